"""Identity of the kernel sources a measurement was taken on."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash():
    """sha256 over the HIP sources and headers of the library (opus_amd/csrc/*.hip, *.h and the public
    header), file names included: profiles/traffic.json stores it with every measured configuration and
    bench.py reports a traffic figure only while the sources still hash to it."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "opus_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "opus_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "opus_msmv.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
