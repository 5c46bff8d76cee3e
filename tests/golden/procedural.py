"""Exact procedural tensors shared by the fixture generator and the tests."""
import numpy as np


def procedural_feats(shape, salt):
    """Exact, platform-independent pseudo-random features in [-1, 1): integer hash only."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    x = i + np.uint64((int(salt) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)   # wraps mod 2^64
    x ^= x >> np.uint64(33)
    x = x * np.uint64(0xFF51AFD7ED558CCD)
    x ^= x >> np.uint64(33)
    x = x * np.uint64(0xC4CEB9FE1A85EC53)
    x ^= x >> np.uint64(33)
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)   # 24 bits -> exact in f32
    return (u * 2.0 - 1.0).astype(np.float32).reshape(shape)


# the camera rig lives with the benchmark tooling; the fixture generator and the tests import it here
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
from tools.synthetic_cameras import synthetic_occ2img  # noqa: E402,F401
