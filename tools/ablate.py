#!/usr/bin/env python3
"""Time the backward (owner) entry point of several builds of the library in one process
(interleaved rounds), e.g. ablation builds under opus_amd/csrc/abl/.
    python tools/ablate.py [variant] lib1.so lib2.so ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opus_amd import _lib  # noqa: E402
from opus_amd.csrc import wrapper  # noqa: E402
from tools.devbench import VARIANTS, make_inputs  # noqa: E402

variant = sys.argv[1]
what = sys.argv[2]
libs = sys.argv[3:]
Q, P = VARIANTS[variant]
feats, loc, w, g = make_inputs(32, Q, P)
handles = {}
for path in libs:
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    handles[path] = _lib.load()
times = {p: [] for p in libs}
for rnd in range(6):
    for path in libs:
        _lib._lib = handles[path]
        for _ in range(2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            if what == "fwd":
                wrapper.msmv_forward(feats, loc, w)
            else:
                wrapper.msmv_backward(g, feats, loc, w, algo=what)
            b.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[path].append(a.elapsed_time(b) * 1e3)
for path in libs:
    t = sorted(times[path])
    print(f"{os.path.basename(path):28s} median {t[len(t) // 2]:9.1f} us   min {t[0]:9.1f} us")
