#!/usr/bin/env python3
"""Time the op's backward when only one group of gradients is asked for (OPUS-L shape, one layer):
    python tools/partial_grad_probe.py
all three groups / d loc + d w only (frozen feature maps) / d feats only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opus_amd.csrc import wrapper  # noqa: E402
from tools.devbench import VARIANTS, make_inputs  # noqa: E402

Q, P = VARIANTS[sys.argv[1] if len(sys.argv) > 1 else "L"]
feats, loc, w, g = make_inputs(32, Q, P)
for need in ((True, True), (False, True), (True, False)):
    times = []
    for i in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        wrapper.msmv_backward(g, feats, loc, w, need=need)
        b.record()
        torch.cuda.synchronize()
        if i >= 10:
            times.append(a.elapsed_time(b) * 1e3)
    times.sort()
    print(f"need d feats={need[0]!s:5} d loc/d w={need[1]!s:5}  median {times[len(times) // 2]:7.1f} us", flush=True)
