#!/usr/bin/env python3
"""Run a few calls of one entry point (for rocprofv3 runs).
    python3 tools/run_once.py fwd|bwd_owner|bwd_atomic [variant] [batch] [calls]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opus_amd.csrc import wrapper  # noqa: E402
from tools.devbench import VARIANTS, make_inputs  # noqa: E402

what = sys.argv[1]
variant = sys.argv[2] if len(sys.argv) > 2 else "L"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 3
Q, P = VARIANTS[variant]
feats, loc, w, g = make_inputs(32 * batch, Q, P)
for _ in range(calls):
    if what == "fwd":
        wrapper.msmv_forward(feats, loc, w)
    else:
        wrapper.msmv_backward(g, feats, loc, w, algo=what.split("_")[1])
torch.cuda.synchronize()
