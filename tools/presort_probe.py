#!/usr/bin/env python3
"""Probe: the backward's sort reads only loc / weights - how long do forward + whole backward take
with the sort issued on a second stream beside the forward (the backward then runs PRESORTED), against
everything on one stream?
    python tools/presort_probe.py [L|M|S|T] [uniform|cameras]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opus_amd import _lib  # noqa: E402
from opus_amd.csrc import wrapper  # noqa: E402
from tools.devbench import VARIANTS, make_inputs  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "L"
Q, P = VARIANTS[variant]
feats, loc, w, g = make_inputs(32, Q, P)
dfeats = [torch.empty_like(f) for f in feats]
dloc, dw = torch.empty_like(loc), torch.empty_like(w)
ws = torch.empty(wrapper.backward_workspace_bytes(feats, loc, w, _lib.BWD_OWNER), dtype=torch.uint8, device="cuda")
args = (g, feats, loc, w, dfeats, dloc, dw, ws)
O = _lib.BWD_OWNER
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def serial():
    wrapper.msmv_forward(feats, loc, w)
    wrapper.msmv_backward_into(*args, flags=O)


def forked():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        wrapper.msmv_backward_into(*args, flags=O | _lib.BWD_STAGE_BIN)
    wrapper.msmv_forward(feats, loc, w)
    main.wait_stream(side)
    wrapper.msmv_backward_into(*args, flags=O | _lib.BWD_PRESORTED)


def timeit(fn, iters=60):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for rnd in range(3):
    print(f"OPUS-{variant}: one stream {timeit(serial):7.1f} us   sort beside the forward {timeit(forked):7.1f} us", flush=True)
