#!/usr/bin/env python3
"""Probe: the backward's d-feat pass and its d loc / d w gather need only the sort and write
disjoint outputs - how long do they take one after the other on one stream, and side by side on
two streams?
    python tools/overlap_probe.py [L|M|S|T] [order]      order: dg (d feat first) | gd
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opus_amd import _lib  # noqa: E402
from opus_amd.csrc import wrapper  # noqa: E402
from tools.devbench import VARIANTS, make_inputs  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "L"
order = sys.argv[2] if len(sys.argv) > 2 else "dg"
Q, P = VARIANTS[variant]
feats, loc, w, g = make_inputs(32, Q, P)
dfeats = [torch.empty_like(f) for f in feats]
dloc, dw = torch.empty_like(loc), torch.empty_like(w)
ws = torch.empty(wrapper.backward_workspace_bytes(feats, loc, w, _lib.BWD_OWNER), dtype=torch.uint8, device="cuda")
args = (g, feats, loc, w, dfeats, dloc, dw, ws)
O = _lib.BWD_OWNER
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def stage(bit):
    wrapper.msmv_backward_into(*args, flags=O | bit)


def serial():
    stage(_lib.BWD_STAGE_BIN)
    stage(_lib.BWD_STAGE_DFEAT)
    stage(_lib.BWD_STAGE_GATHER)


def forked():
    stage(_lib.BWD_STAGE_BIN)
    side.wait_stream(main)
    first, second = (_lib.BWD_STAGE_DFEAT, _lib.BWD_STAGE_GATHER) if order == "dg" else (_lib.BWD_STAGE_GATHER, _lib.BWD_STAGE_DFEAT)
    stage(first)
    with torch.cuda.stream(side):
        stage(second)
    main.wait_stream(side)


def timeit(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def step(order):
    """A whole step as the benchmark runs it: forward, sort, then the two passes in `order`."""
    wrapper.msmv_forward(feats, loc, w)
    stage(_lib.BWD_STAGE_BIN)
    for bit in order:
        stage(bit)


for _ in range(200):
    serial()
torch.cuda.synchronize()
print("sort only           %7.1f us" % timeit(lambda: stage(_lib.BWD_STAGE_BIN)))
print("d feat only         %7.1f us" % timeit(lambda: stage(_lib.BWD_STAGE_DFEAT)))
print("gather only         %7.1f us" % timeit(lambda: stage(_lib.BWD_STAGE_GATHER)))
print("serial              %7.1f us" % timeit(serial))
print("two streams (%s)    %7.1f us" % (order, timeit(forked)))
D, G = _lib.BWD_STAGE_DFEAT, _lib.BWD_STAGE_GATHER
print("step: forward, sort, d feat, gather   %7.1f us" % timeit(lambda: step((D, G))))
print("step: forward, sort, gather, d feat   %7.1f us" % timeit(lambda: step((G, D))))
print("step: forward, sort, d feat, gather   %7.1f us" % timeit(lambda: step((D, G))))
print("step: forward, sort, gather, d feat   %7.1f us" % timeit(lambda: step((G, D))))
