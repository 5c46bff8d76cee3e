#!/usr/bin/env python3
"""Probe: how much does the d-feat kernel's duration depend on what ran just before it?
Times the d-feat stage alone (HIP events around it) after different predecessors.
    python tools/prior_state_probe.py [L|M|S|T]
"""
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
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()      # 1 GiB
big2 = torch.empty(1 << 28, dtype=torch.float32, device="cuda")


def stage(bit):
    wrapper.msmv_backward_into(*args, flags=O | bit)


def measure(name, before, reps=30):
    stage(_lib.BWD_STAGE_BIN)
    ts = []
    for i in range(reps + 5):
        before()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        stage(_lib.BWD_STAGE_DFEAT)
        b.record()
        torch.cuda.synchronize()
        if i >= 5:
            ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print(f"{name:58s} d feat {ts[len(ts) // 2]:7.1f} us  (min {ts[0]:.1f})", flush=True)


for _ in range(100):
    stage(_lib.BWD_STAGE_DFEAT)
torch.cuda.synchronize()
measure("nothing (host sync only)", lambda: None)
measure("d feat (back to back)", lambda: stage(_lib.BWD_STAGE_DFEAT))
measure("forward + sort", lambda: (wrapper.msmv_forward(feats, loc, w), stage(_lib.BWD_STAGE_BIN)))
measure("sort", lambda: stage(_lib.BWD_STAGE_BIN))
measure("gather", lambda: stage(_lib.BWD_STAGE_GATHER))
measure("forward + sort + gather", lambda: (wrapper.msmv_forward(feats, loc, w), stage(_lib.BWD_STAGE_BIN), stage(_lib.BWD_STAGE_GATHER)))
measure("idle 300 us (spin kernel)", lambda: torch.cuda._sleep(600000))
measure("1 GiB read of another buffer (sum)", lambda: big.sum())
measure("1 GiB write of another buffer (fill)", lambda: big2.fill_(1.0))
measure("1 GiB copy", lambda: big2.copy_(big))
measure("read of grad_out (sum)", lambda: g.sum())
measure("forward + sort + read of grad_out", lambda: (wrapper.msmv_forward(feats, loc, w), stage(_lib.BWD_STAGE_BIN), g.sum()))
measure("fill of the d-feat buffers (zero_)", lambda: [d.zero_() for d in dfeats])
