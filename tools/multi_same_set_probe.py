#!/usr/bin/env python3
"""What bounds the multi-set d-feat pass: the hits it applies or the gradient rows it fetches?
Times opus_msmv_dfeat_multi over six DIFFERENT sorted point sets against the same call over six
copies of ONE set (the same hits to apply, but one layer's gradient rows and entries: they stay in
an XCD's L2 as in the single-set kernel), in the op's layout (512-byte rows at P = 2) and in row mode.
    python tools/multi_same_set_probe.py [L|M|T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opus_amd import _lib  # noqa: E402
from opus_amd.csrc import wrapper  # noqa: E402
from tools.devbench import VARIANTS, make_inputs  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "L"
Q, P = VARIANTS[variant]
NSETS = 6
sets = []
feats = None
for i in range(NSETS):
    f, loc, w, g = make_inputs(32, Q, P, seed=100 + i)
    feats = feats or f
    sets.append((loc, w, g))
dfeats = [torch.empty_like(f) for f in feats]
for rows in (None, (8, 4)):
    wss, gs = [], []
    for loc, w, g in sets:
        ws = torch.empty(wrapper.backward_workspace_bytes(feats, loc, w, _lib.BWD_OWNER), dtype=torch.uint8, device="cuda")
        gg = g if rows is None else g.reshape(1, Q, 4, 8 * P, 64).contiguous()
        wrapper.msmv_backward_into(gg, feats, loc, w, None, torch.empty_like(loc), torch.empty_like(w), ws,
                                   flags=_lib.BWD_OWNER | _lib.BWD_STAGE_BIN | _lib.BWD_STAGE_GATHER, rows=rows)
        wss.append(ws)
        gs.append(gg)
    for name, gl, wl in (("six different sets", gs, wss), ("one set six times", [gs[0]] * NSETS, [wss[0]] * NSETS),
                         ("one set once", gs[:1], wss[:1])):
        times = []
        for i in range(25):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            wrapper.msmv_dfeat_multi(gl, wl, feats, Q, P, dfeats, rows=rows)
            b.record()
            torch.cuda.synchronize()
            if i >= 5:
                times.append(a.elapsed_time(b) * 1e3)
        times.sort()
        print(f"{'op layout' if rows is None else 'row mode '}  {name:20s} median {times[len(times) // 2]:7.1f} us", flush=True)
