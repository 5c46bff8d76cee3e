#!/usr/bin/env python3
"""Time one entry point (or one stage of the owner-computes backward) of several builds of the
library in one process (interleaved rounds), e.g. ablation builds under opus_amd/csrc/abl/.
    python tools/ablate.py VARIANT fwd|owner|atomic|bin|dfeat|gather lib1.so lib2.so ...
Set POINTS=cameras for camera-projected locations instead of uniform ones."""
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
if os.environ.get("POINTS") == "cameras":
    import numpy as np
    from tools.synthetic_cameras import synthetic_occ2img
    from opus_amd import opus_sampling
    gen = torch.Generator(device="cuda").manual_seed(1)
    lo, hi = torch.tensor([-40.0, -40.0, -1.0], device="cuda"), torch.tensor([40.0, 40.0, 5.4], device="cuda")
    pts = torch.rand((1, Q, 8, 4, P, 3), generator=gen, device="cuda") * (hi - lo) + lo
    sw = torch.softmax(torch.randn((1, Q, 4, 8, P, 4), generator=gen, device="cuda"), -1)
    occ = torch.from_numpy(synthetic_occ2img(np.random.default_rng(1), 1, 8, 6, 256, 704)).cuda()
    opus_sampling.project_pack_into(pts, sw, occ, 256, 704, 6, loc, w)
STAGES = {"bin": _lib.BWD_STAGE_BIN, "dfeat": _lib.BWD_STAGE_DFEAT, "gather": _lib.BWD_STAGE_GATHER}
dfeats = [torch.empty_like(f) for f in feats]
dloc, dw = torch.empty_like(loc), torch.empty_like(w)
ws = torch.empty(max(wrapper.backward_workspace_bytes(feats, loc, w, _lib.BWD_OWNER), 16), dtype=torch.uint8, device="cuda")
handles = {}
for path in libs:
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    handles[path] = _lib.load()
times = {p: [] for p in libs}
for rnd in range(6):
    for path in libs:
        _lib._lib = handles[path]
        if what in ("dfeat", "gather"):
            wrapper.msmv_backward_into(g, feats, loc, w, dfeats, dloc, dw, ws, flags=_lib.BWD_OWNER | _lib.BWD_STAGE_BIN)
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            if what == "fwd":
                wrapper.msmv_forward(feats, loc, w)
            elif what in STAGES:
                wrapper.msmv_backward_into(g, feats, loc, w, dfeats, dloc, dw, ws, flags=_lib.BWD_OWNER | STAGES[what])
            else:
                wrapper.msmv_backward(g, feats, loc, w, algo=what)
            b.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[path].append(a.elapsed_time(b) * 1e3)
for path in libs:
    t = sorted(times[path])
    print(f"{os.path.basename(path):28s} median {t[len(t) // 2]:9.1f} us   min {t[0]:9.1f} us")
