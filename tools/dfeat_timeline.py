#!/usr/bin/env python3
"""Per-wave phase timeline of the d-feat kernel from a -DMSMV_DBG_TIMES build of the library
(opus_amd/csrc/abl/build.sh dbg "-DMSMV_DBG_TIMES"):
    OPUS_MSMV_LIB=opus_amd/csrc/abl/lib_dbg.so python tools/dfeat_timeline.py [L|M|S|T] [uniform|cameras]
Timestamps are the 100 MHz constant clock: start, bins known, first entries in, hit loop done,
partial tiles summed, stores drained."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opus_amd import _lib  # noqa: E402
from opus_amd.csrc import wrapper  # noqa: E402
from tools.devbench import VARIANTS, make_inputs  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "L"
Q, P = VARIANTS[variant]
feats, loc, w, g = make_inputs(32, Q, P)
lib = _lib.load()
setbuf = lib.opus_msmv_debug_set_buffer
setbuf.argtypes = [ctypes.c_void_p]
dfeats = [torch.empty_like(f) for f in feats]
dloc, dw = torch.empty_like(loc), torch.empty_like(w)
ws = torch.empty(wrapper.backward_workspace_bytes(feats, loc, w, _lib.BWD_OWNER), dtype=torch.uint8, device="cuda")
wrapper.msmv_backward_into(g, feats, loc, w, dfeats, dloc, dw, ws, flags=_lib.BWD_OWNER | _lib.BWD_STAGE_BIN)
NW = 4 * 32 * 6 * (88 + 22 + 12 + 9) + 4096
buf = torch.zeros((NW, 8), dtype=torch.int64, device="cuda")
assert setbuf(buf.data_ptr()) == 0
for _ in range(5):
    wrapper.msmv_backward_into(g, feats, loc, w, dfeats, dloc, dw, ws, flags=_lib.BWD_OWNER | _lib.BWD_STAGE_DFEAT)
torch.cuda.synchronize()
buf.zero_()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
wrapper.msmv_backward_into(g, feats, loc, w, dfeats, dloc, dw, ws, flags=_lib.BWD_OWNER | _lib.BWD_STAGE_DFEAT)
b.record()
torch.cuda.synchronize()
print("kernel (event) us:", a.elapsed_time(b) * 1e3)
t = buf.cpu().numpy()
t = t[t[:, 0] != 0]
lvl = ((t[:, 6] >> 32) & 0xFF).astype(int)
where = (t[:, 6] >> 40).astype(int)          # xcc << 16 | HW_ID
xcc = (where >> 16) & 0xF
cu_key = (xcc << 8) | ((where >> 8) & 0xFF)  # XCC, SE, SH, CU
simd_key = (cu_key << 2) | ((where >> 4) & 3)
total = (t[:, 6] & 0xFFFFFFFF).astype(int)
bslab = (t[:, 7] >> 32).astype(int)
sub = (t[:, 7] & 0xFFFF).astype(int)
t0 = t[:, 0].min()
T = (t[:, :6] - t0) * 0.01          # us
end = np.where(t[:, 5] != 0, T[:, 5], T[:, 4])
print("waves", len(t), "span us", end.max())
for l in range(4):
    m = lvl == l
    own = m & (t[:, 5] != 0)
    print(f"level {l}: waves {m.sum():6d} entries/wave {total[m].mean():7.1f}  start {T[m, 0].min():7.1f}..{T[m, 0].max():7.1f}  "
          f"end {end[m].min():7.1f}..{end[m].max():7.1f}")
    d = lambda i, j, mm=m: (T[mm, j] - T[mm, i])
    print(f"   header+bins {d(0, 1).mean():6.2f}  first entries {np.maximum(d(1, 2), 0).mean():6.2f}  hit loop {d(2, 3)[t[m, 2] != 0].mean() if (t[m, 2] != 0).any() else 0:6.2f}  "
          f"tile sum {d(3, 4).mean():6.2f}  store+drain {(T[own, 5] - T[own, 4]).mean():6.2f}  life {(end[m] - T[m, 0]).mean():6.2f} (max {(end[m] - T[m, 0]).max():6.2f})")
# concurrency over time
edges = np.arange(0, end.max() + 5, 5.0)
print("time(us)  resident waves by level [L0 L1 L2 L3]  | waves in store+drain")
for lo in edges:
    hi = lo + 5
    res = [(((T[:, 0] < hi) & (end > lo)) & (lvl == l)).sum() for l in range(4)]
    drain = ((t[:, 5] != 0) & (T[:, 4] < hi) & (T[:, 5] > lo)).sum()
    print(f"{lo:6.0f}   {res}   | {drain}")

# instantaneous residency: waves alive at sample times, chip-wide and per CU / SIMD
print("CUs seen", len(np.unique(cu_key)), "SIMDs seen", len(np.unique(simd_key)))
print("t(us)  resident  [L0 L1 L2 L3]  per-CU min/median/max  per-SIMD max  CUs with < 8 waves")
for ts in np.arange(2.5, end.max(), 5.0):
    alive = (T[:, 0] <= ts) & (end > ts)
    per_cu = np.bincount(np.unique(cu_key, return_inverse=True)[1][alive], minlength=len(np.unique(cu_key)))
    per_simd = np.bincount(np.unique(simd_key, return_inverse=True)[1][alive], minlength=len(np.unique(simd_key)))
    print(f"{ts:6.1f} {alive.sum():7d}  {[int((alive & (lvl == l)).sum()) for l in range(4)]}  "
          f"{per_cu.min()}/{int(np.median(per_cu))}/{per_cu.max()}  {per_simd.max()}  {(per_cu < 8).sum()}")
# per XCC: when its first wave starts, when its last ends, and how many waves it ran
print("XCC  waves  first start  last end")
for x in np.unique(xcc):
    m = xcc == x
    print(f"{x:3d} {m.sum():7d} {T[m, 0].min():8.1f} {end[m].max():8.1f}")
