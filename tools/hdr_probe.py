import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/opus_amd") else os.getcwd())
from opus_amd import _lib
from opus_amd.csrc import wrapper
from tools.devbench import VARIANTS, make_inputs
Q, P = VARIANTS["L"]
feats, loc, w, g = make_inputs(32, Q, P)
lib = _lib.load()
setbuf = lib.opus_msmv_debug_set_buffer; setbuf.argtypes = [ctypes.c_void_p]
dfeats = [torch.empty_like(f) for f in feats]
dloc, dw = torch.empty_like(loc), torch.empty_like(w)
ws = torch.empty(wrapper.backward_workspace_bytes(feats, loc, w, _lib.BWD_OWNER), dtype=torch.uint8, device="cuda")
wrapper.msmv_backward_into(g, feats, loc, w, dfeats, dloc, dw, ws, flags=_lib.BWD_OWNER | _lib.BWD_STAGE_BIN)
NW = 4 * 32 * 6 * (88 + 22 + 12 + 9) + 4096
buf = torch.zeros((NW, 8), dtype=torch.int64, device="cuda")
assert setbuf(buf.data_ptr()) == 0
for _ in range(5):
    wrapper.msmv_backward_into(g, feats, loc, w, dfeats, dloc, dw, ws, flags=_lib.BWD_OWNER | _lib.BWD_STAGE_DFEAT)
torch.cuda.synchronize(); buf.zero_()
wrapper.msmv_backward_into(g, feats, loc, w, dfeats, dloc, dw, ws, flags=_lib.BWD_OWNER | _lib.BWD_STAGE_DFEAT)
torch.cuda.synchronize()
t = buf.cpu().numpy(); t = t[t[:, 0] != 0]
lvl = ((t[:, 6] >> 32) & 0xFF).astype(int)
T = (t[:, :6] - t[:, :1]) * 0.01
for l in range(4):
    m = (lvl == l) & (t[:, 5] != 0)
    print("level", l, "waves", m.sum(), "level known %.2f  geometry in %.2f  bins known %.2f  first entries %.2f  end %.2f" % tuple(T[m, k].mean() for k in (1, 2, 3, 4, 5)))
