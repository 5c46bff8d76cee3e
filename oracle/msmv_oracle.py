"""numpy front-end of the C oracle (``oracle/msmv_oracle.c``).  TEST INFRASTRUCTURE ONLY.

The C file restates the reference device kernels
(models/csrc/msmv_sampling/msmv_sampling_forward.cu:27-164,
models/csrc/msmv_sampling/msmv_sampling_backward.cu:29-224); this module only marshals
numpy arrays into it.  Feature maps are channel-last ``[B, N, H, W, C]`` exactly as the
reference's extension takes them (models/csrc/msmv_sampling/msmv_sampling.cpp:98-104).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmsmv_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile the oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "msmv_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libmsmv_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.msmv_oracle_forward.restype = None
        _lib.msmv_oracle_backward.restype = None
        _lib.msmv_oracle_unique_pixels.restype = ctypes.c_long
    return _lib


def _as_f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _ptr(a):
    return a.ctypes.data_as(_f32p)


def _ptr_array(arrs):
    return (_f32p * len(arrs))(*[_ptr(a) for a in arrs])


def _dims(feats, loc, w):
    L = len(feats)
    B, N, _, _, C = feats[0].shape
    _, Q, P, three = loc.shape
    assert three == 3 and w.shape == (B, Q, P, L), (loc.shape, w.shape, L)
    for f in feats:
        assert f.shape[0] == B and f.shape[1] == N and f.shape[4] == C
    hw = np.array([d for f in feats for d in f.shape[2:4]], dtype=np.int32)
    return L, B, N, C, Q, P, hw


def forward(feats, loc, w):
    """feats: list of [B,N,H,W,C]; loc [B,Q,P,3]; w [B,Q,P,L] -> out [B,Q,C,P]."""
    feats = [_as_f32(f) for f in feats]
    loc, w = _as_f32(loc), _as_f32(w)
    L, B, N, C, Q, P, hw = _dims(feats, loc, w)
    out = np.empty((B, Q, C, P), dtype=np.float32)
    lib().msmv_oracle_forward(
        ctypes.c_int(L), _ptr_array(feats), hw.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
        _ptr(loc), _ptr(w), B, C, N, Q, P, _ptr(out))
    return out


def backward(grad_out, feats, loc, w):
    """-> (dfeats list [B,N,H,W,C], dloc [B,Q,P,3] (channel 2 == 0), dw [B,Q,P,L])."""
    feats = [_as_f32(f) for f in feats]
    loc, w, grad_out = _as_f32(loc), _as_f32(w), _as_f32(grad_out)
    L, B, N, C, Q, P, hw = _dims(feats, loc, w)
    assert grad_out.shape == (B, Q, C, P)
    dfeats = [np.empty_like(f) for f in feats]
    dloc = np.empty((B, Q, P, 3), dtype=np.float32)
    dw = np.empty((B, Q, P, L), dtype=np.float32)
    lib().msmv_oracle_backward(
        ctypes.c_int(L), _ptr(grad_out), _ptr_array(feats),
        hw.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
        _ptr(loc), _ptr(w), B, C, N, Q, P, _ptr_array(dfeats), _ptr(dloc), _ptr(dw))
    return dfeats, dloc, dw


def unique_pixels(hw_list, loc, B, N):
    """Exact number of distinct feature pixels the locations touch (all levels)."""
    loc = _as_f32(loc)
    _, Q, P, _ = loc.shape
    hw = np.array([d for pair in hw_list for d in pair], dtype=np.int32)
    total = sum(B * N * h * w for h, w in hw_list)
    bitmap = np.empty(total, dtype=np.uint8)
    return int(lib().msmv_oracle_unique_pixels(
        ctypes.c_int(len(hw_list)), hw.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
        _ptr(loc), B, N, Q, P, bitmap.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))))
