"""ctypes binding of libopus_msmv.so (declared in include/opus_msmv.h).

This replaces the reference's pybind11 module ``_msmv_sampling_cuda``
(models/csrc/msmv_sampling/msmv_sampling.cpp:362-368): the same four operations, taken
through a plain C ABI with raw device pointers, sizes and the current HIP stream.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OPUS_MSMV_LIB points at another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("OPUS_MSMV_LIB") or os.path.join(_HERE, "csrc", "libopus_msmv.so")

OK = 0
BWD_AUTO, BWD_ATOMIC, BWD_OWNER = 0, 1, 2
BWD_ACCUMULATE = 0x10
FWD_GENERIC = 0x1
PACK_SHARED_FRAMES = 0x1
BWD_STAGE_BIN, BWD_STAGE_DFEAT, BWD_STAGE_GATHER = 0x100, 0x200, 0x400
BWD_PRESORTED = 0x800
FWD_PRESORTED = 0x4
MAX_LEVELS = 8
MAX_POINTS = 32
NOALLOC_SCRATCH_BYTES = 32 << 20
ESCRATCH = -9
EINVAL_FLAGS = -7        # the call's form does not apply to these shapes / pointers (or unknown flag bits)

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); every symbol include/opus_msmv.h declares
SYMBOLS = {
    "opus_msmv_version": (ctypes.c_int, []),
    "opus_msmv_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "opus_msmv_forward": (ctypes.c_int, [
        ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), _i32p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_void_p]),
    "opus_msmv_forward_ex": (ctypes.c_int, [
        ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), _i32p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "opus_msmv_backward_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 7),
    "opus_msmv_backward": (ctypes.c_int, [
        ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), _i32p,
        ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "opus_msmv_backward_noalloc": (ctypes.c_int, [
        ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), _i32p,
        ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_void_p]),
    "opus_msmv_forward_rows": (ctypes.c_int, [
        ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), _i32p, ctypes.c_void_p, ctypes.c_void_p]
        + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "opus_msmv_backward_rows": (ctypes.c_int, [
        ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), _i32p,
        ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 7
        + [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p,
           ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "opus_msmv_forward_sorted": (ctypes.c_int, [
        ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), _i32p, ctypes.c_void_p, ctypes.c_void_p]
        + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "opus_msmv_dfeat_max_sets": (ctypes.c_int, []),
    "opus_msmv_dfeat_multi": (ctypes.c_int, [
        ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
        ctypes.c_size_t, _i32p] + [ctypes.c_int] * 7
        + [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p]),
    "opus_msmv_project_pack": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 7
        + [ctypes.c_float] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "opus_msmv_project_pack_backward": (ctypes.c_int, [
        ctypes.c_void_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_float] * 3
        + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "opus_msmv_relayout": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 8
                           + [ctypes.c_void_p]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C opus_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`)")
        # torch first: it loads the HIP runtime bundled in its wheel by path.  Were this library
        # loaded before it, the system's copy of libamdhip64 would come in with it, torch would
        # add its own, and kernels registered with one runtime would be launched on streams of the
        # other ("HIP kernel launch failed").  With torch's copy resident, the soname resolves to it.
        import torch  # noqa: F401
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:            # missing libamdhip64 etc.
            raise ImportError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (restype, argtypes) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def strerror(code):
    return load().opus_msmv_strerror(code).decode()


def check(code, what):
    if code != OK:
        raise RuntimeError(f"{what}: {strerror(code)} (code {code})")


def ptr_array(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


def int_array(vals):
    return (ctypes.c_int * len(vals))(*vals)
