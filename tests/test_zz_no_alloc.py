"""opus_msmv_backward_noalloc (the backward whose sort scratch is a static device array of the
library) against opus_msmv_backward with a caller-owned workspace, through the C ABI, on two
calls in a row - the second on a smaller input, so that it runs on whatever the first left in
the shared array.  Bar: gradients 1e-4 relative to each tensor's max magnitude
(helpers.GRAD_RTOL); both entry points run the same kernels, only the order inside a sort bin
may differ between runs."""
import numpy as np
import pytest
import torch

from helpers import GRAD_RTOL, REAL_LEVELS, assert_close, random_case

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _both(lib_mod, wr, case, flags, ref_flags=None):
    """(reference entry point, no-alloc entry point) -> two (dfeats, dloc, dw) tuples, as numpy."""
    levels, B, N, C, Q, P, seed = case
    feats, loc, w, gout = random_case(seed, levels, B, N, C, Q, P, oob_frac=0.25)
    f = [_dev(x) for x in feats]
    l, ww, g = _dev(loc), _dev(w), _dev(gout)
    lib = lib_mod.load()
    L = len(levels)
    hw = lib_mod.int_array([d for h_w in levels for d in h_w])
    fptrs = lib_mod.ptr_array([x.data_ptr() for x in f])
    stream = torch.cuda.current_stream().cuda_stream

    def fresh():
        # NaN-filled: every element must be written by the call
        return ([torch.full_like(x, float("nan")) for x in f], torch.full_like(l, float("nan")),
                torch.full_like(ww, float("nan")))

    # existing entry point, workspace from the caller
    df0, dl0, dw0 = fresh()
    need = lib.opus_msmv_backward_workspace_bytes(L, B, C, N, Q, P, flags)
    ref_flags = flags if ref_flags is None else ref_flags
    ws = torch.full((max(lib.opus_msmv_backward_workspace_bytes(L, B, C, N, Q, P, ref_flags), 16),), 0xFF,
                    dtype=torch.uint8, device="cuda")
    rc = lib.opus_msmv_backward(L, g.data_ptr(), fptrs, hw, l.data_ptr(), ww.data_ptr(), B, C, N, Q, P,
                                lib_mod.ptr_array([x.data_ptr() for x in df0]), dl0.data_ptr(), dw0.data_ptr(),
                                ws.data_ptr(), ws.numel(), ref_flags, stream)
    lib_mod.check(rc, "opus_msmv_backward")

    # the entry point that allocates nothing: torch's allocator must see no request during the call
    df1, dl1, dw1 = fresh()
    dfp = lib_mod.ptr_array([x.data_ptr() for x in df1])
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    free_before, _ = torch.cuda.mem_get_info()
    rc = lib.opus_msmv_backward_noalloc(L, g.data_ptr(), fptrs, hw, l.data_ptr(), ww.data_ptr(), B, C, N, Q, P,
                                        dfp, dl1.data_ptr(), dw1.data_ptr(), flags, stream)
    lib_mod.check(rc, "opus_msmv_backward_noalloc")
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == before
    free_after, _ = torch.cuda.mem_get_info()
    to_np = lambda t: (([x.cpu().numpy() for x in t[0]]), t[1].cpu().numpy(), t[2].cpu().numpy())   # noqa: E731
    return to_np((df0, dl0, dw0)), to_np((df1, dl1, dw1)), need, free_before - free_after


def _compare(ref, got, what):
    for i, (a, b) in enumerate(zip(got[0], ref[0])):
        assert_close(a, b, GRAD_RTOL, f"{what} dfeat{i}")
    assert_close(got[1], ref[1], GRAD_RTOL, f"{what} dloc")
    assert_close(got[2], ref[2], GRAD_RTOL, f"{what} dw")
    assert (got[1][..., 2] == 0).all()


def test_noalloc_backward_twice_second_call_smaller():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from opus_amd import _lib as lib_mod
    from opus_amd.csrc import wrapper as wr
    assert wr.MSMV_CUDA
    # first: slabs of 4000 points (two-pass sort: global counters that must start at zero);
    # second: fewer slabs of 300 points (one-workgroup sort) on the array the first call used
    big = (REAL_LEVELS, 3, 6, 64, 2000, 2, 101)
    small = (REAL_LEVELS, 2, 6, 64, 150, 2, 202)
    grown = []
    for name, case in (("first call", big), ("second call (smaller)", small), ("third call (first size again)", big)):
        ref, got, need, grew = _both(lib_mod, wr, case, lib_mod.BWD_OWNER)
        assert 0 < need <= lib_mod.NOALLOC_SCRATCH_BYTES
        _compare(ref, got, name)
        grown.append(grew)
    # device memory in use may grow once (the first call loads the code object that holds the
    # array); later calls leave it where it was
    assert grown[1] <= 0 and grown[2] <= 0, grown


def test_noalloc_backward_other_paths():
    from opus_amd import _lib as lib_mod
    from opus_amd.csrc import wrapper as wr
    # the atomic algorithm and a shape the owner algorithm does not take (C = 8) need no scratch
    for case, flags in (((REAL_LEVELS, 2, 6, 64, 40, 2, 7), lib_mod.BWD_ATOMIC),
                        ((REAL_LEVELS, 2, 3, 8, 40, 3, 8), lib_mod.BWD_AUTO),
                        ((REAL_LEVELS + [(4, 11)], 2, 6, 64, 700, 4, 9), lib_mod.BWD_AUTO)):
        ref, got, _, _ = _both(lib_mod, wr, case, flags)
        _compare(ref, got, f"flags {flags}")


def test_noalloc_backward_rejects_sizes_beyond_the_static_buffer():
    from opus_amd import _lib as lib_mod
    lib = lib_mod.load()
    L, B, C, N, Q, P = 4, 128, 64, 6, 4800, 2            # 1.2 M points: 39 MB of sort entries
    assert lib.opus_msmv_backward_workspace_bytes(L, B, C, N, Q, P, 0) > lib_mod.NOALLOC_SCRATCH_BYTES
    hw = lib_mod.int_array([d for h_w in REAL_LEVELS for d in h_w])
    # the owner algorithm asked for by name: rejected before any pointer is looked at or anything is launched
    rc = lib.opus_msmv_backward_noalloc(L, None, None, hw, None, None, B, C, N, Q, P, None, None, None,
                                        lib_mod.BWD_OWNER, None)
    assert rc == lib_mod.ESCRATCH
    assert "scratch" in lib_mod.strerror(rc)


def test_noalloc_backward_auto_falls_back_to_the_atomic_algorithm_beyond_the_buffer():
    """With the choice of algorithm left to the library, sizes whose sort does not fit the static
    array run the atomic algorithm (which needs no scratch) instead of failing."""
    from opus_amd import _lib as lib_mod
    from opus_amd.csrc import wrapper as wr
    from helpers import TINY_LEVELS
    case = (TINY_LEVELS, 64, 6, 64, 8400, 2, 303)            # 1.08 M points: 34 MB of sort entries
    ref, got, need, _ = _both(lib_mod, wr, case, lib_mod.BWD_AUTO, ref_flags=lib_mod.BWD_ATOMIC)
    assert need > lib_mod.NOALLOC_SCRATCH_BYTES
    _compare(ref, got, "auto beyond the scratch size")
