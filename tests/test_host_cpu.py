"""CPU-side checks (no GPU): the C-ABI library loads and exports what the header declares,
argument validation returns before any launch, the host mirror of the reference interface
behaves like the reference's, and the caller restatement reproduces the golden packing."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import FWD_RTOL, assert_close, golden_feats, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from opus_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def header_functions():
    text = open(os.path.join(ROOT, "include", "opus_msmv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(opus_msmv_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    from opus_amd import _lib
    declared = header_functions()
    assert declared == sorted(_lib.SYMBOLS), "ctypes table and header disagree"
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.opus_msmv_version() == 200
    assert lib.opus_msmv_strerror(0) == b"ok"
    assert b"num_point exceed limits" in lib.opus_msmv_strerror(-2)
    assert lib.opus_msmv_strerror(-99) == b"unknown error code"


def test_argument_validation_returns_before_any_launch(lib):
    from opus_amd import _lib
    null = ctypes.c_void_p(0)
    hw = _lib.int_array([4, 4])
    ptrs = _lib.ptr_array([0])
    fwd = lib.opus_msmv_forward
    assert fwd(0, ptrs, hw, null, null, 1, 8, 6, 1, 1, null, null) == -1       # levels
    assert fwd(9, ptrs, hw, null, null, 1, 8, 6, 1, 1, null, null) == -1
    assert fwd(1, ptrs, hw, null, null, 1, 8, 6, 1, 33, null, null) == -2      # P > 32
    assert fwd(1, ptrs, hw, null, null, 1, 8, 6, 1, 0, null, null) == -2
    assert fwd(1, ptrs, hw, null, null, 1, 0, 6, 1, 1, null, null) == -3       # C == 0
    assert fwd(1, ptrs, hw, null, null, -1, 8, 6, 1, 1, null, null) == -3
    assert fwd(1, ptrs, _lib.int_array([0, 4]), null, null, 1, 8, 6, 1, 1, null, null) == -3
    assert fwd(1, ptrs, hw, null, null, 1, 8, 6, 1, 1, null, null) == -4       # null data
    assert fwd(1, ptrs, hw, null, null, 1 << 20, 64, 6, 1 << 10, 32, null, null) == -5
    assert fwd(1, ptrs, hw, null, null, 0, 8, 6, 5, 1, null, null) == 0        # empty batch
    assert fwd(1, ptrs, hw, null, null, 3, 8, 6, 0, 1, null, null) == 0        # no queries
    assert lib.opus_msmv_forward_ex(1, ptrs, hw, null, null, 1, 8, 6, 1, 1, null, 0x80, null) == -7
    bwd = lib.opus_msmv_backward
    args = (1, null, ptrs, hw, null, null, 1, 8, 6, 1, 1, ptrs, null, null, null, 0)
    assert bwd(*args, 3, null) == -7                                           # unknown algorithm
    assert bwd(*args, 0x12, null) == -4                                        # accumulate + owner is legal: next check
    assert bwd(*args, 0x20, null) == -7                                        # unknown flag bit
    assert bwd(*args, 0x101, null) == -7                                       # stage bits need owner
    assert bwd(*args, 1, null) == -4                                           # null data
    # workspace: none for the atomic algorithm, header + 32 B per point for the owner one
    ws = lib.opus_msmv_backward_workspace_bytes
    assert ws(4, 32, 64, 6, 4800, 2, 1) == 0
    assert ws(4, 32, 8, 6, 4800, 2, 0) == 0                                    # C != 64 -> atomic
    n = ws(4, 32, 64, 6, 4800, 2, 0)
    assert ws(4, 32, 64, 6, 4800, 2, 0x12) == n and ws(4, 32, 64, 6, 4800, 2, 0x11) == 0   # with ACCUMULATE
    # row mode: the batch is Bs*T*G; the re-layout and projection entry points validate too
    assert lib.opus_msmv_forward_rows(1, ptrs, hw, null, null, 1, 0, 4, 8, 6, 1, 1, null, 0, null) == -3
    assert lib.opus_msmv_forward_rows(1, ptrs, hw, null, null, 1, 8, 4, 8, 6, 1, 1, null, 0, null) == -4
    assert lib.opus_msmv_forward_rows(1, ptrs, hw, null, null, 0, 8, 4, 8, 6, 5, 1, null, 0, null) == 0
    assert lib.opus_msmv_relayout(null, null, 1, 8, 6, 4, 64, 0, 4, 0, null) == -3
    assert lib.opus_msmv_relayout(null, null, 1, 8, 6, 4, 64, 4, 4, 0, null) == -4
    assert lib.opus_msmv_relayout(null, null, 0, 8, 6, 4, 64, 4, 4, 0, null) == 0
    pp = lib.opus_msmv_project_pack
    assert pp(null, null, null, 1, 4, 8, 4, 2, 4, 1, 256.0, 704.0, 1e-5, 0, null, null, null) == -3   # N < 2
    assert pp(null, null, null, 1, 4, 8, 4, 33, 4, 6, 256.0, 704.0, 1e-5, 0, null, null, null) == -2
    assert pp(null, null, null, 1, 4, 8, 4, 2, 4, 6, 256.0, 704.0, 1e-5, 1, null, null, null) == -4
    assert pp(null, null, null, 1, 4, 8, 4, 2, 4, 6, 256.0, 704.0, 1e-5, 2, null, null, null) == -7   # unknown flag
    assert pp(null, null, null, 1, 0, 8, 4, 2, 4, 6, 256.0, 704.0, 1e-5, 0, null, null, null) == 0
    assert n >= 32 * 4800 * 2 * 32 and n < 32 * 4800 * 2 * 32 + (4 << 20)
    # sort / gather stages alone need no d feats; the deferred d-feat pass validates its sets
    assert bwd(1, null, ptrs, hw, null, null, 1, 64, 6, 1, 1, None, null, null, null, 0, 0x502, null) == -4
    assert bwd(1, null, ptrs, hw, null, null, 1, 64, 6, 1, 1, None, null, null, null, 0, 0x202, null) == -4
    assert bwd(1, null, ptrs, hw, null, null, 0, 64, 6, 1, 1, None, null, null, null, 0, 0x502, null) == 0
    multi = lib.opus_msmv_dfeat_multi
    assert lib.opus_msmv_dfeat_max_sets() == 8
    real = _lib.int_array([64, 176, 32, 88, 16, 44, 8, 22])
    p4 = _lib.ptr_array([0, 0, 0, 0])
    assert multi(4, 0, ptrs, ptrs, 0, real, 32, 0, 0, 64, 6, 10, 2, p4, 0, null) == -3      # no sets
    assert multi(4, 9, ptrs, ptrs, 0, real, 32, 0, 0, 64, 6, 10, 2, p4, 0, null) == -3      # too many
    assert multi(4, 1, ptrs, ptrs, 0, real, 32, 0, 0, 32, 6, 10, 2, p4, 0, null) == -7      # C != 64
    assert multi(4, 1, ptrs, ptrs, 0, real, 32, 0, 0, 64, 6, 10, 2, p4, 1, null) == -7      # unknown flag
    assert multi(4, 1, ptrs, ptrs, 0, real, 32, 0, 0, 64, 6, 10, 2, p4, 0, null) == -4      # null d feats
    assert multi(4, 1, ptrs, ptrs, 0, real, 4, 8, 0, 64, 6, 10, 2, p4, 0, null) == -3       # rows: G < 1
    assert multi(4, 1, ptrs, ptrs, 0, real, 0, 0, 0, 64, 6, 10, 2, p4, 0, null) == 0        # empty batch


def test_wrapper_mirrors_reference_names_and_checks():
    from opus_amd.csrc import wrapper
    for name in ("msmv_sampling", "msmv_sampling_pytorch", "MSMVSamplingC2345",
                 "MSMVSamplingC23456", "MSMV_CUDA"):
        assert hasattr(wrapper, name)
    assert wrapper.MSMV_CUDA is True
    g = load_golden("g1_tiny4")
    feats = [torch.from_numpy(f) for f in golden_feats(g)]
    loc, w = torch.from_numpy(g["loc"]), torch.from_numpy(g["w"])
    # CPU tensors are rejected with the reference's message (msmv_sampling.cpp:113), never
    # routed to a fallback
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        wrapper.msmv_sampling(feats, loc, w)
    with pytest.raises(RuntimeError, match="contiguous"):
        wrapper.msmv_sampling([feats[0].transpose(2, 3)] + feats[1:], loc, w)


def test_msmv_sampling_pytorch_matches_golden():
    """The channel-first grid_sample formulation kept for API parity equals the reference's."""
    from opus_amd.csrc import wrapper
    for name in ("g1_tiny4", "g3_tiny5"):
        g = load_golden(name)
        feats_cf = [torch.from_numpy(f).permute(0, 4, 1, 2, 3).contiguous() for f in golden_feats(g)]
        out = wrapper.msmv_sampling_pytorch(feats_cf, torch.from_numpy(g["loc"]), torch.from_numpy(g["w"]))
        assert_close(out.numpy(), g["out"], 1e-6, f"{name} msmv_sampling_pytorch")


def test_sampling_4d_packing_matches_golden(oracle):
    """project_and_pack_torch / unpack reproduce what the reference's sampling_4d hands to and takes
    from the op (incl. first-valid-camera selection and the (B,T,G) / (B,G,T) packing)."""
    from opus_amd import opus_sampling
    g = load_golden("g5_sampling4d")
    pts = torch.from_numpy(g["sample_points"])
    B, Q, T, G, P, _ = pts.shape
    loc, w = opus_sampling.project_and_pack_torch(
        pts, torch.from_numpy(g["scale_weights"]), torch.from_numpy(g["occ2img"]),
        int(g["image_h"]), int(g["image_w"]), int(g["num_views"]))
    assert loc.shape == g["op_loc"].shape and w.shape == g["op_w"].shape
    # the selected camera must agree exactly; projected coordinates to float rounding
    assert np.array_equal(loc[..., 2].numpy(), g["op_loc"][..., 2])
    assert_close(loc.numpy(), g["op_loc"], 1e-6, "packed locations")
    assert np.array_equal(w.numpy(), g["op_w"])
    # op (oracle on channel-last features) + unpack == the reference's final output
    feats_cl = [np.ascontiguousarray(g[f"feat_cf{i}"].transpose(0, 2, 3, 4, 1)) for i in range(4)]
    op_out = oracle.forward(feats_cl, g["op_loc"], g["op_w"])
    final = opus_sampling.unpack(torch.from_numpy(op_out), B, T, G)
    assert_close(final.numpy(), g["out"], FWD_RTOL, "sampling_4d output")


def test_relayout_features_matches_decoder_convention():
    from opus_amd import opus_sampling
    B, T, N, G, C, H, W = 2, 3, 2, 2, 4, 3, 5
    feat = torch.arange(B * T * N * G * C * H * W, dtype=torch.float32).reshape(B, T * N, G * C, H, W)
    out = opus_sampling.relayout_features_torch([feat], T, N, G)[0]
    assert out.shape == (B * T * G, N, H, W, C) and out.is_contiguous()
    b, t, g, n, c, h, w = 1, 2, 1, 1, 3, 2, 4
    assert out[(b * T + t) * G + g, n, h, w, c] == feat[b, t * N + n, g * C + c, h, w]


def test_sampling_inputs_follow_the_decoder_module():
    """Restatement of models/opus_transformer.py:337-358: centre + offset * std, the same points in
    every frame, softmax over levels; a query with a single point has scale 0."""
    from opus_amd import opus_sampling
    B, Q, R, E, T, G, P, L = 2, 5, 8, 16, 3, 2, 4, 4
    gen = torch.Generator().manual_seed(0)
    qp, feat = torch.rand((B, Q, R, 3), generator=gen), torch.randn((B, Q, E), generator=gen)
    ow, ob = torch.randn((G * P * 3, E), generator=gen), torch.randn((G * P * 3,), generator=gen)
    sw_w, sw_b = torch.randn((G * P * L, E), generator=gen), torch.randn((G * P * L,), generator=gen)
    rng = [-40.0, -40.0, -1.0, 40.0, 40.0, 5.4]
    sp, sw = opus_sampling.sampling_inputs(qp, feat, ow, ob, sw_w, sw_b, T, G, P, L, rng)
    assert sp.shape == (B, Q, T, G, P, 3) and sw.shape == (B, Q, G, T, P, L)
    assert torch.equal(sp[:, :, 0], sp[:, :, T - 1]) and torch.equal(sw[:, :, :, 0], sw[:, :, :, T - 1])
    assert torch.allclose(sw.sum(-1), torch.ones(B, Q, G, T, P))
    lo, hi = torch.tensor(rng[:3]), torch.tensor(rng[3:])
    pts = qp * (hi - lo) + lo
    off = (feat @ ow.T + ob).view(B, Q, G, P, 3)
    want = pts.mean(2)[:, :, None, None] + off * pts.std(2)[:, :, None, None]
    assert torch.allclose(sp[:, :, 0], want, atol=1e-5)
    sp1, _ = opus_sampling.sampling_inputs(qp[:, :, :1], feat, ow, ob, sw_w, sw_b, T, G, P, L, rng)
    assert torch.allclose(sp1[:, :, 0, 0, 0], pts[:, :, 0])


def test_workspace_bytes_is_zero_where_the_owner_algorithm_does_not_apply(lib):
    """More than 16 views exceed the sort's key table: no owner algorithm, hence no workspace
    (FeatureGradSink reads a non-zero size as 'the owner algorithm applies')."""
    ws = lib.opus_msmv_backward_workspace_bytes
    assert ws(4, 32, 64, 16, 100, 2, 0) > 0
    assert ws(4, 32, 64, 17, 100, 2, 0) == 0
    assert ws(4, 32, 64, 17, 100, 2, 2) == 0
    # the d-feat stage alone needs no d loc / d weights (it never writes them): the NULL check
    # that fires is the one for the data it does read
    null = ctypes.c_void_p(0)
    from opus_amd import _lib
    hw, ptrs = _lib.int_array([4, 4]), _lib.ptr_array([1 << 12])
    one = ctypes.c_void_p(1 << 12)
    bwd = lib.opus_msmv_backward
    # workspace missing is the first complaint once all needed pointers are there
    assert bwd(1, one, ptrs, hw, one, one, 1, 64, 6, 1, 1, ptrs, null, null, null, 0, 0x202, null) == -7   # L = 1: no owner
    real = _lib.int_array([64, 176, 32, 88, 16, 44, 8, 22])
    p4 = _lib.ptr_array([1 << 12] * 4)
    assert bwd(4, one, p4, real, one, one, 1, 64, 6, 1, 1, p4, null, null, null, 0, 0x202, null) == -6    # d-feat stage: d loc may be NULL
    assert bwd(4, one, p4, real, one, one, 1, 64, 6, 1, 1, p4, null, null, null, 0, 0x402, null) == -4    # gather stage: it may not


_CHILD_ENV = {"OPUS_MSMV_LIB": "/nonexistent/libopus_msmv.so"}


def _run_child(code, **env):
    import subprocess
    import sys
    e = dict(os.environ, **_CHILD_ENV, **env)
    e["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")])
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_dispatcher_without_the_library_raises_by_default():
    out = _run_child("""
import torch
from opus_amd.csrc import wrapper
assert wrapper.MSMV_CUDA is False
try:
    wrapper.msmv_sampling([torch.zeros(1, 1, 1, 2, 2)] * 4, torch.zeros(1, 1, 1, 3), torch.zeros(1, 1, 1, 4))
except RuntimeError as e:
    assert 'not loaded' in str(e)
    print('raised')
""")
    assert "Warning: failed to load the HIP extension" in out and "raised" in out


def test_dispatcher_fallback_arm_is_the_reference_s(oracle):
    """OPUS_MSMV_ALLOW_FALLBACK=1 without the library: MSMV_CUDA is False and msmv_sampling is
    msmv_sampling_pytorch on channel-first features (reference wrapper.py:8-11, :87-93) - checked
    on the 4- and 5-level fixtures and, through sampling_4d, on the caller fixture."""
    out = _run_child("""
import numpy as np, torch
from helpers import assert_close, golden_feats, load_golden, FWD_RTOL
from opus_amd.csrc import wrapper
from opus_amd import opus_sampling
assert wrapper.MSMV_CUDA is False
for name in ('g1_tiny4', 'g3_tiny5'):
    g = load_golden(name)
    cf = [torch.from_numpy(f).permute(0, 4, 1, 2, 3).contiguous() for f in golden_feats(g)]
    out = wrapper.msmv_sampling(cf, torch.from_numpy(g['loc']), torch.from_numpy(g['w']))
    assert_close(out.numpy(), g['out'], 1e-6, name)
g = load_golden('g5_sampling4d')
cf = [torch.from_numpy(g[f'feat_cf{i}']) for i in range(4)]
out = opus_sampling.sampling_4d(torch.from_numpy(g['sample_points']), cf, torch.from_numpy(g['scale_weights']),
                                torch.from_numpy(g['occ2img']), int(g['image_h']), int(g['image_w']),
                                int(g['num_views']), fused=False)
assert_close(out.numpy(), g['out'], FWD_RTOL, 'sampling_4d through the fallback')
print('fallback ok')
""", OPUS_MSMV_ALLOW_FALLBACK="1")
    assert "performance may be hurt" in out and "fallback ok" in out


def test_opus_t_batch1_on_cpu_through_the_fallback(oracle):
    """The plumbing configuration: OPUS-T (Q = 600, P = 4), batch 1, 8 frames x 6 views x 4
    levels at full size on the CPU - decoder layout prep (channel-first), sampling_4d's
    projection / packing, msmv_sampling's fallback arm, un-pack - against the C oracle on the
    same packed locations (first and last slab)."""
    out = _run_child("""
import time
import numpy as np, torch
from helpers import REAL_LEVELS, FWD_RTOL, assert_close, sampling4d_case
from opus_amd.csrc import wrapper
from opus_amd import opus_sampling
from oracle import msmv_oracle
assert wrapper.MSMV_CUDA is False
B, Q, T, G, P, N, C = 1, 600, 8, 4, 4, 6, 64
pts, sw, occ2img = sampling4d_case(11, B, Q, T, G, P, N=N)
gen = torch.Generator().manual_seed(3)
backbone = [torch.randn((B, T * N, G * C, h, w), generator=gen) for h, w in REAL_LEVELS]
cf = opus_sampling.relayout_features_channel_first(backbone, T, N, G)
assert cf[0].shape == (B * T * G, C, N, 64, 176)
t0 = time.perf_counter()
out = opus_sampling.sampling_4d(torch.from_numpy(pts), cf, torch.from_numpy(sw), torch.from_numpy(occ2img),
                                256, 704, N, fused=False)
dt = time.perf_counter() - t0
assert out.shape == (B, Q, G, T * P, C)
loc, w = opus_sampling.project_and_pack_torch(torch.from_numpy(pts), torch.from_numpy(sw),
                                              torch.from_numpy(occ2img), 256, 704, N)
cl = opus_sampling.relayout_features_torch(backbone, T, N, G)
packed = out.reshape(B, Q, G, T, P, C).permute(0, 3, 2, 1, 5, 4).reshape(B * T * G, Q, C, P)
for slab in (0, B * T * G - 1):
    ref = msmv_oracle.forward([f[slab:slab + 1].numpy() for f in cl], loc[slab:slab + 1].numpy(),
                              w[slab:slab + 1].numpy())
    assert_close(packed[slab:slab + 1].numpy(), ref, FWD_RTOL, f'slab {slab}')
print('opus-t cpu ok: %d points in %.2f s on %d threads' % (B * T * G * Q * P, dt, torch.get_num_threads()))
""", OPUS_MSMV_ALLOW_FALLBACK="1")
    assert "opus-t cpu ok" in out


def test_feature_grad_sink_is_released_without_a_cyclic_collection():
    """sink -> token -> grad_fn must not lead back to the sink: dropping the last reference
    frees it (and its aliases of the feature maps) at once, gc disabled."""
    import gc
    import weakref
    from opus_amd.csrc import wrapper
    gc.collect()
    gc.disable()
    try:
        feats = [torch.zeros(1, 2, 2, 2, 4, requires_grad=True) for _ in range(4)]
        sink = wrapper.FeatureGradSink(feats, defer=False)
        assert sink.token.grad_fn is not None
        ref = weakref.ref(sink)
        alias = weakref.ref(sink.feats[0])
        del sink
        assert ref() is None and alias() is None
    finally:
        gc.enable()


def test_feature_grad_sink_notices_in_place_updates():
    from opus_amd.csrc import wrapper
    feats = [torch.zeros(1, 2, 2, 2, 4) for _ in range(4)]
    sink = wrapper.FeatureGradSink(feats, defer=False)
    sink.check_versions()
    feats[2].add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        sink.check_versions()
