"""GPU parity tests of the fused caller-side kernels (opus_msmv_project_pack and its backward,
through the C ABI via opus_amd.opus_sampling) against the golden fixtures of the reference's
sampling_4d and the caller-side CPU oracle.

Bars: against the float32 oracle (whose operation order the kernel follows) the chosen camera,
the projected coordinates and the packed weights are bit-identical; against a device matrix
product (summation order of the BLAS library) a differing camera is accepted only where a
visibility test sits within 1e-4 of its threshold, coordinates 1e-6 relative; gradients 1e-5
relative against the float64 analytic oracle and 1e-4 end to end (helpers.GRAD_RTOL)."""
import numpy as np
import pytest
import torch

from helpers import (FWD_RTOL, GRAD_RTOL, REAL_LEVELS, TINY_LEVELS, assert_close, load_golden,
                     random_case, sampling4d_case)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from opus_amd import opus_sampling
    from opus_amd.csrc import wrapper
    assert wrapper.MSMV_CUDA, "libopus_msmv.so must be loaded on the GPU box"
    return opus_sampling


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def visibility_margin(pts, occ2img, image_h, image_w, N, eps=1e-5):
    """Per point (in [B,T,G,Q,P] order): the smallest distance of any camera's (u, v, z)
    from a visibility threshold, in float64."""
    B, Q, T, G, P, _ = pts.shape
    mats = occ2img.astype(np.float64).reshape(B, T, N, 4, 4)
    p = np.concatenate([pts.astype(np.float64), np.ones(pts.shape[:-1] + (1,))], -1)
    p = p.transpose(0, 2, 3, 1, 4, 5)                                  # [B,T,G,Q,P,4]
    cam = np.einsum("btnij,btgqpj->btngqpi", mats[..., :3, :], p)
    z = cam[..., 2]
    hz = np.maximum(z, eps)
    u, v = cam[..., 0] / hz / image_w, cam[..., 1] / hz / image_h
    m = np.minimum.reduce([np.abs(u), np.abs(u - 1), np.abs(v), np.abs(v - 1), np.abs(z - eps)])
    return m.min(axis=2)


def check_packed(loc, w, ref_loc, ref_w, pts, occ2img, H, W, N, what, exact=True):
    """exact (the reference statement is the float32 oracle, whose operation order the kernel
    follows): the chosen camera AND the projected coordinates must be bit-identical - camera
    choice is index work.  Not exact (the statement is a device matrix product, whose summation
    order is the BLAS library's): a differing camera is accepted where a visibility test sits
    within 1e-4 of its threshold."""
    same = loc[..., 2] == ref_loc[..., 2]
    if exact:
        assert same.all(), f"{what}: {int((~same).sum())} points chose a different camera"
        assert np.array_equal(loc, ref_loc), \
            f"{what}: projected coordinates differ, max {np.abs(loc.astype(np.float64) - ref_loc).max():.3e}"
        assert np.array_equal(w, ref_w), f"{what}: packed weights"
        return same
    if not same.all():
        B, Q, T, G, P, _ = pts.shape
        margin = visibility_margin(pts, occ2img, H, W, N).reshape(B * T * G, Q, P)
        assert (margin[~same] < 1e-4).all(), f"{what}: camera differs away from any threshold"
        assert (~same).mean() < 1e-3, f"{what}: too many threshold cases to be rounding"
    scale = np.abs(ref_loc[same]).max()
    err = np.abs(loc[same].astype(np.float64) - ref_loc[same]).max()
    assert err <= 1e-6 * scale, f"{what}: loc max|diff| {err:.3e} vs scale {scale:.3e}"
    assert np.array_equal(w, ref_w), f"{what}: packed weights"
    return same


def test_project_pack_matches_golden(ops):
    g = load_golden("g5_sampling4d")
    H, W, N = int(g["image_h"]), int(g["image_w"]), int(g["num_views"])
    loc, w = ops.project_and_pack(dev(g["sample_points"]), dev(g["scale_weights"]), dev(g["occ2img"]), H, W, N)
    loc, w = loc.cpu().numpy(), w.cpu().numpy()
    assert loc.shape == g["op_loc"].shape and w.shape == g["op_w"].shape
    assert np.array_equal(loc[..., 2], g["op_loc"][..., 2])
    assert_close(loc, g["op_loc"], 1e-6, "packed locations")
    assert np.array_equal(w, g["op_w"])


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "tensor-ops"])
def test_sampling_4d_end_to_end_matches_golden(ops, fused):
    """Forward and every gradient of caller + op against the reference's own run."""
    g, gg = load_golden("g5_sampling4d"), load_golden("g6_sampling4d_grad")
    H, W, N = int(g["image_h"]), int(g["image_w"]), int(g["num_views"])
    feats = [dev(g[f"feat_cf{i}"].transpose(0, 2, 3, 4, 1)).requires_grad_(True) for i in range(4)]
    pts = dev(g["sample_points"]).requires_grad_(True)
    sw = dev(g["scale_weights"]).requires_grad_(True)
    out = ops.sampling_4d(pts, feats, sw, dev(g["occ2img"]), H, W, N, fused=fused)
    assert_close(out.detach().cpu().numpy(), g["out"], FWD_RTOL, "sampling_4d")
    out.backward(dev(gg["grad_out"]))
    assert_close(pts.grad.cpu().numpy(), gg["d_sample_points"], GRAD_RTOL, "d sample_points")
    assert_close(sw.grad.cpu().numpy(), gg["d_scale_weights"], GRAD_RTOL, "d scale_weights")
    for i, f in enumerate(feats):
        assert_close(f.grad.permute(0, 4, 1, 2, 3).cpu().numpy(), gg[f"d_feat_cf{i}"], GRAD_RTOL, f"d feat {i}")


def test_forward_order_switch_gives_the_same_results(ops, monkeypatch):
    """set_forward_sorted(False) (forward in input order, the backward sorting for itself) and
    the default (one sort serving forward and backward) are the same function through
    sampling_4d's autograd path: forward values, d sample_points and d scale_weights equal bit
    for bit (same arithmetic per point), d feats within the accumulation-order bar.  64 channels
    and real level sizes, so that the bin-ordered forward is the one taken when switched on."""
    from opus_amd.csrc import wrapper
    B, Q, T, G, P, N, L = 1, 400, 2, 2, 2, 6, 4
    H, W = 256, 704
    pts, sw, occ2img = sampling4d_case(31, B, Q, T, G, P, N, L, H, W)
    gen = torch.Generator(device="cuda").manual_seed(31)
    base = [torch.randn((B * T * G, N, h, w_, 64), generator=gen, device="cuda") for h, w_ in REAL_LEVELS]
    gout = None
    got, taken = {}, []
    real = wrapper.msmv_forward_sorted

    def counted(*args, **kw):
        res = real(*args, **kw)
        taken.append(res is not None)
        return res
    monkeypatch.setattr(wrapper, "msmv_forward_sorted", counted)
    try:
        for flag in (True, False):
            wrapper.set_forward_sorted(flag)
            feats = [f.clone().requires_grad_(True) for f in base]
            d_pts, d_sw = dev(pts).requires_grad_(True), dev(sw).requires_grad_(True)
            out = ops.sampling_4d(d_pts, feats, d_sw, dev(occ2img), H, W, N)
            if gout is None:
                gout = torch.randn(out.shape, generator=gen, device="cuda")
            out.backward(gout)
            got[flag] = [out.detach(), d_pts.grad, d_sw.grad, *[f.grad for f in feats]]
    finally:
        wrapper.set_forward_sorted(True)
    assert taken == [True], "the bin-ordered forward runs exactly once: when switched on"
    a, b = got[True], got[False]
    assert a[0].abs().max() > 0 and a[1].abs().max() > 0
    assert torch.equal(a[0], b[0]), "forward differs between the two orders"
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), "d points / d weights differ"
    for i in range(L):
        assert_close(a[3 + i].cpu().numpy(), b[3 + i].cpu().numpy(), 1e-5, f"d feat {i} between the two orders")


@pytest.mark.parametrize("shape", [(2, 300, 8, 4, 2, 6, 4), (1, 37, 3, 5, 4, 6, 5), (1, 64, 2, 1, 1, 2, 4),
                                   (3, 5, 1, 2, 7, 7, 4)],
                         ids=["T8G4P2", "T3G5P4L5", "N2", "P7N7"])
def test_project_pack_matches_oracle(ops, shape):
    from oracle import sampling4d_oracle as so
    B, Q, T, G, P, N, L = shape
    H, W = 256, 704
    pts, sw, occ2img = sampling4d_case(77 + Q, B, Q, T, G, P, N, L, H, W)
    ref_loc, ref_w = so.project_pack(pts, sw, occ2img, H, W, N)
    d_pts, d_sw = dev(pts).requires_grad_(True), dev(sw).requires_grad_(True)
    loc, w = ops.project_and_pack(d_pts, d_sw, dev(occ2img), H, W, N)
    same = check_packed(loc.detach().cpu().numpy(), w.detach().cpu().numpy(), ref_loc, ref_w,
                        pts, occ2img, H, W, N, "fused vs oracle")

    rng = np.random.default_rng(5)
    dloc = rng.standard_normal(ref_loc.shape).astype(np.float32)
    dloc[~same] = 0            # a different camera is a different function: compare the rest
    dw = rng.standard_normal(ref_w.shape).astype(np.float32)
    torch.autograd.backward([loc, w], [dev(dloc), dev(dw)])
    ref_dpts, ref_dsw = so.project_pack_backward(dloc, dw, pts, occ2img, ref_loc, H, W, N)
    assert_close(d_pts.grad.cpu().numpy(), ref_dpts, 1e-5, "d sample_points")
    assert np.array_equal(d_sw.grad.cpu().numpy(), ref_dsw)


def test_project_pack_full_size_matches_tensor_ops(ops):
    """OPUS-L caller shape (B=1, Q=4800, T=8, G=4, P=2): the kernel against the reference's
    own formulation run as tensor ops on the same device, forward and backward."""
    B, Q, T, G, P, N, L = 1, 4800, 8, 4, 2, 6, 4
    H, W = 256, 704
    pts, sw, occ2img = sampling4d_case(9, B, Q, T, G, P, N, L, H, W)
    a_pts, a_sw = dev(pts).requires_grad_(True), dev(sw).requires_grad_(True)
    b_pts, b_sw = dev(pts).requires_grad_(True), dev(sw).requires_grad_(True)
    loc, w = ops.project_and_pack(a_pts, a_sw, dev(occ2img), H, W, N)
    rloc, rw = ops.project_and_pack_torch(b_pts, b_sw, dev(occ2img), H, W, N)
    same = check_packed(loc.detach().cpu().numpy(), w.detach().cpu().numpy(),
                        rloc.detach().cpu().numpy(), rw.detach().cpu().numpy(),
                        pts, occ2img, H, W, N, "fused vs tensor ops", exact=False)
    # ... and bit for bit against the float32 oracle at this size
    from oracle import sampling4d_oracle as so
    o_loc, o_w = so.project_pack(pts, sw, occ2img, H, W, N)
    check_packed(loc.detach().cpu().numpy(), w.detach().cpu().numpy(), o_loc, o_w,
                 pts, occ2img, H, W, N, "fused vs oracle, OPUS-L size")
    views = np.rint(loc.detach().cpu().numpy()[..., 2] * (N - 1)).astype(int)
    assert (np.bincount(views.ravel(), minlength=N) > 0).all(), "every camera should be chosen somewhere"
    gen = torch.Generator(device="cuda").manual_seed(3)
    dloc = torch.randn(loc.shape, generator=gen, device="cuda") * dev(same.astype(np.float32))[..., None]
    dw = torch.randn(w.shape, generator=gen, device="cuda")
    torch.autograd.backward([loc, w], [dloc, dw])
    torch.autograd.backward([rloc, rw], [dloc, dw])
    assert_close(a_pts.grad.cpu().numpy(), b_pts.grad.cpu().numpy(), 1e-5, "d sample_points")
    assert torch.equal(a_sw.grad, b_sw.grad)


def test_project_pack_edge_cases(ops):
    from opus_amd import _lib
    H, W, N = 256, 704, 6
    # empty query set
    pts, sw, occ2img = sampling4d_case(1, 1, 0, 2, 2, 2, N)
    loc, w = ops.project_and_pack(dev(pts), dev(sw), dev(occ2img), H, W, N)
    assert loc.shape == (4, 0, 2, 3) and w.shape == (4, 0, 2, 4)
    # N = 1 would divide by zero in the view coordinate; more than 32 points per query is
    # beyond the op's limit
    pts, sw, occ2img = sampling4d_case(1, 1, 4, 2, 2, 2, 1)
    with pytest.raises(RuntimeError):
        ops.project_and_pack(dev(pts), dev(sw), dev(occ2img), H, W, 1)
    pts, sw, occ2img = sampling4d_case(1, 1, 4, 1, 1, 33, N)
    with pytest.raises(RuntimeError):
        ops.project_and_pack(dev(pts), dev(sw), dev(occ2img), H, W, N)
    # CPU tensors are refused, not silently computed elsewhere
    pts, sw, occ2img = sampling4d_case(1, 1, 4, 2, 2, 2, N)
    with pytest.raises(RuntimeError):
        ops.project_and_pack(torch.from_numpy(pts), torch.from_numpy(sw), torch.from_numpy(occ2img), H, W, N)
    assert _lib.load().opus_msmv_version() >= 100


# ---------------------------------------------------------------- row mode (op + un-pack)
ROW_CASES = [
    # Bs, T, G, Q, P, C, levels
    (1, 8, 4, 50, 2, 64, REAL_LEVELS),
    (2, 3, 2, 33, 4, 64, REAL_LEVELS),
    (1, 2, 5, 17, 1, 64, REAL_LEVELS),
    (1, 3, 1, 9, 7, 64, REAL_LEVELS),
    (2, 1, 3, 21, 2, 64, REAL_LEVELS + [(4, 11)]),
    (2, 2, 3, 11, 3, 8, TINY_LEVELS),
    (1, 1, 1, 1, 1, 64, REAL_LEVELS),
]


@pytest.mark.parametrize("case", ROW_CASES, ids=lambda c: "Bs%dT%dG%dQ%dP%dC%dL%d" % (*c[:6], len(c[6])))
def test_row_mode_equals_op_plus_unpack(ops, case):
    """opus_msmv_forward_rows / _backward_rows against the plain entry points with the
    reference's permute applied around them: same arithmetic, another store / load address."""
    from opus_amd.csrc import wrapper as wr
    Bs, T, G, Q, P, C, levels = case
    B = Bs * T * G
    feats, loc, w, _ = random_case(4242 + Q, levels, B, 6, C, Q, P)
    f = [dev(x) for x in feats]
    l, ww = dev(loc), dev(w)
    plain = wr.msmv_forward(f, l, ww)
    rows = wr.msmv_forward(f, l, ww, rows=(T, G))
    assert rows.shape == (Bs, Q, G, T * P, C)
    assert torch.equal(rows, ops.unpack(plain, Bs, T, G).contiguous()), "forward must be bit-identical"

    gen = torch.Generator(device="cuda").manual_seed(7)
    g_rows = torch.randn(rows.shape, generator=gen, device="cuda")
    # the same gradient in the op's own layout: adjoint of unpack
    g_plain = g_rows.reshape(Bs, Q, G, T, P, C).permute(0, 3, 2, 1, 5, 4).reshape(B, Q, C, P).contiguous()
    algos = ["atomic", "owner"] if C == 64 and len(levels) in (4, 5) else ["atomic"]
    for algo in algos:
        df_p, dl_p, dw_p = wr.msmv_backward(g_plain, f, l, ww, algo=algo)
        df_r, dl_r, dw_r = wr.msmv_backward(g_rows, f, l, ww, algo=algo, rows=(T, G))
        for i, (a, b) in enumerate(zip(df_r, df_p)):
            assert_close(a.cpu().numpy(), b.cpu().numpy(), GRAD_RTOL, f"{algo} dfeat{i}")
        assert_close(dl_r.cpu().numpy(), dl_p.cpu().numpy(), 1e-5, f"{algo} dloc")
        assert_close(dw_r.cpu().numpy(), dw_p.cpu().numpy(), 1e-5, f"{algo} dw")
        assert (dl_r[..., 2] == 0).all()


def test_row_mode_rejects_bad_batch(ops):
    from opus_amd.csrc import wrapper as wr
    feats, loc, w, _ = random_case(1, TINY_LEVELS, 6, 6, 8, 4, 2)
    with pytest.raises(RuntimeError):
        wr.msmv_forward([dev(x) for x in feats], dev(loc), dev(w), rows=(4, 1))     # 6 % 4 != 0


def test_row_mode_full_size(ops):
    """OPUS-L and OPUS-T sizes: forward bit-identical to op + permute; backward agrees."""
    from opus_amd.csrc import wrapper as wr
    for Q, P in ((4800, 2), (600, 4)):
        gen = torch.Generator(device="cuda").manual_seed(Q)
        T, G, B, N, C = 8, 4, 32, 6, 64
        f = [torch.randn((B, N, h, w_, C), generator=gen, device="cuda") for h, w_ in REAL_LEVELS]
        xy = torch.rand((B, Q, P, 2), generator=gen, device="cuda") * 1.2 - 0.1
        view = torch.randint(0, N, (B, Q, P, 1), generator=gen, device="cuda").float() / (N - 1)
        l = torch.cat([xy, view], -1).contiguous()
        ww = torch.softmax(torch.randn((B, Q, P, 4), generator=gen, device="cuda"), -1)
        plain = wr.msmv_forward(f, l, ww)
        rows = wr.msmv_forward(f, l, ww, rows=(T, G))
        assert torch.equal(rows, ops.unpack(plain, 1, T, G).contiguous())
        g_rows = torch.randn(rows.shape, generator=gen, device="cuda")
        g_plain = g_rows.reshape(1, Q, G, T, P, C).permute(0, 3, 2, 1, 5, 4).reshape(B, Q, C, P).contiguous()
        df_p, dl_p, dw_p = wr.msmv_backward(g_plain, f, l, ww, algo="owner")
        df_r, dl_r, dw_r = wr.msmv_backward(g_rows, f, l, ww, algo="owner", rows=(T, G))
        for i, (a, b) in enumerate(zip(df_r, df_p)):
            assert_close(a.cpu().numpy(), b.cpu().numpy(), GRAD_RTOL, f"P{P} dfeat{i}")
        assert_close(dl_r.cpu().numpy(), dl_p.cpu().numpy(), 1e-5, f"P{P} dloc")
        assert_close(dw_r.cpu().numpy(), dw_p.cpu().numpy(), 1e-5, f"P{P} dw")


# ---------------------------------------------------------------- producer-side re-layout
RELAYOUT_CASES = [
    # B, T, N, G, C, H, W
    (1, 2, 6, 4, 64, 16, 44),      # real level 2 (S = 704 = 11 tiles)
    (1, 1, 2, 2, 64, 8, 22),       # real level 3 (S = 176: a ragged last tile)
    (2, 3, 2, 2, 8, 3, 5),         # tiny, C < tile, S odd: scalar path
    (1, 2, 1, 3, 100, 7, 9),       # C > tile, not a multiple of it
    (1, 1, 1, 1, 64, 1, 1),
    (1, 2, 3, 2, 12, 4, 6),        # C % 4 == 0, S % 4 == 0 but both smaller than a tile
]


@pytest.mark.parametrize("case", RELAYOUT_CASES, ids=lambda c: "x".join(map(str, c)))
def test_relayout_is_the_reference_permute(ops, case):
    """A pure data movement: bit-exact against reshape/permute/contiguous, both directions."""
    B, T, N, G, C, H, W = case
    gen = torch.Generator(device="cuda").manual_seed(11)
    feat = torch.randn((B, T * N, G * C, H, W), generator=gen, device="cuda").requires_grad_(True)
    ref = ops.relayout_features_torch([feat.detach()], T, N, G)[0]
    out = ops.relayout_features([feat], T, N, G)[0]
    assert out.shape == ref.shape and out.is_contiguous()
    assert torch.equal(out, ref)
    g = torch.randn(out.shape, generator=gen, device="cuda")
    out.backward(g)
    # the adjoint of a permutation is its inverse
    ref_grad = g.reshape(B, T, G, N, H, W, C).permute(0, 1, 3, 2, 6, 4, 5).reshape(B, T * N, G * C, H, W)
    assert torch.equal(feat.grad, ref_grad)


def test_relayout_full_size_round_trip(ops):
    """All four levels of one OPUS sample (735 MB): forward equals the permute, inverse(forward) = id."""
    from opus_amd import _lib
    import ctypes
    T, N, G, C = 8, 6, 4, 64
    gen = torch.Generator(device="cuda").manual_seed(5)
    for H, W in REAL_LEVELS:
        feat = torch.randn((1, T * N, G * C, H, W), generator=gen, device="cuda")
        out = ops.relayout_features([feat], T, N, G)[0]
        assert torch.equal(out, ops.relayout_features_torch([feat], T, N, G)[0])
        back = torch.empty_like(feat)
        _lib.check(_lib.load().opus_msmv_relayout(
            ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(back.data_ptr()), 1, T, N, G, C, H, W, 1,
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "relayout")
        assert torch.equal(back, feat)


# ---------------------------------------------------------------- shared d-feat buffers
@pytest.mark.parametrize("defer", [True, False], ids=["deferred", "accumulated"])
@pytest.mark.parametrize("rows", [None, (2, 2)], ids=["plain", "rows"])
def test_feature_grad_sink_equals_autograd_sum(ops, rows, defer):
    """Three layers sampling the same feature maps: one shared, accumulated d-feat buffer must
    give what autograd's sum of three separate gradients gives; d loc / d w per layer too."""
    from opus_amd.csrc import wrapper as wr
    B, N, C, Q, P = 4, 6, 64, 40, 2
    feats, _, _, _ = random_case(300, REAL_LEVELS, B, N, C, Q, P)
    layers = [random_case(301 + i, TINY_LEVELS, B, N, 1, Q, P)[1:3] for i in range(3)]     # (loc, w) per layer
    gen = torch.Generator(device="cuda").manual_seed(9)

    def run(use_sink):
        f = [dev(x).requires_grad_(True) for x in feats]
        lw = [(dev(l).requires_grad_(True), dev(w).requires_grad_(True)) for l, w in layers]
        src = wr.FeatureGradSink(f, defer=defer) if use_sink else f
        assert not use_sink or src.defer == defer
        total = 0
        gen.manual_seed(9)
        for l, w in lw:
            out = wr.msmv_sampling(src, l, w) if rows is None else wr.msmv_sampling_rows(src, l, w, *rows)
            total = total + (out * torch.randn(out.shape, generator=gen, device="cuda")).sum()
        total.backward()
        return f, lw

    f_ref, lw_ref = run(False)
    f_snk, lw_snk = run(True)
    for i, (a, b) in enumerate(zip(f_snk, f_ref)):
        assert a.grad is not None
        assert_close(a.grad.cpu().numpy(), b.grad.cpu().numpy(), GRAD_RTOL, f"dfeat{i} through the sink")
    for (la, wa), (lb, wb) in zip(lw_snk, lw_ref):
        assert_close(la.grad.cpu().numpy(), lb.grad.cpu().numpy(), 1e-5, "dloc")
        assert_close(wa.grad.cpu().numpy(), wb.grad.cpu().numpy(), 1e-5, "dw")


@pytest.mark.parametrize("case", [
    # (levels, B, Q, P, rows, layers): few points on real level sizes; crowded coarse levels whose
    # patches are shared by several waves; more layers than one launch takes; row mode; no points
    ("real", 4, 150, 2, None, 3),
    ("tiny", 2, 900, 4, None, 3),
    ("real", 4, 40, 1, None, 10),
    ("real", 4, 60, 2, (2, 2), 4),
    ("tiny", 2, 33, 3, None, 2),
    ("real", 4, 0, 2, None, 2),
], ids=["real-p2", "crowded-p4", "ten-layers-p1", "rows", "p3-generic", "no-points"])
def test_deferred_dfeat_over_several_layers_matches_oracle_sum(ops, case):
    """Every layer sorts and gathers on its own (d loc / d w per layer equal the one-call
    backward's); one opus_msmv_dfeat_multi per batch of layers then writes d feats, started on
    NaN-filled buffers: equal to the oracle's sum over the layers at the gradient tolerance and to
    the sum of the one-call backward's d feats at 1e-5."""
    from opus_amd import _lib
    from opus_amd.csrc import wrapper as wr
    from oracle import msmv_oracle
    levels_name, B, Q, P, rows, nlayers = case
    levels = REAL_LEVELS if levels_name == "real" else TINY_LEVELS
    N, C = 6, 64
    feats, _, _, _ = random_case(600, levels, B, N, C, max(Q, 1), P)
    f = [dev(x) for x in feats]
    layers = []
    for i in range(nlayers):
        _, loc, w, _ = random_case(601 + i, levels, B, N, 1, Q, P)
        shape = (B, Q, C, P) if rows is None else (B // (rows[0] * rows[1]), Q, rows[1], rows[0] * P, C)
        gout = np.random.default_rng(650 + i).standard_normal(shape).astype(np.float32)
        layers.append((loc, w, gout))
    flags = _lib.BWD_OWNER | _lib.BWD_STAGE_BIN | _lib.BWD_STAGE_GATHER
    kept, dense = [], [torch.zeros_like(x) for x in f]
    for loc, w, gout in layers:
        l, ww, g = dev(loc), dev(w), dev(gout)
        ws = torch.empty(max(wr.backward_workspace_bytes(f, l, ww, flags), 16), dtype=torch.uint8, device="cuda")
        dloc, dw = torch.full_like(l, float("nan")), torch.full_like(ww, float("nan"))
        wr.msmv_backward_into(g, f, l, ww, None, dloc, dw, workspace=ws, flags=flags, rows=rows)
        one, dl_ref, dw_ref = wr.msmv_backward(g, f, l, ww, algo="owner", rows=rows)
        if Q:
            assert_close(dloc.cpu().numpy(), dl_ref.cpu().numpy(), 1e-6, "dloc, sort + gather only")
            assert_close(dw.cpu().numpy(), dw_ref.cpu().numpy(), 1e-6, "dw, sort + gather only")
        for d, o in zip(dense, one):
            d += o
        kept.append((g, ws))
    out = [torch.full_like(x, float("nan")) for x in f]
    cap = _lib.load().opus_msmv_dfeat_max_sets()
    assert cap == 8
    for i in range(0, nlayers, cap):
        part = kept[i:i + cap]
        wr.msmv_dfeat_multi([g for g, _ in part], [w_ for _, w_ in part], f, Q, P, out, rows=rows,
                            accumulate=i > 0)
    torch.cuda.synchronize()
    ref = [np.zeros_like(x) for x in feats]
    for loc, w, gout in layers:
        if rows is not None:            # [Bs,Q,G,T*P,C] -> [Bs*T*G,Q,C,P]
            T, G = rows
            gout = gout.reshape(B // (T * G), Q, G, T, P, C).transpose(0, 3, 2, 1, 5, 4).reshape(B, Q, C, P)
        if Q:
            df, _, _ = msmv_oracle.backward(np.ascontiguousarray(gout), feats, loc, w)
            for r, d in zip(ref, df):
                r += d
    for i, (a, d, r) in enumerate(zip(out, dense, ref)):
        assert not torch.isnan(a).any(), f"level {i}: elements left unwritten"
        assert_close(a.cpu().numpy(), d.cpu().numpy(), 1e-5, f"dfeat{i}, deferred against per-layer sum",
                     atol_floor=1e-30)
        assert_close(a.cpu().numpy(), r, GRAD_RTOL, f"dfeat{i}, deferred against the oracle", atol_floor=1e-30)


def test_deferred_dfeat_randomised_shapes(ops):
    """Twenty seeded random configurations of the multi-set pass (4 / 5 levels of random sizes,
    up to 16 views, 1-8 sets, any P, plain and row layout) against the oracle's sum."""
    from opus_amd import _lib, opus_sampling
    from opus_amd.csrc import wrapper as wr
    from oracle import msmv_oracle
    rng = np.random.default_rng(77)
    flags = _lib.BWD_OWNER | _lib.BWD_STAGE_BIN
    for case in range(20):
        L = int(rng.choice([4, 5]))
        levels = [(int(rng.integers(1, 41)), int(rng.integers(1, 41))) for _ in range(L)]
        B, N = int(rng.integers(1, 7)), int(rng.integers(1, 17))
        Q, P = int(rng.integers(1, 400)), int(rng.choice([1, 2, 3, 4, 8]))
        nsets = int(rng.integers(1, 9))
        rows = (B, 1) if case % 2 else None
        what = f"case {case}: levels {levels} B{B} N{N} Q{Q} P{P} sets {nsets} rows {rows}"
        feats, _, _, _ = random_case(7000 + case, levels, B, N, 64, 1, P)
        f = [dev(x) for x in feats]
        ref = [np.zeros_like(x) for x in feats]
        gs, wss = [], []
        for i in range(nsets):
            _, loc, w, gout = random_case(7100 + 10 * case + i, levels, B, N, 64, Q, P,
                                          oob_frac=float(rng.choice([0.0, 0.3, 0.8])))
            df, _, _ = msmv_oracle.backward(gout, feats, loc, w)
            for r, d in zip(ref, df):
                r += d
            l, ww, g = dev(loc), dev(w), dev(gout)
            if rows is not None:
                g = opus_sampling.unpack(g, 1, *rows).contiguous()
            ws = torch.empty(max(wr.backward_workspace_bytes(f, l, ww, flags), 16), dtype=torch.uint8, device="cuda")
            wr.msmv_backward_into(g, f, l, ww, None, torch.empty_like(l), torch.empty_like(ww), workspace=ws,
                                  flags=flags, rows=rows)
            gs.append(g)
            wss.append(ws)
        out = [torch.full_like(x, float("nan")) for x in f]
        wr.msmv_dfeat_multi(gs, wss, f, Q, P, out, rows=rows)
        for i, (a, r) in enumerate(zip(out, ref)):
            assert_close(a.cpu().numpy(), r, GRAD_RTOL, f"{what}: dfeat{i}", atol_floor=1e-30)


def test_deferred_dfeat_rejects_bad_arguments(ops):
    from opus_amd.csrc import wrapper as wr
    B, N, Q, P = 2, 6, 8, 2
    feats, loc, w, gout = random_case(690, TINY_LEVELS, B, N, 64, Q, P)
    f = [dev(x) for x in feats]
    g = dev(gout)
    ws = torch.empty(max(wr.backward_workspace_bytes(f, dev(loc), dev(w), 2), 16), dtype=torch.uint8, device="cuda")
    d = [torch.empty_like(x) for x in f]
    with pytest.raises(RuntimeError, match="dimensions"):
        wr.msmv_dfeat_multi([g] * 9, [ws] * 9, f, Q, P, d)                 # more sets than one launch takes
    with pytest.raises(RuntimeError, match="workspace"):
        wr.msmv_dfeat_multi([g], [ws[:64]], f, Q, P, d)
    with pytest.raises(RuntimeError, match="grad_output"):
        wr.msmv_dfeat_multi([g[:1]], [ws], f, Q, P, d)
    feats32 = [dev(x[..., :32].copy()) for x in feats]                     # C = 32: no owner algorithm
    with pytest.raises(RuntimeError, match="flag"):
        wr.msmv_dfeat_multi([g[:, :, :32].contiguous()], [ws], feats32, Q, P, [torch.empty_like(x) for x in feats32])
    assert not wr.FeatureGradSink([x.requires_grad_(True) for x in feats32]).defer
    assert wr.FeatureGradSink([x.requires_grad_(True) for x in f]).defer


def test_feature_grad_sink_through_relayout_and_sampling_4d(ops):
    """Backbone-layout features -> relayout -> sink -> two sampling_4d layers: gradients reach the
    backbone layout and equal those of the tensor-op chain."""
    B, Q, T, G, P, N, C = 1, 24, 2, 2, 2, 6, 64
    H, W = 256, 704
    pts, sw, occ2img = sampling4d_case(21, B, Q, T, G, P, N)
    gen = torch.Generator(device="cuda").manual_seed(2)
    raw0 = [torch.randn((B, T * N, G * C, h, w), generator=gen, device="cuda") for h, w in REAL_LEVELS]
    gouts = [torch.randn((B, Q, G, T * P, C), generator=gen, device="cuda") for _ in range(2)]
    from opus_amd.csrc import wrapper as wr

    def run(fused):
        raw = [r.clone().requires_grad_(True) for r in raw0]
        p = dev(pts).requires_grad_(True)
        s_ = dev(sw).requires_grad_(True)
        if fused:
            f = wr.FeatureGradSink(ops.relayout_features(raw, T, N, G))
        else:
            f = ops.relayout_features_torch(raw, T, N, G)
        total = 0
        for k, g in enumerate(gouts):
            out = ops.sampling_4d(p + 0.25 * k, f, s_, dev(occ2img), H, W, N, fused=fused)
            total = total + (out * g).sum()
        total.backward()
        return raw, p, s_

    raw_a, p_a, s_a = run(True)
    raw_b, p_b, s_b = run(False)
    for i, (a, b) in enumerate(zip(raw_a, raw_b)):
        assert_close(a.grad.cpu().numpy(), b.grad.cpu().numpy(), GRAD_RTOL, f"d raw feat {i}")
    assert_close(p_a.grad.cpu().numpy(), p_b.grad.cpu().numpy(), GRAD_RTOL, "d sample_points")
    assert_close(s_a.grad.cpu().numpy(), s_b.grad.cpu().numpy(), GRAD_RTOL, "d scale_weights")


def test_project_pack_randomised_shapes(ops):
    """Twenty seeded random caller shapes: packed locations / weights and both gradients against
    the numpy oracle (same acceptance rule for near-threshold camera choices as above)."""
    from oracle import sampling4d_oracle as so
    rng = np.random.default_rng(77)
    H, W = 256, 704
    for case in range(20):
        B, Q = int(rng.integers(1, 4)), int(rng.integers(1, 120))
        T, G, P = int(rng.integers(1, 9)), int(rng.integers(1, 5)), int(rng.integers(1, 9))
        N, L = int(rng.integers(2, 8)), int(rng.choice([1, 4, 5, 8]))
        pts, sw, occ2img = sampling4d_case(500 + case, B, Q, T, G, P, N, L, H, W)
        ref_loc, ref_w = so.project_pack(pts, sw, occ2img, H, W, N)
        d_pts, d_sw = dev(pts).requires_grad_(True), dev(sw).requires_grad_(True)
        loc, w = ops.project_and_pack(d_pts, d_sw, dev(occ2img), H, W, N)
        what = f"case {case}: B{B} Q{Q} T{T} G{G} P{P} N{N} L{L}"
        same = check_packed(loc.detach().cpu().numpy(), w.detach().cpu().numpy(), ref_loc, ref_w,
                            pts, occ2img, H, W, N, what)
        dloc = rng.standard_normal(ref_loc.shape).astype(np.float32)
        dloc[~same] = 0
        dw = rng.standard_normal(ref_w.shape).astype(np.float32)
        torch.autograd.backward([loc, w], [dev(dloc), dev(dw)])
        ref_dpts, ref_dsw = so.project_pack_backward(dloc, dw, pts, occ2img, ref_loc, H, W, N)
        # A point a few centimetres from its camera's plane has a gradient ~ 1/z^2 in the millions
        # and z itself is a float32 sum with cancellation: such elements are held to 5e-3 of
        # their own value, everything else to 1e-5 of the tensor's largest ordinary element.
        got = d_pts.grad.cpu().numpy().astype(np.float64)
        err = np.abs(got - ref_dpts)
        ordinary = np.abs(ref_dpts) < 1e3
        scale = np.abs(ref_dpts[ordinary]).max() if ordinary.any() else 1.0
        ok = (err <= 1e-5 * scale) | (err <= 5e-3 * np.abs(ref_dpts))
        assert np.isfinite(got).all() and ok.all(), f"{what} d sample_points: max|diff| {err[~ok].max():.3e}"
        assert np.array_equal(d_sw.grad.cpu().numpy(), ref_dsw), what


def test_shared_frames_equal_the_expanded_inputs(ops):
    """The decoder expands one set of points / weights over the T frames before it calls
    sampling_4d (models/opus_transformer.py:350-358).  Feeding the un-expanded tensors with
    num_frames=T must give the same output, and gradients equal to autograd's sum over T."""
    B, Q, T, G, P, N, L, C = 1, 40, 4, 2, 3, 6, 4, 64
    H, W = 256, 704
    pts, sw, occ2img = sampling4d_case(91, B, Q, T, G, P, N, L, H, W)
    base_pts, base_sw = pts[:, :, 0].copy(), sw[:, :, :, 0].copy()          # [B,Q,G,P,3], [B,Q,G,P,L]
    gen = torch.Generator(device="cuda").manual_seed(4)
    feats0 = [torch.randn((B * T * G, N, h, w_, C), generator=gen, device="cuda") for h, w_ in REAL_LEVELS]
    gout = torch.randn((B, Q, G, T * P, C), generator=gen, device="cuda")

    def run(shared):
        p0, s0 = dev(base_pts).requires_grad_(True), dev(base_sw).requires_grad_(True)
        feats = [f.clone().requires_grad_(True) for f in feats0]
        if shared:
            out = ops.sampling_4d(p0, feats, s0, dev(occ2img), H, W, N, num_frames=T)
        else:
            pe = p0[:, :, None].expand(B, Q, T, G, P, 3)
            se = s0[:, :, :, None].expand(B, Q, G, T, P, L)
            out = ops.sampling_4d(pe, feats, se, dev(occ2img), H, W, N)
        out.backward(gout)
        return out.detach(), p0.grad, s0.grad, [f.grad for f in feats]

    out_a, dp_a, ds_a, df_a = run(True)
    out_b, dp_b, ds_b, df_b = run(False)
    assert torch.equal(out_a, out_b)
    assert_close(dp_a.cpu().numpy(), dp_b.cpu().numpy(), 1e-5, "d sample_points (summed over frames)")
    assert_close(ds_a.cpu().numpy(), ds_b.cpu().numpy(), 1e-5, "d scale_weights (summed over frames)")
    for a_, b_ in zip(df_a, df_b):
        assert_close(a_.cpu().numpy(), b_.cpu().numpy(), GRAD_RTOL, "d feats")


def test_feature_grad_sink_under_activation_checkpointing(ops):
    """The reference wraps every sampling layer in torch.utils.checkpoint (use_reentrant=False,
    models/opus_transformer.py:372-375): the forward is recomputed inside the backward.  The shared
    d-feat buffer must still receive each layer exactly once."""
    from torch.utils.checkpoint import checkpoint
    from opus_amd.csrc import wrapper as wr
    B, N, C, Q, P, T, G = 4, 6, 64, 30, 2, 2, 2
    feats, _, _, _ = random_case(700, REAL_LEVELS, B, N, C, Q, P)
    layers = [random_case(701 + i, TINY_LEVELS, B, N, 1, Q, P)[1:3] for i in range(3)]
    gen = torch.Generator(device="cuda")

    def run(use_sink, use_checkpoint):
        f = [dev(x).requires_grad_(True) for x in feats]
        lw = [(dev(l).requires_grad_(True), dev(w).requires_grad_(True)) for l, w in layers]
        src = wr.FeatureGradSink(f) if use_sink else f
        gen.manual_seed(1)
        total = 0
        for l, w in lw:
            fn = lambda l_, w_: wr.msmv_sampling_rows(src, l_, w_, T, G)          # noqa: E731
            out = checkpoint(fn, l, w, use_reentrant=False) if use_checkpoint else fn(l, w)
            total = total + (out * torch.randn(out.shape, generator=gen, device="cuda")).sum()
        total.backward()
        return f, lw

    f_ref, lw_ref = run(False, False)
    f_chk, lw_chk = run(True, True)
    for i, (a, b) in enumerate(zip(f_chk, f_ref)):
        assert_close(a.grad.cpu().numpy(), b.grad.cpu().numpy(), GRAD_RTOL, f"dfeat{i}, sink + checkpoint")
    for (la, wa), (lb, wb) in zip(lw_chk, lw_ref):
        assert_close(la.grad.cpu().numpy(), lb.grad.cpu().numpy(), 1e-5, "dloc")
        assert_close(wa.grad.cpu().numpy(), wb.grad.cpu().numpy(), 1e-5, "dw")


def test_feature_grad_sink_releases_the_feature_maps_without_a_cyclic_collection(ops):
    """After a step through a sink, dropping the step's objects returns the allocator to where it
    was with the interpreter's cycle collector off: the sink must not sit in a reference cycle
    (sink -> token -> grad_fn -> sink) that only a generational collection would break."""
    import gc
    from opus_amd.csrc import wrapper as wr
    B, N, C, Q, P = 4, 6, 64, 40, 2
    feats, loc, w, _ = random_case(310, REAL_LEVELS, B, N, C, Q, P)
    gc.collect()
    gc.disable()
    try:
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        for defer in (True, False):
            f = [dev(x).requires_grad_(True) for x in feats]
            sink = wr.FeatureGradSink(f, defer=defer)
            l, ww = dev(loc).requires_grad_(True), dev(w).requires_grad_(True)
            total = wr.msmv_sampling(sink, l, ww).sum() + wr.msmv_sampling(sink, l, ww).square().sum()
            total.backward()
            assert f[0].grad is not None
            del f, sink, l, ww, total
            torch.cuda.synchronize()
            assert torch.cuda.memory_allocated() == base, f"defer={defer}: the step's tensors are still alive"
    finally:
        gc.enable()


def test_feature_grad_sink_with_more_views_than_the_sort_takes(ops):
    """17 views: the owner algorithm (and with it the deferred d-feat pass) does not apply; the
    sink must pick the layer-by-layer accumulation on the atomic kernels by itself."""
    from opus_amd.csrc import wrapper as wr
    B, N, C, Q, P = 2, 17, 64, 30, 2
    feats, loc, w, _ = random_case(320, TINY_LEVELS, B, N, C, Q, P)
    layers = [random_case(321 + i, TINY_LEVELS, B, N, 1, Q, P)[1:3] for i in range(2)]

    def run(use_sink):
        f = [dev(x).requires_grad_(True) for x in feats]
        src = wr.FeatureGradSink(f) if use_sink else f
        assert not use_sink or src.defer is False
        total = 0
        for i, (l, ww) in enumerate(layers):
            total = total + (wr.msmv_sampling(src, dev(l), dev(ww)) * (i + 1.5)).sum()
        total.backward()
        return f

    for i, (a, b) in enumerate(zip(run(True), run(False))):
        assert_close(a.grad.cpu().numpy(), b.grad.cpu().numpy(), GRAD_RTOL, f"dfeat{i}, 17 views")


SORTED_FWD_CASES = [
    # Bs, T, G, Q, P, levels, oob
    (1, 2, 2, 37, 2, REAL_LEVELS, 0.25),
    (2, 3, 1, 300, 4, REAL_LEVELS, 0.25),          # slabs of 1200 points
    (1, 8, 4, 450, 8, REAL_LEVELS, 0.4),           # slabs of 3600 points: the two-pass sort
    (1, 2, 2, 19, 3, REAL_LEVELS + [(4, 11)], 0.25),
    (1, 1, 3, 0, 2, REAL_LEVELS, 0.0),
]


@pytest.mark.parametrize("case", SORTED_FWD_CASES, ids=lambda c: "Bs%dT%dG%dQ%dP%dL%d" % (*c[:5], len(c[5])))
def test_bin_ordered_forward_is_bit_identical_and_its_sort_serves_the_backward(ops, case):
    """opus_msmv_forward_sorted against opus_msmv_forward_rows on the same inputs (same arithmetic
    per point: equal bit for bit), incl. points with no valid view (zero rows written by the sort),
    out-of-range locations, a re-run on the kept sort (PRESORTED), and the backward on that sort
    (OPUS_MSMV_BWD_PRESORTED) against the backward that sorts for itself."""
    from opus_amd import _lib
    from opus_amd.csrc import wrapper as wr
    Bs, T, G, Q, P, levels, oob = case
    B, N, C = Bs * T * G, 6, 64
    feats, loc, w, _ = random_case(900 + Q, levels, B, N, C, max(Q, 1), P, oob)
    loc, w = loc[:, :Q], w[:, :Q]
    if Q:
        loc[0, 0, 0, 2] = 1.7            # no such view
        loc[-1, Q // 2, -1, 2] = float("nan")
    f, l, ww = [dev(x) for x in feats], dev(loc), dev(w)
    want = wr.msmv_forward(f, l, ww, rows=(T, G))
    res = wr.msmv_forward_sorted(f, l, ww, rows=(T, G))
    assert res is not None
    out, ws = res
    assert torch.equal(out, want)
    out2, _ = wr.msmv_forward_sorted(f, l, ww, rows=(T, G), workspace=ws, presorted=True)
    assert torch.equal(out2, want)
    gout = dev(np.random.default_rng(5).standard_normal(tuple(want.shape)).astype(np.float32))
    df_ref, dl_ref, dw_ref = wr.msmv_backward(gout, f, l, ww, algo="owner", rows=(T, G))
    df, dl, dw_ = wr.msmv_backward(gout, f, l, ww, rows=(T, G), sorted_ws=ws)
    for i, (a, b) in enumerate(zip(df, df_ref)):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), 1e-5, f"dfeat{i} on the forward's sort")
    assert torch.equal(dl, dl_ref) and torch.equal(dw_, dw_ref)
    # the deferred form: gather alone on the kept sort
    if Q:
        dl2, dw2 = torch.full_like(l, float("nan")), torch.full_like(ww, float("nan"))
        wr.msmv_backward_into(gout, f, l, ww, None, dl2, dw2, workspace=ws,
                              flags=_lib.BWD_OWNER | _lib.BWD_PRESORTED | _lib.BWD_STAGE_GATHER, rows=(T, G))
        assert torch.equal(dl2, dl_ref) and torch.equal(dw2, dw_ref)


def test_bin_ordered_forward_plain_layout_and_refusals(ops):
    from opus_amd import _lib
    from opus_amd.csrc import wrapper as wr
    feats, loc, w, gout = random_case(950, REAL_LEVELS, 3, 6, 64, 700, 1, 0.25)
    f, l, ww = [dev(x) for x in feats], dev(loc), dev(w)
    out, ws = wr.msmv_forward_sorted(f, l, ww)                      # [B,Q,C,1]: a point's channels are contiguous
    assert torch.equal(out, wr.msmv_forward(f, l, ww))
    # through autograd: the Function keeps the sort for its backward
    fa = [x.clone().requires_grad_(True) for x in f]
    la, wa = l.clone().requires_grad_(True), ww.clone().requires_grad_(True)
    o = wr.msmv_sampling(fa, la, wa)
    o.backward(dev(gout))
    df_ref, dl_ref, dw_ref = wr.msmv_backward(dev(gout), f, l, ww, algo="owner")
    assert torch.equal(la.grad, dl_ref) and torch.equal(wa.grad, dw_ref)
    for a, b in zip(fa, df_ref):
        assert_close(a.grad.cpu().numpy(), b.cpu().numpy(), 1e-5, "d feat through autograd")
    # P > 1 in the [B,Q,C,P] layout, C != 64: not taken (None -> the caller runs the plain forward)
    feats2, loc2, w2, _ = random_case(951, REAL_LEVELS, 2, 6, 64, 30, 2)
    assert wr.msmv_forward_sorted([dev(x) for x in feats2], dev(loc2), dev(w2)) is None
    feats3, loc3, w3, _ = random_case(952, TINY_LEVELS, 2, 6, 8, 30, 1)
    assert wr.msmv_forward_sorted([dev(x) for x in feats3], dev(loc3), dev(w3)) is None
    # flag checks of the C entry points
    lib = _lib.load()
    hw = _lib.int_array([d for h_w in REAL_LEVELS for d in h_w])
    fp = _lib.ptr_array([x.data_ptr() for x in f])
    args = (4, fp, hw, l.data_ptr(), ww.data_ptr(), 3, 0, 0, 64, 6, 700, 1, out.data_ptr(), ws.data_ptr())
    assert lib.opus_msmv_forward_sorted(*args, ws.numel(), 0x8, None) == -7          # unknown flag
    assert lib.opus_msmv_forward_sorted(*args, 16, 0, None) == -6                    # workspace too small
    dl, dw_ = torch.empty_like(l), torch.empty_like(ww)
    with pytest.raises(RuntimeError, match="flag"):                                  # PRESORTED + STAGE_BIN
        wr.msmv_backward_into(dev(gout), f, l, ww, None, dl, dw_, workspace=ws,
                              flags=_lib.BWD_OWNER | _lib.BWD_PRESORTED | _lib.BWD_STAGE_BIN)
    with pytest.raises(RuntimeError, match="flag"):                                  # PRESORTED needs the owner algorithm
        wr.msmv_backward_into(dev(gout), f, l, ww, [torch.empty_like(x) for x in f], dl, dw_, workspace=ws,
                              flags=_lib.BWD_ATOMIC | _lib.BWD_PRESORTED)


def test_bin_ordered_forward_full_size(ops):
    """OPUS-L caller shape in row mode: equal to the input-order forward bit for bit."""
    from opus_amd.csrc import wrapper as wr
    gen = torch.Generator(device="cuda").manual_seed(31)
    Bs, T, G, Q, P, N, C = 1, 8, 4, 4800, 2, 6, 64
    B = Bs * T * G
    f = [torch.randn((B, N, h, w_, C), generator=gen, device="cuda") for h, w_ in REAL_LEVELS]
    xy = torch.rand((B, Q, P, 2), generator=gen, device="cuda") * 1.2 - 0.1
    view = torch.randint(0, N, (B, Q, P, 1), generator=gen, device="cuda").float() / (N - 1)
    l = torch.cat([xy, view], -1).contiguous()
    ww = torch.softmax(torch.randn((B, Q, P, 4), generator=gen, device="cuda"), -1)
    out, _ = wr.msmv_forward_sorted(f, l, ww, rows=(T, G))
    assert torch.equal(out, wr.msmv_forward(f, l, ww, rows=(T, G)))


def test_bin_ordered_forward_and_backward_on_its_sort_replay_from_a_hip_graph(ops):
    """The bin-ordered forward, its re-run on the kept sort and the backward on that sort allocate
    nothing inside the library: captured once with caller-owned buffers (the wrapper's output
    allocation aside, which the graph's pool serves), replayed on new data."""
    from opus_amd import _lib
    from opus_amd.csrc import wrapper as wr
    Bs, T, G, Q, P, N, C = 1, 2, 2, 150, 2, 6, 64
    B = Bs * T * G
    feats, loc, w, _ = random_case(31, REAL_LEVELS, B, N, C, Q, P, 0.25)
    f, l, ww = [dev(x) for x in feats], dev(loc), dev(w)
    g = dev(np.random.default_rng(1).standard_normal((Bs, Q, G, T * P, C)).astype(np.float32))
    need = wr.backward_workspace_bytes(f, l, ww, _lib.BWD_OWNER)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    out = torch.empty((Bs, Q, G, T * P, C), device="cuda")
    out2 = torch.empty_like(out)
    dfeats = [torch.empty_like(x) for x in f]
    dloc, dw_ = torch.empty_like(l), torch.empty_like(ww)

    def step():
        out.copy_(wr.msmv_forward_sorted(f, l, ww, rows=(T, G), workspace=ws)[0])
        out2.copy_(wr.msmv_forward_sorted(f, l, ww, rows=(T, G), workspace=ws, presorted=True)[0])
        wr.msmv_backward_into(g, f, l, ww, dfeats, dloc, dw_, workspace=ws,
                              flags=_lib.BWD_OWNER | _lib.BWD_PRESORTED, rows=(T, G))

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    feats2, loc2, w2, _ = random_case(32, REAL_LEVELS, B, N, C, Q, P, 0.25)
    for dst, src in zip(f, feats2):
        dst.copy_(dev(src))
    l.copy_(dev(loc2)); ww.copy_(dev(w2))
    g.copy_(dev(np.random.default_rng(2).standard_normal(tuple(g.shape)).astype(np.float32)))
    for t in dfeats + [dloc, dw_, out, out2]:
        t.fill_(float("nan"))
    ws.fill_(0xFF)
    graph.replay()
    torch.cuda.synchronize()
    want = wr.msmv_forward(f, l, ww, rows=(T, G))
    assert torch.equal(out, want) and torch.equal(out2, want)
    df_ref, dl_ref, dw_ref = wr.msmv_backward(g, f, l, ww, algo="owner", rows=(T, G))
    assert torch.equal(dloc, dl_ref) and torch.equal(dw_, dw_ref)
    for i, (a, b) in enumerate(zip(dfeats, df_ref)):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), 1e-5, f"replayed dfeat{i}")


@pytest.mark.parametrize("workload", ["chain", "infer"])
def test_bench_chain_workloads_equal_the_tensor_op_chain(ops, workload):
    """bench.py's chain-* / infer-* steps - re-layout, six layers of projection + forward (twice in
    training) + gather + projection backward, one d-feat pass, inverse re-layout, all as raw
    launches through the C ABI - at a reduced query count against the same chain written with
    tensor ops and autograd (the reference's formulation, models/opus_transformer.py:113-136 and
    models/opus_sampling.py:19-124, around the plain op)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    bench.load_product()
    Q, P, T, G, N, C, n_layers = 48, 2, 8, 4, 6, 64, 6
    train = workload == "chain"
    device = torch.device("cuda", 0)
    feats, sets = bench.make_inputs(T * G, Q, P, 5, device, n_sets=n_layers)
    cam = bench.camera_sets(T * G, Q, P, 5, device, n_layers)
    sets = [(loc, w, g.reshape(1, Q, G, T * P, C)) for loc, w, g in sets]
    gen = torch.Generator(device=device).manual_seed(9)
    backbone = [torch.randn((1, T * N, G * C, h, w), generator=gen, device=device) for h, w in bench.LEVELS]
    step = bench.Step(feats, sets, train, "owner", n_fwd=2 if train else 1, caller=cam, defer=True,
                      fwd_sorted=train, backbone=backbone)
    outs = []
    if not train:
        # the forward writes into a fresh tensor per call: keep every layer's output
        from opus_amd.csrc import wrapper as wr
        orig = wr.msmv_forward
        wr.msmv_forward = lambda *a, **k: outs.append(orig(*a, **k)) or outs[-1]
        try:
            step.run(False)
        finally:
            wr.msmv_forward = orig
    else:
        step.run(False)
    torch.cuda.synchronize()

    raw = [b.clone().requires_grad_(train) for b in backbone]
    f_ref = ops.relayout_features_torch(raw, T, N, G)
    for a, b in zip(step.feats, f_ref):
        assert torch.equal(a, b.detach()), "re-layout"
    total, last = 0, None
    for k, ((_, _, g), (pts, sw, occ2img)) in enumerate(zip(sets, cam)):
        p = pts.clone().requires_grad_(train)
        s_ = sw.clone().requires_grad_(train)
        out = ops.sampling_4d(p, f_ref, s_, occ2img, 256, 704, N, fused=False)
        if train:
            total = total + (out * g).sum()
            last = (p, s_)
        else:
            # the tensor-op projection sums its 4x4 products in the BLAS library's order, the kernel in
            # the reference statement's: coordinates agree to 1e-6, which levels up to 176 pixels wide
            # turn into 1e-4 of a pixel - hence the gradient bar here; the step's own projection through
            # the plain op and the reference's permute must give the very same bits
            assert_close(outs[k].cpu().numpy(), out.detach().cpu().numpy(), GRAD_RTOL, f"layer {k} output")
            from opus_amd.csrc import wrapper as wr
            loc_k, w_k, _ = sets[k]
            assert torch.equal(outs[k], ops.unpack(wr.msmv_forward(step.feats, loc_k, w_k), 1, T, G)), f"layer {k}: row mode"
    if train:
        total.backward()
        for i, (a, b) in enumerate(zip(step.dbackbone, raw)):
            assert_close(a.cpu().numpy(), b.grad.cpu().numpy(), GRAD_RTOL, f"d backbone feat {i}")
        assert_close(step.dpts.cpu().numpy(), last[0].grad.cpu().numpy(), GRAD_RTOL, "d sample_points (last layer)")
        assert_close(step.dsw.cpu().numpy(), last[1].grad.cpu().numpy(), GRAD_RTOL, "d scale_weights (last layer)")
