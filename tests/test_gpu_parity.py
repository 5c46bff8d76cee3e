"""GPU parity tests: the HIP path (through the C ABI, via opus_amd.csrc.wrapper) against
the golden fixtures and the CPU oracle.  Bars: forward 1e-5 relative, gradients 1e-4
relative (helpers.FWD_RTOL / GRAD_RTOL), relative to each tensor's max magnitude."""
import numpy as np
import pytest
import torch

from helpers import (FWD_RTOL, GRAD_RTOL, REAL_LEVELS, TINY_LEVELS, assert_close,
                     check_dfeat_digest, golden_feats, load_golden, random_case)

pytestmark = pytest.mark.gpu

DENSE = ["g1_tiny4", "g2_edges", "g2_p32", "g3_tiny5"]
REAL = ["g4_real_p2", "g4_real_p4"]
ALGOS = ["atomic", "owner"]


@pytest.fixture(scope="module")
def wr():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from opus_amd.csrc import wrapper
    assert wrapper.MSMV_CUDA, "libopus_msmv.so must be loaded on the GPU box"
    return wrapper


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_hip(wr, feats, loc, w, grad_out=None, algo=None, flags=0):
    f = [dev(x) for x in feats]
    l, ww = dev(loc), dev(w)
    out = wr.msmv_forward(f, l, ww, flags=flags)
    if grad_out is None:
        return out.cpu().numpy()
    dfeats, dloc, dw = wr.msmv_backward(dev(grad_out), f, l, ww, algo=algo)
    torch.cuda.synchronize()
    return (out.cpu().numpy(), [d.cpu().numpy() for d in dfeats], dloc.cpu().numpy(), dw.cpu().numpy())


def supports_owner(feats):
    return feats[0].shape[-1] == 64 and len(feats) in (4, 5)


# ---------------------------------------------------------------- golden fixtures
@pytest.mark.parametrize("name", DENSE + REAL)
def test_forward_matches_golden(wr, name):
    g = load_golden(name)
    out = run_hip(wr, golden_feats(g), g["loc"], g["w"])
    assert_close(out, g["out"], FWD_RTOL, f"{name} forward")


@pytest.mark.parametrize("name", DENSE)
def test_backward_matches_golden(wr, name):
    g = load_golden(name)
    feats = golden_feats(g)
    _, dfeats, dloc, dw = run_hip(wr, feats, g["loc"], g["w"], g["grad_out"], algo="auto")
    for i, d in enumerate(dfeats):
        assert_close(d, g[f"dfeat{i}"], GRAD_RTOL, f"{name} dfeat{i}")
    assert_close(dloc[..., :2], g["dloc_xy"], GRAD_RTOL, f"{name} dloc xy")
    assert_close(dw, g["dw"], GRAD_RTOL, f"{name} dw")
    assert (dloc[..., 2] == 0).all()


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("name", REAL)
def test_backward_matches_golden_real_size(wr, name, algo):
    g = load_golden(name)
    feats = golden_feats(g)
    _, dfeats, dloc, dw = run_hip(wr, feats, g["loc"], g["w"], g["grad_out"], algo=algo)
    check_dfeat_digest(g, dfeats)
    assert_close(dloc[..., :2], g["dloc_xy"], GRAD_RTOL, f"{name} dloc xy")
    assert_close(dw, g["dw"], GRAD_RTOL, f"{name} dw")
    assert (dloc[..., 2] == 0).all()


# ---------------------------------------------------------------- oracle, seeded random
CASES = [
    # levels, B, N, C, Q, P, oob
    (REAL_LEVELS, 3, 6, 64, 50, 1, 0.25),
    (REAL_LEVELS, 3, 6, 64, 37, 2, 0.25),
    (REAL_LEVELS, 2, 6, 64, 33, 3, 0.25),
    (REAL_LEVELS, 2, 6, 64, 21, 4, 0.25),
    (REAL_LEVELS, 1, 6, 64, 9, 8, 0.5),
    (REAL_LEVELS + [(4, 11)], 2, 6, 64, 19, 2, 0.25),
    (REAL_LEVELS + [(4, 11)], 2, 6, 64, 11, 4, 0.25),
    (TINY_LEVELS, 2, 6, 8, 13, 3, 0.25),
    (TINY_LEVELS, 2, 3, 96, 7, 2, 0.25),
    (TINY_LEVELS[:3], 2, 2, 64, 7, 2, 0.25),
    (TINY_LEVELS[:1], 1, 1, 5, 4, 1, 0.25),
    (REAL_LEVELS + TINY_LEVELS, 1, 2, 16, 6, 2, 0.25),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"L{len(c[0])}B{c[1]}N{c[2]}C{c[3]}Q{c[4]}P{c[5]}")
def test_forward_backward_match_oracle(wr, oracle, case):
    levels, B, N, C, Q, P, oob = case
    feats, loc, w, grad_out = random_case(hash(case[1:]) % 10000, levels, B, N, C, Q, P, oob)
    ref_out = oracle.forward(feats, loc, w)
    ref_df, ref_dloc, ref_dw = oracle.backward(grad_out, feats, loc, w)
    algos = ALGOS if supports_owner(feats) else ["atomic"]
    for algo in algos:
        out, dfeats, dloc, dw = run_hip(wr, feats, loc, w, grad_out, algo=algo)
        assert_close(out, ref_out, FWD_RTOL, f"forward [{algo}]")
        for i, d in enumerate(dfeats):
            assert_close(d, ref_df[i], GRAD_RTOL, f"dfeat{i} [{algo}]")
        assert_close(dloc, ref_dloc, GRAD_RTOL, f"dloc [{algo}]")
        assert_close(dw, ref_dw, GRAD_RTOL, f"dw [{algo}]")
    # the any-shape forward kernel on the same input
    out_generic = run_hip(wr, feats, loc, w, flags=1)
    assert_close(out_generic, ref_out, FWD_RTOL, "forward [generic kernel]")


def _check_all(wr, oracle, feats, loc, w, grad_out, algos=ALGOS):
    ref_out = oracle.forward(feats, loc, w)
    ref_df, ref_dloc, ref_dw = oracle.backward(grad_out, feats, loc, w)
    for algo in algos:
        out, dfeats, dloc, dw = run_hip(wr, feats, loc, w, grad_out, algo=algo)
        assert_close(out, ref_out, FWD_RTOL, f"forward [{algo}]")
        for i, d in enumerate(dfeats):
            assert_close(d, ref_df[i], GRAD_RTOL, f"dfeat{i} [{algo}]")
        assert_close(dloc, ref_dloc, GRAD_RTOL, f"dloc [{algo}]")
        assert_close(dw, ref_dw, GRAD_RTOL, f"dw [{algo}]")


def test_hot_spot_thousands_of_points_in_one_patch(wr, oracle):
    """Every point lands in the same few pixels of one view: one bin holds all entries and
    single patches receive thousands of hits (many scan chunks, many queue rounds)."""
    feats, loc, w, grad_out = random_case(31, REAL_LEVELS, B=2, N=6, C=64, Q=1500, P=2, oob_frac=0)
    rng = np.random.default_rng(5)
    loc[..., 0] = 0.40 + 0.01 * rng.random(loc.shape[:-1])
    loc[..., 1] = 0.70 + 0.01 * rng.random(loc.shape[:-1])
    loc[..., 2] = 2 / 5
    loc[1, :, :, 2] = (rng.integers(0, 2, loc.shape[1:3]) * 3) / 5      # slab 1: views 0 and 3 only
    _check_all(wr, oracle, feats, loc, w, grad_out)


def test_patches_shared_by_several_waves(wr, oracle):
    """Many points on small levels: the owner kernel puts 2 or 4 waves on one patch (each
    takes every K-th bin entry) and sums their partial tiles; includes ragged patches, slabs
    whose patch count is not a multiple of the patches per workgroup, and accumulate mode."""
    for levels, Q in (([(8, 22), (4, 11), (2, 6), (1, 3)], 2000), ([(16, 44), (8, 22), (5, 13), (3, 3)], 1200)):
        feats, loc, w, grad_out = random_case(60 + Q, levels, B=2, N=6, C=64, Q=Q, P=2, oob_frac=0.1)
        _check_all(wr, oracle, feats, loc, w, grad_out)
    f = [dev(x) for x in feats]
    pre = [torch.full_like(x, -2.0) for x in f]
    wr.msmv_backward(dev(grad_out), f, dev(loc), dev(w), dfeats=pre, algo="owner")
    ref_df, _, _ = oracle.backward(grad_out, feats, loc, w)
    for i, d in enumerate(pre):
        assert_close(d.cpu().numpy(), ref_df[i] - 2.0, GRAD_RTOL, f"shared patches, accumulate, dfeat{i}")


def test_odd_level_sizes_and_partial_patches(wr, oracle):
    levels = [(13, 31), (5, 7), (3, 9), (1, 1)]
    for P in (1, 2, 4, 5):
        feats, loc, w, grad_out = random_case(40 + P, levels, B=2, N=3, C=64, Q=40, P=P, oob_frac=0.3)
        _check_all(wr, oracle, feats, loc, w, grad_out)


def test_locations_on_band_and_pixel_boundaries(wr, oracle):
    """Coordinates exactly on, one ulp below and one ulp above every bin-band boundary (k/32 in x,
    k/8 in y) and every pixel boundary of every level: the sort's band of a point and the band
    interval a patch scans are computed differently (float product vs integer division), and a
    disagreement would drop a contribution from d feat."""
    levels = REAL_LEVELS
    def candidates(bands, sizes):
        v = [k / bands for k in range(bands + 1)]
        for n in sizes:
            v += [i / (n - 1) for i in range(n)]
        v = np.asarray(v, dtype=np.float32)
        return np.unique(np.concatenate([v, np.nextafter(v, np.float32(-1)), np.nextafter(v, np.float32(2))]))
    xs = candidates(32, [w_ for _, w_ in levels])
    ys = candidates(8, [h for h, _ in levels])
    rng = np.random.default_rng(88)
    B, N, C, P = 2, 6, 64, 2
    Q = 1200
    feats, loc, w, grad_out = random_case(89, levels, B, N, C, Q, P, oob_frac=0.0)
    loc[..., 0] = rng.choice(xs, size=loc.shape[:-1])
    loc[..., 1] = rng.choice(ys, size=loc.shape[:-1])
    # half of the points: one coordinate on a boundary, the other anywhere
    free = rng.random(loc.shape[:-1]) < 0.5
    axis = rng.integers(0, 2, loc.shape[:-1])
    for ax in (0, 1):
        m = free & (axis == ax)
        loc[..., ax][m] = rng.random(int(m.sum())).astype(np.float32)
    _check_all(wr, oracle, feats, loc, w, grad_out)


def test_axes_of_thousands_of_pixels(wr, oracle):
    """Levels with one very long axis (beyond 4096 pixels the owner kernel widens its band
    interval by a band on each side instead of relying on exact rounding), points spread over
    the whole axis and clustered at both ends."""
    levels = [(3, 5000), (4500, 2), (7, 4097), (1, 4096)]
    feats, loc, w, grad_out = random_case(70, levels, B=1, N=2, C=64, Q=300, P=2, oob_frac=0.2)
    loc[0, :40, :, 0] *= 1e-3                      # first pixels of the long axes
    loc[0, 40:80, :, 0] = 1.0 - loc[0, 40:80, :, 0] * 1e-3
    loc[0, 80:120, :, 1] *= 1e-3
    loc[0, 120:160, :, 1] = 1.0 - loc[0, 120:160, :, 1] * 1e-3
    _check_all(wr, oracle, feats, loc, w, grad_out)


def test_many_views_and_max_points(wr, oracle):
    # 16 views is the most the owner algorithm's bin table holds; 17 must use the atomic one
    feats, loc, w, grad_out = random_case(50, TINY_LEVELS, B=1, N=16, C=64, Q=30, P=32, oob_frac=0.2)
    _check_all(wr, oracle, feats, loc, w, grad_out)
    feats, loc, w, grad_out = random_case(51, TINY_LEVELS, B=1, N=17, C=64, Q=12, P=2, oob_frac=0.2)
    _check_all(wr, oracle, feats, loc, w, grad_out, algos=["atomic", "auto"])
    with pytest.raises(RuntimeError, match="algorithm"):
        run_hip(wr, feats, loc, w, grad_out, algo="owner")


@pytest.mark.parametrize("P", [1, 2, 3, 4])
def test_backward_ignores_what_earlier_kernels_left_in_lds(wr, oracle, P):
    """LDS is not cleared between kernels.  The d-feat kernel pads its hit queue with null
    records (weights 0): the gradient rows they point to must be loaded like the others, or
    0 x stale LDS contents (NaN here: a tiled transpose of a NaN tensor runs first, on every CU)
    ends up in d feat.  Odd hit counts per round are the case that matters."""
    from opus_amd import opus_sampling
    B, N, C, Q = 2, 6, 64, 37
    feats, loc, w, grad_out = random_case(4200 + P, REAL_LEVELS, B, N, C, Q, P, oob_frac=0.1)
    f = [dev(x) for x in feats]
    l, ww, g = dev(loc), dev(w), dev(grad_out)
    ref = oracle.backward(grad_out, feats, loc, w)
    poison = torch.full((1, 8 * 6, 4 * 64, 64, 176), float("nan"), device="cuda")
    for _ in range(3):
        opus_sampling.relayout_features([poison], 8, 6, 4)
        dfeats, dloc, dw = wr.msmv_backward(g, f, l, ww, algo="owner")
        torch.cuda.synchronize()
        for i, (a, r) in enumerate(zip(dfeats, ref[0])):
            assert_close(a.cpu().numpy(), r, GRAD_RTOL, f"dfeat{i} after an LDS-poisoning kernel, P={P}")
        assert_close(dloc.cpu().numpy()[..., :2], ref[1][..., :2], GRAD_RTOL, "dloc")
        assert_close(dw.cpu().numpy(), ref[2], GRAD_RTOL, "dw")


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("P", [1, 2, 3, 4])
def test_backward_overwrites_poisoned_buffers(wr, oracle, algo, P):
    """Outputs pre-filled with NaN and a workspace pre-filled with 0xFF bytes: the backward's
    contract is that it ignores what they hold on entry and writes every output element."""
    from opus_amd import _lib
    levels = REAL_LEVELS + [(4, 11)]
    B, N, C, Q = 3, 5, 64, 90
    feats, loc, w, grad_out = random_case(4300 + P, levels, B, N, C, Q, P, oob_frac=0.5)
    loc[0, :5, :, 2] = 7.0                        # some points without a valid view
    f = [dev(x) for x in feats]
    l, ww, g = dev(loc), dev(w), dev(grad_out)
    flags = _lib.BWD_OWNER if algo == "owner" else _lib.BWD_ATOMIC
    dfeats = [torch.full_like(x, float("nan")) for x in f]
    dloc, dw = torch.full_like(l, float("nan")), torch.full_like(ww, float("nan"))
    ws = torch.full((max(wr.backward_workspace_bytes(f, l, ww, flags), 16),), 0xFF, dtype=torch.uint8, device="cuda")
    wr.msmv_backward_into(g, f, l, ww, dfeats, dloc, dw, workspace=ws, flags=flags)
    ref = oracle.backward(grad_out, feats, loc, w)
    for i, (a, r) in enumerate(zip(dfeats, ref[0])):
        assert_close(a.cpu().numpy(), r, GRAD_RTOL, f"dfeat{i}, poisoned buffers, {algo}, P={P}", atol_floor=1e-30)
    assert_close(dloc.cpu().numpy()[..., :2], ref[1][..., :2], GRAD_RTOL, "dloc")
    assert float(dloc[..., 2].abs().max()) == 0.0
    assert_close(dw.cpu().numpy(), ref[2], GRAD_RTOL, "dw")


def test_out_of_range_view_and_nan_location(wr, oracle):
    feats, loc, w, grad_out = random_case(5, REAL_LEVELS, B=1, N=6, C=64, Q=8, P=2, oob_frac=0)
    loc[0, 0, 0, 2] = 1.3
    loc[0, 1, 1, 2] = -0.2
    loc[0, 2, 0, 0] = np.nan
    loc[0, 3, 1, 1] = np.inf
    ref_out = oracle.forward(feats, loc, w)
    for algo in ALGOS:
        out, dfeats, dloc, dw = run_hip(wr, feats, loc, w, grad_out, algo=algo)
        assert_close(out, ref_out, FWD_RTOL, "forward")
        assert (out[0, 0, :, 0] == 0).all() and (out[0, 2, :, 0] == 0).all()
        assert (dloc[0, 0, 0] == 0).all() and (dw[0, 2, 0] == 0).all()
        assert all(np.isfinite(d).all() for d in dfeats)


@pytest.mark.parametrize("Q,P", [(700, 1), (1700, 2), (1500, 1)], ids=["one-pass sort", "two-pass sort", "P1 forward"])
def test_slabs_with_many_points_in_no_view(wr, oracle, Q, P):
    """Points without a valid view are the tail entries of their slab's sort: a slab where every
    point is one, a slab where every other point is one, and an ordinary slab - zero output rows,
    zero gradients, d feats from the valid points only, from both sort kernels (a slab of up to
    3072 points is sorted by one workgroup, larger ones in two passes); with one point per query
    the bin-ordered forward and a backward on its sort run over the same entries."""
    feats, loc, w, grad_out = random_case(77 + Q, REAL_LEVELS, B=3, N=6, C=64, Q=Q, P=P)
    loc[0, :, :, 2] = 1.4                        # no such view anywhere in slab 0
    loc[1, ::2, :, 2] = np.nan                   # every other query of slab 1
    grad_out[0, 0] = np.inf                      # (a gradient row nobody may read a number from)
    ref_out = oracle.forward(feats, loc, w)
    g_fin = grad_out.copy()
    g_fin[0, 0] = 0.0
    ref_df, ref_dloc, ref_dw = oracle.backward(g_fin, feats, loc, w)
    out, dfeats, dloc, dw = run_hip(wr, feats, loc, w, grad_out, algo="owner")
    assert_close(out, ref_out, FWD_RTOL, "forward")
    assert (out[0] == 0).all() and (out[1, ::2] == 0).all()
    assert (dloc[0] == 0).all() and (dw[0] == 0).all() and (dloc[1, ::2] == 0).all() and (dw[1, ::2] == 0).all()
    for i in range(4):
        assert (dfeats[i][0] == 0).all()
        assert_close(dfeats[i], ref_df[i], GRAD_RTOL, f"dfeat{i}")
    assert_close(dloc[..., :2], ref_dloc[..., :2], GRAD_RTOL, "dloc")
    assert_close(dw, ref_dw, GRAD_RTOL, "dw")
    if P == 1:
        f, l, ww = [dev(x) for x in feats], dev(loc), dev(w)
        res = wr.msmv_forward_sorted(f, l, ww)
        assert res is not None
        o2, ws = res
        assert torch.equal(o2, wr.msmv_forward(f, l, ww))
        df2, dl2, dw2 = wr.msmv_backward(dev(grad_out), f, l, ww, sorted_ws=ws)
        assert torch.equal(dl2.cpu(), torch.from_numpy(dloc)) and torch.equal(dw2.cpu(), torch.from_numpy(dw))


def test_slab_larger_than_the_sort_s_workgroups_take_in_one_round(wr, oracle):
    """A slab of more than 16 x 1024 points: the sort's workgroups (at most 16 per slab) walk their
    share in several rounds.  Backward against the oracle, bin-ordered forward against the plain one."""
    levels = [(16, 44), (8, 22), (4, 11), (2, 6)]
    feats, loc, w, grad_out = random_case(4242, levels, B=2, N=6, C=64, Q=20011, P=1, oob_frac=0.1)
    loc[1, 5::7, :, 2] = -3.0                     # some points without a valid view, spread over the rounds
    ref_df, ref_dloc, ref_dw = oracle.backward(grad_out, feats, loc, w)
    out, dfeats, dloc, dw = run_hip(wr, feats, loc, w, grad_out, algo="owner")
    assert_close(out, oracle.forward(feats, loc, w), FWD_RTOL, "forward")
    for i in range(4):
        assert_close(dfeats[i], ref_df[i], GRAD_RTOL, f"dfeat{i}")
    assert_close(dloc[..., :2], ref_dloc[..., :2], GRAD_RTOL, "dloc")
    assert_close(dw, ref_dw, GRAD_RTOL, "dw")
    f, l, ww = [dev(x) for x in feats], dev(loc), dev(w)
    o2, ws = wr.msmv_forward_sorted(f, l, ww)
    assert torch.equal(o2.cpu(), torch.from_numpy(out))
    # every slot of the sort holds a different point of its slab
    lib = __import__("opus_amd._lib", fromlist=["x"]).load()
    need = lib.opus_msmv_backward_workspace_bytes(4, 2, 64, 6, 20011, 1, 2)
    ids = ws[need - 2 * 20011 * 32:].view(torch.int32).view(2 * 20011, 8)[:, 0].cpu().numpy()
    assert sorted(ids[:20011]) == list(range(20011)) and sorted(ids[20011:]) == list(range(20011, 2 * 20011))


# ---------------------------------------------------------------- API behaviour
def test_autograd_functions_and_dispatch(wr, oracle):
    feats, loc, w, grad_out = random_case(21, REAL_LEVELS, B=2, N=6, C=64, Q=10, P=2)
    f = [dev(x).requires_grad_(True) for x in feats]
    l, ww = dev(loc).requires_grad_(True), dev(w).requires_grad_(True)
    out = wr.msmv_sampling(f, l, ww)
    assert out.shape == (2, 10, 64, 2) and out.is_contiguous()
    # a non-contiguous upstream gradient is made contiguous (reference wrapper.py:56)
    g = dev(grad_out.transpose(0, 2, 1, 3).copy()).permute(0, 2, 1, 3)
    assert not g.is_contiguous()
    out.backward(g)
    ref_df, ref_dloc, ref_dw = oracle.backward(grad_out, feats, loc, w)
    for i in range(4):
        assert_close(f[i].grad.cpu().numpy(), ref_df[i], GRAD_RTOL, f"dfeat{i}")
    assert_close(l.grad.cpu().numpy(), ref_dloc, GRAD_RTOL, "dloc")
    assert_close(ww.grad.cpu().numpy(), ref_dw, GRAD_RTOL, "dw")
    out2 = wr.MSMVSamplingC2345.apply(*[x.detach() for x in f], l.detach(), ww.detach())
    assert torch.equal(out2, out.detach())
    with pytest.raises(TypeError):
        wr.MSMVSamplingC23456.apply(*[x.detach() for x in f], l.detach(), ww.detach())


@pytest.mark.parametrize("algo", ["auto", "atomic"])
@pytest.mark.parametrize("which", ["feats", "points", "weights only"])
def test_backward_computes_only_the_gradient_groups_asked_for(wr, oracle, which, algo):
    """Frozen feature maps (or constant locations / weights): autograd asks for one group of
    gradients only; the other group's pass is skipped where the owner algorithm applies (P = 1: on
    the sort the bin-ordered forward left; P = 2: on a sort of its own; C = 8: the atomic
    algorithm computes everything and the unwanted part is dropped).  The wanted gradients are the
    oracle's either way."""
    from opus_amd import opus_sampling  # noqa: F401  (the rows Function below lives in the wrapper)
    wr.set_backward_algo(algo)
    try:
        for levels, C, P in ((REAL_LEVELS, 64, 1), (REAL_LEVELS, 64, 2), (TINY_LEVELS, 8, 3)):
            feats, loc, w, grad_out = random_case(77 + P, levels, B=2, N=6, C=C, Q=23, P=P)
            ref_df, ref_dloc, ref_dw = oracle.backward(grad_out, feats, loc, w)
            f = [dev(x).requires_grad_(which == "feats") for x in feats]
            l = dev(loc).requires_grad_(which == "points")
            ww = dev(w).requires_grad_(which in ("points", "weights only"))
            out = wr.msmv_sampling(f, l, ww)
            out.backward(dev(grad_out))
            torch.cuda.synchronize()
            if which == "feats":
                for i in range(4):
                    assert_close(f[i].grad.cpu().numpy(), ref_df[i], GRAD_RTOL, f"dfeat{i} alone, P={P}")
                assert l.grad is None and ww.grad is None
            else:
                assert all(x.grad is None for x in f)
                assert_close(ww.grad.cpu().numpy(), ref_dw, GRAD_RTOL, f"dw without d feats, P={P}")
                if which == "points":
                    assert_close(l.grad.cpu().numpy(), ref_dloc, GRAD_RTOL, f"dloc without d feats, P={P}")
                else:
                    assert l.grad is None
        # the raw form says what was computed: a group that was not asked for comes back as None
        feats, loc, w, grad_out = random_case(5, REAL_LEVELS, B=1, N=6, C=64, Q=9, P=2)
        f, l, ww, g = [dev(x) for x in feats], dev(loc), dev(w), dev(grad_out)
        df, dl, dww = wr.msmv_backward(g, f, l, ww, need=(False, True))
        assert df == [None] * 4 and dl is not None and dww is not None
        df, dl, dww = wr.msmv_backward(g, f, l, ww, need=(True, False))
        assert dl is None and dww is None
        ref_df, _, _ = oracle.backward(grad_out, feats, loc, w)
        for i in range(4):
            assert_close(df[i].cpu().numpy(), ref_df[i], GRAD_RTOL, f"dfeat{i}, raw form")
        assert wr.msmv_backward(g, f, l, ww, need=(False, False)) == ([None] * 4, None, None)
    finally:
        wr.set_backward_algo("auto")


def test_rows_function_with_frozen_feature_maps(wr, oracle):
    """The caller-layout Function with feature maps that need no gradient: sort + gather only."""
    T, G, Bs, Q, P = 2, 2, 1, 17, 2
    feats, loc, w, grad_out = random_case(31, REAL_LEVELS, B=Bs * T * G, N=6, C=64, Q=Q, P=P)
    _, ref_dloc, ref_dw = oracle.backward(grad_out, feats, loc, w)
    f = [dev(x) for x in feats]
    l, ww = dev(loc).requires_grad_(True), dev(w).requires_grad_(True)
    out = wr.msmv_sampling_rows(f, l, ww, T, G)
    # grad_out [B', Q, C, P] -> the caller's [Bs, Q, G, T*P, C] (models/opus_sampling.py:119-122)
    g_rows = dev(grad_out).reshape(Bs, T, G, Q, 64, P).permute(0, 3, 2, 1, 5, 4).reshape(Bs, Q, G, T * P, 64)
    out.backward(g_rows.contiguous())
    torch.cuda.synchronize()
    assert_close(l.grad.cpu().numpy(), ref_dloc, GRAD_RTOL, "dloc, frozen feature maps")
    assert_close(ww.grad.cpu().numpy(), ref_dw, GRAD_RTOL, "dw, frozen feature maps")


def test_precondition_errors(wr):
    feats, loc, w, _ = random_case(3, TINY_LEVELS, B=1, N=6, C=8, Q=4, P=2)
    f = [dev(x) for x in feats]
    l, ww = dev(loc), dev(w)
    with pytest.raises(RuntimeError, match="contiguous"):
        wr.msmv_sampling([f[0].transpose(2, 3)] + f[1:], l, ww)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        wr.msmv_sampling(f, l.cpu(), ww)
    with pytest.raises(RuntimeError, match="float32"):
        wr.msmv_sampling(f, l.double(), ww)
    # P = 33 exceeds MAX_POINT (msmv_sampling.cpp:125)
    feats, loc, w, _ = random_case(3, TINY_LEVELS, B=1, N=6, C=8, Q=2, P=33)
    with pytest.raises(RuntimeError, match="num_point exceed limits"):
        wr.msmv_sampling([dev(x) for x in feats], dev(loc), dev(w))


def test_c_abi_error_codes(wr):
    import ctypes
    from opus_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    hw = _lib.int_array([4, 4])
    ptrs = _lib.ptr_array([0])
    assert lib.opus_msmv_forward(0, ptrs, hw, null, null, 1, 8, 6, 1, 1, null, null) == -1
    assert lib.opus_msmv_forward(1, ptrs, hw, null, null, 1, 8, 6, 1, 33, null, null) == -2
    assert lib.opus_msmv_forward(1, ptrs, hw, null, null, 1, 0, 6, 1, 1, null, null) == -3
    assert lib.opus_msmv_forward(1, ptrs, hw, null, null, 1, 8, 6, 1, 1, null, null) == -4
    assert lib.opus_msmv_forward(1, ptrs, hw, null, null, 1 << 20, 64, 6, 1 << 10, 32, null, null) == -5
    assert lib.opus_msmv_forward(1, ptrs, hw, null, null, 0, 8, 6, 5, 1, null, null) == 0   # empty batch


def test_empty_query_set(wr):
    feats, loc, w, grad_out = random_case(4, REAL_LEVELS, B=1, N=6, C=64, Q=0, P=2)
    for algo in ALGOS:
        out, dfeats, dloc, dw = run_hip(wr, feats, loc, w, grad_out, algo=algo)
        assert out.shape == (1, 0, 64, 2)
        assert all((d == 0).all() for d in dfeats)


def test_non_default_stream_and_accumulate(wr, oracle):
    feats, loc, w, grad_out = random_case(9, REAL_LEVELS, B=1, N=6, C=64, Q=16, P=4)
    f = [dev(x) for x in feats]
    l, ww, g = dev(loc), dev(w), dev(grad_out)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out = wr.msmv_forward(f, l, ww)
        dfeats, dloc, dw = wr.msmv_backward(g, f, l, ww, algo="owner")
        # accumulate two more backwards onto the first, one per algorithm: d feats triple
        _, dloc_a, dw_a = wr.msmv_backward(g, f, l, ww, dfeats=dfeats, algo="atomic")
        _, dloc_o, dw_o = wr.msmv_backward(g, f, l, ww, dfeats=dfeats, algo="owner")
        # and onto a buffer that holds something else: with 64 points most patches are reached
        # by no point and must keep their contents
        pre = [torch.full_like(x, 3.0) for x in f]
        wr.msmv_backward(g, f, l, ww, dfeats=pre, algo="owner")
    s.synchronize()
    ref_df, ref_dloc, ref_dw = oracle.backward(grad_out, feats, loc, w)
    assert_close(out.cpu().numpy(), oracle.forward(feats, loc, w), FWD_RTOL, "forward on side stream")
    for i, d in enumerate(dfeats):
        assert_close(d.cpu().numpy(), 3 * ref_df[i], GRAD_RTOL, f"accumulated dfeat{i}")
        assert_close(pre[i].cpu().numpy(), 3.0 + ref_df[i], GRAD_RTOL, f"dfeat{i} onto a filled buffer")
    # d loc / d w are overwritten, never accumulated
    for dl, dww in ((dloc_a, dw_a), (dloc_o, dw_o)):
        assert_close(dl.cpu().numpy(), ref_dloc, GRAD_RTOL, "dloc with ACCUMULATE")
        assert_close(dww.cpu().numpy(), ref_dw, GRAD_RTOL, "dw with ACCUMULATE")


def test_sampling_4d_matches_golden(wr):
    """The caller restatement + HIP op against the reference's sampling_4d output."""
    from opus_amd import opus_sampling
    g = load_golden("g5_sampling4d")
    feats_cl = [dev(np.ascontiguousarray(g[f"feat_cf{i}"].transpose(0, 2, 3, 4, 1))) for i in range(4)]
    out = opus_sampling.sampling_4d(
        dev(g["sample_points"]), feats_cl, dev(g["scale_weights"]), dev(g["occ2img"]),
        int(g["image_h"]), int(g["image_w"]), int(g["num_views"]))
    assert_close(out.cpu().numpy(), g["out"], FWD_RTOL, "sampling_4d")


# ---------------------------------------------------------------- full-size properties
def _device_case(B, Q, P, seed, levels=REAL_LEVELS, N=6, C=64):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    feats = [torch.randn((B, N, h, w, C), generator=gen, device="cuda") for h, w in levels]
    xy = torch.rand((B, Q, P, 2), generator=gen, device="cuda") * 1.2 - 0.1
    view = torch.randint(0, N, (B, Q, P, 1), generator=gen, device="cuda").float() / (N - 1)
    loc = torch.cat([xy, view], -1).contiguous()
    w = torch.softmax(torch.randn((B, Q, P, len(levels)), generator=gen, device="cuda"), -1)
    g = torch.randn((B, Q, C, P), generator=gen, device="cuda")
    return feats, loc, w, g


@pytest.mark.parametrize("shape", [(32, 4800, 2), (32, 2400, 2), (32, 1200, 2), (32, 600, 4)],
                         ids=["OPUS-L", "OPUS-M", "OPUS-S", "OPUS-T"])
def test_full_size_adjoint_identities(wr, shape):
    """At the benchmark sizes the oracle is too slow; use what linearity gives for free:
    the op is linear in feats and in weights, so <g, F(feats, w)> = <dfeats, feats> = <dw, w>,
    and the two backward algorithms must agree with each other."""
    B, Q, P = shape
    feats, loc, w, g = _device_case(B, Q, P, seed=1234)
    out = wr.msmv_forward(feats, loc, w)
    lhs = (out.double() * g.double()).sum().item()
    scale = (out.double().abs() * g.double().abs()).sum().item()
    res = {}
    for algo in ALGOS:
        dfeats, dloc, dw = wr.msmv_backward(g, feats, loc, w, algo=algo)
        via_feats = sum((d.double() * f.double()).sum().item() for d, f in zip(dfeats, feats))
        via_w = (dw.double() * w.double()).sum().item()
        assert abs(via_feats - lhs) <= 1e-5 * scale, (algo, via_feats, lhs)
        assert abs(via_w - lhs) <= 1e-5 * scale, (algo, via_w, lhs)
        assert (dloc[..., 2] == 0).all()
        res[algo] = (dfeats, dloc, dw)
    for i in range(len(feats)):
        a, o = res["atomic"][0][i], res["owner"][0][i]
        assert (a - o).abs().max().item() <= GRAD_RTOL * a.abs().max().item(), f"dfeat{i} algos disagree"
    for k in (1, 2):
        a, o = res["atomic"][k], res["owner"][k]
        assert (a - o).abs().max().item() <= GRAD_RTOL * a.abs().max().item()


def test_full_size_forward_generic_vs_fast_and_subset_oracle(wr, oracle):
    B, Q, P = 32, 4800, 2
    feats, loc, w, g = _device_case(B, Q, P, seed=77)
    fast = wr.msmv_forward(feats, loc, w)
    generic = wr.msmv_forward(feats, loc, w, flags=1)
    assert (fast - generic).abs().max().item() <= FWD_RTOL * fast.abs().max().item()
    # the oracle on the last batch slab only (exercises the largest offsets)
    sl = slice(B - 1, B)
    ref = oracle.forward([f[sl].cpu().numpy() for f in feats], loc[sl].cpu().numpy(), w[sl].cpu().numpy())
    assert_close(fast[sl].cpu().numpy(), ref, FWD_RTOL, "last slab vs oracle")
    # backward of the last slab against the oracle, both algorithms
    ref_df, ref_dloc, ref_dw = oracle.backward(
        g[sl].cpu().numpy(), [f[sl].cpu().numpy() for f in feats],
        loc[sl].cpu().numpy(), w[sl].cpu().numpy())
    for algo in ALGOS:
        dfeats, dloc, dw = wr.msmv_backward(g, feats, loc, w, algo=algo)
        for i, d in enumerate(dfeats):
            assert_close(d[sl].cpu().numpy(), ref_df[i], GRAD_RTOL, f"dfeat{i} last slab [{algo}]")
        assert_close(dloc[sl].cpu().numpy(), ref_dloc, GRAD_RTOL, f"dloc last slab [{algo}]")
        assert_close(dw[sl].cpu().numpy(), ref_dw, GRAD_RTOL, f"dw last slab [{algo}]")


def test_forward_backward_replay_from_a_hip_graph(wr, oracle):
    """The library allocates nothing and never synchronises, so a forward + backward issued
    with caller-owned buffers can be captured into a HIP graph once and replayed on new data."""
    from opus_amd import _lib
    feats, loc, w, grad_out = random_case(77, REAL_LEVELS, B=2, N=6, C=64, Q=64, P=2, oob_frac=0.2)
    f = [dev(x) for x in feats]
    l, ww, g = dev(loc), dev(w), dev(grad_out)
    dfeats = [torch.empty_like(x) for x in f]
    dloc, dw = torch.empty_like(l), torch.empty_like(ww)
    ws = torch.empty(max(wr.backward_workspace_bytes(f, l, ww, _lib.BWD_OWNER), 16), dtype=torch.uint8, device="cuda")
    out = torch.empty((2, 64, 64, 2), device="cuda")

    def step():
        out.copy_(wr.msmv_forward(f, l, ww))
        wr.msmv_backward_into(g, f, l, ww, dfeats, dloc, dw, ws, flags=_lib.BWD_OWNER)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()                                   # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    # new inputs in the captured buffers, then replay
    feats2, loc2, w2, grad2 = random_case(78, REAL_LEVELS, B=2, N=6, C=64, Q=64, P=2, oob_frac=0.2)
    for dst, src in zip(f, feats2):
        dst.copy_(dev(src))
    l.copy_(dev(loc2)); ww.copy_(dev(w2)); g.copy_(dev(grad2))
    for t in dfeats + [dloc, dw, out]:
        t.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    assert_close(out.cpu().numpy(), oracle.forward(feats2, loc2, w2), FWD_RTOL, "replayed forward")
    ref_df, ref_dloc, ref_dw = oracle.backward(grad2, feats2, loc2, w2)
    for i, d in enumerate(dfeats):
        assert_close(d.cpu().numpy(), ref_df[i], GRAD_RTOL, f"replayed dfeat{i}")
    assert_close(dloc.cpu().numpy(), ref_dloc, GRAD_RTOL, "replayed dloc")
    assert_close(dw.cpu().numpy(), ref_dw, GRAD_RTOL, "replayed dw")


def test_offsets_beyond_32_bits(wr, oracle):
    """520 batch slabs of the real level sizes: feature offsets pass 2^31 elements from slab 496
    on (where the reference's int arithmetic wraps, msmv_sampling_forward.cu:127).  The last
    slabs are checked against the oracle, forward and backward, both algorithms."""
    B, N, C, Q, P = 520, 6, 64, 8, 2
    gen = torch.Generator(device="cuda").manual_seed(520)
    f = [torch.randn((B, N, h, w_, C), generator=gen, device="cuda") for h, w_ in REAL_LEVELS]
    assert f[0].numel() > 2 ** 31
    _, loc, w, grad_out = random_case(521, TINY_LEVELS, B, N, 1, Q, P, oob_frac=0.2)
    grad_out = np.random.default_rng(3).standard_normal((B, Q, C, P)).astype(np.float32)
    l, ww, g = dev(loc), dev(w), dev(grad_out)
    out = wr.msmv_forward(f, l, ww)
    tail = slice(B - 3, B)
    feats_tail = [x[tail].cpu().numpy() for x in f]
    ref_out = oracle.forward(feats_tail, loc[tail], w[tail])
    assert_close(out[tail].cpu().numpy(), ref_out, FWD_RTOL, "forward, slabs past 2^31")
    ref_df, ref_dloc, ref_dw = oracle.backward(grad_out[tail], feats_tail, loc[tail], w[tail])
    for algo in ALGOS:
        dfeats, dloc, dw = wr.msmv_backward(g, f, l, ww, algo=algo)
        for i, d in enumerate(dfeats):
            assert_close(d[tail].cpu().numpy(), ref_df[i], GRAD_RTOL, f"{algo} dfeat{i}, slabs past 2^31")
        assert_close(dloc[tail].cpu().numpy(), ref_dloc, GRAD_RTOL, f"{algo} dloc")
        assert_close(dw[tail].cpu().numpy(), ref_dw, GRAD_RTOL, f"{algo} dw")
        del dfeats


def test_randomised_shapes_against_the_oracle(wr, oracle):
    """Forty seeded random configurations (level count and sizes, views, queries, points per query,
    out-of-range share; C = 64 so that both backward algorithms apply), every tensor against the
    oracle; where the batch allows it, row mode against the same results re-ordered."""
    from opus_amd import opus_sampling
    rng = np.random.default_rng(2024)
    for case in range(40):
        L = int(rng.choice([4, 5]))
        levels = [(int(rng.integers(1, 41)), int(rng.integers(1, 41))) for _ in range(L)]
        B, N = int(rng.integers(1, 7)), int(rng.integers(1, 9))
        Q, P = int(rng.integers(1, 301)), int(rng.choice([1, 2, 3, 4, 5, 8]))
        feats, loc, w, grad_out = random_case(9000 + case, levels, B, N, 64, Q, P,
                                              oob_frac=float(rng.choice([0.0, 0.3, 0.8])))
        what = f"case {case}: levels {levels} B{B} N{N} Q{Q} P{P}"
        try:
            _check_all(wr, oracle, feats, loc, w, grad_out)
        except AssertionError as e:
            raise AssertionError(f"{what}: {e}") from e
        # row mode with (T, G) = (B, 1) and (1, B): the op's results in the caller's layout
        f = [dev(x) for x in feats]
        l, ww, g = dev(loc), dev(w), dev(grad_out)
        plain = wr.msmv_forward(f, l, ww)
        for T, G in ((B, 1), (1, B)):
            rows = wr.msmv_forward(f, l, ww, rows=(T, G))
            assert torch.equal(rows, opus_sampling.unpack(plain, 1, T, G).contiguous()), what
            g_rows = opus_sampling.unpack(g, 1, T, G).contiguous()
            df_r, dl_r, dw_r = wr.msmv_backward(g_rows, f, l, ww, algo="owner", rows=(T, G))
            df_p, dl_p, dw_p = wr.msmv_backward(g, f, l, ww, algo="owner")
            for a_, b_ in zip(df_r, df_p):
                assert_close(a_.cpu().numpy(), b_.cpu().numpy(), GRAD_RTOL, what + " row-mode dfeat")
            assert_close(dl_r.cpu().numpy(), dl_p.cpu().numpy(), 1e-5, what + " row-mode dloc")
            assert_close(dw_r.cpu().numpy(), dw_p.cpu().numpy(), 1e-5, what + " row-mode dw")


def test_non_finite_border_pixel_is_treated_as_the_reference_does(wr, oracle):
    """An out-of-image corner has the VALUE zero in the reference (msmv_sampling_forward.cu:48-67:
    the load is skipped), so an infinite feature next to the border reaches the output only through
    the corners that really lie on it - as +-inf, never as the NaN of 0 * inf.  All forward and
    backward kernels must agree with the oracle on where the non-finite values are."""
    feats, loc, w, grad_out = random_case(99, REAL_LEVELS, B=2, N=6, C=64, Q=40, P=2, oob_frac=0.0)
    w = np.abs(w) + 0.05
    w /= w.sum(-1, keepdims=True)
    w = w.astype(np.float32)
    H, W = REAL_LEVELS[0]
    feats[0][0, 2, 10, W - 1, 7] = np.inf           # right border
    feats[0][1, 4, 0, 33, 5] = -np.inf              # top border
    loc[0, :8, 0] = [1.0, (10 + 0.25) / (H - 1), 2 / 5]          # exactly on the last column: corner 2 is outside
    loc[0, 8:16, 0] = [1.0 + 0.3 / (W - 1), 10 / (H - 1), 2 / 5]  # beyond it, still in range
    loc[1, :8, 1] = [(33 + 0.5) / (W - 1), -0.4 / (H - 1), 4 / 5]  # above the first row
    ref_out = oracle.forward(feats, loc, w)
    assert np.isinf(ref_out).any() and not np.isnan(ref_out).any()
    _, ref_dloc, ref_dw = oracle.backward(grad_out, feats, loc, w)

    def same_non_finite(a, b, what):
        assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: NaN pattern"
        assert np.array_equal(np.isposinf(a), np.isposinf(b)) and np.array_equal(np.isneginf(a), np.isneginf(b)), \
            f"{what}: inf pattern"
        fin = np.isfinite(b)
        assert_close(a[fin], b[fin], GRAD_RTOL, what)

    for flags, name in ((0, "fast"), (1, "generic")):
        same_non_finite(run_hip(wr, feats, loc, w, flags=flags), ref_out, f"forward [{name}]")
    for algo in ALGOS:
        _, _, dloc, dw = run_hip(wr, feats, loc, w, grad_out, algo=algo)
        same_non_finite(dw, ref_dw, f"dw [{algo}]")
        same_non_finite(dloc, ref_dloc, f"dloc [{algo}]")


@pytest.mark.parametrize("mode", ["deferred", "accumulate"])
def test_full_size_six_layer_decoder_step(wr, oracle, mode):
    """The full decoder configuration at OPUS-L size: six layers (six point sets) sample the same
    feature maps (B' = 32 slabs, Q = 4800, P = 2), forward + backward with the layers' d feats
    summed into ONE buffer - deferred to a single multi-set pass, or accumulated layer by layer.
    The sum must equal the sum of six stand-alone backwards of the atomic algorithm (the other
    implementation) everywhere, and the oracle's sum over the six layers on the last slab."""
    from opus_amd import _lib
    B, Q, P, NL = 32, 4800, 2, 6
    feats, loc0, w0, g0 = _device_case(B, Q, P, seed=4321)
    layers = [(loc0, w0, g0)] + [_device_case(B, Q, P, seed=4322 + i)[1:] for i in range(NL - 1)]
    # reference sum: atomic algorithm, one layer at a time, accumulating (its own flag path)
    ref = [torch.zeros_like(f) for f in feats]
    ref_small = []
    for loc, w, g in layers:
        _, dloc, dw = wr.msmv_backward(g, feats, loc, w, algo="atomic", dfeats=ref)
        ref_small.append((dloc, dw))
    got = [torch.full_like(f, float("nan")) for f in feats]
    small = []
    if mode == "deferred":
        flags = _lib.BWD_OWNER | _lib.BWD_STAGE_BIN | _lib.BWD_STAGE_GATHER
        need = wr.backward_workspace_bytes(feats, loc0, w0, flags)
        kept = []
        for loc, w, g in layers:
            ws = torch.empty(need, dtype=torch.uint8, device="cuda")
            dloc, dw = torch.full_like(loc, float("nan")), torch.full_like(w, float("nan"))
            wr.msmv_backward_into(g, feats, loc, w, None, dloc, dw, workspace=ws, flags=flags)
            kept.append(ws)
            small.append((dloc, dw))
        wr.msmv_dfeat_multi([g for _, _, g in layers], kept, feats, Q, P, got)
    else:
        for i, (loc, w, g) in enumerate(layers):
            _, dloc, dw = wr.msmv_backward(g, feats, loc, w, algo="owner", dfeats=got if i else None,
                                           out_dfeats=None if i else got)
            small.append((dloc, dw))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got, ref)):
        assert torch.isfinite(a).all(), f"dfeat{i}: not every element written"
        assert (a - b).abs().max().item() <= GRAD_RTOL * b.abs().max().item(), f"dfeat{i} [{mode}] vs atomic sum"
    for (dl, dw), (rl, rw) in zip(small, ref_small):
        assert (dl - rl).abs().max().item() <= GRAD_RTOL * rl.abs().max().item()
        assert (dw - rw).abs().max().item() <= GRAD_RTOL * rw.abs().max().item()
    sl = slice(B - 1, B)
    f_np = [f[sl].cpu().numpy() for f in feats]
    total = None
    for loc, w, g in layers:
        df, _, _ = oracle.backward(g[sl].cpu().numpy(), f_np, loc[sl].cpu().numpy(), w[sl].cpu().numpy())
        total = df if total is None else [a + b for a, b in zip(total, df)]
    for i, d in enumerate(got):
        assert_close(d[sl].cpu().numpy(), total[i], GRAD_RTOL, f"dfeat{i} last slab, six layers [{mode}]")


def test_where_a_non_finite_gradient_lands(wr):
    """An inf in grad_out of one valid point: the reference (and the atomic algorithm) carry it to
    that point's corner pixels; the owner algorithm forms whole 4 x 8 patches as matrix products, so
    the channel goes non-finite on the patches around the point - never on another channel, view or
    batch slab - and everything that stays finite equals the atomic algorithm's result."""
    B, N, C, Q, P = 2, 6, 64, 300, 2
    feats, loc, w, gout = random_case(8800, REAL_LEVELS, B, N, C, Q, P, 0.0)
    b0, q0, p0, ch = 1, 17, 1, 5
    loc[b0, q0, p0] = (0.4, 0.6, 2 / (N - 1))
    gout[b0, q0, ch, p0] = np.inf
    f, l, ww, g = [dev(x) for x in feats], dev(loc), dev(w), dev(gout)
    own = wr.msmv_backward(g, f, l, ww, algo="owner")
    ato = wr.msmv_backward(g, f, l, ww, algo="atomic")
    for lvl, (a, o) in enumerate(zip(ato[0], own[0])):
        a, o = a.cpu().numpy(), o.cpu().numpy()
        bad_a, bad_o = ~np.isfinite(a), ~np.isfinite(o)
        assert bad_a.any() and (bad_o | ~bad_a).all(), f"level {lvl}: the corner pixels themselves"
        where = np.argwhere(bad_o)                       # (b, view, y, x, channel)
        assert (where[:, 0] == b0).all() and (where[:, 1] == 2).all() and (where[:, 4] == ch).all()
        H, W = REAL_LEVELS[lvl]
        y, x = 0.6 * (H - 1), 0.4 * (W - 1)
        # within the patches that scan the point's image bin (a bin is 1/8 x 1/32 of the view, a patch
        # 4 x 8 pixels and scans one pixel beyond its edges)
        assert (np.abs(where[:, 2] - y) <= H / 8 + 6).all() and (np.abs(where[:, 3] - x) <= W / 32 + 10).all()
        scale = np.abs(a[~bad_a]).max()
        assert np.abs(a[~bad_o] - o[~bad_o]).max() <= GRAD_RTOL * scale
    for a, o, what in ((ato[1], own[1], "dloc"), (ato[2], own[2], "dw")):
        a, o = a.cpu().numpy(), o.cpu().numpy()
        bad = ~np.isfinite(o)
        assert np.array_equal(bad, ~np.isfinite(a)), what
        only = np.zeros_like(bad)
        only[b0, q0, p0] = True
        assert not (bad & ~only).any(), f"{what}: other points stay finite"
        assert_close(np.where(bad, 0, o), np.where(bad, 0, a), GRAD_RTOL, what)


# ---------------------------------------------------------------- every hit loop of the d-feat kernel
_HIT_LOOP_CHECK = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, {tests!r} + "/golden")
sys.path.insert(0, {tests!r})
sys.path.insert(0, {root!r})
from helpers import GRAD_RTOL, REAL_LEVELS, assert_close, random_case
from opus_amd.csrc import wrapper as wr
from opus_amd import _lib
from oracle import msmv_oracle as oracle
oracle.build()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for levels, P, Q in ((REAL_LEVELS, 1, 300), (REAL_LEVELS, 2, 211), (REAL_LEVELS + [(4, 11)], 2, 97)):
    # one set, in the op's layout
    feats, loc, w, g = random_case(900 + P + Q, levels, B=2, N=6, C=64, Q=Q, P=P)
    ref_df, _, _ = oracle.backward(g, feats, loc, w)
    f, l, ww = [dev(x) for x in feats], dev(loc), dev(w)
    df, _, _ = wr.msmv_backward(dev(g), f, l, ww, algo="owner")
    for i, d in enumerate(df):
        assert_close(d.cpu().numpy(), ref_df[i], GRAD_RTOL, "dfeat%d, one set, P=%d" % (i, P))
    # three sets in one pass (deferred d feats)
    sets = [random_case(1900 + P + Q + k, levels, B=2, N=6, C=64, Q=Q, P=P) for k in range(3)]
    total = None
    gs, wss = [], []
    for _, lo, we, go in sets:
        rdf, _, _ = oracle.backward(go, feats, lo, we)
        total = rdf if total is None else [a + b for a, b in zip(total, rdf)]
        lo_d, we_d, go_d = dev(lo), dev(we), dev(go)
        ws = torch.empty(wr.backward_workspace_bytes(f, lo_d, we_d, _lib.BWD_OWNER), dtype=torch.uint8, device="cuda")
        wr.msmv_backward_into(go_d, f, lo_d, we_d, None, torch.empty_like(lo_d), torch.empty_like(we_d), ws,
                              flags=_lib.BWD_OWNER | _lib.BWD_STAGE_BIN | _lib.BWD_STAGE_GATHER)
        gs.append(go_d)
        wss.append(ws)
    out = [torch.full_like(x, float("nan")) for x in f]
    wr.msmv_dfeat_multi(gs, wss, f, Q, P, out)
    torch.cuda.synchronize()
    for i, d in enumerate(out):
        assert_close(d.cpu().numpy(), total[i], GRAD_RTOL, "dfeat%d, three sets, P=%d" % (i, P))
print("hit loop ok")
"""


@pytest.mark.parametrize("hit_loop", ["0", "1", "2"])
def test_every_hit_loop_of_the_dfeat_kernel(hit_loop):
    """The d-feat kernel picks one of three hit loops by layout and size; OPUS_MSMV_DFEAT_HL (read once
    per process) forces one wherever it applies.  Each loop, forced in a process of its own, against the
    oracle: one set and three sets in one pass, P = 1 and 2, 4 and 5 levels."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    env = dict(os.environ, OPUS_MSMV_DFEAT_HL=hit_loop)
    p = subprocess.run([sys.executable, "-c", _HIT_LOOP_CHECK.format(tests=here, root=root)], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "hit loop ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
