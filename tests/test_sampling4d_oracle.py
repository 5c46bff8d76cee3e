"""Pins the caller-side oracle (oracle/sampling4d_oracle.py) against what the reference's own
sampling_4d produced (tests/golden/g5_sampling4d.npz, g6_sampling4d_grad.npz)."""
import numpy as np
import torch

from helpers import FWD_RTOL, GRAD_RTOL, assert_close, load_golden


def _g5_feats_cl(g):
    return [np.ascontiguousarray(g[f"feat_cf{i}"].transpose(0, 2, 3, 4, 1)) for i in range(4)]


def test_project_pack_oracle_matches_golden():
    from oracle import sampling4d_oracle as so
    g = load_golden("g5_sampling4d")
    loc, w = so.project_pack(g["sample_points"], g["scale_weights"], g["occ2img"],
                             int(g["image_h"]), int(g["image_w"]), int(g["num_views"]))
    assert np.array_equal(loc[..., 2], g["op_loc"][..., 2])          # same camera, exactly
    assert_close(loc, g["op_loc"], 1e-6, "oracle packed locations")
    assert np.array_equal(w, g["op_w"])


def test_sampling4d_oracle_chain_matches_golden(oracle):
    """project_pack -> op -> unpack forward, and the whole chain's gradients."""
    from oracle import sampling4d_oracle as so
    g, gg = load_golden("g5_sampling4d"), load_golden("g6_sampling4d_grad")
    B, Q, T, G, P, _ = g["sample_points"].shape
    H, W, N = int(g["image_h"]), int(g["image_w"]), int(g["num_views"])
    feats = _g5_feats_cl(g)
    loc, w = so.project_pack(g["sample_points"], g["scale_weights"], g["occ2img"], H, W, N)
    out = so.unpack(oracle.forward(feats, loc, w), B, T, G)
    assert_close(out, g["out"], FWD_RTOL, "sampling_4d forward")

    gop = so.unpack_backward(gg["grad_out"], B, T, G)
    dfeats, dloc, dw = oracle.backward(gop, feats, loc, w)
    dpts, dsw = so.project_pack_backward(dloc, dw, g["sample_points"], g["occ2img"], loc, H, W, N)
    assert_close(dpts, gg["d_sample_points"], GRAD_RTOL, "d sample_points")
    assert_close(dsw, gg["d_scale_weights"], GRAD_RTOL, "d scale_weights")
    for i, d in enumerate(dfeats):
        assert_close(d.transpose(0, 4, 1, 2, 3), gg[f"d_feat_cf{i}"], GRAD_RTOL, f"d feat {i}")


def test_tensor_op_restatement_matches_oracle_gradients():
    """autograd through project_and_pack_torch == the oracle's analytic backward."""
    from opus_amd import opus_sampling
    from oracle import sampling4d_oracle as so
    g = load_golden("g5_sampling4d")
    H, W, N = int(g["image_h"]), int(g["image_w"]), int(g["num_views"])
    pts = torch.from_numpy(g["sample_points"]).requires_grad_(True)
    sw = torch.from_numpy(g["scale_weights"]).requires_grad_(True)
    loc, w = opus_sampling.project_and_pack_torch(pts, sw, torch.from_numpy(g["occ2img"]), H, W, N)
    rng = np.random.default_rng(1)
    dloc = rng.standard_normal(tuple(loc.shape)).astype(np.float32)
    dw = rng.standard_normal(tuple(w.shape)).astype(np.float32)
    torch.autograd.backward([loc, w], [torch.from_numpy(dloc), torch.from_numpy(dw)])
    dpts, dsw = so.project_pack_backward(dloc, dw, g["sample_points"], g["occ2img"],
                                         loc.detach().numpy(), H, W, N)
    assert_close(pts.grad.numpy(), dpts, 1e-5, "d sample_points")
    assert np.array_equal(sw.grad.numpy(), dsw)
