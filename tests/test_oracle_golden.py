"""Pins the CPU oracle (oracle/msmv_oracle.c, oracle/torch_ref.py) against the golden
fixtures generated from the reference's own fallback (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from helpers import (FWD_RTOL, GRAD_RTOL, assert_close, check_dfeat_digest, golden_feats,
                     load_golden, random_case, TINY_LEVELS)

DENSE = ["g1_tiny4", "g2_edges", "g2_p32", "g3_tiny5"]
REAL = ["g4_real_p2", "g4_real_p4"]


@pytest.mark.parametrize("name", DENSE + REAL)
def test_oracle_forward_matches_golden(oracle, name):
    g = load_golden(name)
    out = oracle.forward(golden_feats(g), g["loc"], g["w"])
    assert_close(out, g["out"], FWD_RTOL, f"{name} forward")


@pytest.mark.parametrize("name", DENSE)
def test_oracle_backward_matches_golden(oracle, name):
    g = load_golden(name)
    feats = golden_feats(g)
    dfeats, dloc, dw = oracle.backward(g["grad_out"], feats, g["loc"], g["w"])
    for i, d in enumerate(dfeats):
        assert_close(d, g[f"dfeat{i}"], GRAD_RTOL, f"{name} dfeat{i}")
    assert_close(dloc[..., :2], g["dloc_xy"], GRAD_RTOL, f"{name} dloc xy")
    assert_close(dw, g["dw"], GRAD_RTOL, f"{name} dw")
    # the device kernels never write the view-coordinate gradient
    assert (dloc[..., 2] == 0).all()


@pytest.mark.parametrize("name", REAL)
def test_oracle_backward_matches_golden_real_size(oracle, name):
    g = load_golden(name)
    feats = golden_feats(g)
    dfeats, dloc, dw = oracle.backward(g["grad_out"], feats, g["loc"], g["w"])
    check_dfeat_digest(g, dfeats)
    assert_close(dloc[..., :2], g["dloc_xy"], GRAD_RTOL, f"{name} dloc xy")
    assert_close(dw, g["dw"], GRAD_RTOL, f"{name} dw")
    assert (dloc[..., 2] == 0).all()


def test_torch_ref_matches_golden():
    """The grid_sample restatement used as CPU baseline equals the reference fallback."""
    from oracle import torch_ref
    for name in DENSE:
        g = load_golden(name)
        feats = [torch.from_numpy(f) for f in golden_feats(g)]
        out = torch_ref.grid_sample_ref_channel_last(
            feats, torch.from_numpy(g["loc"]), torch.from_numpy(g["w"]))
        assert_close(out.numpy(), g["out"], 1e-6, f"{name} torch_ref forward")


def test_oracle_out_of_range_view_contributes_zero(oracle):
    feats, loc, w, grad_out = random_case(7, TINY_LEVELS, B=1, N=6, C=4, Q=4, P=2, oob_frac=0)
    loc[0, 0, 0, 2] = 1.3      # rounds to view 7 (> N-1)
    loc[0, 1, 1, 2] = -0.2     # rounds to view -1
    out = oracle.forward(feats, loc, w)
    assert (out[0, 0, :, 0] == 0).all() and (out[0, 1, :, 1] == 0).all()
    assert np.abs(out[0, 0, :, 1]).max() > 0
    dfeats, dloc, dw = oracle.backward(grad_out, feats, loc, w)
    assert (dloc[0, 0, 0] == 0).all() and (dw[0, 0, 0] == 0).all()


def test_oracle_empty_inputs(oracle):
    feats, loc, w, grad_out = random_case(8, TINY_LEVELS, B=1, N=6, C=4, Q=0, P=2)
    out = oracle.forward(feats, loc, w)
    assert out.shape == (1, 0, 4, 2)
    dfeats, dloc, dw = oracle.backward(grad_out, feats, loc, w)
    assert all((d == 0).all() for d in dfeats)


def test_oracle_gradients_match_finite_differences(oracle):
    """fp64-free sanity check of the analytic gradients: directional finite differences
    of the forward oracle (central, on loc and weights) agree with backward."""
    feats, loc, w, grad_out = random_case(11, TINY_LEVELS, B=1, N=6, C=4, Q=6, P=2, oob_frac=0)
    # keep clear of pixel-cell borders so the difference quotient stays on one bilinear patch
    loc[..., :2] = 0.05 + 0.9 * loc[..., :2]
    _, dloc, dw = oracle.backward(grad_out, feats, loc, w)
    rng = np.random.default_rng(0)
    eps = 1e-3
    dirw = rng.standard_normal(w.shape).astype(np.float32)
    fd = ((oracle.forward(feats, loc, w + eps * dirw).astype(np.float64)
           - oracle.forward(feats, loc, w - eps * dirw)) * grad_out).sum() / (2 * eps)
    an = (dw.astype(np.float64) * dirw).sum()
    assert abs(fd - an) <= 2e-3 * max(1.0, abs(an))
