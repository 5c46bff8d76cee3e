"""Shared helpers for the parity tests."""
import os

import numpy as np

from procedural import procedural_feats, synthetic_occ2img  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

REAL_LEVELS = [(64, 176), (32, 88), (16, 44), (8, 22)]
TINY_LEVELS = [(8, 22), (4, 11), (2, 6), (1, 3)]

# parity bars: forward 1e-5 relative, gradients 1e-4 relative (atomic / summation order).
# "relative" is taken against the tensor's max magnitude because outputs cross zero.
FWD_RTOL = 1e-5
GRAD_RTOL = 1e-4


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_feats(g):
    """Channel-last feature list of a fixture (stored, or regenerated procedurally)."""
    L = int(g["n_levels"])
    if "feat0" in g:
        return [g[f"feat{i}"] for i in range(L)]
    B, N, C = int(g["B"]), int(g["N"]), int(g["C"])
    salt0 = int(g["feat_salt0"])
    return [procedural_feats((B, N, int(h), int(w), C), salt=salt0 + i)
            for i, (h, w) in enumerate(g["levels"])]


def assert_close(actual, expected, rtol, what, atol_floor=0.0):
    actual = np.asarray(actual, dtype=np.float64)
    expected = np.asarray(expected, dtype=np.float64)
    assert actual.shape == expected.shape, (what, actual.shape, expected.shape)
    assert np.isfinite(actual).all(), f"{what}: non-finite values"
    scale = max(np.abs(expected).max(), atol_floor, 1e-30)
    err = np.abs(actual - expected).max() if actual.size else 0.0
    assert err <= rtol * scale, f"{what}: max|diff|={err:.3e} > {rtol:g} * {scale:.3e}"


def check_dfeat_digest(g, dfeats, rtol=GRAD_RTOL):
    """Compare d feats with the digests a real-size fixture stores."""
    for i, d in enumerate(dfeats):
        C = d.shape[-1]
        flat = np.asarray(d, dtype=np.float64).reshape(-1, C)
        assert_close(flat.sum(0), g[f"dfeat{i}_chan_sum"], rtol, f"dfeat{i} channel sums",
                     atol_floor=float(g[f"dfeat{i}_abs_sum"]) / C * 1e-2)
        proj = procedural_feats(d.shape, salt=7000 + i).reshape(-1, C).astype(np.float64)
        got = (flat * proj).sum()
        assert abs(got - float(g[f"dfeat{i}_proj"])) <= rtol * float(g[f"dfeat{i}_abs_sum"]), \
            f"dfeat{i} projection"
        idx = g[f"dfeat{i}_pix_idx"]
        assert_close(flat[idx], g[f"dfeat{i}_pix_val"], rtol, f"dfeat{i} sampled pixels")
        # (the stored touched-pixel count is informational: the fallback's trilinear view
        # axis leaves ~1e-8 residues in neighbouring views that the device kernels do not)


def random_case(seed, levels, B, N, C, Q, P, oob_frac=0.25, dtype=np.float32):
    """Seeded random inputs of the op in the device layout (channel-last)."""
    rng = np.random.default_rng(seed)
    L = len(levels)
    feats = [rng.standard_normal((B, N, h, w, C)).astype(dtype) for h, w in levels]
    xy = rng.uniform(0, 1, (B, Q, P, 2))
    if oob_frac > 0:
        oob = rng.uniform(0, 1, (B, Q, P, 2)) < oob_frac / 2
        xy = np.where(oob, rng.uniform(-0.3, 1.3, xy.shape), xy)
    view = rng.integers(0, N, (B, Q, P, 1)).astype(np.float64) / max(N - 1, 1)
    loc = np.concatenate([xy, view], -1).astype(dtype)
    logits = rng.standard_normal((B, Q, P, L))
    w = (np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)).astype(dtype)
    grad_out = rng.standard_normal((B, Q, C, P)).astype(dtype)
    return feats, loc, w, grad_out


def sampling4d_case(seed, B, Q, T, G, P, N=6, L=4, image_h=256, image_w=704):
    """Random caller-side inputs in the reference's shapes: sample_points [B,Q,T,G,P,3] in the
    occupancy range, softmax scale weights [B,Q,G,T,P,L], ring-of-cameras occ2img [B,T*N,4,4]."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform([-40, -40, -1.0], [40, 40, 5.4], (B, Q, T, G, P, 3)).astype(np.float32)
    if Q > 1:
        pts[0, 0] = 0.0                      # at the ego origin: visible in no camera
    logits = rng.standard_normal((B, Q, G, T, P, L))
    sw = (np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)).astype(np.float32)
    return pts, sw, synthetic_occ2img(rng, B, T, N, image_h, image_w)
