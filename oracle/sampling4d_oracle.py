"""TEST INFRASTRUCTURE - CPU oracle for the caller side of the sampling op.

A numpy restatement of what the reference's ``sampling_4d`` does around the op
(models/opus_sampling.py:19-124): projection into the cameras, first-visible-camera
selection, packing, un-packing, and the analytic gradient of that chain.  Written as plain
loops over cameras / vectorised numpy over points, independently of the tensor-op
restatement in ``opus_amd/opus_sampling.py`` and of the HIP kernel.

Pinned by ``tests/golden/g5_sampling4d.npz`` (forward: the loc / weight tensors the
reference hands to the op) and ``tests/golden/g6_sampling4d_grad.npz`` (backward: the
reference's autograd gradients of the whole chain), both generated from the reference's own
code by ``tests/golden/make_golden.py``.

Only tests, ``__graft_entry__.smoke()`` and bench.py's cpu_baseline leg may import this.
"""
import numpy as np


def _project(pts, M):
    """rows 0..2 of M @ (x, y, z, 1) for pts [..., 3] in float32 (opus_sampling.py:54)."""
    x, y, z = pts[..., 0], pts[..., 1], pts[..., 2]
    f = np.float32
    cx = f(M[0, 0]) * x + f(M[0, 1]) * y + f(M[0, 2]) * z + f(M[0, 3])
    cy = f(M[1, 0]) * x + f(M[1, 1]) * y + f(M[1, 2]) * z + f(M[1, 3])
    cz = f(M[2, 0]) * x + f(M[2, 1]) * y + f(M[2, 2]) * z + f(M[2, 3])
    return cx, cy, cz


def project_pack(sample_points, scale_weights, occ2img, image_h, image_w, num_views=6, eps=1e-5):
    """sample_points [B,Q,T,G,P,3], scale_weights [B,Q,G,T,P,L], occ2img [B,T*N,4,4]
    -> loc [B*T*G,Q,P,3], weights [B*G*T,Q,P,L]   (opus_sampling.py:36-113)."""
    pts = np.asarray(sample_points, np.float32)
    B, Q, T, G, P, _ = pts.shape
    N = int(num_views)
    mats = np.asarray(occ2img, np.float32).reshape(B, T, N, 4, 4)
    eps32 = np.float32(eps)
    loc = np.zeros((B, T, G, Q, P, 3), np.float32)
    for b in range(B):
        for t in range(T):
            p_bt = pts[b, :, t].transpose(1, 0, 2, 3)          # [G, Q, P, 3]
            chosen = np.full((G, Q, P), -1, np.int64)
            for n in range(N):
                cx, cy, cz = _project(p_bt, mats[b, t, n])
                hz = np.maximum(cz, eps32)                                     # :58
                u = (cx / hz) / np.float32(image_w)                            # :59, :62
                v = (cy / hz) / np.float32(image_h)                            # :63
                valid = (cz > eps32) & (v > 0) & (v < 1) & (u > 0) & (u < 1)   # :66-70
                take = valid & (chosen < 0)
                if n == 0:                      # camera 0 is the default when none is valid (:94)
                    loc[b, t, ..., 0], loc[b, t, ..., 1] = u, v
                loc[b, t, ..., 0] = np.where(take, u, loc[b, t, ..., 0])
                loc[b, t, ..., 1] = np.where(take, v, loc[b, t, ..., 1])
                chosen = np.where(take, n, chosen)
            chosen = np.maximum(chosen, 0)
            loc[b, t, ..., 2] = chosen.astype(np.float32) / np.float32(N - 1)   # :101
    loc = loc.reshape(B * T * G, Q, P, 3)                                      # (B,T,G) order, :104-106
    sw = np.asarray(scale_weights, np.float32)
    w = np.ascontiguousarray(sw.transpose(0, 2, 3, 1, 4, 5)).reshape(B * G * T, Q, P, -1)   # (B,G,T), :110-112
    return loc, w


def unpack(final, B, T, G):
    """[B*T*G,Q,C,P] -> [B,Q,G,T*P,C]   (opus_sampling.py:119-122)."""
    _, Q, C, P = final.shape
    x = final.reshape(B, T, G, Q, C, P).transpose(0, 3, 2, 1, 5, 4)
    return np.ascontiguousarray(x).reshape(B, Q, G, T * P, C)


def unpack_backward(grad_final, B, T, G):
    """Adjoint of ``unpack``: [B,Q,G,T*P,C] -> [B*T*G,Q,C,P]."""
    _, Q, _, TP, C = grad_final.shape
    P = TP // T
    x = grad_final.reshape(B, Q, G, T, P, C).transpose(0, 3, 2, 1, 5, 4)
    return np.ascontiguousarray(x).reshape(B * T * G, Q, C, P)


def project_pack_backward(dloc, dweights, sample_points, occ2img, loc, image_h, image_w, num_views=6, eps=1e-5):
    """Gradient of ``project_pack`` in float64: (d sample_points [B,Q,T,G,P,3],
    d scale_weights [B,Q,G,T,P,L]).  Only the chosen camera's projection carries gradient;
    max(z, eps) passes it where z > eps; the view slot of d loc is ignored."""
    pts = np.asarray(sample_points, np.float64)
    B, Q, T, G, P, _ = pts.shape
    N = int(num_views)
    mats = np.asarray(occ2img, np.float64).reshape(B, T, N, 4, 4)
    loc = np.asarray(loc).reshape(B, T, G, Q, P, 3)
    dloc = np.asarray(dloc, np.float64).reshape(B, T, G, Q, P, 3)
    view = np.rint(loc[..., 2].astype(np.float64) * (N - 1)).astype(np.int64)
    dpts = np.zeros_like(pts)
    for b in range(B):
        for t in range(T):
            M = mats[b, t][view[b, t]]                         # [G, Q, P, 4, 4]
            p = pts[b, :, t].transpose(1, 0, 2, 3)             # [G, Q, P, 3]
            ph = np.concatenate([p, np.ones_like(p[..., :1])], -1)
            cam = np.einsum("...ij,...j->...i", M[..., :3, :], ph)
            cx, cy, cz = cam[..., 0], cam[..., 1], cam[..., 2]
            hz = np.maximum(cz, eps)
            gcx = dloc[b, t, ..., 0] / (hz * image_w)
            gcy = dloc[b, t, ..., 1] / (hz * image_h)
            gcz = np.where(cz > eps, -(gcx * cx + gcy * cy) / hz, 0.0)
            g = (gcx[..., None] * M[..., 0, :3] + gcy[..., None] * M[..., 1, :3]
                 + gcz[..., None] * M[..., 2, :3])             # [G, Q, P, 3]
            dpts[b, :, t] = g.transpose(1, 0, 2, 3)
    dw = np.asarray(dweights)
    L = dw.shape[-1]
    dsw = np.ascontiguousarray(dw.reshape(B, G, T, Q, P, L).transpose(0, 3, 1, 2, 4, 5))
    return dpts.astype(np.float32), dsw
