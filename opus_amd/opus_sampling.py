"""Caller of the sampling op: mirror of the reference's ``models/opus_sampling.py``.

``sampling_4d`` keeps the reference's signature and semantics (models/opus_sampling.py:19-124):
project every 3-D sampling point into the T*N camera views, keep the first view in which
it is visible, pack points and scale weights with (frame, group) folded into the batch
dimension, run ``msmv_sampling`` and unpack to [B, Q, G, T*P, C].

Two inherited behaviours are reproduced on purpose, because trained weights depend on them:
  * points are packed in (B, T, G) order but scale weights in (B, G, T) order
    (opus_sampling.py:104-106 vs :110-112), so row i of the weights is not the (t, g) of
    row i of the points whenever T != G;
  * the visibility mask only selects the view; a point visible in no camera is still
    sampled in camera 0 at wherever it projects (opus_sampling.py:94-98) and relies on the
    op's zero padding.

Feature maps are expected in the layout the HIP op takes, [B*T*G, N, H, W, C] (channel-last),
i.e. what the reference's decoder prepares when MSMV_CUDA is true
(models/opus_transformer.py:119-121); ``relayout_features`` restates that step.
"""
import ctypes

import torch

from . import _lib
from .csrc.wrapper import msmv_sampling, msmv_sampling_rows


def relayout_features_torch(mlvl_feats, num_frames, num_views, num_groups):
    """[B, T*N, G*C, H, W] per level -> [B*T*G, N, H, W, C] contiguous (channel-last).

    Restates models/opus_transformer.py:113-121 as the reference writes it (reshape, permute,
    contiguous): channel index = g*C + c, image index = t*N + n.
    """
    out = []
    for feat in mlvl_feats:
        B, TN, GC, H, W = feat.shape
        T, N, G = num_frames, num_views, num_groups
        assert TN == T * N and GC % G == 0
        C = GC // G
        feat = feat.reshape(B, T, N, G, C, H, W).permute(0, 1, 3, 2, 5, 6, 4)
        out.append(feat.reshape(B * T * G, N, H, W, C).contiguous())
    return out


def relayout_features_channel_first(mlvl_feats, num_frames, num_views, num_groups):
    """[B, T*N, G*C, H, W] per level -> [B*T*G, C, N, H, W] contiguous: the layout the decoder
    prepares for the grid_sample fallback when MSMV_CUDA is false
    (models/opus_transformer.py:122-124)."""
    out = []
    for feat in mlvl_feats:
        B, TN, GC, H, W = feat.shape
        T, N, G = num_frames, num_views, num_groups
        assert TN == T * N and GC % G == 0
        C = GC // G
        feat = feat.reshape(B, T, N, G, C, H, W).permute(0, 1, 3, 4, 2, 5, 6)
        out.append(feat.reshape(B * T * G, C, N, H, W).contiguous())
    return out


class _Relayout(torch.autograd.Function):
    """opus_msmv_relayout behind autograd: a tiled transpose forward, its inverse backward."""

    @staticmethod
    def forward(ctx, feat, T, N, G):
        if not feat.is_cuda:
            raise RuntimeError("relayout_features runs on the GPU only; "
                               "relayout_features_torch is the tensor-op statement of it")
        B, TN, GC, H, W = feat.shape
        assert TN == T * N and GC % G == 0
        C = GC // G
        src = feat.contiguous().float()
        dst = torch.empty((B * T * G, N, H, W, C), dtype=torch.float32, device=feat.device)
        ctx.dims = (B, T, N, G, C, H, W)
        _Relayout._run(src, dst, ctx.dims, 0)
        return dst

    @staticmethod
    def backward(ctx, grad):
        B, T, N, G, C, H, W = ctx.dims
        src = grad.contiguous().float()
        dst = torch.empty((B, T * N, G * C, H, W), dtype=torch.float32, device=grad.device)
        _Relayout._run(src, dst, ctx.dims, 1)
        return dst, None, None, None

    @staticmethod
    def _run(src, dst, dims, inverse):
        lib = _lib.load()
        with torch.cuda.device(src.device):
            _lib.check(lib.opus_msmv_relayout(
                _vp(src), _vp(dst), *dims, inverse,
                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "opus_msmv_relayout")


def relayout_features(mlvl_feats, num_frames, num_views, num_groups):
    """Same result as ``relayout_features_torch`` from one tiled-transpose launch per level
    (and its inverse for the gradient)."""
    return [_Relayout.apply(f, num_frames, num_views, num_groups) for f in mlvl_feats]


def project_and_pack_torch(sample_points, scale_weights, occ2img, image_h, image_w, num_views=6, eps=1e-5):
    """Everything ``sampling_4d`` does before the op (opus_sampling.py:36-113), as the
    reference writes it: ~20 tensor ops over [B, T, N, Q, G*P, ...] temporaries.  Kept as the
    statement of the semantics that ``project_and_pack`` (one HIP kernel) is tested against.

    sample_points [B, Q, T, G, P, 3] (metres), scale_weights [B, Q, G, T, P, L],
    occ2img [B, T*N, 4, 4].  Returns (loc [B*T*G, Q, P, 3], weights [B*G*T, Q, P, L]).
    """
    B, Q, T, G, P, _ = sample_points.shape
    N = num_views
    GP = G * P
    pts = sample_points.reshape(B, Q, T, GP, 3)
    pts = torch.cat([pts, torch.ones_like(pts[..., :1])], dim=-1)            # homogeneous
    pts = pts.permute(0, 2, 1, 3, 4)                                           # [B, T, Q, GP, 4]
    mats = occ2img.reshape(B, T, N, 4, 4)
    # cam[b,t,n,q,k,i] = sum_j mats[b,t,n,i,j] * pts[b,t,q,k,j]   (opus_sampling.py:54)
    cam = torch.matmul(mats[:, :, :, None, None], pts[:, :, None, ..., None]).squeeze(-1)

    homo = cam[..., 2:3]
    homo_nonzero = torch.maximum(homo, torch.zeros_like(homo) + eps)           # :58
    uv = cam[..., 0:2] / homo_nonzero                                          # :59
    uv = torch.stack([uv[..., 0] / image_w, uv[..., 1] / image_h], dim=-1)     # :62-63

    valid = ((homo > eps) & (uv[..., 1:2] > 0.0) & (uv[..., 1:2] < 1.0)
             & (uv[..., 0:1] > 0.0) & (uv[..., 0:1] < 1.0)).squeeze(-1).float()   # [B, T, N, Q, GP]
    valid = valid.permute(0, 1, 3, 4, 2)                                       # [B, T, Q, GP, N]
    uv = uv.permute(0, 1, 3, 4, 2, 5)                                          # [B, T, Q, GP, N, 2]

    # first valid camera, camera 0 if none (argmax of a 0/1 mask, opus_sampling.py:94)
    i_view = torch.argmax(valid, dim=-1, keepdim=True)                         # [B, T, Q, GP, 1]
    uv = torch.gather(uv, 4, i_view[..., None].expand(B, T, Q, GP, 1, 2)).squeeze(4)
    loc = torch.cat([uv, i_view.float() / (N - 1)], dim=-1)                    # [B, T, Q, GP, 3]

    loc = loc.reshape(B, T, Q, G, P, 3).permute(0, 1, 3, 2, 4, 5)              # (B, T, G) order
    loc = loc.reshape(B * T * G, Q, P, 3).contiguous()
    w = scale_weights.reshape(B, Q, G, T, P, -1).permute(0, 2, 3, 1, 4, 5)     # (B, G, T) order
    w = w.reshape(B * G * T, Q, P, -1).contiguous()
    return loc, w


def _vp(t):
    return ctypes.c_void_p(t.data_ptr())


class _ProjectPack(torch.autograd.Function):
    """opus_msmv_project_pack / _backward behind autograd (one kernel each way)."""

    @staticmethod
    def forward(ctx, sample_points, scale_weights, occ2img, image_h, image_w, num_views, eps, num_frames=None):
        if not sample_points.is_cuda:
            raise RuntimeError("project_and_pack runs on the GPU only; "
                               "project_and_pack_torch is the tensor-op statement of it")
        L = scale_weights.shape[-1]
        N = int(num_views)
        shared = num_frames is not None       # one set of points / weights for all T frames
        if shared:
            B, Q, G, P, three = sample_points.shape
            T = int(num_frames)
            assert scale_weights.shape == (B, Q, G, P, L), "scale_weights must be [B,Q,G,P,L]"
        else:
            B, Q, T, G, P, three = sample_points.shape
            assert scale_weights.shape == (B, Q, G, T, P, L), "scale_weights must be [B,Q,G,T,P,L]"
        assert three == 3
        assert occ2img.shape == (B, T * N, 4, 4), "occ2img must be [B,T*N,4,4]"
        flags = _lib.PACK_SHARED_FRAMES if shared else 0
        pts = sample_points.contiguous().float()
        sw = scale_weights.contiguous().float()
        mats = occ2img.contiguous().float()
        loc = torch.empty((B * T * G, Q, P, 3), dtype=torch.float32, device=pts.device)
        w = torch.empty((B * G * T, Q, P, L), dtype=torch.float32, device=pts.device)
        lib = _lib.load()
        with torch.cuda.device(pts.device):
            _lib.check(lib.opus_msmv_project_pack(
                _vp(pts), _vp(sw), _vp(mats), B, Q, T, G, P, L, N,
                float(image_h), float(image_w), float(eps), flags, _vp(loc), _vp(w),
                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "opus_msmv_project_pack")
        ctx.save_for_backward(pts, mats, loc)
        ctx.dims = (B, Q, T, G, P, L, N, float(image_h), float(image_w), float(eps), flags)
        ctx.sw_shape = tuple(sw.shape)
        return loc, w

    @staticmethod
    def backward(ctx, dloc, dw):
        pts, mats, loc = ctx.saved_tensors
        B, Q, T, G, P, L, N, image_h, image_w, eps, flags = ctx.dims
        dloc = dloc.contiguous().float()
        dw = dw.contiguous().float()
        dpts = torch.empty_like(pts)
        dsw = torch.empty(ctx.sw_shape, dtype=torch.float32, device=pts.device)
        lib = _lib.load()
        with torch.cuda.device(pts.device):
            _lib.check(lib.opus_msmv_project_pack_backward(
                _vp(dloc), _vp(dw), _vp(pts), _vp(mats), _vp(loc), B, Q, T, G, P, L, N,
                image_h, image_w, eps, flags, _vp(dpts), _vp(dsw),
                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                "opus_msmv_project_pack_backward")
        return dpts, dsw, None, None, None, None, None, None


def project_pack_into(sample_points, scale_weights, occ2img, image_h, image_w, num_views, loc, weights,
                      eps=1e-5):
    """Raw opus_msmv_project_pack into caller-owned ``loc`` / ``weights`` (no autograd, no
    allocation): for benchmarks and callers that manage their own buffers."""
    B, Q, T, G, P, _ = sample_points.shape
    L = scale_weights.shape[-1]
    assert loc.shape == (B * T * G, Q, P, 3) and weights.shape == (B * G * T, Q, P, L)
    for t in (sample_points, scale_weights, occ2img, loc, weights):
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
    with torch.cuda.device(loc.device):
        _lib.check(_lib.load().opus_msmv_project_pack(
            _vp(sample_points), _vp(scale_weights), _vp(occ2img), B, Q, T, G, P, L, int(num_views),
            float(image_h), float(image_w), float(eps), 0, _vp(loc), _vp(weights),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "opus_msmv_project_pack")


def project_pack_backward_into(dloc, dweights, sample_points, occ2img, loc, image_h, image_w, num_views,
                               dsample_points, dscale_weights, eps=1e-5):
    """Raw opus_msmv_project_pack_backward into caller-owned gradient buffers."""
    B, Q, T, G, P, _ = sample_points.shape
    L = dweights.shape[-1]
    assert dsample_points.shape == sample_points.shape and dscale_weights.shape == (B, Q, G, T, P, L)
    for t in (dloc, dweights, sample_points, occ2img, loc, dsample_points, dscale_weights):
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
    with torch.cuda.device(loc.device):
        _lib.check(_lib.load().opus_msmv_project_pack_backward(
            _vp(dloc), _vp(dweights), _vp(sample_points), _vp(occ2img), _vp(loc), B, Q, T, G, P, L,
            int(num_views), float(image_h), float(image_w), float(eps), 0, _vp(dsample_points),
            _vp(dscale_weights),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "opus_msmv_project_pack_backward")


def project_and_pack(sample_points, scale_weights, occ2img, image_h, image_w, num_views=6, eps=1e-5,
                     num_frames=None):
    """Same result as ``project_and_pack_torch`` from one HIP kernel (and one for the
    backward): (loc [B*T*G, Q, P, 3], weights [B*G*T, Q, P, L]).  occ2img gets no gradient
    (the reference builds it from image metadata, models/opus_transformer.py:105-110).

    ``num_frames=T``: the frames share their points and weights - sample_points [B,Q,G,P,3] and
    scale_weights [B,Q,G,P,L], i.e. what the decoder has BEFORE it expands them over T
    (models/opus_transformer.py:350-358); their gradients then come back summed over the frames."""
    return _ProjectPack.apply(sample_points, scale_weights, occ2img, image_h, image_w, num_views, eps,
                              num_frames)


def unpack(final, B, T, G):
    """[B*T*G, Q, C, P] -> [B, Q, G, T*P, C] (opus_sampling.py:119-122)."""
    _, Q, C, P = final.shape
    final = final.reshape(B, T, G, Q, C, P).permute(0, 3, 2, 1, 5, 4)
    return final.flatten(3, 4)


def sampling_4d(sample_points, mlvl_feats, scale_weights, occ2img, image_h, image_w, num_views=6, eps=1e-5,
                fused=True, num_frames=None):
    """Same contract as the reference's sampling_4d with channel-last features
    [B*T*G, N, H, W, C]; returns [B, Q, G, T*P, C].  ``fused=False`` runs the projection and
    the un-pack as the reference's tensor ops around the plain op (same result, for
    comparison)."""
    if num_frames is not None:
        # un-expanded inputs ([B,Q,G,P,3] / [B,Q,G,P,L], see project_and_pack): fused path only
        assert fused, "num_frames needs the fused path"
        B, Q, G, P, _ = sample_points.shape
        loc, w = project_and_pack(sample_points, scale_weights, occ2img, image_h, image_w, num_views, eps,
                                  num_frames)
        return msmv_sampling_rows(mlvl_feats, loc, w, int(num_frames), G)
    B, Q, T, G, P, _ = sample_points.shape
    if fused:
        # two launches forward (project + pack, then the op writing [B,Q,G,T*P,C] directly);
        # no [B,T,N,Q,G*P,...] temporaries and no permute copy in either direction
        loc, w = project_and_pack(sample_points, scale_weights, occ2img, image_h, image_w, num_views, eps)
        return msmv_sampling_rows(mlvl_feats, loc, w, T, G)
    loc, w = project_and_pack_torch(sample_points, scale_weights, occ2img, image_h, image_w, num_views, eps)
    final = msmv_sampling(mlvl_feats, loc, w)
    return unpack(final, B, T, G)


def sampling_inputs(query_points, query_feat, offset_weight, offset_bias, scale_weight, scale_bias,
                    num_frames, num_groups, num_points, num_levels, pc_range):
    """What the decoder's sampling module feeds ``sampling_4d`` (restates
    models/opus_transformer.py:337-358 and models/bbox/utils.py:88-93; not pinned by a fixture:
    the module itself needs mmcv to import).

    query_points [B, Q, R, 3] in [0, 1] (R points per query), query_feat [B, Q, E];
    offset_* / scale_* are the weights of the two Linear layers (E -> G*P*3 and E -> G*P*L).
    Returns sample_points [B, Q, T, G, P, 3] (metres, the same points for every frame: an
    expanded view) and scale_weights [B, Q, G, T, P, L] (softmax over levels, expanded over T).
    """
    B, Q = query_points.shape[:2]
    T, G, P, L = num_frames, num_groups, num_points, num_levels
    lo = query_points.new_tensor(pc_range[:3])
    hi = query_points.new_tensor(pc_range[3:])
    pts = query_points * (hi - lo) + lo                                        # decode_points
    if pts.shape[2] == 1:
        center, scale = pts, torch.zeros_like(pts)
    else:
        center = pts.mean(dim=2, keepdim=True)
        scale = pts.std(dim=2, keepdim=True)
    offset = torch.nn.functional.linear(query_feat, offset_weight, offset_bias).view(B, Q, -1, 3)
    sp = (center + offset * scale).view(B, Q, 1, G, P, 3).expand(B, Q, T, G, P, 3)
    sw = torch.nn.functional.linear(query_feat, scale_weight, scale_bias).view(B, Q, G, 1, P, L)
    sw = torch.softmax(sw, dim=-1).expand(B, Q, G, T, P, L)
    return sp, sw
