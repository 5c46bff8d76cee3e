#!/usr/bin/env python3
"""Caller-side fusion benchmark: the projection / camera selection / packing that sampling_4d
does around the op (models/opus_sampling.py:36-113 of the reference), as ONE HIP kernel per
direction (opus_msmv_project_pack[_backward]) against the same arithmetic as ~20 tensor ops on
the same GPU, and the whole sampling_4d forward + backward either way.

    python tools/bench_sampling4d.py [--variant L] [--iters 50]

Prints one JSON line.  Algorithmic bytes of the fused kernels: 12 B of point + 4L B of weights
read and the same written per sampled point (the T*N 4x4 matrices are 3 KB and stay in cache).
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from opus_amd import _lib, opus_sampling  # noqa: E402
from opus_amd.csrc import wrapper  # noqa: E402
from procedural import synthetic_occ2img  # noqa: E402

LEVELS = [(64, 176), (32, 88), (16, 44), (8, 22)]
VARIANTS = {"T": (600, 4), "S": (1200, 2), "M": (2400, 2), "L": (4800, 2)}
T, G, N, C, H, W = 8, 4, 6, 64, 256, 704


def timed(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="L", choices=list(VARIANTS))
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    Q, P = VARIANTS[args.variant]
    B, L = args.batch, len(LEVELS)
    dev = torch.device("cuda", 0)
    import numpy as np
    rng = np.random.default_rng(0)
    gen = torch.Generator(device=dev).manual_seed(0)
    lo = torch.tensor([-40.0, -40.0, -1.0], device=dev)
    hi = torch.tensor([40.0, 40.0, 5.4], device=dev)
    pts = (torch.rand((B, Q, T, G, P, 3), generator=gen, device=dev) * (hi - lo) + lo).requires_grad_(True)
    sw = torch.softmax(torch.randn((B, Q, G, T, P, L), generator=gen, device=dev), -1).requires_grad_(True)
    occ2img = torch.from_numpy(synthetic_occ2img(rng, B, T, N, H, W)).to(dev)
    feats = [torch.randn((B * T * G, N, h, w, C), generator=gen, device=dev).requires_grad_(True) for h, w in LEVELS]
    npts = B * T * G * Q * P
    dloc = torch.randn((B * T * G, Q, P, 3), generator=gen, device=dev)
    dw = torch.randn((B * T * G, Q, P, L), generator=gen, device=dev)
    gout = torch.randn((B, Q, G, T * P, C), generator=gen, device=dev)

    res = {"variant": f"OPUS-{args.variant}", "points": npts, "iters": args.iters}
    for name, fn in (("fused", opus_sampling.project_and_pack), ("tensor_ops", opus_sampling.project_and_pack_torch)):
        with torch.no_grad():
            res[f"pack_fwd_us_{name}"] = timed(lambda: fn(pts, sw, occ2img, H, W, N), args.iters)

        def fwd_bwd():
            pts.grad = sw.grad = None
            loc, w = fn(pts, sw, occ2img, H, W, N)
            torch.autograd.backward([loc, w], [dloc, dw])
        res[f"pack_fwdbwd_us_{name}"] = timed(fwd_bwd, args.iters)

        def chain(fused=(name == "fused")):
            pts.grad = sw.grad = None
            for f in feats:
                f.grad = None
            out = opus_sampling.sampling_4d(pts, feats, sw, occ2img, H, W, N, fused=fused)
            out.backward(gout)
        res[f"sampling_4d_fwdbwd_us_{name}"] = timed(chain, max(args.iters // 2, 5))
    # what the decoder really hands over: ONE set of points / weights, expanded over the frames
    # (models/opus_transformer.py:350-358) - as expanded views, and un-expanded with num_frames
    bp = pts.detach()[:, :, 0].contiguous().requires_grad_(True)             # [B,Q,G,P,3]
    bs = sw.detach()[:, :, :, 0].contiguous().requires_grad_(True)           # [B,Q,G,P,L]

    def chain_expanded():
        bp.grad = bs.grad = None
        for f in feats:
            f.grad = None
        out = opus_sampling.sampling_4d(bp[:, :, None].expand(B, Q, T, G, P, 3), feats,
                                        bs[:, :, :, None].expand(B, Q, G, T, P, L), occ2img, H, W, N)
        out.backward(gout)

    def chain_shared():
        bp.grad = bs.grad = None
        for f in feats:
            f.grad = None
        opus_sampling.sampling_4d(bp, feats, bs, occ2img, H, W, N, num_frames=T).backward(gout)
    res["sampling_4d_fwdbwd_us_fused_expanded_views"] = timed(chain_expanded, max(args.iters // 2, 5))
    res["sampling_4d_fwdbwd_us_fused_shared_frames"] = timed(chain_shared, max(args.iters // 2, 5))

    # a stack of six layers sampling the same feature maps, forward + backward through autograd:
    # six separate 735 MB d feats summed by autograd, one buffer the layers accumulate into, and
    # d feats deferred to one pass over all six layers' points
    def stack(kind):
        bp.grad = bs.grad = None
        for f in feats:
            f.grad = None
        src = feats if kind == "autograd" else wrapper.FeatureGradSink(feats, defer=kind == "deferred")
        outs = [opus_sampling.sampling_4d(bp, src, bs, occ2img, H, W, N, num_frames=T) for _ in range(6)]
        torch.autograd.backward(outs, [gout] * 6)
    for kind in ("autograd", "accumulate", "deferred"):
        res[f"six_layers_fwdbwd_us_{kind}"] = timed(lambda: stack(kind), max(args.iters // 5, 5), warmup=2)

    # the op alone, in its own output layout + the caller's permute copy, against row mode
    with torch.no_grad():
        loc, w = opus_sampling.project_and_pack(pts, sw, occ2img, H, W, N)
        fd = [f.detach() for f in feats]
        plain = wrapper.msmv_forward(fd, loc, w)
        res["op_fwd_us_plain"] = timed(lambda: wrapper.msmv_forward(fd, loc, w), args.iters)
        res["unpack_copy_us"] = timed(lambda: opus_sampling.unpack(plain, B, T, G).contiguous(), args.iters)
        res["op_fwd_us_rows"] = timed(lambda: wrapper.msmv_forward(fd, loc, w, rows=(T, G)), args.iters)
        g_plain = torch.randn_like(plain)
        res["op_bwd_us_plain"] = timed(lambda: wrapper.msmv_backward(g_plain, fd, loc, w), args.iters)
        res["op_bwd_us_rows"] = timed(lambda: wrapper.msmv_backward(gout, fd, loc, w, rows=(T, G)), args.iters)
        # per-stage times of the owner-computes backward, both layouts
        dfeats = [torch.empty_like(f) for f in fd]
        dloc, dw = torch.empty_like(loc), torch.empty_like(w)
        ws = torch.empty(max(wrapper.backward_workspace_bytes(fd, loc, w, _lib.BWD_OWNER), 16),
                         dtype=torch.uint8, device=dev)
        for tag, g, rows in (("plain", g_plain, None), ("rows", gout, (T, G))):
            for stage, bit in (("bin", _lib.BWD_STAGE_BIN), ("dfeat", _lib.BWD_STAGE_DFEAT),
                               ("gather", _lib.BWD_STAGE_GATHER)):
                res[f"bwd_{stage}_us_{tag}"] = timed(lambda: wrapper.msmv_backward_into(
                    g, fd, loc, w, dfeats, dloc, dw, ws, flags=_lib.BWD_OWNER | bit, rows=rows), args.iters)
        views = torch.round(loc[..., 2] * (N - 1)).long()
        res["view_histogram"] = torch.bincount(views.flatten(), minlength=N).tolist()
        res["frac_points_outside_image"] = float(((loc[..., :2] < 0) | (loc[..., :2] > 1)).any(-1).float().mean())
    # producer-side re-layout of all four levels (735 MB in, 735 MB out per sample)
    with torch.no_grad():
        raw = [torch.randn((B, T * N, G * C, h, w), generator=gen, device=dev) for h, w in LEVELS]
        res["relayout_us_fused"] = timed(lambda: opus_sampling.relayout_features(raw, T, N, G), 20)
        res["relayout_us_tensor_ops"] = timed(lambda: opus_sampling.relayout_features_torch(raw, T, N, G), 20)
        rl_bytes = 2 * sum(r.numel() * 4 for r in raw)
        res["relayout_bytes"] = rl_bytes
        res["relayout_GBps_fused"] = rl_bytes / res["relayout_us_fused"] / 1e3
        res["relayout_GBps_tensor_ops"] = rl_bytes / res["relayout_us_tensor_ops"] / 1e3
        del raw
    alg = npts * (12 + 4 * L) * 2
    res["pack_alg_bytes"] = alg
    res["pack_fwd_GBps_fused"] = alg / res["pack_fwd_us_fused"] / 1e3
    res["speedup_pack_fwdbwd"] = res["pack_fwdbwd_us_tensor_ops"] / res["pack_fwdbwd_us_fused"]
    res["speedup_sampling_4d_fwdbwd"] = res["sampling_4d_fwdbwd_us_tensor_ops"] / res["sampling_4d_fwdbwd_us_fused"]
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
