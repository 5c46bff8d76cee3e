#!/usr/bin/env python3
"""Developer timing of the individual entry points (not the contract benchmark: bench.py).

    python tools/devbench.py [--variant L|M|S|T] [--batch 1] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opus_amd.csrc import wrapper  # noqa: E402

VARIANTS = {"T": (600, 4), "S": (1200, 2), "M": (2400, 2), "L": (4800, 2)}
LEVELS = [(64, 176), (32, 88), (16, 44), (8, 22)]


def make_inputs(B, Q, P, seed=1234, N=6, C=64, levels=LEVELS):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    feats = [torch.randn((B, N, h, w, C), generator=gen, device="cuda") for h, w in levels]
    xy = torch.rand((B, Q, P, 2), generator=gen, device="cuda")
    view = torch.randint(0, N, (B, Q, P, 1), generator=gen, device="cuda").float() / (N - 1)
    loc = torch.cat([xy, view], -1).contiguous()
    w = torch.softmax(torch.randn((B, Q, P, len(levels)), generator=gen, device="cuda"), -1)
    g = torch.randn((B, Q, C, P), generator=gen, device="cuda")
    return feats, loc, w, g


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="L")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    Q, P = VARIANTS[args.variant]
    B = 32 * args.batch
    feats, loc, w, g = make_inputs(B, Q, P)
    npts = B * Q * P
    fbytes = sum(f.numel() * 4 for f in feats)
    fwd_bytes = npts * (4 * 4 * 64 * 4 + 256 + 12 + 16)
    bwd_bytes = npts * 8504 + fbytes
    print(f"variant {args.variant} B'={B} Q={Q} P={P} points={npts} feats={fbytes / 1e6:.1f} MB")

    med, mn = timeit(lambda: wrapper.msmv_forward(feats, loc, w), args.iters)
    print(f"forward fast   : {med * 1e3:8.1f} us (min {mn * 1e3:.1f})  {npts / med / 1e6:8.2f} Gpts/s  {fwd_bytes / med / 1e9:6.2f} TB/s alg")
    med, mn = timeit(lambda: wrapper.msmv_forward(feats, loc, w, flags=1), max(3, args.iters // 4))
    print(f"forward generic: {med * 1e3:8.1f} us (min {mn * 1e3:.1f})")
    for algo in ("owner", "atomic"):
        med, mn = timeit(lambda: wrapper.msmv_backward(g, feats, loc, w, algo=algo), max(3, args.iters // 2))
        print(f"backward {algo:6s}: {med * 1e3:8.1f} us (min {mn * 1e3:.1f})  {npts / med / 1e6:8.2f} Gpts/s  {bwd_bytes / med / 1e9:6.2f} TB/s alg")
    x = torch.empty(fbytes // 4, device="cuda")
    med, mn = timeit(lambda: x.zero_(), 10)
    print(f"zero-fill of d feats ({fbytes / 1e6:.0f} MB): {med * 1e3:.1f} us -> {fbytes / med / 1e9:.2f} TB/s")
    y = torch.empty_like(x)
    med, mn = timeit(lambda: y.copy_(x), 10)
    print(f"device copy ({fbytes / 1e6:.0f} MB): {med * 1e3:.1f} us -> {2 * fbytes / med / 1e9:.2f} TB/s (read+write)")


if __name__ == "__main__":
    main()
