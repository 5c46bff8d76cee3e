#!/usr/bin/env python3
"""Benchmark of the msmv_sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--batch B]

One "step" is one pass of the op over one batch of synthetic input that is already
resident in HBM: by default one decoder layer's forward + backward at the OPUS-L sampling
shape (B' = 32*batch slabs, Q = 4800, P = 2, 4 levels of 256x704 features, C = 64).  For
N > 1 the driver launches one rank per GPU (torch.distributed.run); every rank works on
its own batch, nothing is exchanged on the data path (weak scaling), the timed region is
bracketed by a barrier + synchronize and the slowest rank's time is the job's time.

Started without a launcher (`python bench.py --gpus N`, no WORLD_SIZE in the environment) it
starts the N ranks itself, as fresh child processes, before anything in this process has
touched a GPU, and still prints one line.

Prints ONE JSON line on rank 0: whole-job sampled points/s, the roofline record of the
dominant kernel (HIP-event durations measured inside the timed region; `achieved` counts the
kernel's COMPULSORY bytes - distinct feature pixels + its dense reads and writes, computed from
the inputs - so `frac` <= 1; the no-reuse algorithmic figure is kept beside it as
`achieved_alg`), and - at N = 1 - the CPU baseline (the grid_sample formulation of
oracle/torch_ref.py on the host cores, on a bounded sample).
"""
import argparse
import gc
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from opus_amd import dist as odist  # noqa: E402  (host-side helpers only: loads no native code)

# bound in load_product(): the HIP library and its host mirror are loaded by the ranks that
# compute, never by the process that only starts them (or by a --stub-compute rehearsal)
_lib = opus_sampling = wrapper = None


def load_product():
    global _lib, opus_sampling, wrapper
    from opus_amd import _lib as lib_mod
    from opus_amd import opus_sampling as ops_mod
    from opus_amd.csrc import wrapper as wr_mod
    _lib, opus_sampling, wrapper = lib_mod, ops_mod, wr_mod

LEVELS = [(64, 176), (32, 88), (16, 44), (8, 22)]     # 256x704 input, strides 4..32
N_VIEWS, CHANNELS = 6, 64
SLABS_PER_SAMPLE = 32                                  # T=8 frames x G=4 channel groups
VARIANTS = {"T": (600, 4), "S": (1200, 2), "M": (2400, 2), "L": (4800, 2)}
HBM_PEAK_GBPS = 8000.0                                 # MI355X HBM3E, spec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="L-fwdbwd",
                    help="{T,S,M,L}-fwd | {T,S,M,L}-fwdbwd | decoder-{T,S,M,L} (6 layers fwd+bwd) | "
                         "train-M (6 x (2 fwd + 1 bwd) + RCCL all-reduce of a 0.3 GB gradient) | "
                         "sampling4d-{T,S,M,L} (projection + op in the caller's layout, fwd+bwd) | "
                         "chain-{T,S,M,L} (the decoder's whole sampling chain in training: feature re-layout, 6 x "
                         "(projection + bin-ordered forward, both again for the checkpoint's recomputation, "
                         "d loc / d w gather, projection backward), one d-feat pass over the 6 layers, inverse "
                         "re-layout) | infer-{T,S,M,L} (re-layout + 6 x (projection + forward))")
    ap.add_argument("--points", default="uniform", choices=["uniform", "cameras", "queries"],
                    help="uniform: u,v ~ U(0,1), view ~ U{0..5} (every corner in range: the byte count is "
                         "exact); cameras: 3-D points in the occupancy range projected into a ring of 6 "
                         "synthetic cameras x 8 ego poses (some points fall outside every image); queries: as "
                         "cameras, but the 3-D points come from the decoder's sampling module at its initial "
                         "weights (each query's G*P points scattered around the query's own points, the same "
                         "points in all 8 frames)")
    ap.add_argument("--batch", type=int, default=1, help="samples per GPU (B' = 32 * batch)")
    ap.add_argument("--bwd-algo", default="owner", choices=["owner", "atomic"])
    ap.add_argument("--dfeat", default="deferred", choices=["deferred", "accumulate"],
                    help="decoder workloads: how the layers' d feats are summed - 'deferred': every layer sorts "
                         "and gathers, one d-feat pass over all layers at the end writes d feats once; "
                         "'accumulate': the first layer's backward writes them, the others add to them")
    ap.add_argument("--fwd-order", default=None, choices=["bin", "input"],
                    help="caller-layout workloads (sampling4d-*): 'bin' sorts the points once per layer and runs "
                         "the forward in bin order, the backward re-using the sort; 'input': forward in input "
                         "order, the backward sorts for itself (default: bin, input for infer-*)")
    ap.add_argument("--no-checkpoint", action="store_true",
                    help="chain-* workloads: one forward per layer instead of two - the decoder without the "
                         "activation checkpointing of its sampling module (models/opus_transformer.py:372-378), "
                         "which trades a second forward for 79 MB of sampled features per layer and sample at "
                         "OPUS-L; with 288 GB of HBM per GPU the memory is there")
    ap.add_argument("--split-sample", action="store_true",
                    help="single-sample latency mode: the 32*batch slabs are split across the ranks (strong "
                         "scaling, still no data-path collective) instead of one whole batch per rank")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step into a HIP graph and time its replays (no launch gaps); the "
                         "per-kernel durations then come from evented steps run after the timed region")
    ap.add_argument("--event-steps", type=int, default=24,
                    help="steps run AFTER the timed region with a HIP event between all launches: the per-kernel "
                         "durations (and the roofline, from their median) rest on this many launches whatever "
                         "--steps is")
    ap.add_argument("--event-every", type=int, default=0,
                    help="additionally record the per-kernel events in every n-th step of the timed region "
                         "(0 = never: an event record idles the GPU for several microseconds, five of them per "
                         "step cost ~8 %% of an OPUS-L step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="set-up, outside warm-up and timed region: run the step back to back for this long "
                         "so that every buffer has been touched and the device has left its idle power state "
                         "before the W warm-up steps (a 2 ms warm-up alone leaves the first ~25 steps 10 %% slow)")
    ap.add_argument("--init-group", action="store_true",
                    help="with one rank: initialise the process group all the same, so that barriers, the "
                         "timing reduction and train-M's all-reduce go through the backend (a group of one)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 ('nccl' is RCCL on ROCm)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal of the real multi-rank path on a box with fewer GPUs than ranks: rank r computes "
                         "on device r mod (devices visible); gloo only (RCCL refuses two ranks on one device), "
                         "workloads without the gradient all-reduce; the line is marked config.shared_device "
                         "and is no scaling measurement")
    ap.add_argument("--rank-timeout", type=float, default=None,
                    help="self-launched ranks (no launcher, --gpus > 1): seconds after which ranks still running "
                         "are terminated, then killed, and the exit status is 124 (default: OPUS_RANK_TIMEOUT_S "
                         "or 600)")
    ap.add_argument("--stub-compute", action="store_true",
                    help="rehearsal of the multi-rank control flow on a host without GPUs: the step is a sleep "
                         "(rank r: 2 ms x (1 + r)), nothing native is loaded; the line is marked data = 'stub'")
    return ap.parse_args(argv)


def make_inputs(Bp, Q, P, seed, device, n_sets=1):
    """Synthetic op inputs: N(0,1) features, every point in range at every level (so the
    byte count below is exact), uniform views, softmax weights, N(0,1) upstream gradient."""
    gen = torch.Generator(device=device).manual_seed(seed)
    feats = [torch.randn((Bp, N_VIEWS, h, w, CHANNELS), generator=gen, device=device) for h, w in LEVELS]
    sets = []
    for _ in range(n_sets):
        xy = torch.rand((Bp, Q, P, 2), generator=gen, device=device)
        view = torch.randint(0, N_VIEWS, (Bp, Q, P, 1), generator=gen, device=device).float() / (N_VIEWS - 1)
        loc = torch.cat([xy, view], -1).contiguous()
        w = torch.softmax(torch.randn((Bp, Q, P, len(LEVELS)), generator=gen, device=device), -1)
        g = torch.randn((Bp, Q, CHANNELS, P), generator=gen, device=device)
        sets.append((loc, w, g))
    return feats, sets


def camera_sets(Bp, Q, P, seed, device, n_sets, from_queries=False):
    """Caller-side inputs of sampling_4d: 3-D points in pc_range = [-40,-40,-1, 40,40,5.4]
    (configs/opus-*: point_cloud_range), softmax scale weights, ring-of-cameras occ2img."""
    import numpy as np
    from tools.synthetic_cameras import synthetic_occ2img
    T, G, N, L = 8, 4, N_VIEWS, len(LEVELS)
    B = Bp // (T * G)
    gen = torch.Generator(device=device).manual_seed(seed)
    lo = torch.tensor([-40.0, -40.0, -1.0], device=device)
    hi = torch.tensor([40.0, 40.0, 5.4], device=device)
    occ2img = torch.from_numpy(synthetic_occ2img(np.random.default_rng(seed), B, T, N, 256, 704)).to(device)
    sets = []
    for _ in range(n_sets):
        if from_queries:
            # models/opus_transformer.py:321-327: zero offset weights, bias ~ U(-0.5, 0.5); a query
            # holds 8 points within about a metre of its centre
            E, R = 256, 8
            qc = torch.rand((B, Q, 1, 3), generator=gen, device=device)
            qp = (qc + 0.02 * torch.randn((B, Q, R, 3), generator=gen, device=device)).clamp(0, 1)
            feat = torch.randn((B, Q, E), generator=gen, device=device)
            ob = torch.rand((G * P * 3,), generator=gen, device=device) - 0.5
            sw_w = 0.02 * torch.randn((G * P * L, E), generator=gen, device=device)
            pts, sw = opus_sampling.sampling_inputs(
                qp, feat, torch.zeros((G * P * 3, E), device=device), ob, sw_w,
                torch.zeros((G * P * L,), device=device), T, G, P, L, [-40.0, -40.0, -1.0, 40.0, 40.0, 5.4])
            pts, sw = pts.contiguous(), sw.contiguous()
        else:
            pts = torch.rand((B, Q, T, G, P, 3), generator=gen, device=device) * (hi - lo) + lo
            sw = torch.softmax(torch.randn((B, Q, G, T, P, L), generator=gen, device=device), -1)
        sets.append((pts, sw, occ2img))
    return sets


def unique_feature_bytes(loc, n_views):
    """Compulsory feature traffic of one pass: 256 B x the number of DISTINCT (slab, view, level,
    pixel) corners the points touch (what an infinite cache would fetch)."""
    Bp = loc.shape[0]
    x, y = loc[..., 0].flatten(1), loc[..., 1].flatten(1)
    view = torch.round(loc[..., 2].flatten(1) * (n_views - 1)).long()
    slab = torch.arange(Bp, device=loc.device)[:, None] * n_views + view
    total = 0
    for H, W in LEVELS:
        h_im, w_im = y * (H - 1), x * (W - 1)
        ok = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        h0, w0 = torch.floor(h_im).long(), torch.floor(w_im).long()
        keys = []
        for dh in (0, 1):
            for dw_ in (0, 1):
                hh, ww = h0 + dh, w0 + dw_
                inside = ok & (hh >= 0) & (hh < H) & (ww >= 0) & (ww < W)
                keys.append(((slab * H + hh) * W + ww)[inside])
        total += torch.unique(torch.cat(keys)).numel() * CHANNELS * 4
    return total


def copy_bandwidth(device, nbytes=512 << 20, reps=5):
    """Measured device-to-device copy rate (read + write bytes per second) on this box."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def alg_bytes(npts, feat_bytes):
    """Algorithmic (no-reuse) bytes per launch, L = 4 levels, C = 64 fp32: a corner vector is 256 B,
    16 corners per point.  Whole passes are the contract figures: forward 4380 B/point, backward
    8504 B/point + F (F = the d-feat buffer, written once).  The backward's 8504 B are split over
    its kernels without counting anything twice: the d-feat kernel gets grad_out, loc / weights and
    the d-feat updates (4380 B + F), the gather the feature re-read and the d loc / d w stores
    (4124 B); what the implementation moves on top (the sort's entries, the gather's second read of
    grad_out) is overhead, not algorithm, and appears in no pass figure."""
    L, C = len(LEVELS), CHANNELS
    corner = L * 4 * C * 4                    # 4096 B of feature reads or d-feat updates per point
    vec = C * 4                               # one 256-B output / gradient vector per point
    small = 12 + 4 * L                        # loc + weights (or their gradients)
    return {
        "msmv_fwd_c64": npts * (corner + vec + small),
        "msmv_fwd_sorted": npts * (corner + vec + small),
        "msmv_sort+fwd_sorted": npts * (corner + vec + small),
        "msmv_bwd_owner_dfeat": npts * (vec + small + corner) + feat_bytes,
        "msmv_bwd_owner_dfeat(accumulate)": npts * (vec + small + corner) + 2 * feat_bytes,
        "msmv_bwd_gather_c64": npts * (corner + small),
        # caller-side kernels: 12 B point + 4L B weights in, the same out
        "msmv_project_pack": npts * (12 + 4 * L) * 2,
        "msmv_project_pack_bwd": npts * ((12 + 4 * L) * 2 + 12 + 12),
        "forward": npts * (corner + vec + small),
        "backward": npts * (vec + small + corner + corner + small) + feat_bytes,
    }


def compulsory_bytes(npts, feat_bytes, uniq, n_layers):
    """Bytes each kernel cannot avoid moving across the L2 <-> fabric boundary, per launch, for
    THESE inputs: `uniq` = 256 B x the distinct (slab, view, level, pixel) corners the points touch
    (mean over the layers' point sets), every dense tensor once.  The sort's 32-B entries count
    where a kernel reads them: they are how this implementation finds its points."""
    L = len(LEVELS)
    vec, small, entry = CHANNELS * 4, 12 + 4 * L, 32
    per_set = npts * (vec + entry)            # a layer's gradient rows + sorted entries
    return {
        "msmv_fwd_c64": uniq + npts * (vec + small),
        "msmv_fwd_sorted": uniq + npts * (vec + entry),
        "msmv_sort+fwd_sorted": uniq + npts * (vec + entry) + npts * (small + entry),
        "msmv_bwd_owner_dfeat": feat_bytes + per_set,
        "msmv_bwd_owner_dfeat(accumulate)": 2 * uniq + per_set,       # touched pixels read and written back
        f"msmv_bwd_owner_dfeat({n_layers} layers)": feat_bytes + n_layers * per_set,
        "msmv_bwd_gather_c64": uniq + per_set + npts * small,
        "msmv_bin_sort": npts * (small + entry),
        "msmv_bwd_atomic(+clear)": feat_bytes + 2 * uniq + npts * (vec + 2 * small),
        "msmv_bwd_atomic(accumulate)": 2 * uniq + npts * (vec + 2 * small),
        "msmv_project_pack": npts * small * 2,
        "msmv_project_pack_bwd": npts * (small * 2 + 12 + 12),
        # every feature (or d-feat) byte read once and written once
        "msmv_relayout(4 levels)": 2 * feat_bytes,
        "msmv_relayout(4 levels, inverse)": 2 * feat_bytes,
    }


class Step:
    """One decoder layer's forward (+ backward) with an event pair around every launch."""

    def __init__(self, feats, sets, do_bwd, algo, n_fwd=1, caller=None, defer=False, fwd_sorted=False, backbone=None):
        self.feats, self.sets, self.do_bwd, self.n_fwd = feats, sets, do_bwd, n_fwd
        # backbone = the feature maps as the backbone emits them, [B, T*N, G*C, H, W] per level: the
        # step then begins with their re-layout into `feats` (models/opus_transformer.py:113-126) and,
        # with a backward, ends with the inverse re-layout of the d feats
        self.backbone = backbone
        self.dbackbone = [torch.empty_like(b) for b in backbone] if backbone and do_bwd else None
        self.algo = algo
        # deferred d feats: every layer sorts and gathers into a workspace of its own, one
        # d-feat pass over all layers closes the step (opus_msmv_dfeat_multi)
        self.defer = defer and do_bwd and algo == "owner" and len(sets) > 1
        # caller = list of (sample_points, scale_weights, occ2img), one per set: the step then
        # includes the projection kernels and the op runs in row mode ([B,Q,G,T*P,C] tensors)
        self.caller = caller
        self.rows = (8, 4) if caller else None
        # bin-ordered forward (row mode only): the layer's sort is made by its first forward and
        # serves its other forwards and its backward
        self.fwd_sorted = bool(fwd_sorted and caller and algo == "owner")
        loc, w, _ = sets[0]
        if caller:
            self.dpts = torch.empty_like(caller[0][0])
            self.dsw = torch.empty_like(caller[0][1])
        self.dfeats = [torch.empty_like(f) for f in feats]
        self.dloc, self.dw = torch.empty_like(loc), torch.empty_like(w)
        flags = _lib.BWD_OWNER if algo == "owner" else _lib.BWD_ATOMIC
        need = wrapper.backward_workspace_bytes(feats, loc, w, flags)
        self.ws = torch.empty(max(need, 16), dtype=torch.uint8, device=loc.device)
        self.ws_per_layer = [torch.empty_like(self.ws) for _ in sets] if self.defer else None
        self.events = []          # (kernel name, start, end)
        self._boundary = None

    def _timed(self, name, fn, record):
        """One event per launch boundary: the event that ends a launch also starts the next one
        of the step (two records between launches doubled the idle time between kernels)."""
        if record:
            a = self._boundary
            if a is None:
                a = torch.cuda.Event(enable_timing=True)
                a.record()
            fn()
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self.events.append((name, a, b))
            self._boundary = b
        else:
            fn()

    def run(self, record=True, after_backward=None):
        """`after_backward(layer)` is called when a layer's backward launches have been issued (the
        training workload hands the layer's gradient buckets to the all-reduce there)."""
        self._boundary = None
        if self.backbone:
            self._timed("msmv_relayout(4 levels)", lambda: self._relayout(self.backbone, self.feats, 0), record)
        for layer, (loc, w, g) in enumerate(self.sets):
            # layers of one decoder sample the same feature maps: the first backward writes
            # d feats, the others add to it (no zero-fill, no separate add pass)
            acc = _lib.BWD_ACCUMULATE if layer else 0
            rows = self.rows
            for _ in range(self.n_fwd):
                if self.caller:
                    pts, sw, occ2img = self.caller[layer]
                    self._timed("msmv_project_pack", lambda: opus_sampling.project_pack_into(
                        pts, sw, occ2img, 256, 704, N_VIEWS, loc, w), record)
                if self.fwd_sorted:
                    ws_l = self.ws_per_layer[layer] if self.ws_per_layer else self.ws
                    self._timed("msmv_sort+fwd_sorted" if _ == 0 else "msmv_fwd_sorted",
                                lambda pre=_ > 0: wrapper.msmv_forward_sorted(self.feats, loc, w, rows=rows, workspace=ws_l,
                                                                              presorted=pre), record)
                else:
                    self._timed("msmv_fwd_c64", lambda: wrapper.msmv_forward(self.feats, loc, w, rows=rows), record)
            if not self.do_bwd:
                continue
            args = (g, self.feats, loc, w, self.dfeats, self.dloc, self.dw, self.ws)
            if self.fwd_sorted and self.defer:
                self._timed("msmv_bwd_gather_c64", lambda: wrapper.msmv_backward_into(
                    g, self.feats, loc, w, None, self.dloc, self.dw, self.ws_per_layer[layer],
                    flags=_lib.BWD_OWNER | _lib.BWD_PRESORTED | _lib.BWD_STAGE_GATHER, rows=rows), record)
            elif self.fwd_sorted and not record:
                wrapper.msmv_backward_into(*args, flags=_lib.BWD_OWNER | _lib.BWD_PRESORTED | acc, rows=rows)
            elif self.fwd_sorted:
                # (in the order the one-call form issues them: the gather first)
                for name, bit in (("msmv_bwd_gather_c64", _lib.BWD_STAGE_GATHER), ("msmv_bwd_owner_dfeat", _lib.BWD_STAGE_DFEAT)):
                    self._timed(name if not (acc and bit == _lib.BWD_STAGE_DFEAT) else name + "(accumulate)",
                                lambda bit=bit: wrapper.msmv_backward_into(
                                    *args, flags=_lib.BWD_OWNER | _lib.BWD_PRESORTED | bit | acc, rows=rows), record)
            elif self.defer:
                dargs = (g, self.feats, loc, w, None, self.dloc, self.dw, self.ws_per_layer[layer])
                if not record:
                    wrapper.msmv_backward_into(*dargs, flags=_lib.BWD_OWNER | _lib.BWD_STAGE_BIN | _lib.BWD_STAGE_GATHER,
                                               rows=rows)
                else:
                    for name, bit in (("msmv_bin_sort", _lib.BWD_STAGE_BIN), ("msmv_bwd_gather_c64", _lib.BWD_STAGE_GATHER)):
                        self._timed(name, lambda bit=bit: wrapper.msmv_backward_into(
                            *dargs, flags=_lib.BWD_OWNER | bit, rows=rows), record)
            elif self.algo == "owner" and not record:
                # the product path: one call issues the sort, the gather and the d-feat kernels
                wrapper.msmv_backward_into(*args, flags=_lib.BWD_OWNER | acc, rows=rows)
            elif self.algo == "owner":
                # the same launches stage by stage, so that each kernel gets its own event pair
                for name, bit in (("msmv_bin_sort", _lib.BWD_STAGE_BIN),
                                  ("msmv_bwd_gather_c64", _lib.BWD_STAGE_GATHER),
                                  ("msmv_bwd_owner_dfeat", _lib.BWD_STAGE_DFEAT)):
                    self._timed(name if not (acc and bit == _lib.BWD_STAGE_DFEAT) else name + "(accumulate)",
                                lambda bit=bit: wrapper.msmv_backward_into(
                                    *args, flags=_lib.BWD_OWNER | bit | acc, rows=rows), record)
            else:
                self._timed("msmv_bwd_atomic(accumulate)" if acc else "msmv_bwd_atomic(+clear)",
                            lambda: wrapper.msmv_backward_into(*args, flags=_lib.BWD_ATOMIC | acc, rows=rows), record)
            if self.caller:
                self._timed("msmv_project_pack_bwd", lambda: opus_sampling.project_pack_backward_into(
                    self.dloc, self.dw, pts, occ2img, loc, 256, 704, N_VIEWS, self.dpts, self.dsw), record)
            if after_backward is not None:
                after_backward(layer)
        if self.defer:
            _, Q, P, _ = self.sets[0][0].shape
            cap = _lib.load().opus_msmv_dfeat_max_sets()
            for i in range(0, len(self.sets), cap):
                part = range(i, min(i + cap, len(self.sets)))
                self._timed(f"msmv_bwd_owner_dfeat({len(part)} layers)", lambda: wrapper.msmv_dfeat_multi(
                    [self.sets[k][2] for k in part], [self.ws_per_layer[k] for k in part], self.feats, Q, P,
                    self.dfeats, rows=self.rows, accumulate=i > 0), record)

        if self.dbackbone:
            self._timed("msmv_relayout(4 levels, inverse)", lambda: self._relayout(self.dfeats, self.dbackbone, 1), record)

    def _relayout(self, src, dst, inverse):
        T, G = self.rows
        for a, b in zip(src, dst):
            raw = b if inverse else a                         # the backbone-layout tensor of the pair
            B, TN, GC, H, W = raw.shape
            opus_sampling._Relayout._run(a, b, (B, T, TN // T, G, GC // G, H, W), inverse)

    def kernel_ms(self):
        """name -> (mean, launches, p10, p50, p90) in ms."""
        per = {}
        for name, a, b in self.events:
            per.setdefault(name, []).append(a.elapsed_time(b))
        out = {}
        for k, v in per.items():
            v = sorted(v)
            pick = lambda f: v[min(len(v) - 1, int(f * len(v)))]      # noqa: E731
            out[k] = (statistics.mean(v), len(v), pick(0.1), pick(0.5), pick(0.9))
        return out


def cpu_baseline(variant, do_bwd, slabs=8, reps=5):
    """The reference's grid_sample formulation (oracle/torch_ref.py) on the host cores, on
    `slabs` of the workload's 32 batch slabs; median of `reps` after one warm-up."""
    from oracle import torch_ref
    Q, P = VARIANTS[variant]
    gen = torch.Generator().manual_seed(1234)
    feats = [torch.randn((slabs, CHANNELS, N_VIEWS, h, w), generator=gen).requires_grad_(do_bwd)
             for h, w in LEVELS]
    xy = torch.rand((slabs, Q, P, 2), generator=gen)
    view = torch.randint(0, N_VIEWS, (slabs, Q, P, 1), generator=gen).float() / (N_VIEWS - 1)
    loc = torch.cat([xy, view], -1).requires_grad_(do_bwd)
    w = torch.softmax(torch.randn((slabs, Q, P, len(LEVELS)), generator=gen), -1).requires_grad_(do_bwd)
    g = torch.randn((slabs, Q, CHANNELS, P), generator=gen)
    times = []
    for i in range(reps + 1):
        for t in feats + [loc, w]:
            t.grad = None
        t0 = time.perf_counter()
        out = torch_ref.grid_sample_ref(feats, loc, w)
        if do_bwd:
            out.backward(g)
        dt = time.perf_counter() - t0
        if i:
            times.append(dt)
    npts = slabs * Q * P
    return {"value": npts / statistics.median(times), "unit": "sampled points/s",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{slabs} of the {SLABS_PER_SAMPLE} batch slabs of the OPUS-{variant} workload "
                      f"({npts} points), {'forward + backward' if do_bwd else 'forward'}, PyTorch grid_sample "
                      f"formulation (oracle/torch_ref.py), median of {reps} runs after 1 warm-up"}


def wl_has_allreduce(workload):
    return workload == "train-M"


def traffic_key(workload, batch, points):
    return f"{workload}|batch={batch}|points={points}"


def load_traffic(workload, batch, points, kernel):
    """L2 <-> fabric bytes per launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE; Infinity-Cache hits
    included) of `kernel` from the committed PMC passes of exactly this workload, batch and point
    distribution (profiles/traffic.json, written by tools/make_profile_summary.py), else None:
    a figure measured on another configuration is no figure."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        conf = t.get(traffic_key(workload, batch, points), {})
        # a figure measured on other kernel sources is no figure either
        from tools.buildinfo import csrc_hash
        if conf.get("_csrc_sha256") != csrc_hash():
            return None
        rec = conf.get(kernel)
        return rec if rec is None or isinstance(rec, dict) else {"bytes": rec}
    except (OSError, ValueError):
        return None


class StubStep:
    """--stub-compute: stands in for Step so that the rank logic around it (sharding, seeds,
    fences, the slowest rank's time, rank 0's line) can be rehearsed with gloo on CPUs."""

    def __init__(self, rank, n_layers=1):
        self.rank = rank
        self.n_layers = n_layers
        self.defer = False

    def run(self, record=True, after_backward=None):
        for layer in range(self.n_layers):
            time.sleep(0.002 * (1 + self.rank) / self.n_layers)
            if after_backward is not None:
                after_backward(layer)

    def kernel_ms(self):
        ms = 2.0 * (1 + self.rank)
        return {"stub_sleep": (ms, 1, ms, ms, ms)}


class GradReducer:
    """The DDP-style gradient all-reduce of the training configuration (train.py:123 of the reference
    wraps the model in DistributedDataParallel, which reduces a bucket as soon as the backward sweep
    has produced its gradients): the step's buckets are dealt out over its layers, a layer's share is
    handed to the collective when that layer's backward has been issued, and the step joins the
    collective once, at its end.  Nothing of it is on the op's data path.

    Device tensors (RCCL): every bucket is reduced on a side stream behind an event recorded on the
    compute stream - the buckets of layer i are on the wire while layers i + 1 .. run (the last
    layer's while the deferred d-feat pass runs); `join` makes the compute stream wait for the side
    stream.  Host tensors (gloo): asynchronous work handles, waited for in `join`."""

    def __init__(self, dist, buckets, n_layers, comm_stream=None):
        self.dist, self.buckets, self.n_layers, self.comm_stream = dist, buckets, n_layers, comm_stream
        self.per_layer = -(-len(buckets) // n_layers)
        self.pending = []
        self.log = []                 # (layer, bucket) in issue order, first step only
        self._logging = True

    def after_backward(self, layer):
        lo = layer * self.per_layer
        mine = range(lo, min(lo + self.per_layer, len(self.buckets)))
        if not mine:
            return
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record()
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                for i in mine:
                    self.dist.all_reduce(self.buckets[i])
        else:
            for i in mine:
                self.pending.append(self.dist.all_reduce(self.buckets[i], async_op=True))
        if self._logging:
            self.log.extend((layer, i) for i in mine)

    def join(self):
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        for w in self.pending:
            w.wait()
        self.pending = []
        self._logging = False


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the ranks here.  Nothing in this process has touched a GPU (importing
        # torch and counting devices does not), and the ranks are new processes, not a re-exec.
        if not args.stub_compute and not args.share_device and torch.cuda.device_count() < args.gpus:
            sys.exit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} HIP device(s) visible")
        sys.exit(odist.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:], timeout_s=args.rank_timeout))
    rank, world, local_rank = odist.env_rank()
    if world != args.gpus:
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    # stdout carries ONE line, the result.  Native libraries write there too (RCCL prints a version
    # banner when its first communicator is made): from here on file descriptor 1 is stderr, and the
    # result line goes to the descriptor saved here.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    stub = args.stub_compute
    if stub:
        device = torch.device("cpu")
        dist = odist.init("gloo")
    else:
        load_product()
        assert torch.cuda.is_available(), "bench.py needs a HIP device"
        assert wrapper.MSMV_CUDA, "libopus_msmv.so is not loaded"
        if args.share_device:
            assert args.backend == "gloo", "--share-device: use --backend gloo"
            local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        # "nccl" is RCCL on ROCm; None when single-process
        dist = odist.init(args.backend, device if args.backend == "nccl" else None, force=args.init_group)

    # ---- workload
    wl = args.workload
    n_layers, n_fwd, allreduce_bytes = 1, 1, 0
    chain = wl.startswith("chain-") or wl.startswith("infer-")
    if chain:
        # the decoder's sampling chain (models/opus_transformer.py:113-136, 372-378): re-layout once,
        # six layers; in training every layer's forward runs twice (activation checkpointing)
        variant, do_bwd, n_layers = wl.split("-")[1], wl.startswith("chain-"), 6
        n_fwd = 2 if do_bwd and not args.no_checkpoint else 1
        assert variant in VARIANTS, wl
    elif wl.startswith("decoder-"):
        variant, do_bwd, n_layers = wl.split("-")[1], True, 6
        assert variant in VARIANTS, wl
    elif wl == "train-M":
        variant, do_bwd, n_layers, n_fwd, allreduce_bytes = "M", True, 6, 2, 300 * 1000 * 1000
    elif wl.startswith("sampling4d-"):
        variant, do_bwd = wl.split("-")[1], True
        assert variant in VARIANTS, wl
    else:
        variant, mode = wl.split("-")
        do_bwd = mode == "fwdbwd"
        assert variant in VARIANTS and mode in ("fwd", "fwdbwd"), wl
    Q, P = VARIANTS[variant]
    Bp = SLABS_PER_SAMPLE * args.batch
    total_slabs = Bp * world
    slab_lo = rank * Bp                                  # this rank's slabs of the job's total_slabs
    if args.split_sample:
        assert args.points == "uniform" and not wl.startswith("sampling4d-"), "--split-sample: op workloads only"
        lo, hi = odist.shard_range(Bp, world, rank)      # every slab is independent: any split is valid
        total_slabs, Bp, slab_lo = Bp, hi - lo, lo
        assert Bp > 0, "more ranks than slabs"
    seed = 1234 + rank
    caller = None
    if stub:
        step, feats, sets = StubStep(rank, n_layers), [], []
    else:
        feats, sets = make_inputs(Bp, Q, P, seed, device, n_sets=n_layers)
        backbone = None
        if args.points != "uniform" or wl.startswith("sampling4d-") or chain:
            cam = camera_sets(Bp, Q, P, seed, device, n_layers, from_queries=args.points == "queries")
            for (loc, w, g), (pts, sw, occ2img) in zip(sets, cam):
                opus_sampling.project_pack_into(pts, sw, occ2img, 256, 704, N_VIEWS, loc, w)
            if wl.startswith("sampling4d-") or chain:
                caller = cam
                sets = [(loc, w, g.reshape(args.batch, Q, 4, 8 * P, CHANNELS)) for loc, w, g in sets]
        if chain:
            gen = torch.Generator(device=device).manual_seed(seed + 77)
            backbone = [torch.randn((args.batch, 8 * N_VIEWS, 4 * CHANNELS, h, w_), generator=gen, device=device)
                        for h, w_ in LEVELS]
        fwd_order = args.fwd_order or ("input" if wl.startswith("infer-") else "bin")
        step = Step(feats, sets, do_bwd, args.bwd_algo, n_fwd=n_fwd, caller=caller, defer=args.dfeat == "deferred",
                    fwd_sorted=fwd_order == "bin", backbone=backbone)
    npts_step = Bp * Q * P * n_layers                    # sampled points per step per rank
    feat_bytes = sum(f.numel() * 4 for f in feats)

    reducer = None
    if allreduce_bytes and dist is not None:
        bucket = 25 * 1000 * 1000 // 4
        # RCCL reduces device tensors on a side stream; gloo is given host tensors only (its staging of
        # device tensors from two processes that share one GPU is where a rehearsal once hung)
        on_device = not stub and args.backend == "nccl"
        grad_buckets = [torch.zeros(bucket, device=device if on_device else "cpu")
                        for _ in range(allreduce_bytes // (bucket * 4))]
        reducer = GradReducer(dist, grad_buckets, n_layers, torch.cuda.Stream() if on_device else None)

    def one_step(record, reduce=True):
        if reducer is not None and reduce:
            step.run(record, after_backward=reducer.after_backward)
            reducer.join()
        else:
            step.run(record)

    # a generational collection of the interpreter's heap (tens of ms with torch loaded) inside
    # the timed loop would stall the launches of one step and be billed to the GPU path
    gc.collect()
    gc.disable()
    # set-up, outside warm-up and timed region: touch every buffer and bring the device out of its
    # idle power state by running the step back to back for --settle-ms
    settle_steps = 0
    if not stub and args.settle_ms > 0:
        t_end = time.perf_counter() + args.settle_ms * 1e-3
        while time.perf_counter() < t_end:
            step.run(False)
            settle_steps += 1
            if settle_steps % 8 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        one_step(False)
    graph = None
    if args.graph:
        assert reducer is None and not stub, "--graph does not cover the all-reduce workload"
        # the library neither allocates nor synchronises: a step is capturable as it is
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step.run(False)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step.run(False)
        graph.replay()
    odist.fence(dist, device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            one_step(args.event_every > 0 and i % args.event_every == 0)
    odist.fence(dist, device)
    # (gloo reduces host tensors; handing it device tensors of a GPU two ranks share can hang)
    elapsed = odist.max_over_ranks(time.perf_counter() - t0, dist,
                                   device if args.backend == "nccl" and not stub else torch.device("cpu"))
    # what the collective costs the step: the same steps without it, timed the same way
    ms_no_comm = None
    if reducer is not None:
        n_cmp = min(args.steps, 20)
        odist.fence(dist, device)
        t1 = time.perf_counter()
        for _ in range(n_cmp):
            one_step(False, reduce=False)
        odist.fence(dist, device)
        ms_no_comm = odist.max_over_ranks(time.perf_counter() - t1, dist,
                                          device if args.backend == "nccl" and not stub else torch.device("cpu")) / n_cmp * 1e3
    # per-kernel durations: evented steps outside the timed region (the all-reduce is not part of them)
    event_steps = 0 if stub else max(args.event_steps, 0)
    for _ in range(event_steps):
        step.run(True)
    if not stub:
        torch.cuda.synchronize()
    gc.enable()

    # what every rank worked on, for rank 0's line
    mine = {"rank": rank, "seed": seed, "slabs": [slab_lo, slab_lo + Bp], "device": str(device)}
    ranks = [mine]
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)

    if rank == 0:
        kms = step.kernel_ms()
        npts_set = Bp * Q * P
        ab = alg_bytes(npts_set, feat_bytes)
        uniq_sets = [unique_feature_bytes(loc, N_VIEWS) for loc, _, _ in sets]
        uniq = sum(uniq_sets) / len(uniq_sets) if uniq_sets else 0
        cb = compulsory_bytes(npts_set, feat_bytes, uniq, n_layers)
        exact = args.points == "uniform"       # every corner in range: the closed forms are exact
        kernels = {}
        for name, (ms, n, p10, p50, p90) in kms.items():
            # rates from the MEDIAN launch duration (one slow launch among a few dozen moves a mean by
            # several per cent); the mean is kept beside it
            rec = {"avg_us": ms * 1e3, "launches": n, "p10_us": p10 * 1e3, "p50_us": p50 * 1e3, "p90_us": p90 * 1e3}
            if name in ab and (exact or name.startswith("msmv_project")):
                rec["alg_bytes"] = ab[name]
                rec["alg_GBps"] = ab[name] / (p50 * 1e-3) / 1e9
            if name in cb:
                rec["compulsory_bytes"] = cb[name]
                rec["compulsory_GBps"] = cb[name] / (p50 * 1e-3) / 1e9
                rec["compulsory_GBps_mean"] = cb[name] / (ms * 1e-3) / 1e9
            t = None if stub else load_traffic(wl, args.batch, args.points, name)
            if t is not None:
                rec["traffic_bytes"] = t["bytes"]
                rec["traffic_GBps"] = t["bytes"] / (p50 * 1e-3) / 1e9
                for k in ("template", "l2_hit_rate"):
                    if k in t:
                        rec[k] = t[k]
            kernels[name] = rec

        def roofline_of(name):
            k = kernels[name]
            return {"bound": "hbm", "kernel": name,
                    # compulsory bytes of the kernel / its median HIP-event launch time: <= peak
                    "achieved": k.get("compulsory_GBps"), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": (k["compulsory_GBps"] / HBM_PEAK_GBPS) if "compulsory_GBps" in k else None,
                    "frac_from_mean": (k["compulsory_GBps_mean"] / HBM_PEAK_GBPS) if "compulsory_GBps_mean" in k else None,
                    # measured L2 <-> fabric bytes per launch (PMC passes of this configuration on this
                    # build of the kernels) or null
                    "traffic": k.get("traffic_bytes"),
                    "traffic_GBps": k.get("traffic_GBps"),
                    "traffic_frac": (k["traffic_GBps"] / HBM_PEAK_GBPS) if "traffic_GBps" in k else None,
                    "compulsory_bytes_per_launch": k.get("compulsory_bytes"),
                    # the no-reuse figure of the metric's definition: above the peak where caches
                    # serve re-touched pixels
                    "achieved_alg": k.get("alg_GBps"), "alg_bytes_per_launch": k.get("alg_bytes"),
                    "frac_alg": (k["alg_GBps"] / HBM_PEAK_GBPS) if "alg_GBps" in k else None,
                    "avg_launch_us": k["avg_us"], "p50_launch_us": k["p50_us"], "launches": k["launches"],
                    "share_of_kernel_time": k["p50_us"] * k["launches"] / kernel_time_total}

        kernel_time_total = sum(k["p50_us"] * k["launches"] for k in kernels.values())
        dominant = max(kernels, key=lambda k: kernels[k]["p50_us"] * kernels[k]["launches"])
        roofline = roofline_of(dominant)
        roofline["note"] = ("achieved = compulsory bytes (distinct feature pixels x 256 B + the kernel's dense "
                            "reads / writes, from the inputs) / median HIP-event launch time over `launches` "
                            "launches; traffic = rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE at the L2 -> fabric "
                            "boundary (Infinity-Cache hits included); achieved_alg = no-reuse bytes of the "
                            "metric's definition")
        # the kernel furthest below the roofline among those that matter (>= 10 % of the step's kernel
        # time: the launch-latency-sized sort and projection kernels are not bandwidth work)
        heavy = [k for k in kernels if "compulsory_GBps" in kernels[k]
                 and kernels[k]["p50_us"] * kernels[k]["launches"] >= 0.1 * kernel_time_total]
        roofline_worst = roofline_of(min(heavy, key=lambda k: kernels[k]["compulsory_GBps"])) if heavy else None
        pass_bytes = (ab["forward"] * n_fwd + (ab["backward"] if do_bwd else 0)) * n_layers
        pass_compulsory = sum(kernels[k]["compulsory_bytes"] * kms[k][1] for k in kms if "compulsory_bytes" in kernels[k])
        evented_steps = event_steps + (len(range(0, args.steps, args.event_every))
                                       if args.event_every > 0 and graph is None else 0)
        evented_steps = max(evented_steps, 1)
        ms_per_step = elapsed / args.steps * 1e3
        outside = 0.0
        if sets:
            loc0 = sets[0][0]
            outside = ((loc0[..., :2] < 0) | (loc0[..., :2] > 1)).any(-1).float().mean().item()
        copy_gbps = None if stub else copy_bandwidth(device)
        result = {
            "metric": "sampled points/sec (fwd+bwd) + % HBM roofline, OPUS-T/L shape, 1/2/4/8 GPUs",
            "value": total_slabs * Q * P * n_layers * args.steps / elapsed,
            "unit": "sampled points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if args.split_sample else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "stub" if stub else "synthetic",
            "config": {"workload": f"OPUS-{variant} sampling shape, "
                                   f"{'projection + op in the caller layout, ' if caller else ''}"
                                   f"{'forward + backward' if do_bwd else 'forward'}"
                                   f"{', %d decoder layers' % n_layers if n_layers > 1 else ''}"
                                   f"{', 2 forwards per backward + gradient all-reduce' if wl == 'train-M' else ''}"
                                   f"{', with the feature re-layout' + ((', 2 forwards per backward (activation checkpointing)' if n_fwd == 2 else ', no activation checkpointing') + ', d feats back in the backbone layout' if do_bwd else '') if chain else ''}",
                       "name": wl, "Q": Q, "P": P, "slabs_per_gpu": Bp, "views": N_VIEWS,
                       "levels": LEVELS, "channels": CHANNELS, "batch_per_gpu": args.batch,
                       "points_per_step_per_gpu": npts_step, "bwd_algo": args.bwd_algo if do_bwd else None,
                       "dfeat_sum": ("deferred" if step.defer else "accumulate") if n_layers > 1 and do_bwd else None,
                       "fwd_order": "bin" if getattr(step, "fwd_sorted", False) else "input",
                       "points": args.points if (not caller or args.points != "uniform") else "cameras",
                       "launch": ("HIP graph replay" if args.graph else
                                  "one call per kernel, no events inside the timed region"
                                  + (f" but in every {args.event_every}. step" if args.event_every > 0 else ""))
                                 + f"; per-kernel durations from {event_steps} evented steps after the timed region"
                                 + f"; set-up before the warm-up: {settle_steps} untimed steps "
                                   f"({args.settle_ms:g} ms) to touch the buffers and leave the idle power state",
                       "frac_points_outside_image": outside,
                       "backend": None if dist is None else ("gloo" if stub else args.backend),
                       "shared_device": bool(args.share_device),
                       "allreduce_bytes_per_step": sum(b.numel() * 4 for b in reducer.buckets) if reducer else 0,
                       # (layer, bucket) in the order the buckets were handed to the collective
                       "allreduce_schedule": reducer.log if reducer else None,
                       "allreduce_overlap": (None if reducer is None else
                                             "a layer's buckets are reduced on a side stream (RCCL) / by asynchronous "
                                             "work handles (gloo, host tensors) while the following layers run; one "
                                             "join at the end of the step"),
                       "ranks": ranks,
                       "parallelism": (f"slab-parallel over {world} GPUs (one sample split, no data-path collective)"
                                       if args.split_sample else
                                       f"dp{world} (batch-parallel, no data-path collective)")},
            "roofline": roofline,
            "roofline_worst": roofline_worst,
            # step time with the gradient all-reduce minus step time without it (None: no collective)
            "allreduce_exposed_us": None if ms_no_comm is None else (ms_per_step - ms_no_comm) * 1e3,
            "ms_per_step_without_allreduce": ms_no_comm,
            "step_alg_GBps": pass_bytes / (ms_per_step * 1e-3) / 1e9,
            "step_alg_frac_of_peak": pass_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            # the whole step on compulsory bytes: sum over its kernels / step time
            "step_compulsory_bytes": pass_compulsory / evented_steps,
            "step_compulsory_frac_of_peak": pass_compulsory / evented_steps / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "compulsory_feature_bytes_per_pass": uniq,
            "gathered_feature_bytes_per_pass": Bp * Q * P * len(LEVELS) * 4 * CHANNELS * 4,
            "device_copy_GBps_measured": copy_gbps,
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline and not stub:
            result["cpu_baseline"] = cpu_baseline(variant, do_bwd)
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    os.close(result_fd)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
