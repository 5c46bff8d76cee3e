"""Host-side mirror of the reference's ``models/csrc/wrapper.py`` for MI355X.

Same public names and call signatures as the reference module, so that
``from .csrc.wrapper import msmv_sampling, msmv_sampling_pytorch`` (models/opus_sampling.py:5)
and ``from .csrc.wrapper import MSMV_CUDA`` (models/opus_transformer.py:14) keep working:

    MSMV_CUDA                 True when libopus_msmv.so is loaded   (wrapper.py:4-11)
    msmv_sampling_pytorch     channel-first grid_sample formulation  (wrapper.py:14-38)
    MSMVSamplingC2345         autograd Function, 4 levels            (wrapper.py:41-61)
    MSMVSamplingC23456        autograd Function, 5 levels            (wrapper.py:64-84)
    msmv_sampling             dispatcher                             (wrapper.py:87-93)

Differences, all deliberate:
  * the native code is a C-ABI shared library reached through ctypes (opus_amd/_lib.py),
    launched on the caller's current HIP stream and device;
  * by default ``msmv_sampling`` never silently falls back: with the library missing it raises
    (and on a machine with a GPU the import itself fails).  The reference's behaviour - warn at
    import, ``MSMV_CUDA = False``, ``msmv_sampling`` -> ``msmv_sampling_pytorch`` on channel-first
    features (wrapper.py:8-11, :87-93) - is an explicit opt-in: environment variable
    ``OPUS_MSMV_ALLOW_FALLBACK=1`` (read at import) or ``allow_fallback(True)``.  With the
    library loaded the HIP path always runs; the opt-in never routes around it;
  * launch failures raise ``RuntimeError`` (the reference prints and continues,
    msmv_sampling_forward.cu:295-298);
  * the gradient buffers are allocated here with ``torch.empty`` and fully written by the
    library (the reference's host shim allocates zeros, msmv_sampling.cpp:244-249).
"""
import os
import weakref

import torch
import torch.nn.functional as F

from .. import _lib

_allow_fallback = os.environ.get("OPUS_MSMV_ALLOW_FALLBACK", "0") not in ("", "0")

try:
    _lib.load()
    MSMV_CUDA = True
except ImportError as e:
    if torch.cuda.is_available() and not _allow_fallback:
        # a GPU is present but the extension is not: that is a broken install, not a
        # configuration to limp along in
        raise
    if _allow_fallback:
        print('Warning: failed to load the HIP extension libopus_msmv.so, performance may be hurt: '
              'msmv_sampling() uses the grid_sample formulation (OPUS_MSMV_ALLOW_FALLBACK is set).')
    else:
        print('Warning: failed to load the HIP extension libopus_msmv.so; '
              'msmv_sampling() will raise, only msmv_sampling_pytorch() is usable '
              '(set OPUS_MSMV_ALLOW_FALLBACK=1 for the reference\'s silent fallback).')
    print('Error message:', e)
    MSMV_CUDA = False


def allow_fallback(flag=True):
    """Opt in to (or out of) the reference dispatcher's fallback arm (wrapper.py:87-93): when the
    library is not loaded, ``msmv_sampling`` returns ``msmv_sampling_pytorch(...)`` on
    channel-first features instead of raising."""
    global _allow_fallback
    _allow_fallback = bool(flag)

_BWD_ALGOS = {"auto": _lib.BWD_AUTO, "atomic": _lib.BWD_ATOMIC, "owner": _lib.BWD_OWNER}
_bwd_algo = os.environ.get("OPUS_MSMV_BWD_ALGO", "auto")
if _bwd_algo not in _BWD_ALGOS:
    raise ValueError(f"OPUS_MSMV_BWD_ALGO must be one of {sorted(_BWD_ALGOS)}, got {_bwd_algo!r}")


# Bin-ordered forward (opus_msmv_forward_sorted) in the autograd Functions that can use it (row
# mode and P == 1, owner backward, some input needs a gradient): the sort made for the forward is
# kept and serves the backward.
_fwd_sorted = os.environ.get("OPUS_MSMV_FWD_SORTED", "1") not in ("", "0")


def set_forward_sorted(flag=True):
    """Whether msmv_sampling_rows (and msmv_sampling with one point per query) run their forward
    in bin order and hand the sort on to their backward.  Default on; OPUS_MSMV_FWD_SORTED=0 at
    import turns it off."""
    global _fwd_sorted
    _fwd_sorted = bool(flag)


def set_backward_algo(name):
    """'auto' (default), 'atomic' or 'owner' - see include/opus_msmv.h."""
    global _bwd_algo
    if name not in _BWD_ALGOS:
        raise ValueError(f"unknown backward algorithm {name!r}")
    _bwd_algo = name


def msmv_sampling_pytorch(mlvl_feats, sampling_locations, scale_weights):
    """grid_sample formulation on channel-FIRST features (reference wrapper.py:14-38).

    mlvl_feats: list of [B, C, N, H, W]; sampling_locations [B, Q, P, 3] in [0, 1];
    scale_weights [B, Q, P, L].  Returns [B, Q, C, P].  The view coordinate is the depth
    axis of a trilinear sample here, whereas the device kernels round it to a view.
    """
    assert scale_weights.shape[-1] == len(mlvl_feats)
    B, C = mlvl_feats[0].shape[:2]
    _, Q, P, _ = sampling_locations.shape
    grid = (sampling_locations * 2 - 1).unsqueeze(3)          # [B, Q, P, 1, 3]
    total = torch.zeros([B, C, Q, P], device=mlvl_feats[0].device)
    for lvl, feat in enumerate(mlvl_feats):
        sampled = F.grid_sample(feat, grid, mode='bilinear', padding_mode='zeros',
                                align_corners=True)[..., 0]   # [B, C, Q, P]
        total += sampled * scale_weights[..., lvl].reshape(B, 1, Q, P)
    return total.permute(0, 2, 1, 3)


# --------------------------------------------------------------------------------------
# argument checks (the reference's AT_ASSERTM list, msmv_sampling.cpp:106-125, plus dtype /
# device / shape consistency, which the reference leaves unchecked)
# --------------------------------------------------------------------------------------
def _check_inputs(feats, loc, w):
    if not MSMV_CUDA:
        raise RuntimeError("libopus_msmv.so is not loaded: build it with `make -C opus_amd/csrc`")
    L = len(feats)
    for f in feats:
        if not f.is_contiguous():
            raise RuntimeError("value tensor has to be contiguous")
        if not f.is_cuda:
            raise RuntimeError("value must be a CUDA tensor")
    if not loc.is_contiguous():
        raise RuntimeError("sampling_loc tensor has to be contiguous")
    if not w.is_contiguous():
        raise RuntimeError("attn_weight tensor has to be contiguous")
    if not loc.is_cuda:
        raise RuntimeError("sampling_loc must be a CUDA tensor")
    if not w.is_cuda:
        raise RuntimeError("attn_weight must be a CUDA tensor")
    dev = feats[0].device
    for t in list(feats) + [loc, w]:
        if t.dtype != torch.float32:
            raise RuntimeError("msmv_sampling expects float32 tensors")
        if t.device != dev:
            raise RuntimeError("msmv_sampling expects all tensors on one device")
    if feats[0].dim() != 5 or loc.dim() != 4 or w.dim() != 4 or loc.shape[-1] != 3:
        raise RuntimeError("msmv_sampling: expected feats [B,N,H,W,C], loc [B,Q,P,3], weights [B,Q,P,L]")
    B, N, _, _, C = feats[0].shape
    _, Q, P, _ = loc.shape
    if P > _lib.MAX_POINTS:
        raise RuntimeError("num_point exceed limits")
    for f in feats:
        if f.dim() != 5 or f.shape[0] != B or f.shape[1] != N or f.shape[4] != C:
            raise RuntimeError("msmv_sampling: feature levels disagree on batch, views or channels")
    if loc.shape[0] != B or tuple(w.shape) != (B, Q, P, L):
        raise RuntimeError("msmv_sampling: sampling_loc / attn_weight shape mismatch")
    hw = [d for f in feats for d in f.shape[2:4]]
    return L, B, C, N, Q, P, hw


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def _check_rows(B, rows):
    """rows = (T, G): the op's batch is Bs*T*G and out / grad_out are [Bs, Q, G, T*P, C]."""
    T, G = int(rows[0]), int(rows[1])
    if T < 1 or G < 1 or B % (T * G) != 0:
        raise RuntimeError(f"row mode: batch {B} is not a multiple of T*G = {T}*{G}")
    return B // (T * G), T, G


def msmv_forward(feats, loc, w, flags=0, rows=None):
    """Raw forward: feats channel-last [B,N,H,W,C] -> out [B,Q,C,P] (no autograd).
    With ``rows=(T, G)``: B = Bs*T*G and out is [Bs, Q, G, T*P, C], i.e. already in the layout
    the reference's sampling_4d returns (models/opus_sampling.py:119-122)."""
    L, B, C, N, Q, P, hw = _check_inputs(feats, loc, w)
    lib = _lib.load()
    fptrs, hw_arr = _lib.ptr_array([f.data_ptr() for f in feats]), _lib.int_array(hw)
    with torch.cuda.device(loc.device):
        if rows is None:
            out = torch.empty((B, Q, C, P), dtype=torch.float32, device=loc.device)
            rc = lib.opus_msmv_forward_ex(L, fptrs, hw_arr, loc.data_ptr(), w.data_ptr(), B, C, N, Q, P,
                                          out.data_ptr(), flags, _stream_ptr(loc.device))
        else:
            Bs, T, G = _check_rows(B, rows)
            out = torch.empty((Bs, Q, G, T * P, C), dtype=torch.float32, device=loc.device)
            rc = lib.opus_msmv_forward_rows(L, fptrs, hw_arr, loc.data_ptr(), w.data_ptr(), Bs, T, G, C, N, Q, P,
                                            out.data_ptr(), flags, _stream_ptr(loc.device))
    _lib.check(rc, "opus_msmv_forward")
    return out


def msmv_forward_sorted(feats, loc, w, rows=None, workspace=None, presorted=False):
    """Forward in bin order -> (out, workspace) or None where it does not apply (then use
    msmv_forward).  The workspace holds the sort: hand it to a later call with ``presorted=True``
    (the same loc / weights) or to msmv_backward_into with ``_lib.BWD_PRESORTED``."""
    L, B, C, N, Q, P, hw = _check_inputs(feats, loc, w)
    lib = _lib.load()
    if rows is None and P != 1:
        return None
    if not _aligned16(w, *feats):               # (what the library would refuse, before anything is allocated)
        return None
    need = lib.opus_msmv_backward_workspace_bytes(L, B, C, N, Q, P, _lib.BWD_OWNER)
    if need == 0:
        return None
    with torch.cuda.device(loc.device):
        if workspace is None:
            if presorted:
                raise RuntimeError("presorted=True needs the workspace that holds the sort")
            workspace = torch.empty(need, dtype=torch.uint8, device=loc.device)
        if rows is None:
            dims, shape = (B, 0, 0), (B, Q, C, P)
        else:
            Bs, T, G = _check_rows(B, rows)
            dims, shape = (Bs, T, G), (Bs, Q, G, T * P, C)
        out = torch.empty(shape, dtype=torch.float32, device=loc.device)
        rc = lib.opus_msmv_forward_sorted(
            L, _lib.ptr_array([f.data_ptr() for f in feats]), _lib.int_array(hw), loc.data_ptr(), w.data_ptr(),
            *dims, C, N, Q, P, out.data_ptr(), workspace.data_ptr(), workspace.numel(),
            _lib.FWD_PRESORTED if presorted else 0, _stream_ptr(loc.device))
    if rc == _lib.EINVAL_FLAGS and not presorted:     # a shape the sort does not take (e.g. more than 16 views)
        return None
    _lib.check(rc, "opus_msmv_forward_sorted")
    return out, workspace


def _aligned16(*tensors):
    return all(t.data_ptr() % 16 == 0 for t in tensors)


def backward_workspace_bytes(feats, loc, w, flags=0):
    """Device scratch the backward needs for these tensors (0 for the atomic algorithm)."""
    L, B, C, N, Q, P, _ = _check_inputs(feats, loc, w)
    return _lib.load().opus_msmv_backward_workspace_bytes(L, B, C, N, Q, P, flags)


def msmv_backward_into(grad_out, feats, loc, w, dfeats, dloc, dw, workspace=None, flags=0, rows=None):
    """Raw backward into caller-owned buffers (no allocation here when ``workspace`` is
    given).  ``flags``: algorithm | ACCUMULATE | STAGE_* bits of include/opus_msmv.h.
    ``rows=(T, G)``: grad_out is [Bs, Q, G, T*P, C] (see msmv_forward).  ``dfeats`` may be None
    when the stages asked for do not include the d-feat pass, ``dloc`` / ``dw`` when they do not
    include the gather."""
    L, B, C, N, Q, P, hw = _check_inputs(feats, loc, w)
    if rows is None:
        want = (B, Q, C, P)
    else:
        Bs, T, G = _check_rows(B, rows)
        want = (Bs, Q, G, T * P, C)
    if not grad_out.is_contiguous() or tuple(grad_out.shape) != want \
            or grad_out.dtype != torch.float32 or grad_out.device != loc.device:
        raise RuntimeError(f"grad_output must be a contiguous float32 {list(want)} tensor on the inputs' device")
    for d, f in zip(dfeats or [], feats):
        if d.shape != f.shape or not d.is_contiguous() or d.dtype != torch.float32 or d.device != f.device:
            raise RuntimeError("d feats must match feats in shape, dtype and device and be contiguous")
    if (dloc is None) != (dw is None):
        raise RuntimeError("dloc and dweights are given or left out together")
    if dloc is not None and (dloc.shape != loc.shape or dw.shape != w.shape or not dloc.is_contiguous()
                             or not dw.is_contiguous()):
        raise RuntimeError("dloc / dweights must match sampling_loc / attn_weight and be contiguous")
    lib = _lib.load()
    with torch.cuda.device(loc.device):
        need = lib.opus_msmv_backward_workspace_bytes(L, B, C, N, Q, P, flags)
        if workspace is None:
            workspace = torch.empty(max(need, 1), dtype=torch.uint8, device=loc.device)
        head = (L, grad_out.data_ptr(), _lib.ptr_array([f.data_ptr() for f in feats]),
                _lib.int_array(hw), loc.data_ptr(), w.data_ptr())
        tail = (_lib.ptr_array([d.data_ptr() for d in dfeats]) if dfeats is not None else None,
                dloc.data_ptr() if dloc is not None else None, dw.data_ptr() if dw is not None else None,
                workspace.data_ptr(), workspace.numel() * workspace.element_size(), flags,
                _stream_ptr(loc.device))
        if rows is None:
            rc = lib.opus_msmv_backward(*head, B, C, N, Q, P, *tail)
        else:
            rc = lib.opus_msmv_backward_rows(*head, Bs, T, G, C, N, Q, P, *tail)
    _lib.check(rc, "opus_msmv_backward")


def msmv_dfeat_multi(grad_outs, workspaces, feats, Q, P, dfeats, rows=None, accumulate=False):
    """One d-feat pass over several layers' sorted point sets (opus_msmv_dfeat_multi): set i is
    ``grad_outs[i]`` + the workspace its layer's sort stage filled.  ``feats`` gives the sizes."""
    lib = _lib.load()
    L = len(feats)
    B, N, _, _, C = feats[0].shape
    hw = [d for f in feats for d in f.shape[2:4]]
    dev = feats[0].device
    if rows is None:
        dims, want = (B, 0, 0), (B, Q, C, P)
    else:
        Bs, T, G = _check_rows(B, rows)
        dims, want = (Bs, T, G), (Bs, Q, G, T * P, C)
    for g in grad_outs:
        if not g.is_contiguous() or tuple(g.shape) != want or g.dtype != torch.float32 or g.device != dev:
            raise RuntimeError(f"grad_output must be a contiguous float32 {list(want)} tensor on the inputs' device")
    for d, f in zip(dfeats, feats):
        if d.shape != f.shape or not d.is_contiguous() or d.dtype != torch.float32 or d.device != dev:
            raise RuntimeError("d feats must match feats in shape, dtype and device and be contiguous")
    if len(grad_outs) != len(workspaces) or not grad_outs:
        raise RuntimeError("one workspace per grad_output, at least one")
    ws_bytes = min(x.numel() * x.element_size() for x in workspaces)
    with torch.cuda.device(dev):
        rc = lib.opus_msmv_dfeat_multi(L, len(grad_outs), _lib.ptr_array([g.data_ptr() for g in grad_outs]),
                                       _lib.ptr_array([x.data_ptr() for x in workspaces]), ws_bytes,
                                       _lib.int_array(hw), *dims, C, N, Q, P,
                                       _lib.ptr_array([d.data_ptr() for d in dfeats]),
                                       _lib.BWD_ACCUMULATE if accumulate else 0, _stream_ptr(dev))
    _lib.check(rc, "opus_msmv_dfeat_multi")


def msmv_backward(grad_out, feats, loc, w, algo=None, dfeats=None, rows=None, out_dfeats=None, sorted_ws=None,
                  need=(True, True)):
    """Raw backward -> (dfeats list, dloc, dw).  ``dfeats`` given => accumulate into them;
    ``out_dfeats`` given => overwrite those buffers instead of allocating new ones; ``sorted_ws``:
    the workspace msmv_forward_sorted left for these loc / weights (owner algorithm, sort skipped).

    ``need = (d feats wanted, d loc / d weights wanted)``: the owner algorithm computes the two groups
    in separate passes (d-feat pass, gather), so a group nobody asked for - frozen feature maps, or
    locations and weights that are constants - is not computed and comes back as None(s).  The
    reference always computes all three (msmv_sampling.cpp:212-281)."""
    need_dfeat, need_dlw = bool(need[0]), bool(need[1])
    if not (need_dfeat or need_dlw):
        return [None] * len(feats), None, None
    name = _bwd_algo if algo is None else algo
    flags = _BWD_ALGOS[name]
    if sorted_ws is not None:
        flags = _lib.BWD_OWNER | _lib.BWD_PRESORTED
    accumulate = dfeats is not None
    if sorted_ws is not None and (grad_out.data_ptr() % 16 or any(d.data_ptr() % 16 for d in (dfeats or out_dfeats or []))):
        sorted_ws, flags = None, _BWD_ALGOS[name]
    if not (need_dfeat and need_dlw):
        # one group only: possible where the owner algorithm takes the tensors (its passes are stages)
        staged = sorted_ws is not None or (
            name != "atomic" and _aligned16(grad_out, *feats, *(dfeats or out_dfeats or []))
            and backward_workspace_bytes(feats, loc, w, _lib.BWD_OWNER) > 0)
        if staged:
            flags = _lib.BWD_OWNER | (_lib.BWD_PRESORTED if sorted_ws is not None else _lib.BWD_STAGE_BIN)
            flags |= _lib.BWD_STAGE_DFEAT if need_dfeat else _lib.BWD_STAGE_GATHER
    if need_dfeat or not (flags & _lib.BWD_STAGE_GATHER):
        if dfeats is None:
            dfeats = out_dfeats if out_dfeats is not None else [torch.empty_like(f) for f in feats]
    else:
        dfeats = None
    if accumulate:
        flags |= _lib.BWD_ACCUMULATE
    dloc = dw = None
    if need_dlw or not (flags & _lib.BWD_STAGE_DFEAT):
        dloc = torch.empty_like(loc)
        dw = torch.empty_like(w)
    msmv_backward_into(grad_out, feats, loc, w, dfeats, dloc, dw, workspace=sorted_ws, flags=flags, rows=rows)
    if not need_dfeat:
        dfeats = [None] * len(feats)
    if not need_dlw:
        dloc = dw = None
    return dfeats, dloc, dw


class _MSMVSampling(torch.autograd.Function):
    """Any level count in [1, 8]; the two named Functions below fix it to 4 and 5."""

    @staticmethod
    def forward(ctx, *args):
        *feats, sampling_locations, scale_weights = args
        ctx.save_for_backward(*feats, sampling_locations, scale_weights)
        ctx.sorted_ws = None
        # the point sort pays for itself only where a backward will use it too (it is kept, 32 bytes
        # per point, until the graph is freed); without gradients the input-order kernel is as fast
        if _fwd_sorted and _bwd_algo != "atomic" and any(ctx.needs_input_grad):
            res = msmv_forward_sorted(feats, sampling_locations, scale_weights)
            if res is not None:
                out, ctx.sorted_ws = res
                return out
        return msmv_forward(feats, sampling_locations, scale_weights)

    @staticmethod
    def backward(ctx, grad_output):
        *feats, sampling_locations, scale_weights = ctx.saved_tensors
        n = len(feats)
        need = (any(ctx.needs_input_grad[:n]), any(ctx.needs_input_grad[n:n + 2]))
        dfeats, dloc, dw = msmv_backward(grad_output.contiguous(), feats,
                                         sampling_locations, scale_weights, sorted_ws=ctx.sorted_ws, need=need)
        return (*dfeats, dloc, dw)


class MSMVSamplingC2345(_MSMVSampling):
    @staticmethod
    def forward(ctx, feat_c2, feat_c3, feat_c4, feat_c5, sampling_locations, scale_weights):
        return _MSMVSampling.forward(ctx, feat_c2, feat_c3, feat_c4, feat_c5,
                                     sampling_locations, scale_weights)


class MSMVSamplingC23456(_MSMVSampling):
    @staticmethod
    def forward(ctx, feat_c2, feat_c3, feat_c4, feat_c5, feat_c6, sampling_locations, scale_weights):
        return _MSMVSampling.forward(ctx, feat_c2, feat_c3, feat_c4, feat_c5, feat_c6,
                                     sampling_locations, scale_weights)


class MSMVSamplingRows(torch.autograd.Function):
    """The op fused with its caller's un-pack (no counterpart in the reference's wrapper: it is
    sampling_4d's msmv_sampling call + the permute of models/opus_sampling.py:116-122 as one
    launch each way).  apply(T, G, *feats, loc, weights) -> [Bs, Q, G, T*P, C]."""

    @staticmethod
    def forward(ctx, T, G, *args):
        *feats, sampling_locations, scale_weights = args
        ctx.save_for_backward(*feats, sampling_locations, scale_weights)
        ctx.rows = (T, G)
        ctx.sorted_ws = None
        if _fwd_sorted and _bwd_algo != "atomic" and any(ctx.needs_input_grad):
            res = msmv_forward_sorted(feats, sampling_locations, scale_weights, rows=ctx.rows)
            if res is not None:
                out, ctx.sorted_ws = res
                return out
        return msmv_forward(feats, sampling_locations, scale_weights, rows=ctx.rows)

    @staticmethod
    def backward(ctx, grad_output):
        *feats, sampling_locations, scale_weights = ctx.saved_tensors
        n = len(feats)
        need = (any(ctx.needs_input_grad[2:2 + n]), any(ctx.needs_input_grad[2 + n:4 + n]))
        dfeats, dloc, dw = msmv_backward(grad_output.contiguous(), feats,
                                         sampling_locations, scale_weights, rows=ctx.rows,
                                         sorted_ws=ctx.sorted_ws, need=need)
        return (None, None, *dfeats, dloc, dw)


class FeatureGradSink:
    """One d-feat buffer per level shared by every decoder layer that samples ``mlvl_feats``.

    The reference lets each of the 6 layers allocate and zero a fresh 735 MB (per sample)
    gradient and autograd add them (msmv_sampling.cpp:244-247).  With a sink the first
    backward that runs writes the buffers, the following ones accumulate into them
    (OPUS_MSMV_BWD_ACCUMULATE), and the total is handed to autograd once, after the last one.
    Where the owner algorithm applies (``defer=None`` decides so; True / False force it) the d
    feats are deferred altogether: a layer's backward only sorts its points and computes its d loc
    / d weights, keeps its grad_output and sort, and ONE pass after the last layer accumulates
    all layers' points (opus_msmv_dfeat_multi) - d feats are written once, never read:

        sink = FeatureGradSink(mlvl_feats)          # channel-last [B',N,H,W,C] tensors
        for layer in layers:
            out = msmv_sampling_rows(sink, loc, w, T, G)    # or msmv_sampling(sink, loc, w)

    Ordering rests on autograd's own dependency count: every op takes the sink's scalar
    ``token`` as an extra input and returns a zero gradient for it, so the node that releases
    the buffers cannot run before all of them have.
    """

    def __init__(self, mlvl_feats, defer=None):
        self.feats = [f.detach() for f in mlvl_feats]
        # the detached aliases are invisible to autograd's saved-tensor version check: an in-place
        # update of a feature map between forward and backward is caught by comparing these
        self._versions = [f._version for f in mlvl_feats]
        if defer is None:
            B, N, _, _, C = self.feats[0].shape     # non-zero workspace <=> the owner algorithm applies
            defer = _bwd_algo != "atomic" and all(f.data_ptr() % 16 == 0 for f in self.feats) and \
                _lib.load().opus_msmv_backward_workspace_bytes(len(self.feats), B, C, N, 1, 1, 0) > 0
        self.defer = bool(defer)
        self.pending = []                       # deferred: (grad_output, workspace, Q, P, rows) per layer
        self.buffers = None                     # allocated by the first backward
        self.token = _SinkOpen.apply(self, *mlvl_feats)

    def __len__(self):
        return len(self.feats)

    def check_versions(self):
        for i, (f, v) in enumerate(zip(self.feats, self._versions)):
            if f._version != v:
                raise RuntimeError(
                    f"one of the variables needed for gradient computation has been modified by an inplace "
                    f"operation: feature level {i} of a FeatureGradSink is at version {f._version}; "
                    f"expected version {v} instead")


class _SinkOpen(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sink, *feats):
        # weak: sink.token's grad_fn is this node, a strong reference would close the cycle
        # sink -> token -> node -> sink and keep the sink's aliases of the feature maps (735 MB
        # per sample) alive until a cyclic collection.  The layers' nodes hold the sink strongly.
        ctx.sink_ref = weakref.ref(sink)
        ctx.n_feats = len(feats)
        return torch.zeros((), dtype=torch.float32, device=feats[0].device)

    @staticmethod
    def backward(ctx, _gtoken):
        sink = ctx.sink_ref()
        if sink is None:                        # no layer ever sampled through the sink
            return (None,) * (1 + ctx.n_feats)
        sink.check_versions()
        if sink.pending:
            # one pass per group of layers with equal sizes, at most max_sets layers per launch
            first = sink.buffers is None
            if first:
                sink.buffers = [torch.empty_like(f) for f in sink.feats]
            groups = {}
            for g, ws, Q, P, rows in sink.pending:
                groups.setdefault((Q, P, rows), []).append((g, ws))
            cap = _lib.load().opus_msmv_dfeat_max_sets()
            for (Q, P, rows), sets in groups.items():
                for i in range(0, len(sets), cap):
                    part = sets[i:i + cap]
                    msmv_dfeat_multi([g for g, _ in part], [ws for _, ws in part], sink.feats, Q, P,
                                     sink.buffers, rows=rows, accumulate=not first)
                    first = False
            sink.pending = []
        grads = sink.buffers if sink.buffers is not None else [None] * len(sink.feats)
        sink.buffers = None                     # a later backward (retain_graph) starts afresh
        return (None, *grads)


class _MSMVSamplingSink(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sink, rows, token, sampling_locations, scale_weights):
        ctx.sink, ctx.rows = sink, rows
        ctx.save_for_backward(sampling_locations, scale_weights)
        ctx.sorted_ws = None
        if _fwd_sorted and _bwd_algo != "atomic":
            res = msmv_forward_sorted(sink.feats, sampling_locations, scale_weights, rows=rows)
            if res is not None:
                out, ctx.sorted_ws = res
                return out
        return msmv_forward(sink.feats, sampling_locations, scale_weights, rows=rows)

    @staticmethod
    def backward(ctx, grad_output):
        sink = ctx.sink
        sampling_locations, scale_weights = ctx.saved_tensors
        sink.check_versions()
        if sink.defer:
            g = grad_output.contiguous()
            if g.data_ptr() % 16:               # (a contiguous view at an odd offset)
                g = g.clone()
            if ctx.sorted_ws is not None:       # the forward's sort: only the gather is left to do here
                flags = _lib.BWD_OWNER | _lib.BWD_PRESORTED | _lib.BWD_STAGE_GATHER
                ws = ctx.sorted_ws
            else:
                flags = _lib.BWD_OWNER | _lib.BWD_STAGE_BIN | _lib.BWD_STAGE_GATHER
                need = backward_workspace_bytes(sink.feats, sampling_locations, scale_weights, flags)
                ws = torch.empty(max(need, 16), dtype=torch.uint8, device=g.device)
            dloc, dw = torch.empty_like(sampling_locations), torch.empty_like(scale_weights)
            msmv_backward_into(g, sink.feats, sampling_locations, scale_weights, None, dloc, dw,
                               workspace=ws, flags=flags, rows=ctx.rows)
            _, Q, P, _ = sampling_locations.shape
            sink.pending.append((g, ws, Q, P, ctx.rows))
            return None, None, grad_output.new_zeros(()), dloc, dw
        first = sink.buffers is None
        if first:
            sink.buffers = [torch.empty_like(f) for f in sink.feats]
        _, dloc, dw = msmv_backward(grad_output.contiguous(), sink.feats, sampling_locations, scale_weights,
                                    dfeats=None if first else sink.buffers, rows=ctx.rows,
                                    out_dfeats=sink.buffers if first else None, sorted_ws=ctx.sorted_ws)
        return None, None, grad_output.new_zeros(()), dloc, dw


def _sink_or_none(mlvl_feats):
    return mlvl_feats if isinstance(mlvl_feats, FeatureGradSink) else None


def msmv_sampling_rows(mlvl_feats, sampling_locations, scale_weights, num_frames, num_groups):
    """msmv_sampling + sampling_4d's un-pack: [Bs*T*G, ...] inputs -> [Bs, Q, G, T*P, C].
    ``mlvl_feats`` may be a FeatureGradSink."""
    sink = _sink_or_none(mlvl_feats)
    if sink is not None:
        return _MSMVSamplingSink.apply(sink, (num_frames, num_groups), sink.token,
                                       sampling_locations, scale_weights)
    return MSMVSamplingRows.apply(num_frames, num_groups, *mlvl_feats, sampling_locations, scale_weights)


def msmv_sampling(mlvl_feats, sampling_locations, scale_weights):
    """Channel-last features [B,N,H,W,C] per level -> [B,Q,C,P] (reference wrapper.py:87-93).
    ``mlvl_feats`` may be a FeatureGradSink (no counterpart in the reference).

    Without the library: raises, unless the fallback was opted into (``allow_fallback`` /
    OPUS_MSMV_ALLOW_FALLBACK=1) - then, as the reference's last arm (wrapper.py:92-93), the
    features are taken channel-FIRST [B,C,N,H,W] (what the decoder prepares when MSMV_CUDA is
    false, models/opus_transformer.py:122-124) and ``msmv_sampling_pytorch`` computes the result."""
    if not MSMV_CUDA and _allow_fallback:
        return msmv_sampling_pytorch(mlvl_feats, sampling_locations, scale_weights)
    sink = _sink_or_none(mlvl_feats)
    if sink is not None:
        return _MSMVSamplingSink.apply(sink, None, sink.token, sampling_locations, scale_weights)
    if len(mlvl_feats) == 4:
        return MSMVSamplingC2345.apply(*mlvl_feats, sampling_locations, scale_weights)
    if len(mlvl_feats) == 5:
        return MSMVSamplingC23456.apply(*mlvl_feats, sampling_locations, scale_weights)
    return _MSMVSampling.apply(*mlvl_feats, sampling_locations, scale_weights)
