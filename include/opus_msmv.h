/*
 * opus_msmv.h - C ABI of libopus_msmv.so: multi-scale multi-view feature sampling
 * (forward + backward) for MI355X / gfx950.
 *
 * This is the drop-in boundary for the native half of the reference's sampling op.  The
 * reference binds four pybind11 functions (models/csrc/msmv_sampling/msmv_sampling.cpp:362-368)
 * over raw-pointer launchers (msmv_sampling.cpp:5-96); the entry points below take the
 * same raw pointers and sizes, with the level count as a runtime argument instead of a
 * _c2345 / _c23456 suffix, an explicit stream, and an error code instead of a printf.
 *
 * Conventions
 *   - All tensors are fp32, contiguous, resident on the device that is current when the
 *     call is made.  Nothing is allocated, freed or synchronised inside the library; the
 *     launches are asynchronous on `stream` (a hipStream_t passed as void*, NULL = the
 *     default stream).  The library keeps no state between calls and is thread-safe; the one
 *     exception is opus_msmv_backward_noalloc(), whose scratch is a buffer of the library's own.
 *   - feats[l]  : [B, N, H_l, W_l, C]  channel-last   (msmv_sampling.cpp:99-102)
 *     loc       : [B, Q, P, 3]  (x/width, y/height, view/(N-1)), each nominally in [0,1]
 *     weights   : [B, Q, P, L]
 *     out       : [B, Q, C, P]                       (msmv_sampling.cpp:136)
 *   - `feats` / `dfeats` are HOST arrays of L device pointers; `hw` is a HOST array of
 *     2*L ints {H_0, W_0, H_1, W_1, ...}.
 *   - Return value: OPUS_MSMV_OK (0) or a negative OPUS_MSMV_E* code; text via
 *     opus_msmv_strerror().
 */
#ifndef OPUS_MSMV_H
#define OPUS_MSMV_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPUS_MSMV_VERSION 200          /* major*10000 + minor*100 + patch */
#define OPUS_MSMV_MAX_LEVELS 8
#define OPUS_MSMV_MAX_POINTS 32        /* MAX_POINT, msmv_sampling_forward.cu:21, msmv_sampling.cpp:125 */

enum {
    OPUS_MSMV_OK = 0,
    OPUS_MSMV_EINVAL_LEVELS = -1,   /* L outside [1, OPUS_MSMV_MAX_LEVELS] */
    OPUS_MSMV_EINVAL_POINTS = -2,   /* P outside [1, 32]: "num_point exceed limits" (msmv_sampling.cpp:125) */
    OPUS_MSMV_EINVAL_DIMS = -3,     /* a negative size, or zero C / N / H / W */
    OPUS_MSMV_EINVAL_PTR = -4,      /* a NULL pointer where data is required */
    OPUS_MSMV_ETOOLARGE = -5,       /* B*Q*P*C or a feature slab count exceeds the 31-bit point index */
    OPUS_MSMV_EWORKSPACE = -6,      /* workspace missing or smaller than opus_msmv_backward_workspace_bytes() */
    OPUS_MSMV_EINVAL_FLAGS = -7,    /* unknown algorithm / flag combination */
    OPUS_MSMV_ELAUNCH = -8,         /* a HIP call or kernel launch of this call failed */
    OPUS_MSMV_ESCRATCH = -9         /* opus_msmv_backward_noalloc: the sizes need more than OPUS_MSMV_NOALLOC_SCRATCH_BYTES */
};

/* backward algorithm selector (low 4 bits of `flags`) */
enum {
    OPUS_MSMV_BWD_AUTO = 0,      /* owner-computes where it applies (C == 64), else atomic scatter */
    OPUS_MSMV_BWD_ATOMIC = 1,    /* wave-reduced dloc/dw + fp32 atomics into d feats (zero-filled first) */
    OPUS_MSMV_BWD_OWNER = 2      /* every d-feat tile is accumulated in LDS by one workgroup: no atomics, no zero-fill */
};
/* OR-ed into `flags`: add onto the current contents of d feats instead of overwriting them
 * (either algorithm; d loc and d weights are still overwritten).  For a stack of decoder
 * layers that sample the same feature maps: the first layer's backward writes d feats, the
 * others accumulate, and no zero-fill or separate add ever touches the 735 MB per sample.
 * The reference always clears and lets autograd add (msmv_sampling.cpp:244-247). */
#define OPUS_MSMV_BWD_ACCUMULATE 0x10
/* OR-ed into `flags` (owner algorithm only): run just the named stage(s) of the backward,
 * so that a profiler or benchmark can time each kernel with events of its own.  STAGE_BIN
 * must come first, on the same stream and with the same arguments and workspace as the other
 * two, which are independent of each other (a call with both runs the gather first: it leaves
 * the gradient rows in the Infinity Cache for the d-feat pass); no stage bit set means all three.  `dfeats` may be NULL when STAGE_DFEAT is not among the
 * stages asked for (see opus_msmv_dfeat_multi); `dloc` / `dweights` when STAGE_GATHER is not
 * (the sort reads `loc` and `weights` only).  No counterpart in the reference. */
#define OPUS_MSMV_BWD_STAGE_BIN 0x100      /* sort the points into (batch, view, band) bins */
#define OPUS_MSMV_BWD_STAGE_DFEAT 0x200    /* d feats, one owner per run of pixels */
#define OPUS_MSMV_BWD_STAGE_GATHER 0x400   /* d loc and d weights */
#define OPUS_MSMV_BWD_STAGE_MASK 0x700
/* OR-ed into `flags` (owner algorithm only): `workspace` already holds the sort of these very
 * `loc` / `weights`, left there by opus_msmv_forward_sorted() or by a backward's sort stage with
 * the same sizes (and, in row mode, the same T and G): the sort stage is skipped.  Not to be combined with
 * OPUS_MSMV_BWD_STAGE_BIN.  No counterpart in the reference. */
#define OPUS_MSMV_BWD_PRESORTED 0x800

/* Library version (OPUS_MSMV_VERSION it was built with). */
int opus_msmv_version(void);

/* Text for a return code; never NULL. */
const char *opus_msmv_strerror(int code);

/*
 * Forward: out[b,q,c,p] = sum_l weights[b,q,p,l] * bilinear(feats[l][b, view(b,q,p)], x, y)[c]
 * with align_corners=True pixel mapping, zero padding per corner, view = round(loc_v*(N-1)).
 * Replaces ms_deform_attn_cuda_c2345_forward / _c23456_forward
 * (msmv_sampling.cpp:98-150, :152-210; kernels msmv_sampling_forward.cu:75-164, :166-267).
 * Every element of `out` is written.  A view index outside [0, N-1] (undefined behaviour
 * in the reference) contributes zero.
 */
int opus_msmv_forward(int L, const float *const *feats, const int *hw,
                      const float *loc, const float *weights,
                      int B, int C, int N, int Q, int P,
                      float *out, void *stream);

/*
 * Same as opus_msmv_forward with a flags word.  OPUS_MSMV_FWD_GENERIC selects the
 * any-shape kernel even where the specialised one (C == 64, L in {4,5}, 16-byte aligned
 * pointers) applies; it exists so that tests can cover both kernels on one input.  No
 * counterpart in the reference.
 */
#define OPUS_MSMV_FWD_GENERIC 0x1
int opus_msmv_forward_ex(int L, const float *const *feats, const int *hw,
                         const float *loc, const float *weights,
                         int B, int C, int N, int Q, int P,
                         float *out, int flags, void *stream);

/*
 * Bytes of device scratch the backward call needs for these sizes and `flags`
 * (0 for the atomic algorithm).  Has no counterpart in the reference, whose backward
 * needs no scratch.
 */
size_t opus_msmv_backward_workspace_bytes(int L, int B, int C, int N, int Q, int P, int flags);

/*
 * Backward of the above for a given grad_out [B,Q,C,P] (contiguous).  Writes
 *   dfeats[l] [B,N,H_l,W_l,C]  - completely (cleared + accumulated; see ACCUMULATE),
 *   dloc      [B,Q,P,3]        - x and y gradients; the view slot is written 0,
 *   dweights  [B,Q,P,L].
 * Replaces ms_deform_attn_cuda_c2345_backward / _c23456_backward (msmv_sampling.cpp:212-281,
 * :283-360; kernels msmv_sampling_backward.cu:108-224, :226-361), including the
 * zeros_like() clears the reference's host shim performs (msmv_sampling.cpp:244-249).
 * `workspace` must point to at least opus_msmv_backward_workspace_bytes(...) bytes of
 * device memory (may be NULL when that is 0); its contents on entry are ignored.
 * Non-finite gradients: the reference adds w * g to the four corner pixels of a point only
 * (msmv_sampling_backward.cu:60-99), so an inf / NaN in grad_out reaches those pixels.  The atomic
 * algorithm does the same.  The owner algorithm accumulates a 4 x 8-pixel patch as one matrix
 * product over the patch's points, zero weights included: a non-finite grad_out value of a point
 * turns that channel of EVERY pixel of the patches the point's corners fall into (at every level)
 * into inf / NaN, not just of the corner pixels, and may do the same to patches next to them (a
 * patch reads the gradient rows of points in the image bins it scans, with weight zero where they
 * miss it).  Other channels, other views and batch slabs, and d loc / d weights of other points are
 * unaffected.
 */
int opus_msmv_backward(int L, const float *grad_out, const float *const *feats, const int *hw,
                       const float *loc, const float *weights,
                       int B, int C, int N, int Q, int P,
                       float *const *dfeats, float *dloc, float *dweights,
                       void *workspace, size_t workspace_bytes, int flags, void *stream);

/*
 * opus_msmv_backward for callers that cannot allocate device memory and have no workspace to
 * hand in (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
 * It has no counterpart in the reference.  Same arguments, results and flags as
 * opus_msmv_backward without `workspace` / `workspace_bytes`: the sort's scratch (histogram
 * table, offsets and the sorted point entries, opus_msmv_backward_workspace_bytes() bytes) lives in a
 * static device array of OPUS_MSMV_NOALLOC_SCRATCH_BYTES inside the library, one copy per GPU,
 * held until the process ends.  Sizes that need more (32 MiB covers B*Q*P <= about 0.85 M points
 * at N = 6, e.g. two OPUS-L samples): with the owner algorithm asked for by name the call returns
 * OPUS_MSMV_ESCRATCH and launches nothing; with OPUS_MSMV_BWD_AUTO it runs the atomic algorithm,
 * which needs no scratch (what AUTO falls back to wherever the owner algorithm does not apply).
 * The call neither allocates nor frees nor synchronises.  The first call on a device looks the
 * array's address up (hipGetSymbolAddress, which may load the library's code object): make it
 * outside any graph capture.  All calls on one GPU share that GPU's array: calls that overlap
 * in time on one GPU (two streams, two threads) race on it; calls on one stream do not.  With
 * OPUS_MSMV_BWD_STAGE_* flags the stages of one backward must run before another backward's.
 */
#define OPUS_MSMV_NOALLOC_SCRATCH_BYTES (32u << 20)
int opus_msmv_backward_noalloc(int L, const float *grad_out, const float *const *feats, const int *hw,
                               const float *loc, const float *weights,
                               int B, int C, int N, int Q, int P,
                               float *const *dfeats, float *dloc, float *dweights,
                               int flags, void *stream);

/*
 * Row-mode variants: the op fused with the permute its caller applies to the result
 * (models/opus_sampling.py:119-122).  The op's batch is B' = Bs*T*G with slab
 * b' = (bs*T + t)*G + g, as sampling_4d packs it (opus_sampling.py:104-107); out / grad_out are
 * [Bs, Q, G, T*P, C] - what sampling_4d returns - instead of [B', Q, C, P], so the
 * 78.6 MB (OPUS-L) permute-copy after the op, and its mirror image in the backward, disappear
 * and every point's C channels are one contiguous row.  Everything else (feats, loc, weights,
 * gradients, workspace size for B = Bs*T*G, flags, errors) is as in opus_msmv_forward_ex /
 * opus_msmv_backward.
 */
int opus_msmv_forward_rows(int L, const float *const *feats, const int *hw,
                           const float *loc, const float *weights,
                           int Bs, int T, int G, int C, int N, int Q, int P,
                           float *out, int flags, void *stream);

int opus_msmv_backward_rows(int L, const float *grad_out, const float *const *feats, const int *hw,
                            const float *loc, const float *weights,
                            int Bs, int T, int G, int C, int N, int Q, int P,
                            float *const *dfeats, float *dloc, float *dweights,
                            void *workspace, size_t workspace_bytes, int flags, void *stream);

/*
 * Forward in bin order.  No counterpart in the reference.  The points of every batch slab are
 * first sorted by (view, image band) - the backward's sort, written into `workspace`
 * (>= opus_msmv_backward_workspace_bytes(L, B, C, N, Q, P, OPUS_MSMV_BWD_OWNER) bytes) - and the
 * forward then walks them in that order, so that neighbouring points fetch neighbouring pixels
 * and the feature traffic approaches the distinct pixels touched.  Results are bit-identical to
 * opus_msmv_forward_rows / opus_msmv_forward.  In a training step the sort serves three passes:
 * the forward, the forward re-run under activation checkpointing (OPUS_MSMV_FWD_PRESORTED) and
 * the backward (OPUS_MSMV_BWD_PRESORTED), as long as `loc`, `weights` and the workspace are kept.
 *   T >= 1: row mode, B = Bs*T*G, out [Bs, Q, G, T*P, C] (see opus_msmv_forward_rows).
 *   T == 0: out [B = Bs, Q, C, P]; P must be 1 (for larger P a point's channels are P-strided in
 *           that layout and bin order would scatter 4-byte pieces).
 * flags: 0 or OPUS_MSMV_FWD_PRESORTED (the workspace already holds the sort of these loc /
 * weights).  OPUS_MSMV_EINVAL_FLAGS where the sort does not apply (C != 64, L outside {4,5},
 * N > 16, pointers not 16-byte aligned, T == 0 with P > 1): call opus_msmv_forward_rows then.
 */
#define OPUS_MSMV_FWD_PRESORTED 0x4
int opus_msmv_forward_sorted(int L, const float *const *feats, const int *hw,
                             const float *loc, const float *weights,
                             int Bs, int T, int G, int C, int N, int Q, int P,
                             float *out, void *workspace, size_t workspace_bytes, int flags, void *stream);

/*
 * Deferred d feats for a stack of layers that sample the same feature maps (the decoder calls
 * the op once per layer, models/opus_transformer.py:57-75, and autograd sums the layers'
 * 735 MB d feats; nothing in a layer's own backward reads them).  No counterpart in the
 * reference.  Each layer's backward runs only OPUS_MSMV_BWD_OWNER | OPUS_MSMV_BWD_STAGE_BIN |
 * OPUS_MSMV_BWD_STAGE_GATHER (its d loc / d weights; `dfeats` may then be NULL) into a
 * workspace of its own and keeps that workspace and its grad_out alive; after the last layer
 * ONE call of opus_msmv_dfeat_multi() accumulates all of them, so that d feats are written
 * once instead of once per layer (and never read back):
 *   nsets in [1, opus_msmv_dfeat_max_sets()] sets, all of the same sizes; grad_outs[i] is
 *   [B,Q,C,P] (T == 0, B = Bs) or [Bs,Q,G,T*P,C] (T >= 1, row mode); workspaces[i] (each of
 *   workspace_bytes >= opus_msmv_backward_workspace_bytes()) holds set i's sort.
 *   flags: 0 (d feats are overwritten, every element) or OPUS_MSMV_BWD_ACCUMULATE (for a
 *   further batch of sets).  EINVAL_FLAGS where the owner algorithm does not apply (C != 64,
 *   L outside {4,5}, pointers not 16-byte aligned).
 */
int opus_msmv_dfeat_max_sets(void);
int opus_msmv_dfeat_multi(int L, int nsets, const float *const *grad_outs, void *const *workspaces,
                          size_t workspace_bytes, const int *hw,
                          int Bs, int T, int G, int C, int N, int Q, int P,
                          float *const *dfeats, int flags, void *stream);

/*
 * Caller-side fusion: everything sampling_4d does before it calls the op
 * (models/opus_sampling.py:36-113), in one launch.
 *   sample_points [B,Q,T,G,P,3] (metres), scale_weights [B,Q,G,T,P,L], occ2img [B,T*N,4,4]
 *   -> loc [B*T*G,Q,P,3] = (u, v, view/(N-1)) of the first camera (of the point's frame) in
 *      which the point is visible (z > eps, 0 < u,v < 1), camera 0 if none (opus_sampling.py:94);
 *      weights [B*G*T,Q,P,L] = the scale weights re-packed.  Points are packed in (B,T,G)
 *      order and weights in (B,G,T) order, as the reference does (opus_sampling.py:104-112).
 * Requires N >= 2 (the reference divides by N-1, opus_sampling.py:101).
 * flags: OPUS_MSMV_PACK_SHARED_FRAMES - the T frames share one set of points and weights, as the
 * decoder produces them before it expands them over T (models/opus_transformer.py:350-358):
 * sample_points is then [B,Q,G,P,3] and scale_weights [B,Q,G,P,L], outputs unchanged.
 */
#define OPUS_MSMV_PACK_SHARED_FRAMES 0x1
int opus_msmv_project_pack(const float *sample_points, const float *scale_weights, const float *occ2img,
                           int B, int Q, int T, int G, int P, int L, int N,
                           float image_h, float image_w, float eps, int flags,
                           float *loc, float *weights, void *stream);

/*
 * Backward of opus_msmv_project_pack: d sample_points [B,Q,T,G,P,3] from d loc (the view
 * slot carries no gradient, nor do the visibility mask and the camera choice) through the
 * chosen camera's projection, and d scale_weights [B,Q,G,T,P,L] = d weights un-packed.
 * `loc` is the forward's output (it holds the chosen camera).  occ2img gets no gradient:
 * it is built from image metadata in the reference (models/opus_transformer.py:105-110).
 * With OPUS_MSMV_PACK_SHARED_FRAMES the gradients have the shared shapes [B,Q,G,P,3] /
 * [B,Q,G,P,L] and hold the sum over the T frames.
 */
int opus_msmv_project_pack_backward(const float *dloc, const float *dweights,
                                    const float *sample_points, const float *occ2img, const float *loc,
                                    int B, int Q, int T, int G, int P, int L, int N,
                                    float image_h, float image_w, float eps, int flags,
                                    float *dsample_points, float *dscale_weights, void *stream);

/*
 * Producer-side re-layout of one feature level: what the decoder does before its first layer,
 *   feat.reshape(B,T,N,G,C,H,W).permute(0,1,3,2,5,6,4).reshape(B*T*G,N,H,W,C).contiguous()
 * (models/opus_transformer.py:113-126, MSMV_CUDA branch), as one tiled transpose.
 *   inverse == 0: src [B, T*N, G*C, H, W] (backbone / FPN layout) -> dst [B*T*G, N, H, W, C]
 *   inverse != 0: src [B*T*G, N, H, W, C] (e.g. d feats of the op)  -> dst [B, T*N, G*C, H, W]
 * src and dst must not overlap.
 */
int opus_msmv_relayout(const float *src, float *dst, int B, int T, int N, int G, int C, int H, int W,
                       int inverse, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OPUS_MSMV_H */
