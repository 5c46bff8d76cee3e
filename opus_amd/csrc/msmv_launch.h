// Host-side launchers shared between the kernel translation units and the C ABI.
#pragma once
#include "msmv_common.h"
#include <stddef.h>

// Launch errors are read from the thread's "last error" slot, which other users of the HIP
// runtime in the process (e.g. torch) share.  A launcher notes the slot's state before its first
// launch (`stale = hipPeekAtLastError()`) and ends with msmv_launch_status(stale): an error its own
// launches raised is returned and taken out of the slot; one that was already there is neither
// reported as ours nor consumed.
inline hipError_t msmv_launch_status(hipError_t stale)
{
    const hipError_t now = hipPeekAtLastError();
    if (now == hipSuccess || now == stale)
        return hipSuccess;
    (void)hipGetLastError();
    return now;
}

// force_generic != 0 selects the any-shape kernel even when the fast path applies (testing).
hipError_t msmv_launch_forward(const MsmvArgs &a, hipStream_t s, int force_generic);

// Atomic-scatter backward: clears d feats unless `accumulate`, then one wave per point.
hipError_t msmv_launch_backward_atomic(const MsmvArgs &a, hipStream_t s, bool accumulate);

// Owner-computes backward (C == 64 only).
bool msmv_backward_owner_supported(const MsmvArgs &a);
size_t msmv_backward_owner_workspace(int B, int N, int Q, int P);
// stages: bit 0 = bin, bit 1 = d feat tiles, bit 2 = d loc / d w gather
hipError_t msmv_launch_backward_owner(const MsmvArgs &a, void *workspace, hipStream_t s, int stages, bool accumulate);

// Forward in bin order (owner-supported shapes; row mode, or P == 1): sorts into `workspace` unless
// `presorted`.
hipError_t msmv_launch_forward_sorted(const MsmvArgs &a, void *workspace, hipStream_t s, bool presorted);

// Caller-side fusion (msmv_project.hip): projection + view selection + packing, and its backward.
hipError_t msmv_launch_project_pack(const float *pts, const float *sw, const float *occ2img,
                                    int B, int Q, int T, int G, int P, int L, int N,
                                    float image_h, float image_w, float eps, int shared,
                                    float *loc, float *w, hipStream_t s);
hipError_t msmv_launch_project_pack_backward(const float *dloc, const float *dw, const float *pts,
                                             const float *occ2img, const float *loc,
                                             int B, int Q, int T, int G, int P, int L, int N,
                                             float image_h, float image_w, float eps, int shared,
                                             float *dpts, float *dsw, hipStream_t s);

// Producer-side re-layout (msmv_relayout.hip): [B,T*N,G*C,H,W] <-> [B*T*G,N,H,W,C].
hipError_t msmv_launch_relayout(const float *src, float *dst, int B, int T, int N, int G, int C, int H, int W,
                                bool inverse, hipStream_t s);

// One d-feat pass over several point sets sorted by earlier OPUS_MSMV_BWD_STAGE_BIN calls with
// these sizes (workspaces[i]) and their gradients of the op's output (gouts[i]).
int msmv_dfeat_max_sets();
hipError_t msmv_launch_dfeat_sets(const MsmvArgs &a, int nsets, const float *const *gouts,
                                  void *const *workspaces, hipStream_t s, bool accumulate);
