// C ABI of libopus_msmv.so (see include/opus_msmv.h): argument validation, the kernel
// argument block, and dispatch to the launchers.  No allocation, no synchronisation; the only
// state is the per-device address of the static scratch of opus_msmv_backward_noalloc.
#include "../../include/opus_msmv.h"
#include "msmv_launch.h"

#include <atomic>
#include <limits.h>
#include <stdint.h>

// Scratch of opus_msmv_backward_noalloc: the owner backward's sort workspace for callers that
// cannot allocate.  Every GPU the library's code object is loaded on holds its own copy at its
// own address.  In host code the name below is only a handle for hipGetSymbolAddress - it is
// NOT a device address and must never be passed to a kernel or a HIP memory call.  (No `used`
// attribute: it would pin the device-side symbol to local linkage, where the runtime's lookup
// by name cannot find it; the reference from host code below is what keeps the array emitted.)
static __device__ __align__(256) unsigned char scratch[OPUS_MSMV_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int MAX_DEVICES = 64;
std::atomic<void *> scratch_ptr[MAX_DEVICES];      // device address of `scratch`, per device ordinal

// Device address of the current device's copy of `scratch` (looked up once per device; threads
// that race here compute the same value).
int scratch_of_current_device(void **out)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES)
        return OPUS_MSMV_ELAUNCH;
    void *p = scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p)
            return OPUS_MSMV_ELAUNCH;
        scratch_ptr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return OPUS_MSMV_OK;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Row mode (out / grad_out in [Bs, Q, G, T*P, C]): B must be Bs*T*G.
int fill_rows(MsmvArgs &a, int T, int G)
{
    if (T < 1 || G < 1 || a.B % (T * G) != 0)
        return OPUS_MSMV_EINVAL_DIMS;
    a.rows = 1;
    a.T = T;
    a.G = G;
    a.div_p = msmv_make_div(a.P);
    a.div_q = msmv_make_div(a.Q);
    a.div_t = msmv_make_div(T);
    a.div_g = msmv_make_div(G);
    a.div_tg = msmv_make_div(T * G);
    return OPUS_MSMV_OK;
}

// Validates sizes and fills everything in MsmvArgs except the data pointers.
int fill_common(MsmvArgs &a, int L, const int *hw, int B, int C, int N, int Q, int P)
{
    if (L < 1 || L > OPUS_MSMV_MAX_LEVELS)
        return OPUS_MSMV_EINVAL_LEVELS;
    if (P < 1 || P > OPUS_MSMV_MAX_POINTS)
        return OPUS_MSMV_EINVAL_POINTS;
    if (B < 0 || Q < 0 || C < 1 || N < 1)
        return OPUS_MSMV_EINVAL_DIMS;
    if (!hw)
        return OPUS_MSMV_EINVAL_PTR;
    for (int l = 0; l < L; ++l) {
        if (hw[2 * l] < 1 || hw[2 * l + 1] < 1)
            return OPUS_MSMV_EINVAL_DIMS;
        if ((long long)hw[2 * l] * hw[2 * l + 1] > INT_MAX / 64)
            return OPUS_MSMV_ETOOLARGE;
        a.lvl[l].H = hw[2 * l];
        a.lvl[l].W = hw[2 * l + 1];
    }
    // point indices and grad_out element offsets are 31-bit inside the kernels
    const long long npts = (long long)B * Q * P;
    if (npts * (long long)(C > L ? C : L) > INT_MAX || (long long)B * N > INT_MAX || (long long)Q * P > INT_MAX)
        return OPUS_MSMV_ETOOLARGE;
    a.L = L; a.B = B; a.C = C; a.N = N; a.Q = Q; a.P = P;
    a.npts = (int)npts;
    a.qp = Q * P;
    const MsmvDiv dqp = msmv_make_div(a.qp);
    a.qp_magic = dqp.magic;
    a.qp_shift = dqp.shift;
    return OPUS_MSMV_OK;
}

int owner_applicable(const MsmvArgs &a, const float *gout)
{
    if (!msmv_backward_owner_supported(a) || !aligned16(gout))
        return 0;
    for (int l = 0; l < a.L; ++l)
        if (!aligned16(a.lvl[l].feat) || !aligned16(a.lvl[l].dfeat))
            return 0;
    return 1;
}

}  // namespace

extern "C" {

int opus_msmv_version(void) { return OPUS_MSMV_VERSION; }

const char *opus_msmv_strerror(int code)
{
    switch (code) {
    case OPUS_MSMV_OK: return "ok";
    case OPUS_MSMV_EINVAL_LEVELS: return "number of feature levels must be in [1, 8]";
    case OPUS_MSMV_EINVAL_POINTS: return "num_point exceed limits (must be in [1, 32])";
    case OPUS_MSMV_EINVAL_DIMS: return "invalid tensor dimensions";
    case OPUS_MSMV_EINVAL_PTR: return "null pointer argument";
    case OPUS_MSMV_ETOOLARGE: return "problem too large for 31-bit point / element indices";
    case OPUS_MSMV_EWORKSPACE: return "workspace missing or too small (see opus_msmv_backward_workspace_bytes)";
    case OPUS_MSMV_EINVAL_FLAGS: return "unknown algorithm or flag combination";
    case OPUS_MSMV_ELAUNCH: return "HIP call or kernel launch failed";
    case OPUS_MSMV_ESCRATCH: return "sizes need more scratch than the library's static buffer (OPUS_MSMV_NOALLOC_SCRATCH_BYTES)";
    default: return "unknown error code";
    }
}

// T == 0: out is [B, Q, C, P]; otherwise row mode with B = Bs*T*G
static int forward_impl(int L, const float *const *feats, const int *hw,
                        const float *loc, const float *weights,
                        int B, int T, int G, int C, int N, int Q, int P,
                        float *out, int flags, void *stream)
{
    MsmvArgs a = {};
    int rc = fill_common(a, L, hw, B, C, N, Q, P);
    if (rc == OPUS_MSMV_OK && T)
        rc = fill_rows(a, T, G);
    if (rc != OPUS_MSMV_OK)
        return rc;
    if (flags & ~OPUS_MSMV_FWD_GENERIC)
        return OPUS_MSMV_EINVAL_FLAGS;
    if (a.npts == 0)
        return OPUS_MSMV_OK;
    if (!feats || !loc || !weights || !out)
        return OPUS_MSMV_EINVAL_PTR;
    bool fast_ok = aligned16(out) && aligned16(weights);
    for (int l = 0; l < L; ++l) {
        if (!feats[l])
            return OPUS_MSMV_EINVAL_PTR;
        a.lvl[l].feat = feats[l];
        fast_ok = fast_ok && aligned16(feats[l]);
    }
    a.loc = loc;
    a.w = weights;
    a.out = out;
    const int generic = (flags & OPUS_MSMV_FWD_GENERIC) || !fast_ok;
    return msmv_launch_forward(a, (hipStream_t)stream, generic) == hipSuccess ? OPUS_MSMV_OK : OPUS_MSMV_ELAUNCH;
}

int opus_msmv_forward_ex(int L, const float *const *feats, const int *hw,
                         const float *loc, const float *weights,
                         int B, int C, int N, int Q, int P,
                         float *out, int flags, void *stream)
{
    return forward_impl(L, feats, hw, loc, weights, B, 0, 0, C, N, Q, P, out, flags, stream);
}

int opus_msmv_forward(int L, const float *const *feats, const int *hw,
                      const float *loc, const float *weights,
                      int B, int C, int N, int Q, int P,
                      float *out, void *stream)
{
    return forward_impl(L, feats, hw, loc, weights, B, 0, 0, C, N, Q, P, out, 0, stream);
}

int opus_msmv_forward_rows(int L, const float *const *feats, const int *hw,
                           const float *loc, const float *weights,
                           int Bs, int T, int G, int C, int N, int Q, int P,
                           float *out, int flags, void *stream)
{
    if (Bs < 0 || T < 1 || G < 1 || (long long)Bs * T * G > INT_MAX)
        return OPUS_MSMV_EINVAL_DIMS;
    return forward_impl(L, feats, hw, loc, weights, Bs * T * G, T, G, C, N, Q, P, out, flags, stream);
}

int opus_msmv_forward_sorted(int L, const float *const *feats, const int *hw,
                             const float *loc, const float *weights,
                             int Bs, int T, int G, int C, int N, int Q, int P,
                             float *out, void *workspace, size_t workspace_bytes, int flags, void *stream)
{
    if (Bs < 0 || T < 0 || (T > 0 && G < 1) || (T > 0 && (long long)Bs * T * G > INT_MAX))
        return OPUS_MSMV_EINVAL_DIMS;
    const int B = T ? Bs * T * G : Bs;
    MsmvArgs a = {};
    int rc = fill_common(a, L, hw, B, C, N, Q, P);
    if (rc == OPUS_MSMV_OK && T)
        rc = fill_rows(a, T, G);
    if (rc != OPUS_MSMV_OK)
        return rc;
    if (flags & ~OPUS_MSMV_FWD_PRESORTED)
        return OPUS_MSMV_EINVAL_FLAGS;
    if (!msmv_backward_owner_supported(a) || (!T && P != 1))
        return OPUS_MSMV_EINVAL_FLAGS;
    if (B == 0)
        return OPUS_MSMV_OK;
    if (!feats || (a.npts > 0 && (!loc || !weights || !out)))
        return OPUS_MSMV_EINVAL_PTR;
    if (!aligned16(out) || !aligned16(weights))
        return OPUS_MSMV_EINVAL_FLAGS;
    for (int l = 0; l < L; ++l) {
        if (!feats[l])
            return OPUS_MSMV_EINVAL_PTR;
        if (!aligned16(feats[l]))
            return OPUS_MSMV_EINVAL_FLAGS;
        a.lvl[l].feat = feats[l];
    }
    if (!workspace || workspace_bytes < msmv_backward_owner_workspace(B, N, Q, P) || !aligned16(workspace))
        return OPUS_MSMV_EWORKSPACE;
    a.loc = loc;
    a.w = weights;
    a.out = out;
    return msmv_launch_forward_sorted(a, workspace, (hipStream_t)stream, (flags & OPUS_MSMV_FWD_PRESORTED) != 0) == hipSuccess
               ? OPUS_MSMV_OK : OPUS_MSMV_ELAUNCH;
}

size_t opus_msmv_backward_workspace_bytes(int L, int B, int C, int N, int Q, int P, int flags)
{
    const int algo = flags & 0xF;
    if (algo == OPUS_MSMV_BWD_ATOMIC)
        return 0;
    if (L < 1 || B < 0 || N < 1 || Q < 0 || P < 1)
        return 0;
    MsmvArgs a = {};
    a.C = C; a.L = L; a.N = N;
    if (!msmv_backward_owner_supported(a))
        return 0;
    return msmv_backward_owner_workspace(B, N, Q, P);
}

static int backward_impl(int L, const float *grad_out, const float *const *feats, const int *hw,
                         const float *loc, const float *weights,
                         int B, int T, int G, int C, int N, int Q, int P,
                         float *const *dfeats, float *dloc, float *dweights,
                         void *workspace, size_t workspace_bytes, int flags, void *stream)
{
    MsmvArgs a = {};
    int rc = fill_common(a, L, hw, B, C, N, Q, P);
    if (rc == OPUS_MSMV_OK && T)
        rc = fill_rows(a, T, G);
    if (rc != OPUS_MSMV_OK)
        return rc;
    const int algo = flags & 0xF;
    const bool accumulate = (flags & OPUS_MSMV_BWD_ACCUMULATE) != 0;
    int stage_bits = flags & OPUS_MSMV_BWD_STAGE_MASK;
    const bool presorted = (flags & OPUS_MSMV_BWD_PRESORTED) != 0;
    if ((flags & ~(0xF | OPUS_MSMV_BWD_ACCUMULATE | OPUS_MSMV_BWD_STAGE_MASK | OPUS_MSMV_BWD_PRESORTED)) ||
        algo > OPUS_MSMV_BWD_OWNER)
        return OPUS_MSMV_EINVAL_FLAGS;
    if ((stage_bits || presorted) && algo != OPUS_MSMV_BWD_OWNER)
        return OPUS_MSMV_EINVAL_FLAGS;
    if (presorted) {
        if (stage_bits & OPUS_MSMV_BWD_STAGE_BIN)
            return OPUS_MSMV_EINVAL_FLAGS;
        if (!stage_bits)
            stage_bits = OPUS_MSMV_BWD_STAGE_DFEAT | OPUS_MSMV_BWD_STAGE_GATHER;
    }
    // d feats may be absent when the stages asked for leave them alone (sort and / or gather only)
    const bool need_dfeat = algo != OPUS_MSMV_BWD_OWNER || !stage_bits || (stage_bits & OPUS_MSMV_BWD_STAGE_DFEAT);
    if (!feats || (need_dfeat && !dfeats))
        return OPUS_MSMV_EINVAL_PTR;
    for (int l = 0; l < L; ++l) {
        if (B > 0 && (!feats[l] || (need_dfeat && !dfeats[l])))
            return OPUS_MSMV_EINVAL_PTR;
        a.lvl[l].feat = feats[l];
        a.lvl[l].dfeat = dfeats ? dfeats[l] : nullptr;
    }
    // d loc / d weights are written by the gather stage only
    const bool need_dlw = algo != OPUS_MSMV_BWD_OWNER || !stage_bits || (stage_bits & OPUS_MSMV_BWD_STAGE_GATHER);
    if (a.npts > 0 && (!grad_out || !loc || !weights || (need_dlw && (!dloc || !dweights))))
        return OPUS_MSMV_EINVAL_PTR;
    a.loc = loc;
    a.w = weights;
    a.gout = grad_out;
    a.dloc = dloc;
    a.dw = dweights;
    if (B == 0)
        return OPUS_MSMV_OK;

    bool use_owner;
    if (algo == OPUS_MSMV_BWD_OWNER) {
        if (!owner_applicable(a, grad_out))
            return OPUS_MSMV_EINVAL_FLAGS;
        use_owner = true;
    } else if (algo == OPUS_MSMV_BWD_ATOMIC) {
        use_owner = false;
    } else {
        use_owner = owner_applicable(a, grad_out) != 0;
    }

    hipError_t e;
    if (use_owner) {
        const size_t need = msmv_backward_owner_workspace(B, N, Q, P);
        if (!workspace || workspace_bytes < need || !aligned16(workspace))
            return OPUS_MSMV_EWORKSPACE;
        e = msmv_launch_backward_owner(a, workspace, (hipStream_t)stream, stage_bits ? stage_bits >> 8 : 7, accumulate);
    } else {
        e = msmv_launch_backward_atomic(a, (hipStream_t)stream, accumulate);
    }
    return e == hipSuccess ? OPUS_MSMV_OK : OPUS_MSMV_ELAUNCH;
}

int opus_msmv_backward(int L, const float *grad_out, const float *const *feats, const int *hw,
                       const float *loc, const float *weights,
                       int B, int C, int N, int Q, int P,
                       float *const *dfeats, float *dloc, float *dweights,
                       void *workspace, size_t workspace_bytes, int flags, void *stream)
{
    return backward_impl(L, grad_out, feats, hw, loc, weights, B, 0, 0, C, N, Q, P, dfeats, dloc, dweights,
                         workspace, workspace_bytes, flags, stream);
}

int opus_msmv_backward_noalloc(int L, const float *grad_out, const float *const *feats, const int *hw,
                               const float *loc, const float *weights,
                               int B, int C, int N, int Q, int P,
                               float *const *dfeats, float *dloc, float *dweights,
                               int flags, void *stream)
{
    // 0 where no scratch is needed (atomic algorithm, shapes the owner algorithm does not take)
    const size_t need = opus_msmv_backward_workspace_bytes(L, B, C, N, Q, P, flags);
    if (need > OPUS_MSMV_NOALLOC_SCRATCH_BYTES) {
        // the caller left the choice to the library: the atomic algorithm needs no scratch (it is what
        // OPUS_MSMV_BWD_AUTO falls back to wherever the owner algorithm does not apply)
        if ((flags & 0xF) == OPUS_MSMV_BWD_AUTO && !(flags & (OPUS_MSMV_BWD_STAGE_MASK | OPUS_MSMV_BWD_PRESORTED)))
            return backward_impl(L, grad_out, feats, hw, loc, weights, B, 0, 0, C, N, Q, P, dfeats, dloc, dweights,
                                 nullptr, 0, (flags & ~0xF) | OPUS_MSMV_BWD_ATOMIC, stream);
        return OPUS_MSMV_ESCRATCH;
    }
    void *ptr = nullptr;
    const int rc = scratch_of_current_device(&ptr);
    if (rc != OPUS_MSMV_OK)
        return rc;
    // nothing in the array needs clearing: the sort writes every word (histogram table, offsets,
    // entries) that a later stage reads, so whatever an earlier call left there is never read
    return backward_impl(L, grad_out, feats, hw, loc, weights, B, 0, 0, C, N, Q, P, dfeats, dloc, dweights,
                         ptr, OPUS_MSMV_NOALLOC_SCRATCH_BYTES, flags, stream);
}

int opus_msmv_backward_rows(int L, const float *grad_out, const float *const *feats, const int *hw,
                            const float *loc, const float *weights,
                            int Bs, int T, int G, int C, int N, int Q, int P,
                            float *const *dfeats, float *dloc, float *dweights,
                            void *workspace, size_t workspace_bytes, int flags, void *stream)
{
    if (Bs < 0 || T < 1 || G < 1 || (long long)Bs * T * G > INT_MAX)
        return OPUS_MSMV_EINVAL_DIMS;
    return backward_impl(L, grad_out, feats, hw, loc, weights, Bs * T * G, T, G, C, N, Q, P, dfeats, dloc, dweights,
                         workspace, workspace_bytes, flags, stream);
}

int opus_msmv_dfeat_max_sets(void) { return msmv_dfeat_max_sets(); }

int opus_msmv_dfeat_multi(int L, int nsets, const float *const *grad_outs, void *const *workspaces,
                          size_t workspace_bytes, const int *hw,
                          int Bs, int T, int G, int C, int N, int Q, int P,
                          float *const *dfeats, int flags, void *stream)
{
    int B = Bs;
    if (T) {
        if (Bs < 0 || T < 1 || G < 1 || (long long)Bs * T * G > INT_MAX)
            return OPUS_MSMV_EINVAL_DIMS;
        B = Bs * T * G;
    }
    MsmvArgs a = {};
    int rc = fill_common(a, L, hw, B, C, N, Q, P);
    if (rc == OPUS_MSMV_OK && T)
        rc = fill_rows(a, T, G);
    if (rc != OPUS_MSMV_OK)
        return rc;
    if (flags & ~OPUS_MSMV_BWD_ACCUMULATE)
        return OPUS_MSMV_EINVAL_FLAGS;
    if (nsets < 1 || nsets > msmv_dfeat_max_sets())
        return OPUS_MSMV_EINVAL_DIMS;
    if (!msmv_backward_owner_supported(a))
        return OPUS_MSMV_EINVAL_FLAGS;          // the sorted sets exist for the owner algorithm only
    if (B == 0)
        return OPUS_MSMV_OK;
    if (!grad_outs || !workspaces || !dfeats)
        return OPUS_MSMV_EINVAL_PTR;
    for (int l = 0; l < L; ++l) {
        if (!dfeats[l])
            return OPUS_MSMV_EINVAL_PTR;
        if (!aligned16(dfeats[l]))
            return OPUS_MSMV_EINVAL_FLAGS;
        a.lvl[l].dfeat = dfeats[l];
    }
    if (workspace_bytes < msmv_backward_owner_workspace(B, N, Q, P))
        return OPUS_MSMV_EWORKSPACE;
    for (int i = 0; i < nsets; ++i) {
        if (!workspaces[i] || (a.npts > 0 && !grad_outs[i]))
            return OPUS_MSMV_EINVAL_PTR;
        if (!aligned16(workspaces[i]) || !aligned16(grad_outs[i]))
            return OPUS_MSMV_EINVAL_FLAGS;
    }
    const hipError_t e = msmv_launch_dfeat_sets(a, nsets, grad_outs, workspaces, (hipStream_t)stream,
                                                (flags & OPUS_MSMV_BWD_ACCUMULATE) != 0);
    return e == hipSuccess ? OPUS_MSMV_OK : OPUS_MSMV_ELAUNCH;
}

static int check_project_dims(int B, int Q, int T, int G, int P, int L, int N)
{
    if (L < 1 || L > OPUS_MSMV_MAX_LEVELS)
        return OPUS_MSMV_EINVAL_LEVELS;
    if (P < 1 || P > OPUS_MSMV_MAX_POINTS)
        return OPUS_MSMV_EINVAL_POINTS;
    if (B < 0 || Q < 0 || T < 1 || G < 1 || N < 2)
        return OPUS_MSMV_EINVAL_DIMS;
    if ((long long)B * T * G * Q * P * (L > 3 ? L : 3) > INT_MAX)
        return OPUS_MSMV_ETOOLARGE;
    return OPUS_MSMV_OK;
}

int opus_msmv_project_pack(const float *sample_points, const float *scale_weights, const float *occ2img,
                           int B, int Q, int T, int G, int P, int L, int N,
                           float image_h, float image_w, float eps, int flags,
                           float *loc, float *weights, void *stream)
{
    const int rc = check_project_dims(B, Q, T, G, P, L, N);
    if (rc != OPUS_MSMV_OK)
        return rc;
    if (flags & ~OPUS_MSMV_PACK_SHARED_FRAMES)
        return OPUS_MSMV_EINVAL_FLAGS;
    if ((long long)B * Q == 0)
        return OPUS_MSMV_OK;
    if (!sample_points || !scale_weights || !occ2img || !loc || !weights)
        return OPUS_MSMV_EINVAL_PTR;
    return msmv_launch_project_pack(sample_points, scale_weights, occ2img, B, Q, T, G, P, L, N,
                                    image_h, image_w, eps, flags & OPUS_MSMV_PACK_SHARED_FRAMES, loc, weights,
                                    (hipStream_t)stream) == hipSuccess
               ? OPUS_MSMV_OK : OPUS_MSMV_ELAUNCH;
}

int opus_msmv_project_pack_backward(const float *dloc, const float *dweights,
                                    const float *sample_points, const float *occ2img, const float *loc,
                                    int B, int Q, int T, int G, int P, int L, int N,
                                    float image_h, float image_w, float eps, int flags,
                                    float *dsample_points, float *dscale_weights, void *stream)
{
    const int rc = check_project_dims(B, Q, T, G, P, L, N);
    if (rc != OPUS_MSMV_OK)
        return rc;
    if (flags & ~OPUS_MSMV_PACK_SHARED_FRAMES)
        return OPUS_MSMV_EINVAL_FLAGS;
    if ((long long)B * Q == 0)
        return OPUS_MSMV_OK;
    if (!dloc || !dweights || !sample_points || !occ2img || !loc || !dsample_points || !dscale_weights)
        return OPUS_MSMV_EINVAL_PTR;
    return msmv_launch_project_pack_backward(dloc, dweights, sample_points, occ2img, loc, B, Q, T, G, P, L, N,
                                             image_h, image_w, eps, flags & OPUS_MSMV_PACK_SHARED_FRAMES,
                                             dsample_points, dscale_weights, (hipStream_t)stream) == hipSuccess
               ? OPUS_MSMV_OK : OPUS_MSMV_ELAUNCH;
}

int opus_msmv_relayout(const float *src, float *dst, int B, int T, int N, int G, int C, int H, int W,
                       int inverse, void *stream)
{
    if (B < 0 || T < 1 || N < 1 || G < 1 || C < 1 || H < 1 || W < 1)
        return OPUS_MSMV_EINVAL_DIMS;
    if ((long long)H * W > INT_MAX / 64 || (long long)T * N * G > INT_MAX)
        return OPUS_MSMV_ETOOLARGE;
    if (B == 0)
        return OPUS_MSMV_OK;
    if (!src || !dst)
        return OPUS_MSMV_EINVAL_PTR;
    const hipError_t e = msmv_launch_relayout(src, dst, B, T, N, G, C, H, W, inverse != 0, (hipStream_t)stream);
    if (e == hipErrorInvalidValue)
        return OPUS_MSMV_ETOOLARGE;
    return e == hipSuccess ? OPUS_MSMV_OK : OPUS_MSMV_ELAUNCH;
}

}  // extern "C"
