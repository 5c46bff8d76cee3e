// Caller-side fusion for the sampling op: everything sampling_4d does before it calls the op
// (models/opus_sampling.py:36-113 of the reference) in one kernel per direction instead of
// ~20 tensor ops over [B,T,N,Q,G*P,...] temporaries.
//
// forward : project each 3-D sampling point into the N cameras of its frame, keep the first
//           camera in which it is visible (camera 0 if none), write (u, v, view/(N-1)) in the
//           op's [B*T*G, Q, P, 3] layout, and copy the scale weights from [B,Q,G,T,P,L] into
//           the op's [B*G*T, Q, P, L] layout.  Points are packed in (B,T,G) order and weights
//           in (B,G,T) order, exactly as the reference does (opus_sampling.py:104-106 vs
//           :110-112): trained weights depend on that pairing.
// backward: d sample_points through the chosen camera's projection (the visibility mask and
//           the argmax carry no gradient; max(z, eps) passes gradient only where z > eps), and
//           d scale_weights as the inverse permutation.
#include "msmv_common.h"
#include "msmv_launch.h"

namespace {

struct ProjDims {
    int B, Q, T, G, P, L, N;
    int shared;          // the T frames share one set of points / weights: sources are [B,Q,G,P,*]
    float inv_w, inv_h, image_w, image_h, eps;
    long long npts;      // B*T*G*Q*P (the host guarantees < 2^31: slots are split with 32-bit magic divisions)
    MsmvDiv div_p, div_q, div_g, div_t;
};

// index math shared by both directions: point slot `idx` of the op's loc tensor
struct ProjIndex {
    int b, t, g, q, p;
    size_t src_pt;       // element offset of the point in sample_points [B,Q,T,G,P,3]
    size_t src_w;        // element offset of its weights in scale_weights [B,Q,G,T,P,L]
    size_t dst_w;        // element offset in the op's weights [B*G*T,Q,P,L]
};

__device__ __forceinline__ ProjIndex proj_index(uint32_t idx, const ProjDims &d)
{
    ProjIndex r;
    uint32_t x = idx, y = msmv_div(x, d.div_p);
    r.p = (int)(x - y * d.P); x = y; y = msmv_div(x, d.div_q);
    r.q = (int)(x - y * d.Q); x = y; y = msmv_div(x, d.div_g);
    r.g = (int)(x - y * d.G); x = y; y = msmv_div(x, d.div_t);
    r.t = (int)(x - y * d.T);
    r.b = (int)y;
    if (d.shared) {
        const size_t e = (((size_t)r.b * d.Q + r.q) * d.G + r.g) * d.P + r.p;
        r.src_pt = e * 3;
        r.src_w = e * d.L;
    } else {
        r.src_pt = (((((size_t)r.b * d.Q + r.q) * d.T + r.t) * d.G + r.g) * d.P + r.p) * 3;
        r.src_w = (((((size_t)r.b * d.Q + r.q) * d.G + r.g) * d.T + r.t) * d.P + r.p) * d.L;
    }
    r.dst_w = (((((size_t)r.b * d.G + r.g) * d.T + r.t) * d.Q + r.q) * d.P + r.p) * d.L;
    return r;
}

// Row r of M * (x, y, z, 1) as three rounded products summed left to right, every operation
// rounded on its own (no fused multiply-add): the camera choice below is index work, and a
// visibility test that sits on its threshold must fall the same way as in the plain float32
// statement of the projection (opus_sampling.py:54 computes it as a float32 matrix product).
// (The pragma, not __fmul_rn / __fadd_rn: those are plain * and + to the compiler, which
// contracts them into fused multiply-adds under HIP's default fp-contract=fast.)
__device__ __forceinline__ float proj_row(const float *M, float x, float y, float z)
{
#pragma clang fp contract(off)
    const float a = M[0] * x, b = M[1] * y, c = M[2] * z;
    return ((a + b) + c) + M[3];
}

__global__ void __launch_bounds__(256)
msmv_project_pack_fwd(const float *__restrict__ pts, const float *__restrict__ sw,
                      const float *__restrict__ occ2img, ProjDims d,
                      float *__restrict__ loc, float *__restrict__ w)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.npts)
        return;
    const ProjIndex ix = proj_index((uint32_t)idx, d);
    const float x = pts[ix.src_pt], y = pts[ix.src_pt + 1], z = pts[ix.src_pt + 2];
    const float *M = occ2img + ((size_t)ix.b * d.T * d.N + (size_t)ix.t * d.N) * 16;

    float u_sel = 0.f, v_sel = 0.f;
    int view = -1;
    for (int n = 0; n < d.N; ++n, M += 16) {
        // rows 0..2 of M * (x, y, z, 1)   (opus_sampling.py:54)
        const float cx = proj_row(M, x, y, z);
        const float cy = proj_row(M + 4, x, y, z);
        const float cz = proj_row(M + 8, x, y, z);
        const float hz = fmaxf(cz, d.eps);                            // :58
        const float u = (cx / hz) / d.image_w;                        // :59, :62 (two correctly rounded divisions)
        const float v = (cy / hz) / d.image_h;                        // :63
        const bool valid = cz > d.eps && v > 0.f && v < 1.f && u > 0.f && u < 1.f;   // :66-70
        // first valid camera wins; camera 0 is the default when none is valid (:94)
        if (n == 0 || (valid && view < 0)) {
            u_sel = u;
            v_sel = v;
        }
        if (valid && view < 0)
            view = n;
    }
    if (view < 0)
        view = 0;
    float *o = loc + (size_t)idx * 3;
    o[0] = u_sel;
    o[1] = v_sel;
    o[2] = (float)view / (float)(d.N - 1);                            // :101
    for (int l = 0; l < d.L; ++l)
        w[ix.dst_w + l] = sw[ix.src_w + l];
}

__global__ void __launch_bounds__(256)
msmv_project_pack_bwd(const float *__restrict__ dloc, const float *__restrict__ dw,
                      const float *__restrict__ pts, const float *__restrict__ occ2img,
                      const float *__restrict__ loc, ProjDims d,
                      float *__restrict__ dpts, float *__restrict__ dsw)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.npts)
        return;
    const ProjIndex ix = proj_index((uint32_t)idx, d);
    const float x = pts[ix.src_pt], y = pts[ix.src_pt + 1], z = pts[ix.src_pt + 2];
    int view = (int)roundf(loc[(size_t)idx * 3 + 2] * (float)(d.N - 1));
    view = min(max(view, 0), d.N - 1);
    const float *M = occ2img + ((size_t)ix.b * d.T * d.N + (size_t)ix.t * d.N + view) * 16;
    const float cx = M[0] * x + M[1] * y + M[2] * z + M[3];
    const float cy = M[4] * x + M[5] * y + M[6] * z + M[7];
    const float cz = M[8] * x + M[9] * y + M[10] * z + M[11];
    const float hz = fmaxf(cz, d.eps);
    const float du = dloc[(size_t)idx * 3], dv = dloc[(size_t)idx * 3 + 1];
    // u = cx / hz / image_w, v = cy / hz / image_h
    const float gcx = du / (hz * d.image_w), gcy = dv / (hz * d.image_h);
    const float ghz = -(gcx * cx + gcy * cy) / hz;
    const float gcz = cz > d.eps ? ghz : 0.f;                         // d max(z, eps) / d z
    dpts[ix.src_pt + 0] = gcx * M[0] + gcy * M[4] + gcz * M[8];
    dpts[ix.src_pt + 1] = gcx * M[1] + gcy * M[5] + gcz * M[9];
    dpts[ix.src_pt + 2] = gcx * M[2] + gcy * M[6] + gcz * M[10];
    for (int l = 0; l < d.L; ++l)
        dsw[ix.src_w + l] = dw[ix.dst_w + l];
}

// Backward when the frames share their points: one thread per source point (b, q, g, p) sums
// the T frames' contributions (the expand of models/opus_transformer.py:353,358 run backwards).
__global__ void __launch_bounds__(256)
msmv_project_pack_bwd_shared(const float *__restrict__ dloc, const float *__restrict__ dw,
                             const float *__restrict__ pts, const float *__restrict__ occ2img,
                             const float *__restrict__ loc, ProjDims d,
                             float *__restrict__ dpts, float *__restrict__ dsw)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nsrc = (long long)d.B * d.Q * d.G * d.P;
    if (e >= nsrc)
        return;
    uint32_t r = (uint32_t)e, r1 = msmv_div(r, d.div_p);
    const int p = (int)(r - r1 * d.P); r = r1; r1 = msmv_div(r, d.div_g);
    const int g = (int)(r - r1 * d.G); r = r1; r1 = msmv_div(r, d.div_q);
    const int q = (int)(r - r1 * d.Q);
    const int b = (int)r1;
    const float x = pts[e * 3], y = pts[e * 3 + 1], z = pts[e * 3 + 2];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    float gw[MSMV_MAX_LEVELS];
#pragma unroll
    for (int l = 0; l < MSMV_MAX_LEVELS; ++l)
        gw[l] = 0.f;
    for (int t = 0; t < d.T; ++t) {
        const size_t idx = ((((size_t)b * d.T + t) * d.G + g) * d.Q + q) * d.P + p;       // loc slot
        const size_t wrow = (((((size_t)b * d.G + g) * d.T + t) * d.Q + q) * d.P + p) * d.L;
        int view = (int)roundf(loc[idx * 3 + 2] * (float)(d.N - 1));
        view = min(max(view, 0), d.N - 1);
        const float *M = occ2img + ((size_t)b * d.T * d.N + (size_t)t * d.N + view) * 16;
        const float cx = M[0] * x + M[1] * y + M[2] * z + M[3];
        const float cy = M[4] * x + M[5] * y + M[6] * z + M[7];
        const float cz = M[8] * x + M[9] * y + M[10] * z + M[11];
        const float hz = fmaxf(cz, d.eps);
        const float du = dloc[idx * 3], dv = dloc[idx * 3 + 1];
        const float gcx = du / (hz * d.image_w), gcy = dv / (hz * d.image_h);
        const float ghz = -(gcx * cx + gcy * cy) / hz;
        const float gcz = cz > d.eps ? ghz : 0.f;
        gx += gcx * M[0] + gcy * M[4] + gcz * M[8];
        gy += gcx * M[1] + gcy * M[5] + gcz * M[9];
        gz += gcx * M[2] + gcy * M[6] + gcz * M[10];
#pragma unroll
        for (int l = 0; l < MSMV_MAX_LEVELS; ++l)
            if (l < d.L)
                gw[l] += dw[wrow + l];
    }
    dpts[e * 3] = gx;
    dpts[e * 3 + 1] = gy;
    dpts[e * 3 + 2] = gz;
#pragma unroll
    for (int l = 0; l < MSMV_MAX_LEVELS; ++l)
        if (l < d.L)
            dsw[e * d.L + l] = gw[l];
}

ProjDims make_dims(int B, int Q, int T, int G, int P, int L, int N, float image_h, float image_w, float eps,
                   int shared)
{
    ProjDims d;
    d.shared = shared;
    d.B = B; d.Q = Q; d.T = T; d.G = G; d.P = P; d.L = L; d.N = N;
    d.image_w = image_w; d.image_h = image_h;
    d.inv_w = 1.f / image_w; d.inv_h = 1.f / image_h;
    d.eps = eps;
    d.npts = (long long)B * T * G * Q * P;
    d.div_p = msmv_make_div(P); d.div_q = msmv_make_div(Q); d.div_g = msmv_make_div(G); d.div_t = msmv_make_div(T);
    return d;
}

}  // namespace

hipError_t msmv_launch_project_pack(const float *pts, const float *sw, const float *occ2img,
                                    int B, int Q, int T, int G, int P, int L, int N,
                                    float image_h, float image_w, float eps, int shared,
                                    float *loc, float *w, hipStream_t s)
{
    const ProjDims d = make_dims(B, Q, T, G, P, L, N, image_h, image_w, eps, shared);
    if (d.npts == 0)
        return hipSuccess;
    const unsigned nblk = (unsigned)((d.npts + 255) / 256);
    const hipError_t stale = hipPeekAtLastError();
    hipLaunchKernelGGL(msmv_project_pack_fwd, dim3(nblk), dim3(256), 0, s, pts, sw, occ2img, d, loc, w);
    return msmv_launch_status(stale);
}

hipError_t msmv_launch_project_pack_backward(const float *dloc, const float *dw, const float *pts,
                                             const float *occ2img, const float *loc,
                                             int B, int Q, int T, int G, int P, int L, int N,
                                             float image_h, float image_w, float eps, int shared,
                                             float *dpts, float *dsw, hipStream_t s)
{
    const ProjDims d = make_dims(B, Q, T, G, P, L, N, image_h, image_w, eps, shared);
    if (d.npts == 0)
        return hipSuccess;
    const hipError_t stale = hipPeekAtLastError();
    if (shared) {
        const long long nsrc = (long long)B * Q * G * P;
        hipLaunchKernelGGL(msmv_project_pack_bwd_shared, dim3((unsigned)((nsrc + 255) / 256)), dim3(256), 0, s,
                           dloc, dw, pts, occ2img, loc, d, dpts, dsw);
        return msmv_launch_status(stale);
    }
    const unsigned nblk = (unsigned)((d.npts + 255) / 256);
    hipLaunchKernelGGL(msmv_project_pack_bwd, dim3(nblk), dim3(256), 0, s, dloc, dw, pts, occ2img, loc, d, dpts, dsw);
    return msmv_launch_status(stale);
}
