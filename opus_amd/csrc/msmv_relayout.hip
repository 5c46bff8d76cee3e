// Producer-side re-layout for the sampling op: the decoder's
//   feat.reshape(B,T,N,G,C,H,W).permute(0,1,3,2,5,6,4).reshape(B*T*G,N,H,W,C).contiguous()
// (models/opus_transformer.py:113-126 of the reference, MSMV_CUDA branch) as a tiled
// transpose, and its inverse (the gradient's way back to the backbone's layout).
//
// Per (b, t, n, g) the source is a [C][S] matrix (S = H*W contiguous pixels per channel) and
// the destination a [S][C] matrix in slab (b*T + t)*G + g, view n.  A workgroup moves one
// 64-channel x 64-pixel tile through LDS: 256-byte reads along the source's contiguous axis,
// 256-byte writes along the destination's; with C == 64 a tile's destination is one
// contiguous 16-KiB run.  HBM-bound: every byte is read once and written once.
#include "msmv_launch.h"

namespace {

constexpr int RL_TILE = 64;
constexpr int RL_THREADS = 256;
constexpr int RL_STRIDE = RL_TILE + 1;       // odd row stride: conflict-free column access

// Every byte is read once and written once: streaming (non-temporal) accesses keep the copy
// from displacing what the neighbouring kernels hold in L2 / Infinity Cache.
typedef float rl_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 rl_load_stream(const float *p)
{
    const rl_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const rl_f32x4 *>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void rl_store_stream(float *p, const float (&v)[4])
{
    const rl_f32x4 x = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(x, reinterpret_cast<rl_f32x4 *>(p));
}

struct RelayoutDims {
    int B, T, N, G, C, S;                     // S = H*W
    int s_tiles, c_tiles;
};

// INVERSE == false: chw -> hwc (forward);  true: hwc -> chw.
// `chw` is [B, T*N, G*C, S], `hwc` is [B*T*G, N, S, C].
template <bool INVERSE, bool VEC>
__global__ void __launch_bounds__(RL_THREADS)
msmv_relayout_kernel(const float *__restrict__ src, float *__restrict__ dst, RelayoutDims d)
{
    __shared__ float tile[RL_TILE * RL_STRIDE];          // [channel][pixel]
    int id = blockIdx.x;
    const int st = id % d.s_tiles; id /= d.s_tiles;
    const int ct = id % d.c_tiles; id /= d.c_tiles;
    const int g = id % d.G; id /= d.G;
    const int n = id % d.N; id /= d.N;
    const int t = id % d.T;
    const int b = id / d.T;
    const int s0 = st * RL_TILE, c0 = ct * RL_TILE;
    const int ns = min(RL_TILE, d.S - s0), nc = min(RL_TILE, d.C - c0);

    // element offsets of the tile's origin in both layouts
    const size_t chw0 = ((((size_t)b * d.T + t) * d.N + n) * d.G + g) * (size_t)d.C * d.S + (size_t)c0 * d.S + s0;
    const size_t hwc0 = (((((size_t)b * d.T + t) * d.G + g) * d.N + n) * (size_t)d.S + s0) * d.C + c0;
    const float *chw_src = src + chw0;
    const float *hwc_src = src + hwc0;
    float *chw_dst = dst + chw0;
    float *hwc_dst = dst + hwc0;

    const int q = threadIdx.x & 15;           // which group of 4 along the contiguous axis
    const int r = threadIdx.x >> 4;           // row within a pass of 16 rows

    if constexpr (!INVERSE) {
        // read channel rows (pixels contiguous)
#pragma unroll
        for (int k = 0; k < RL_TILE / 16; ++k) {
            const int c = r + 16 * k;
            if (c < nc) {
                const float *p = chw_src + (size_t)c * d.S + q * 4;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (VEC && q * 4 + 3 < ns) {
                    const float4 x = rl_load_stream(p);
                    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (q * 4 + j < ns)
                            v[j] = p[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    tile[c * RL_STRIDE + q * 4 + j] = v[j];
            }
        }
        __syncthreads();
        // write pixel rows (channels contiguous)
#pragma unroll
        for (int k = 0; k < RL_TILE / 16; ++k) {
            const int s = r + 16 * k;
            if (s < ns) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j] = tile[(q * 4 + j) * RL_STRIDE + s];
                float *p = hwc_dst + (size_t)s * d.C + q * 4;
                if (VEC && q * 4 + 3 < nc) {
                    rl_store_stream(p, v);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (q * 4 + j < nc)
                            p[j] = v[j];
                }
            }
        }
    } else {
        // read pixel rows (channels contiguous)
#pragma unroll
        for (int k = 0; k < RL_TILE / 16; ++k) {
            const int s = r + 16 * k;
            if (s < ns) {
                const float *p = hwc_src + (size_t)s * d.C + q * 4;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (VEC && q * 4 + 3 < nc) {
                    const float4 x = rl_load_stream(p);
                    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (q * 4 + j < nc)
                            v[j] = p[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    tile[(q * 4 + j) * RL_STRIDE + s] = v[j];
            }
        }
        __syncthreads();
        // write channel rows (pixels contiguous)
#pragma unroll
        for (int k = 0; k < RL_TILE / 16; ++k) {
            const int c = r + 16 * k;
            if (c < nc) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j] = tile[c * RL_STRIDE + q * 4 + j];
                float *p = chw_dst + (size_t)c * d.S + q * 4;
                if (VEC && q * 4 + 3 < ns) {
                    rl_store_stream(p, v);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (q * 4 + j < ns)
                            p[j] = v[j];
                }
            }
        }
    }
}

}  // namespace

hipError_t msmv_launch_relayout(const float *src, float *dst, int B, int T, int N, int G, int C, int H, int W,
                                bool inverse, hipStream_t s)
{
    RelayoutDims d;
    d.B = B; d.T = T; d.N = N; d.G = G; d.C = C; d.S = H * W;
    d.s_tiles = (d.S + RL_TILE - 1) / RL_TILE;
    d.c_tiles = (C + RL_TILE - 1) / RL_TILE;
    const long long nblk = (long long)B * T * N * G * d.s_tiles * d.c_tiles;
    if (nblk == 0)
        return hipSuccess;
    if (nblk > 0x7FFFFFFFll)
        return hipErrorInvalidValue;
    // 16-byte accesses need both row pitches (S and C floats) and both bases to allow them
    const bool vec = (d.S % 4 == 0) && (C % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
    const dim3 grid((unsigned)nblk), block(RL_THREADS);
    const hipError_t stale = hipPeekAtLastError();
    if (!inverse) {
        if (vec) hipLaunchKernelGGL((msmv_relayout_kernel<false, true>), grid, block, 0, s, src, dst, d);
        else     hipLaunchKernelGGL((msmv_relayout_kernel<false, false>), grid, block, 0, s, src, dst, d);
    } else {
        if (vec) hipLaunchKernelGGL((msmv_relayout_kernel<true, true>), grid, block, 0, s, src, dst, d);
        else     hipLaunchKernelGGL((msmv_relayout_kernel<true, false>), grid, block, 0, s, src, dst, d);
    }
    return msmv_launch_status(stale);
}
