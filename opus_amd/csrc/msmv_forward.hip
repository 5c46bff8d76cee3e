// Forward kernels of msmv sampling for gfx950.
//
// out[b,q,c,p] = sum_l w[b,q,p,l] * bilinear(feat_l[b, view(b,q,p)], x, y)[c]
// (what msmv_sampling_forward.cu:75-164 / :166-267 of the reference compute).
//
// Fast path (C == 64, L in {4,5}): a sampled point is owned by a ROW of 16 lanes, each
// lane holding 4 consecutive channels, so one wave works on 4 points at once and every
// corner fetch is one 16-byte load per lane = four fully used 256-byte segments per wave
// instruction.  All 4*L corner loads of a point are issued before any is consumed.  Taps
// are computed redundantly by the 16 lanes of a row (cheaper than a broadcast).  The
// [B,Q,C,P] output is P-strided per channel; for P in {1,2,4} the 4 points of a wave are
// whole queries and the rows exchange values (2x2 / 4x4 lane transposes) so that every
// lane still issues one 16-byte store into a contiguous 1-KiB run per wave.  In row mode
// (PT == MSMV_PT_ROWS: output [Bs,Q,G,T*P,C]) a point's 64 channels are one contiguous
// 256-byte row and no exchange is needed.
#include "msmv_common.h"
#include "msmv_launch.h"

namespace {

constexpr int FWD_THREADS = 256;          // 4 waves
#ifndef MSMV_STREAM_LEVELS
#define MSMV_STREAM_LEVELS 1        // the finest levels whose corner loads carry the streaming hint
#endif
#ifndef MSMV_FWD_STREAM_L0
#define MSMV_FWD_STREAM_L0 1
#endif
#ifndef MSMV_FWD_GPW
#define MSMV_FWD_GPW 1
#endif
constexpr int FWD_GROUPS_PER_WAVE = MSMV_FWD_GPW;    // 4-point groups each wave walks through

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
// The finest level is the largest (3/4 of the feature bytes) and the least re-used (a layer's
// points touch under half of its pixels): its corners are fetched with a streaming hint so that
// they do not displace the coarser levels, which every point re-reads, from L2.
typedef float fwd_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_level(const float *p, bool stream)   // (constant after unrolling)
{
    if (stream && MSMV_FWD_STREAM_L0) {
        const fwd_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const fwd_f32x4 *>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    } else {
        return ld4(p);
    }
}

template <int NL, int PT>
__global__ void __launch_bounds__(FWD_THREADS)
msmv_fwd_c64(const MsmvArgs a)
{
    const int lane = threadIdx.x & 63;
    const int sub = lane >> 4;        // which of the wave's 4 points
    const int l16 = lane & 15;        // channel quad: channels 4*l16 .. 4*l16+3
    const int wave = threadIdx.x >> 6;
    const int blk = msmv_xcd_remap(blockIdx.x, gridDim.x);
    const int ngroups = (a.npts + 3) >> 2;
    const int g0 = (blk * (FWD_THREADS / 64) + wave) * FWD_GROUPS_PER_WAVE;

#pragma unroll 1
    for (int gi = 0; gi < FWD_GROUPS_PER_WAVE; ++gi) {
        const int g = g0 + gi;
        if (g >= ngroups)
            break;                                   // wave-uniform
        const int pt_raw = g * 4 + sub;
        const bool act = pt_raw < a.npts;
        const int pt = act ? pt_raw : a.npts - 1;    // keep addresses valid for idle rows

        const float *lp = a.loc + (size_t)pt * 3;
        const float lx = lp[0], ly = lp[1], lv = lp[2];
        float wl[NL];
        if constexpr (NL == 4) {
            const float4 w4 = ld4(a.w + (size_t)pt * 4);
            wl[0] = w4.x; wl[1] = w4.y; wl[2] = w4.z; wl[3] = w4.w;
        } else {
#pragma unroll
            for (int l = 0; l < NL; ++l)
                wl[l] = a.w[(size_t)pt * NL + l];
        }

        const int view = msmv_view_of(lv, a.N);
        const bool view_ok = view >= 0;
        const int b = msmv_batch_of(pt, a);
        const size_t slab = (size_t)b * a.N + (view_ok ? view : 0);
        [[maybe_unused]] int orow = 0;
        if constexpr (PT == MSMV_PT_ROWS)
            orow = msmv_row_of(pt, a);

        MsmvTap tap[NL];
        float4 v[NL][4];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int H = a.lvl[l].H, W = a.lvl[l].W;
            tap[l] = msmv_make_tap(lx, ly, H, W, view_ok);
            const float *base = a.lvl[l].feat + (slab * (size_t)(H * W)) * 64 + l16 * 4;
#ifdef MSMV_ABL_SKIP_COARSE
            // development aid (wrong results): levels 2.. cost no memory access at all - the bound on
            // what serving them from an LDS-staged copy could save
            if (l >= 2) {
                const float c = __int_as_float(0x3f800000 + tap[l].o1 + tap[l].o4);
                v[l][0] = v[l][1] = v[l][2] = v[l][3] = make_float4(c, c, c, c);
                continue;
            }
#endif
            v[l][0] = ld4_level(base + (size_t)tap[l].o1 * 64, l < MSMV_STREAM_LEVELS);
            v[l][1] = ld4_level(base + (size_t)tap[l].o2 * 64, l < MSMV_STREAM_LEVELS);
            v[l][2] = ld4_level(base + (size_t)tap[l].o3 * 64, l < MSMV_STREAM_LEVELS);
            v[l][3] = ld4_level(base + (size_t)tap[l].o4 * 64, l < MSMV_STREAM_LEVELS);
        }

        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const MsmvTap &t = tap[l];
            // an out-of-image corner counts as a zero VALUE (forward.cu:48-67): what was loaded for
            // it (a clamped in-image pixel) is discarded, so that a non-finite neighbour cannot
            // turn 0 * inf into NaN
#define MSMV_FWD_CH(X)                                                             \
            {                                                                      \
                const float v1 = t.c1 ? v[l][0].X : 0.f, v2 = t.c2 ? v[l][1].X : 0.f; \
                const float v3 = t.c3 ? v[l][2].X : 0.f, v4 = t.c4 ? v[l][3].X : 0.f; \
                acc.X = msmv_accum_level(t, acc.X, msmv_blend4(t, v1, v2, v3, v4), wl[l]); \
            }
            MSMV_FWD_CH(x) MSMV_FWD_CH(y) MSMV_FWD_CH(z) MSMV_FWD_CH(w)
#undef MSMV_FWD_CH
        }

        if constexpr (PT == MSMV_PT_ROWS) {
            if (act)
                *reinterpret_cast<float4 *>(a.out + (size_t)orow * 64 + l16 * 4) = acc;
        } else if constexpr (PT == 1) {
            if (act)
                *reinterpret_cast<float4 *>(a.out + (size_t)pt * 64 + l16 * 4) = acc;
        } else if constexpr (PT == 2) {
            // rows (2k, 2k+1) are p = 0, 1 of one query; memory order is [c][p]
            const bool odd = sub & 1;
            const float s0 = odd ? acc.x : acc.z, s1 = odd ? acc.y : acc.w;
            const float r0 = __shfl_xor(s0, 16), r1 = __shfl_xor(s1, 16);
            const float4 o = odd ? make_float4(r0, acc.z, r1, acc.w) : make_float4(acc.x, r0, acc.y, r1);
            if (act) {
                const size_t bq = (size_t)(pt >> 1);
                *reinterpret_cast<float4 *>(a.out + (bq * 64 + l16 * 4 + (odd ? 2 : 0)) * 2) = o;
            }
        } else if constexpr (PT == 4) {
            // rows 0..3 are p = 0..3 of one query: 4x4 transpose across the rows
            const bool b0 = sub & 1, b1 = sub & 2;
            float sa = b0 ? acc.x : acc.y, sb = b0 ? acc.z : acc.w;
            float ra = __shfl_xor(sa, 16), rb = __shfl_xor(sb, 16);
            float4 m = b0 ? make_float4(ra, acc.y, rb, acc.w) : make_float4(acc.x, ra, acc.z, rb);
            sa = b1 ? m.x : m.z;
            sb = b1 ? m.y : m.w;
            ra = __shfl_xor(sa, 32);
            rb = __shfl_xor(sb, 32);
            const float4 o = b1 ? make_float4(ra, rb, m.z, m.w) : make_float4(m.x, m.y, ra, rb);
            if (act) {
                const size_t bq = (size_t)(pt >> 2);
                *reinterpret_cast<float4 *>(a.out + (bq * 64 + l16 * 4 + sub) * 4) = o;
            }
        } else {
            if (act) {
                const int bq = pt / a.P, p = pt - bq * a.P;
                float *o = a.out + ((size_t)bq * 64 + l16 * 4) * a.P + p;
                o[0] = acc.x;
                o[(size_t)a.P] = acc.y;
                o[(size_t)a.P * 2] = acc.z;
                o[(size_t)a.P * 3] = acc.w;
            }
        }
    }
}

// Any C, L, P: one wave per point, lanes stride over the channels.
__global__ void __launch_bounds__(256)
msmv_fwd_generic(const MsmvArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    for (int pt = blockIdx.x * (blockDim.x >> 6) + wave; pt < a.npts; pt += nwaves) {
        const float lx = a.loc[(size_t)pt * 3], ly = a.loc[(size_t)pt * 3 + 1];
        const int view = msmv_view_of(a.loc[(size_t)pt * 3 + 2], a.N);
        const bool view_ok = view >= 0;
        const int b = msmv_batch_of(pt, a);
        const int bq = pt / a.P, p = pt - bq * a.P;
        const size_t slab = (size_t)b * a.N + (view_ok ? view : 0);
        const size_t row = a.rows ? (size_t)msmv_row_of(pt, a) : 0;
        for (int c = lane; c < a.C; c += 64) {
            float acc = 0.f;
            for (int l = 0; l < a.L; ++l) {
                const int H = a.lvl[l].H, W = a.lvl[l].W;
                const MsmvTap t = msmv_make_tap(lx, ly, H, W, view_ok);
                if (!t.ok)
                    continue;
                const float *base = a.lvl[l].feat + (slab * (size_t)(H * W)) * a.C + c;
                const float v1 = t.c1 ? base[(size_t)t.o1 * a.C] : 0.f;
                const float v2 = t.c2 ? base[(size_t)t.o2 * a.C] : 0.f;
                const float v3 = t.c3 ? base[(size_t)t.o3 * a.C] : 0.f;
                const float v4 = t.c4 ? base[(size_t)t.o4 * a.C] : 0.f;
                const float val = t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4;
                acc += val * a.w[(size_t)pt * a.L + l];
            }
            if (a.rows)
                a.out[row * a.C + c] = acc;
            else
                a.out[((size_t)bq * a.C + c) * a.P + p] = acc;
        }
    }
}

template <int NL>
void launch_c64(const MsmvArgs &a, hipStream_t s)
{
    const int ngroups = (a.npts + 3) / 4;
    const int per_block = (FWD_THREADS / 64) * FWD_GROUPS_PER_WAVE;
    const int nblk = (ngroups + per_block - 1) / per_block;
    switch (a.rows ? MSMV_PT_ROWS : a.P) {
    case MSMV_PT_ROWS: hipLaunchKernelGGL((msmv_fwd_c64<NL, MSMV_PT_ROWS>), dim3(nblk), dim3(FWD_THREADS), 0, s, a); break;
    case 1: hipLaunchKernelGGL((msmv_fwd_c64<NL, 1>), dim3(nblk), dim3(FWD_THREADS), 0, s, a); break;
    case 2: hipLaunchKernelGGL((msmv_fwd_c64<NL, 2>), dim3(nblk), dim3(FWD_THREADS), 0, s, a); break;
    case 4: hipLaunchKernelGGL((msmv_fwd_c64<NL, 4>), dim3(nblk), dim3(FWD_THREADS), 0, s, a); break;
    default: hipLaunchKernelGGL((msmv_fwd_c64<NL, 0>), dim3(nblk), dim3(FWD_THREADS), 0, s, a); break;
    }
}

}  // namespace

hipError_t msmv_launch_forward(const MsmvArgs &a, hipStream_t s, int force_generic)
{
    if (a.npts == 0)
        return hipSuccess;
    const hipError_t stale = hipPeekAtLastError();
    if (!force_generic && a.C == 64 && a.L == 4) {
        launch_c64<4>(a, s);
    } else if (!force_generic && a.C == 64 && a.L == 5) {
        launch_c64<5>(a, s);
    } else {
        const int waves_per_block = 4;
        int nblk = (a.npts + waves_per_block - 1) / waves_per_block;
        if (nblk > 256 * 32)
            nblk = 256 * 32;
        hipLaunchKernelGGL(msmv_fwd_generic, dim3(nblk), dim3(64 * waves_per_block), 0, s, a);
    }
    return msmv_launch_status(stale);
}
