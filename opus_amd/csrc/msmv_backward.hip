// Backward kernels of msmv sampling for gfx950.
//
// Gradients of out[b,q,c,p] = sum_l w_l * bilinear(feat_l[b, view], x, y)[c]
// (what msmv_sampling_backward.cu:29-105, :108-224, :226-361 of the reference compute):
//   d feat_l[corner] += corner_weight * w_l * g          (scatter)
//   d w_l             = sum_c g[c] * val_l[c]
//   d x               = sum_l (W_l-1) * w_l * sum_c g[c] * (hh*(v2-v1) + lh*(v4-v3))
//   d y               = sum_l (H_l-1) * w_l * sum_c g[c] * (hw*(v3-v1) + lw*(v4-v2))
//   d view            = 0  (the reference never writes that slot, backward.cu:103-104)
//
// Two algorithms:
//  * atomic  - one wave per point, lane = channel.  d w / d loc are summed across the wave
//              with DPP and stored once by one lane (the reference issues 64 same-address
//              atomics per value); d feat uses fp32 global atomics, each wave instruction
//              covering the 256 contiguous bytes of one corner.  Works for every shape.
//              Bounded by the chip's float-atomic rate (~1.3 TB/s of added bytes).
//  * owner   - no atomics and no clear of d feat: points are first binned by (batch, view);
//              every run of TILE_PIX consecutive pixels of a (batch, view, level) slab is
//              owned by one workgroup, which scans its bin, accumulates the contributions
//              that land in its run in LDS (ds_add_f32) and then writes the run once with
//              plain 16-byte stores (zeros included).  d loc / d w come from a separate
//              gather kernel shaped like the forward (row of 16 lanes per point).  C == 64.
#include "msmv_common.h"
#include "msmv_launch.h"

namespace {

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// ======================================================================================
// atomic algorithm
// ======================================================================================
__global__ void __launch_bounds__(256)
msmv_bwd_atomic(const MsmvArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = blockDim.x >> 6;
    const int nwaves = gridDim.x * wpb;
    const int blk = msmv_xcd_remap(blockIdx.x, gridDim.x);
    // contiguous chunk of points per wave keeps a batch slab on one XCD
    const int per_wave = (a.npts + nwaves - 1) / nwaves;
    const int first = (blk * wpb + wave) * per_wave;
    const int last = min(first + per_wave, a.npts);

    for (int pt = first; pt < last; ++pt) {
        const float lx = a.loc[(size_t)pt * 3], ly = a.loc[(size_t)pt * 3 + 1];
        const int view = msmv_view_of(a.loc[(size_t)pt * 3 + 2], a.N);
        const bool view_ok = view >= 0;
        const int b = msmv_batch_of(pt, a);
        const int bq = pt / a.P, p = pt - bq * a.P;
        const size_t slab = (size_t)b * a.N + (view_ok ? view : 0);
        float gx = 0.f, gy = 0.f;
        for (int l = 0; l < a.L; ++l) {
            const int H = a.lvl[l].H, W = a.lvl[l].W;
            const MsmvTap t = msmv_make_tap(lx, ly, H, W, view_ok);
            float dwl = 0.f;
            if (t.ok) {                                   // wave-uniform
                const float wl = a.w[(size_t)pt * a.L + l];
                const size_t off = (slab * (size_t)(H * W)) * a.C;
                const float *base = a.lvl[l].feat + off;
                float *gbase = a.lvl[l].dfeat + off;
                for (int c = lane; c < a.C; c += 64) {
                    const float top = a.gout[((size_t)bq * a.C + c) * a.P + p];
                    const float tgv = top * wl;
                    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
                    if (t.c1) { v1 = base[(size_t)t.o1 * a.C + c]; unsafeAtomicAdd(gbase + (size_t)t.o1 * a.C + c, t.w1 * tgv); }
                    if (t.c2) { v2 = base[(size_t)t.o2 * a.C + c]; unsafeAtomicAdd(gbase + (size_t)t.o2 * a.C + c, t.w2 * tgv); }
                    if (t.c3) { v3 = base[(size_t)t.o3 * a.C + c]; unsafeAtomicAdd(gbase + (size_t)t.o3 * a.C + c, t.w3 * tgv); }
                    if (t.c4) { v4 = base[(size_t)t.o4 * a.C + c]; unsafeAtomicAdd(gbase + (size_t)t.o4 * a.C + c, t.w4 * tgv); }
                    const float gww = t.hh * (v2 - v1) + t.lh * (v4 - v3);
                    const float ghw = t.hw * (v3 - v1) + t.lw * (v4 - v2);
                    const float val = t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4;
                    dwl += top * val;
                    gx += (float)(W - 1) * gww * tgv;
                    gy += (float)(H - 1) * ghw * tgv;
                }
            }
            dwl = msmv_wave_sum(dwl);
            if (lane == 0)
                a.dw[(size_t)pt * a.L + l] = dwl;
        }
        gx = msmv_wave_sum(gx);
        gy = msmv_wave_sum(gy);
        if (lane == 0) {
            a.dloc[(size_t)pt * 3 + 0] = gx;
            a.dloc[(size_t)pt * 3 + 1] = gy;
            a.dloc[(size_t)pt * 3 + 2] = 0.f;
        }
    }
}

// ======================================================================================
// owner algorithm: binning
// ======================================================================================
struct BinEntry {       // 16 bytes, one aligned load
    int pt;
    float x, y;
    int pad;
};

// workspace layout: [ counts: B*N ints ][ cursors: B*N ints ][ pad to 256 B ][ entries: npts * 16 B ]
struct OwnerWs {
    int *counts;
    int *cursors;
    BinEntry *entries;
};

__host__ __device__ inline size_t owner_ws_header_bytes(int B, int N)
{
    const size_t h = (size_t)B * N * 2 * sizeof(int);
    return (h + 255) & ~(size_t)255;
}

// FILL == false: count points per (batch, view).  FILL == true: write each point's entry at
// b*QP + (points of b in lower views) + its slot among the points of its own (b, view).
template <bool FILL>
__global__ void __launch_bounds__(256)
msmv_bin_points(const MsmvArgs a, OwnerWs ws)
{
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int key = -1;
    float x = 0.f, y = 0.f;
    int b = 0, view = -1;
    if (pt < a.npts) {
        x = a.loc[(size_t)pt * 3];
        y = a.loc[(size_t)pt * 3 + 1];
        view = msmv_view_of(a.loc[(size_t)pt * 3 + 2], a.N);
        b = msmv_batch_of(pt, a);
        if (view >= 0)
            key = b * a.N + view;
    }
    // group the wave's lanes by key; one atomic per distinct key per wave
    unsigned long long todo = __ballot(key >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __builtin_amdgcn_readlane(key, leader);
        const unsigned long long same = __ballot(key == k);
        const int n = __popcll(same);
        if (!FILL) {
            if (lane == leader)
                atomicAdd(ws.counts + k, n);
        } else {
            int slot0 = 0;
            if (lane == leader)
                slot0 = atomicAdd(ws.cursors + k, n);
            slot0 = __builtin_amdgcn_readlane(slot0, leader);
            if (key == k) {
                const int rank = __popcll(same & ((1ull << lane) - 1ull));
                int lower = 0;
                for (int v = 0; v < view; ++v)
                    lower += ws.counts[b * a.N + v];
                BinEntry e;
                e.pt = pt; e.x = x; e.y = y; e.pad = 0;
                ws.entries[(size_t)b * a.qp + lower + slot0 + rank] = e;
            }
        }
        todo &= ~same;
    }
}

// ======================================================================================
// owner algorithm: d feat tiles
// ======================================================================================
constexpr int TILE_PIX = 176;            // pixels per tile: 176 * 256 B = 44 KiB of LDS, 3 workgroups per CU
constexpr int OWN_THREADS = 256;

struct OwnerGeom {
    int tiles_per_level[MSMV_MAX_LEVELS];   // tiles per (batch, view) slab of the level
    int tiles_per_b;                         // N * sum over levels
    int p_magic_valid;                       // unused placeholder for alignment
};

__global__ void __launch_bounds__(OWN_THREADS)
msmv_bwd_owner_dfeat(const MsmvArgs a, OwnerWs ws, OwnerGeom geo)
{
    __shared__ float tile[TILE_PIX * 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- which tile: logical order [b][level, coarse first][view][tile]
    int id = msmv_xcd_remap(blockIdx.x, gridDim.x);
    const int b = id / geo.tiles_per_b;
    id -= b * geo.tiles_per_b;
    int l = a.L - 1;
    for (; l > 0; --l) {
        const int n = geo.tiles_per_level[l] * a.N;
        if (id < n)
            break;
        id -= n;
    }
    const int view = id / geo.tiles_per_level[l];
    const int tile_idx = id - view * geo.tiles_per_level[l];
    const int H = a.lvl[l].H, W = a.lvl[l].W;
    const int S = H * W;
    const int s0 = tile_idx * TILE_PIX;                 // first pixel of the run
    const int npix = min(TILE_PIX, S - s0);

    // ---- clear the tile
    {
        float4 *t4 = reinterpret_cast<float4 *>(tile);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = tid; i < npix * 16; i += OWN_THREADS)
            t4[i] = z;
    }
    __syncthreads();

    // ---- scan this (batch, view)'s bin
    int lower = 0;
    for (int v = 0; v < view; ++v)
        lower += ws.counts[b * a.N + v];
    const int n = ws.counts[b * a.N + view];
    const BinEntry *list = ws.entries + (size_t)b * a.qp + lower;

    for (int i0 = wave * 64; i0 < n; i0 += (OWN_THREADS / 64) * 64) {
        const int i = i0 + lane;
        const bool valid = i < n;
        BinEntry e;
        if (valid) {
            const int4 raw = *reinterpret_cast<const int4 *>(list + i);
            e.pt = raw.x; e.x = __int_as_float(raw.y); e.y = __int_as_float(raw.z);
        } else {
            e.pt = 0; e.x = 0.f; e.y = 0.f;
        }
        const MsmvTap t = msmv_make_tap(e.x, e.y, H, W, valid);
        // corners that fall into this tile's run of pixels
        const bool in1 = t.c1 && (unsigned)(t.o1 - s0) < (unsigned)npix;
        const bool in2 = t.c2 && (unsigned)(t.o2 - s0) < (unsigned)npix;
        const bool in3 = t.c3 && (unsigned)(t.o3 - s0) < (unsigned)npix;
        const bool in4 = t.c4 && (unsigned)(t.o4 - s0) < (unsigned)npix;
        const float cw1 = in1 ? t.w1 : 0.f, cw2 = in2 ? t.w2 : 0.f;
        const float cw3 = in3 ? t.w3 : 0.f, cw4 = in4 ? t.w4 : 0.f;
        const int co1 = in1 ? (t.o1 - s0) * 64 : -1, co2 = in2 ? (t.o2 - s0) * 64 : -1;
        const int co3 = in3 ? (t.o3 - s0) * 64 : -1, co4 = in4 ? (t.o4 - s0) * 64 : -1;
        const int bq = e.pt / a.P, p = e.pt - bq * a.P;
        const int goff = bq * 64 * a.P + p;            // < 2^31: host checks B*Q*C*P
        const int woff = e.pt * a.L + l;

        unsigned long long hits = __ballot(in1 || in2 || in3 || in4);
        while (hits) {
            // up to two hits per round so that both gradient loads are in flight together
            const int j0 = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            const bool two = hits != 0;
            const int j1 = two ? __ffsll((long long)hits) - 1 : j0;
            if (two)
                hits &= hits - 1;

            const int goff0 = __builtin_amdgcn_readlane(goff, j0), goff1 = __builtin_amdgcn_readlane(goff, j1);
            const int woff0 = __builtin_amdgcn_readlane(woff, j0), woff1 = __builtin_amdgcn_readlane(woff, j1);
            const float g0 = a.gout[(size_t)goff0 + (size_t)lane * a.P] * a.w[woff0];
            const float g1 = a.gout[(size_t)goff1 + (size_t)lane * a.P] * a.w[woff1];

#define MSMV_OWNER_ADD(J, G)                                                                   \
            {                                                                                  \
                const int o1 = __builtin_amdgcn_readlane(co1, J), o2 = __builtin_amdgcn_readlane(co2, J); \
                const int o3 = __builtin_amdgcn_readlane(co3, J), o4 = __builtin_amdgcn_readlane(co4, J); \
                const float k1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cw1), J)); \
                const float k2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cw2), J)); \
                const float k3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cw3), J)); \
                const float k4 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cw4), J)); \
                if (o1 >= 0) unsafeAtomicAdd(&tile[o1 + lane], k1 * (G));                       \
                if (o2 >= 0) unsafeAtomicAdd(&tile[o2 + lane], k2 * (G));                       \
                if (o3 >= 0) unsafeAtomicAdd(&tile[o3 + lane], k3 * (G));                       \
                if (o4 >= 0) unsafeAtomicAdd(&tile[o4 + lane], k4 * (G));                       \
            }
            MSMV_OWNER_ADD(j0, g0)
            if (two)
                MSMV_OWNER_ADD(j1, g1)
#undef MSMV_OWNER_ADD
        }
    }
    __syncthreads();

    // ---- write the run once
    {
        const float4 *t4 = reinterpret_cast<const float4 *>(tile);
        float4 *dst = reinterpret_cast<float4 *>(
            a.lvl[l].dfeat + (((size_t)b * a.N + view) * (size_t)S + s0) * 64);
        for (int i = tid; i < npix * 16; i += OWN_THREADS)
            dst[i] = t4[i];
    }
}

// ======================================================================================
// owner algorithm: d loc / d w gather (row of 16 lanes per point, like the forward)
// ======================================================================================
constexpr int GAT_THREADS = 256;
constexpr int GAT_GROUPS_PER_WAVE = 4;

template <int NL, int PT>
__global__ void __launch_bounds__(GAT_THREADS)
msmv_bwd_gather_c64(const MsmvArgs a)
{
    const int lane = threadIdx.x & 63;
    const int sub = lane >> 4;
    const int l16 = lane & 15;
    const int wave = threadIdx.x >> 6;
    const int blk = msmv_xcd_remap(blockIdx.x, gridDim.x);
    const int ngroups = (a.npts + 3) >> 2;
    const int g0 = (blk * (GAT_THREADS / 64) + wave) * GAT_GROUPS_PER_WAVE;

#pragma unroll 1
    for (int gi = 0; gi < GAT_GROUPS_PER_WAVE; ++gi) {
        const int g = g0 + gi;
        if (g >= ngroups)
            break;
        const int pt_raw = g * 4 + sub;
        const bool act = pt_raw < a.npts;
        const int pt = act ? pt_raw : a.npts - 1;

        const float *lp = a.loc + (size_t)pt * 3;
        const float lx = lp[0], ly = lp[1], lv = lp[2];
        float wl[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l)
            wl[l] = a.w[(size_t)pt * NL + l];

        const int view = msmv_view_of(lv, a.N);
        const bool view_ok = view >= 0;
        const int b = msmv_batch_of(pt, a);
        const size_t slab = (size_t)b * a.N + (view_ok ? view : 0);

        MsmvTap tap[NL];
        float4 v[NL][4];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int H = a.lvl[l].H, W = a.lvl[l].W;
            tap[l] = msmv_make_tap(lx, ly, H, W, view_ok);
            const float *base = a.lvl[l].feat + (slab * (size_t)(H * W)) * 64 + l16 * 4;
            v[l][0] = ld4(base + (size_t)tap[l].o1 * 64);
            v[l][1] = ld4(base + (size_t)tap[l].o2 * 64);
            v[l][2] = ld4(base + (size_t)tap[l].o3 * 64);
            v[l][3] = ld4(base + (size_t)tap[l].o4 * 64);
        }

        // this row's 4 channels of grad_out for its point (inverse of the forward's store)
        float4 gq;
        if constexpr (PT == 1) {
            gq = ld4(a.gout + (size_t)pt * 64 + l16 * 4);
        } else if constexpr (PT == 2) {
            const bool odd = sub & 1;
            const float4 o = ld4(a.gout + ((size_t)(pt >> 1) * 64 + l16 * 4 + (odd ? 2 : 0)) * 2);
            const float s0 = odd ? o.x : o.y, s1 = odd ? o.z : o.w;
            const float r0 = __shfl_xor(s0, 16), r1 = __shfl_xor(s1, 16);
            gq = odd ? make_float4(r0, r1, o.y, o.w) : make_float4(o.x, o.z, r0, r1);
        } else if constexpr (PT == 4) {
            const bool b0 = sub & 1, b1 = sub & 2;
            const float4 o = ld4(a.gout + ((size_t)(pt >> 2) * 64 + l16 * 4 + sub) * 4);
            float sa = b0 ? o.x : o.y, sb = b0 ? o.z : o.w;
            float ra = __shfl_xor(sa, 16), rb = __shfl_xor(sb, 16);
            float4 m = b0 ? make_float4(ra, o.y, rb, o.w) : make_float4(o.x, ra, o.z, rb);
            sa = b1 ? m.x : m.z;
            sb = b1 ? m.y : m.w;
            ra = __shfl_xor(sa, 32);
            rb = __shfl_xor(sb, 32);
            gq = b1 ? make_float4(ra, rb, m.z, m.w) : make_float4(m.x, m.y, ra, rb);
        } else {
            const int bq = pt / a.P, p = pt - bq * a.P;
            const float *gp = a.gout + ((size_t)bq * 64 + l16 * 4) * a.P + p;
            gq = make_float4(gp[0], gp[(size_t)a.P], gp[(size_t)a.P * 2], gp[(size_t)a.P * 3]);
        }

        float gx = 0.f, gy = 0.f;
        float dwl[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const MsmvTap &t = tap[l];
            const float m1 = t.c1 ? 1.f : 0.f, m2 = t.c2 ? 1.f : 0.f;
            const float m3 = t.c3 ? 1.f : 0.f, m4 = t.c4 ? 1.f : 0.f;
            float sval = 0.f, sgw = 0.f, sgh = 0.f;
#define MSMV_GATHER_CH(X)                                                          \
            {                                                                      \
                const float v1 = m1 * v[l][0].X, v2 = m2 * v[l][1].X;              \
                const float v3 = m3 * v[l][2].X, v4 = m4 * v[l][3].X;              \
                const float val = t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4;   \
                sval += gq.X * val;                                                \
                sgw += gq.X * (t.hh * (v2 - v1) + t.lh * (v4 - v3));               \
                sgh += gq.X * (t.hw * (v3 - v1) + t.lw * (v4 - v2));               \
            }
            MSMV_GATHER_CH(x) MSMV_GATHER_CH(y) MSMV_GATHER_CH(z) MSMV_GATHER_CH(w)
#undef MSMV_GATHER_CH
            dwl[l] = msmv_row16_sum(sval);
            gx += (float)(a.lvl[l].W - 1) * wl[l] * sgw;
            gy += (float)(a.lvl[l].H - 1) * wl[l] * sgh;
        }
        gx = msmv_row16_sum(gx);
        gy = msmv_row16_sum(gy);
        if (act && l16 == 0) {
            float *dl = a.dloc + (size_t)pt * 3;
            dl[0] = gx; dl[1] = gy; dl[2] = 0.f;
#pragma unroll
            for (int l = 0; l < NL; ++l)
                a.dw[(size_t)pt * NL + l] = dwl[l];
        }
    }
}

template <int NL>
hipError_t launch_gather(const MsmvArgs &a, hipStream_t s)
{
    const int ngroups = (a.npts + 3) / 4;
    const int per_block = (GAT_THREADS / 64) * GAT_GROUPS_PER_WAVE;
    const int nblk = (ngroups + per_block - 1) / per_block;
    switch (a.P) {
    case 1: hipLaunchKernelGGL((msmv_bwd_gather_c64<NL, 1>), dim3(nblk), dim3(GAT_THREADS), 0, s, a); break;
    case 2: hipLaunchKernelGGL((msmv_bwd_gather_c64<NL, 2>), dim3(nblk), dim3(GAT_THREADS), 0, s, a); break;
    case 4: hipLaunchKernelGGL((msmv_bwd_gather_c64<NL, 4>), dim3(nblk), dim3(GAT_THREADS), 0, s, a); break;
    default: hipLaunchKernelGGL((msmv_bwd_gather_c64<NL, 0>), dim3(nblk), dim3(GAT_THREADS), 0, s, a); break;
    }
    return hipGetLastError();
}

}  // namespace

// ======================================================================================
// host launchers
// ======================================================================================
hipError_t msmv_launch_backward_atomic(const MsmvArgs &a, hipStream_t s, bool accumulate)
{
    if (!accumulate) {
        for (int l = 0; l < a.L; ++l) {
            const size_t bytes = (size_t)a.B * a.N * a.lvl[l].H * a.lvl[l].W * a.C * sizeof(float);
            if (bytes) {
                hipError_t e = hipMemsetAsync(a.lvl[l].dfeat, 0, bytes, s);
                if (e != hipSuccess)
                    return e;
            }
        }
    }
    if (a.npts == 0)
        return hipSuccess;
    // a few points per wave; enough waves to fill the chip many times over
    const int wpb = 4;
    int nblk = (a.npts + wpb * 4 - 1) / (wpb * 4);
    if (nblk < 1)
        nblk = 1;
    hipLaunchKernelGGL(msmv_bwd_atomic, dim3(nblk), dim3(64 * wpb), 0, s, a);
    return hipGetLastError();
}

bool msmv_backward_owner_supported(const MsmvArgs &a)
{
    return a.C == 64 && (a.L == 4 || a.L == 5);
}

size_t msmv_backward_owner_workspace(int B, int N, int Q, int P)
{
    return owner_ws_header_bytes(B, N) + (size_t)B * Q * P * sizeof(BinEntry);
}

hipError_t msmv_launch_backward_owner(const MsmvArgs &a, void *workspace, hipStream_t s)
{
    OwnerWs ws;
    ws.counts = reinterpret_cast<int *>(workspace);
    ws.cursors = ws.counts + (size_t)a.B * a.N;
    ws.entries = reinterpret_cast<BinEntry *>(reinterpret_cast<char *>(workspace) + owner_ws_header_bytes(a.B, a.N));

    hipError_t e = hipMemsetAsync(workspace, 0, owner_ws_header_bytes(a.B, a.N), s);
    if (e != hipSuccess)
        return e;
    if (a.npts > 0) {
        const int nblk = (a.npts + 255) / 256;
        hipLaunchKernelGGL((msmv_bin_points<false>), dim3(nblk), dim3(256), 0, s, a, ws);
        hipLaunchKernelGGL((msmv_bin_points<true>), dim3(nblk), dim3(256), 0, s, a, ws);
        if ((e = hipGetLastError()) != hipSuccess)
            return e;
    }

    OwnerGeom geo = {};
    int per_view = 0;
    for (int l = 0; l < a.L; ++l) {
        geo.tiles_per_level[l] = (a.lvl[l].H * a.lvl[l].W + TILE_PIX - 1) / TILE_PIX;
        per_view += geo.tiles_per_level[l];
    }
    geo.tiles_per_b = per_view * a.N;
    const long long ntiles = (long long)geo.tiles_per_b * a.B;
    if (ntiles > 0) {
        hipLaunchKernelGGL(msmv_bwd_owner_dfeat, dim3((unsigned)ntiles), dim3(OWN_THREADS), 0, s, a, ws, geo);
        if ((e = hipGetLastError()) != hipSuccess)
            return e;
    }
    if (a.npts > 0)
        return a.L == 4 ? launch_gather<4>(a, s) : launch_gather<5>(a, s);
    return hipSuccess;
}
