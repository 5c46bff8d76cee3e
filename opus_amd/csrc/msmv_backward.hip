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
//  * owner   - no atomics and no clear of d feat: points are first counting-sorted into bins
//              by (batch, view, y-band, x-band); every 4x8-pixel patch of a (batch, view,
//              level) slab is owned by one wave, which scans the bins that can reach it,
//              accumulates the contributions in fp32 MFMA registers and stores the patch
//              once (zeros included).  d loc / d w come from a separate gather kernel shaped
//              like the forward (row of 16 lanes per point).  C == 64, L in {4, 5}.
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "msmv_common.h"
#include "msmv_launch.h"

namespace {

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
#ifndef MSMV_STREAM_LEVELS
#define MSMV_STREAM_LEVELS 1        // the finest levels whose corner loads carry the streaming hint
#endif
#ifndef MSMV_GAT_STREAM_L0
#define MSMV_GAT_STREAM_L0 1
#endif
typedef float bwd_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_level(const float *p, bool stream)   // (constant after unrolling)
{
    if (stream && MSMV_GAT_STREAM_L0) {
        const bwd_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const bwd_f32x4 *>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    }
    return ld4(p);
}

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
        const float *grow = a.rows ? a.gout + (size_t)msmv_row_of(pt, a) * a.C : nullptr;
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
                    const float top = grow ? grow[c] : a.gout[((size_t)bq * a.C + c) * a.P + p];
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
struct BinEntry {       // 32 bytes: everything the d-feat pass needs about a point
    int pt;             // point index; in row mode the point's grad_out row instead
    float x, y;
    float w[5];         // the point's scale weights (L <= 5; unused slots are 0)
};

// (8, 32) measured best of (16,8) (16,16) (16,32) (32,8) (8,32): a patch is 8 pixels wide, so
// narrow x-bands cut the entries a wave scans in vain; every y-band is one more segment to walk
#ifndef MSMV_BIN_YB
#define MSMV_BIN_YB 8
#endif
#ifndef MSMV_BIN_XB
#define MSMV_BIN_XB 32
#endif
constexpr int BIN_YB = MSMV_BIN_YB;      // y-bands per (batch, view)
constexpr int BIN_XB = MSMV_BIN_XB;      // x-bands per y-band
constexpr int BIN_PER_VIEW = BIN_YB * BIN_XB;
constexpr int BIN_THREADS = 1024;
constexpr int BIN_MAX_KEYS = 4096;       // N * BIN_PER_VIEW must not exceed this (N <= 16)

constexpr int BIN_MAX_CHUNKS = 16;       // binning workgroups per batch slab, at most

// workspace layout:
//   [ table: B * chunks * (keys+1) ints ]   per binning workgroup: its points per key, and in the
//                                            last column its points without a valid view
//   [ offsets: B*(keys+1) ints ]  (padded to 256 B)   exclusive prefix of the counts per batch
//   [ entries: npts * 32 B ]
// entries of batch b occupy [b*QP, b*QP + offsets[b][keys]), sorted by
// key = (view*BIN_YB + yband)*BIN_XB + xband (inside a key: by binning workgroup, i.e. by blocks of
// consecutive points).  The slab's other slots, [b*QP + offsets[b][keys], (b+1)*QP), hold its points
// without a valid view (in no bin, view -1 in the entry): the d-feat pass, which walks bins, never
// meets them; the passes that walk all QP slots of a slab (gather, bin-ordered forward) write their
// all-zero results.  The sort thus reads loc and weights only, and one sort serves forward and
// backward passes alike.  Nothing in the workspace needs clearing: every word that is read was
// written by the same sort.
struct OwnerWs {
    int *table;
    int *offsets;
    BinEntry *entries;
    int keys;             // N * BIN_PER_VIEW
    int chunks;           // binning workgroups per batch slab (<= BIN_MAX_CHUNKS)
    int chunk_pts;        // points per binning workgroup
};

// Slabs of more than BIN_MAX_CHUNKS * BIN_THREADS points give each workgroup several rounds of
// BIN_THREADS points instead of more workgroups: the table stays small against the entries.
__host__ __device__ inline int owner_ws_chunks(int qp)
{
    const int c = (qp + BIN_THREADS - 1) / BIN_THREADS;
    return c < 1 ? 1 : (c > BIN_MAX_CHUNKS ? BIN_MAX_CHUNKS : c);
}

__host__ __device__ inline size_t owner_ws_header_bytes(int B, int N, int qp)
{
    const size_t keys1 = (size_t)N * BIN_PER_VIEW + 1;
    const size_t h = ((size_t)B * owner_ws_chunks(qp) * keys1 + (size_t)B * keys1) * sizeof(int);
    return (h + 255) & ~(size_t)255;
}

// Band of a normalised coordinate: monotone in the coordinate, so a pixel interval maps to
// a band interval.  Coordinates outside [0,1) go to the first / last band; NaN goes to 0.
template <int NB>
__device__ __forceinline__ int msmv_band_of(float y)
{
    const float f = y * (float)NB;
    return f >= (float)(NB - 1) ? NB - 1 : (f > 0.f ? (int)f : 0);
}

// Conservative band interval of the locations whose 2x2 footprint can reach pixels
// lo..hi (inclusive) of an axis with `size` pixels: floor(v*(size-1)) is lo-1..hi, i.e.
// v*(size-1) in [lo-1, hi+1), so the band floor(v*NB) lies between floor(NB*(lo-1)/(size-1))
// and floor(NB*(hi+1)/(size-1)).  Integer arithmetic only (the caller is wave-uniform: this
// runs on the scalar unit; `sm1` divides by size-1).  The rounding of the two float products
// (location -> pixel, location -> band) is below 1e-6 bands, which only matters when a bound is
// an exact integer (the lower one then steps down a band) - or on axes of thousands of pixels,
// where the interval is simply widened by a band on each side.  size == 1 maps every location
// to pixel 0.
template <int NB>
__device__ __forceinline__ void msmv_band_range(int lo, int hi, int size, MsmvDiv sm1, int &b_lo, int &b_hi)
{
    b_lo = 0;
    b_hi = NB - 1;
    if (size > 1) {
        const int den = size - 1;
        const int wide = den >= 4096 ? 1 : 0;
        const int tl = NB * (lo - 1);
        if (tl > 0) {
            const int q = (int)msmv_div((uint32_t)tl, sm1);
            const int lower = (tl - q * den == 0 || wide) ? q - 1 : q;
            b_lo = min(max(lower, 0), NB - 1);
        }
        const int qh = (int)msmv_div((uint32_t)(NB * (hi + 1)), sm1) + wide;
        b_hi = min(qh, NB - 1);
    }
}

// The scale weights of point pt, padded to the entry's five slots.
struct BinWeights {
    float w[5];
};

__device__ __forceinline__ BinWeights msmv_load_weights(const MsmvArgs &a, size_t pt)
{
    const float *wp = a.w + pt * a.L;
    BinWeights r;
#pragma unroll
    for (int k = 0; k < 5; ++k)
        r.w[k] = k < a.L ? wp[k] : 0.f;
    return r;
}

// The entry of point pt (location x, y, weights, bin key or -1) at slot `pos` of its slab.
__device__ __forceinline__ void msmv_put_entry(const MsmvArgs &a, const OwnerWs &ws, int b, int pos, size_t pt,
                                               float x, float y, BinWeights wv, int key)
{
    if (a.L <= 4)
        wv.w[4] = __int_as_float(key < 0 ? -1 : key / BIN_PER_VIEW);   // the view, for the passes that walk the slots
    float4 *e = reinterpret_cast<float4 *>(ws.entries + (size_t)b * a.qp + pos);
    const int id = a.rows ? msmv_row_of((int)pt, a) : (int)pt;
    e[0] = make_float4(__int_as_float(id), x, y, wv.w[0]);
    e[1] = make_float4(wv.w[1], wv.w[2], wv.w[3], wv.w[4]);
}

__device__ __forceinline__ int msmv_bin_key(const float *lp, int N)
{
    const int view = msmv_view_of(lp[2], N);
    return view < 0 ? -1 : (view * BIN_YB + msmv_band_of<BIN_YB>(lp[1])) * BIN_XB + msmv_band_of<BIN_XB>(lp[0]);
}

// In-place inclusive prefix sum of v[0..keys) (keys <= BIN_MAX_KEYS) by a workgroup of
// BIN_THREADS threads: a serial run of 4 keys per thread, a shuffle scan across the wave, the 16
// wave totals through LDS - two barriers (a Hillis-Steele scan over the keys took 22).  Ends
// with a barrier; v must be complete (and a barrier passed) on entry.
__device__ __forceinline__ void msmv_block_scan(int *v, int keys, int *wave_tot)
{
    constexpr int PER = BIN_MAX_KEYS / BIN_THREADS;
    static_assert(PER * BIN_THREADS == BIN_MAX_KEYS && BIN_THREADS % 64 == 0, "every key has a thread");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int x[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = tid * PER + i;
        sum += k < keys ? v[k] : 0;
        x[i] = sum;
    }
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(inc, d);
        if (lane >= d)
            inc += up;
    }
    if (lane == 63)
        wave_tot[wave] = inc;
    __syncthreads();
    int before = inc - sum;                       // keys of earlier threads of this wave
    for (int w = 0; w < wave; ++w)
        before += wave_tot[w];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = tid * PER + i;
        if (k < keys)
            v[k] = before + x[i];
    }
    __syncthreads();
}

// Counting sort of the points of every batch slab by key, in two short launches; a slab is cut
// into `chunks` runs of consecutive points, one workgroup each.  No global atomics and nothing to
// clear: pass 1 writes every workgroup's histogram to the table, pass 2 reads the slab's rows of it.
// Pass 1: per-workgroup LDS histogram -> table[b][chunk][0..keys], last column = points in no bin.
__global__ void __launch_bounds__(BIN_THREADS)
msmv_bin_count(const MsmvArgs a, OwnerWs ws)
{
    __shared__ int hist[BIN_MAX_KEYS + 1];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / ws.chunks, chunk = blockIdx.x % ws.chunks;
    const int keys = ws.keys;
    for (int k = tid; k <= keys; k += BIN_THREADS)
        hist[k] = 0;
    __syncthreads();
    const int lo = chunk * ws.chunk_pts, hi = min(lo + ws.chunk_pts, a.qp);
    for (int i = lo + tid; i < hi; i += BIN_THREADS) {
        const int key = msmv_bin_key(a.loc + ((size_t)b * a.qp + i) * 3, a.N);
        atomicAdd(&hist[key >= 0 ? key : keys], 1);
    }
    __syncthreads();
    int *row = ws.table + ((size_t)b * ws.chunks + chunk) * (keys + 1);
    for (int k = tid; k <= keys; k += BIN_THREADS)
        row[k] = hist[k];
}

// Pass 2: every workgroup sums the slab's table rows - all of them for the slab's counts per key,
// those of the workgroups before it for where its own points of a key begin - scans the counts,
// and places its points (LDS cursors).
__global__ void __launch_bounds__(BIN_THREADS)
msmv_bin_fill(const MsmvArgs a, OwnerWs ws)
{
    __shared__ int base[BIN_MAX_KEYS + 1];    // points of the key placed by earlier workgroups, then the next free slot
    __shared__ int incl[BIN_MAX_KEYS];        // the slab's counts, then their inclusive prefix
    __shared__ int wave_tot[BIN_THREADS / 64];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / ws.chunks, chunk = blockIdx.x % ws.chunks;
    const int keys = ws.keys;

    // this workgroup's points: location loads in flight while the table is summed
    const int lo = chunk * ws.chunk_pts, hi = min(lo + ws.chunk_pts, a.qp);
    const int i0 = lo + tid;
    float x = 0.f, y = 0.f, v = 0.f;
    BinWeights wv = {};
    if (i0 < hi) {
        const float *lp = a.loc + ((size_t)b * a.qp + i0) * 3;
        x = lp[0]; y = lp[1]; v = lp[2];
        wv = msmv_load_weights(a, (size_t)b * a.qp + i0);
    }
    const int *rows = ws.table + (size_t)b * ws.chunks * (keys + 1);
    for (int k = tid; k <= keys; k += BIN_THREADS) {
        int all = 0, before = 0;
        for (int c = 0; c < ws.chunks; ++c) {
            const int n = rows[(size_t)c * (keys + 1) + k];
            all += n;
            before += c < chunk ? n : 0;
        }
        base[k] = before;
        if (k < keys)
            incl[k] = all;
    }
    __syncthreads();
    msmv_block_scan(incl, keys, wave_tot);
    for (int k = tid; k < keys; k += BIN_THREADS) {
        const int excl = k ? incl[k - 1] : 0;
        base[k] += excl;
        if (chunk == 0) {
            int *off = ws.offsets + (size_t)b * (keys + 1);
            off[k] = excl;
            if (k == keys - 1)
                off[keys] = incl[k];
        }
    }
    if (chunk == 0 && keys == 0 && tid == 0)
        ws.offsets[(size_t)b] = 0;
    __syncthreads();
    for (int i = i0; i < hi; i += BIN_THREADS) {
        const size_t pt = (size_t)b * a.qp + i;
        if (i != i0) {
            const float *lp = a.loc + pt * 3;
            x = lp[0]; y = lp[1]; v = lp[2];
            wv = msmv_load_weights(a, pt);
        }
        const float l3[3] = {x, y, v};
        const int key = msmv_bin_key(l3, a.N);
        const int r = atomicAdd(&base[key >= 0 ? key : keys], 1);
        // (points in no bin fill the slab's slots from the last one down)
        msmv_put_entry(a, ws, b, key >= 0 ? r : a.qp - 1 - r, pt, x, y, wv, key);
    }
}

// Small slabs (at most BIN_ONE_PPT * BIN_THREADS points, e.g. OPUS-T / S): the whole counting
// sort of a slab in ONE workgroup and one launch - histogram, scan and placement in LDS, no
// global counters, nothing to clear beforehand.  Three launches' worth of latency is most of
// the two-pass sort's time at these sizes.
#ifndef MSMV_BIN_ONE_PPT
#define MSMV_BIN_ONE_PPT 3        // 10 (slabs up to 10240 points in one workgroup) measured 49 vs 26 us at OPUS-L
#endif
constexpr int BIN_ONE_PPT = MSMV_BIN_ONE_PPT;
__global__ void __launch_bounds__(BIN_THREADS)
msmv_bin_one(const MsmvArgs a, OwnerWs ws)
{
    __shared__ int hist[BIN_MAX_KEYS];
    __shared__ int incl[BIN_MAX_KEYS];
    __shared__ int wave_tot[BIN_THREADS / 64];
    __shared__ int tail_n;                    // points without a valid view
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int keys = ws.keys;
    for (int k = tid; k < keys; k += BIN_THREADS)
        hist[k] = 0;
    if (tid == 0)
        tail_n = 0;
    __syncthreads();
    int key[BIN_ONE_PPT], rank[BIN_ONE_PPT];
    float x[BIN_ONE_PPT], y[BIN_ONE_PPT];
#pragma unroll
    for (int u = 0; u < BIN_ONE_PPT; ++u) {
        const int i = u * BIN_THREADS + tid;
        key[u] = -1;
        rank[u] = 0;
        x[u] = y[u] = 0.f;
        if (i < a.qp) {
            const float *lp = a.loc + ((size_t)b * a.qp + i) * 3;
            x[u] = lp[0];
            y[u] = lp[1];
            key[u] = msmv_bin_key(lp, a.N);
            rank[u] = atomicAdd(key[u] >= 0 ? &hist[key[u]] : &tail_n, 1);
        }
    }
    __syncthreads();
    for (int k = tid; k < keys; k += BIN_THREADS)
        incl[k] = hist[k];
    __syncthreads();
    msmv_block_scan(incl, keys, wave_tot);
    int *off = ws.offsets + (size_t)b * (keys + 1);
    for (int k = tid; k < keys; k += BIN_THREADS) {
        const int excl = incl[k] - hist[k];
        off[k] = excl;
        if (k == keys - 1)
            off[keys] = incl[k];
    }
#pragma unroll
    for (int u = 0; u < BIN_ONE_PPT; ++u) {
        const int i = u * BIN_THREADS + tid;
        const size_t pt = (size_t)b * a.qp + i;
        if (i < a.qp) {
            const int pos = key[u] >= 0 ? incl[key[u]] - hist[key[u]] + rank[u] : a.qp - 1 - rank[u];
            msmv_put_entry(a, ws, b, pos, pt, x[u], y[u], msmv_load_weights(a, pt), key[u]);
        }
    }
}

// ======================================================================================
// owner algorithm: d feat tiles
// ======================================================================================
// d feat of a patch of PATCH_H x PATCH_W = 32 pixels of one (batch, view, level) slab is
//     D[pixel i][channel c] = sum over hits h of  W[i][h] * G[h][c],
// W[i][h] = corner weight * scale weight of the corner of point h that lands on pixel i (at
// most four non-zeros per column), G[h] = the point's 64-channel gradient vector.  Each
// WAVE owns one patch and keeps D in registers as two 32x32 fp32 MFMA accumulators
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, K = 2 hits per instruction; tile 0 holds
// channels 0..31, tile 1 channels 32..63, lane = channel within the tile, so that a half-wave
// stores 128 contiguous bytes of one pixel per instruction straight from an accumulator
// register, with a wave-uniform base and an immediate offset - no address arithmetic).
// The dense product does 8x the multiply-adds the sparse one needs, but it runs on the
// otherwise idle matrix pipe: the earlier versions of this kernel, which applied every hit with
// LDS read-modify-writes, were bound by VALU / scalar / LDS instruction issue (45 CU cycles per
// hit), and LDS float atomics (ds_add_f32) cost ~190 LDS cycles per wave instruction.
// No pixel is shared between waves, nothing is cleared, every d-feat byte is stored once.
// The patch is 4 rows x 8 columns rather than 32 pixels of one row because a point's 2x2
// footprint then falls into 1.4 patches on average instead of 2.1: a third fewer hits.
//
// That is hit loop 0 of the kernel (template parameter HL).  Hit loops 1 and 2 work on the same
// patch as two halves of 4 x 4 pixels with 16 x 16 x 4 tiles (v_mfma_f32_16x16x4_f32: four hits per
// instruction, 4 of 16 pixels useful instead of 4 of 32), which halves the matrix-pipe time of a hit
// - the pass is paid for per hit, and the waves of a SIMD take turns at its one matrix pipe - and
// lets a lane load its 16 bytes of a gradient row straight into the B operand's registers and store
// 16 contiguous bytes of d feat per instruction; see the comment in front of the kernel.  They take
// contiguous gradient rows (row mode, P = 1) and the op's layout with P = 2; everything else
// (P = 4, other P, ACC) and - by measurement - several sets in the op's layout run hit loop 0.
//
// A wave scans only the bins whose bands can reach its rows / columns; lanes that find a
// hit (a point with a corner in the patch) write a record into a small per-wave LDS queue;
// the wave then consumes the queue two records at a time (one per half-wave).  The kernel is
// bound by instruction issue, so the record is laid out for the consumer: bilinear weights are
// separable, W[pixel (r, c)] = wy[r] * wx[c], and a record holds wy for the patch's 4 rows and
// wx for its 8 columns as two small zero-filled tables into which the scanning lane drops
// (hh * w_l, lh * w_l) at the footprint's rows and (hw, lw) at its columns (one guard slot on
// each side takes what falls outside the patch).  The consuming lane, which stands for one
// pixel, reads its row's and its column's entry and multiplies: one VALU instruction per pair
// of hits where four compare / select pairs used to pick the corner that lands on the pixel.
// Corners outside the image need no test: their rows / columns belong to no stored pixel.
//
// A wave applies its hits one after the other with a few units of gradient rows in flight (a
// queue round is straight-line code, see the kernel), so its run time grows with the hits of its patch: a patch of a coarse level (or of a
// crowded image region) collects hundreds of hits where a fine one collects a handful, and
// those few long waves would finish last.  A level whose patches expect many hits is
// therefore worked on by K = 2 or 4 waves per patch: wave k of the K takes entries
// k, k + K, k + 2K, ... of the patch's bins, and the K partial accumulators are summed through
// LDS (log2 K rounds) before the store.  A workgroup is always OWN_WAVES waves = OWN_WAVES / K patches.
constexpr int PATCH_H = 4, PATCH_W = 8;
static_assert(PATCH_H * PATCH_W == 32, "a patch is the M = 32 dimension of the MFMA tile");
#ifndef MSMV_OWN_WAVES
#define MSMV_OWN_WAVES 4
#endif
// waves per workgroup (a power of two); at most all of them share one patch.  With 72 VGPRs a SIMD
// holds 7 waves: workgroups of 4 waves fill them (7 per CU), workgroups of 8 only 6 of 7
// (measured 183 vs 198 us at OPUS-L; 2 waves: 175 us uniform but no better on crowded images)
constexpr int OWN_WAVES = MSMV_OWN_WAVES;
constexpr int OWN_THREADS = OWN_WAVES * 64;
// Workgroup order of the d-feat pass: views per group (0 = all N: level-major; 1: view-major).  Measured
// at OPUS-L: one set 159.5 / 165.0 us level- / view-major, six sets 621 / 686 us, six OPUS-M sets 353 /
// 404 us - view-major mixes the long waves of a view's coarse levels with the store-bound ones of its
// neighbour's finest level, and two views' rows still exceed the L2.
#ifndef MSMV_DFEAT_VPU_SINGLE
#define MSMV_DFEAT_VPU_SINGLE 0
#endif
#ifndef MSMV_DFEAT_VPU_MULTI
#define MSMV_DFEAT_VPU_MULTI 0
#endif
#ifndef MSMV_OWN_SPLIT_HITS
#define MSMV_OWN_SPLIT_HITS 64                   // expected hits per wave above which a level's patches are shared
#endif
#ifndef MSMV_OWN_SPLIT_HITS_MULTI
#define MSMV_OWN_SPLIT_HITS_MULTI 128            // the same for a pass over several point sets: per-patch set-up is spread over
#endif                                           // more hits there (six OPUS-L layers: 64 -> 795 us, 256 -> 764, 1024 -> 768 but 1017 on crowded images)
constexpr int OWN_QCAP = 16;             // hit records a wave queues before consuming them

// A queued hit record, in dwords:
//   [0] point index (or grad_out row)     [1] point set (MULTI)
//   [2..7]  wy for patch rows -1..4       (rows 0..3 are real, -1 and 4 are guards)
//   [8..17] wx for patch columns -1..8    (columns 0..7 are real)
//   [18..19] padding to a multiple of 16 bytes
constexpr int REC_DW = 20;
constexpr int REC_Y0 = 3, REC_X0 = 9;    // dword of row 0 / column 0

// Geometry of one level, packed into 64 bytes so that a wave fetches it with one wave-uniform load
// once it knows its level.
struct OwnerLevel {
    float *dfeat;
    int H, W;
    int tiles;                  // workgroups per (batch, view) slab of the level
    int patch_cols;             // patches per row of patches: ceil(W / PATCH_W)
    int patches;                // patches per slab: ceil(H / PATCH_H) * patch_cols
    int split_log2;             // log2 K: waves per patch at the level (K in {1,2,4,8})
    MsmvDiv div_tiles, div_cols, div_hm1, div_wm1;   // divisors: tiles, patch_cols, H - 1, W - 1
};
static_assert(sizeof(OwnerLevel) == 64, "one 64-byte scalar load per level");

struct OwnerGeom {
    OwnerLevel lv[MSMV_MAX_LEVELS];
    // workgroups of a batch slab are ordered [view group][level, coarse first][view][tile], a group
    // being vpu views (N: level-major, 1: view-major): seg_end[k] is the end of the k-th level in that
    // order (level L-1-k) within a group; INT_MAX from the last level on
    int seg_end[MSMV_MAX_LEVELS];
    MsmvDiv div_tiles_per_b;
    int tiles_per_b;            // N * sum over levels of tiles
    int L, N, qp, P;
    int vpu;                    // views per group: a divisor of N
    int tiles_per_group;        // vpu * sum over levels of tiles
    MsmvDiv div_tiles_per_group;
};

// The sorted point sets one d-feat launch accumulates: normally one (this call's), several when
// a stack of layers that sample the same feature maps defers its d feats to one pass - d feat
// is then written once instead of once per layer.  All sets share the sizes in OwnerGeom.
constexpr int OWN_MAX_SETS = 8;
struct OwnerSets {
    const float *gout[OWN_MAX_SETS];        // gradient of the op's output, per set
    const int *offsets[OWN_MAX_SETS];       // OwnerWs::offsets of the set's sort
    const BinEntry *entries[OWN_MAX_SETS];  // OwnerWs::entries
    int n;
};

typedef float msmv_f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void *msmv_gptr_t;
typedef __attribute__((address_space(3))) void *msmv_lptr_t;
#ifndef MSMV_OWN_NT_STORE
#define MSMV_OWN_NT_STORE 1
#endif
#ifndef MSMV_OWN_ROWS_NT_MULTI
#define MSMV_OWN_ROWS_NT_MULTI 0                 // several point sets: gradient rows copied with the streaming hint
#endif
#ifndef MSMV_OWN_STAGE_F4
#define MSMV_OWN_STAGE_F4 256                    // 4 KiB per wave: with the queue, LDS never limits the occupancy
#endif
#ifndef MSMV_OWN_OCC
#define MSMV_OWN_OCC (MULTI || ACC || HL != 0 ? 4 : 7), 8  // the plain single-set kernel fits 7 waves per SIMD (72 VGPRs)
#endif

// d feat is written once and not read again by this pass: a streaming (non-temporal) store
// keeps it from displacing the feature maps and gradients that the following kernels read.
__device__ __forceinline__ void msmv_store_stream(float *p, float x)
{
#if MSMV_OWN_NT_STORE
    __builtin_nontemporal_store(x, p);
#else
    *p = x;
#endif
}

__device__ __forceinline__ float msmv_load_stream(const float *p)
{
#if MSMV_OWN_NT_STORE
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// global -> LDS copy of 16 bytes per lane (lane i's land at dst + 16 i) without a register
// destination, as an opaque instruction: the compiler's wait-count pass, which knows nothing of the
// copies in flight, then leaves the waiting to the counted waits written here (it would put a full
// wait in front of every LDS read that follows such a copy in straight-line code).
// (M0 carries the LDS address of such a copy; the compiler, which warns about it being named as
// clobbered, keeps nothing in it in these kernels.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <bool STREAM = false>
__device__ __forceinline__ void msmv_load_to_lds_b128(const float *src, const float4 *dst)
{
    const unsigned lds = (unsigned)(uintptr_t)(msmv_lptr_t)dst;
    if constexpr (STREAM)         // streaming hint: the copy does not displace what the L2 holds
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt"
                     :: "v"(src), "s"(lds) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                     :: "v"(src), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

template <int N> __device__ __forceinline__ void msmv_wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int... I, class F>
__device__ __forceinline__ void msmv_static_for_impl(std::integer_sequence<int, I...>, F &&f)
{
    (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant
template <int N, class F> __device__ __forceinline__ void msmv_static_for(F &&f)
{
    msmv_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// Inclusive prefix sum over the first LANES lanes of a wave (LANES = 8: within one DPP row;
// LANES = 64: the whole wave) with DPP shifts: no LDS round trips.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int msmv_dpp_shift_add(int v)
{
    // lanes without a source (and rows outside ROW_MASK) add the `old` operand, 0
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
}

template <int LANES>
__device__ __forceinline__ int msmv_wave_incl_scan(int v)
{
    static_assert(LANES == 8 || LANES == 64, "a quarter row or the wave");
    v = msmv_dpp_shift_add<0x111, 0xF>(v);      // row_shr:1
    v = msmv_dpp_shift_add<0x112, 0xF>(v);      // row_shr:2
    v = msmv_dpp_shift_add<0x114, 0xF>(v);      // row_shr:4
    if constexpr (LANES == 64) {
        v = msmv_dpp_shift_add<0x118, 0xF>(v);  // row_shr:8
        v = msmv_dpp_shift_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
        v = msmv_dpp_shift_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
    }
    return v;
}

// What one lane knows about the bin entry it scanned (one point against this wave's patch).
struct OwnHit {
    int pt;
    bool hit;                        // the footprint's rows and columns both reach the patch
    int dr0, dc0;                    // row / column of corner 1 relative to the patch, -1 .. 3 / -1 .. 7 for a hit
    float y0, y1, x0, x1;            // hh * w_l, lh * w_l, hw, lw
};

// ACC: add the patch to what d feat already holds instead of overwriting it (one decoder
// layer after another accumulating into one buffer); a patch that no point reaches is then
// left alone.
// MULTI: more than one point set.  The segments of all sets (one per set and y-band, at most
// OWN_MAX_SETS * BIN_YB = 64: one per lane) are then walked as ONE concatenation, so that a
// patch still costs one chain of dependent loads (offsets -> entries -> gradient rows) and not
// one per set; a hit record carries its set, which selects the gradient array.  The tables
// this needs cost registers, hence occupancy: the single-set kernel is compiled without them.
#ifdef MSMV_DBG_TIMES
// development aid: per-wave phase timestamps (100 MHz constant clock) of the d-feat kernel
__device__ unsigned long long *msmv_dbg_buf;
#ifdef MSMV_DBG_HDR
// header study (tools/hdr_probe.py): stamps 1, 2 = level known, level geometry in; 3, 4 = bins
// known, first entries in
#define MSMV_DBG_STAMP(K) if ((K) == 1) dbg_t[3] = __builtin_amdgcn_s_memrealtime(); else if ((K) == 2) dbg_t[4] = __builtin_amdgcn_s_memrealtime(); else if ((K) == 5) dbg_t[5] = __builtin_amdgcn_s_memrealtime();
#define MSMV_DBG_HSTAMP(K) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dbg_t[K] = __builtin_amdgcn_s_memrealtime(); }
#else
#define MSMV_DBG_STAMP(K) if (lane == 0) dbg_t[K] = __builtin_amdgcn_s_memrealtime();
#define MSMV_DBG_HSTAMP(K)
#endif
#else
#define MSMV_DBG_STAMP(K)
#define MSMV_DBG_HSTAMP(K)
#endif

// Hit loops 1 and 2 (HL != 0): 16 x 16 x 4 matrix tiles (v_mfma_f32_16x16x4_f32, 32 matrix-pipe cycles
// for four hits x 16 pixels x 16 channels) instead of 32 x 32 x 2 ones.  The patch is worked on as two
// HALVES of 4 x 4 pixels (columns 0..3 / 4..7): a hit is queued for the half - or the two halves - its
// footprint reaches (10 / 9 halves per hit), and a group of four queued hits costs four matrix
// instructions, one per tile of 16 channel quadruples: 35 matrix-pipe cycles per hit instead of 64.
// Lane (n, k) = (lane % 16, lane / 16) supplies for hit k of a group the weight of half-pixel n
// (A operand) and channels 4n .. 4n+3 of the hit's gradient row (B operands of the four tiles: tile
// t holds channel 4n + t in column n) - 16 contiguous bytes, loaded straight into registers, the
// row's 256 bytes by 16 neighbouring lanes, no staging in LDS.  Accumulator register i of tile t
// in lane (n, r) is pixel (row r, column i) of the half, channel 4n + t: the four tiles' registers
// i form 16 contiguous bytes of d feat, and one store instruction writes a column of the half -
// 4 rows x 256 contiguous bytes.  The wave's queue (64 records of 64 bytes: the row's address and
// the separable weight tables for the half) lives in what hit loop 0 uses as its staging area.
// HL (hit loop): 0 = 32 x 32 x 2 tiles over whole patches; 1 = 16 x 16 x 4 tiles over half patches, the
// rows of two groups of four hits requested together, then both groups applied; 2 = the same tiles, one
// group at a time with the next group's rows requested before this group is applied.
template <int PT, bool ACC, bool MULTI, int HL = 0>
__global__ void __launch_bounds__(OWN_THREADS) __attribute__((amdgpu_waves_per_eu(MSMV_OWN_OCC)))
msmv_bwd_owner_dfeat(const OwnerGeom geo, const OwnerSets sets)
{
    constexpr bool H16 = HL != 0;
    static_assert(HL >= 0 && HL <= 2, "hit loop kind");
    static_assert(!H16 || ((PT == 1 || PT == 2) && !ACC), "the 16 x 16 x 4 hit loop: contiguous gradient rows, plain stores");
#ifdef MSMV_DBG_TIMES
    unsigned long long dbg_t[6] = {};
    dbg_t[0] = __builtin_amdgcn_s_memrealtime();
    unsigned dbg_hw, dbg_xcc;            // where the wave runs: HW_ID (wave, SIMD, CU, SH, SE) and the XCC
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(dbg_hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(dbg_xcc));
    const unsigned long long dbg_where = ((unsigned long long)((dbg_xcc & 0xF) << 16 | (dbg_hw & 0xFFFF))) << 40;
#endif
    // STAGED (P in {1,2,4} and row mode): the gradient rows of a whole queue of hits are copied
    // global -> LDS by direct loads (no VGPR destination), all in flight at once, and read from
    // there when the hits are applied.  P-strided rows (other P) are gathered into registers.
    constexpr bool STAGED = PT != 0;
    constexpr int ROW_F4 = PT == 4 ? 64 : PT == 2 ? 32 : 16;   // float4 per staged gradient row
    constexpr int STAGE_F4 = MSMV_OWN_STAGE_F4;                // staging area per wave, in float4
    constexpr int QC = OWN_QCAP;                               // records per queue round
    static_assert(QC >= 2 && QC % 2 == 0 && QC <= OWN_QCAP, "records are consumed in pairs");
    static_assert(REC_DW % 4 == 0, "records are cleared with 16-byte writes");
    __shared__ float4 queue_all[OWN_WAVES][OWN_QCAP * REC_DW / 4];
    // staging areas; after the hit loop the same memory carries partial tiles between waves
    __shared__ float4 stage_all[OWN_WAVES][STAGE_F4];
    static_assert(sizeof(stage_all) >= (OWN_WAVES / 2) * 32 * 64 * sizeof(float), "partial tiles fit");
    float (*red)[32 * 64] = reinterpret_cast<float (*)[32 * 64]>(&stage_all[0][0]);

    const int lane = threadIdx.x & 63;
    const int l32 = lane & 31;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 *queue4 = queue_all[wave];
    float *queue = reinterpret_cast<float *>(queue4);
    [[maybe_unused]] float4 *stage = stage_all[wave];
    // MULTI: the sets' gradient arrays, and per wave the concatenation's segment table
    // (entries before the segment, address of its first entry)
    __shared__ const float *gout_tab[MULTI ? OWN_MAX_SETS : 1];
    __shared__ int excl_tab_all[MULTI ? OWN_WAVES : 1][64];
    __shared__ unsigned long long addr_tab_all[MULTI ? OWN_WAVES : 1][64];
    [[maybe_unused]] int *excl_tab = excl_tab_all[MULTI ? wave : 0];
    [[maybe_unused]] unsigned long long *addr_tab = addr_tab_all[MULTI ? wave : 0];
    if constexpr (MULTI) {
        static_assert(OWN_MAX_SETS * BIN_YB <= 64, "one lane per (set, y-band) segment");
        if (threadIdx.x < OWN_MAX_SETS) {
            // static indices only: a lane-indexed read would copy the argument block to scratch
            const float *g = sets.gout[0];
#pragma unroll
            for (int i = 1; i < OWN_MAX_SETS; ++i)
                g = (int)threadIdx.x == i ? sets.gout[i] : g;
            gout_tab[threadIdx.x] = g;
        }
        __syncthreads();
    }

    // ---- which patch: logical order [b][view group][level, coarse first][view][tile][wave / K].  A
    // group is geo.vpu views: all N of them (level-major: one phase per level and slab), one (view-major:
    // a view's gradient rows and bin entries are then what an XCD's L2 has to hold from one level to
    // the next - with several point sets a whole slab's no longer fit) or a divisor of N in between
    int id = msmv_xcd_remap(blockIdx.x, gridDim.x);
    const int b = (int)msmv_div((uint32_t)id, geo.div_tiles_per_b);
    id -= b * geo.tiles_per_b;
    const int vgroup = (int)msmv_div((uint32_t)id, geo.div_tiles_per_group);
    id -= vgroup * geo.tiles_per_group;
    int k = 0, seg_start = 0;
#pragma unroll
    for (int j = 0; j < MSMV_MAX_LEVELS - 1; ++j) {          // (seg_end is INT_MAX from the last level on)
        const bool past = id >= geo.seg_end[j];
        k = past ? j + 1 : k;
        seg_start = past ? geo.seg_end[j] : seg_start;
    }
    const int l = geo.L - 1 - k;
    MSMV_DBG_HSTAMP(1)
#ifdef MSMV_ABL_LEVELS
    if (!((MSMV_ABL_LEVELS >> l) & 1))       // development aid: only the levels of the mask are worked on
        return;
#endif
    id -= seg_start;
    const OwnerLevel G = geo.lv[l];
    const int N = geo.N;
    const int view_in_group = (int)msmv_div((uint32_t)id, G.div_tiles);
    const int view = vgroup * geo.vpu + view_in_group;
    const int tile_idx = id - view_in_group * G.tiles;
    const int klog = G.split_log2;
    const int K = 1 << klog;                                 // waves sharing a patch
    const int sub = wave & (K - 1);                          // this wave's share: entries sub, sub + K, ...
    const int patch_raw = tile_idx * (OWN_WAVES >> klog) + (wave >> klog);
    const bool idle = patch_raw >= G.patches;                // past the slab's last patch: takes part in
    const int patch = idle ? 0 : patch_raw;                  // the barriers below, stores nothing
    const int H = G.H, W = G.W;
    const int S = H * W;
    const int prow = (int)msmv_div((uint32_t)patch, G.div_cols);
    const int r0 = prow * PATCH_H, c0 = (patch - prow * G.patch_cols) * PATCH_W;
    const int nrow = min(PATCH_H, H - r0), ncol = min(PATCH_W, W - c0);
    MSMV_DBG_HSTAMP(2)

    [[maybe_unused]] msmv_f32x16 acc0 = {}, acc1 = {};       // channels 0..31 / 32..63
    typedef float msmv_f32x4v __attribute__((ext_vector_type(4)));
    [[maybe_unused]] msmv_f32x4v hacc[2][4] = {};            // H16: [half][tile]
    [[maybe_unused]] bool any_hit = false;
    [[maybe_unused]] bool touched = false;                   // ACC: pixel l32 of this half's records got a weight

    // ---- the bins that can reach this patch: one segment of the sorted entries per y-band
    // (all of them adjacent, i.e. one segment, when every x-band is wanted)
    int yb_lo, yb_hi, xb_lo, xb_hi;
    msmv_band_range<BIN_YB>(r0, r0 + nrow - 1, H, G.div_hm1, yb_lo, yb_hi);
    msmv_band_range<BIN_XB>(c0, c0 + ncol - 1, W, G.div_wm1, xb_lo, xb_hi);
    const bool all_x = xb_lo == 0 && xb_hi == BIN_XB - 1;
    const int nseg = all_x ? 1 : yb_hi - yb_lo + 1;          // <= BIN_YB
    [[maybe_unused]] const float *gout = sets.gout[0];
    const size_t off_slab = (size_t)b * (N * BIN_PER_VIEW + 1) + view * BIN_PER_VIEW;
    [[maybe_unused]] const float4 *list = reinterpret_cast<const float4 *>(sets.entries[0] + (size_t)b * geo.qp);

    // Lane s holds segment s (MULTI: segment s % BIN_YB of set s / BIN_YB): its length, the
    // count of entries before it (excl) and delta = begin - excl, so that virtual entry v of
    // the concatenation is list[v + delta] (MULTI: the address of its first entry instead).
    int seg_len = 0, seg_begin = 0;
    [[maybe_unused]] unsigned long long seg_addr = 0;
    if constexpr (MULTI) {
        const int sset = lane / BIN_YB, sseg = lane % BIN_YB;
        if (sset < sets.n && sseg < nseg) {
            const int *offp = sets.offsets[0];
            const BinEntry *entp = sets.entries[0];
#pragma unroll
            for (int i = 1; i < OWN_MAX_SETS; ++i) {
                offp = sset == i ? sets.offsets[i] : offp;
                entp = sset == i ? sets.entries[i] : entp;
            }
            const int k_lo = all_x ? yb_lo * BIN_XB : (yb_lo + sseg) * BIN_XB + xb_lo;
            const int k_hi = all_x ? (yb_hi + 1) * BIN_XB : (yb_lo + sseg) * BIN_XB + xb_hi + 1;
            seg_begin = offp[off_slab + k_lo];
            seg_len = offp[off_slab + k_hi] - seg_begin;
            seg_addr = reinterpret_cast<unsigned long long>(entp + (size_t)b * geo.qp + seg_begin);
        }
    } else if (lane < nseg) {
        const int *off = sets.offsets[0] + off_slab;
        const int k_lo = all_x ? yb_lo * BIN_XB : (yb_lo + lane) * BIN_XB + xb_lo;
        const int k_hi = all_x ? (yb_hi + 1) * BIN_XB : (yb_lo + lane) * BIN_XB + xb_hi + 1;
        seg_begin = off[k_lo];
        seg_len = off[k_hi] - seg_begin;
    }
    static_assert(BIN_YB <= 8, "the single-set prefix runs over 8 lanes");
    const int incl = msmv_wave_incl_scan<MULTI ? 64 : 8>(seg_len);
    const int seg_excl = incl - seg_len;
    [[maybe_unused]] const int seg_delta = seg_begin - seg_excl;
    const int total = idle ? 0 : __builtin_amdgcn_readlane(incl, MULTI ? 63 : nseg - 1);
    MSMV_DBG_STAMP(1)
    if constexpr (MULTI) {
        excl_tab[lane] = seg_excl;          // non-decreasing; lanes past the last segment hold `total`
        addr_tab[lane] = seg_addr;
    }

    // This wave's j-th entry is virtual entry j*K + sub of the concatenated segments.
    // entry_index: its index in the entry array for j = j0 + lane (clamped into range so that
    // every lane loads something).
    auto entry_index = [&](int j0) {
        const int v = min(((j0 + lane) << klog) + sub, total - 1);
        if constexpr (MULTI) {
            // the last segment that begins at or before v (empty segments share their successor's
            // count and are skipped by that rule): six probes of the wave's table
            int g = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1)
                g = excl_tab[g + step] <= v ? g + step : g;
            return ((long long)g << 32) | (unsigned)(v - excl_tab[g]);   // segment, entry within it
        } else {
            int idx = v + __builtin_amdgcn_readlane(seg_delta, 0);
            for (int sgm = 1; sgm < nseg; ++sgm) {
                const int cs = __builtin_amdgcn_readlane(seg_excl, sgm);
                const int ds = __builtin_amdgcn_readlane(seg_delta, sgm);
                idx = v >= cs ? v + ds : idx;
            }
            return idx;
        }
    };
    // weight l of an entry sits in dword 3 + l; eset: the set the entry belongs to (MULTI)
    auto load_entry = [&](auto ref, float4 &e0, float &wl, int &eset) {
        if constexpr (MULTI) {
            const int g = (int)(ref >> 32);
            const unsigned within = (unsigned)ref;
            const float4 *e = reinterpret_cast<const float4 *>(addr_tab[g] + (unsigned long long)within * sizeof(BinEntry));
            e0 = e[0];
            wl = reinterpret_cast<const float *>(e)[3 + l];
            eset = g / BIN_YB;
        } else {
            e0 = list[(size_t)ref * 2];
            wl = reinterpret_cast<const float *>(list + (size_t)ref * 2)[3 + l];
        }
    };
    const float fHm1 = (float)(H - 1), fWm1 = (float)(W - 1), fH = (float)H, fW = (float)W;
    auto scan = [&](int j0, const float4 &e0, float wl) {
        OwnHit h;
        const bool valid = ((j0 + lane) << klog) + sub < total;
        h.pt = __float_as_int(e0.x);
        // the tap's arithmetic (msmv_make_tap): pixel coordinates, range test, floor, fractions
        float h_im = e0.z * fHm1, w_im = e0.y * fWm1;
        const bool ok = valid && (h_im > -1.f) && (w_im > -1.f) && (h_im < fH) && (w_im < fW);
        h_im = ok ? h_im : 0.f;
        w_im = ok ? w_im : 0.f;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const float lh = h_im - hf, lw = w_im - wf;
        h.dr0 = (int)hf - r0;
        h.dc0 = (int)wf - c0;
        // rows dr0, dr0 + 1 and columns dc0, dc0 + 1 against the patch's nrow x ncol pixels.  A
        // corner that is off the image (row -1 or H, column -1 or W) lies on no pixel that is
        // stored, so per-corner zero padding needs no test here.
        h.hit = ok && (unsigned)(h.dr0 + 1) <= (unsigned)nrow && (unsigned)(h.dc0 + 1) <= (unsigned)ncol;
        h.y0 = (1.f - lh) * wl;
        h.y1 = lh * wl;
        h.x0 = 1.f - lw;
        h.x1 = lw;
        return h;
    };

    // lane-constant pieces of the queue reads: record `half` of a pair, this pixel's row / column
    const float *q_wy = queue + half * REC_DW + REC_Y0 + (l32 >> 3);
    const float *q_wx = queue + half * REC_DW + REC_X0 + (l32 & 7);
    const float *q_pt = queue + half * REC_DW;

#ifdef MSMV_ABL_NOSCAN
    if (false)
#endif
    if (sub < total) {
        float4 e0;
        float wl;
        [[maybe_unused]] int eset = 0;
        load_entry(entry_index(0), e0, wl, eset);
#ifdef MSMV_DBG_TIMES
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MSMV_DBG_STAMP(2)
#endif
        for (int j0 = 0; (j0 << klog) + sub < total; j0 += 64) {
            const OwnHit h = scan(j0, e0, wl);
            [[maybe_unused]] const float hset = __int_as_float(eset);
            [[maybe_unused]] const int eset_of_h = eset;
            if (((j0 + 64) << klog) + sub < total)             // next chunk's entries, in flight
                load_entry(entry_index(j0 + 64), e0, wl, eset);     // while this chunk's hits are applied

            if constexpr (H16) {
                const int hn = lane & 15, hk = lane >> 4;    // half-pixel / channel quadruple; hit within a group of four
                // Columns 4 hf .. 4 hf + 3 of the patch are reached by a footprint that begins at column
                // 4 hf - 1 .. 4 hf + 3 (one on columns 3 and 4 is queued for both halves).  The chunk's
                // records: those of the left half first, padded to whole groups of four, then the right
                // half's, padded likewise - a group belongs to one half, whose tiles its matrix
                // instructions name.  A record is HREC_DW dwords: the gradient row's address, then the
                // separable weight tables for rows -1 .. 4 and the half's columns -1 .. 4.
                constexpr int HREC_DW = 16, HREC_Y0 = 3, HREC_X0 = 9;
                constexpr int GPB = HL == 2 ? 1 : 2;                   // groups of four records per batch of row loads
                constexpr int HPAD = HL == 2 ? 4 * GPB : 4;            // a half's records are padded to a multiple of this
                constexpr int HQ = STAGE_F4 * 4 / HREC_DW;             // records the wave's queue holds (its staging area)
                static_assert(HQ >= 64 && HQ % HPAD == 0, "a chunk's records of one half fit the queue");
                float *hqueue = reinterpret_cast<float *>(stage);
                const bool inL = h.hit && h.dc0 <= 3, inR = h.hit && h.dc0 >= 3;
                unsigned long long hitsL = __ballot(inL), hitsR = __ballot(inR);
#ifdef MSMV_ABL_NOHITS
                asm volatile("" ::"v"(h.dr0), "v"(h.dc0), "v"(h.y0), "v"(h.y1), "v"(h.x0), "v"(h.x1));
                hitsL = hitsR = 0;
#endif
                const int nL = __popcll(hitsL), nR = __popcll(hitsR);
                const int padL = (nL + HPAD - 1) / HPAD * HPAD, padR = (nR + HPAD - 1) / HPAD * HPAD;
                const int slotL = __builtin_amdgcn_mbcnt_hi((unsigned)(hitsL >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hitsL, 0u));
                const int slotR = __builtin_amdgcn_mbcnt_hi((unsigned)(hitsR >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hitsR, 0u));
                // what a record says about its gradient row: the address of the row's first float (P = 2: of
                // the query's [channel][point] pairs, the point in bit 2 - rows are 16-byte aligned)
                const float *hrow = gout;
                if constexpr (MULTI)
                    hrow = gout_tab[eset_of_h];
                hrow += (size_t)(PT == 2 ? h.pt >> 1 : h.pt) * (PT * 64);
                const unsigned long long hrow_bits = reinterpret_cast<unsigned long long>(hrow) | (PT == 2 ? (unsigned long long)(h.pt & 1) << 2 : 0ull);
                const float hrow_lo = __int_as_float((int)(unsigned)hrow_bits), hrow_hi = __int_as_float((int)(unsigned)(hrow_bits >> 32));

                // one half's records -> the queue -> the half's tiles (hf is a constant: the tiles are registers)
                auto half_pass = [&](auto hfc, bool mine, int slot, int n_pad, int dch) {
                    constexpr int hf = decltype(hfc)::value;
                    if (n_pad == 0)
                        return;                                        // wave-uniform
                    // null records first (a valid row, all weights 0), then the hit lanes' own: whole
                    // groups of four are then loaded and applied without per-record tests
                    typedef float rec_f32x4 __attribute__((ext_vector_type(4)));
                    if (lane < n_pad) {
                        rec_f32x4 *rec = reinterpret_cast<rec_f32x4 *>(hqueue + lane * HREC_DW);
                        const rec_f32x4 head = {hrow_lo, hrow_hi, 0.f, 0.f};
                        const rec_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        rec[0] = head;
#pragma unroll
                        for (int i = 1; i < HREC_DW / 4; ++i)
                            rec[i] = zero;
                    }
                    if (mine) {
                        float *rec = hqueue + slot * HREC_DW;
                        rec[0] = hrow_lo;
                        rec[1] = hrow_hi;
                        rec[HREC_Y0 + h.dr0] = h.y0;
                        rec[HREC_Y0 + 1 + h.dr0] = h.y1;
                        rec[HREC_X0 + dch] = h.x0;
                        rec[HREC_X0 + 1 + dch] = h.x1;
                    }
                    // A batch = GPB groups of four records: every record's row address first (one LDS read
                    // each, all in flight together), then the batch's row loads - plain loads, the compiler
                    // counts those in flight - then its matrix instructions.
                    struct Batch {
                        msmv_f32x4v g[GPB][PT];
                        int odd[GPB];
                    };
                    auto load_batch = [&](auto ngc, int g0) {
                        constexpr int NG = decltype(ngc)::value;
                        Batch bt;
                        const float *q0 = hqueue + (g0 * 4 + hk) * HREC_DW;
                        unsigned long long rbits[NG];
#pragma unroll
                        for (int gi = 0; gi < NG; ++gi)
                            rbits[gi] = *reinterpret_cast<const unsigned long long *>(q0 + gi * 4 * HREC_DW);
#pragma unroll
                        for (int gi = 0; gi < NG; ++gi) {
                            typedef __attribute__((address_space(1))) const msmv_f32x4v *grow_t;
                            const grow_t src = (grow_t)(rbits[gi] & ~15ull) + hn * PT;
#pragma unroll
                            for (int j = 0; j < PT; ++j)
                                bt.g[gi][j] = src[j];
                            bt.odd[gi] = (int)(rbits[gi] & 4ull);
                        }
                        return bt;
                    };
                    auto apply_batch = [&](auto ngc, const Batch &bt, int g0) {
                        constexpr int NG = decltype(ngc)::value;
                        const float *q0 = hqueue + (g0 * 4 + hk) * HREC_DW;
                        float wa[NG];
#pragma unroll
                        for (int gi = 0; gi < NG; ++gi) {
                            const float *r = q0 + gi * 4 * HREC_DW;
                            wa[gi] = r[HREC_Y0 + (hn >> 2)] * r[HREC_X0 + (hn & 3)];
                        }
#pragma unroll
                        for (int gi = 0; gi < NG; ++gi) {
                            msmv_f32x4v b = bt.g[gi][0];
                            if constexpr (PT == 2) {
                                // [channel][point] pairs: the record's point picks the even or the odd floats
                                const msmv_f32x4v lo = bt.g[gi][0], hi = bt.g[gi][PT - 1];
                                const bool odd = bt.odd[gi] != 0;
                                b = msmv_f32x4v{odd ? lo[1] : lo[0], odd ? lo[3] : lo[2], odd ? hi[1] : hi[0], odd ? hi[3] : hi[2]};
                            }
#ifdef MSMV_ABL_NORMW
                            asm volatile("" ::"v"(wa[gi]), "v"(b));
#else
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                hacc[hf][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[gi], b[t], hacc[hf][t], 0, 0, 0);
#endif
                        }
                    };
                    const int ngroups = n_pad >> 2;
                    constexpr std::integral_constant<int, GPB> whole = {};
                    if constexpr (HL == 2) {
                        // the rows of batch b + 1 are requested before batch b is applied: their latency
                        // runs beside its matrix instructions (one unconditional batch per iteration)
                        const int nbatch = ngroups / GPB;
                        Batch cur = load_batch(whole, 0);
                        for (int b = 1; b < nbatch; ++b) {
                            const Batch nxt = load_batch(whole, b * GPB);
                            apply_batch(whole, cur, (b - 1) * GPB);
                            cur = nxt;
                        }
                        apply_batch(whole, cur, (nbatch - 1) * GPB);
                    } else {
                        int g0 = 0;
                        for (; g0 + GPB <= ngroups; g0 += GPB)
                            apply_batch(whole, load_batch(whole, g0), g0);
                        msmv_static_for<GPB - 1>([&](auto kc) {
                            constexpr std::integral_constant<int, decltype(kc)::value + 1> part = {};
                            if (ngroups - g0 == part.value)            // wave-uniform
                                apply_batch(part, load_batch(part, g0), g0);
                        });
                    }
                };
                half_pass(std::integral_constant<int, 0>{}, inL, slotL, padL, h.dc0);
                half_pass(std::integral_constant<int, 1>{}, inR, slotR, padR, h.dc0 - 4);
            } else {
                unsigned long long hits = __ballot(h.hit);
#ifdef MSMV_ABL_NOHITS
                asm volatile("" ::"v"(h.dr0), "v"(h.dc0), "v"(h.y0), "v"(h.y1), "v"(h.x0), "v"(h.x1));
                hits = 0;
#endif
                const int nh = __popcll(hits);
                if constexpr (ACC)
                    any_hit = any_hit || nh > 0;
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(hits >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((unsigned)hits, 0u));
                for (int qbase = 0; qbase < nh; qbase += QC) {
                    // every slot first becomes a null record (a valid point, all weights 0), then
                    // hit lanes qbase .. qbase+QC-1 (in lane order) fill theirs in: the queue can
                    // then be consumed in whole pairs without branches around the loads
                    if (lane < QC) {
                        typedef float rec_f32x4 __attribute__((ext_vector_type(4)));
                        rec_f32x4 *rec = reinterpret_cast<rec_f32x4 *>(queue4 + lane * (REC_DW / 4));
                        const rec_f32x4 head = {__int_as_float(h.pt), MULTI ? hset : 0.f, 0.f, 0.f};
                        const rec_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        rec[0] = head;
#pragma unroll
                        for (int i = 1; i < REC_DW / 4; ++i)
                            rec[i] = zero;
                    }
                    const unsigned slot = (unsigned)(rank - qbase);
                    if (h.hit && slot < (unsigned)QC) {
                        float *rec = queue + slot * REC_DW;
                        rec[0] = __int_as_float(h.pt);
                        rec[1] = MULTI ? hset : 0.f;
                        rec[REC_Y0 + h.dr0] = h.y0;
                        rec[REC_Y0 + 1 + h.dr0] = h.y1;
                        rec[REC_X0 + h.dc0] = h.x0;
                        rec[REC_X0 + 1 + h.dc0] = h.x1;
                    }
                    const int cnt = min(QC, nh - qbase);
                    [[maybe_unused]] const int npairs = (cnt + 1) >> 1;

#ifdef MSMV_ABL_NORMW
#define MSMV_OWN_MFMA(WA, G0, G1) asm volatile("" ::"v"(WA), "v"(G0), "v"(G1));
#else
#define MSMV_OWN_MFMA(WA, G0, G1)                                                             \
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(WA, G0, acc0, 0, 0, 0);       \
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(WA, G1, acc1, 0, 0, 0);
#endif

                    if constexpr (STAGED) {
                        // ---- the round's gradient rows: global -> LDS by direct loads, 1 KiB per wave
                        // instruction, as a pipeline of UNITS (whole pairs of records and whole load
                        // instructions): the staging area holds NS units, the loads of unit v + D are
                        // issued before unit v is applied, and a counted wait leaves them in flight
                        // (vector memory operations complete in order).  Every slot of the queue holds a
                        // record with a valid point (null records have zero weights), so whole units are
                        // loaded and applied without per-record tests - 0 times whatever an earlier
                        // kernel left in LDS could be NaN, hence no row is ever left unloaded.
                        constexpr int RPI = 64 / ROW_F4;                   // records per load instruction
                        constexpr int UNIT_REC = RPI < 2 ? 2 : RPI;        // records per unit
                        constexpr int IPU = UNIT_REC / RPI;                // load instructions per unit
                        constexpr int PPU = UNIT_REC / 2;                  // pairs per unit
                        constexpr int UNIT_F4 = UNIT_REC * ROW_F4;
                        constexpr int NS = STAGE_F4 / UNIT_F4;             // units the staging area holds
                        constexpr int D = NS - 1;                          // units in flight ahead of the one applied
                        constexpr int NU = QC / UNIT_REC;                  // units per full round
                        static_assert(NS >= 2 && NU * UNIT_REC == QC && D * IPU <= 3, "pipeline shape");
                        const int nunits = (cnt + UNIT_REC - 1) / UNIT_REC;
                        auto issue = [&](int v) {                          // (v is a constant after unrolling)
#pragma unroll
                            for (int j = 0; j < IPU; ++j) {
                                const int rec = v * UNIT_REC + j * RPI + lane / ROW_F4;
                                const int rpt = __float_as_int(queue[rec * REC_DW]);
                                const float *go = gout;
                                if constexpr (MULTI)
                                    go = gout_tab[__float_as_int(queue[rec * REC_DW + 1])];
                                const int row = PT == 4 ? rpt >> 2 : PT == 2 ? rpt >> 1 : rpt;
                                const float *src = go + ((size_t)row * ROW_F4 + (lane % ROW_F4)) * 4;
                                msmv_load_to_lds_b128<MULTI && MSMV_OWN_ROWS_NT_MULTI>(src, stage + (v % NS) * UNIT_F4 + j * 64);
                            }
                        };
                        // what the applying lane needs of unit v's records: its weight per pair and, for
                        // P > 1, where in the staged row its point's channels sit
                        auto load_q = [&](int v, float (&wa)[PPU], int (&po)[PPU]) {
#pragma unroll
                            for (int k2 = 0; k2 < PPU; ++k2) {
                                const int u = v * PPU + k2;                // pair: records 2u and 2u + 1
                                wa[k2] = q_wy[u * 2 * REC_DW] * q_wx[u * 2 * REC_DW];
                                po[k2] = PT == 1 ? 0 : __float_as_int(q_pt[u * 2 * REC_DW]) & (PT - 1);
                            }
                        };
                        auto apply = [&](int v, const float (&wa)[PPU], const int (&po)[PPU]) {
                            const float *sp0 = reinterpret_cast<const float *>(stage + (v % NS) * UNIT_F4 + half * ROW_F4) +
                                               l32 * (PT == 1 ? 1 : PT);
#pragma unroll
                            for (int k2 = 0; k2 < PPU; ++k2) {
                                const float *sp = sp0 + k2 * 2 * ROW_F4 * 4;
                                const float g0 = sp[po[k2]], g1 = sp[32 * (PT == 1 ? 1 : PT) + po[k2]];
                                MSMV_OWN_MFMA(wa[k2], g0, g1)
                                if constexpr (ACC)
                                    touched = touched || wa[k2] != 0.f;
                            }
                        };
                        // A round of NUV units as straight-line code (every bound a constant: no branch
                        // keeps the compiler from moving loads up): the queue values of unit v + 1 are
                        // fetched and the rows of unit v + D requested before unit v's rows are waited for.
                        auto run_round = [&](auto nuv_c) {
                            constexpr int NUV = decltype(nuv_c)::value;
                            // (a unit of one pair fetches its queue values one unit ahead; units of
                            // several pairs overlap within themselves and keep the registers)
                            constexpr bool AHEAD = PPU == 1;
                            float wa[2][PPU];
                            int po[2][PPU];
                            if constexpr (AHEAD)
                                load_q(0, wa[0], po[0]);
                            msmv_static_for<(D < NUV ? D : NUV)>([&](auto vc) { issue(decltype(vc)::value); });
                            msmv_static_for<NUV>([&](auto vc) {
                                constexpr int v = decltype(vc)::value;
                                if constexpr (v + D < NUV)
                                    issue(v + D);
                                if constexpr (AHEAD && v + 1 < NUV)
                                    load_q(v + 1, wa[(v + 1) & 1], po[(v + 1) & 1]);
                                if constexpr (!AHEAD)
                                    load_q(v, wa[v & 1], po[v & 1]);
                                // loads issued after unit v's: those of min(D, NUV - 1 - v) units
                                msmv_wait_vmcnt<(D < NUV - 1 - v ? D : NUV - 1 - v) * IPU>();
                                apply(v, wa[v & 1], po[v & 1]);
                            });
                        };
                        msmv_static_for<NU>([&](auto kc) {
                            constexpr int k = decltype(kc)::value + 1;
                            if (nunits == k)                               // wave-uniform
                                run_round(std::integral_constant<int, k>{});
                        });
                        // the next round's copies and queue writes must not overtake this round's LDS reads
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    } else {
                        // P-strided gradient rows: channels l32 and l32 + 32 of the record's point, gathered
                        // into registers one pair ahead of their use
                        auto fetch = [&](int u, float &g0, float &g1) {
                            const int pt = __float_as_int(q_pt[u * 2 * REC_DW]);
                            const float *go = gout;
                            if constexpr (MULTI)
                                go = gout_tab[__float_as_int(q_pt[u * 2 * REC_DW + 1])];
                            const int bq = pt / geo.P, p = pt - bq * geo.P;
                            const float *gp = go + ((size_t)bq * 64 + l32) * geo.P + p;
                            g0 = gp[0];
                            g1 = gp[(size_t)32 * geo.P];
                        };
                        float g0, g1;
                        fetch(0, g0, g1);
                        for (int u = 0; u < npairs; ++u) {
                            float n0 = 0.f, n1 = 0.f;
                            if (u + 1 < npairs)                            // wave-uniform
                                fetch(u + 1, n0, n1);
                            const float wa = q_wy[u * 2 * REC_DW] * q_wx[u * 2 * REC_DW];
                            MSMV_OWN_MFMA(wa, g0, g1)
                            if constexpr (ACC)
                                touched = touched || wa != 0.f;
                            g0 = n0;
                            g1 = n1;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
#undef MSMV_OWN_MFMA
                }
            }
        }
    }

    MSMV_DBG_STAMP(3)
    // ---- K > 1: sum the K partial tiles of a patch, halving the live waves per round.  The
    // round count is the workgroup's (one level, one K): every wave reaches every barrier.
    if (K > 1)
        __syncthreads();                 // every wave is done with its staging area: it is reused below
    for (int stride = K >> 1; stride >= 1; stride >>= 1) {
        const int slot = (wave >> klog) * (K >> 1) + (sub & (stride - 1));   // < OWN_WAVES / 2, one per pair
        if (sub >= stride && sub < 2 * stride) {
            float *dstw = red[slot] + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (H16) {
                    dstw[r * 64] = hacc[0][r >> 2][r & 3];
                    dstw[(16 + r) * 64] = hacc[1][r >> 2][r & 3];
                } else {
                    dstw[r * 64] = acc0[r];
                    dstw[(16 + r) * 64] = acc1[r];
                }
            }
        }
        __syncthreads();
        if (sub < stride) {
            const float *srcw = red[slot] + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (H16) {
                    hacc[0][r >> 2][r & 3] += srcw[r * 64];
                    hacc[1][r >> 2][r & 3] += srcw[(16 + r) * 64];
                } else {
                    acc0[r] += srcw[r * 64];
                    acc1[r] += srcw[(16 + r) * 64];
                }
            }
        }
        __syncthreads();
    }
    MSMV_DBG_STAMP(4)
#ifdef MSMV_DBG_TIMES
    if (sub != 0 || idle) {
        if (lane == 0 && msmv_dbg_buf) {
            unsigned long long *o = msmv_dbg_buf + ((size_t)blockIdx.x * OWN_WAVES + wave) * 8;
            o[0] = dbg_t[0]; o[1] = dbg_t[1]; o[2] = dbg_t[2]; o[3] = dbg_t[3]; o[4] = dbg_t[4]; o[5] = 0;
            o[6] = dbg_where | ((unsigned long long)l << 32) | (unsigned)total;
            o[7] = ((unsigned long long)b << 32) | ((unsigned)view << 16) | (unsigned)(idle ? 0xFFFF : sub);
        }
        return;
    }
#endif
    if (sub != 0 || idle)
        return;
    if constexpr (ACC) {
        if (K > 1)
            any_hit = total > 0;         // (a patch whose bins hold entries but no hit adds zeros)
    }

    // ---- store the patch once.  Accumulator register r of lane (half, l32) is patch pixel
    // i = (r & 3) + 8 * (r >> 2) + 4 * half, i.e. row r >> 2, column (r & 3) + 4 * half, channel
    // l32 (tile 0) or 32 + l32 (tile 1): each half-wave stores 128 contiguous bytes of one pixel
    // per instruction.  The row base is wave-uniform, the lane's part (its channel and the
    // half-wave's 4 columns) is one register for all 32 stores, the rest an immediate.
    float *dst = G.dfeat + (((size_t)b * N + view) * (size_t)S + (size_t)r0 * W + c0) * 64;
    const unsigned voff = (unsigned)(half * 4 * 64 + l32);
#ifdef MSMV_ABL_NOSTORE
    if (nrow < 0)
#endif
    if constexpr (H16) {
        // lane (n, r): row r of the patch, channels 4n .. 4n+3; register i of the four tiles of half hf
        // is column 4 hf + i: one instruction stores a column of the half, 4 rows x 256 contiguous bytes
        const int hn = lane & 15, hr = lane >> 4;
        float *o = dst + (size_t)hr * W * 64 + 4 * hn;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = 4 * hf + i;
                if (hr < nrow && col < ncol) {
                    const msmv_f32x4v v = {hacc[hf][0][i], hacc[hf][1][i], hacc[hf][2][i], hacc[hf][3][i]};
#if MSMV_OWN_NT_STORE
                    __builtin_nontemporal_store(v, reinterpret_cast<msmv_f32x4v *>(o + col * 64));
#else
                    *reinterpret_cast<msmv_f32x4v *>(o + col * 64) = v;
#endif
                }
            }
        }
    } else if constexpr (ACC) {
        if (!any_hit)
            return;                                          // wave-uniform
        // Only the pixels that received a weight are read, added to and written back: on the
        // finest level a layer's points reach about half of them.  Bit i of `pixels`: patch
        // pixel i (either half-wave may have applied it).  Patches shared by several waves are
        // on crowded levels, where every pixel is reached: all in-range pixels there.
        const unsigned long long tb = __ballot(touched);
        const unsigned pixels = K > 1 ? 0xFFFFFFFFu : (unsigned)tb | (unsigned)(tb >> 32);
        // in two batches of 8 pixels: all loads of a batch first (clamped to the patch's first
        // pixel where nothing is to be done, so that no branch separates them), then the adds
        // and stores; two batches keep the kernel's register count that of its main loop
#pragma unroll
        for (int r0b = 0; r0b < 16; r0b += 8) {
            float old0[8], old1[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int r = r0b + kk;
                const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int pr = r >> 2, pc = (r & 3) + 4 * half;
                const bool in = pr < nrow && pc < ncol && ((pixels >> i) & 1u);
                const float *src = dst + (in ? ((size_t)pr * W + pc) * 64 : 0) + l32;
                old0[kk] = msmv_load_stream(src);
                old1[kk] = msmv_load_stream(src + 32);
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int r = r0b + kk;
                const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int pr = r >> 2, pc = (r & 3) + 4 * half;
                if (pr < nrow && pc < ncol && ((pixels >> i) & 1u)) {
                    float *o = dst + ((size_t)pr * W + pc) * 64 + l32;
                    msmv_store_stream(o, old0[kk] + acc0[r]);
                    msmv_store_stream(o + 32, old1[kk] + acc1[r]);
                }
            }
        }
    } else if (nrow == PATCH_H && ncol == PATCH_W) {
        // a whole patch (all but the last row / column of patches): no per-pixel tests
#pragma unroll
        for (int pr = 0; pr < PATCH_H; ++pr) {
            float *rowp = dst + (size_t)pr * W * 64;         // wave-uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = pr * 4 + q;
                msmv_store_stream(rowp + (voff + q * 64), acc0[r]);
                msmv_store_stream(rowp + (voff + q * 64 + 32), acc1[r]);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pr = r >> 2, pc = (r & 3) + 4 * half;
            if (pr < nrow && pc < ncol) {
                float *o = dst + ((size_t)pr * W + pc) * 64 + l32;
                msmv_store_stream(o, acc0[r]);
                msmv_store_stream(o + 32, acc1[r]);
            }
        }
    }
#ifdef MSMV_DBG_TIMES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (what s_endpgm would wait for anyway: the stores' drain)
    MSMV_DBG_STAMP(5)
    if (lane == 0 && msmv_dbg_buf) {
        unsigned long long *o = msmv_dbg_buf + ((size_t)blockIdx.x * OWN_WAVES + wave) * 8;
        o[0] = dbg_t[0]; o[1] = dbg_t[1]; o[2] = dbg_t[2]; o[3] = dbg_t[3]; o[4] = dbg_t[4]; o[5] = dbg_t[5];
        o[6] = dbg_where | ((unsigned long long)l << 32) | (unsigned)total;
        o[7] = ((unsigned long long)b << 32) | ((unsigned)view << 16) | (unsigned)sub;
    }
#endif
}

// ======================================================================================
// owner algorithm: d loc / d w gather (row of 16 lanes per point, like the forward)
// ======================================================================================
#ifndef MSMV_GAT_THREADS
#define MSMV_GAT_THREADS 256
#endif
constexpr int GAT_THREADS = MSMV_GAT_THREADS;
#ifndef MSMV_GAT_GPW
#define MSMV_GAT_GPW 1
#endif
constexpr int GAT_GROUPS_PER_WAVE = MSMV_GAT_GPW;

// SORTED: rows walk the bin entries (points of a slab ordered by view and image band) instead
// of the points in input order, so that the rows a CU works on at one time touch
// neighbouring pixels; location, weights (and the view for L <= 4) come from the entry.
template <int NL, int PT, bool SORTED>
__global__ void __launch_bounds__(GAT_THREADS)
msmv_bwd_gather_c64(const MsmvArgs a, const OwnerWs ws)
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
        bool act = pt_raw < a.npts;
        int pt = act ? pt_raw : a.npts - 1;
        float lx, ly, wl[NL];
        int view, b;
        [[maybe_unused]] int grow = 0;           // row mode: the point's grad_out row
        static_assert(PT != MSMV_PT_ROWS || SORTED, "row mode walks the bin entries");
        if constexpr (SORTED) {
            // slot pt of the entry array: every slot of slab b = [b*QP, (b+1)*QP) holds a point of it
            b = msmv_batch_of(pt, a);
            const float4 *e = reinterpret_cast<const float4 *>(ws.entries + pt);
            const float4 e0 = e[0], e1 = e[1];
            pt = __float_as_int(e0.x);
            if constexpr (PT == MSMV_PT_ROWS) {
                grow = pt;
                pt = msmv_pt_of_row(grow, a);
            }
            lx = e0.y;
            ly = e0.z;
            const float wv[5] = {e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
            for (int l = 0; l < NL; ++l)
                wl[l] = wv[l];
            if constexpr (NL <= 4)
                view = __float_as_int(e1.w);
            else
                view = msmv_view_of(a.loc[(size_t)pt * 3 + 2], a.N);
        } else {
            const float *lp = a.loc + (size_t)pt * 3;
            lx = lp[0];
            ly = lp[1];
#pragma unroll
            for (int l = 0; l < NL; ++l)
                wl[l] = a.w[(size_t)pt * NL + l];
            view = msmv_view_of(lp[2], a.N);
            b = msmv_batch_of(pt, a);
        }
        const bool view_ok = view >= 0;
        const size_t slab = (size_t)b * a.N + (view_ok ? view : 0);

        MsmvTap tap[NL];
        float4 v[NL][4];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int H = a.lvl[l].H, W = a.lvl[l].W;
            tap[l] = msmv_make_tap(lx, ly, H, W, view_ok);
            const float *base = a.lvl[l].feat + (slab * (size_t)(H * W)) * 64 + l16 * 4;
            // (finest level: streaming hint, as in the forward)
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

        // this row's 4 channels of grad_out for its point (inverse of the forward's store)
        float4 gq;
        if constexpr (PT == MSMV_PT_ROWS) {
            gq = ld4(a.gout + (size_t)grow * 64 + l16 * 4);
        } else if constexpr (PT == 1) {
            gq = ld4(a.gout + (size_t)pt * 64 + l16 * 4);
        } else if constexpr (SORTED && PT == 2) {
            // rows no longer hold the p = 0,1 pair of one query: fetch the row's 4 channels x 2 p
            const float4 *gp = reinterpret_cast<const float4 *>(a.gout + ((size_t)(pt >> 1) * 64 + l16 * 4) * 2);
            const float4 A = gp[0], B = gp[1];
            const bool odd = pt & 1;
            gq.x = odd ? A.y : A.x;
            gq.y = odd ? A.w : A.z;
            gq.z = odd ? B.y : B.x;
            gq.w = odd ? B.w : B.z;
        } else if constexpr (SORTED && PT == 4) {
            const float4 *gp = reinterpret_cast<const float4 *>(a.gout + ((size_t)(pt >> 2) * 64 + l16 * 4) * 4);
            const float4 A = gp[0], B = gp[1], C = gp[2], D = gp[3];
            const bool p1 = pt & 1, p2 = pt & 2;
#define MSMV_PICK(V) (p2 ? (p1 ? (V).w : (V).z) : (p1 ? (V).y : (V).x))
            gq.x = MSMV_PICK(A);
            gq.y = MSMV_PICK(B);
            gq.z = MSMV_PICK(C);
            gq.w = MSMV_PICK(D);
#undef MSMV_PICK
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
            // (an out-of-image corner's value is discarded, not multiplied by 0: see the forward)
            float sval = 0.f, sgw = 0.f, sgh = 0.f;
#define MSMV_GATHER_CH(X)                                                          \
            {                                                                      \
                const float v1 = t.c1 ? v[l][0].X : 0.f, v2 = t.c2 ? v[l][1].X : 0.f; \
                const float v3 = t.c3 ? v[l][2].X : 0.f, v4 = t.c4 ? v[l][3].X : 0.f; \
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
            // (a point without a valid view: exact zeros, whatever its gradient row holds)
            float *dl = a.dloc + (size_t)pt * 3;
            dl[0] = view_ok ? gx : 0.f; dl[1] = view_ok ? gy : 0.f; dl[2] = 0.f;
#pragma unroll
            for (int l = 0; l < NL; ++l)
                a.dw[(size_t)pt * NL + l] = view_ok ? dwl[l] : 0.f;
        }
    }
}

template <int NL>
void launch_gather(const MsmvArgs &a, const OwnerWs &ws, hipStream_t s)
{
    const int ngroups = (a.npts + 3) / 4;
    const int per_block = (GAT_THREADS / 64) * GAT_GROUPS_PER_WAVE;
    const int nblk = (ngroups + per_block - 1) / per_block;
    switch (a.rows ? MSMV_PT_ROWS : a.P) {
    // row mode: a point's gradient is one contiguous row whatever P is, so bin order always pays
    case MSMV_PT_ROWS: hipLaunchKernelGGL((msmv_bwd_gather_c64<NL, MSMV_PT_ROWS, true>), dim3(nblk), dim3(GAT_THREADS), 0, s, a, ws); break;
    case 1: hipLaunchKernelGGL((msmv_bwd_gather_c64<NL, 1, true>), dim3(nblk), dim3(GAT_THREADS), 0, s, a, ws); break;
    case 2: hipLaunchKernelGGL((msmv_bwd_gather_c64<NL, 2, true>), dim3(nblk), dim3(GAT_THREADS), 0, s, a, ws); break;
    // P = 4 / other P: bin order would fetch 4x (Px) the gradient bytes per row; input order wins
    case 4: hipLaunchKernelGGL((msmv_bwd_gather_c64<NL, 4, false>), dim3(nblk), dim3(GAT_THREADS), 0, s, a, ws); break;
    default: hipLaunchKernelGGL((msmv_bwd_gather_c64<NL, 0, false>), dim3(nblk), dim3(GAT_THREADS), 0, s, a, ws); break;
    }
}

// ======================================================================================
// forward in bin order (row mode, or P == 1): the same row-of-16-lanes kernel as msmv_fwd_c64
// (msmv_forward.hip), but walking the sorted entries, so that the rows a CU works on at one time
// touch neighbouring pixels and the feature traffic approaches the distinct pixels touched.  In
// training the sort is then the backward's too (OPUS_MSMV_BWD_PRESORTED).  A point's 64 channels
// are one contiguous 256-byte row of the output; a point without a valid view (the tail slots of
// its slab) gets a row of zeros like in input order.
// ======================================================================================
template <int NL>
__global__ void __launch_bounds__(GAT_THREADS)
msmv_fwd_sorted_c64(const MsmvArgs a, const OwnerWs ws)
{
    const int lane = threadIdx.x & 63;
    const int sub = lane >> 4;
    const int l16 = lane & 15;
    const int wave = threadIdx.x >> 6;
    const int blk = msmv_xcd_remap(blockIdx.x, gridDim.x);
    const int ngroups = (a.npts + 3) >> 2;
    const int g = blk * (GAT_THREADS / 64) + wave;
    if (g >= ngroups)
        return;
    const int slot_raw = g * 4 + sub;
    const bool act = slot_raw < a.npts;
    const int slot = act ? slot_raw : a.npts - 1;
    // slot of the entry array: every slot of slab b = [b*QP, (b+1)*QP) holds a point of it
    const int b = msmv_batch_of(slot, a);
    const float4 *e = reinterpret_cast<const float4 *>(ws.entries + slot);
    const float4 e0 = e[0], e1 = e[1];
    const int row = __float_as_int(e0.x);                    // output row (row mode) or point (P == 1)
    const float lx = e0.y, ly = e0.z;
    const float wv[5] = {e0.w, e1.x, e1.y, e1.z, e1.w};
    float wl[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l)
        wl[l] = wv[l];
    int view;
    if constexpr (NL <= 4) {
        view = __float_as_int(e1.w);
    } else {
        const int pt = a.rows ? msmv_pt_of_row(row, a) : row;
        view = msmv_view_of(a.loc[(size_t)pt * 3 + 2], a.N);
    }
    const bool view_ok = view >= 0;
    const size_t slab = (size_t)b * a.N + (view_ok ? view : 0);

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
        // the arithmetic of msmv_fwd_c64, operation for operation: results are bit-identical
#define MSMV_FWDS_CH(X)                                                            \
        {                                                                          \
            const float v1 = t.c1 ? v[l][0].X : 0.f, v2 = t.c2 ? v[l][1].X : 0.f;  \
            const float v3 = t.c3 ? v[l][2].X : 0.f, v4 = t.c4 ? v[l][3].X : 0.f;  \
            acc.X = msmv_accum_level(t, acc.X, msmv_blend4(t, v1, v2, v3, v4), wl[l]); \
        }
        MSMV_FWDS_CH(x) MSMV_FWDS_CH(y) MSMV_FWDS_CH(z) MSMV_FWDS_CH(w)
#undef MSMV_FWDS_CH
    }
    if (act)
        *reinterpret_cast<float4 *>(a.out + (size_t)row * 64 + l16 * 4) = acc;
}

}  // namespace

// ======================================================================================
// host launchers
// ======================================================================================
hipError_t msmv_launch_backward_atomic(const MsmvArgs &a, hipStream_t s, bool accumulate)
{
    const hipError_t stale = hipPeekAtLastError();
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
    return msmv_launch_status(stale);
}

bool msmv_backward_owner_supported(const MsmvArgs &a)
{
    return a.C == 64 && (a.L == 4 || a.L == 5) && a.N * BIN_PER_VIEW <= BIN_MAX_KEYS;
}

size_t msmv_backward_owner_workspace(int B, int N, int Q, int P)
{
    return owner_ws_header_bytes(B, N, Q * P) + (size_t)B * Q * P * sizeof(BinEntry);
}

template <bool ACC, bool MULTI>
static void launch_owner_dfeat(int pt, dim3 grid, dim3 block, hipStream_t s, const OwnerGeom &geo, const OwnerSets &sets,
                               int hit_loop)
{
    if constexpr (!ACC) {
        if (hit_loop == 1 && pt == 1) {
            hipLaunchKernelGGL((msmv_bwd_owner_dfeat<1, false, MULTI, 1>), grid, block, 0, s, geo, sets);
            return;
        }
        if (hit_loop == 2 && pt == 1) {
            hipLaunchKernelGGL((msmv_bwd_owner_dfeat<1, false, MULTI, 2>), grid, block, 0, s, geo, sets);
            return;
        }
        if (hit_loop == 1 && pt == 2) {
            hipLaunchKernelGGL((msmv_bwd_owner_dfeat<2, false, MULTI, 1>), grid, block, 0, s, geo, sets);
            return;
        }
        if (hit_loop == 2 && pt == 2) {
            hipLaunchKernelGGL((msmv_bwd_owner_dfeat<2, false, MULTI, 2>), grid, block, 0, s, geo, sets);
            return;
        }
    }
    switch (pt) {
    case 1: hipLaunchKernelGGL((msmv_bwd_owner_dfeat<1, ACC, MULTI>), grid, block, 0, s, geo, sets); break;
    case 2: hipLaunchKernelGGL((msmv_bwd_owner_dfeat<2, ACC, MULTI>), grid, block, 0, s, geo, sets); break;
    case 4: hipLaunchKernelGGL((msmv_bwd_owner_dfeat<4, ACC, MULTI>), grid, block, 0, s, geo, sets); break;
    default: hipLaunchKernelGGL((msmv_bwd_owner_dfeat<0, ACC, MULTI>), grid, block, 0, s, geo, sets); break;
    }
}

static OwnerWs owner_ws_of(const MsmvArgs &a, void *workspace)
{
    OwnerWs ws;
    ws.keys = a.N * BIN_PER_VIEW;
    ws.chunks = owner_ws_chunks(a.qp);
    const int per = (a.qp + ws.chunks - 1) / ws.chunks;
    ws.chunk_pts = per > 0 ? per : 1;
    ws.table = reinterpret_cast<int *>(workspace);
    ws.offsets = ws.table + (size_t)a.B * ws.chunks * (ws.keys + 1);
    ws.entries = reinterpret_cast<BinEntry *>(reinterpret_cast<char *>(workspace) + owner_ws_header_bytes(a.B, a.N, a.qp));
    return ws;
}

// Workgroup order of the d-feat pass (see the kernel): views per group - N = level-major, 1 = view-major.
// OPUS_MSMV_DFEAT_ORDER = "view" / "level" / a number of views overrides the default for measurements; a
// value that does not divide N falls back to N.
static int msmv_dfeat_views_per_group(int nsets, int N)
{
    static const int forced = [] {
        const char *e = getenv("OPUS_MSMV_DFEAT_ORDER");
        return !e ? -1 : (e[0] == 'v' ? 1 : (e[0] == 'l' ? 0 : atoi(e)));
    }();
    int vpu = forced >= 0 ? forced : (nsets > 1 ? MSMV_DFEAT_VPU_MULTI : MSMV_DFEAT_VPU_SINGLE);
    if (vpu < 1 || vpu > N || N % vpu != 0)
        vpu = N;
    return vpu;
}

// Which hit loop (see the kernel's HL) for a pass over `nsets` point sets with gradient rows of kind
// `pt` (1: contiguous rows, 2: the op's layout with two points per query).  Measured at the OPUS shapes:
// the 16 x 16 x 4 loops win wherever rows are contiguous and for one set in the op's layout; several sets
// in the op's layout (512-byte rows, the pass at the fabric's rate already) stay on the 32 x 32 x 2 loop.
// There, loop 2 is ahead while the finest level's patches expect few hits (OPUS-S / M: 2-5 %) and behind
// when they expect many (OPUS-L: 3-6 %).  OPUS_MSMV_DFEAT_HL = 0 / 1 / 2 overrides the choice for measurements.
#ifndef MSMV_DFEAT_HL_SINGLE_ROWS
#define MSMV_DFEAT_HL_SINGLE_ROWS 1
#endif
#ifndef MSMV_DFEAT_HL_SINGLE_P2
#define MSMV_DFEAT_HL_SINGLE_P2 2
#endif
#ifndef MSMV_DFEAT_HL_MULTI_ROWS
#define MSMV_DFEAT_HL_MULTI_ROWS 1
#endif
#ifndef MSMV_DFEAT_HL_MULTI_P2
#define MSMV_DFEAT_HL_MULTI_P2 0
#endif
#ifndef MSMV_DFEAT_HL_MULTI_P2_SPARSE
#define MSMV_DFEAT_HL_MULTI_P2_SPARSE 2          // ... while a patch of the finest level expects fewer hits than
#endif
#ifndef MSMV_DFEAT_HL_MULTI_P2_SPARSE_HITS
#define MSMV_DFEAT_HL_MULTI_P2_SPARSE_HITS 30    // this (six OPUS-M layers: 19, six OPUS-L layers: 38)
#endif
static int msmv_dfeat_hit_loop(int nsets, int pt, double finest_hits_per_patch)
{
    static const int forced = [] {
        const char *e = getenv("OPUS_MSMV_DFEAT_HL");
        return !e ? -1 : (e[0] == '2' ? 2 : (e[0] == '1' ? 1 : 0));
    }();
    if (forced >= 0)
        return forced;
    if (pt == 1)
        return nsets > 1 ? MSMV_DFEAT_HL_MULTI_ROWS : MSMV_DFEAT_HL_SINGLE_ROWS;
    if (pt == 2 && nsets > 1)
        return finest_hits_per_patch < MSMV_DFEAT_HL_MULTI_P2_SPARSE_HITS ? MSMV_DFEAT_HL_MULTI_P2_SPARSE : MSMV_DFEAT_HL_MULTI_P2;
    if (pt == 2)
        return MSMV_DFEAT_HL_SINGLE_P2;
    return 0;
}

// The d-feat pass over `sets.n` sorted point sets (each of a.qp points per batch slab).
static hipError_t launch_dfeat_sets(const MsmvArgs &a, const OwnerSets &sets, hipStream_t s, bool accumulate)
{
    OwnerGeom geo = {};
    geo.L = a.L; geo.N = a.N; geo.qp = a.qp; geo.P = a.P;
    int per_view = 0;
    for (int l = 0; l < a.L; ++l) {
        OwnerLevel &g = geo.lv[l];
        g.dfeat = a.lvl[l].dfeat;
        g.H = a.lvl[l].H;
        g.W = a.lvl[l].W;
        g.patch_cols = (g.W + PATCH_W - 1) / PATCH_W;
        g.patches = ((g.H + PATCH_H - 1) / PATCH_H) * g.patch_cols;
        // a point touches ~1.4 patches of a level: expected hits per patch, uniform points
        const double expect = 1.4 * (double)a.qp * sets.n / ((double)a.N * g.patches);
        int klog = 0;
        const double split_hits = sets.n > 1 ? MSMV_OWN_SPLIT_HITS_MULTI : MSMV_OWN_SPLIT_HITS;
        while ((2 << klog) <= OWN_WAVES && expect > split_hits * (1 << klog))
            ++klog;
        g.split_log2 = klog;
        const int per_wg = OWN_WAVES >> klog;
        g.tiles = (g.patches + per_wg - 1) / per_wg;
        g.div_tiles = msmv_make_div(g.tiles);
        g.div_cols = msmv_make_div(g.patch_cols);
        g.div_hm1 = msmv_make_div(g.H - 1);
        g.div_wm1 = msmv_make_div(g.W - 1);
        per_view += g.tiles;
    }
    // workgroup order within a batch slab (or within a view of it): coarse level first
    geo.vpu = msmv_dfeat_views_per_group(sets.n, a.N);
    geo.tiles_per_group = per_view * geo.vpu;
    geo.div_tiles_per_group = msmv_make_div(geo.tiles_per_group);
    long long end = 0;
    for (int k = 0; k < MSMV_MAX_LEVELS; ++k) {
        if (k < a.L - 1) {
            end += (long long)geo.lv[a.L - 1 - k].tiles * geo.vpu;
            geo.seg_end[k] = (int)end;
        } else {
            geo.seg_end[k] = 0x7FFFFFFF;
        }
    }
    geo.tiles_per_b = per_view * a.N;
    geo.div_tiles_per_b = msmv_make_div(geo.tiles_per_b);
    const long long ntiles = (long long)geo.tiles_per_b * a.B;
    if (ntiles > 0x7FFFFFFFll)
        return hipErrorInvalidValue;
    if (ntiles > 0) {
        const dim3 grid((unsigned)ntiles), block(OWN_THREADS);
        const int pt = a.rows ? 1 : a.P;
        const double finest_hits = 1.4 * (double)a.qp * sets.n / ((double)a.N * geo.lv[0].patches);
        const int hl = msmv_dfeat_hit_loop(sets.n, pt, finest_hits);
        if (sets.n > 1 && accumulate)
            launch_owner_dfeat<true, true>(pt, grid, block, s, geo, sets, hl);
        else if (sets.n > 1)
            launch_owner_dfeat<false, true>(pt, grid, block, s, geo, sets, hl);
        else if (accumulate)
            launch_owner_dfeat<true, false>(pt, grid, block, s, geo, sets, hl);
        else
            launch_owner_dfeat<false, false>(pt, grid, block, s, geo, sets, hl);
    }
    return hipSuccess;
}

hipError_t msmv_launch_dfeat_sets(const MsmvArgs &a, int nsets, const float *const *gouts,
                                  void *const *workspaces, hipStream_t s, bool accumulate)
{
    if (nsets < 1 || nsets > OWN_MAX_SETS)
        return hipErrorInvalidValue;
    OwnerSets sets = {};
    sets.n = nsets;
    for (int i = 0; i < nsets; ++i) {
        const OwnerWs ws = owner_ws_of(a, workspaces[i]);
        sets.gout[i] = gouts[i];
        sets.offsets[i] = ws.offsets;
        sets.entries[i] = ws.entries;
    }
    const hipError_t stale = hipPeekAtLastError();
    const hipError_t e = launch_dfeat_sets(a, sets, s, accumulate);
    return e != hipSuccess ? e : msmv_launch_status(stale);
}

int msmv_dfeat_max_sets() { return OWN_MAX_SETS; }

#ifdef MSMV_DBG_TIMES
extern "C" int opus_msmv_debug_set_buffer(void *p)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(msmv_dbg_buf), &p, sizeof(p));
}
#endif

// The counting sort of every batch slab's points into `workspace` (see OwnerWs).
static hipError_t launch_sort(const MsmvArgs &a, const OwnerWs &ws, hipStream_t s)
{
    if (a.qp > 0 && a.qp <= BIN_ONE_PPT * BIN_THREADS) {
        hipLaunchKernelGGL(msmv_bin_one, dim3((unsigned)a.B), dim3(BIN_THREADS), 0, s, a, ws);
        return hipSuccess;
    }
    // (no queries: one workgroup per slab still writes the offsets, "every bin empty")
    const dim3 grid((unsigned)(a.B * ws.chunks)), block(BIN_THREADS);
    hipLaunchKernelGGL(msmv_bin_count, grid, block, 0, s, a, ws);
    hipLaunchKernelGGL(msmv_bin_fill, grid, block, 0, s, a, ws);
    return hipSuccess;
}

hipError_t msmv_launch_forward_sorted(const MsmvArgs &a, void *workspace, hipStream_t s, bool presorted)
{
    const OwnerWs ws = owner_ws_of(a, workspace);
    const hipError_t stale = hipPeekAtLastError();
    hipError_t e;
    if (!presorted && (e = launch_sort(a, ws, s)) != hipSuccess)
        return e;
    if (a.npts > 0) {
        const int ngroups = (a.npts + 3) / 4;
        const int per_block = GAT_THREADS / 64;
        const dim3 grid((unsigned)((ngroups + per_block - 1) / per_block)), block(GAT_THREADS);
        if (a.L == 4)
            hipLaunchKernelGGL((msmv_fwd_sorted_c64<4>), grid, block, 0, s, a, ws);
        else
            hipLaunchKernelGGL((msmv_fwd_sorted_c64<5>), grid, block, 0, s, a, ws);
    }
    return msmv_launch_status(stale);
}

hipError_t msmv_launch_backward_owner(const MsmvArgs &a, void *workspace, hipStream_t s, int stages, bool accumulate)
{
    const OwnerWs ws = owner_ws_of(a, workspace);
    const hipError_t stale = hipPeekAtLastError();

    hipError_t e;
    if ((stages & 1) && (e = launch_sort(a, ws, s)) != hipSuccess)
        return e;
    if ((e = msmv_launch_status(stale)) != hipSuccess)
        return e;

    // The gather runs BEFORE the d-feat pass: both need only the sort and write disjoint outputs, and
    // the gather reads every gradient row anyway - it leaves them in the Infinity Cache, where the
    // d-feat pass, whose waves are bound by the latency of exactly those rows, finds them (measured:
    // d feat 171 -> 164 us at OPUS-L after a forward, 158 us with nothing else in between).
    if (a.npts > 0 && (stages & 4)) {
        if (a.L == 4)
            launch_gather<4>(a, ws, s);
        else
            launch_gather<5>(a, ws, s);
    }
    if (stages & 2) {
        OwnerSets sets = {};
        sets.n = 1;
        sets.gout[0] = a.gout;
        sets.offsets[0] = ws.offsets;
        sets.entries[0] = ws.entries;
        if ((e = launch_dfeat_sets(a, sets, s, accumulate)) != hipSuccess)
            return e;
    }
    return msmv_launch_status(stale);
}
