// Shared device-side definitions of the msmv sampling kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MSMV_MAX_LEVELS 8
#define MSMV_WAVE 64
#define MSMV_PT_ROWS (-1)     // template value of "points per query" that selects row mode

struct MsmvLevel {
    const float *feat;   // [B, N, H, W, C]
    float *dfeat;        // same shape (backward only)
    int H, W;
};

// Unsigned division by a run-time constant d without a divide: for d >= 2 the host sets
// l = ceil(log2 d), magic = ceil(2^(31+l) / d) (< 2^32) and shift = l - 1; then
// x / d == umulhi(x, magic) >> shift for every x < 2^31.  d == 1 is encoded as magic == 0.
struct MsmvDiv {
    uint32_t magic, shift;
};

// Passed by value as the kernel argument block.
struct MsmvArgs {
    MsmvLevel lvl[MSMV_MAX_LEVELS];
    const float *loc;      // [B, Q, P, 3]
    const float *w;        // [B, Q, P, L]
    float *out;            // forward: [B, Q, C, P]
    const float *gout;     // backward: grad of out
    float *dloc;           // [B, Q, P, 3]
    float *dw;             // [B, Q, P, L]
    int L, B, C, N, Q, P;
    int npts;              // B*Q*P  (host guarantees < 2^31)
    int qp;                // Q*P
    uint32_t qp_magic;     // floor-division of a point index by Q*P without a divide:
    uint32_t qp_shift;     //   pt / qp == umulhi(pt, qp_magic) >> qp_shift   (pt < 2^31)
    // Row mode (rows != 0): B = Bs*T*G with batch slab b = (bs*T + t)*G + g, and out / gout are
    // [Bs, Q, G, T*P, C]: the C channels of point (b, q, p) are contiguous at row
    // (((bs*Q + q)*G + g)*T + t)*P + p - the layout the caller of the reference's op permutes
    // its result into (models/opus_sampling.py:119-122).
    int rows, T, G;
    MsmvDiv div_p, div_q, div_t, div_g, div_tg;
};

inline MsmvDiv msmv_make_div(int d)
{
    MsmvDiv r = {0, 0};
    if (d >= 2) {
        int l2 = 0;
        while ((1ll << l2) < d)
            ++l2;                                                // ceil(log2(d)), >= 1
        const unsigned __int128 one = 1;
        const unsigned __int128 num = one << (31 + l2);
        r.magic = (uint32_t)((num + (unsigned)d - 1) / (unsigned)d);
        r.shift = (uint32_t)(l2 - 1);
    }
    return r;
}

__device__ __forceinline__ uint32_t msmv_div(uint32_t x, MsmvDiv d)
{
    return d.magic ? __umulhi(x, d.magic) >> d.shift : x;
}

// pt / (Q*P) for 0 <= pt < 2^31.  For d = Q*P >= 2 the host sets l = ceil(log2 d),
// qp_magic = ceil(2^(31+l) / d) (< 2^32) and qp_shift = l - 1, which is exact for every
// 31-bit numerator; d = 1 has no 32-bit magic and is handled by the comparison.
__device__ __forceinline__ int msmv_batch_of(int pt, const MsmvArgs &a)
{
    return a.qp == 1 ? pt : (int)(__umulhi((uint32_t)pt, a.qp_magic) >> a.qp_shift);
}

// Row mode: output row of point pt, and its inverse.
__device__ __forceinline__ int msmv_row_of(int pt, const MsmvArgs &a)
{
    const uint32_t b = (uint32_t)msmv_batch_of(pt, a);
    const uint32_t rem = (uint32_t)pt - b * (uint32_t)a.qp;           // q*P + p
    const uint32_t q = msmv_div(rem, a.div_p), p = rem - q * a.P;
    const uint32_t bs = msmv_div(b, a.div_tg), r = b - bs * (uint32_t)(a.T * a.G);
    const uint32_t t = msmv_div(r, a.div_g), g = r - t * a.G;
    return (int)((((bs * a.Q + q) * a.G + g) * a.T + t) * a.P + p);
}

__device__ __forceinline__ int msmv_pt_of_row(int row, const MsmvArgs &a)
{
    uint32_t x = (uint32_t)row;
    uint32_t y = msmv_div(x, a.div_p);
    const uint32_t p = x - y * a.P;
    x = msmv_div(y, a.div_t);
    const uint32_t t = y - x * a.T;
    y = msmv_div(x, a.div_g);
    const uint32_t g = x - y * a.G;
    const uint32_t bs = msmv_div(y, a.div_q), q = y - bs * a.Q;
    return (int)((((bs * a.T + t) * a.G + g) * a.Q + q) * a.P + p);
}

// One bilinear tap of one level: geometry shared by all channels of a point.
// Mirrors the arithmetic of the reference tap (msmv_sampling_forward.cu:32-69):
// align_corners=True pixel coordinates, floor, per-corner zero padding.
struct MsmvTap {
    float w1, w2, w3, w4;   // hh*hw, hh*lw, lh*hw, lh*lw  with invalid corners' weights as computed (not zeroed)
    float lh, lw, hh, hw;
    int o1, o2, o3, o4;     // pixel offsets (h*W + w) of the four corners, clamped into the slab
    int h_low, w_low;       // row / column of corner 1 (may be -1)
    bool ok;                // level in range: -1 < h_im < H and -1 < w_im < W
    bool c1, c2, c3, c4;    // corner inside the image
};

__device__ __forceinline__ MsmvTap msmv_make_tap(float loc_w, float loc_h, int H, int W, bool view_ok)
{
    MsmvTap t;
    float h_im = loc_h * (float)(H - 1);
    float w_im = loc_w * (float)(W - 1);
    // NaN compares false, so a NaN location is out of range like in the reference (forward.cu:126)
    t.ok = view_ok && (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
    h_im = t.ok ? h_im : 0.f;
    w_im = t.ok ? w_im : 0.f;
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h_low = (int)hf, w_low = (int)wf;
    t.h_low = h_low;
    t.w_low = w_low;
    t.lh = h_im - hf;
    t.lw = w_im - wf;
    t.hh = 1.f - t.lh;
    t.hw = 1.f - t.lw;
    const bool top = h_low >= 0, left = w_low >= 0;
    const bool bot = h_low + 1 <= H - 1, right = w_low + 1 <= W - 1;
    t.c1 = t.ok && top && left;
    t.c2 = t.ok && top && right;
    t.c3 = t.ok && bot && left;
    t.c4 = t.ok && bot && right;
    const int r0 = top ? h_low : 0, r1 = bot ? h_low + 1 : H - 1;
    const int q0 = left ? w_low : 0, q1 = right ? w_low + 1 : W - 1;
    t.o1 = r0 * W + q0;
    t.o2 = r0 * W + q1;
    t.o3 = r1 * W + q0;
    t.o4 = r1 * W + q1;
    t.w1 = t.hh * t.hw;
    t.w2 = t.hh * t.lw;
    t.w3 = t.lh * t.hw;
    t.w4 = t.lh * t.lw;
    return t;
}

// The bilinear blend and the accumulation over levels with ONE rounding order for every kernel that
// promises bit-identical results (a sum of products leaves the compiler a choice of which product to
// fuse into the multiply-add; written out, there is none).
__device__ __forceinline__ float msmv_blend4(const MsmvTap &t, float v1, float v2, float v3, float v4)
{
    return fmaf(t.w4, v4, fmaf(t.w3, v3, fmaf(t.w2, v2, t.w1 * v1)));
}

__device__ __forceinline__ float msmv_accum_level(const MsmvTap &t, float acc, float val, float wl)
{
    // a level that is out of range adds nothing at all (msmv_sampling_forward.cu:126)
    return t.ok ? fmaf(val, wl, acc) : acc;
}

// view index of a point: round-half-away-from-zero of loc_v*(N-1) (forward.cu:110).
// Returns -1 for anything that does not land in [0, N-1] (NaN, infinities, out of range):
// the reference would index out of bounds there; such a point contributes nothing here.
__device__ __forceinline__ int msmv_view_of(float loc_v, int N)
{
    const float r = roundf(loc_v * (float)(N - 1));
    return (r >= 0.f && r < (float)N) ? (int)r : -1;
}

// ---- cross-lane sums -------------------------------------------------------------------

template <int CTRL>
__device__ __forceinline__ float msmv_dpp_add(float v)
{
    const int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false);
    return v + __int_as_float(x);
}

// Sum over each row of 16 lanes; every lane of the row ends up holding the row's sum.
__device__ __forceinline__ float msmv_row16_sum(float v)
{
    v = msmv_dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = msmv_dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = msmv_dpp_add<0x141>(v);   // row_half_mirror
    v = msmv_dpp_add<0x140>(v);   // row_mirror
    return v;
}

// Sum over all 64 lanes, returned wave-uniform.
__device__ __forceinline__ float msmv_wave_sum(float v)
{
    v = msmv_row16_sum(v);
    const int x = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(x, 0)) +
            __int_as_float(__builtin_amdgcn_readlane(x, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(x, 32)) +
            __int_as_float(__builtin_amdgcn_readlane(x, 48)));
}

// Logical block index for a launch of `nblk` blocks such that the blocks that share an XCD
// (observed: blockIdx % 8) work on one contiguous range of logical indices, so that one
// batch slab's feature maps are served by one XCD's L2.  Bijective for every nblk.
__device__ __forceinline__ int msmv_xcd_remap(int bid, int nblk)
{
#ifdef MSMV_XCD_SHIFT
    const int xcd = (bid + MSMV_XCD_SHIFT) & 7, within = bid >> 3;     // development aid: XCD x works on another XCD's range
#else
    const int xcd = bid & 7, within = bid >> 3;
#endif
    const int q = nblk >> 3, r = nblk & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + within;
}
