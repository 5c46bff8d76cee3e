/*
 * CPU oracle for the multi-scale multi-view sampling op.  TEST INFRASTRUCTURE ONLY:
 * nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's library; the product path never does.
 *
 * It restates, in plain C, what the reference's device kernels compute:
 *   forward  : models/csrc/msmv_sampling/msmv_sampling_forward.cu:27-73   (bilinear tap)
 *              models/csrc/msmv_sampling/msmv_sampling_forward.cu:75-164  (4-level kernel)
 *              models/csrc/msmv_sampling/msmv_sampling_forward.cu:166-267 (5-level kernel)
 *   backward : models/csrc/msmv_sampling/msmv_sampling_backward.cu:29-105 (bilinear tap)
 *              models/csrc/msmv_sampling/msmv_sampling_backward.cu:108-224, :226-361
 * Parity is pinned by the .npz fixtures under tests/golden, generated in the build container from the
 * reference's own PyTorch fallback (models/csrc/wrapper.py:14-38), see
 * tests/golden/make_golden.py.
 *
 * Layouts (all fp32, contiguous):
 *   feats[l] : [B, N, H_l, W_l, C]   channel-last
 *   loc      : [B, Q, P, 3]          (x/width, y/height, view/(N-1)), each in [0,1]
 *   w        : [B, Q, P, L]
 *   out      : [B, Q, C, P]
 * The level count L is a runtime value (the reference ships L=4 and L=5 kernels that
 * differ only in the unrolled level list and the weight stride).
 *
 * One deliberate definition where the reference is undefined: a view index that rounds
 * outside [0, N-1] would read out of bounds in the reference kernels
 * (msmv_sampling_forward.cu:110,127); here such a point contributes zero.
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

typedef struct {
    int ok;              /* level in range: -1 < h < H and -1 < w < W (forward.cu:126) */
    int h_low, w_low;
    int c1, c2, c3, c4;  /* per-corner validity (forward.cu:49,54,59,64) */
    float lh, lw, hh, hw;
} tap_t;

static tap_t make_tap(float loc_w, float loc_h, int H, int W)
{
    tap_t t;
    /* align_corners=True mapping, forward.cu:123-124 */
    const float h_im = loc_h * (float)(H - 1);
    const float w_im = loc_w * (float)(W - 1);
    t.ok = (h_im > -1 && w_im > -1 && h_im < H && w_im < W);
    if (!t.ok) {
        memset(&t, 0, sizeof t);
        return t;
    }
    t.h_low = (int)floorf(h_im);
    t.w_low = (int)floorf(w_im);
    t.lh = h_im - (float)t.h_low;
    t.lw = w_im - (float)t.w_low;
    t.hh = 1 - t.lh;
    t.hw = 1 - t.lw;
    const int h_high = t.h_low + 1, w_high = t.w_low + 1;
    t.c1 = (t.h_low >= 0 && t.w_low >= 0);
    t.c2 = (t.h_low >= 0 && w_high <= W - 1);
    t.c3 = (h_high <= H - 1 && t.w_low >= 0);
    t.c4 = (h_high <= H - 1 && w_high <= W - 1);
    return t;
}

/* view index: round-half-away-from-zero of loc_v * (N-1), forward.cu:110 */
static int view_of(float loc_v, int N)
{
    return (int)roundf(loc_v * (float)(N - 1));
}

void msmv_oracle_forward(int L, const float *const *feats, const int *hw,
                         const float *loc, const float *w,
                         int B, int C, int N, int Q, int P, float *out)
{
#pragma omp parallel for schedule(static)
    for (long bq = 0; bq < (long)B * Q; ++bq) {
        const long b = bq / Q;
        for (int p = 0; p < P; ++p) {
            const long pt = bq * P + p;
            const float loc_w = loc[pt * 3 + 0];
            const float loc_h = loc[pt * 3 + 1];
            const int v = view_of(loc[pt * 3 + 2], N);
            const int v_ok = (v >= 0 && v < N);
            for (int c = 0; c < C; ++c)
                out[(bq * C + c) * P + p] = 0.f;
            if (!v_ok)
                continue;
            for (int l = 0; l < L; ++l) {
                const int H = hw[2 * l], W = hw[2 * l + 1];
                const tap_t t = make_tap(loc_w, loc_h, H, W);
                if (!t.ok)
                    continue;
                const float wl = w[pt * L + l];
                const float *base = feats[l] + ((size_t)(b * N + v) * H * W) * C;
                const float w1 = t.hh * t.hw, w2 = t.hh * t.lw, w3 = t.lh * t.hw, w4 = t.lh * t.lw;
                const size_t o1 = ((size_t)t.h_low * W + t.w_low) * C;
                const size_t o2 = o1 + C, o3 = o1 + (size_t)W * C, o4 = o3 + C;
                for (int c = 0; c < C; ++c) {
                    const float v1 = t.c1 ? base[o1 + c] : 0.f;
                    const float v2 = t.c2 ? base[o2 + c] : 0.f;
                    const float v3 = t.c3 ? base[o3 + c] : 0.f;
                    const float v4 = t.c4 ? base[o4 + c] : 0.f;
                    const float val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                    out[(bq * C + c) * P + p] += val * wl;
                }
            }
        }
    }
}

/*
 * Backward.  d feats are zero-filled here and then accumulated (the reference's host
 * shim does the zero-fill, msmv_sampling.cpp:244-249); dloc[...,2] is exactly 0 because
 * the reference never writes it (backward.cu:103-104).  Per-point sums over channels are
 * kept in double before the final cast, since the reference's atomic order is arbitrary.
 */
void msmv_oracle_backward(int L, const float *grad_out, const float *const *feats,
                          const int *hw, const float *loc, const float *w,
                          int B, int C, int N, int Q, int P,
                          float *const *dfeats, float *dloc, float *dw)
{
    for (int l = 0; l < L; ++l)
        memset(dfeats[l], 0, sizeof(float) * (size_t)B * N * hw[2 * l] * hw[2 * l + 1] * C);
    memset(dloc, 0, sizeof(float) * (size_t)B * Q * P * 3);
    memset(dw, 0, sizeof(float) * (size_t)B * Q * P * L);

    /* batch slabs are independent: no two b write the same d feat element */
#pragma omp parallel for schedule(static)
    for (long b = 0; b < B; ++b) {
        for (long q = 0; q < Q; ++q) {
            const long bq = b * Q + q;
            for (int p = 0; p < P; ++p) {
                const long pt = bq * P + p;
                const float loc_w = loc[pt * 3 + 0];
                const float loc_h = loc[pt * 3 + 1];
                const int v = view_of(loc[pt * 3 + 2], N);
                if (v < 0 || v >= N)
                    continue;
                double gx = 0, gy = 0;
                for (int l = 0; l < L; ++l) {
                    const int H = hw[2 * l], W = hw[2 * l + 1];
                    const tap_t t = make_tap(loc_w, loc_h, H, W);
                    if (!t.ok)
                        continue;
                    const float wl = w[pt * L + l];
                    const size_t slab = ((size_t)(b * N + v) * H * W) * C;
                    const float *base = feats[l] + slab;
                    float *gbase = dfeats[l] + slab;
                    const float w1 = t.hh * t.hw, w2 = t.hh * t.lw, w3 = t.lh * t.hw, w4 = t.lh * t.lw;
                    const size_t o1 = ((size_t)t.h_low * W + t.w_low) * C;
                    const size_t o2 = o1 + C, o3 = o1 + (size_t)W * C, o4 = o3 + C;
                    double gwl = 0;
                    for (int c = 0; c < C; ++c) {
                        const float top = grad_out[(bq * C + c) * P + p];
                        const float tgv = top * wl;                 /* backward.cu:55 */
                        float ghw = 0, gww = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                        if (t.c1) { v1 = base[o1 + c]; ghw -= t.hw * v1; gww -= t.hh * v1; gbase[o1 + c] += w1 * tgv; }
                        if (t.c2) { v2 = base[o2 + c]; ghw -= t.lw * v2; gww += t.hh * v2; gbase[o2 + c] += w2 * tgv; }
                        if (t.c3) { v3 = base[o3 + c]; ghw += t.hw * v3; gww -= t.lh * v3; gbase[o3 + c] += w3 * tgv; }
                        if (t.c4) { v4 = base[o4 + c]; ghw += t.lw * v4; gww += t.lh * v4; gbase[o4 + c] += w4 * tgv; }
                        const float val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                        gwl += (double)(top * val);                               /* backward.cu:102 */
                        gx += (double)((float)(W - 1) * gww * tgv);               /* backward.cu:103 */
                        gy += (double)((float)(H - 1) * ghw * tgv);               /* backward.cu:104 */
                    }
                    dw[pt * L + l] = (float)gwl;
                }
                dloc[pt * 3 + 0] = (float)gx;
                dloc[pt * 3 + 1] = (float)gy;
            }
        }
    }
}

/* exact count of distinct feature pixels touched (for the unique-bytes companion figure) */
long msmv_oracle_unique_pixels(int L, const int *hw, const float *loc,
                               int B, int N, int Q, int P, unsigned char *scratch_bitmap)
{
    long total_pix = 0;
    for (int l = 0; l < L; ++l)
        total_pix += (long)B * N * hw[2 * l] * hw[2 * l + 1];
    memset(scratch_bitmap, 0, (size_t)total_pix);
    long level_off = 0, uniq = 0;
    for (int l = 0; l < L; ++l) {
        const int H = hw[2 * l], W = hw[2 * l + 1];
        for (long pt = 0; pt < (long)B * Q * P; ++pt) {
            const long b = pt / ((long)Q * P);
            const int v = view_of(loc[pt * 3 + 2], N);
            if (v < 0 || v >= N)
                continue;
            const tap_t t = make_tap(loc[pt * 3], loc[pt * 3 + 1], H, W);
            if (!t.ok)
                continue;
            unsigned char *bm = scratch_bitmap + level_off + (size_t)(b * N + v) * H * W;
            const long o1 = (long)t.h_low * W + t.w_low;
            if (t.c1 && !bm[o1]) { bm[o1] = 1; ++uniq; }
            if (t.c2 && !bm[o1 + 1]) { bm[o1 + 1] = 1; ++uniq; }
            if (t.c3 && !bm[o1 + W]) { bm[o1 + W] = 1; ++uniq; }
            if (t.c4 && !bm[o1 + W + 1]) { bm[o1 + W + 1] = 1; ++uniq; }
        }
        level_off += (long)B * N * H * W;
    }
    return uniq;
}
