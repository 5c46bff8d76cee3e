// Probe for the sparse d-feat update: does gfx950 apply VGPR-index mode (s_set_gpr_idx_on) to the
// destination and src2 of v_fma_f32, and what does one "hit" cost when it is applied as four such
// multiply-adds into a 32-register accumulator tuple (lane = channel, register = patch pixel)?
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/gpr_idx_probe tools/gpr_idx_probe.hip && tools/gpr_idx_probe
//
// Part 1 (semantics): every wave applies a list of hits {pixel index, 4 weights, gradient row} and
// the result is compared with a host loop, bit for bit.
// Part 2 (rate, only with --at-scale): the same loop over many hits per wave, full chip, cycles per hit
// per SIMD.  CAUTION: kernels that hold index mode at full occupancy faulted the GPU on the
// development machines (a product kernel built on this instruction, and MODE 5 here); the rate runs
// of the other modes completed, unchecked.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x32 __attribute__((ext_vector_type(32)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// acc[idx + {0, 1, 8, 9}] += w{0..3} * g, idx wave-uniform.  Index mode adds M0[7:0] to the register
// number of the operands selected by the mode bits (here src2 and dst: 0xC); the tuple is pinned so
// that the instruction text can name its first register.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void apply_hit(f32x32 &acc, int idx, float w0, float w1, float w2, float w3, float g)
{
    asm volatile("s_set_gpr_idx_on %1, 0xc\n\t"
                 "v_fma_f32 v32, %2, %6, v32\n\t"
                 "v_fma_f32 v33, %3, %6, v33\n\t"
                 "v_fma_f32 v40, %4, %6, v40\n\t"
                 "v_fma_f32 v41, %5, %6, v41\n\t"
                 "s_set_gpr_idx_off"
                 : "+{v[32:63]}"(acc)
                 : "s"(idx), "s"(w0), "s"(w1), "s"(w2), "s"(w3), "v"(g)
                 : "m0");
}
#pragma clang diagnostic pop

struct Hit {
    int idx;
    float w[4];
    int row;
    int pad[2];
};

// lanes hold the hits (one per lane, 64 per chunk); the wave walks them in lane order, GROUP hits
// at a time: their gradient rows are requested first (one dword per lane, wave-uniform base in
// SGPRs), then applied in order.
// MODE 0: record through v_readlane (index, row, four weights); MODE 1: as 0 without the row loads;
// MODE 2: weights through a per-wave LDS queue read back with wave-uniform (broadcast) addresses.
constexpr int GROUP = 8;
template <int MODE>
__global__ void __launch_bounds__(256)
probe(const Hit *hits, int nchunks, const float *rows, float *out, long long *cycles)
{
    __shared__ float4 wq_all[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float4 *wq = wq_all[threadIdx.x >> 6];
    f32x32 acc = {};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < nchunks; ++c) {
        const Hit h = hits[((size_t)wave * nchunks + c) * 64 + lane];
        if (MODE == 2 || MODE >= 6)
            wq[lane] = make_float4(h.w[0], h.w[1], h.w[2], h.w[3]);
        for (int i0 = 0; i0 < 64; i0 += GROUP) {
            float g[GROUP];
#pragma unroll
            for (int k = 0; k < GROUP; ++k) {
                const int row = __builtin_amdgcn_readlane(h.row, i0 + k);
                unsigned long long base = (unsigned long long)rows + (unsigned long long)(unsigned)row * 256u;
                asm volatile("" : "+s"(base));       // keep the wave-uniform base apart from the lane's offset
                g[k] = MODE == 1 ? __int_as_float(row + lane) : reinterpret_cast<const __attribute__((address_space(1))) float *>(base)[lane];
            }
#pragma unroll
            for (int k = 0; k < GROUP; ++k) {
                const int i = i0 + k;
                const int idx = min(__builtin_amdgcn_readlane(h.idx, i), 22);     // (never beyond the tuple, whatever the data)
                float w0, w1, w2, w3;
                if (MODE == 6 || MODE == 7 || MODE == 8) {
                    // the two-operand multiply-add (VOP2, destination = addend) under index mode
                    const float4 w = wq[i];
                    w0 = w.x; w1 = w.y; w2 = w.z; w3 = w.w;
                    if (MODE == 6)
                        asm volatile("s_set_gpr_idx_on %1, 0xc\n\t"
                                     "v_fmac_f32 v32, %2, %6\n\tv_fmac_f32 v33, %3, %6\n\tv_fmac_f32 v40, %4, %6\n\tv_fmac_f32 v41, %5, %6\n\t"
                                     "s_set_gpr_idx_off"
                                     : "+{v[32:63]}"(acc) : "s"(idx), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(g[k]) : "m0");
                    else if (MODE == 7)
                        asm volatile("s_set_gpr_idx_on %1, 0x8\n\t"
                                     "v_fmac_f32 v32, %2, %6\n\tv_fmac_f32 v33, %3, %6\n\tv_fmac_f32 v40, %4, %6\n\tv_fmac_f32 v41, %5, %6\n\t"
                                     "s_set_gpr_idx_off"
                                     : "+{v[32:63]}"(acc) : "s"(idx), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(g[k]) : "m0");
                    else
                        asm volatile("s_set_gpr_idx_on %1, 0xc\n\t"
                                     "v_fmac_f32_dpp v32, %2, %6 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n\t"
                                     "v_fmac_f32_dpp v33, %3, %6 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n\t"
                                     "v_fmac_f32_dpp v40, %4, %6 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n\t"
                                     "v_fmac_f32_dpp v41, %5, %6 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n\t"
                                     "s_set_gpr_idx_off"
                                     : "+{v[32:63]}"(acc) : "s"(idx), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(g[k]) : "m0");
                } else if (MODE == 2) {
                    const float4 w = wq[i];
                    w0 = w.x; w1 = w.y; w2 = w.z; w3 = w.w;
                    asm volatile("s_set_gpr_idx_on %1, 0xc\n\t"
                                 "v_fma_f32 v32, %2, %6, v32\n\t"
                                 "v_fma_f32 v33, %3, %6, v33\n\t"
                                 "v_fma_f32 v40, %4, %6, v40\n\t"
                                 "v_fma_f32 v41, %5, %6, v41\n\t"
                                 "s_set_gpr_idx_off"
                                 : "+{v[32:63]}"(acc)
                                 : "s"(idx), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(g[k])
                                 : "m0");
                } else if (MODE == 4) {
                    // index mode switched on once for four hits, the index alone changed in between
                    if (k % 4 == 0) {
                        int ix[4];
                        float ww[4][4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            ix[u] = __builtin_amdgcn_readlane(h.idx, i + u);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                ww[u][e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.w[e]), i + u));
                        }
#define HIT4(I, A, B, C, D, G) "s_set_gpr_idx_idx " I "\n\tv_fma_f32 v32, " A ", " G ", v32\n\tv_fma_f32 v33, " B ", " G ", v33\n\tv_fma_f32 v40, " C ", " G ", v40\n\tv_fma_f32 v41, " D ", " G ", v41\n\t"
                        asm volatile("s_set_gpr_idx_on %1, 0xc\n\t"
                                     HIT4("%1", "%5", "%6", "%7", "%8", "%21")
                                     HIT4("%2", "%9", "%10", "%11", "%12", "%22")
                                     HIT4("%3", "%13", "%14", "%15", "%16", "%23")
                                     HIT4("%4", "%17", "%18", "%19", "%20", "%24")
                                     "s_set_gpr_idx_off"
                                     : "+{v[32:63]}"(acc)
                                     : "s"(ix[0]), "s"(ix[1]), "s"(ix[2]), "s"(ix[3]),
                                       "s"(ww[0][0]), "s"(ww[0][1]), "s"(ww[0][2]), "s"(ww[0][3]),
                                       "s"(ww[1][0]), "s"(ww[1][1]), "s"(ww[1][2]), "s"(ww[1][3]),
                                       "s"(ww[2][0]), "s"(ww[2][1]), "s"(ww[2][2]), "s"(ww[2][3]),
                                       "s"(ww[3][0]), "s"(ww[3][1]), "s"(ww[3][2]), "s"(ww[3][3]),
                                       "v"(g[k]), "v"(g[k + 1]), "v"(g[k + 2]), "v"(g[k + 3])
                                     : "m0");
                    }
                } else if (MODE == 3) {
                    // no index mode: fixed registers (timing floor of 6 readlanes + 4 multiply-adds)
                    w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.w[0]), i));
                    w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.w[1]), i));
                    w2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.w[2]), i));
                    w3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.w[3]), i));
                    asm volatile("v_fma_f32 v32, %2, %6, v32\n\t"
                                 "v_fma_f32 v33, %3, %6, v33\n\t"
                                 "v_fma_f32 v40, %4, %6, v40\n\t"
                                 "v_fma_f32 v41, %5, %6, v41"
                                 : "+{v[32:63]}"(acc)
                                 : "s"(idx), "s"(w0), "s"(w1), "s"(w2), "s"(w3), "v"(g[k]));
                } else {
                    w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.w[0]), i));
                    w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.w[1]), i));
                    w2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.w[2]), i));
                    w3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.w[3]), i));
                    apply_hit(acc, idx, w0, w1, w2, w3, g[k]);
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int p = 0; p < 32; ++p)
        out[((size_t)wave * 32 + p) * 64 + lane] = acc[p];
    if (lane == 0)
        cycles[wave] = t1 - t0;
}

// MODE 5: the candidate hit loop.  A round of 16 hit records sits in a per-wave LDS queue; lane i
// holds record i % 16 in registers (so every row of 16 lanes holds the round), the gradient rows are
// copied global -> LDS by direct loads with per-lane addresses (4 rows per instruction), and hit k
// is applied as four v_fmac_f32 whose weight operand is lane k of the row (DPP row_newbcast:k) and
// whose destination is index-relative.  The pixel indices of four consecutive records are packed
// into one dword so that one v_readlane serves four hits.
#ifndef VARIANT
#define VARIANT 0
#endif
#ifndef IDXMODE
#define IDXMODE "0xc"
#endif
typedef __attribute__((address_space(3))) void *lds_ptr_t;
__device__ __forceinline__ void load_to_lds_b128(const float *src, const float4 *dst)
{
    const unsigned lds = (unsigned)(uintptr_t)(lds_ptr_t)dst;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory", "m0");
}

#if VARIANT == 1
// weights broadcast into temporaries first (outside index mode), plain v_fmac inside
#define FMAC4N(K, G) \
    "s_set_gpr_idx_off\n\t" \
    "v_mov_b32_dpp %10, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
    "v_mov_b32_dpp %11, %3 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
    "v_mov_b32_dpp %12, %4 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
    "v_mov_b32_dpp %13, %5 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
    "s_set_gpr_idx_on %1, " IDXMODE "\n\t" \
    "v_fmac_f32 v32, %10, " G "\n\tv_fmac_f32 v33, %11, " G "\n\tv_fmac_f32 v40, %12, " G "\n\tv_fmac_f32 v41, %13, " G "\n\t"
#elif VARIANT == 2
// as 1 with the three-operand form
#define FMAC4N(K, G) \
    "s_set_gpr_idx_off\n\t" \
    "v_mov_b32_dpp %10, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
    "v_mov_b32_dpp %11, %3 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
    "v_mov_b32_dpp %12, %4 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
    "v_mov_b32_dpp %13, %5 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
    "s_set_gpr_idx_on %1, 0xc\n\t" \
    "v_fma_f32 v32, %10, " G ", v32\n\tv_fma_f32 v33, %11, " G ", v33\n\tv_fma_f32 v40, %12, " G ", v40\n\tv_fma_f32 v41, %13, " G ", v41\n\t"
#else
#define FMAC4N(K, G) \
    "v_fmac_f32_dpp v32, %2, " G " row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp v33, %3, " G " row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp v40, %4, " G " row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp v41, %5, " G " row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
#endif
// hits K0 .. K0+3 of the round: packed = their four pixel indices, one per byte
#define APPLY4(K0, K1, K2, K3)                                                                              \
    asm volatile("s_set_gpr_idx_on %1, " IDXMODE "\n\t" FMAC4N(K0, "%6")                                     \
                 "s_lshr_b32 %1, %1, 8\n\ts_set_gpr_idx_idx %1\n\t" FMAC4N(K1, "%7")                         \
                 "s_lshr_b32 %1, %1, 8\n\ts_set_gpr_idx_idx %1\n\t" FMAC4N(K2, "%8")                         \
                 "s_lshr_b32 %1, %1, 8\n\ts_set_gpr_idx_idx %1\n\t" FMAC4N(K3, "%9")                         \
                 "s_set_gpr_idx_off"                                                                          \
                 : "+{v[32:63]}"(acc), "+s"(packed)                                                          \
                 : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(g[K0]), "v"(g[K1]), "v"(g[K2]), "v"(g[K3]),       \
                   "v"(t0), "v"(t1), "v"(t2), "v"(t3)                                                        \
                 : "m0")
#ifndef IDXMODE
#define IDXMODE "0xc"
#endif

template <int K> __device__ __forceinline__ float bcast(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + K, 0xF, 0xF, false));
}

struct Rec {            // 32 bytes in the LDS queue
    float w[4];
    int idx, row;
    int pad[2];
};

__global__ void __launch_bounds__(256)
probe5(const Hit *hits, int nchunks, const float *rows, float *out, long long *cycles, int rowmask)
{
    __shared__ float4 queue_all[4][64 * 2];
    __shared__ float4 stage_all[4][256];                 // 16 rows of 256 B
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 *queue = queue_all[wv];
    float4 *stage = stage_all[wv];
    const float *stage_f = reinterpret_cast<const float *>(stage) + lane;
    f32x32 acc = {};
    for (int c = 0; c < nchunks; ++c) {
        const Hit h = hits[((size_t)wave * nchunks + c) * 64 + lane];
        queue[lane * 2] = make_float4(h.w[0], h.w[1], h.w[2], h.w[3]);
        queue[lane * 2 + 1] = make_float4(__int_as_float(h.idx), __int_as_float(h.row), 0.f, 0.f);
        for (int r = 0; r < 4; ++r) {
            // the round's rows: instruction j copies the rows of records 4j .. 4j+3
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = reinterpret_cast<const int *>(queue + (r * 16 + j * 4 + (lane >> 4)) * 2 + 1)[1];
                load_to_lds_b128(rows + (size_t)(row & rowmask) * 64 + (lane & 15) * 4, stage + j * 64);
            }
            const float4 rw = queue[(r * 16 + (lane & 15)) * 2];
            const int ridx = reinterpret_cast<const int *>(queue + (r * 16 + (lane & 15)) * 2 + 1)[0];
            const float w0 = rw.x, w1 = rw.y, w2 = rw.z, w3 = rw.w;
            // byte b of lane i: pixel index of record i + b (lanes 0, 4, 8, 12 are read)
            int pk = ridx;
            pk |= __builtin_amdgcn_update_dpp(0, ridx, 0x101, 0xF, 0xF, true) << 8;     // row_shl:1
            pk |= __builtin_amdgcn_update_dpp(0, ridx, 0x102, 0xF, 0xF, true) << 16;    // row_shl:2
            pk |= __builtin_amdgcn_update_dpp(0, ridx, 0x103, 0xF, 0xF, true) << 24;    // row_shl:3
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float g[16];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                g[k] = stage_f[k * 64];
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;      // (variants 1, 2: scratch registers, clobbered)
            asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
#if VARIANT == 3
            // weights broadcast by the compiler's DPP move, the index-mode block is MODE 2's
#define V3HIT(k)                                                                                            \
            {                                                                                                   \
                const int ix = min(__builtin_amdgcn_readlane(ridx, k), 22);                                     \
                const float b0 = bcast<k>(w0), b1 = bcast<k>(w1), b2 = bcast<k>(w2), b3 = bcast<k>(w3);         \
                asm volatile("s_set_gpr_idx_on %1, 0xc\n\t"                                                     \
                             "v_fma_f32 v32, %2, %6, v32\n\tv_fma_f32 v33, %3, %6, v33\n\tv_fma_f32 v40, %4, %6, v40\n\tv_fma_f32 v41, %5, %6, v41\n\t" \
                             "s_set_gpr_idx_off"                                                                 \
                             : "+{v[32:63]}"(acc) : "s"(ix), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(g[k]) : "m0"); \
            }
            V3HIT(0) V3HIT(1) V3HIT(2) V3HIT(3) V3HIT(4) V3HIT(5) V3HIT(6) V3HIT(7)
            V3HIT(8) V3HIT(9) V3HIT(10) V3HIT(11) V3HIT(12) V3HIT(13) V3HIT(14) V3HIT(15)
            if (false) {
#else
            {
#endif
            pk &= 0x1f1f1f1f;                                   // (never beyond the tuple, whatever the data)
            int packed = __builtin_amdgcn_readlane(pk, 0);
            APPLY4(0, 1, 2, 3);
            packed = __builtin_amdgcn_readlane(pk, 4);
            APPLY4(4, 5, 6, 7);
            packed = __builtin_amdgcn_readlane(pk, 8);
            APPLY4(8, 9, 10, 11);
            packed = __builtin_amdgcn_readlane(pk, 12);
            APPLY4(12, 13, 14, 15);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
#pragma unroll
    for (int p = 0; p < 32; ++p)
        out[((size_t)wave * 32 + p) * 64 + lane] = acc[p];
}

int main(int argc, char **argv)
{
    // Index-mode kernels at full occupancy ended in NaN results and GPU memory faults on the machines
    // this was developed on (the small checked runs never did): only the checked 8-wave runs are made
    // unless the rate runs are asked for by name.
    const bool at_scale = argc > 1 && strcmp(argv[1], "--at-scale") == 0;
    const int nrows = 1 << 12;                                 // 1 MiB of rows: they stay in L2
    std::vector<float> rows((size_t)nrows * 64);
    srand(7);
    for (auto &x : rows)
        x = (float)(rand() % 2001 - 1000) / 512.f;
    float *d_rows;
    CHECK(hipMalloc(&d_rows, rows.size() * 4));
    CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));

    static const int modes[] = {0, 2, 6, 7, 8, 5};
    for (int part = 0; part < 12; ++part) {
        const int mode = modes[part >> 1];
        if ((part & 1) && !at_scale)
            continue;
        if (mode == 5 && (part & 1))
            continue;                                          // (this loop's rate run faulted the GPU)
        // part 0: 8 waves x 2 chunks, checked; part 1: 7 waves per SIMD on 256 CUs x 16 chunks, timed
        const int nwaves = (part & 1) == 0 ? 8 : 256 * 4 * 7 * 2;
        const int nchunks = (part & 1) == 0 ? 2 : 16;
        std::vector<Hit> hits((size_t)nwaves * nchunks * 64);
        for (auto &h : hits) {
            const int r = rand() % 3, c = rand() % 7;
            h.idx = r * 8 + c;
            for (int k = 0; k < 4; ++k)
                h.w[k] = (float)(rand() % 1025) / 1024.f;
            h.row = rand() % nrows;
        }
        Hit *d_hits;
        float *d_out;
        long long *d_cyc;
        CHECK(hipMalloc(&d_hits, hits.size() * sizeof(Hit)));
        CHECK(hipMalloc(&d_out, (size_t)nwaves * 32 * 64 * 4));
        CHECK(hipMalloc(&d_cyc, (size_t)nwaves * 8));
        CHECK(hipMemcpy(d_hits, hits.data(), hits.size() * sizeof(Hit), hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        float ms = 0.f;
        for (int rep = 0; rep < ((part & 1) ? 3 : 1); ++rep) {
            CHECK(hipEventRecord(e0));
            if (mode == 0)
                hipLaunchKernelGGL(probe<0>, dim3(nwaves / 4), dim3(256), 0, 0, d_hits, nchunks, d_rows, d_out, d_cyc);
            else if (mode == 1)
                hipLaunchKernelGGL(probe<1>, dim3(nwaves / 4), dim3(256), 0, 0, d_hits, nchunks, d_rows, d_out, d_cyc);
            else if (mode == 3)
                hipLaunchKernelGGL(probe<3>, dim3(nwaves / 4), dim3(256), 0, 0, d_hits, nchunks, d_rows, d_out, d_cyc);
            else if (mode == 6)
                hipLaunchKernelGGL(probe<6>, dim3(nwaves / 4), dim3(256), 0, 0, d_hits, nchunks, d_rows, d_out, d_cyc);
            else if (mode == 7)
                hipLaunchKernelGGL(probe<7>, dim3(nwaves / 4), dim3(256), 0, 0, d_hits, nchunks, d_rows, d_out, d_cyc);
            else if (mode == 8)
                hipLaunchKernelGGL(probe<8>, dim3(nwaves / 4), dim3(256), 0, 0, d_hits, nchunks, d_rows, d_out, d_cyc);
            else if (mode == 5)
                hipLaunchKernelGGL(probe5, dim3(nwaves / 4), dim3(256), 0, 0, d_hits, nchunks, d_rows, d_out, d_cyc, nrows - 1);
            else if (mode == 4)
                hipLaunchKernelGGL(probe<4>, dim3(nwaves / 4), dim3(256), 0, 0, d_hits, nchunks, d_rows, d_out, d_cyc);
            else
                hipLaunchKernelGGL(probe<2>, dim3(nwaves / 4), dim3(256), 0, 0, d_hits, nchunks, d_rows, d_out, d_cyc);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::vector<float> out((size_t)nwaves * 32 * 64);
        CHECK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
        if ((part & 1) == 0 && mode != 1 && mode != 3) {
            long long bad = 0;
            for (int w = 0; w < nwaves; ++w) {
                std::vector<float> ref(32 * 64, 0.f);
                for (int i = 0; i < nchunks * 64; ++i) {
                    const Hit &h = hits[(size_t)w * nchunks * 64 + i];
                    const int off[4] = {0, 1, 8, 9};
                    for (int k = 0; k < 4; ++k)
                        for (int ch = 0; ch < 64; ++ch)
                            ref[(h.idx + off[k]) * 64 + ch] =
                                __builtin_fmaf(h.w[k], rows[(size_t)h.row * 64 + ch], ref[(h.idx + off[k]) * 64 + ch]);
                }
                for (int j = 0; j < 32 * 64; ++j)
                    if (memcmp(&ref[j], &out[(size_t)w * 32 * 64 + j], 4) != 0 && bad++ < 5)
                        printf("wave %d pixel %d ch %d: got %g, expected %g\n", w, j / 64, j % 64,
                               out[(size_t)w * 32 * 64 + j], ref[j]);
            }
            printf(bad ? "mode %d index-mode fma: %lld MISMATCHES\n" : "mode %d index-mode fma: dst and src2 relative, bit-exact (%lld mismatches)\n", mode, bad);
        } else if (part & 1) {
            const double nhits = (double)nwaves * nchunks * 64;
            // 1024 SIMDs; the shader clock is not read here: report time per hit per SIMD in ns
            printf("mode %d rate: %.0f hits in %.1f us: %.2f ns per hit per SIMD (x 2.4 GHz = %.1f cycles)\n", mode, nhits, ms * 1e3,
                   ms * 1e6 / (nhits / 1024.0), ms * 1e6 / (nhits / 1024.0) * 2.4);
        }
        CHECK(hipFree(d_hits));
        CHECK(hipFree(d_out));
        CHECK(hipFree(d_cyc));
    }
    return 0;
}
