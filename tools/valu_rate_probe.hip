// Instruction-rate probe for gfx950: what a SIMD spends on the pieces of a sparse hit update, with
// every SIMD holding WPS waves that run the same register-only loop (no memory in the loop).
//   hipcc -O3 --offload-arch=gfx950 -Wno-inline-asm -o tools/valu_rate_probe tools/valu_rate_probe.hip
// Prints ns per loop body per SIMD (= per wave body / waves per SIMD) for each body, and the ratio of
// s_memtime to s_memrealtime ticks (the latter is a constant 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x32 __attribute__((ext_vector_type(32)));

#define FMA4 "v_fma_f32 v32, s20, %1, v32\n\tv_fma_f32 v33, s21, %1, v33\n\tv_fma_f32 v40, s22, %1, v40\n\tv_fma_f32 v41, s23, %1, v41\n\t"
#define RL6 "v_readlane_b32 s20, %2, s24\n\tv_readlane_b32 s21, %2, s24\n\tv_readlane_b32 s22, %2, s24\n\tv_readlane_b32 s23, %2, s24\n\tv_readlane_b32 s25, %2, s24\n\tv_readlane_b32 s26, %2, s24\n\t"
#define RL2 "v_readlane_b32 s25, %2, s24\n\tv_readlane_b32 s26, %2, s24\n\t"
#define SALU6 "s_lshl_b64 s[28:29], s[26:27], 8\n\ts_add_u32 s28, s28, s30\n\ts_addc_u32 s29, s29, s31\n\ts_add_i32 s24, s24, 1\n\ts_and_b32 s24, s24, 63\n\ts_and_b32 s25, s25, 15\n\t"

template <int BODY>
__global__ void __launch_bounds__(256)
k(int iters, float *out, long long *ticks)
{
    f32x32 acc = {};
    float g = (float)threadIdx.x;
    int v = threadIdx.x & 15;
    const long long t0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        // every body is unrolled four times inside one statement
#define X4(S) S S S S
        if (BODY == 0)        // 4 multiply-adds, fixed registers
            asm volatile("s_mov_b32 s20, 0.5\n\ts_mov_b32 s21, 0.5\n\ts_mov_b32 s22, 0.5\n\ts_mov_b32 s23, 0.5\n\t" X4(FMA4)
                         : "+{v[32:63]}"(acc) : "v"(g), "v"(v) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "m0");
        else if (BODY == 1)   // index mode on / 4 multiply-adds / off
            asm volatile("s_mov_b32 s20, 0.5\n\ts_mov_b32 s21, 0.5\n\ts_mov_b32 s22, 0.5\n\ts_mov_b32 s23, 0.5\n\ts_mov_b32 s25, 3\n\t"
                         X4("s_set_gpr_idx_on s25, 0xc\n\t" FMA4 "s_set_gpr_idx_off\n\t")
                         : "+{v[32:63]}"(acc) : "v"(g), "v"(v) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "m0");
        else if (BODY == 2)   // 6 readlanes
            asm volatile("s_mov_b32 s24, 5\n\t" X4(RL6)
                         : "+{v[32:63]}"(acc) : "v"(g), "v"(v) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "m0");
        else if (BODY == 3)   // 6 readlanes + 4 multiply-adds
            asm volatile("s_mov_b32 s24, 5\n\t" X4(RL6 FMA4)
                         : "+{v[32:63]}"(acc) : "v"(g), "v"(v) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "m0");
        else if (BODY == 4)   // the whole hit: 6 readlanes, 6 scalar instructions, index mode, 4 multiply-adds
            asm volatile("s_mov_b32 s24, 5\n\ts_mov_b32 s27, 0\n\t"
                         X4(RL6 SALU6 "s_set_gpr_idx_on s25, 0xc\n\t" FMA4 "s_set_gpr_idx_off\n\t")
                         : "+{v[32:63]}"(acc) : "v"(g), "v"(v) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "m0");
        else if (BODY == 5)   // 2 readlanes + index mode + 4 multiply-adds with vector weights
            asm volatile("s_mov_b32 s24, 5\n\t"
                         X4(RL2 "s_and_b32 s25, s25, 15\n\ts_set_gpr_idx_on s25, 0xc\n\t"
                            "v_fma_f32 v32, %1, %1, v32\n\tv_fma_f32 v33, %1, %1, v33\n\tv_fma_f32 v40, %1, %1, v40\n\tv_fma_f32 v41, %1, %1, v41\n\t"
                            "s_set_gpr_idx_off\n\t")
                         : "+{v[32:63]}"(acc) : "v"(g), "v"(v) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "m0");
        else if (BODY == 6)   // 4 multiply-adds with vector weights, fixed registers, dependent chain broken over 4 registers
            asm volatile(X4("v_fma_f32 v32, %1, %1, v32\n\tv_fma_f32 v33, %1, %1, v33\n\tv_fma_f32 v40, %1, %1, v40\n\tv_fma_f32 v41, %1, %1, v41\n\t")
                         : "+{v[32:63]}"(acc) : "v"(g), "v"(v) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "m0");
        else if (BODY == 7)   // 16 independent multiply-adds (no back-to-back dependence on one register)
            asm volatile("v_fma_f32 v32, %1, %1, v32\n\tv_fma_f32 v33, %1, %1, v33\n\tv_fma_f32 v34, %1, %1, v34\n\tv_fma_f32 v35, %1, %1, v35\n\t"
                         "v_fma_f32 v36, %1, %1, v36\n\tv_fma_f32 v37, %1, %1, v37\n\tv_fma_f32 v38, %1, %1, v38\n\tv_fma_f32 v39, %1, %1, v39\n\t"
                         "v_fma_f32 v40, %1, %1, v40\n\tv_fma_f32 v41, %1, %1, v41\n\tv_fma_f32 v42, %1, %1, v42\n\tv_fma_f32 v43, %1, %1, v43\n\t"
                         "v_fma_f32 v44, %1, %1, v44\n\tv_fma_f32 v45, %1, %1, v45\n\tv_fma_f32 v46, %1, %1, v46\n\tv_fma_f32 v47, %1, %1, v47\n\t"
                         : "+{v[32:63]}"(acc) : "v"(g), "v"(v) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "m0");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 32; ++p)
        s += acc[p];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        ticks[blockIdx.x * 2] = t1 - t0;
        ticks[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

template <int BODY>
void run(const char *what, int wps)
{
    const int nblk = 256 * wps;          // 4 waves per block: one per SIMD of a CU when blocks spread evenly
    const int iters = 4096;
    float *out;
    long long *ticks;
    CHECK(hipMalloc(&out, (size_t)nblk * 256 * 4));
    CHECK(hipMalloc(&ticks, (size_t)nblk * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<BODY>, dim3(nblk), dim3(256), 0, 0, iters, out, ticks);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    long long t[2];
    CHECK(hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost));
    const double bodies_per_simd = (double)iters * 4 * wps;      // 4 unrolled bodies per iteration
    printf("%-64s %d waves/SIMD: %7.1f us, %6.2f ns per body per SIMD; memtime/realtime ticks %.2f\n", what, wps, ms * 1e3,
           ms * 1e6 / bodies_per_simd, (double)t[0] / (double)t[1]);
    CHECK(hipFree(out));
    CHECK(hipFree(ticks));
}

int main()
{
    for (int wps : {1, 2, 8}) {
        run<7>("16 independent v_fma (per 4: divide by 4)", wps);
        run<6>("4 v_fma, vector weights", wps);
        run<0>("4 v_fma, scalar weights", wps);
        run<1>("idx on + 4 v_fma + idx off", wps);
        run<2>("6 v_readlane", wps);
        run<3>("6 v_readlane + 4 v_fma", wps);
        run<4>("6 v_readlane + 6 salu + idx on + 4 v_fma + idx off", wps);
        run<5>("2 v_readlane + 1 salu + idx on + 4 v_fma (vector weights) + off", wps);
    }
    return 0;
}
