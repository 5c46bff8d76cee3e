// Probe: rate of store-only kernels over a 738 MB buffer shaped like 256 level-0 slabs
// (64 x 176 pixels x 256 B): sequential fill vs the d-feat kernel's patch pattern (PH x PW pixels per
// wave, a half-wave storing 128 contiguous bytes per instruction), plain vs non-temporal stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

constexpr int H = 64, W = 160, SLABS = 282;     // (W a multiple of 32 so that every shape tiles it)

template <bool NT> __device__ __forceinline__ void st(float *p, float v)
{
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <bool NT>
__global__ void __launch_bounds__(256) fill_seq(float *dst, size_t n4)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f4 v = {1.f, 2.f, 3.f, 4.f};
    if (i < n4) {
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(dst) + i);
        else reinterpret_cast<f4 *>(dst)[i] = v;
    }
}

// one wave per patch of PH x PW = 32 pixels; 4 waves per workgroup take 4 consecutive patches
template <int PH, int PW, bool NT>
__global__ void __launch_bounds__(256) fill_patch(float *dst)
{
    const int lane = threadIdx.x & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = threadIdx.x >> 6;
    constexpr int PCOLS = W / PW, PROWS = H / PH, PATCHES = PCOLS * PROWS;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, within = bid >> 3, q = gridDim.x >> 3;
    const int id = xcd * q + within;                       // contiguous ranges per XCD (grid is a multiple of 8)
    const int slab = id / (PATCHES / 4);
    const int patch = (id % (PATCHES / 4)) * 4 + wave;
    const int prow = patch / PCOLS, pcol = patch % PCOLS;
    float *base = dst + ((size_t)slab * H * W + (size_t)prow * PH * W + pcol * PW) * 64;
    const float v = (float)lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2);              // + 4 * half
        const int pr = i / PW, pc = i % PW + 4 * half;
        float *o = base + ((size_t)pr * W + pc) * 64 + l32;
        st<NT>(o, v);
        st<NT>(o + 32, v);
    }
}

// the same patch, but every lane stores 16 bytes: 16 lanes cover one pixel's 256 B, a wave
// instruction four pixels (what a transposition through LDS would allow)
template <int PH, int PW, bool NT>
__global__ void __launch_bounds__(256) fill_patch_x4(float *dst)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, l16 = lane & 15, sub = lane >> 4;
    const int wave = threadIdx.x >> 6;
    constexpr int PCOLS = W / PW, PROWS = H / PH, PATCHES = PCOLS * PROWS;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, within = bid >> 3, q = gridDim.x >> 3;
    const int id = xcd * q + within;
    const int slab = id / (PATCHES / 4);
    const int patch = (id % (PATCHES / 4)) * 4 + wave;
    const int prow = patch / PCOLS, pcol = patch % PCOLS;
    float *base = dst + ((size_t)slab * H * W + (size_t)prow * PH * W + pcol * PW) * 64;
    const f4 v = {(float)lane, 1.f, 2.f, 3.f};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = r * 4 + sub;                          // 4 consecutive pixels per instruction
        const int pr = i / PW, pc = i % PW;
        f4 *o = reinterpret_cast<f4 *>(base + ((size_t)pr * W + pc) * 64) + l16;
        if (NT) __builtin_nontemporal_store(v, o); else *o = v;
    }
}

// 1 x 32 pixels per wave (8 KiB contiguous), the 4 waves of a workgroup stacked vertically: the
// workgroup covers 4 rows x 32 pixels like four 4 x 8 patches side by side, but each wave writes
// one row of it (what exchanging the accumulators between the waves through LDS would allow)
template <bool NT>
__global__ void __launch_bounds__(256) fill_rows_of_wg(float *dst)
{
    const int lane = threadIdx.x & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = threadIdx.x >> 6;
    constexpr int PCOLS = W / 32, PROWS = H / 4, TILES = PCOLS * PROWS;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, within = bid >> 3, q = gridDim.x >> 3;
    const int id = xcd * q + within;
    const int slab = id / TILES, tile = id % TILES;
    const int trow = tile / PCOLS, tcol = tile % PCOLS;
    float *base = dst + ((size_t)slab * H * W + (size_t)(trow * 4 + wave) * W + tcol * 32) * 64;
    const float v = (float)lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float *o = base + (size_t)(r * 2 + half) * 64 + l32;
        st<NT>(o, v);
        st<NT>(o + 32, v);
    }
}

// the two patch shapes again, but every wave first waits a pseudo-random 0 .. `spread` cycles (the waves
// of the real kernel reach their stores at different times) and holds `lds` bytes of LDS per workgroup
template <int PH, int PW, bool NT, bool BARRIER = false>
__global__ void __launch_bounds__(256) fill_patch_delayed(float *dst, int spread)
{
    const int lane = threadIdx.x & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = threadIdx.x >> 6;
    constexpr int PCOLS = W / PW, PROWS = H / PH, PATCHES = PCOLS * PROWS;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, within = bid >> 3, q = gridDim.x >> 3;
    const int id = xcd * q + within;
    const int slab = id / (PATCHES / 4);
    const int patch = (id % (PATCHES / 4)) * 4 + wave;
    const int prow = patch / PCOLS, pcol = patch % PCOLS;
    float *base = dst + ((size_t)slab * H * W + (size_t)prow * PH * W + pcol * PW) * 64;
    const unsigned h = ((unsigned)(bid * 4 + wave) * 2654435761u) >> 16;
    const long long wait = spread ? (long long)(h % (unsigned)spread) : 0;
    const long long t0 = clock64();
    while (clock64() - t0 < wait) {}
    if (BARRIER)
        __syncthreads();                       // the workgroup's four waves store together
    const float v = (float)lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2);
        const int pr = i / PW, pc = i % PW + 4 * half;
        float *o = base + ((size_t)pr * W + pc) * 64 + l32;
        st<NT>(o, v);
        st<NT>(o + 32, v);
    }
}

// cache-policy sweep on the 4 x 8 patch with desynchronised waves: the policy bits of the vector store
template <int POL> __device__ __forceinline__ void st_pol(float *p, float v)
{
    if (POL == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 5) asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 6) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (POL == 7) asm volatile("global_store_dword %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
}

template <int POL>
__global__ void __launch_bounds__(256) fill_patch_pol(float *dst, int spread)
{
    constexpr int PH = 4, PW = 8;
    const int lane = threadIdx.x & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = threadIdx.x >> 6;
    constexpr int PCOLS = W / PW, PROWS = H / PH, PATCHES = PCOLS * PROWS;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, within = bid >> 3, q = gridDim.x >> 3;
    const int id = xcd * q + within;
    const int slab = id / (PATCHES / 4);
    const int patch = (id % (PATCHES / 4)) * 4 + wave;
    const int prow = patch / PCOLS, pcol = patch % PCOLS;
    float *base = dst + ((size_t)slab * H * W + (size_t)prow * PH * W + pcol * PW) * 64;
    const unsigned h = ((unsigned)(bid * 4 + wave) * 2654435761u) >> 16;
    const long long wait = spread ? (long long)(h % (unsigned)spread) : 0;
    const long long t0 = clock64();
    while (clock64() - t0 < wait) {}
    const float v = (float)lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2);
        const int pr = i / PW, pc = i % PW + 4 * half;
        float *o = base + ((size_t)pr * W + pc) * 64 + l32;
        st_pol<POL>(o, v);
        st_pol<POL>(o + 32, v);
    }
}

template <typename F> float time_it(F f, int reps = 20)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> t;
    for (int i = 0; i < reps + 3; ++i) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (i >= 3) t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    const size_t bytes = (size_t)SLABS * H * W * 256;
    float *d; hipMalloc(&d, bytes);
    const size_t n4 = bytes / 16;
    auto rep = [&](const char *name, float us) { printf("%-56s %8.1f us  %6.2f TB/s\n", name, us, bytes / us / 1e6); };
    rep("sequential plain", time_it([&] { fill_seq<false><<<(unsigned)((n4 + 255) / 256), 256>>>(d, n4); }));
    rep("sequential nt", time_it([&] { fill_seq<true><<<(unsigned)((n4 + 255) / 256), 256>>>(d, n4); }));
#define PATCH(PH, PW)                                                                                  \
    rep("patch " #PH "x" #PW " plain", time_it([&] { fill_patch<PH, PW, false><<<SLABS * (H / PH) * (W / PW) / 4, 256>>>(d); })); \
    rep("patch " #PH "x" #PW " nt", time_it([&] { fill_patch<PH, PW, true><<<SLABS * (H / PH) * (W / PW) / 4, 256>>>(d); }));
    PATCH(4, 8)
    PATCH(2, 16)
    PATCH(1, 32)
#define PATCH4(PH, PW)                                                                                 \
    rep("patch " #PH "x" #PW " 16B/lane plain", time_it([&] { fill_patch_x4<PH, PW, false><<<SLABS * (H / PH) * (W / PW) / 4, 256>>>(d); })); \
    rep("patch " #PH "x" #PW " 16B/lane nt", time_it([&] { fill_patch_x4<PH, PW, true><<<SLABS * (H / PH) * (W / PW) / 4, 256>>>(d); }));
    PATCH4(4, 8)
    PATCH4(2, 16)
    rep("4 rows x 32 px per WG, wave = row plain", time_it([&] { fill_rows_of_wg<false><<<SLABS * (H / 4) * (W / 32), 256>>>(d); }));
    rep("4 rows x 32 px per WG, wave = row nt", time_it([&] { fill_rows_of_wg<true><<<SLABS * (H / 4) * (W / 32), 256>>>(d); }));
    // the same patch kernel with the d-feat kernel's LDS footprint per workgroup (21.5 KiB: 7 workgroups per CU)
    rep("patch 4x8 nt + 21.5 KiB LDS/WG", time_it([&] { fill_patch<4, 8, true><<<SLABS * (H / 4) * (W / 8) / 4, 256, 21504>>>(d); }));
    rep("patch 4x8 nt + 40 KiB LDS/WG", time_it([&] { fill_patch<4, 8, true><<<SLABS * (H / 4) * (W / 8) / 4, 256, 40960>>>(d); }));
    rep("patch 1x32 nt + 21.5 KiB LDS/WG", time_it([&] { fill_patch<1, 32, true><<<SLABS * (H / 1) * (W / 32) / 4, 256, 21504>>>(d); }));
    for (int spread : {0, 2000, 8000}) {
        char name[96];
        snprintf(name, sizeof name, "patch 4x8 nt, waves delayed 0..%d cycles", spread);
        rep(name, time_it([&] { fill_patch_delayed<4, 8, true><<<SLABS * (H / 4) * (W / 8) / 4, 256>>>(d, spread); }));
        snprintf(name, sizeof name, "patch 1x32 nt, waves delayed 0..%d cycles", spread);
        rep(name, time_it([&] { fill_patch_delayed<1, 32, true><<<SLABS * (H / 1) * (W / 32) / 4, 256>>>(d, spread); }));
        snprintf(name, sizeof name, "patch 4x8 nt, delayed 0..%d, barrier before stores", spread);
        rep(name, time_it([&] { fill_patch_delayed<4, 8, true, true><<<SLABS * (H / 4) * (W / 8) / 4, 256>>>(d, spread); }));
        snprintf(name, sizeof name, "patch 1x32 nt, delayed 0..%d, barrier before stores", spread);
        rep(name, time_it([&] { fill_patch_delayed<1, 32, true, true><<<SLABS * (H / 1) * (W / 32) / 4, 256>>>(d, spread); }));
    }
    {
        const unsigned grid = SLABS * (H / 4) * (W / 8) / 4;
#define POL(N, NAME) rep("patch 4x8 delayed 0..2000, policy " NAME, time_it([&] { fill_patch_pol<N><<<grid, 256>>>(d, 2000); }));
        POL(0, "(none)") POL(1, "nt") POL(2, "sc1") POL(3, "sc0 sc1") POL(4, "sc1 nt") POL(5, "sc0 sc1 nt") POL(6, "sc0") POL(7, "sc0 nt")
        POL(0, "(none)") POL(1, "nt")
    }
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(e));
    return 0;
}
