// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access pattern of the msmv
// kernels: rows of 16 lanes read 256-byte vectors (16 B per lane) at random 256-byte-aligned
// positions of a table far larger than the 256 MiB Infinity Cache, each position exactly
// once, and write a 256-byte result vector per 16 reads.  Known bytes are printed; compare
// with the counters of kernel `calib_gather`:
//   hipcc --offload-arch=gfx950 -O3 -o calib_fetch tools/calib_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE -- ./calib_fetch ;  rocprofv3 --pmc WRITE_SIZE -- ./calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) calib_gather(const float4 *table, const int *rows, int ngroups, float4 *out)
{
    const int lane = threadIdx.x & 63, sub = lane >> 4, l16 = lane & 15;
    const int g = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub;   // one row of lanes per group of 16 vectors
    if (g >= ngroups)
        return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
        v[k] = table[(size_t)rows[g * 16 + k] * 16 + l16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w;
    }
    out[(size_t)g * 16 + l16] = acc;
}

int main()
{
    const size_t nrows = (size_t)1 << 22;             // 4M rows x 256 B = 1 GiB table
    const int nread = 1 << 21;                        // 2M distinct rows read = 512 MiB
    const int ngroups = nread / 16;
    float4 *table, *out;
    int *rows;
    CHECK(hipMalloc(&table, nrows * 256));
    CHECK(hipMalloc(&out, (size_t)ngroups * 256));
    CHECK(hipMalloc(&rows, nread * sizeof(int)));
    CHECK(hipMemset(table, 0, nrows * 256));
    std::vector<int> perm(nrows);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 rng(1);
    for (int i = 0; i < nread; ++i) {                 // partial Fisher-Yates: distinct rows
        std::uniform_int_distribution<size_t> d(i, nrows - 1);
        std::swap(perm[i], perm[d(rng)]);
    }
    CHECK(hipMemcpy(rows, perm.data(), nread * sizeof(int), hipMemcpyHostToDevice));
    const int nblk = (ngroups + 15) / 16;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int it = 0; it < 3; ++it) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(calib_gather, dim3(nblk), dim3(256), 0, 0, table, rows, ngroups, out);
        CHECK(hipEventRecord(b));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        const double rd = (double)nread * 256 + (double)nread * 4, wr = (double)ngroups * 256;
        printf("launch %d: %.1f us  known read bytes %.0f (%.1f MiB)  known write bytes %.0f  -> %.2f TB/s\n",
               it, ms * 1e3, rd, rd / 1048576.0, wr, (rd + wr) / ms / 1e9);
    }
    return 0;
}
