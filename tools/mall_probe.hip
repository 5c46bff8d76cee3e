// Probe: what the 256 MiB Infinity Cache does for streamed writes and re-reads.
//   writes: one sequential fill of S bytes after an idle gap (nothing dirty anywhere), plain and
//           non-temporal stores - a rate far above the HBM write rate for S below the cache size
//           means the cache absorbs the burst and drains it later;
//   reads:  a sequential read of S bytes right after another read of the same bytes (re-read rate)
//           and after a read of a different 1 GiB buffer (cold rate).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
#include <algorithm>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(256) fill(f4 *dst, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    if (i < n4) {
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}

template <bool NT>
__global__ void __launch_bounds__(256) rd(const f4 *src, size_t n4, float *sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        const f4 v = NT ? __builtin_nontemporal_load(src + i) : src[i];
        if (v.x == 12345.678f)                    // never true: keeps the load
            *sink = v.y;
    }
}

static float timed(hipEvent_t a, hipEvent_t b)
{
    float ms;
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f;
}

int main()
{
    const size_t big = (size_t)1 << 30;
    f4 *buf, *other;
    float *sink;
    hipMalloc(&buf, big); hipMalloc(&other, big); hipMalloc(&sink, 4);
    hipMemset(buf, 0, big); hipMemset(other, 0, big);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const size_t sizes_mb[] = {32, 64, 128, 192, 256, 384, 512, 768};
    printf("%8s | %22s | %22s | %22s | %22s\n", "MB", "fill plain after idle", "fill nt after idle", "re-read (plain)", "cold read (plain)");
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20, n4 = bytes / 16;
        const unsigned grid = (unsigned)((n4 + 255) / 256);
        std::vector<float> t[4];
        for (int rep = 0; rep < 7; ++rep) {
            // writes after an idle gap: first read `other` so that the cache holds clean lines only
            for (int nt = 0; nt < 2; ++nt) {
                rd<false><<<(unsigned)(big / 16 / 256), 256>>>(other, big / 16, sink);
                hipDeviceSynchronize();
                usleep(300);
                hipEventRecord(a);
                if (nt) fill<true><<<grid, 256>>>(buf, n4); else fill<false><<<grid, 256>>>(buf, n4);
                hipEventRecord(b);
                t[nt].push_back(timed(a, b));
            }
            // re-read: read S bytes, then time a second read of them
            rd<false><<<(unsigned)(big / 16 / 256), 256>>>(other, big / 16, sink);
            rd<false><<<grid, 256>>>(buf, n4, sink);
            hipEventRecord(a);
            rd<false><<<grid, 256>>>(buf, n4, sink);
            hipEventRecord(b);
            t[2].push_back(timed(a, b));
            // cold: the other buffer went through the caches in between
            rd<false><<<(unsigned)(big / 16 / 256), 256>>>(other, big / 16, sink);
            hipEventRecord(a);
            rd<false><<<grid, 256>>>(buf, n4, sink);
            hipEventRecord(b);
            t[3].push_back(timed(a, b));
        }
        printf("%8zu |", mb);
        for (int k = 0; k < 4; ++k) {
            std::sort(t[k].begin(), t[k].end());
            const float us = t[k][t[k].size() / 2];
            printf(" %9.1f us %6.2f TB/s |", us, bytes / us / 1e6);
        }
        printf("\n");
    }
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
