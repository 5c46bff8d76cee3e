// Probe: does a workgroup barrier release when some waves of the workgroup have already ended?
// Waves 1 and 2 of every 4-wave workgroup return at once (or after a long spin); waves 0 and 3
// exchange a value through LDS across two barriers.  Prints OK when every exchange is right.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void __launch_bounds__(256) k(int *out, int spin)
{
    __shared__ int box[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 1)
        return;                                   // ends before the others reach the barrier
    if (wave == 2) {                              // ends late: the others are already waiting
        long long t0 = clock64();
        while (clock64() - t0 < spin) {}
        return;
    }
    if (lane == 0)
        box[wave] = 100 + wave + blockIdx.x;
    __syncthreads();
    const int other = box[3 - wave];
    __syncthreads();
    if (lane == 0)
        out[blockIdx.x * 2 + (wave ? 1 : 0)] = other;
}

int main()
{
    const int nb = 4096;
    int *d;
    hipMalloc(&d, nb * 2 * sizeof(int));
    hipMemset(d, 0, nb * 2 * sizeof(int));
    k<<<nb, 256>>>(d, 200000);
    hipError_t e = hipDeviceSynchronize();
    std::vector<int> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 2 * sizeof(int), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < nb; ++b)
        bad += (h[b * 2] != 103 + b) + (h[b * 2 + 1] != 100 + b);
    printf("%s, %d wrong of %d\n", hipGetErrorString(e), bad, nb * 2);
    return bad != 0;
}
