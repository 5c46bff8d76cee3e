// Probe: address of a static __device__ array through hipGetSymbolAddress, used as a kernel argument.
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifndef PROBE_BYTES
#define PROBE_BYTES (32u << 20)
#endif
static __device__ __align__(256) unsigned char scratch[PROBE_BYTES];

__global__ void fill(unsigned *p, unsigned n) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i * 3u;
}

int main() {
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    printf("hipGetSymbolAddress: %s ptr=%p align256=%d\n", hipGetErrorString(e), ptr, (int)(((unsigned long long)ptr & 255) == 0));
    if (e != hipSuccess) return 1;
    size_t sz = 0;
    e = hipGetSymbolSize(&sz, HIP_SYMBOL(scratch));
    printf("size: %s %zu\n", hipGetErrorString(e), sz);
    unsigned n = PROBE_BYTES / 4;
    fill<<<(n + 255) / 256, 256>>>((unsigned *)ptr, n);
    e = hipDeviceSynchronize();
    printf("kernel: %s\n", hipGetErrorString(e));
    unsigned back[4];
    e = hipMemcpy(back, (unsigned *)ptr + n - 4, 16, hipMemcpyDeviceToHost);
    printf("copy: %s last=%u expect=%u\n", hipGetErrorString(e), back[3], (n - 1) * 3u);
    return 0;
}
