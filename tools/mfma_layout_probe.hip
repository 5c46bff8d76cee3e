// Probe: operand / result layout of v_mfma_f32_4x4x1_16b_f32 (16 blocks of a 4x4 += 4x1 * 1x4 update).
// A = lane number, B = 100 + lane number: D[lane][v] = A[src lane] * B[src lane] tells which lanes meet.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out)
{
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), (float)(1000 * (l + 1)), c, 0, 0, 0);
    for (int v = 0; v < 4; ++v)
        out[l * 4 + v] = c[v];
}
int main()
{
    float *d, h[256];
    hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    // expectation: D[lane l][v] = A[lane 4 * (l / 4) + v] * B[lane l]
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            const long long p = (long long)(h[l * 4 + v] + 0.5f);
            const int ea = (l / 4) * 4 + v + 1, eb = l + 1;
            if (p != 1000LL * ea * eb) {
                ok = 0;
                printf("lane %d v %d: got %lld, expected %lld\n", l, v, p, 1000LL * ea * eb);
            }
        }
    printf(ok ? "layout as expected: D[lane l][v] = A[lane 4*(l/4)+v] * B[lane l]\n" : "layout differs\n");
    return 0;
}
