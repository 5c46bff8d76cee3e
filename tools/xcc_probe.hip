// Which XCD does block i of a 1-D grid run on?  Prints, for several grid sizes, how many
// distinct XCC ids each residue class (blockIdx % 8) sees - 1 everywhere means "blocks b and
// b+8 share an XCD", the assumption behind msmv_xcd_remap().
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
__global__ void probe(int *out)
{
    if (threadIdx.x == 0) {
        int id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        out[blockIdx.x] = id & 0xF;
    }
    // keep the block alive a little so that blocks spread over all CUs
    for (int i = 0; i < 2000; ++i) asm volatile("s_nop 15");
}
int main()
{
    for (int nblk : {64, 1024, 4800, 16320}) {
        for (int threads : {256, 512}) {
            int *d; hipMalloc(&d, nblk * sizeof(int));
            hipLaunchKernelGGL(probe, dim3(nblk), dim3(threads), 0, 0, d);
            std::vector<int> h(nblk);
            hipMemcpy(h.data(), d, nblk * sizeof(int), hipMemcpyDeviceToHost);
            int worst = 0; std::set<int> all;
            for (int r = 0; r < 8; ++r) {
                std::set<int> s;
                for (int i = r; i < nblk; i += 8) { s.insert(h[i]); all.insert(h[i]); }
                worst = std::max<int>(worst, s.size());
            }
            int mism = 0;
            for (int i = 8; i < nblk; ++i) mism += h[i] != h[i - 8];
            printf("grid %5d x %3d: %zu XCC ids, max ids per residue class %d, blocks differing from block-8: %d; first 16:", nblk, threads, all.size(), worst, mism);
            for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
            printf("\n");
            hipFree(d);
        }
    }
    return 0;
}
