// Diagnostic for the pieces of the candidate hit loop of gpr_idx_probe.hip, one wave, every value
// dumped and compared on the host: (1) DPP row_newbcast on a 32-bit move, (2) DPP row_shl packing of
// four bytes, (3) gradient rows staged global -> LDS with per-lane addresses and read back.
//   hipcc -O3 --offload-arch=gfx950 -Wno-inline-asm -o tools/dpp_stage_probe tools/dpp_stage_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void *lds_ptr_t;
__device__ __forceinline__ void load_to_lds_b128(const float *src, const float4 *dst)
{
    const unsigned lds = (unsigned)(uintptr_t)(lds_ptr_t)dst;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory", "m0");
}

constexpr int NROWS = 256;

__global__ void __launch_bounds__(64)
k(const int *rowidx, const float *rows, int *out_i, float *out_f)
{
    __shared__ float4 stage[256];
    __shared__ int rq[16];
    const int lane = threadIdx.x;
    // (1) row_newbcast:K of x = lane * 10
    int x = lane * 10, b3, b12;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf"
                 : "=&v"(b3), "=&v"(b12) : "v"(x));
    out_i[lane] = b3;
    out_i[64 + lane] = b12;
    // (2) bytes of lanes i .. i+3 of the row
    const int idx = lane & 31;
    int pk = idx;
    pk |= __builtin_amdgcn_update_dpp(0, idx, 0x101, 0xF, 0xF, true) << 8;
    pk |= __builtin_amdgcn_update_dpp(0, idx, 0x102, 0xF, 0xF, true) << 16;
    pk |= __builtin_amdgcn_update_dpp(0, idx, 0x103, 0xF, 0xF, true) << 24;
    out_i[128 + lane] = pk;
    // (3) 16 rows staged by four direct loads; row numbers pass through LDS like in the hit loop
    if (lane < 16)
        rq[lane] = rowidx[lane] & (NROWS - 1);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = rq[j * 4 + (lane >> 4)];
        load_to_lds_b128(rows + (size_t)row * 64 + (lane & 15) * 4, stage + j * 64);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float *sf = reinterpret_cast<const float *>(stage) + lane;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
        out_f[kk * 64 + lane] = sf[kk * 64];
}

int main()
{
    std::vector<int> rowidx(16);
    std::vector<float> rows((size_t)NROWS * 64);
    srand(3);
    for (auto &r : rowidx)
        r = rand() % NROWS;
    for (size_t i = 0; i < rows.size(); ++i)
        rows[i] = (float)i;
    int *d_ri, *d_oi;
    float *d_rows, *d_of;
    CHECK(hipMalloc(&d_ri, 64));
    CHECK(hipMalloc(&d_rows, rows.size() * 4));
    CHECK(hipMalloc(&d_oi, 192 * 4));
    CHECK(hipMalloc(&d_of, 1024 * 4));
    CHECK(hipMemcpy(d_ri, rowidx.data(), 64, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_of, 0xFF, 1024 * 4));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_ri, d_rows, d_oi, d_of);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    std::vector<int> oi(192);
    std::vector<float> of(1024);
    CHECK(hipMemcpy(oi.data(), d_oi, 192 * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(of.data(), d_of, 1024 * 4, hipMemcpyDeviceToHost));
    int bad1 = 0, bad2 = 0, bad3 = 0;
    for (int l = 0; l < 64; ++l) {
        bad1 += oi[l] != ((l & ~15) + 3) * 10;
        bad1 += oi[64 + l] != ((l & ~15) + 12) * 10;
        int want = 0;
        for (int b = 0; b < 4; ++b)
            want |= (((l & 15) + b < 16) ? ((l + b) & 31) : 0) << (8 * b);
        bad2 += oi[128 + l] != want;
    }
    for (int kk = 0; kk < 16; ++kk)
        for (int l = 0; l < 64; ++l)
            bad3 += of[kk * 64 + l] != rows[(size_t)rowidx[kk] * 64 + l];
    printf("row_newbcast: %d bad (lane 5 got %d / %d, lane 37 got %d / %d)\n", bad1, oi[5], oi[64 + 5], oi[37], oi[64 + 37]);
    printf("row_shl packing: %d bad (lane 0 %08x, lane 14 %08x, lane 17 %08x)\n", bad2, oi[128], oi[128 + 14], oi[128 + 17]);
    printf("staged rows: %d bad (row 0 lane 0 %g want %g; row 5 lane 9 %g want %g)\n", bad3, of[0], rows[(size_t)rowidx[0] * 64],
           of[5 * 64 + 9], rows[(size_t)rowidx[5] * 64 + 9]);
    return 0;
}
