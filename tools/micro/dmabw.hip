// Global -> LDS fill rate per CU from an L2-resident buffer: (a) LDS-DMA pieces (buffer_load ... lds, 1 KiB per
// wave instruction), (b) the same bytes through registers (buffer_load_dwordx4 + ds_write_b128).  Every wave of
// a 512-thread workgroup fills its own 4 KiB LDS ring; 1 or 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kSrcBytes = 288 * 1024;

template <int MODE>
__global__ __launch_bounds__(512, 4) void fill(const char* src, int iters, int* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, kSrcBytes, 0x00020000);
    // 16 rows of 64 bytes, rows 2304 bytes apart (a weight stage piece)
    const unsigned voff = (unsigned)((lane >> 2) * 2304 + (lane & 3) * 16);
    char* ring = smem + wave * 4096;
    int acc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(((it * 8 + wave) * 64) % 2304 + ((it & 7) * 16 * 2304)));
        char* dst = ring + (it & 3) * 1024;
        if (MODE == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, so, 0, 0);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            v4i v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so, 0);
            *(v4i*)(dst + lane * 16) = v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    acc = ((int*)smem)[threadIdx.x];
    if (acc == 0x7fffffff) out[0] = 1;
}

template <int MODE>
static void run(const char* src, int* out, int per_cu) {
    const int iters = 4000, blocks = 256 * per_cu;
    hipFuncSetAttribute((const void*)fill<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    fill<MODE><<<blocks, 512, 65536>>>(src, 100, out);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); fill<MODE><<<blocks, 512, 65536>>>(src, iters, out); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes_cu = (double)iters * 8 * 1024 * per_cu;
    printf("%s, %d workgroup(s)/CU: %.1f GB/s per CU, %.0f ns per 1 KiB piece per CU (%.3f ms)\n",
           MODE == 0 ? "LDS-DMA          " : "registers + write", per_cu, bytes_cu / (ms * 1e-3) / 1e9, ms * 1e6 / (iters * 8.0 * per_cu), ms);
}
int main() {
    char* src; int* out; hipMalloc(&src, kSrcBytes); hipMemset(src, 1, kSrcBytes); hipMalloc(&out, 4);
    for (int per_cu = 1; per_cu <= 2; ++per_cu) { run<0>(src, out, per_cu); run<1>(src, out, per_cu); }
    return 0;
}
