// LDS read bandwidth per CU: every wave of a 512-thread workgroup streams conflict-free ds_read_b128 from a
// 64 KiB region; prints bytes per shader clock per CU for 1 and 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void lds_read(int iters, int* out, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) ((int*)smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* p = smem + wave * 8192 + lane * 16;
    v4i a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        v4i b0 = *(const v4i*)(p), b1 = *(const v4i*)(p + 1024), b2 = *(const v4i*)(p + 2048), b3 = *(const v4i*)(p + 3072);
        v4i b4 = *(const v4i*)(p + 4096), b5 = *(const v4i*)(p + 5120), b6 = *(const v4i*)(p + 6144), b7 = *(const v4i*)(p + 7168);
        a0 += b0; a1 += b1; a2 += b2; a3 += b3; a4 += b4; a5 += b5; a6 += b6; a7 += b7;
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    v4i s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + s.y + s.z + s.w == 0x7fffffff) out[0] = 1;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
    int* out; unsigned long long* clk; hipMalloc(&out, 4); hipMalloc(&clk, 8 * 1024);
    const int iters = 20000;
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        const int blocks = 256 * per_cu;
        hipFuncSetAttribute((const void*)lds_read, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        lds_read<<<blocks, 512, 65536>>>(100, out, clk);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); lds_read<<<blocks, 512, 65536>>>(iters, out, clk); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8]; hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
        const double bytes_per_block = (double)iters * 8 * 1024 * 8;   // 8 waves x 8 reads x 1 KiB
        printf("%d workgroup(s)/CU: %.1f B/clk/CU by s_memtime (block 0: %llu clk), %.2f TB/s aggregate by events (%.3f ms)\n",
               per_cu, bytes_per_block * per_cu / (double)h[0], h[0], bytes_per_block * blocks / (ms * 1e-3) / 1e12, ms);
    }
    return 0;
}
