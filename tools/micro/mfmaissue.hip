// What slows a lone wave's MFMA stream (one wave per SIMD, every CU busy)?  v_mfma_f32_32x32x16_bf16 through
// inline asm, two alternating accumulation chains, with: the accumulators in vector or accumulation registers,
// an s_waitcnt + s_nop in every gap, the A operand rotating over 8 register groups, one ds_read_b128 per gap,
// and 0 / 2 / 4 / 6 independent v_fma_f32 per gap.  Prints cycles per MFMA from s_memtime.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfmaissue.hip -o tools/micro/mfmaissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool ACCV, bool WAITS, bool ROT, bool LDS, int NFMA>
__global__ __launch_bounds__(256, 1) void k(const bf16x8* in, float* out, unsigned long long* cyc, int iters) {
    __shared__ bf16x8 sm[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = in[i & 127];
    __syncthreads();
    bf16x8 a[8], b = in[64 + lane];
    for (int i = 0; i < 8; ++i) a[i] = in[(lane + i) & 127];
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
    float f[6] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f};
    const unsigned ad = (unsigned)(unsigned long long)(__attribute__((address_space(3))) bf16x8*)sm + (threadIdx.x & 255) * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            f32x16& acc = (u & 1) ? acc1 : acc0;
            const bf16x8& aa = ROT ? a[u] : a[0];
            if constexpr (WAITS) asm volatile("s_waitcnt lgkmcnt(5)");
            if constexpr (ACCV) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(aa), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(aa), "v"(b));
            if constexpr (LDS) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(b) : "v"(ad));
            if constexpr (WAITS) asm volatile("s_nop 0");
#pragma unroll
            for (int q = 0; q < NFMA; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[q]));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" : "+v"(b));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
    for (int q = 0; q < 6; ++q) s += f[q];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)b[0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool ACCV, bool WAITS, bool ROT, bool LDS, int NFMA>
void run(const bf16x8* in, float* out, unsigned long long* cyc, int cus) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<ACCV, WAITS, ROT, LDS, NFMA>), dim3(cus), dim3(256), 0, 0, in, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[1024], mx = 0;
    (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
    for (int i = 0; i < cus; ++i) mx = h[i] > mx ? h[i] : mx;
    printf("acc in %s, waits %d, rotating A %d, ds_read %d, v_fma x%d: %6.1f cycles/MFMA\n", ACCV ? "VGPR" : "AGPR", WAITS, ROT, LDS, NFMA,
           mx / ((double)iters * 8));
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    bf16x8* in; float* out; unsigned long long* cyc;
    (void)hipMalloc(&in, 128 * 16); (void)hipMalloc(&out, cus * 256 * 4); (void)hipMalloc(&cyc, cus * 8);
    (void)hipMemset(in, 0, 128 * 16);
    run<false, false, false, false, 0>(in, out, cyc, cus);
    run<true, false, false, false, 0>(in, out, cyc, cus);
    run<true, true, false, false, 0>(in, out, cyc, cus);
    run<true, true, true, false, 0>(in, out, cyc, cus);
    run<true, true, true, true, 0>(in, out, cyc, cus);
    run<false, true, true, true, 0>(in, out, cyc, cus);
    run<true, true, true, true, 2>(in, out, cyc, cus);
    run<true, true, true, true, 4>(in, out, cyc, cus);
    run<true, true, true, true, 6>(in, out, cyc, cus);
    run<false, true, true, true, 4>(in, out, cyc, cus);
    return 0;
}
