// MFMA issue rate of one wave per SIMD (256-thread workgroups, one per CU, every CU busy): cycles per
// v_mfma_f32_32x32x16_bf16 and per v_mfma_f32_16x16x32_bf16 on 1, 2 and 4 accumulation chains, zero and
// random operands, from s_memtime inside the kernel and from the wall clock.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfmarate.hip -o tools/micro/mfmarate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE, int CHAINS>
__global__ __launch_bounds__(256, 1) void k(const bf16x8* in, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a = in[lane], b = in[64 + lane];
    unsigned long long t0 = 0, t1 = 0;
    if constexpr (SHAPE == 32) {
        f32x16 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c)
            for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int c = 0; c < CHAINS; ++c)
            for (int e = 0; e < 16; ++e) s += acc[c][e];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x4 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0, 0, 0, 0};
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int c = 0; c < CHAINS; ++c)
            for (int e = 0; e < 4; ++e) s += acc[c][e];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int CHAINS>
void run(const char* what, const bf16x8* in, float* out, unsigned long long* cyc, int cus) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, CHAINS>), dim3(cus), dim3(256), 0, 0, in, out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int i = 0; i < cus; ++i) mx = h[i] > mx ? h[i] : mx;
    const double n = (double)iters * 8 * CHAINS;
    const double flop = n * (SHAPE == 32 ? 32768.0 : 16384.0) * 4 * cus;
    printf("%-8s %dx chains=%d: %6.1f cycles/MFMA (s_memtime), wall %.3f ms, %7.1f TFLOP/s, clock %.2f GHz\n", what, SHAPE, CHAINS,
           mx / n, ms, flop / ms / 1e9, mx / (ms * 1e6));
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    bf16x8* in; float* out; unsigned long long* cyc;
    hipMalloc(&in, 128 * 16); hipMalloc(&out, cus * 256 * 4); hipMalloc(&cyc, cus * 8);
    for (int pass = 0; pass < 2; ++pass) {
        unsigned short h[1024];
        for (int i = 0; i < 1024; ++i) h[i] = pass ? (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15)) : 0;
        hipMemcpy(in, h, 2048, hipMemcpyHostToDevice);
        const char* what = pass ? "random" : "zeros";
        run<32, 1>(what, in, out, cyc, cus);
        run<32, 2>(what, in, out, cyc, cus);
        run<32, 4>(what, in, out, cyc, cus);
        run<16, 1>(what, in, out, cyc, cus);
        run<16, 4>(what, in, out, cyc, cus);
    }
    return 0;
}
