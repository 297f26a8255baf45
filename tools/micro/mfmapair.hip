// Do two waves on one SIMD overlap one's MFMAs with the other's vector instructions?  8 waves per workgroup (two per
// SIMD), one workgroup per CU.  Waves 0-3 run v_mfma_f32_32x32x16_bf16 (two chains), waves 4-7 run independent
// v_fma_f32 chains; timed alone and together.  Then both halves run the same mix (1 MFMA + NF v_fma per step), as
// a kernel whose waves all multiply and do vector work would, against one wave per SIMD running that mix.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfmapair.hip -o tools/micro/mfmapair
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// mode bit 0: waves 0-3 run, bit 1: waves 4-7 run.  kMfmaA / kFmaA: per step of waves 0-3; kMfmaB / kFmaB: waves 4-7
template <int kMfmaA, int kFmaA, int kMfmaB, int kFmaB>
__global__ __launch_bounds__(512, 1) void k(const bf16x8* in, float* out, unsigned long long* cyc, int iters, int mode) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = wave >> 2;
    bf16x8 a = in[lane], b = in[64 + lane];
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
    float f[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if ((mode >> half) & 1) {
        if (half == 0) {
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if constexpr (kMfmaA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"((u & 1) ? acc1 : acc0) : "v"(a), "v"(b));
#pragma unroll
                    for (int q = 0; q < kFmaA; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[q]));
                }
            }
        } else {
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if constexpr (kMfmaB) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"((u & 1) ? acc1 : acc0) : "v"(a), "v"(b));
#pragma unroll
                    for (int q = 0; q < kFmaB; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[q]));
                }
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc0), "+v"(acc1));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
    for (int q = 0; q < 8; ++q) s += f[q];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// Every wave alternates a K phase (36 MFMAs back to back) and a vector phase (kV v_fma), as the tiles of a convolution
// kernel do; kPrio: 1 = raise the wave's priority for the K phase, 2 = for the vector phase; kSkew: waves 4-7 start
// with a vector phase (anti-phase start).
template <int kV, int kPrio, int kSkew>
__global__ __launch_bounds__(512, 1) void phases(const bf16x8* in, float* out, unsigned long long* cyc, int iters, int mode) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = wave >> 2;
    bf16x8 a = in[lane], b = in[64 + lane];
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
    float f[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if ((mode >> half) & 1) {
        if (kSkew && half) {
            for (int v = 0; v < kV / 8; ++v)
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[q]));
        }
        for (int i = 0; i < iters; ++i) {
            if (kPrio == 1) __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int u = 0; u < 36; ++u) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"((u & 1) ? acc1 : acc0) : "v"(a), "v"(b));
            if (kPrio == 1) __builtin_amdgcn_s_setprio(0);
            if (kPrio == 2) __builtin_amdgcn_s_setprio(3);
            for (int v = 0; v < kV / 8; ++v)
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[q]));
            if (kPrio == 2) __builtin_amdgcn_s_setprio(0);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc0), "+v"(acc1));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
    for (int q = 0; q < 8; ++q) s += f[q];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int kV, int kPrio, int kSkew>
void run_phases(const char* what, const bf16x8* in, float* out, unsigned long long* cyc, int cus, int mode) {
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((phases<kV, kPrio, kSkew>), dim3(cus), dim3(512), 0, 0, in, out, cyc, iters, mode);
    (void)hipDeviceSynchronize();
    static unsigned long long h[8192];
    (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * cus * 8, hipMemcpyDeviceToHost);
    unsigned long long ma = 0, mb = 0;
    for (int i = 0; i < cus; ++i)
        for (int w = 0; w < 8; ++w) {
            unsigned long long& m = w < 4 ? ma : mb;
            m = h[i * 8 + w] > m ? h[i * 8 + w] : m;
        }
    printf("%-58s waves 0-3: %6.0f  waves 4-7: %6.0f cycles per tile (36 MFMAs = 1152)\n", what, ma / (double)iters, mb / (double)iters);
}

template <int kMfmaA, int kFmaA, int kMfmaB, int kFmaB>
void run(const char* what, const bf16x8* in, float* out, unsigned long long* cyc, int cus, int mode) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<kMfmaA, kFmaA, kMfmaB, kFmaB>), dim3(cus), dim3(512), 0, 0, in, out, cyc, iters, mode);
    (void)hipDeviceSynchronize();
    static unsigned long long h[8192];
    (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * cus * 8, hipMemcpyDeviceToHost);
    unsigned long long ma = 0, mb = 0;
    for (int i = 0; i < cus; ++i)
        for (int w = 0; w < 8; ++w) {
            unsigned long long& m = w < 4 ? ma : mb;
            m = h[i * 8 + w] > m ? h[i * 8 + w] : m;
        }
    printf("%-58s waves 0-3: %6.1f  waves 4-7: %6.1f cycles per step\n", what, ma / (iters * 8.0), mb / (iters * 8.0));
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    bf16x8* in; float* out; unsigned long long* cyc;
    (void)hipMalloc(&in, 128 * 16); (void)hipMalloc(&out, cus * 512 * 4); (void)hipMalloc(&cyc, cus * 64);
    (void)hipMemset(in, 0, 128 * 16);
    run<1, 0, 0, 6>("MFMA alone (waves 0-3)", in, out, cyc, cus, 1);
    run<1, 0, 0, 6>("6 v_fma alone (waves 4-7)", in, out, cyc, cus, 2);
    run<1, 0, 0, 6>("MFMA on 0-3 + 6 v_fma on 4-7", in, out, cyc, cus, 3);
    run<1, 0, 0, 3>("MFMA on 0-3 + 3 v_fma on 4-7", in, out, cyc, cus, 3);
    run<1, 0, 0, 8>("MFMA on 0-3 + 8 v_fma on 4-7", in, out, cyc, cus, 3);
    run<1, 3, 1, 3>("one wave per SIMD: MFMA + 3 v_fma", in, out, cyc, cus, 1);
    run<1, 3, 1, 3>("two waves per SIMD, each MFMA + 3 v_fma", in, out, cyc, cus, 3);
    run<1, 6, 1, 6>("one wave per SIMD: MFMA + 6 v_fma", in, out, cyc, cus, 1);
    run<1, 6, 1, 6>("two waves per SIMD, each MFMA + 6 v_fma", in, out, cyc, cus, 3);
    run<1, 0, 1, 0>("two waves per SIMD, each MFMA only", in, out, cyc, cus, 3);
    run_phases<256, 0, 0>("tiles of 36 MFMA + 256 v_fma, one wave per SIMD", in, out, cyc, cus, 1);
    run_phases<256, 0, 0>("same, two waves per SIMD", in, out, cyc, cus, 3);
    run_phases<256, 0, 1>("same, two waves, second starts with its vector phase", in, out, cyc, cus, 3);
    run_phases<256, 1, 0>("same, two waves, priority raised in the K phase", in, out, cyc, cus, 3);
    run_phases<256, 2, 0>("same, two waves, priority raised in the vector phase", in, out, cyc, cus, 3);
    run_phases<128, 0, 0>("tiles of 36 MFMA + 128 v_fma, one wave per SIMD", in, out, cyc, cus, 1);
    run_phases<128, 0, 0>("same, two waves per SIMD", in, out, cyc, cus, 3);
    run_phases<128, 2, 0>("same, two waves, priority raised in the vector phase", in, out, cyc, cus, 3);
    return 0;
}
