// Microbenchmark behind one design question of the 3x3 large-tile kernels (conv3x3_pp.hip): does the K loop run
// faster when the WEIGHT fragments come straight from L2 into registers (buffer_load_dwordx4 per fragment) instead of
// through LDS by LDS-DMA pieces?  The loop below has the real kernel's skeleton -- 8-wave workgroups, two per CU, the
// two halves of a workgroup one phase apart, per step {DMA issue, fragment reads, counted wait, barrier, 16 MFMAs
// 16x16x32, barrier}, a patch piece per wave on 4 of every 9 steps -- with the addressing reduced to constants; the
// sums it produces mean nothing.
//   MODE 0: weights by LDS-DMA (1 piece per wave and step), 4 weight + 4 patch ds_read_b128 per wave and step
//   MODE 1: weights by 4 buffer_load_dwordx4 per wave and step (64 x 64 wave tile, four waves share each row),
//           issued one step ahead into the registers the MFMAs have just read; 4 patch ds_read_b128
//   MODE 2: 128-pixel x 32-channel wave tiles: 2 weight loads (two waves share each row) + 8 patch ds_read_b128
//   MODE 3, 4: as 1, 2 with the weights pre-packed in fragment order (every load reads 1 KiB of consecutive bytes
//           instead of 16 rows x 64 B)
//   MODE 5: as 0 without any DMA (the loop's floor); MODE 6: as 0 without the weight DMA and the weight reads
//   MODE 7: as 5 without any barrier (free-running waves); MODE 8: as 5 with ONE barrier per step and no phase offset
//   MODE 9: as 0 (all DMA) with ONE barrier per step and no phase offset
// Prints cycles per step (s_memtime, median over workgroups) and the wall time of the launch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int kRowBytes = 4608;            // 9 taps x 256 channels x 2 B
constexpr int kRows = 128;                 // one N tile of weights
constexpr int kWBytes = kRows * kRowBytes;
constexpr int kXBytes = 8 << 20;           // patch source: streams from HBM/L2 like an activation tensor
constexpr int kLds = 80 * 1024;

template <int MODE>
__global__ __launch_bounds__(512, 4) void wloop(const char* w, const char* x, int steps, unsigned long long* stamps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = wave >= 4;
    const int fr = lane & 15, fq = lane >> 4;
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (unsigned)kWBytes, 0x00020000);
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)kXBytes, 0x00020000);
    char* ring = smem;                      // 3 x 8 KiB weight stages
    char* patch = smem + 24 * 1024;         // 2 x 28 KiB
    constexpr bool kDirect = MODE >= 1 && MODE <= 4, kTall = MODE == 2 || MODE == 4, kPacked = MODE == 3 || MODE == 4;
    constexpr bool kWDma = MODE == 0 || MODE == 9, kPDma = MODE != 5 && MODE != 7 && MODE != 8, kWRead = MODE == 0 || MODE == 5 || MODE >= 7;
    constexpr bool kPingPong = MODE <= 6;
    constexpr int FM = kTall ? 8 : 4, FN = kTall ? 2 : 4;
    const int wc = kTall ? (wave & 3) : (wave & 1);
    // per-lane offsets: DMA piece (16 rows x 64 B), weight fragment rows, LDS fragment addresses (conflict-free)
    const unsigned wpiece_vo = (unsigned)((wave * 16 + (lane >> 2)) * kRowBytes + (lane & 3) * 16);
    unsigned wfrag_vo[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j)
        wfrag_vo[j] = kPacked ? (unsigned)((wc * FN + j) * 1024 + lane * 16) : (unsigned)((wc * (FN * 16) + j * 16 + fr) * kRowBytes + fq * 16);
    const unsigned xvo = (unsigned)(lane * 16);
    unsigned xoff = (unsigned)((blockIdx.x * 7919u * 1024u) & (kXBytes - 1));
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) wf[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (kDirect) {
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wfrag_vo[j], 0, 0));
    }
    __syncthreads();
    if (late && kPingPong) asm volatile("s_barrier" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        const int tap = s % 9;
        const bool ppiece = tap < 4;
        if (ppiece && kPDma) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(patch + ((s / 9) & 1) * 28672 + (wave + 8 * tap) % 28 * 1024),
                                                     16, xvo, __builtin_amdgcn_readfirstlane(xoff), 0, 0);
            xoff = (xoff + 8192u) & (kXBytes - 1);
        }
        // packed: one 8 KiB slab of fragments per step; else the step's 64-byte column of every row
        const unsigned kso = __builtin_amdgcn_readfirstlane((unsigned)(((s + 1) % 72) * (kPacked ? 8192 : 64)));
        if (kWDma)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(ring + ((s + 2) % 3) * 8192 + wave * 1024), 16,
                                                     wpiece_vo, kso, 0, 0);
        bf16x8 xf[FM], wl[FN];
        const char* pb = patch + ((s / 9) & 1) * 28672 + tap * 64;
#pragma unroll
        for (int i = 0; i < FM; ++i) xf[i] = *(const bf16x8*)(pb + ((wave >> (kTall ? 2 : 1)) * FM + i) * 1024 % 24576 + fq * 256 + fr * 16);
        if (kWRead) {
            const char* wb = ring + (s % 3) * 8192;
#pragma unroll
            for (int j = 0; j < FN; ++j) wl[j] = *(const bf16x8*)(wb + (wc * 64 + j * 16 + fr) * 64 + ((fq ^ ((fr & 8) ? 3 : 0)) << 4));
        }
        if (MODE == 7) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 5 || MODE == 8) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if (MODE == 9) {
            if (ppiece) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if (MODE == 6) {
            if (ppiece) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if (MODE == 0) {
            if (ppiece) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            // the weight loads of this step were issued before this step's patch piece: leave only that piece in flight
            if (ppiece) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kWRead ? wl[j] : wf[j], xf[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (kDirect) {
#pragma unroll
            for (int j = 0; j < FN; ++j) wf[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wfrag_vo[j], kso, 0));
        }
        if (kPingPong) asm volatile("s_barrier" ::: "memory");
    }
    if (!late && kPingPong) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) sum += acc[i][j][0] + acc[i][j][3];
    if (sum == 12345.678f) sink[0] = sum;
    if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* w, const char* x, unsigned long long* stamps, float* sink, int blocks, int steps, const char* what) {
    hipFuncSetAttribute((const void*)wloop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    wloop<MODE><<<blocks, 512, kLds>>>(w, x, 72, stamps, sink);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    std::vector<unsigned long long> h(blocks);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); wloop<MODE><<<blocks, 512, kLds>>>(w, x, steps, stamps, sink); hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); exit(1); }
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    hipMemcpy(h.data(), stamps, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-58s %7.1f cycles/step (median workgroup), %.3f ms\n", what, (double)h[blocks / 2] / steps, best);
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512, steps = argc > 2 ? atoi(argv[2]) : 7200;
    char *w, *x; unsigned long long* stamps; float* sink;
    hipMalloc(&w, kWBytes); hipMalloc(&x, kXBytes); hipMalloc(&stamps, blocks * 8); hipMalloc(&sink, 64);
    // random bit patterns that are small finite bf16 values (random data holds the clock down as real activations do)
    std::vector<unsigned short> hw(kWBytes / 2), hx(kXBytes / 2);
    srand(1);
    for (auto& v : hw) v = (unsigned short)(0x3c00 | (rand() & 0x83ff));
    for (auto& v : hx) v = (unsigned short)(0x3c00 | (rand() & 0x83ff));
    hipMemcpy(w, hw.data(), kWBytes, hipMemcpyHostToDevice); hipMemcpy(x, hx.data(), kXBytes, hipMemcpyHostToDevice);
    printf("%d workgroups of 8 waves, %d steps of 16 MFMA 16x16x32 per wave (MFMA-bound: 1024 cycles per step with two workgroups per CU = four waves per SIMD)\n", blocks, steps);
    run<0>(w, x, stamps, sink, blocks, steps, "weights by LDS-DMA, 64x64 wave tiles (the kernel today)");
    run<1>(w, x, stamps, sink, blocks, steps, "weights L2 -> registers, 64x64 wave tiles, 4 loads/wave");
    run<2>(w, x, stamps, sink, blocks, steps, "weights L2 -> registers, 128x32 wave tiles, 2 loads/wave");
    run<3>(w, x, stamps, sink, blocks, steps, "  ... 64x64, weights packed in fragment order");
    run<4>(w, x, stamps, sink, blocks, steps, "  ... 128x32, weights packed in fragment order");
    run<5>(w, x, stamps, sink, blocks, steps, "LDS-DMA form without any DMA (floor)");
    run<6>(w, x, stamps, sink, blocks, steps, "LDS-DMA form, patch DMA only, no weight reads");
    run<7>(w, x, stamps, sink, blocks, steps, "no DMA, no barriers (free-running waves)");
    run<8>(w, x, stamps, sink, blocks, steps, "no DMA, one barrier per step, no phase offset");
    run<9>(w, x, stamps, sink, blocks, steps, "all DMA, one barrier per step, no phase offset");
    return 0;
}
