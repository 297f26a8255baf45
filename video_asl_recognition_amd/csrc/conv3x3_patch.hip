// 3x3 / stride 1 / pad 1 convolution with the input patch resident in LDS (gfx950).
//
// The gather-GEMM fetches a pixel's 64-channel slice once per filter tap -- nine times per
// K sweep -- and at 128-wide tiles the L2->LDS fabric, not the MFMA pipe, sets its speed.
// Here a workgroup owns 256 consecutive output pixels (row-major over image, row, column;
// tiles may straddle rows and images), copies the contiguous range of halo'd input pixels
// those outputs touch into LDS ONCE per 64-channel chunk, and reads all nine taps' pixel
// fragments from that patch at a per-tap constant offset.  Only the 64x64 weight tile of each
// (chunk, tap) step still streams, through a 3-deep LDS-DMA ring with counted waits.
//
// MFMA orientation, epilogue and weight packing are those of gather_gemm.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "asl_hip.h"
#include "common.h"
#include "epilogue.h"
#include "patch_args.h"

namespace asl_detail {

// LDS budget of one workgroup: patch (kPatchPixels x 128 B, one 64-channel chunk) + kStages weight
// stages of 64 rows x 128 B = 55,296 + 24,576 = 78 KiB, so two workgroups share a CU's 160 KiB and
// cover each other's prologue, epilogue and patch reloads.
constexpr int kPatchPixels = 432;
constexpr int kPatchBytes = kPatchPixels * 128;
constexpr int kStages = 3;                    // weight ring depth: prefetch distance 2

// NW waves; workgroup tile = up to 256 pixel slots x BN output channels, waves laid out (NW / (BN/64)) x (BN/64),
// each wave a (256 / rows of waves) x 64 tile.  A 64x64 wave tile reads as many LDS bytes per MFMA as the LDS
// pipe can deliver at the MFMA peak rate; the BN = 128 shape gives each wave 128 pixels x 64 channels (a third
// fewer fragment reads per MFMA) and pays for it with K steps of 32 channels so a weight stage stays 8 KiB.
template <int NW, int BN>
__global__ __launch_bounds__(NW * 64) void conv3x3_patch_kernel(const PatchArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WGN = BN / 64, WGM = NW / WGN;
    constexpr int WTM = 256 / WGM;
    constexpr int FM = WTM / 16, FN = 4;
    constexpr int KS = BN == 64 ? 64 : 32;            // input channels per K step
    constexpr int KK = KS / 32, SPT = 64 / KS;        // MFMA k-slices per step; steps per (chunk, tap)
    constexpr int RB = KS * 2, CPR = RB / 16, RPI = 64 / CPR;   // staged weight row: bytes, 16-B chunks, rows per DMA
    constexpr int W_BYTES = BN * RB;                  // one weight stage (8 KiB)
    constexpr int W_PW = W_BYTES / 1024 / NW;         // LDS-DMA instructions per wave per stage
    static_assert(W_PW >= 1 && W_PW * NW * 1024 == W_BYTES, "weight stage / wave count mismatch");
    auto wswz = [](int row) { return KS == 64 ? ((row >> 1) & 7) : ((row & 8) ? 3 : 0); };

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wst = smem + kPatchBytes;              // weight ring follows the patch

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
#if defined(ASL_STAMPS)
    unsigned long long st[5];
    st[0] = __builtin_amdgcn_s_memtime();
#endif

    int lbid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lbid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tn = lbid % a.tiles_n, tm = lbid / a.tiles_n;
    const int m0 = tm * a.bmv;                         // bmv <= 256 valid pixel slots per tile
    // Two workgroups share a CU.  Dispatched together they would load, compute and store in lockstep
    // and leave HBM idle during the K loops; delaying the second resident set once, by about half a
    // tile, keeps one workgroup's epilogue under the other's MFMA phase.  Speed only: any placement
    // is correct.
    if (a.stagger_ticks > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)a.stagger_ticks) __builtin_amdgcn_s_sleep(8);
    }
    const int HW = a.H * a.W;

    // centre pixel (in halo'd, image-major pixel units) of output pixel m
    auto centre = [&](int m) -> long long {
        const int n = (int)fdiv(m, a.dOHW), r1 = m - n * HW;
        const int y = (int)fdiv(r1, a.dOW), x = r1 - y * a.W;
        return ((long long)n * a.Hp + y + a.in_halo) * a.Wp + x + a.in_halo;
    };
    const int mend = (m0 + a.bmv < a.M ? m0 + a.bmv : a.M);           // one past this tile's last pixel
    const long long P0 = centre(m0) - a.Wp - 1;                       // first pixel of the patch
    const int NP = (int)(centre(mend - 1) + a.Wp + 1 - P0) + 1;       // pixels in the patch (<= kPatchPixels)
    const int NPI = (NP + 7) >> 3;                                    // 8 pixels per DMA instruction

    // ---- DMA descriptors ---------------------------------------------------------------------
    const long long pix_bytes = 2LL * a.C;
    const char* xbase = a.x + P0 * pix_bytes;
    long long xrem = a.x_bytes - P0 * pix_bytes;
    const unsigned xrec = (unsigned)(xrem > 0x7fffffffLL ? 0x7fffffffLL : xrem);
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, xrec, 0x00020000);
    const char* wbase = a.w + 2LL * (tn * BN) * a.wk;
    const unsigned wrec = (unsigned)(a.w_bytes - 2LL * (tn * BN) * a.wk);
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, wrec, 0x00020000);

    // patch DMA: instruction i covers patch pixels 8i..8i+7; slot (p, c') holds chunk c' ^ ((p>>1)&7).
    // lane -> pixel 8i + lane/8, so (p>>1)&7 = (lane>>4) | ((i&1) << 2): the odd-i offset is the even-i
    // one with byte-offset bit 6 flipped
    const unsigned pvoff0 = (unsigned)((lane >> 3) * pix_bytes + 16 * ((lane & 7) ^ (lane >> 4)));
    unsigned wvoff[W_PW];
#pragma unroll
    for (int i = 0; i < W_PW; ++i) {
        const int row = (wave * W_PW + i) * RPI + lane / CPR;
        wvoff[i] = (unsigned)(2 * row * a.wk + 16 * ((lane % CPR) ^ wswz(row)));
    }
    auto dma_patch = [&](int chunk) {
        for (int i = wave; i < NPI; i += NW) {
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(i * 8 * pix_bytes + chunk * 128));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(smem + i * 1024), 16,
                                                     pvoff0 ^ ((unsigned)(i & 1) << 6), so, 0, 0);
        }
    };
    const int nsteps = a.nchunks * 9 * SPT;
    auto dma_weights = [&](int s) {                    // step s = (chunk * 9 + tap) * SPT + half -> ring slot s % kStages
        const int ct = s / SPT, half = s - ct * SPT;
        const int chunk = ct / 9, tap = ct - 9 * chunk;
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(2 * (tap * a.C + chunk * 64 + half * KS)));
        char* dst = wst + (s % kStages) * W_BYTES + wave * (W_PW * 1024);
#pragma unroll
        for (int i = 0; i < W_PW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                                     wvoff[i], so, 0, 0);
    };

    // ---- per-lane fragment geometry ---------------------------------------------------------
    const int fr = lane & 15, fq = lane >> 4;
    int pc[FM];                                   // patch-local centre pixel of this lane's pixel, per fragment
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        int m = m0 + wm * WTM + i * 16 + fr;
        m = m < mend ? m : mend - 1;              // idle slots recompute the tile's last pixel; never stored
        pc[i] = (int)(centre(m) - P0);
    }
    int wrd[FN][KK];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int row = wn * 64 + j * 16 + fr;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wrd[j][kk] = row * RB + (((kk * 4 + fq) ^ wswz(row)) << 4);
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- output coordinates and residual loads first: the oldest loads in flight, they retire under the
    // prologue's first counted wait and cost the epilogue nothing
    OutGeom g;
    g.scale = a.scale; g.bias = a.bias; g.res = a.res; g.y = a.y; g.M = mend; g.N = a.N;
    g.dOW = a.dOW; g.dOHW = a.dOHW; g.dODHW = a.dODHW;
    g.out_sn = a.out_sn; g.out_sd = 0; g.out_sh = a.out_sh; g.out_sw = a.out_sw; g.out_off0 = a.out_off0;
    g.res_sn = a.res_sn; g.res_sd = 0; g.res_sh = a.res_sh; g.res_sw = a.res_sw; g.res_off0 = a.res_off0;
    g.relu = a.relu; g.out_f32 = 0; g.res_f32 = 0;
    // (the 128-pixel wave tile has no registers to spare for that during the K loop and prepares afterwards)
    constexpr bool kEarlyEpi = FM <= 4;
    EpiState<(kEarlyEpi ? FM : 4), FN> est;
    if constexpr (kEarlyEpi) epilogue_prepare<FM, FN>(est, g, m0 + wm * WTM, tn * BN + wn * 64, lane);

    // ---- prologue: patch of chunk 0, weights of steps 0 and 1 ----------------------------------
    // Waits are counted: vmcnt(W_PW) retires everything but the youngest weight stage, so one stage stays
    // in flight across every barrier.  The barrier is raw (inside the asm) so hipcc adds no vmcnt(0).
    dma_patch(0);
    dma_weights(0);
    if (nsteps > 1) {
        dma_weights(1);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(W_PW) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }

#if defined(ASL_STAMPS)
    st[1] = __builtin_amdgcn_s_memtime();
#endif
    int tap = 0, chunk = 0, half = 0;
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
#if !defined(ASL_ABL_NODMA)
        if (s + 2 < nsteps) dma_weights(s + 2);       // its ring slot was last read in step s-1
#endif
        const int r = tap / 3, q = tap - 3 * r;
        const int toff = (r - 1) * a.Wp + (q - 1);
        const char* wb = wst + (s % kStages) * W_BYTES;
        bf16x8 xf[KK][FM], wf[KK][FN];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int c16 = (half * KK + kk) * 4 + fq;          // 16-byte chunk of the pixel's 128-byte row
#pragma unroll
            for (int j = 0; j < FN; ++j) wf[kk][j] = *(const bf16x8*)(wb + wrd[j][kk]);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int p = pc[i] + toff;
                xf[kk][i] = *(const bf16x8*)(smem + p * 128 + ((c16 ^ ((p >> 1) & 7)) << 4));
            }
        }
        // 4-wave shapes: all fragment reads issue before the first MFMA, so one LDS latency is exposed per
        // step instead of one per fragment group (the 8-wave shape has no registers for that and would
        // lose a resident wave per SIMD)
        if constexpr (NW == 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][j], xf[kk][i], acc[i][j], 0, 0, 0);
        if (++half == SPT) { half = 0; ++tap; }
        if (tap == 9) {
            tap = 0;
            ++chunk;
            if (chunk < a.nchunks) {
                // next 64-channel chunk: every wave is done with the patch after this barrier; reload it
                // (the other resident workgroup covers the bubble), weights in flight keep flying
                asm volatile("s_barrier" ::: "memory");
                dma_patch(chunk);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                continue;
            }
        }
#if defined(ASL_ABL_NOBAR)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        if (s + 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(W_PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    }
#if defined(ASL_ABL_NOEPI)
    if (a.M > 0) { if (acc[0][0][0] == 123.456f) a.y[0] = 1; return; }
#endif
#if defined(ASL_STAMPS)
    st[2] = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue: the staging LDS overlays the patch; the loop's last barrier frees it ------------
    static_assert(NW * kEpiWaveBytes(FN * 16) <= kPatchBytes, "epilogue stage does not fit the patch buffer");
    char* const stage = smem + wave * kEpiWaveBytes(FN * 16);
    if constexpr (kEarlyEpi) {
        epilogue_store<FM, FN>(acc, stage, g, est, lane);
    } else {
#pragma unroll
        for (int hf = 0; hf < FM / 4; ++hf) {
            epilogue_prepare<4, FN>(est, g, m0 + wm * WTM + hf * 64, tn * BN + wn * 64, lane);
            epilogue_store<4, FN>(*(const f32x4(*)[4][FN])&acc[hf * 4], stage, g, est, lane);
        }
    }
#if defined(ASL_STAMPS)
    st[3] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[4] = __builtin_amdgcn_s_memtime();
    if (a.stamps && tid == 0)
        for (int k = 0; k < 5; ++k) a.stamps[(long long)blockIdx.x * 5 + k] = st[k];
#endif
#endif
}

int ensure_dynamic_lds(const void* fn, int bytes, unsigned char* done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev >= 0 && __atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0) __atomic_store_n(&done[dev], (unsigned char)1, __ATOMIC_RELEASE);
    return 0;
}

int device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}

// Largest patch (pixels, halo included) any tile of bmv consecutive output pixels needs
long long max_patch_pixels(int n, int h, int w, int in_halo, int bmv) {
    const int hp = h + 2 * in_halo, wp = w + 2 * in_halo;
    const long long M = (long long)n * h * w, hw = (long long)h * w;
    auto centre = [&](long long m) {
        const long long img = m / hw, r1 = m - img * hw;
        const long long y = r1 / w, x = r1 - y * w;
        return (img * hp + y + in_halo) * wp + x + in_halo;
    };
    const long long tiles = (M + bmv - 1) / bmv;
    const long long scan = tiles < hw ? tiles : hw;      // the tile/image phase repeats within h*w tiles
    long long mx = 0;
    for (long long t = 0; t < scan; ++t) {
        const long long m0 = t * bmv, ml = (m0 + bmv - 1 < M ? m0 + bmv - 1 : M - 1);
        const long long np = centre(ml) - centre(m0) + 2 * wp + 3;
        mx = np > mx ? np : mx;
    }
    return mx;
}

// Pixel slots per tile for this geometry (0 = the patch kernel cannot run it): 256 consecutive pixels
// when their patch always fits; else whole images; else whole rows that tile an image exactly.
int patch_conv_tile(int n, int h, int w, int in_halo) {
    if (max_patch_pixels(n, h, w, in_halo, 256) <= kPatchPixels) return 256;
    const int hw = h * w;
    if (hw <= 256) {
        const int b = 256 / hw * hw;
        if (max_patch_pixels(n, h, w, in_halo, b) <= kPatchPixels) return b;
    }
    for (int k = 256 / w; k >= 1; --k)
        if (h % k == 0 && k * w * 4 >= 256 * 3 && max_patch_pixels(n, h, w, in_halo, k * w) <= kPatchPixels) return k * w;
    return 0;
}

int launch_conv3x3_patch(PatchArgs a, hipStream_t stream) {
    // the kernel trusts these; refuse anything inconsistent rather than index with it
    if (a.W <= 0 || a.H <= 0 || (int)a.dOW.d != a.W || (int)a.dOHW.d != a.H * a.W || a.bmv <= 0 || a.bmv > 256 ||
        a.nchunks * 64 != a.C || a.Wp != a.W + 2 * a.in_halo || a.Hp != a.H + 2 * a.in_halo)
        return ASL_EINVAL;
    a.tiles_m = (a.M + a.bmv - 1) / a.bmv;
    // 128-channel tiles: a third fewer LDS fragment reads per MFMA and half the patch traffic
    // (measured no faster than the 64-channel tile on any ResNet layer: both shapes run ~1.5x the MFMA-bound
    // step time; the wide shape is kept selectable for experiments)
    const bool wide = false;   // the 128-channel tile is compiled for experiments only
    a.tiles_n = wide ? (a.N + 127) / 128 : (a.N + 63) / 64;
    // 100 MHz ticks: ~0.2 us per K step is half of a tile's measured duration; only worth it when the
    // grid runs for many rounds of 512 resident workgroups
    a.stagger_ticks = 0;
    if ((long long)a.tiles_m * a.tiles_n >= 6 * 512) a.stagger_ticks = 20 * a.nchunks * 9 + 200;
    const int lds = kPatchBytes + kStages * 64 * 128;
    // single-chunk layers (C = 64) are prologue/epilogue-bound: 8 waves split that work finer
    const bool w8 = a.nchunks == 1;
    static unsigned char lds_set[3][64];
    const void* fn = wide ? (const void*)conv3x3_patch_kernel<4, 128>
                          : w8 ? (const void*)conv3x3_patch_kernel<8, 64> : (const void*)conv3x3_patch_kernel<4, 64>;
    if (const int e = ensure_dynamic_lds(fn, lds, lds_set[wide ? 0 : w8 ? 1 : 2])) return e;
    const dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
    if (wide) hipLaunchKernelGGL((conv3x3_patch_kernel<4, 128>), grid, dim3(256), lds, stream, a);
    else if (w8) hipLaunchKernelGGL((conv3x3_patch_kernel<8, 64>), grid, dim3(512), lds, stream, a);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<4, 64>), grid, dim3(256), lds, stream, a);
    return (int)hipGetLastError();
}

}  // namespace asl_detail
