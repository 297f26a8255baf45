// 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels: persistent kernel with the whole filter
// bank resident in LDS (gfx950).
//
// A 64 -> 64 layer has only nine K steps per tile, so a per-tile kernel spends most of its time in the
// prologue (patch and first weight stages arriving) and the epilogue.  Here one 8-wave workgroup per CU
// walks a contiguous range of tiles: the 72 KiB of weights are copied to LDS once, the next tile's
// halo'd input patch streams into the second patch buffer while the current tile is multiplied, its
// residual values are already in registers when the K loop ends, and the K loop itself has no
// barrier and no wait on global memory.  Per tile: one barrier.
//
// LDS: weights 9 x 64 x 128 B = 73,728 + two patches of 352 pixels x 128 B = 90,112 -> 163,840 B (all of it).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "asl_hip.h"
#include "common.h"
#include "patch_args.h"

namespace asl_detail {

constexpr int kC64PatchPixels = 352;             // whole 8-pixel DMA pieces: a piece must not spill into the other buffer
static_assert(kC64PatchPixels % 8 == 0, "patch buffers hold whole DMA pieces");
constexpr int kC64PatchBytes = kC64PatchPixels * 128;
constexpr int kC64WeightBytes = 9 * 64 * 128;
constexpr int kC64Lds = kC64WeightBytes + 2 * kC64PatchBytes;

// Weight rows are read in a permuted order so that a lane's accumulators for channel fragments 2j' and
// 2j'+1 are eight CONSECUTIVE output channels of its pixel: MFMA row r of fragment j is channel
// (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3).  The epilogue then stores 16 bytes per lane straight from
// registers -- no LDS staging, no second barrier.
__device__ __forceinline__ int c64_row(int j, int r) { return (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3); }
// chunk swizzle of the resident weight rows, conflict-free for that read order
__device__ __forceinline__ int c64_wswz(int row) { return (row & 3) | ((row >> 1) & 4); }

__global__ __launch_bounds__(512) void conv3x3_c64_kernel(const PatchArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int FM = 2, FN = 4;                      // wave tile: 32 pixel slots x 64 channels; 8 waves = 256 slots
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wts = smem;
    char* const patch0 = smem + kC64WeightBytes;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    // contiguous tile range of this workgroup: neighbouring tiles share halo rows through L2
    const int G = gridDim.x, b = blockIdx.x;
    const int t0 = (int)((long long)b * a.tiles_m / G), t1 = (int)((long long)(b + 1) * a.tiles_m / G);
    const int HW = a.H * a.W;
    auto centre = [&](int m) -> long long {
        const int n = (int)fdiv(m, a.dOHW), r1 = m - n * HW;
        const int y = (int)fdiv(r1, a.dOW), x = r1 - y * a.W;
        return ((long long)n * a.Hp + y + a.in_halo) * a.Wp + x + a.in_halo;
    };

    // ---- weights: 72 pieces of 8 rows x 128 B, nine per wave; slot (row, c') holds chunk c' ^ c64_wswz(row)
    {
        __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)a.w_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int piece = wave * 9 + k, tap = piece >> 3, r8 = piece & 7;
            const int row = r8 * 8 + (lane >> 3);
            const unsigned vo = (unsigned)(2 * row * a.wk + 16 * ((lane & 7) ^ c64_wswz(row)));
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(2 * tap * 64));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(wts + piece * 1024), 16, vo, so, 0, 0);
        }
    }
    int wrd[FN][2];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int row = c64_row(j, fr);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) wrd[j][kk] = row * 128 + (((kk * 4 + fq) ^ c64_wswz(row)) << 4);
    }

    // ---- patch DMA of tile t into buffer `buf`: piece i covers patch pixels 8i..8i+7 (128 B each) --------
    const unsigned pvoff0 = (unsigned)((lane >> 3) * 128 + 16 * ((lane & 7) ^ (lane >> 4)));
    auto dma_patch = [&](int t, char* buf) {
        const int m0 = t * a.bmv, mend = (m0 + a.bmv < a.M ? m0 + a.bmv : a.M);
        const long long P0 = centre(m0) - a.Wp - 1;
        const int NP = (int)(centre(mend - 1) + a.Wp + 1 - P0) + 1;
        const int NPI = (NP + 7) >> 3;
        long long xrem = a.x_bytes - P0 * 128;
        const unsigned xrec = (unsigned)(xrem > 0x7fffffffLL ? 0x7fffffffLL : xrem);
        __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + P0 * 128), 0, xrec, 0x00020000);
        for (int i = wave; i < NPI; i += 8) {
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(i * 1024));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(buf + i * 1024), 16,
                                                     pvoff0 ^ ((unsigned)(i & 1) << 6), so, 0, 0);
        }
    };

    // ---- per-lane channel constants: this lane's 8 channels of each 32-channel half -----------------
    f32x4 sc[2][2], bi[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ch = h * 32 + fq * 8 + q * 4;
            sc[h][q] = a.scale ? *(const f32x4*)(a.scale + ch) : f32x4{1.f, 1.f, 1.f, 1.f};
            bi[h][q] = a.bias ? *(const f32x4*)(a.bias + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
        }

    // per-tile, per-lane pixel state: patch-local centre, output / residual element offsets, store mask
    int pc[FM];
    long long oe[FM], re[FM];
    bool ok[FM];
    bf16x8 rv[FM][2];
    auto geometry = [&](int t) {
        const int m0 = t * a.bmv, mend = (m0 + a.bmv < a.M ? m0 + a.bmv : a.M);
        const long long P0 = centre(m0) - a.Wp - 1;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            int m = m0 + wave * 32 + i * 16 + fr;
            ok[i] = m < mend;
            m = ok[i] ? m : mend - 1;                  // idle slots recompute the tile's last pixel; never stored
            const int n = (int)fdiv(m, a.dOHW), r1 = m - n * HW;
            const int y = (int)fdiv(r1, a.dOW), x = r1 - y * a.W;
            pc[i] = (int)(((long long)n * a.Hp + y + a.in_halo) * a.Wp + x + a.in_halo - P0);
            oe[i] = (long long)n * a.out_sn + (long long)y * a.out_sh + (long long)x * a.out_sw + a.out_off0 + fq * 8;
            re[i] = (long long)n * a.res_sn + (long long)y * a.res_sh + (long long)x * a.res_sw + a.res_off0 + fq * 8;
        }
    };
    auto load_residual = [&]() {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                rv[i][h] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (a.res && ok[i]) rv[i][h] = *(const bf16x8*)(a.res + 2 * (re[i] + h * 32));
            }
    };
    f32x4 acc[FM][FN];
    auto epilogue = [&]() {                           // consumes acc, rv, oe, ok of the tile just multiplied
        bf16x8 ov[FM][2];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 v0 = acc[i][2 * h] * sc[h][0] + bi[h][0];
                f32x4 v1 = acc[i][2 * h + 1] * sc[h][1] + bi[h][1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float u0 = v0[e] + bf16_to_f32(rv[i][h][e]), u1 = v1[e] + bf16_to_f32(rv[i][h][4 + e]);
                    if (a.relu) { u0 = fmaxf(u0, 0.f); u1 = fmaxf(u1, 0.f); }
                    ov[i][h][e] = f32_to_bf16(u0);
                    ov[i][h][4 + e] = f32_to_bf16(u1);
                }
            }
        // all stores after the last use of a loaded value: no wait on a store sits between them
#pragma unroll
        for (int i = 0; i < FM; ++i)
            if (ok[i]) {
#pragma unroll
                for (int h = 0; h < 2; ++h) *(bf16x8*)(a.y + 2 * (oe[i] + h * 32)) = ov[i][h];
            }
    };
    // vmcnt(0) lgkmcnt(0) + barrier.  The wait is the builtin, not inline asm, so that the compiler's own
    // wait-count bookkeeping knows every load has returned and adds no vmcnt(0) of its own behind the
    // LDS-DMA pieces issued next.
    auto drain_and_barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

#if defined(ASL_STAMPS)
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
    unsigned long long sumK = 0, sumE = 0, sumO = 0, tq;
#endif
    dma_patch(t0, patch0);
    drain_and_barrier();                               // weights and the first patch have landed
#if defined(ASL_STAMPS)
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
    tq = st1;
#endif

    // Iteration t, entered behind a barrier every wave reached after its K loop of tile t-1 and after its own
    // pieces of patch t had landed:  epilogue of tile t-1 | DMA of patch t+1 (its buffer was tile t-1's) |
    // coordinates and residual loads of tile t | K loop of tile t | vmcnt(0) (everything waited for is at
    // least one K loop old) | barrier.
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        char* const pbuf = patch0 + ((t - t0) & 1) * kC64PatchBytes;
        // (the epilogue goes first: its residual registers are loaded values, and the compiler drains every
        // outstanding LDS-DMA piece before the first use of one)
        if (t > t0) epilogue();
        if (t + 1 < t1) dma_patch(t + 1, patch0 + ((t + 1 - t0) & 1) * kC64PatchBytes);
        geometry(t);
        load_residual();
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#if defined(ASL_STAMPS)
        { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sumE += n_ - tq; tq = n_; }
#endif
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3 - 1) * a.Wp + (tap % 3 - 1);
            const char* wb = wts + tap * 8192;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 xf[FM], wf[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    int p = pc[i];
                    asm volatile("" : "+v"(p));        // keeps the 18 per-tap addresses from being hoisted into VGPRs
                    p += toff;
                    xf[i] = *(const bf16x8*)(pbuf + p * 128 + (((kk * 4 + fq) ^ ((p >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) wf[j] = *(const bf16x8*)(wb + wrd[j][kk]);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            }
        }
#if defined(ASL_STAMPS)
        { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sumK += n_ - tq; tq = n_; }
#endif
        drain_and_barrier();
#if defined(ASL_STAMPS)
        { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sumO += n_ - tq; tq = n_; }
#endif
    }
    epilogue();
#if defined(ASL_STAMPS)
    // diagnostic builds: per workgroup {start, prologue end, + K loops, + epilogue/coordinate phases, + waits}
    if (a.stamps && tid == 0) {
        unsigned long long* o = a.stamps + (long long)blockIdx.x * 5;
        o[0] = st0; o[1] = st1; o[2] = st1 + sumK; o[3] = o[2] + sumE; o[4] = o[3] + sumO;
    }
#endif
#endif
}

// pixel slots per tile for the patch buffers (0 = geometry does not fit)
static int c64_tile(int n, int h, int w, int in_halo) {
    if (max_patch_pixels(n, h, w, in_halo, 256) <= kC64PatchPixels) return 256;
    const int hw = h * w;
    if (hw <= 256) {
        const int b = 256 / hw * hw;
        if (max_patch_pixels(n, h, w, in_halo, b) <= kC64PatchPixels) return b;
    }
    for (int k = 256 / w; k >= 1; --k)
        if (h % k == 0 && k * w * 4 >= 256 * 3 && max_patch_pixels(n, h, w, in_halo, k * w) <= kC64PatchPixels) return k * w;
    return 0;
}

// Runs a 64 -> 64 channel request on the persistent kernel; -1 when it does not apply.
int try_launch_conv3x3_c64(PatchArgs a, int n, hipStream_t stream) {
    if (a.C != 64 || a.N != 64 || getenv("ASL_NO_C64_CONV")) return -1;
    if (a.W <= 0 || a.H <= 0 || (int)a.dOW.d != a.W || (int)a.dOHW.d != a.H * a.W || a.Wp != a.W + 2 * a.in_halo ||
        a.Hp != a.H + 2 * a.in_halo || a.w_bytes < 2LL * 64 * a.wk || a.wk != 9 * 64)
        return -1;
    a.bmv = c64_tile(n, a.H, a.W, a.in_halo);
    if (!a.bmv) return -1;
    a.tiles_m = (a.M + a.bmv - 1) / a.bmv;
    a.tiles_n = 1;
    int ncu = 256;
    if (const char* e = getenv("ASL_C64_GRID")) ncu = atoi(e) > 0 ? atoi(e) : ncu;
    const int grid = a.tiles_m < ncu ? a.tiles_m : ncu;
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kC64Lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv3x3_c64_kernel, dim3((unsigned)grid), dim3(512), kC64Lds, stream, a);
    return (int)hipGetLastError();
}

}  // namespace asl_detail
