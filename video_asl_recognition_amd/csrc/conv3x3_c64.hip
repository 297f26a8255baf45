// 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels: persistent kernel with the whole filter
// bank resident in LDS (gfx950).
//
// A 64 -> 64 layer has only nine K steps per tile and moves 86 KB of HBM traffic per 256-pixel tile, so a
// per-tile kernel spends most of its time in prologues and epilogues.  Here one 8-wave workgroup per CU
// walks a contiguous range of tiles:
//  * the 72 KiB of weights are copied to LDS once;
//  * waves 0-3 and waves 4-7 (one of each per SIMD) are two groups that take alternate tiles and run in
//    anti-phase: while one group multiplies its tile (MFMA, LDS reads; no barrier and no wait on global
//    memory inside), the other stores its previous tile, computes the coordinates of its next one and
//    streams that tile's halo'd input patch into its own LDS patch buffer -- so each SIMD always has one
//    wave feeding the matrix pipe and one wave doing vector/scalar work; one workgroup barrier per phase;
//  * weight rows are read in a permuted order so each lane owns eight consecutive output channels and the
//    epilogue stores 16 bytes per lane straight from registers (no LDS staging); residuals are prefetched
//    two phases ahead; output and residual are addressed through buffer descriptors so masked lanes need
//    no branches.
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
// (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3).
__device__ __forceinline__ int c64_row(int j, int r) { return (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3); }
// chunk swizzle of the resident weight rows, conflict-free for that read order
__device__ __forceinline__ int c64_wswz(int row) { return (row & 3) | ((row >> 1) & 4); }

__global__ __launch_bounds__(512) void conv3x3_c64_kernel(const PatchArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int FM = 4, FN = 4;                      // wave tile: 64 pixel slots x 64 channels; a group of 4 waves = 256 slots
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wts = smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;          // group (tile parity) and wave within the group
    const int fr = lane & 15, fq = lane >> 4;
    char* const pbuf = smem + kC64WeightBytes + grp * kC64PatchBytes;      // this group's patch buffer

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
    int wrd[FN];                                       // kk = 1 is the same address with bit 6 flipped
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int row = c64_row(j, fr);
        wrd[j] = row * 128 + ((fq ^ c64_wswz(row)) << 4);
    }

    // ---- patch DMA of tile t into this group's buffer: piece i covers patch pixels 8i..8i+7 (128 B each);
    // slot (p, c') holds chunk c' ^ ((p >> 1) & 7); the group's four waves share the pieces
    const unsigned pvoff0 = (unsigned)((lane >> 3) * 128 + 16 * ((lane & 7) ^ (lane >> 4)));
    auto dma_patch = [&](int t) {
        const int m0 = t * a.bmv, mend = (m0 + a.bmv < a.M ? m0 + a.bmv : a.M);
        const long long P0 = centre(m0) - a.Wp - 1;
        const int NP = (int)(centre(mend - 1) + a.Wp + 1 - P0) + 1;
        const int NPI = (NP + 7) >> 3;
        long long xrem = a.x_bytes - P0 * 128;
        const unsigned xrec = (unsigned)(xrem > 0x7fffffffLL ? 0x7fffffffLL : xrem);
        __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + P0 * 128), 0, xrec, 0x00020000);
        for (int i = w4; i < NPI; i += 4) {
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(i * 1024));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(pbuf + i * 1024), 16,
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
    const float floor_v = a.relu ? 0.f : -__builtin_inff();      // ReLU as one max per value
    // Output and residual go through buffer descriptors: a masked lane uses an out-of-range offset (stores are
    // dropped, loads return 0), so neither needs a branch.  The host guarantees both tensors are under 4 GiB.
    constexpr unsigned kOob = 0xffffffffu;
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.y), 0,
                                                                   a.res ? (unsigned)a.res_bytes : 0u, 0x00020000);

    // per-lane pixel state of a tile: byte offset of (centre pixel, chunk fq) in an unswizzled patch, and the
    // output / residual byte offsets (kOob = masked)
    int pc[FM];
    unsigned oe[FM], re[FM];
    i32x4 rv[FM][2];
    auto geometry = [&](int t) {                       // t may be past the range: everything masked
        const bool live = t < t1;
        int m0 = t * a.bmv;
        m0 = live ? m0 : t0 * a.bmv;
        const int mend = (m0 + a.bmv < a.M ? m0 + a.bmv : a.M);
        const long long P0 = centre(m0) - a.Wp - 1;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            int m = m0 + w4 * 64 + i * 16 + fr;
            const bool ok = live && m < mend;
            m = m < mend ? m : mend - 1;               // idle slots recompute the tile's last pixel; never stored
            const int n = (int)fdiv(m, a.dOHW), r1 = m - n * HW;
            const int y = (int)fdiv(r1, a.dOW), x = r1 - y * a.W;
            pc[i] = (int)(((long long)n * a.Hp + y + a.in_halo) * a.Wp + x + a.in_halo - P0) * 128 + fq * 16;
            const long long o = (long long)n * a.out_sn + (long long)y * a.out_sh + (long long)x * a.out_sw + a.out_off0 + fq * 8;
            const long long r = (long long)n * a.res_sn + (long long)y * a.res_sh + (long long)x * a.res_sw + a.res_off0 + fq * 8;
            oe[i] = ok ? (unsigned)(2 * o) : kOob;
            re[i] = ok ? (unsigned)(2 * r) : kOob;
        }
    };
    auto load_residual = [&]() {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                rv[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rrs, re[i] == kOob ? kOob : re[i] + h * 64, 0, 0);
    };
    f32x4 acc[FM][FN];
    auto epilogue = [&]() {                           // scale/bias, residual, ReLU, 16-byte stores; consumes acc, rv, oe
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v0 = acc[i][2 * h] * sc[h][0] + bi[h][0];
                const f32x4 v1 = acc[i][2 * h + 1] * sc[h][1] + bi[h][1];
                const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[i][h]);
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ov[e] = f32_to_bf16(fmaxf(v0[e] + bf16_to_f32(r8[e]), floor_v));
                    ov[4 + e] = f32_to_bf16(fmaxf(v1[e] + bf16_to_f32(r8[4 + e]), floor_v));
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, ov), yrs, oe[i] == kOob ? kOob : oe[i] + h * 64, 0, 0);
            }
    };
    auto multiply = [&]() {                           // K loop of one tile: 9 taps x 64 channels from LDS only
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = ((tap / 3 - 1) * a.Wp + (tap % 3 - 1)) * 128;
            const char* wb = wts + tap * 8192;
            bf16x8 xf[2][FM], wf[2][FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                // byte offset u = 128 p + 16 fq; the chunk swizzle (p >> 1) & 7 is bits 8..10 of u moved to bits 4..6;
                // the second K half is the same address with bit 6 flipped
                int u = pc[i];
                asm volatile("" : "+v"(u));            // keeps the per-tap addresses from being hoisted into VGPRs
                u += toff;
                u ^= (u >> 4) & 0x70;
                xf[0][i] = *(const bf16x8*)(pbuf + u);
                xf[1][i] = *(const bf16x8*)(pbuf + (u ^ 64));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                wf[0][j] = *(const bf16x8*)(wb + wrd[j]);
                wf[1][j] = *(const bf16x8*)(wb + (wrd[j] ^ 64));
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][j], xf[kk][i], acc[i][j], 0, 0, 0);
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

    // ---- group g takes tiles t0 + g, t0 + g + 2, ...; its k-th tile is multiplied in phase 2k + g and stored in
    // phase 2k + g + 1, so in every phase one group multiplies while the other stores / prefetches
    const int ntl = t1 - t0;
    const int mine = (ntl - grp + 1) >> 1;             // tiles of this group
    const int nphases = ntl + 1;                       // last multiply is phase ntl - 1, its store phase ntl
    int t = t0 + grp;                                  // next tile to multiply
    if (mine > 0) dma_patch(t);
    geometry(t);
    load_residual();
    drain_and_barrier();                               // weights, both first patches and residuals have landed

#pragma unroll 1
    for (int ph = 0; ph < nphases; ++ph) {
        const int k2 = ph - grp;                       // even and in range: multiply phase of this group's tile k2 / 2
        if (k2 >= 0 && !(k2 & 1) && (k2 >> 1) < mine) {
            multiply();                                // tile t
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();              // every wave of the group is done reading the patch
            asm volatile("" ::: "memory");
        } else if (k2 > 0 && (k2 & 1) && (k2 >> 1) < mine) {
            epilogue();                                // tile t: stores first, they have the whole phase to complete
            t += 2;
            if (t < t1) dma_patch(t);                  // the group's next patch; waited for at the end of this phase
            geometry(t);
            load_residual();
            // the patch pieces (and the stores before them) have landed; the FM * 2 residual loads just issued
            // stay in flight through the multiply phase: vmcnt(8) lgkmcnt(0)
            static_assert(FM * 2 == 8, "counted wait below assumes eight residual loads");
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0078);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();              // nothing to do in this phase (first / last phases of a group)
            asm volatile("" ::: "memory");
        }
    }
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
    if (a.variant != ASL_CONV_AUTO && a.variant != ASL_CONV_C64) return -1;
    if (a.C != 64 || a.N != 64) return -1;
    if (a.y_bytes <= 0 || a.y_bytes >= 0xffffffffLL || (a.res && (a.res_bytes <= 0 || a.res_bytes >= 0xffffffffLL))) return -1;
    if (a.W <= 0 || a.H <= 0 || (int)a.dOW.d != a.W || (int)a.dOHW.d != a.H * a.W || a.Wp != a.W + 2 * a.in_halo ||
        a.Hp != a.H + 2 * a.in_halo || a.w_bytes < 2LL * 64 * a.wk || a.wk != 9 * 64)
        return -1;
    a.bmv = c64_tile(n, a.H, a.W, a.in_halo);
    if (!a.bmv) return -1;
    a.tiles_m = (a.M + a.bmv - 1) / a.bmv;
    a.tiles_n = 1;
    const int ncu = device_cus();                      // one persistent workgroup per CU
    const int grid = a.tiles_m < ncu ? a.tiles_m : ncu;
    static unsigned char lds_set[64];
    if (const int e = ensure_dynamic_lds((const void*)conv3x3_c64_kernel, kC64Lds, lds_set)) return e;
    hipLaunchKernelGGL(conv3x3_c64_kernel, dim3((unsigned)grid), dim3(512), kC64Lds, stream, a);
    return (int)hipGetLastError();
}

}  // namespace asl_detail
