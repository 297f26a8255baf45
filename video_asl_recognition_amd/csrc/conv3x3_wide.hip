// 3x3 / stride 1 / pad 1 convolution, 16-wave "wide" ping-pong kernel for C >= 128 (gfx950).
//
// Same patch-in-LDS scheme as conv3x3_pp.hip (chunk-major half-chunk patch image, permuted weight rows,
// register-direct epilogue), rebuilt so that ONE weight stage serves 512 pixels while every wave keeps the
// 64-pixel x 64-channel register tile that lets four waves share a SIMD:
//
//  * 16 waves as 8 (pixels) x 2 (channels); tile = 512 pixel slots x 128 output channels; one workgroup per
//    CU (160 KiB of LDS: 3 x 16 KiB weight ring, 2 x 56 KiB patch images);
//  * the K loop is bound by how many LDS-DMA pieces a CU issues per MFMA: here a (tap, 32-channel) step of
//    512 pixels needs 8 weight pieces + 6.2 patch pieces, against 22 for two 256-pixel workgroups;
//  * weight stages travel in PAIRS (2 x 8 KiB = 16 pieces, one per wave) issued on even steps, the next
//    half-chunk's patch (56 pieces) one piece per wave on odd steps: every wave issues exactly one piece on
//    the steps that issue at all, with no role split and compile-time s_waitcnt counts;
//  * waves 0-7 / 8-15 (two of each per SIMD) run one phase apart, as in conv3x3_pp.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "asl_hip.h"
#include "common.h"
#include "patch_args.h"

namespace asl_detail {

struct WideShape {
    static constexpr int WPX = 8, WCO = 2, FM = 4, NW = WPX * WCO;
    static constexpr int TP = WPX * FM * 16;             // 512 pixel slots per tile
    static constexpr int TN = WCO * 64;                  // 128 output channels per tile
    static constexpr int PCAP = 896;                     // patch buffer, pixels (64 B each)
    static constexpr int PIECES = PCAP / 16;             // 56 LDS-DMA pieces per half-chunk patch
    static constexpr int PPW = (PIECES + NW - 1) / NW;   // 4 pieces per wave (the last round is partly surplus)
    // the surplus pieces of the last round zero-fill the buffer's last piece, which therefore never holds live pixels
    static constexpr int PLIVE = PCAP - 16;
    static constexpr int PATCH_BYTES = PCAP * 64;
    static constexpr int WST = TN * 64;                  // one weight stage: TN rows x 32 k = 8 KiB
    static constexpr int PAIR = 2 * WST;                 // ring slot: two consecutive stages, one piece per wave
    static constexpr int RING = 3;                       // pairs; prefetch distance 2 pairs = 4 steps
    static constexpr int LDS = RING * PAIR + 2 * PATCH_BYTES;
    static_assert(PAIR == NW * 1024 && LDS == 160 * 1024 && PPW == 4, "shape");
};

__global__ __launch_bounds__(1024) void conv3x3_wide_kernel(const PatchArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    using S = WideShape;
    constexpr int WCO = S::WCO, FM = S::FM, FN = 4;
    constexpr int WST = S::WST;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // weight ring first: its fragment reads then reach every slot through the instruction's 16-bit offset field
    char* const ring = smem;
    char* const patches = smem + S::RING * S::PAIR;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave / WCO, wc = wave % WCO;
    const bool late = wave >= 8;                        // the half that runs one phase behind
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
    const int m0 = tm * a.bmv;
    const int HW = a.H * a.W;
    auto centre = [&](int m) -> long long {             // halo'd, image-major pixel index of output pixel m
        const int n = (int)fdiv(m, a.dOHW), r1 = m - n * HW;
        const int y = (int)fdiv(r1, a.dOW), x = r1 - y * a.W;
        return ((long long)n * a.Hp + y + a.in_halo) * a.Wp + x + a.in_halo;
    };
    const int mend = (m0 + a.bmv < a.M ? m0 + a.bmv : a.M);
    const long long P0 = centre(m0) - a.Wp - 1;
    const int NP = (int)(centre(mend - 1) + a.Wp + 1 - P0) + 1;       // <= PCAP (checked on the host)
    const int nhc = a.C >> 5;                                         // 32-channel half-chunks

    // ---- DMA descriptors and per-lane source offsets ------------------------------------------
    const long long pix_bytes = 2LL * a.C;
    const char* xbase = a.x + P0 * pix_bytes;
    long long xrem = a.x_bytes - P0 * pix_bytes;
    const unsigned xrec = (unsigned)(xrem > 0x7fffffffLL ? 0x7fffffffLL : xrem);
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, xrec, 0x00020000);
    const char* wbase = a.w + 2LL * (tn * S::TN) * a.wk;
    const unsigned wrec = (unsigned)(a.w_bytes - 2LL * (tn * S::TN) * a.wk);
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, wrec, 0x00020000);
    constexpr unsigned kOob = 0x80000000u;              // beyond num_records: the DMA writes zeros

    // Patch image in LDS: piece j (1 KiB) holds patch pixels 16j..16j+15 chunk-major, byte (p, c) at
    // (p >> 4) * 1024 + c * 256 + (p & 15) * 16 (c = 16-byte chunk of the pixel's 64).  The 16-byte slot within
    // the 256-byte bank row is p & 15 whatever the chunk, so a fragment read of 16 consecutive pixels is
    // conflict-free with no address swizzle.  DMA lane -> pixel 16j + lane % 16, chunk lane / 16.
    const unsigned pvoff = (unsigned)((lane & 15) * pix_bytes + (lane >> 4) * 16);
    auto dma_patch_piece = [&](int hc, int k) {         // piece (wave + 16k) of half-chunk hc
        const int j = wave + 16 * k;
        const bool live = hc < nhc && 16 * j < NP;      // else: zeros, into the piece itself or the spare last piece
        const int jd = j < S::PIECES ? j : S::PIECES - 1;
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(16LL * j * pix_bytes + hc * 64));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(patches + (hc & 1) * S::PATCH_BYTES + jd * 1024),
                                                 16, live ? pvoff : kOob, live ? so : 0u, 0, 0);
    };
    // Weight pair p = stages 2p, 2p+1 -> ring slot p % RING.  Wave w issues piece (w & 7) of stage 2p + (w >> 3):
    // tile rows 16 (w & 7) .. +15 (64 B each), chunk swizzle (row & 8) ? 3 : 0.  Pairs past the last stage are
    // still issued: they read the first half-chunk of a later tap of the same rows, in bounds, into slots
    // nobody reads any more.
    const int w8 = wave & 7, h8 = wave >> 3;
    unsigned wvoff;
    {
        const int row = w8 * 16 + (lane >> 2);
        wvoff = (unsigned)(2 * row * a.wk + 16 * ((lane & 3) ^ ((row & 8) ? 3 : 0)));
    }
    auto dma_weight_pair = [&](int p, int slot) {
        const int s = 2 * p + h8;
        const int hc = s / 9, tap = s - 9 * hc;
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(2 * (tap * a.C + hc * 32)));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(ring + slot * S::PAIR + wave * 1024), 16,
                                                 wvoff, so, 0, 0);
    };

    // ---- per-lane fragment geometry ----------------------------------------------------------
    const int fr = lane & 15, fq = lane >> 4;
    // u[i]: image address of (top-left tap pixel, chunk 3) of this lane's pixel in fragment i.  The chunk field
    // (bits 8-9) is all ones, so adding a tap's (pixel >> 4) << 10 | (pixel & 15) << 4 carries from the low
    // pixel bits straight into the high ones; the lane's own chunk is then inserted over the field.
    int u[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        int m = m0 + wp * (FM * 16) + i * 16 + fr;
        m = m < mend ? m : mend - 1;                    // idle slots recompute the tile's last pixel; never stored
        const int tl = (int)(centre(m) - P0) - a.Wp - 1;
        u[i] = ((tl >> 4) << 10) | 0x300 | ((tl & 15) << 4);
    }
    const int chunk_bits = fq << 8;
    constexpr int chunk_mask = 0x300;
    // LDS address of the dynamic segment's first byte
    const int lds0 = (int)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    // Weight rows are read in a permuted order so that a lane's accumulators for channel fragments 2h and
    // 2h+1 are eight CONSECUTIVE output channels of its pixel (MFMA row r of fragment j is channel
    // (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3) of the wave's 64): the epilogue stores 16 bytes per lane straight
    // from registers.  The 64-byte-row swizzle is conflict-free for this order too.
    int wrd[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int row = wc * 64 + (j >> 1) * 32 + (fr >> 2) * 8 + (j & 1) * 4 + (fr & 3);
        wrd[j] = row * 64 + ((fq ^ ((row & 8) ? 3 : 0)) << 4);
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: patch of half-chunk 0, weight pairs 0 and 1 -----------------------------------
#pragma unroll
    for (int k = 0; k < S::PPW; ++k) dma_patch_piece(0, k);
    dma_weight_pair(0, 0);
    dma_weight_pair(1, 1);
    asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");                     // patch 0 and pair 0 landed
    if (late) asm volatile("s_barrier" ::: "memory");                                 // run one phase behind
#if defined(ASL_STAMPS)
    st[1] = __builtin_amdgcn_s_memtime();
#endif

    // ---- K loop ---------------------------------------------------------------------------------
    // The loop body covers two half-chunks = 18 steps q (C is a multiple of 64 here).  Step q: load phase =
    // {the 8 fragment reads of the step, at most one DMA piece, counted wait} | barrier | 16 MFMAs | barrier.
    // Even q issues weight pair q/2 + 2 of this body; odd q in kPatchSteps issues one piece of the following
    // half-chunk's patch (4 per wave and half-chunk: odd taps of the first, even taps < 8 of the second).
    int hp = 0;                                         // index of the half-chunk pair
    int pboff = lds0 + (int)(S::RING * S::PAIR);
    auto is_patch_step = [](int q) constexpr { return q == 1 || q == 3 || q == 5 || q == 7 || q == 9 || q == 11 || q == 13 || q == 15; };
    auto step = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int tap = q % 9, hcl = q / 9;          // tap and half-chunk within the pair
        constexpr int slot = (q / 2) % 3, half = q & 1;  // 9 pairs per body = 0 (mod 3)
        const int o = (tap / 3) * a.Wp + tap % 3;       // tap pixel relative to the top-left one
        const int so = __builtin_amdgcn_readfirstlane((((o >> 4) << 10) | ((o & 15) << 4)) + pboff);
        const char* wb = ring + slot * S::PAIR + half * WST;
        bf16x8 xf[FM], wf[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = *(const bf16x8*)(wb + wrd[j]);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            // two vector instructions per address.  The tap term passes through an empty asm per use: otherwise
            // the compiler hoists every tap address out of the K loop into registers.
            int so_i = so;
            asm volatile("" : "+s"(so_i));
            const int r = u[i] + so_i;
            const int ad = (r & ~chunk_mask) | chunk_bits;
            xf[i] = *(const __attribute__((address_space(3))) bf16x8*)(unsigned long long)(unsigned)ad;
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((q & 1) == 0) {
            dma_weight_pair(9 * hp + q / 2 + 2, (q / 2 + 2) % 3);
        } else if constexpr (is_patch_step(q)) {
            dma_patch_piece(2 * hp + hcl + 1, (q % 9) / 2);     // k = 0..3 within either half-chunk
        }
        // Counted waits (a wave's DMA pieces retire in issue order).  Odd q: the pair issued at q - 3 (read
        // from the next step on) has landed; everything issued after it may stay in flight.  q = 8 and 17: the
        // patch of the next half-chunk has landed (only the newest weight piece is younger).
        constexpr int qm2 = (q + 16) % 18;
        constexpr int allow_w = 1 + (is_patch_step(qm2) ? 1 : 0) + (is_patch_step(q) ? 1 : 0);
        if constexpr (q == 8 || q == 17) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if constexpr (q & 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(allow_w) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        if constexpr (q == 8) pboff = lds0 + (int)(S::RING * S::PAIR) + S::PATCH_BYTES;
    };
    const int npairs = nhc >> 1;
#pragma unroll 1
    for (hp = 0; hp < npairs; ++hp) {
        pboff = lds0 + (int)(S::RING * S::PAIR);
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{});
        step(std::integral_constant<int, 9>{});
        step(std::integral_constant<int, 10>{});
        step(std::integral_constant<int, 11>{});
        step(std::integral_constant<int, 12>{});
        step(std::integral_constant<int, 13>{});
        step(std::integral_constant<int, 14>{});
        step(std::integral_constant<int, 15>{});
        step(std::integral_constant<int, 16>{});
        step(std::integral_constant<int, 17>{});
    }
    if (!late) asm volatile("s_barrier" ::: "memory");  // the early half waits out the late half's last phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // trailing zero-fill pieces have landed
#if defined(ASL_STAMPS)
    st[2] = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue, straight from registers: per 32-channel half, the residual loads of all pixel fragments
    // are issued first, then scale/bias, residual, ReLU and one 16-byte store per (fragment, half).  Output
    // and residual are addressed through buffer descriptors: masked lanes use an out-of-range offset
    // (stores dropped, loads return 0), so nothing branches.  The host guarantees both are under 4 GiB.
    {
        constexpr unsigned kOobOff = 0xffffffffu;
        __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.y), 0,
                                                                       a.res ? (unsigned)a.res_bytes : 0u, 0x00020000);
        const float floor_v = a.relu ? 0.f : -__builtin_inff();
        const int ch0 = tn * S::TN + wc * 64 + fq * 8;   // first of this lane's 8 channels in half 0
        unsigned oe[FM], re[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            int m = m0 + wp * (FM * 16) + i * 16 + fr;
            asm volatile("" : "+v"(m));                  // recompute (n, y, x) here: reusing the prologue's values
                                                        // would carry 3 registers per fragment across the K loop
            const bool ok = m < mend;
            m = ok ? m : mend - 1;
            const int n = (int)fdiv(m, a.dOHW), r1 = m - n * HW;
            const int y = (int)fdiv(r1, a.dOW), x = r1 - y * a.W;
            const long long o = (long long)n * a.out_sn + (long long)y * a.out_sh + (long long)x * a.out_sw + a.out_off0 + ch0;
            const long long r = (long long)n * a.res_sn + (long long)y * a.res_sh + (long long)x * a.res_sw + a.res_off0 + ch0;
            oe[i] = ok ? (unsigned)(2 * o) : kOobOff;
            re[i] = ok ? (unsigned)(2 * r) : kOobOff;
        }
        // (FM = 4: all eight loads fit the registers the fragments have just vacated, one exposed latency)
        constexpr int HB = FM <= 4 ? 2 : 1;              // halves whose residual loads are batched
        i32x4 rv[HB][FM];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h % HB == 0) {
#pragma unroll
                for (int hh = 0; hh < HB; ++hh)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
                        rv[hh][i] = __builtin_amdgcn_raw_buffer_load_b128(rrs, re[i] == kOobOff ? kOobOff : re[i] + (h + hh) * 64, 0, 0);
            }
            f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, bi0 = {0.f, 0.f, 0.f, 0.f}, bi1 = bi0;
            if (a.scale) { sc0 = *(const f32x4*)(a.scale + ch0 + h * 32); sc1 = *(const f32x4*)(a.scale + ch0 + h * 32 + 4); }
            if (a.bias) { bi0 = *(const f32x4*)(a.bias + ch0 + h * 32); bi1 = *(const f32x4*)(a.bias + ch0 + h * 32 + 4); }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const f32x4 v0 = acc[i][2 * h] * sc0 + bi0;
                const f32x4 v1 = acc[i][2 * h + 1] * sc1 + bi1;
                const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[h % HB][i]);
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ov[e] = f32_to_bf16(fmaxf(v0[e] + bf16_to_f32(r8[e]), floor_v));
                    ov[4 + e] = f32_to_bf16(fmaxf(v1[e] + bf16_to_f32(r8[4 + e]), floor_v));
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, ov), yrs, oe[i] == kOobOff ? kOobOff : oe[i] + h * 64, 0, 0);
            }
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


// Runs the request on the 16-wave kernel when it fits: C a multiple of 64 and at least 128, N a multiple of
// 128, a 512-pixel (or whole-image) tile whose patch fits.  Returns -1 when it does not apply.
int try_launch_conv3x3_wide(PatchArgs a, int n, hipStream_t stream) {
    using S = WideShape;
    if (a.variant != ASL_CONV_AUTO && a.variant != ASL_CONV_WIDE) return -1;
    if (a.C % 64 || a.C < 128 || a.N % 128) return -1;
    if (a.variant == ASL_CONV_AUTO && a.C < 512) return -1;   // measured: ahead of the 8-wave kernel from 512 channels up
    if (a.y_bytes <= 0 || a.y_bytes >= 0xffffffffLL || (a.res && (a.res_bytes <= 0 || a.res_bytes >= 0xffffffffLL))) return -1;
    if (a.W <= 0 || a.H <= 0 || (int)a.dOW.d != a.W || (int)a.dOHW.d != a.H * a.W || a.Wp != a.W + 2 * a.in_halo ||
        a.Hp != a.H + 2 * a.in_halo)
        return -1;
    // the trailing weight pairs read up to four stages past the last one: keep that inside the weight rows
    if (a.wk < 9 * a.C) return -1;
    a.bmv = 0;
    if (max_patch_pixels(n, a.H, a.W, a.in_halo, S::TP) <= S::PLIVE) a.bmv = S::TP;
    else if (a.H * a.W <= S::TP) {
        const int b = S::TP / (a.H * a.W) * (a.H * a.W);
        if (b * 8 >= S::TP * 7 && max_patch_pixels(n, a.H, a.W, a.in_halo, b) <= S::PLIVE) a.bmv = b;
    }
    if (!a.bmv) return -1;
    a.tiles_m = (a.M + a.bmv - 1) / a.bmv;
    a.tiles_n = (a.N + S::TN - 1) / S::TN;
    // one workgroup per CU: a grid that leaves a quarter of the CUs idle is better served by the 256-pixel
    // workgroups of the 8-wave kernel (layer 4, 256 frames: 104 tiles, 0.090 ms here against 0.065 there)
    if (a.variant == ASL_CONV_AUTO && (long long)a.tiles_m * a.tiles_n * 4 < 3LL * device_cus()) return -1;
    static unsigned char lds_set[64];
    if (const int e = ensure_dynamic_lds((const void*)conv3x3_wide_kernel, S::LDS, lds_set)) return e;
    hipLaunchKernelGGL(conv3x3_wide_kernel, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(1024), S::LDS, stream, a);
    return (int)hipGetLastError();
}

}  // namespace asl_detail
