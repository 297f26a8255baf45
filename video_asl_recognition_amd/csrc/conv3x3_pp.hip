// 3x3 / stride 1 / pad 1 convolution, large-tile "ping-pong" kernel for C >= 128 (gfx950).
//
// Same idea as conv3x3_patch.hip -- the halo'd input range of a run of consecutive output pixels sits
// in LDS and all nine taps read it at constant offsets -- rebuilt around 8-wave workgroups:
//
//  * 8 waves as 4 (pixels) x 2 (channels); tile = 512 or 256 pixel slots x 128 output channels, every wave a
//    128- or 64-pixel x 64-channel register tile;
//  * K advances in steps of (32-channel half-chunk, tap): the patch of a half-chunk is 64 B per pixel,
//    double-buffered, and the next half-chunk's patch streams in one LDS-DMA piece per step while the
//    current one is consumed -- no reload bubble between chunks;
//  * weight stages (tile channels x 32 k, 64-byte rows) run through a 3- or 4-deep LDS-DMA ring, two or
//    three stages ahead of the reads, retired by counted s_waitcnt vmcnt so loads stay in flight
//    across every barrier;
//  * the two halves of the workgroup (waves 0-3 / 4-7, one of each per SIMD) run one phase apart:
//    while one half issues its DMA pieces and fragment reads, the other half runs its MFMAs, and
//    a raw s_barrier swaps the roles.
//
// MFMA orientation and weight packing are those of gather_gemm.hip.
//
// KD = 3 instance: 3x3x3 / stride 1 / pad 1 on halo'd NDHWC.  Depth planes of a clip lie behind one another exactly as
// the images of a 2-D batch do, so the patch of a run of output pixels for ONE depth tap is the 2-D patch shifted by
// whole planes: K walks (half-chunk, depth tap) pairs of nine steps each -- 27 taps at constant LDS offsets -- and
// everything else is the 2-D kernel.  Tiles never cross a clip (the halo planes between clips would not fit a patch).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "asl_hip.h"
#include "common.h"
#include "patch_args.h"

namespace asl_detail {

// Shape BIG: 512 pixels x 128 channels, 128-pixel wave tiles, 144 KiB of LDS -> one workgroup per CU.
// Shape TWIN: 256 pixels x 128 channels, 64-pixel wave tiles, 80 KiB and <= 128 VGPRs -> two workgroups per
// CU, each covering the other's prologue and epilogue.
template <int FM_>
struct PpShape {
    static constexpr int WPX = 4, WCO = 2, FM = FM_;
    static constexpr int TP = WPX * FM * 16;             // pixel slots per tile
    static constexpr int TN = WCO * 64;                  // output channels per tile
    static constexpr int PCAP = FM == 8 ? 896 : 448;     // patch buffer, pixels (64 B each)
    static constexpr int PIECES = PCAP / 16;             // 1 KiB LDS-DMA pieces per half-chunk patch
    static constexpr int PPW = (PIECES + 7) / 8;         // pieces per wave (<= 7: one per tap step)
    // when PIECES is not a multiple of 8 the surplus pieces of the last round zero-fill the buffer's last
    // piece, which therefore never holds live pixels
    static constexpr int PLIVE = PIECES % 8 ? PCAP - 16 : PCAP;
    static constexpr int PATCH_BYTES = PCAP * 64;
    static constexpr int WST = TN * 64;                  // one weight stage: TN rows x 32 k
    static constexpr int WPW = WST / 1024 / 8;           // weight DMA pieces per wave per stage
    static constexpr int RING = FM == 8 ? 4 : 3;         // weight ring depth; prefetch distance RING - 1
    static constexpr int LDS = 2 * PATCH_BYTES + RING * WST;
    static constexpr int WAVES_PER_EU = FM == 8 ? 2 : 4;
    static_assert(PPW >= 1 && PPW <= 7 && WPW >= 1, "piece counts");
};

template <int FM_, int KD>
__global__ __launch_bounds__(512, PpShape<FM_>::WAVES_PER_EU) void conv3x3_pp_kernel(const PatchArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    using S = PpShape<FM_>;
    constexpr int WCO = S::WCO, FM = S::FM, FN = 4;
    constexpr int PPW = S::PPW, WPW = S::WPW, WST = S::WST, RING = S::RING, DIST = RING - 1;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // weight ring first: its fragment reads then reach every slot through the instruction's 16-bit offset field
    char* const ring = smem;
    char* const patches = smem + S::RING * S::WST;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave / WCO, wc = wave % WCO;
    const bool late = wave >= 4;                        // the half that runs one phase behind
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
    const int HW = a.H * a.W;
    int m0, mend;
    if constexpr (KD == 3) {                            // tile tm = (clip, t): pixels t*bmv .. of that clip
        const int clip = (int)fdiv(tm, a.dTPC), t = tm - clip * a.tiles_per_clip;
        const int DHW = a.D * HW;
        m0 = clip * DHW + t * a.bmv;
        mend = (t + 1) * a.bmv < DHW ? m0 + a.bmv : (clip + 1) * DHW;
    } else {
        m0 = tm * a.bmv;
        mend = (m0 + a.bmv < a.M ? m0 + a.bmv : a.M);
    }
    auto centre = [&](int m) -> long long {             // halo'd, image-major pixel index of output pixel m
        if constexpr (KD == 3) {
            const int n = (int)fdiv(m, a.dODHW), r = m - n * (a.D * HW);
            const int d = (int)fdiv(r, a.dOHW), r1 = r - d * HW;
            const int y = (int)fdiv(r1, a.dOW), x = r1 - y * a.W;
            return (((long long)n * a.Dp + d + a.in_halo_d) * a.Hp + y + a.in_halo) * a.Wp + x + a.in_halo;
        } else {
            const int n = (int)fdiv(m, a.dOHW), r1 = m - n * HW;
            const int y = (int)fdiv(r1, a.dOW), x = r1 - y * a.W;
            return ((long long)n * a.Hp + y + a.in_halo) * a.Wp + x + a.in_halo;
        }
    };
    // P0: first patch pixel of the tile in the plane of the centre depth tap; the descriptor starts one plane earlier
    const long long plane_px = KD == 3 ? (long long)a.Hp * a.Wp : 0;
    const long long P0 = centre(m0) - a.Wp - 1;
    const int NP = (int)(centre(mend - 1) + a.Wp + 1 - P0) + 1;       // <= PCAP (checked on the host)
    const int nhc = a.C >> 5;                                         // 32-channel half-chunks
    const int nq = nhc * KD;                                          // (half-chunk, depth tap) pairs, nine steps each
    const int nsteps = nq * 9;

    // ---- DMA descriptors and per-lane source offsets ------------------------------------------
    const long long pix_bytes = 2LL * a.C;
    const char* xbase = a.x + (P0 - plane_px) * pix_bytes;
    long long xrem = a.x_bytes - (P0 - plane_px) * pix_bytes;
    const unsigned plane_bytes = (unsigned)(plane_px * pix_bytes);
    const unsigned xrec = (unsigned)(xrem > 0x7fffffffLL ? 0x7fffffffLL : xrem);
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, xrec, 0x00020000);
    const char* wbase = a.w + 2LL * (tn * S::TN) * a.wk;
    const unsigned wrec = (unsigned)(a.w_bytes - 2LL * (tn * S::TN) * a.wk);
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, wrec, 0x00020000);
    // Pieces past the end -- of the tile's patch, of the last half-chunk, of the last weight stage -- are issued like any
    // other (every wave issues the same number of pieces per step: the waits below count them) but must not touch
    // memory: the scalar offset of a buffer access is outside the descriptor's range check, only the per-lane offset is
    // inside it.  Their per-lane offset gets bit 31 set (beyond num_records: the DMA writes zeros) by OR-ing a
    // wave-uniform mask, which costs no register across the loop (a select against a constant did: in the 3-D instance
    // the two DMA offsets ended up in scratch and were reloaded at the top of every iteration -- one more operation on
    // the counter the waits count, with an s_waitcnt vmcnt(0) in front of its first use).
    constexpr unsigned kOob = 0x80000000u;

    // Patch image in LDS: piece j (1 KiB) holds patch pixels 16j..16j+15 chunk-major, byte (p, c) at
    // (p >> 4) * 1024 + c * 256 + (p & 15) * 16 (c = 16-byte chunk of the pixel's 64).  The 16-byte slot within
    // the 256-byte bank row is p & 15 whatever the chunk, so a fragment read of 16 consecutive pixels is
    // conflict-free with no address swizzle.  DMA lane -> pixel 16j + lane % 16, chunk lane / 16.
    const unsigned pvoff = (unsigned)((lane & 15) * pix_bytes + (lane >> 4) * 16);
    auto dma_patch_piece = [&](int q, int k) {          // piece (wave + 8k) of pair q = (half-chunk q / KD, depth tap q % KD)
        const int j = wave + 8 * k;
        const unsigned dead = __builtin_amdgcn_readfirstlane(q < nq && 16 * j < NP ? 0u : kOob);
        const int jd = j < S::PIECES ? j : S::PIECES - 1;    // surplus pieces of the last round: into the spare last piece
        const int hc = KD == 3 ? q / 3 : q, kt = q - hc * KD;
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(16LL * j * pix_bytes + hc * 64) + (unsigned)kt * plane_bytes);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(patches + (q & 1) * S::PATCH_BYTES + jd * 1024),
                                                 16, pvoff | dead, dead ? 0u : so, 0, 0);
    };
    // weight piece q = wave * WPW + i covers tile rows 16q..16q+15 (64 B each); chunk swizzle (row & 8) ? 3 : 0
    unsigned wvoff[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int row = (wave * WPW + i) * 16 + (lane >> 2);
        wvoff[i] = (unsigned)(2 * row * a.wk + 16 * ((lane & 3) ^ ((row & 8) ? 3 : 0)));
    }
    auto dma_weights = [&](int s, int slot) {           // stage s = q * 9 + tap -> ring slot s % RING
        const unsigned dead = __builtin_amdgcn_readfirstlane(s < nsteps ? 0u : kOob);
        const int q = s / 9, tap = s - 9 * q;
        const int hc = KD == 3 ? q / 3 : q, kt = q - hc * KD;
        const unsigned so = __builtin_amdgcn_readfirstlane(dead ? 0u : (unsigned)(2 * ((kt * 9 + tap) * a.C + hc * 32)));
        char* dst = ring + slot * WST + wave * (WPW * 1024);
#pragma unroll
        for (int i = 0; i < WPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                                     wvoff[i] | dead, so, 0, 0);
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
    // (fragment j reads row (j >> 1) * 32 + (j & 1) * 4 + row0; bit 3 of the row, which picks the swizzle, is the same for
    // all four, so they are constant offsets from ONE address register)
    int wrd0;
    {
        const int row0 = wc * 64 + (fr >> 2) * 8 + (fr & 3);
        wrd0 = row0 * 64 + ((fq ^ ((row0 & 8) ? 3 : 0)) << 4);
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: patch of half-chunk 0, weight stages 0..2 ------------------------------------
#pragma unroll
    for (int k = 0; k < PPW; ++k) dma_patch_piece(0, k);
#pragma unroll
    for (int d = 0; d < DIST; ++d) dma_weights(d, d);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((DIST - 1) * WPW) : "memory");   // patch 0 and stage 0 landed
    if (late) asm volatile("s_barrier" ::: "memory");                                 // run one phase behind
#if defined(ASL_STAMPS)
    st[1] = __builtin_amdgcn_s_memtime();
#endif

    // ---- K loop ---------------------------------------------------------------------------------
    // Step s: load phase L(s) = {DMA pieces of stage s+3 and of the next half-chunk's patch, the 12 fragment
    // reads of step s, counted wait retiring stage s+1} | barrier | compute phase = 32 MFMAs | barrier.
    // The wait leaves the pieces issued in L(s-1) and L(s) in flight: I(t) = WPW + (t < PPW) pieces in L of tap t.
#if defined(ASL_STAMPS_PHASES)
    unsigned long long ph[4] = {0, 0, 0, 0};   // diagnostic: issue (DMA + reads), wait + barrier, MFMA issue, second barrier
#endif
    int hc = 0;
    int pboff = lds0 + (int)(S::RING * S::WST);        // LDS address of the current half-chunk's patch image
    auto step = [&](auto tapc) {
        constexpr int tap = decltype(tapc)::value;
        const int s = hc * 9 + tap;
#if defined(ASL_STAMPS_PHASES)
        const unsigned long long tA = __builtin_amdgcn_s_memtime();
#endif
        // ring slot of step s: 9 = 0 (mod 3) and 9 = 1 (mod 4)
        const int slot = RING == 3 ? tap % 3 : ((hc + tap) & 3);
        const int slot_next = RING == 3 ? (tap + DIST) % 3 : ((hc + tap + DIST) & 3);
        if constexpr (tap < PPW) dma_patch_piece(hc + 1, tap);
        dma_weights(s + DIST, slot_next);
        const int o = (tap / 3) * a.Wp + tap % 3;       // tap pixel relative to the top-left one
        const int so = __builtin_amdgcn_readfirstlane((((o >> 4) << 10) | ((o & 15) << 4)) + pboff);
        const char* wb = ring + slot * WST + wrd0;
        bf16x8 xf[FM], wf[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = *(const bf16x8*)(wb + (j >> 1) * 2048 + (j & 1) * 256);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            // two vector instructions per address (the K loop is bound by vector issue, not by the MFMA pipe).
            // The tap term passes through an empty asm per use: otherwise the compiler hoists all 72 tap
            // addresses out of the K loop into registers.
            int so_i = so;
            asm volatile("" : "+s"(so_i));
            const int r = u[i] + so_i;
            const int ad = (r & ~chunk_mask) | chunk_bits;
            xf[i] = *(const __attribute__((address_space(3))) bf16x8*)(unsigned long long)(unsigned)ad;
        }
        // pieces that may stay in flight: those of the last DIST - 1 load phases, this one included
        constexpr int kPrev = tap == 0 ? 8 : tap - 1;
        constexpr int kInFlight = (DIST == 3 ? (WPW + (kPrev < PPW ? 1 : 0)) : 0) + (WPW + (tap < PPW ? 1 : 0));
        static_assert(DIST == 2 || DIST == 3, "wait counts are written for prefetch distance 2 or 3");
#if defined(ASL_STAMPS_PHASES)
        const unsigned long long tB = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kInFlight) : "memory");
#if defined(ASL_STAMPS_PHASES)
        const unsigned long long tC = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
#if defined(ASL_STAMPS_PHASES)
        const unsigned long long tD = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_barrier" ::: "memory");
#if defined(ASL_STAMPS_PHASES)
        const unsigned long long tE = __builtin_amdgcn_s_memtime();
        ph[0] += tB - tA; ph[1] += tC - tB; ph[2] += tD - tC; ph[3] += tE - tD;
#endif
    };
#pragma unroll 1
    for (hc = 0; hc < nq; ++hc) {                       // hc counts (half-chunk, depth tap) pairs
        pboff = lds0 + (int)(S::RING * S::WST) + (hc & 1) * S::PATCH_BYTES;
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{});
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
            int n, r1;
            long long od = 0;
            if constexpr (KD == 3) {
                n = (int)fdiv(m, a.dODHW);
                const int r = m - n * (a.D * HW), d = (int)fdiv(r, a.dOHW);
                r1 = r - d * HW;
                od = (long long)d * a.out_sd;
            } else {
                n = (int)fdiv(m, a.dOHW);
                r1 = m - n * HW;
            }
            const int y = (int)fdiv(r1, a.dOW), x = r1 - y * a.W;
            long long o = (long long)n * a.out_sn + od + (long long)y * a.out_sh + (long long)x * a.out_sw + a.out_off0 + ch0;
            if (a.out_s2d)          // cell layout for a stride-2 consumer: cell ((y+1)>>1, (x+1)>>1), plane by the parities
                o = (long long)n * a.out_sn + (long long)((y + 1) >> 1) * a.out_sh + (long long)((x + 1) >> 1) * a.out_sw +
                    (((y + 1) & 1) * 2 + ((x + 1) & 1)) * a.out_s2d + a.out_off0 + ch0;
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
            // the 3-D instance takes any channel count that is a multiple of 8: a lane whose 8 channels lie past N
            // (weight rows there are zero padding) loads no constants and stores nothing
            const bool ch_ok = KD == 1 || ch0 + h * 32 + 8 <= a.N;
            if (a.scale && ch_ok) { sc0 = *(const f32x4*)(a.scale + ch0 + h * 32); sc1 = *(const f32x4*)(a.scale + ch0 + h * 32 + 4); }
            if (a.bias && ch_ok) { bi0 = *(const f32x4*)(a.bias + ch0 + h * 32); bi1 = *(const f32x4*)(a.bias + ch0 + h * 32 + 4); }
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
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, ov), yrs, (oe[i] == kOobOff || !ch_ok) ? kOobOff : oe[i] + h * 64, 0, 0);
            }
        }
    }
#if defined(ASL_STAMPS)
    st[3] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[4] = __builtin_amdgcn_s_memtime();
#if defined(ASL_STAMPS_PHASES)
    if (a.stamps && tid == 0) {
        unsigned long long* o = a.stamps + (long long)blockIdx.x * 5;
        o[0] = 1; o[1] = 1 + ph[0]; o[2] = o[1] + ph[1]; o[3] = o[2] + ph[2]; o[4] = o[3] + ph[3];
    }
#else
    if (a.stamps && tid == 0)
        for (int k = 0; k < 5; ++k) a.stamps[(long long)blockIdx.x * 5 + k] = st[k];
#endif
#endif
#endif
}

// pixel slots per tile for this geometry (0 = it does not fit the shape): a full tile of consecutive
// pixels when its patch always fits, else whole images
template <int FM_>
static int pp_tile(int n, int h, int w, int in_halo) {
    using S = PpShape<FM_>;
    if (max_patch_pixels(n, h, w, in_halo, S::TP) <= S::PLIVE) return S::TP;
    const int hw = h * w;
    if (hw <= S::TP) {
        const int b = S::TP / hw * hw;
        if (b * 8 >= S::TP * 7 && max_patch_pixels(n, h, w, in_halo, b) <= S::PLIVE) return b;
    }
    return 0;
}

template <int FM_>
static int pp_launch(PatchArgs a, hipStream_t stream) {
    using S = PpShape<FM_>;
    a.tiles_m = (a.M + a.bmv - 1) / a.bmv;
    a.tiles_n = (a.N + S::TN - 1) / S::TN;
    static unsigned char lds_set[64];
    if (const int e = ensure_dynamic_lds((const void*)conv3x3_pp_kernel<FM_, 1>, S::LDS, lds_set)) return e;
    hipLaunchKernelGGL((conv3x3_pp_kernel<FM_, 1>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(512), S::LDS, stream, a);
    return (int)hipGetLastError();
}

// 3x3x3 / stride 1 / pad 1 on halo'd NDHWC through the KD = 3 instance of the 256 x 128 kernel.  `a` carries the 2-D
// fields of one depth plane (H, W, Hp, Wp, in_halo) plus D, Dp, in_halo_d, out_sd; n = clips.  Returns -1 when the
// request does not fit (the caller falls back to the gather-GEMM).
int try_launch_conv3x3x3_pp(PatchArgs a, int n, hipStream_t stream) {
    using S = PpShape<4>;
    if (a.variant != ASL_CONV_AUTO && a.variant != ASL_CONV_PP) return -1;
    if (a.C % 32 || a.C < 64 || a.N % 8 || a.res || a.out_s2d) return -1;
    if (a.y_bytes <= 0 || a.y_bytes >= 0xffffffffLL) return -1;
    if (a.D <= 0 || a.in_halo_d < 1 || a.in_halo < 1 || a.Dp != a.D + 2 * a.in_halo_d) return -1;
    if ((long long)a.Hp * a.Wp * 2 * a.C >= 0x40000000LL) return -1;            // one plane's bytes as a 32-bit DMA offset (x3)
    // tiles of consecutive output pixels inside one clip: its depth planes lie like the images of a 2-D batch
    a.bmv = pp_tile<4>(a.D, a.H, a.W, a.in_halo);
    if (!a.bmv) return -1;
    const long long dhw = (long long)a.D * a.H * a.W;
    a.tiles_per_clip = (int)((dhw + a.bmv - 1) / a.bmv);
    a.dTPC = make_fastdiv((unsigned)a.tiles_per_clip);
    const long long tiles_m = (long long)n * a.tiles_per_clip;
    a.tiles_n = (a.N + S::TN - 1) / S::TN;
    if (tiles_m * a.tiles_n > 0x7fffffffLL) return -1;
    a.tiles_m = (int)tiles_m;
    static unsigned char lds_set[64];
    if (const int e = ensure_dynamic_lds((const void*)conv3x3_pp_kernel<4, 3>, S::LDS, lds_set)) return e;
    hipLaunchKernelGGL((conv3x3_pp_kernel<4, 3>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(512), S::LDS, stream, a);
    return (int)hipGetLastError();
}

// Runs the request on the large-tile kernel when it fits: C a multiple of 32 and at least 128, N a multiple
// of 128.  Returns -1 when it does not apply (the caller falls back).
int try_launch_conv3x3_pp(PatchArgs a, int n, hipStream_t stream) {
    if (a.variant != ASL_CONV_AUTO && a.variant != ASL_CONV_PP && a.variant != ASL_CONV_PP_BIG) return -1;
    if (a.C % 32 || a.C < 128 || a.N % 128) return -1;
    if (a.y_bytes <= 0 || a.y_bytes >= 0xffffffffLL || (a.res && (a.res_bytes <= 0 || a.res_bytes >= 0xffffffffLL))) return -1;
    if (a.W <= 0 || a.H <= 0 || (int)a.dOW.d != a.W || (int)a.dOHW.d != a.H * a.W || a.Wp != a.W + 2 * a.in_halo ||
        a.Hp != a.H + 2 * a.in_halo)
        return -1;
    if (a.variant != ASL_CONV_PP_BIG) {
        a.bmv = pp_tile<4>(n, a.H, a.W, a.in_halo);
        if (a.bmv) return pp_launch<4>(a, stream);
    }
    if (a.variant != ASL_CONV_PP) {
        a.bmv = pp_tile<8>(n, a.H, a.W, a.in_halo);
        if (a.bmv) return pp_launch<8>(a, stream);
    }
    return -1;
}

}  // namespace asl_detail
