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
    static constexpr int EW = WST / 1024 / 4;            // weight DMA pieces per EARLY wave per stage (waves 0-3 stream the weights)
    static constexpr int PT = ((PIECES + 3) / 4 + 6) / 7; // patch pieces per LATE wave per tap (waves 4-7 stream the patch over taps 0-6)
    static constexpr int RING = FM == 8 ? 4 : 3;         // weight ring depth; prefetch distance RING - 1
    static constexpr int LDS = 2 * PATCH_BYTES + RING * WST;
    static constexpr int WAVES_PER_EU = FM == 8 ? 2 : 4;
    static_assert(PPW >= 1 && EW >= 1 && PIECES % 4 == 0 && 4 * 9 * EW >= PIECES, "piece counts");
};

template <int FM_>
__global__ __launch_bounds__(512, PpShape<FM_>::WAVES_PER_EU) void conv3x3_pp_kernel(const PatchArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    using S = PpShape<FM_>;
    constexpr int WCO = S::WCO, FM = S::FM, FN = 4;
    constexpr int PPW = S::PPW, EW = S::EW, PT = S::PT, WST = S::WST, RING = S::RING, DIST = RING - 1;

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
    const int nsteps = nhc * 9;

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

    // patch piece j covers patch pixels 16j..16j+15 (64 B each): lane -> pixel 16j + lane/4, slot lane%4,
    // which holds 16-byte chunk (lane%4) ^ swz(p), swz(p) = (p & 4) >> 1 (conflict-free fragment reads
    // from any 16 consecutive pixels)
    const unsigned pvoff = (unsigned)((lane >> 2) * pix_bytes + 16 * ((lane & 3) ^ ((lane >> 3) & 2)));
    auto dma_patch_piece = [&](int hc, int j) {         // prologue only (half-chunk 0); pieces past the patch are skipped
        if (16 * j < NP) {
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(16LL * j * pix_bytes + hc * 64));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(patches + (hc & 1) * S::PATCH_BYTES + j * 1024),
                                                     16, pvoff, so, 0, 0);
        }
    };
    // weight piece q = (wave & 3) * EW + i covers tile rows 16q..16q+15 (64 B each); chunk swizzle (row & 8) ? 3 : 0
    unsigned wvoff[EW];
#pragma unroll
    for (int i = 0; i < EW; ++i) {
        const int row = ((wave & 3) * EW + i) * 16 + (lane >> 2);
        wvoff[i] = (unsigned)(2 * row * a.wk + 16 * ((lane & 3) ^ ((row & 8) ? 3 : 0)));
    }
    auto dma_weights = [&](int s, int slot) {           // prologue only: stage s -> ring slot
        const int hc = s / 9, tap = s - 9 * hc;
        char* dst = ring + slot * WST + (wave & 3) * (EW * 1024);
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(2 * (tap * a.C + hc * 32)));
#pragma unroll
        for (int i = 0; i < EW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                                     s < nsteps ? wvoff[i] : kOob, so, 0, 0);
    };
    // K-loop roles: waves 0-3 stream weight pieces, waves 4-7 patch pieces, through ONE instruction sequence whose
    // descriptor and per-lane offsets are picked once per wave (scalar selects), so the roles cost no extra code.
    const int w4 = wave & 3;
    __amdgpu_buffer_rsrc_t krs = late ? xrs : wrs;
    unsigned kvoff[EW];
#pragma unroll
    for (int i = 0; i < EW; ++i) kvoff[i] = late ? pvoff : wvoff[i];
    const unsigned w4_patch_so = (unsigned)(16LL * w4 * pix_bytes);         // piece w4 of the patch
    const unsigned quad_patch_so = (unsigned)(64LL * pix_bytes);            // four pieces further on

    // ---- per-lane fragment geometry ----------------------------------------------------------
    const int fr = lane & 15, fq = lane >> 4;
    int pc[FM];                                         // byte offset of (centre pixel, chunk fq) in an unswizzled patch
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        int m = m0 + wp * (FM * 16) + i * 16 + fr;
        m = m < mend ? m : mend - 1;                    // idle slots recompute the tile's last pixel; never stored
        pc[i] = (int)(centre(m) - P0) * 64 + fq * 16;
    }
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

    // ---- prologue: patch of half-chunk 0, weight stages 0..2 ------------------------------------
    // Roles in the K loop: the early half (waves 0-3) streams the weight stages, the late half (waves 4-7) the
    // next half-chunk's patch.  vmcnt retires a wave's loads in issue order, so a wave that issued both would
    // sit behind its slow patch pieces (HBM) every time it waits for the next weight stage (L2); split this way
    // the per-step wait only ever covers weight pieces, and the patch is waited for once per half-chunk.
#pragma unroll
    for (int k = 0; k < PPW; ++k) dma_patch_piece(0, wave + 8 * k);
    if (!late) {
#pragma unroll
        for (int d = 0; d < DIST; ++d) dma_weights(d, d);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * EW) : "memory");          // patch 0 pieces and stage 0 landed
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    if (late) asm volatile("s_barrier" ::: "memory");                                 // run one phase behind
#if defined(ASL_STAMPS)
    st[1] = __builtin_amdgcn_s_memtime();
#endif

    // ---- K loop ---------------------------------------------------------------------------------
    // Step s: load phase L(s) = {this wave's DMA pieces (weight stage s + DIST, or a slice of the next half-chunk's
    // patch), the fragment reads of step s, counted wait} | barrier | compute phase = the MFMAs | barrier.
#if defined(ASL_STAMPS_PHASES)
    unsigned long long ph[4] = {0, 0, 0, 0};   // diagnostic: issue (DMA + reads), wait + barrier, MFMA issue, second barrier
#endif
    int hc = 0;
    const char* pbuf = patches;
    auto step = [&](auto tapc) {
        constexpr int tap = decltype(tapc)::value;
        const int s = hc * 9 + tap;
#if defined(ASL_STAMPS_PHASES)
        const unsigned long long tA = __builtin_amdgcn_s_memtime();
#endif
        // ring slot of step s: 9 = 0 (mod 3) and 9 = 1 (mod 4)
        const int slot = RING == 3 ? tap % 3 : ((hc + tap) & 3);
        const int slot_next = RING == 3 ? (tap + DIST) % 3 : ((hc + tap + DIST) & 3);
        {
            // early: stage s + DIST into ring slot slot_next, piece w4 * EW + i.  Past the last stage the pieces
            // are still issued (the counted waits assume a fixed number): they read the first half-chunk of a
            // later tap of the same weight rows, in bounds, into ring slots nobody reads any more.
            // late: piece w4 + 4 (tap * EW + i) of half-chunk hc + 1's patch into the other patch buffer.  Pieces
            // past the patch's last pixel, or of the half-chunk after the last, fetch neighbouring bytes of x
            // (zeros beyond its end: the descriptor's range check) into space nobody reads.
            const int tapn = (tap + DIST) % 9, hcn = hc + (tap + DIST) / 9;
            const unsigned so_w = (unsigned)(2 * (tapn * a.C + hcn * 32));
            char* const dst_w = ring + slot_next * WST + w4 * (EW * 1024);
            char* const dst_p = patches + ((hc + 1) & 1) * S::PATCH_BYTES + w4 * 1024;
#pragma unroll
            for (int i = 0; i < EW; ++i) {
                constexpr int quad = tap * EW;          // + i: which group of four patch pieces
                if (4 * (quad + i) < S::PIECES) {
                    const unsigned so_p = w4_patch_so + (quad + i) * quad_patch_so + (hc + 1) * 64;
                    const unsigned so = __builtin_amdgcn_readfirstlane(late ? so_p : so_w);
                    char* const dst = late ? dst_p + (quad + i) * 4096 : dst_w + i * 1024;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)dst, 16, kvoff[i], so, 0, 0);
                } else if (!late) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)(dst_w + i * 1024), 16,
                                                             kvoff[i], __builtin_amdgcn_readfirstlane(so_w), 0, 0);
                }
            }
        }
        const int toff = ((tap / 3 - 1) * a.Wp + (tap % 3 - 1)) * 64;
        const char* wb = ring + slot * WST;
        bf16x8 xf[FM], wf[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = *(const bf16x8*)(wb + wrd[j]);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            // byte offset t = 64 p + 16 fq; the chunk swizzle (p & 4) >> 1 is bit 8 of t moved to bit 5.
            // The empty asm keeps the compiler from hoisting 72 tap addresses out of the K loop into VGPRs.
            int t = pc[i];
            asm volatile("" : "+v"(t));
            t += toff;
            xf[i] = *(const bf16x8*)(pbuf + (t ^ ((t >> 3) & 0x20)));
        }
        // early waves: the weight pieces of the last DIST - 1 load phases (this one included) may stay in flight;
        // late waves: their patch pieces must have landed before the half-chunk's last barrier, nothing before that
        constexpr int kInFlight = (DIST - 1) * EW;
#if defined(ASL_STAMPS_PHASES)
        const unsigned long long tB = __builtin_amdgcn_s_memtime();
#endif
        // (late waves have at most PT pieces per tap in flight ahead of this count, so outside tap 8 the same
        // wait only ever holds them for a piece issued (DIST - 1) * EW / PT taps earlier)
        if (tap == 8 && late) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kInFlight) : "memory");
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
    for (hc = 0; hc < nhc; ++hc) {
        pbuf = patches + (hc & 1) * S::PATCH_BYTES;
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
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_pp_kernel<FM_>, hipFuncAttributeMaxDynamicSharedMemorySize, S::LDS);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv3x3_pp_kernel<FM_>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(512), S::LDS, stream, a);
    return (int)hipGetLastError();
}

// Runs the request on the large-tile kernel when it fits: C a multiple of 32 and at least 128, N a multiple
// of 128.  Returns -1 when it does not apply (the caller falls back).
int try_launch_conv3x3_pp(PatchArgs a, int n, hipStream_t stream) {
    if (a.C % 32 || a.C < 128 || a.N % 128 || getenv("ASL_NO_PP_CONV")) return -1;
    if (a.y_bytes <= 0 || a.y_bytes >= 0xffffffffLL || (a.res && (a.res_bytes <= 0 || a.res_bytes >= 0xffffffffLL))) return -1;
    if (a.W <= 0 || a.H <= 0 || (int)a.dOW.d != a.W || (int)a.dOHW.d != a.H * a.W || a.Wp != a.W + 2 * a.in_halo ||
        a.Hp != a.H + 2 * a.in_halo)
        return -1;
    if (!getenv("ASL_PP_BIG")) {
        a.bmv = pp_tile<4>(n, a.H, a.W, a.in_halo);
        if (a.bmv) return pp_launch<4>(a, stream);
    }
    a.bmv = pp_tile<8>(n, a.H, a.W, a.in_halo);
    if (a.bmv) return pp_launch<8>(a, stream);
    return -1;
}

}  // namespace asl_detail
