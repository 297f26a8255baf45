// A whole 64-channel ResNet BasicBlock in one kernel, producer / consumer form (gfx950):
//     y = relu( bn2(conv2( relu(bn1(conv1(x))) )) + x ),   conv1, conv2: 3x3 / stride 1 / pad 1, 64 -> 64.
//
// The one-wave-per-SIMD kernel (block_c64.hip) keeps both filter banks in every wave's registers, and with a single
// wave on a SIMD nothing hides its vector instructions.  Here a workgroup has EIGHT waves, two per SIMD, and each
// holds ONE bank (144 registers) as the A operands of v_mfma_f32_32x32x16_bf16:
//
//   * waves 0-3 (producers) run conv1 + bn1 + ReLU of step g and write the intermediate t into LDS;
//   * waves 4-7 (consumers) run conv2 + bn2 + residual + ReLU of step g - 1 from LDS and store y;
//   * one barrier per step.  On every SIMD one producer and one consumer share the matrix pipe: while one of them
//     runs a tile's epilogue or address arithmetic, the other's MFMAs keep the pipe busy.
//
// A workgroup walks "units" = (image, column strip of CW <= 28 output columns) from top to bottom in steps of R
// output rows.  Step k of a unit:
//   consumer: y rows kR .. kR+R-1 from t rows kR-1 .. kR+R (t image: R+2 rows x CW+2 columns, local row 0 = kR-1);
//   producer: t rows kR+1 .. kR+R (and row 0 when k = 0) from x rows kR .. kR+R+1 (x image: R+2 rows x CW+4
//   columns); t rows kR-1, kR were made by step k-1 and are copied over from the other t buffer (LDS to LDS).
// So every t value is computed once per column strip (the strips overlap by two t columns: 30/28 of the MFMAs),
// and x is read from HBM once as conv1's input (rows re-fetched across steps and the residual come from L2).
// x for step g+1 streams into the other x buffer by LDS-DMA, issued by the producers, whose memory counter
// carries nothing else.  The x row above the image (k = 0) is not stored: its taps read a pixel of zeros instead.
// LDS: 2 x images + 2 t images + that zero pixel group + the BatchNorm constants; 56x56: R = 8, 159 KiB.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asl_hip.h"
#include "block_mfma.h"
#include "common.h"
#include "patch_args.h"

#ifndef ASL_PAIR_EXP
#define ASL_PAIR_EXP 0
#endif

namespace asl_detail {

struct PairArgs {
    const char* x;
    const char* w1;
    const char* w2;
    const float* s1;
    const float* b1;
    const float* s2;
    const float* b2;
    char* y;
    int n, H, W, Wp, Hp;
    int in_halo, out_halo;
    int R, K;                      // output rows per step, steps per unit
    int CW, ncs;                   // output columns per strip, strips per image
    int XW, TW;                    // row pitch of the x / t images, pixels (CW + 4, CW + 2)
    int xpix, tpix;                // pixels per x / t image, multiples of 16
    int nunits;
    int wk;
    long long x_bytes;
    long long out_sn;
    int out_sh;
    int out_s2d;                   // 1: y is the cell tensor of the halo-1 output (out_sn / out_sh are per image / cell row)
    FastDiv dTW, dCW, dXW, dK, dncs;
};

template <int S>
struct PairSteps {
#ifndef ASL_PAIR_AHEAD
#define ASL_PAIR_AHEAD 5
#endif
    static constexpr int kAhead = ASL_PAIR_AHEAD, kRing = kAhead + 1;
    // At two waves per SIMD a wave has 128 accumulation registers and 128 ordinary ones: 32 of the bank's 36
    // fragments are addressed in the former, the last four in the latter.
    static constexpr int kBankA = 32;
    // accumulation chains the 36 steps alternate over (1 = diagnostic build: one dependent chain, same results;
    // block_mfma.h says what a one-chain loop must not do)
#ifndef ASL_PAIR_CHAINS
#define ASL_PAIR_CHAINS 2
#endif
    static constexpr int kChains = ASL_PAIR_CHAINS;
    static_assert(kChains == 1 || kChains == 2, "one or two accumulation chains");
    __device__ static __forceinline__ void read(bf16x8 (&xf)[kRing], const unsigned (&ad)[9]) {
#if ASL_PAIR_EXP == 1      // diagnostic build: K loop without its LDS reads (results wrong)
        asm volatile("" : "=v"(xf[S % kRing]) : "v"(ad[S >> 2]));
#else
        xf[S % kRing] = lds_frag<(S & 3) * 512>(ad[S >> 2]);
#endif
    }
    // the read issued right behind the MFMA of step S - kAhead: never into that MFMA's own fragment register
    __device__ static __forceinline__ void read_after(bf16x8 (&xf)[kRing], const unsigned (&ad)[9]) {
#if ASL_PAIR_EXP == 1
        asm volatile("" : "=v"(xf[S % kRing]) : "v"(ad[S >> 2]));
#else
        xf[S % kRing] = lds_frag_after<(S & 3) * 512>(ad[S >> 2], xf[(S - kAhead) % kRing]);
#endif
    }
    __device__ static __forceinline__ void run(f32x16 (&acc)[2], const bf16x8 (&w)[36], bf16x8 (&xf)[kRing], const unsigned (&ad)[9]) {
        constexpr int inflight = 35 - S < kAhead - 1 ? 35 - S : kAhead - 1;      // reads younger than step S's
        if constexpr (S == 35 && kChains == 1) mfma_last1<(S < kBankA), inflight>(acc[0], w[S], xf[S % kRing]);
        else if constexpr (S == 35) mfma_last2<(S < kBankA), inflight, S % 2>(acc, w[S], xf[S % kRing]);
        else mfma_step<(S < kBankA), inflight, S < kChains ? 1 : 0>(acc[S % kChains], w[S], xf[S % kRing]);
        if constexpr (S + kAhead < 36) PairSteps<S + kAhead>::read_after(xf, ad);
        if constexpr (S + 1 < 36) PairSteps<S + 1>::run(acc, w, xf, ad);
    }
};

// 36 K steps of one 32-pixel x 32-channel tile; ad[tap] = LDS address of the lane's pixel for that tap (+ k half)
__device__ __forceinline__ f32x16 pair_tile(const bf16x8 (&w)[36], const unsigned (&ad)[9]) {
    bf16x8 xf[PairSteps<0>::kRing];
    PairSteps<0>::read(xf, ad);
    PairSteps<1>::read(xf, ad);
    PairSteps<2>::read(xf, ad);
    if constexpr (PairSteps<0>::kAhead > 3) PairSteps<3>::read(xf, ad);
    if constexpr (PairSteps<0>::kAhead > 4) PairSteps<4>::read(xf, ad);
    if constexpr (PairSteps<0>::kAhead > 5) PairSteps<5>::read(xf, ad);
    static_assert(PairSteps<0>::kAhead >= 3 && PairSteps<0>::kAhead <= 6, "prefetch distance");
    f32x16 acc[2];
    PairSteps<0>::run(acc, w, xf, ad);
    if constexpr (PairSteps<0>::kChains == 1) return acc[0];
    else return acc[0] + acc[1];
}

// BatchNorm, residual (conv2), ReLU on the rounded pair, masked: the lane's 16 channels as 8 packed dwords
template <bool kConv2>
__device__ __forceinline__ void pair_finish(const f32x16& acc, const float* bnl, const bf16x8 (&res)[2], bool keep, unsigned (&out)[8]) {
    // scale / bias come from LDS here, after the K loop, into registers the fragment ring has just vacated
    f32x4 sc[4], bi[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#if ASL_PAIR_EXP == 2      // diagnostic build: no BatchNorm reads (results wrong)
        sc[g] = f32x4{1.f, 1.f, 1.f, 1.f};
        bi[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
        sc[g] = *(const f32x4*)(bnl + 4 * g);
        bi[g] = *(const f32x4*)(bnl + 64 + 4 * g);
#endif
    }
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = j >> 1, e = 2 * (j & 1);
        float v0 = acc[2 * j] * sc[g][e] + bi[g][e], v1 = acc[2 * j + 1] * sc[g][e + 1] + bi[g][e + 1];
        if constexpr (kConv2) {
            v0 += bf16_to_f32(res[g >> 1][4 * (g & 1) + e]);
            v1 += bf16_to_f32(res[g >> 1][4 * (g & 1) + e + 1]);
        }
        unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, bf16x2_t));
        // as 16-bit integers, negative bf16 values are negative: ReLU of the rounded pair in one instruction
        asm("v_pk_max_i16 %0, %1, 0" : "=v"(pk) : "v"(pk));
        out[j] = keep ? pk : 0u;
    }
}

__global__ __launch_bounds__(512, 1) void block_c64_pair_kernel(const PairArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2;                            // 0: producer (conv1), 1: consumer (conv2)
    const int wr = wave & 3, wc = wr & 1, wq = wr >> 1;    // channel half, pixel-tile parity
    const int lp = lane & 31, lh = lane >> 5;              // MFMA column (pixel), k half
    const int Wp = a.Wp, R = a.R, XW = a.XW, TW = a.TW, CW = a.CW;
    const unsigned xbytes = (unsigned)a.xpix * 128u, tbytes = (unsigned)a.tpix * 128u;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const unsigned zoff = 2 * xbytes + 2 * tbytes;         // the zero pixel group (2 KiB), then [s1 | b1 | s2 | b2]
    float* const bn = (float*)(smem + zoff + 2048);

    // ---- this wave's filter bank.  MFMA row r = 8i + 4h + e is output channel 32*wc + 16h + 4i + e, so accumulator
    // register 4i + e of lane half h is channel 32*wc + 16h + 4i + e: 16 consecutive channels per lane.
    bf16x8 w[36];
    {
        const int ch = 32 * wc + 16 * ((lp >> 2) & 1) + 4 * (lp >> 3) + (lp & 3);
        const short* r = (const short*)(role ? a.w2 : a.w1) + (long long)ch * a.wk + lh * 8;
#pragma unroll
        for (int s = 0; s < 36; ++s) w[s] = *(const bf16x8*)(r + s * 16);   // step s = (tap, 16-channel group)
    }
    if (tid < 128) *(i32x4*)(smem + zoff + tid * 16) = i32x4{0, 0, 0, 0};
    if (tid < 256) bn[tid] = tid < 64 ? a.s1[tid] : tid < 128 ? a.b1[tid - 64] : tid < 192 ? a.s2[tid - 128] : a.b2[tid - 192];

    const int my_units = (a.nunits - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int G = my_units * a.K;                          // steps of this workgroup
    // step g -> image, first column of its strip, step within the unit
    auto decode = [&](int g, int& img, int& c0, int& k) {
        const int ui = (int)fdiv((unsigned)g, a.dK);
        k = g - ui * a.K;
        const int unit = (int)blockIdx.x + ui * (int)gridDim.x;
        img = (int)fdiv((unsigned)unit, a.dncs);
        c0 = (unit - img * a.ncs) * CW;
    };

    const int xpieces = a.xpix >> 3;                       // 1 KiB pieces: 16 pixels x 4 chunks
    // Piece j: pixel group j >> 1, chunks 4*(j & 1) .. +3; lane -> pixel 16*(j >> 1) + lane % 16, chunk 4*(j & 1) + lane / 16.
    // Source of step g's x image: a buffer descriptor based at its image (wave-uniform, scalar work only), so lane
    // offsets are 32-bit whatever the batch size, and rows past the end of the tensor read as zeros by the
    // descriptor's own range check.  p0 = first pixel of the x image within the halo'd image.
    struct StepSrc {
        __amdgpu_buffer_rsrc_t rs;
        int p0;
    };
    auto dma_source = [&](int g) -> StepSrc {
        int img, c0, k;
        decode(g, img, c0, k);
        const long long base = (long long)img * a.Hp * Wp * 128;
        const long long rem = a.x_bytes - base;
        StepSrc s;
        s.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + base), 0, (unsigned)(rem > 0x7fffffffLL ? 0x7fffffffLL : rem), 0x00020000);
        s.p0 = (k * R + a.in_halo) * Wp + (c0 - 2 + a.in_halo);      // >= Wp - 1: never before the image
        return s;
    };
    auto dma_piece = [&](int j, const StepSrc& src, char* dst) {
        const int px = 16 * (j >> 1) + (lane & 15);
        const int i = (int)fdiv((unsigned)px, a.dXW), jc = px - i * XW;
        const unsigned vo = (unsigned)(src.p0 + i * Wp + jc) * 128u + (4 * (j & 1) + (lane >> 4)) * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(src.rs, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, vo, 0, 0, 0);
    };

#if defined(ASL_PAIR_STAMPS)   // diagnostic build: s_memtime stamps of workgroup 0's waves 0 and 4 into y's first halo row
    unsigned long long* const stamp = (unsigned long long*)a.y + 400 * role;
    int stamp_i = 0;
    if (blockIdx.x == 0 && lane == 0) ((unsigned long long*)a.y)[800 + wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
#define STAMP() do { if (blockIdx.x == 0 && wr == 0 && lane == 0 && stamp_i < 400) stamp[stamp_i] = __builtin_amdgcn_s_memtime(); ++stamp_i; } while (0)
#else
#define STAMP() do { } while (0)
#endif
    const int pitch = role ? TW : XW;                      // the image this wave's taps walk over
    unsigned tapenc[9];                                    // wave-uniform
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int o = (tap / 3) * pitch + tap % 3;
        tapenc[tap] = (unsigned)(((o >> 4) << 11) | ((o & 15) << 4));
    }
    const unsigned khalf = (unsigned)lh << 8;
    const int chl = 32 * wc + 16 * lh;                     // the lane's 16 output channels

    if (role == 0) {
        const StepSrc src = dma_source(0);
        for (int j = wr; j < xpieces; j += 4) dma_piece(j, src, smem);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // The bank has arrived: said here, by a use of every fragment, or the compiler retires the 36 loads lazily
    // inside the tile loops with counted waits that would then also wait for every later DMA piece, residual load
    // and store.
#pragma unroll
    for (int s = 0; s < 36; ++s) {
        if (s < PairSteps<0>::kBankA) asm volatile("" : "+a"(w[s]));
        else asm volatile("" : "+v"(w[s]));
    }
    const float* const bnl = bn + 128 * role + chl;        // the lane's scale values, bias 64 floats on

    for (int g = 0; g <= G; ++g) {
        if (role == 0) {
            STAMP();
            if (g < G) {
                // the next step's x image: this wave's pieces go out a few ahead of each tile, not in one burst
                const bool more = g + 1 < G;
                const StepSrc nsrc = dma_source(more ? g + 1 : g);
                char* const ndst = smem + ((g + 1) & 1) * xbytes;
#if ASL_PAIR_EXP == 3      // diagnostic build: no x image DMA after the first (results wrong)
                int jn = xpieces;
#else
                int jn = more ? wr : xpieces;
#endif
                STAMP();
                int img, c0, k;
                decode(g, img, c0, k);
                char* const tb = smem + 2 * xbytes + (g & 1) * tbytes;
                const char* const tprev = smem + 2 * xbytes + ((g & 1) ^ 1) * tbytes;
                // local t rows 0, 1: above the image when k = 0 (row 0 = zeros; row 1 is computed below), otherwise
                // the previous step's last two rows
#if ASL_PAIR_EXP == 4      // diagnostic build: no carry copy (results wrong)
                if (false) {
                } else if (k == 0) {
#else
                if (k == 0) {
#endif
                    for (int i = tid; i < TW * 8; i += 256) *(i32x4*)(tb + img_addr(i >> 3) + (i & 7) * 256) = i32x4{0, 0, 0, 0};
                } else if (ASL_PAIR_EXP != 4) {
                    for (int i = tid; i < 2 * TW * 8; i += 256)
                        *(i32x4*)(tb + img_addr(i >> 3) + (i & 7) * 256) = *(const i32x4*)(tprev + img_addr(R * TW + (i >> 3)) + (i & 7) * 256);
                }
                STAMP();
                const int r_first = k == 0 ? 1 : 2;
                // Pixel slots run over rows of XW, the x image's own pitch (the last XW - TW of a row are idle): the 16
                // lanes of an LDS pass then read 16 consecutive image pixels, which the chunk-major image serves without
                // a bank conflict; rows of TW slots would put a two-pixel jump inside most passes.
                const int n1 = (R + 2 - r_first) * XW;
                const int t1 = ASL_PAIR_EXP == 5 ? 0 : (n1 + 31) >> 5;    // 5: diagnostic build, producers idle
                const int my_tiles = (t1 - wq + 1) >> 1 > 0 ? (t1 - wq + 1) >> 1 : 1;
                const int per_tile = ((xpieces + 3) / 4 + my_tiles - 1) / my_tiles;
                const unsigned xl = lds0 + (g & 1) * xbytes;
                const unsigned zad = (lds0 + zoff) | khalf;
                const bf16x8 nores[2] = {};
                for (int tile = wq; tile < t1; tile += 2) {
                    for (int q = 0; q < per_tile && jn < xpieces; ++q, jn += 4) dma_piece(jn, nsrc, ndst);
                    int m = tile * 32 + lp;
                    m = m < n1 ? m : n1 - 1;
                    const int rt = (int)fdiv((unsigned)m, a.dXW);
                    int ct = m - rt * XW;
                    const bool slot_ok = tile * 32 + lp < n1 && ct < TW;
                    ct = ct < TW ? ct : TW - 1;            // idle slots repeat the row's last pixel
                    const int lr = r_first + rt;           // local t row: image row kR - 1 + lr
                    // t (lr, ct) reads x local rows lr-2 .. lr, columns ct .. ct+2
                    const int umid = (lr - 1) * XW + ct;
                    const bool top = lr >= 2;
                    const unsigned emid = xl + (unsigned)img_addr(umid) + 0x700u;
                    const unsigned etop = xl + (unsigned)img_addr(top ? umid - XW : 0) + 0x700u;
                    unsigned ad[9];
#if ASL_PAIR_EXP == 8      // diagnostic build: no per-tile address arithmetic (results wrong)
#pragma unroll
                    for (int i = 0; i < 9; ++i) ad[i] = (xl + tapenc[i] + lp * 16) | khalf;
#else
#pragma unroll
                    for (int i = 0; i < 3; ++i) ad[i] = top ? (((etop + tapenc[i]) & ~0x700u) | khalf) : zad;
#pragma unroll
                    for (int i = 0; i < 6; ++i) ad[3 + i] = ((emid + tapenc[i]) & ~0x700u) | khalf;
#endif
                    const f32x16 acc = pair_tile(w, ad);
                    const int yt = k * R - 1 + lr, xt = c0 - 1 + ct;
                    const bool inside = yt >= 0 && yt < a.H && xt >= 0 && xt < a.W;   // outside: conv2's zero padding
                    unsigned out[8];
#if ASL_PAIR_EXP == 7      // diagnostic build: no tile epilogue (results wrong)
                    asm volatile("" :: "v"(acc));
                    continue;
#endif
                    pair_finish<false>(acc, bnl, nores, inside, out);
                    char* const dst = tb + img_addr(lr * TW + ct) + (chl >> 3) * 256;
                    if (slot_ok) {
                        *(i32x4*)dst = i32x4{(int)out[0], (int)out[1], (int)out[2], (int)out[3]};
                        *(i32x4*)(dst + 256) = i32x4{(int)out[4], (int)out[5], (int)out[6], (int)out[7]};
                    }
                    STAMP();
                }
                for (; jn < xpieces; jn += 4) dma_piece(jn, nsrc, ndst);
            }
            STAMP();
            // the next x image has landed, this step's t image is written
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            STAMP();
            if (g >= 1) {
                const int gc = g - 1;
                int img, c0, k;
                decode(gc, img, c0, k);
                const unsigned tl = lds0 + 2 * xbytes + (gc & 1) * tbytes;
#if defined(ASL_PAIR_WIDE_SLOTS)
                const int n2 = R * TW;                     // slot rows at the t image's pitch: conflict-free reads, (TW - CW) idle slots a row
#else
                const int n2 = R * CW;
#endif
                const int t2 = ASL_PAIR_EXP == 6 ? 0 : (n2 + 31) >> 5;    // 6: diagnostic build, consumers idle
                for (int tile = wq; tile < t2; tile += 2) {
                    int m = tile * 32 + lp;
#if defined(ASL_PAIR_WIDE_SLOTS)
                    m = m < n2 ? m : n2 - 1;
                    const int ro = (int)fdiv((unsigned)m, a.dTW);
                    int co = m - ro * TW;
                    const bool slot_ok = tile * 32 + lp < n2 && co < CW;
                    co = co < CW ? co : CW - 1;
#else
                    const bool slot_ok = m < n2;
                    m = slot_ok ? m : n2 - 1;
                    const int ro = (int)fdiv((unsigned)m, a.dCW), co = m - ro * CW;
#endif
                    const int yo = k * R + ro, xo = c0 + co;
                    const bool ok = slot_ok && yo < a.H && xo < a.W;
                    const char* rp = a.x + 2 * ((((long long)img * a.Hp + (ok ? yo : 0) + a.in_halo) * Wp + (ok ? xo : 0) + a.in_halo) * 64 + chl);
                    bf16x8 res[2];
                    res[0] = *(const bf16x8*)rp;
                    res[1] = *(const bf16x8*)(rp + 16);
                    const unsigned e0 = tl + (unsigned)img_addr(ro * TW + co) + 0x700u;
                    unsigned ad[9];
#pragma unroll
#if ASL_PAIR_EXP == 8
                    for (int i = 0; i < 9; ++i) ad[i] = (tl + tapenc[i] + lp * 16) | khalf;
#else
                    for (int i = 0; i < 9; ++i) ad[i] = ((e0 + tapenc[i]) & ~0x700u) | khalf;
#endif
                    const f32x16 acc = pair_tile(w, ad);
                    unsigned out[8];
#if ASL_PAIR_EXP == 7
                    asm volatile("" :: "v"(acc), "v"(res[0]), "v"(res[1]));
                    continue;
#endif
                    pair_finish<true>(acc, bnl, res, true, out);
                    const int yh = (ok ? yo : 0) + a.out_halo, xh = (ok ? xo : 0) + a.out_halo;      // halo'd output pixel
                    char* const dst = a.out_s2d
                        ? a.y + 2 * ((long long)img * a.out_sn + (long long)(yh >> 1) * a.out_sh + (long long)(xh >> 1) * 256 +
                                     ((yh & 1) * 2 + (xh & 1)) * 64 + chl)
                        : a.y + 2 * ((long long)img * a.out_sn + (long long)yh * a.out_sh + (long long)xh * 64 + chl);
                    if (ok) {
                        *(i32x4*)dst = i32x4{(int)out[0], (int)out[1], (int)out[2], (int)out[3]};
                        *(i32x4*)(dst + 16) = i32x4{(int)out[4], (int)out[5], (int)out[6], (int)out[7]};
                    }
                    STAMP();
                }
            }
            STAMP();
            // this step's reads of the t image are over (the stores to y need no wait)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
#endif
}

// Launches the producer / consumer kernel, or returns -1 when the shape does not fit it.
int try_launch_block_c64_pair(const asl_block_params* p, const void* x, const void* w1, const float* scale1, const float* bias1,
                              const void* w2, const float* scale2, const float* bias2, void* y, hipStream_t stream) {
    PairArgs a;
    a.x = (const char*)x; a.w1 = (const char*)w1; a.w2 = (const char*)w2; a.y = (char*)y;
    a.s1 = scale1; a.b1 = bias1; a.s2 = scale2; a.b2 = bias2;
    a.n = p->n; a.H = p->h; a.W = p->w; a.in_halo = p->in_halo; a.out_halo = p->out_halo;
    a.Wp = p->w + 2 * p->in_halo; a.Hp = p->h + 2 * p->in_halo;
    a.wk = 9 * 64;
    a.x_bytes = 2LL * p->n * a.Hp * a.Wp * 64;
    if ((long long)a.Hp * a.Wp >= (1LL << 23)) return -1;  // 32-bit byte offsets within one image
    const int owp = p->w + 2 * p->out_halo, ohp = p->h + 2 * p->out_halo;
    a.out_sn = (long long)ohp * owp * 64; a.out_sh = owp * 64;
    a.out_s2d = p->out_s2d ? 1 : 0;
    if (a.out_s2d) { a.out_sn = (long long)(ohp / 2) * (owp / 2) * 256; a.out_sh = (owp / 2) * 256; }
    // the fewest column strips whose images leave room for steps of at least min(8, h) rows
    constexpr int kLds = 160 * 1024, kFixed = 2048 + 1024;
    const int want = p->h < 8 ? p->h : 8;
    int R = 0;
    for (int ncs = 1; ncs <= p->w && !R; ++ncs) {
        const int cw = (p->w + ncs - 1) / ncs;
        for (int r = p->h < 16 ? p->h : 16; r >= want; --r) {
            const int xp = ((r + 2) * (cw + 4) + 15) & ~15, tp = ((r + 2) * (cw + 2) + 15) & ~15;
            if (2LL * (xp + tp) * 128 + kFixed <= kLds) {
                R = r; a.CW = cw; a.ncs = ncs; a.xpix = xp; a.tpix = tp;
                break;
            }
        }
    }
    if (!R) return -1;
    a.R = R; a.K = (p->h + R - 1) / R;
    a.XW = a.CW + 4; a.TW = a.CW + 2;
    if ((long long)p->n * a.ncs > 0x3fffffffLL / a.K) return -1;
    a.nunits = p->n * a.ncs;
    a.dTW = make_fastdiv((unsigned)a.TW); a.dCW = make_fastdiv((unsigned)a.CW); a.dXW = make_fastdiv((unsigned)a.XW);
    a.dK = make_fastdiv((unsigned)a.K); a.dncs = make_fastdiv((unsigned)a.ncs);
    const int lds = 2 * (a.xpix + a.tpix) * 128 + kFixed;
    static unsigned char lds_set[64];
    // the attribute is set once per device, so to the largest size the search above can return, not to this request's
    if (const int e = ensure_dynamic_lds((const void*)block_c64_pair_kernel, kLds, lds_set)) return e;
    const int ncu = device_cus();
    const int grid = a.nunits < ncu ? a.nunits : ncu;
    hipLaunchKernelGGL(block_c64_pair_kernel, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

}  // namespace asl_detail
