// A whole 64-channel ResNet BasicBlock in one kernel (gfx950):
//     y = relu( bn2(conv2( relu(bn1(conv1(x))) )) + x ),   conv1, conv2: 3x3 / stride 1 / pad 1, 64 -> 64.
//
// Run as two launches the block moves 2.0 MB of HBM traffic per 56x56 frame (x in, t out, t in, x again as
// the residual, y out) for 0.46 GFLOP, which makes both halves memory-bound.  Here the intermediate t never
// leaves the CU:
//
//  * one persistent 4-wave workgroup per CU, one wave per SIMD, walks strips of R full-width output rows of one
//    image.  LDS holds only activations: the strip's input patch (R+4 rows, halo columns included) and the
//    intermediate (R+2 rows), both as 128-byte pixels in a chunk-major image whose 16-pixel fragment reads are
//    bank-conflict free (byte (p, c) at (p>>4)*2048 + c*256 + (p&15)*16);
//  * BOTH filter banks live in registers: wave (c, q) owns output channels 32c..32c+31 and every other
//    32-pixel tile, and keeps its 32 x 576 slab of each conv's weights as the A operands of
//    v_mfma_f32_32x32x16_bf16 (2 x 144 VGPRs; the kernel runs at one wave per SIMD by design).  A K step is
//    one ds_read_b128 of 32 pixels x 8 channels and one MFMA -- no weight traffic at all after start-up;
//  * conv1's rows are permuted so a lane ends up with 16 consecutive channels of its pixel: BN + ReLU, two
//    16-byte LDS stores into the intermediate image (zeros for rows outside the image: conv2's padding);
//  * conv2 adds the residual, read back from global memory (L2-warm: the same bytes were the strip's patch
//    a few microseconds earlier), so the patch buffer is dead during conv2 and the NEXT strip's patch streams
//    into it by LDS-DMA meanwhile.
//
// Per strip conv1 recomputes two rows ((R+2)/R) and the patch re-reads four ((R+4)/R); at R = 7 on 56x56 that is
// 1.14x the MFMAs for 0.5x the HBM bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asl_hip.h"
#include "common.h"
#include "patch_args.h"
#include "block_mfma.h"

#ifndef ASL_BLOCK_EXP
#define ASL_BLOCK_EXP 0        // diagnostic builds: 1 = K loop without its LDS reads, 2 = without its MFMAs (results wrong)
#endif

namespace asl_detail {

struct BlockArgs {
    const char* x;
    const char* w1;
    const char* w2;
    const float* s1;
    const float* b1;
    const float* s2;
    const float* b2;
    char* y;
    int n, H, W, Wp, Hp;           // Wp, Hp: halo'd input geometry (in_halo each side)
    int in_halo, out_halo;
    int R, strips_per_img, nstrips;
    int wk;                        // elements per packed weight row (9 * 64)
    int xpix, tpix;                // pixels of the patch / intermediate images, multiples of 16
    long long x_bytes;
    long long out_sn;              // output image stride, elements
    int out_sh;                    // output row stride, elements (pixel stride is 64)
    FastDiv dW;
};

// The epilogue of a tile (16 values per lane: BatchNorm, residual or row mask, ReLU, bf16, two 16-byte stores)
// is cut into slices -- a pair of values each, then the two stores -- that ride between the MFMAs of the NEXT
// tile's K loop, where the matrix pipe covers their issue slots.  kConv2: residual add and global stores;
// otherwise LDS stores.
template <bool kConv2>
struct TileEpilogue {
    f32x16 acc;                    // the finished tile
    f32x4 sc[4], bi[4];            // BatchNorm scale / bias of the lane's 16 channels
    bf16x8 res[2];                 // residual (conv2)
    char* dst;                     // where the lane's 16 channels go: LDS (conv1) or global (conv2)
    bool store;                    // false: nothing pending, or a masked pixel slot
    unsigned out[8];               // the 16 results as bf16 pairs
    static constexpr int kStride2 = kConv2 ? 16 : 256;   // byte distance between the two 16-byte halves at dst
    // slice S: even S < 16 computes values S and S + 1 (one packed dword), 16 and 17 store.  Each result passes
    // through an empty volatile asm: without it the optimiser sinks all of the arithmetic down to the stores.
    template <int S>
    __device__ __forceinline__ void slice() {
        if constexpr (S < 16 && (S & 1) == 0) {
            constexpr int j = S >> 1, g = j >> 1, e = 2 * (j & 1);        // values 2j, 2j + 1 = (group g, element e, e + 1)
            float v0 = acc[2 * j] * sc[g][e] + bi[g][e], v1 = acc[2 * j + 1] * sc[g][e + 1] + bi[g][e + 1];
            if constexpr (kConv2) {
                v0 += bf16_to_f32(res[g >> 1][4 * (g & 1) + e]);
                v1 += bf16_to_f32(res[g >> 1][4 * (g & 1) + e + 1]);
            }
            // ReLU on the rounded pair: as 16-bit integers, negative bf16 values (sign bit set) are negative
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
            unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, bf16x2_t));   // one v_cvt_pk_bf16_f32
            asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(pk));
            out[j] = pk;
        } else if constexpr (S == 16) {
            if (store) *(i32x4*)dst = i32x4{(int)out[0], (int)out[1], (int)out[2], (int)out[3]};
        } else if constexpr (S == 17) {
            if (store) *(i32x4*)(dst + kStride2) = i32x4{(int)out[4], (int)out[5], (int)out[6], (int)out[7]};
        }
    }
    template <int S>
    __device__ __forceinline__ void flush_from() {      // the last tile of a phase has no K loop to hide under
        slice<S>();
        if constexpr (S < 17) flush_from<S + 1>();
    }
};

template <bool kConv2, int S>
struct ConvSteps {
    // kAhead reads in flight, kRing = kAhead + 1 register slots: the slot a read refills was consumed by the MFMA
    // before the one just issued.  (An MFMA that accumulates onto its predecessor's result starts up to 32 cycles
    // after it issues and reads its operands as it goes; refilling the slot of the MFMA just issued let an LDS
    // return overtake that read -- measured: half of all outputs wrong.)
    static constexpr int kAhead = 6, kRing = kAhead + 1;
    // issue the fragment read of step S (tap S / 4, 16-channel group S % 4)
    __device__ static __forceinline__ void read(bf16x8 (&xf)[kRing], const unsigned (&ad)[9]) {
        xf[S % kRing] = lds_frag<(S & 3) * 512>(ad[S >> 2]);
    }
    // the read issued right behind the MFMA of step S - kAhead: never into that MFMA's own fragment register
    __device__ static __forceinline__ void read_after(bf16x8 (&xf)[kRing], const unsigned (&ad)[9]) {
        xf[S % kRing] = lds_frag_after<(S & 3) * 512>(ad[S >> 2], xf[(S - kAhead) % kRing]);
    }
    __device__ static __forceinline__ void run(f32x16 (&acc)[2], const bf16x8 (&w)[36], bf16x8 (&xf)[kRing], const unsigned (&ad)[9],
                                               TileEpilogue<kConv2>& epi) {
        constexpr int inflight = 35 - S < kAhead - 1 ? 35 - S : kAhead - 1;      // reads younger than step S's
#if ASL_BLOCK_EXP != 2
        if constexpr (S == 35) mfma_last2<kConv2, inflight, S % 2>(acc, w[S], xf[S % kRing]);
        else mfma_step<kConv2, inflight, S < 2 ? 1 : 0>(acc[S & 1], w[S], xf[S % kRing]);
#endif
#if ASL_BLOCK_EXP != 1
        if constexpr (S + kAhead < 36) ConvSteps<kConv2, S + kAhead>::read_after(xf, ad);
#endif
        epi.template slice<S>();                          // a slice of the previous tile's epilogue ...
        __builtin_amdgcn_sched_barrier(0);                // ... pinned here: the scheduler would gather them in one place
        if constexpr (S + 1 < 36) ConvSteps<kConv2, S + 1>::run(acc, w, xf, ad, epi);
    }
};

// One 32-pixel x 32-channel tile of a 3x3 convolution over a chunk-major LDS image: 36 K steps of 16 channels.
// `uenc` is the image's LDS byte address plus the address of the lane's top-left tap pixel with the chunk field
// (bits 8-10) all ones; adding a tap's (pixel >> 4) << 11 | (pixel & 15) << 4 then carries from the low pixel bits
// straight into the high ones, and the lane's k half replaces the field: two vector instructions per tap.
// `epi` is the previous tile's pending epilogue.  conv2 takes its filter bank from the accumulation registers.
template <bool kConv2>
__device__ __forceinline__ f32x16 conv_tile(const bf16x8 (&w)[36], unsigned uenc, unsigned khalf, const unsigned (&tapenc)[9],
                                            TileEpilogue<kConv2>& epi) {
    unsigned ad[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) ad[tap] = ((uenc + tapenc[tap]) & ~0x700u) | khalf;
    bf16x8 xf[ConvSteps<kConv2, 0>::kRing];
    ConvSteps<kConv2, 0>::read(xf, ad);
    ConvSteps<kConv2, 1>::read(xf, ad);
    ConvSteps<kConv2, 2>::read(xf, ad);
    ConvSteps<kConv2, 3>::read(xf, ad);
    ConvSteps<kConv2, 4>::read(xf, ad);
    ConvSteps<kConv2, 5>::read(xf, ad);
    f32x16 acc[2];
    ConvSteps<kConv2, 0>::run(acc, w, xf, ad, epi);
    return acc[0] + acc[1];
}

__global__ __launch_bounds__(256, 1) void block_c64_kernel(const BlockArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wq = wave >> 1;              // channel half, pixel-tile parity
    const int lp = lane & 31, lh = lane >> 5;             // MFMA column (pixel) / row, k half
    const int Wp = a.Wp, W = a.W, R = a.R;
    const unsigned xlds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const unsigned tlds = xlds + (unsigned)a.xpix * 128u;
    char* const tbuf = smem + (long long)a.xpix * 128;

    // ---- both filter banks into registers.  MFMA row r = 8i + 4h + e is output channel 32*wc + 16h + 4i + e,
    // so accumulator register 4i + e of lane half h is channel 32*wc + 16h + 4i + e: 16 consecutive channels.
    bf16x8 w1[36], w2[36];
    {
        const int ch = 32 * wc + 16 * ((lp >> 2) & 1) + 4 * (lp >> 3) + (lp & 3);
        const short* r1 = (const short*)a.w1 + (long long)ch * a.wk + lh * 8;
        const short* r2 = (const short*)a.w2 + (long long)ch * a.wk + lh * 8;
#pragma unroll
        for (int s = 0; s < 36; ++s) {                    // step s = (tap, 16-channel group): K index tap*64 + k16*16
            w1[s] = *(const bf16x8*)(r1 + s * 16);
            w2[s] = *(const bf16x8*)(r2 + s * 16);
        }
    }
    // the intermediate image's halo columns are never written: zero the image once
    for (int i = tid; i < a.tpix * 8; i += 256) *(i32x4*)(tbuf + i * 16) = i32x4{0, 0, 0, 0};
    // BatchNorm scale / bias of both convs behind the images: [s1 | b1 | s2 | b2] x 64 floats.  The epilogues read
    // them from LDS so that they never queue behind the patch DMA in the vector-memory counter.
    float* const bn = (float*)(smem + (long long)(a.xpix + a.tpix) * 128);
    bn[tid] = tid < 64 ? a.s1[tid] : tid < 128 ? a.b1[tid - 64] : tid < 192 ? a.s2[tid - 128] : a.b2[tid - 192];

    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)a.x_bytes, 0x00020000);
    constexpr unsigned kOob = 0x80000000u;
    const int xpieces = a.xpix >> 3;                      // 1 KiB pieces: 16 pixels x 4 chunks; two per 16-pixel group
    // piece j of a strip's patch: group j >> 1, chunks 4*(j & 1) .. +3; lane -> pixel 16*(j>>1) + lane % 16,
    // chunk 4*(j & 1) + lane / 16.  Rows above the first / below the last image are out of range: zeros.
    auto dma_piece = [&](int j, long long P0) {
        const long long gp = P0 + 16 * (j >> 1) + (lane & 15);
        const long long off = gp * 128 + (4 * (j & 1) + (lane >> 4)) * 16;
        const unsigned vo = (off < 0 || off >= a.x_bytes) ? kOob : (unsigned)off;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(smem + j * 1024), 16, vo, 0, 0, 0);
    };
    // first halo'd pixel of strip sid's patch: image row y0 - 2 (halo'd row y0 - 2 + in_halo), halo'd column in_halo - 1
    auto patch_origin = [&](int sid, int& img, int& y0) -> long long {
        img = sid / a.strips_per_img;
        y0 = (sid - img * a.strips_per_img) * R;
        return ((long long)img * a.Hp + (y0 - 2 + a.in_halo)) * Wp + (a.in_halo - 1);
    };

    int sid = blockIdx.x;
    if (sid < a.nstrips) {
        int img, y0;
        const long long P0 = patch_origin(sid, img, y0);
        for (int j = wave; j < xpieces; j += 4) dma_piece(j, P0);
    }
    unsigned tapenc[9];                                   // wave-uniform: the compiler keeps them in scalar registers
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int o = (tap / 3) * Wp + tap % 3;
        tapenc[tap] = (unsigned)(((o >> 4) << 11) | ((o & 15) << 4));
    }
    const unsigned khalf = (unsigned)lh << 8;
    const int n1 = (R + 2) * W, n2 = R * W;               // pixel slots of conv1 / conv2 per strip
    const int t1 = (n1 + 31) >> 5, t2 = (n2 + 31) >> 5;

#if defined(ASL_BLOCK_STAMPS)   // diagnostic build: per-phase s_memtime stamps of workgroup 0 into y's first halo row
    unsigned long long* stamp = (unsigned long long*)a.y;
    int stamp_i = 0;
#define STAMP() do { if (blockIdx.x == 0 && tid == 0 && stamp_i < 200) stamp[stamp_i] = __builtin_amdgcn_s_memtime(); ++stamp_i; } while (0)
#else
#define STAMP() do { } while (0)
#endif
    for (; sid < a.nstrips; sid += gridDim.x) {
        int img, y0;
        patch_origin(sid, img, y0);
        STAMP();
        // this strip's patch has landed (and, first time round, the zero fill); the previous strip's conv2 is over
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        STAMP();
        // Intermediate rows outside the image are conv2's zero padding: conv1 never stores them, and the first and
        // last strip of an image clear them here (an earlier strip left its own rows there).
        {
            const int rlo = y0 == 0 ? 1 : 0;                               // rows [0, rlo) ...
            const int rhi = a.H - y0 + 1 < R + 2 ? a.H - y0 + 1 : R + 2;   // ... and [rhi, R + 2) are outside
            for (int i = tid; i < rlo * Wp * 8; i += 256) *(i32x4*)(tbuf + img_addr(i >> 3) + (i & 7) * 256) = i32x4{0, 0, 0, 0};
            for (int i = rhi * Wp * 8 + tid; i < (R + 2) * Wp * 8; i += 256)
                *(i32x4*)(tbuf + img_addr(i >> 3) + (i & 7) * 256) = i32x4{0, 0, 0, 0};
        }
        // ---- conv1: intermediate rows y0-1 .. y0+R (slot m -> row m / W, column m % W) -------------------
        const int chl = 32 * wc + 16 * lh;
        {
            TileEpilogue<false> epi;
            epi.store = false;
            epi.dst = tbuf;
#pragma unroll
            for (int e = 0; e < 16; ++e) epi.acc[e] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                epi.sc[g] = *(const f32x4*)(bn + chl + 4 * g);
                epi.bi[g] = *(const f32x4*)(bn + 64 + chl + 4 * g);
            }
            for (int tile = wq; tile < t1; tile += 2) {
                int m = tile * 32 + lp;
                m = m < n1 ? m : n1 - 1;
                const int rt = (int)fdiv(m, a.dW), ct = m - rt * W;
                const int u = rt * Wp + ct;
                STAMP();
                const f32x16 acc = conv_tile<false>(w1, xlds + (unsigned)img_addr(u) + 0x700u, khalf, tapenc, epi);
                STAMP();
                const int yt = y0 - 1 + rt;
                epi.acc = acc;
                epi.store = tile * 32 + lp < n1 && yt >= 0 && yt < a.H;   // rows outside the image stay zero (see above)
                epi.dst = tbuf + img_addr(u + 1) + (chl >> 3) * 256;
            }
            epi.flush_from<0>();
        }
        STAMP();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // intermediate complete, patch reads over
        STAMP();

        // ---- conv2 + residual -> y; meanwhile the next strip's patch streams into the (dead) patch buffer ----
        const int nsid = sid + gridDim.x;
        long long nP0 = 0;
        int nimg, ny0;
        const bool more = nsid < a.nstrips;
        if (more) nP0 = patch_origin(nsid, nimg, ny0);
        int jn = wave;                                     // next DMA piece of this wave
        const int my_tiles = (t2 - wq + 1) >> 1;
        // all of them within this wave's first two tiles: early enough to land before the strip ends, and behind
        // the residual loads of exactly those tiles
        const int per_tile = more && my_tiles > 0 ? ((xpieces + 3) / 4 + 1) / 2 : 0;
        {
            TileEpilogue<true> epi;
            epi.store = false;
            epi.dst = a.y;
            epi.res[0] = epi.res[1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 16; ++e) epi.acc[e] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                epi.sc[g] = *(const f32x4*)(bn + 128 + chl + 4 * g);
                epi.bi[g] = *(const f32x4*)(bn + 192 + chl + 4 * g);
            }
            for (int tile = wq; tile < t2; tile += 2) {
                int m = tile * 32 + lp;
                const bool slot_ok = m < n2;
                m = slot_ok ? m : n2 - 1;
                const int ro = (int)fdiv(m, a.dW), co = m - ro * W;
                const int yo = y0 + ro;
                const bool ok = slot_ok && yo < a.H;
                // residual first (older than the DMA pieces below in the memory counter's order)
                const char* rp = a.x + 2 * ((((long long)img * a.Hp + (ok ? yo : 0) + a.in_halo) * Wp + co + a.in_halo) * 64 + chl);
                const bf16x8 r0 = *(const bf16x8*)rp, r1 = *(const bf16x8*)(rp + 16);
                for (int k = 0; k < per_tile && jn < xpieces; ++k, jn += 4) dma_piece(jn, nP0);
                const f32x16 acc = conv_tile<true>(w2, tlds + (unsigned)img_addr(ro * Wp + co) + 0x700u, khalf, tapenc, epi);
                epi.acc = acc;
                epi.res[0] = r0;
                epi.res[1] = r1;
                epi.store = ok;
                epi.dst = a.y + 2 * ((long long)img * a.out_sn + (long long)((ok ? yo : 0) + a.out_halo) * a.out_sh +
                                     (long long)(co + a.out_halo) * 64 + chl);
            }
            epi.flush_from<0>();
        }
        if (more)
            for (; jn < xpieces; jn += 4) dma_piece(jn, nP0);
        STAMP();
    }
#endif
}

}  // namespace asl_detail
using namespace asl_detail;

extern "C" int asl_basic_block_c64_bf16(const asl_block_params* p, const void* x, const void* w1, const float* scale1,
                                        const float* bias1, const void* w2, const float* scale2, const float* bias2,
                                        void* y, asl_stream_t stream) {
    if (!p || !x || !w1 || !w2 || !scale1 || !bias1 || !scale2 || !bias2 || !y) return ASL_EINVAL;
    if (p->n <= 0 || p->h <= 0 || p->w <= 0 || p->in_halo < 1 || p->out_halo < 0) return ASL_EINVAL;
    if (p->channels != 64) return ASL_ENOTSUP;
    if (p->w_rows < 64 || p->variant < 0 || p->variant > ASL_BLOCK_PAIRED) return ASL_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)w1 & 15) || ((uintptr_t)w2 & 15) || ((uintptr_t)y & 15) ||
        ((uintptr_t)scale1 & 15) || ((uintptr_t)bias1 & 15) || ((uintptr_t)scale2 & 15) || ((uintptr_t)bias2 & 15))
        return ASL_EINVAL;
    if (x == y) return ASL_EINVAL;                          // strips re-read rows that other strips have already replaced
    if (p->out_s2d && (p->out_s2d != 1 || p->out_halo != 1 || (p->h & 1) || (p->w & 1))) return ASL_EINVAL;
    if (p->variant != ASL_BLOCK_SOLO) {
        const int e = try_launch_block_c64_pair(p, x, w1, scale1, bias1, w2, scale2, bias2, y, (hipStream_t)stream);
        if (e != -1) return e;
        if (p->variant == ASL_BLOCK_PAIRED) return ASL_ENOTSUP;
    }
    if (p->out_s2d) return ASL_ENOTSUP;                     // only the paired kernel writes the cell layout
    BlockArgs a;
    a.x = (const char*)x; a.w1 = (const char*)w1; a.w2 = (const char*)w2; a.y = (char*)y;
    a.s1 = scale1; a.b1 = bias1; a.s2 = scale2; a.b2 = bias2;
    a.n = p->n; a.H = p->h; a.W = p->w; a.in_halo = p->in_halo; a.out_halo = p->out_halo;
    a.Wp = p->w + 2 * p->in_halo; a.Hp = p->h + 2 * p->in_halo;
    a.wk = 9 * 64;
    a.x_bytes = 2LL * p->n * a.Hp * a.Wp * 64;
    if (a.x_bytes >= 0x7fffffffLL) return ASL_EINVAL;      // 32-bit LDS-DMA offsets: split the batch
    const int owp = p->w + 2 * p->out_halo, ohp = p->h + 2 * p->out_halo;
    a.out_sn = (long long)ohp * owp * 64; a.out_sh = owp * 64;
    a.dW = make_fastdiv((unsigned)p->w);
    // the tallest strip whose patch (R+4 rows) and intermediate (R+2 rows) fit the 160 KiB of LDS
    constexpr int kLds = 160 * 1024;
    int R = 0;
    for (int r = p->h < 16 ? p->h : 16; r >= 1; --r) {
        const int xp = ((r + 4) * a.Wp + 15) & ~15, tp = ((r + 2) * a.Wp + 15) & ~15;
        if ((long long)(xp + tp) * 128 + 1024 <= kLds) { R = r; a.xpix = xp; a.tpix = tp; break; }
    }
    if (R < 1) return ASL_ENOTSUP;                          // rows too wide for a one-row strip
    // taps of the last intermediate row reach pixel (R+3)*Wp + W + 1 of the patch image: inside xpix by construction
    a.R = R;
    a.strips_per_img = (p->h + R - 1) / R;
    a.nstrips = p->n * a.strips_per_img;
    const int lds = (a.xpix + a.tpix) * 128 + 1024;      // images + BatchNorm constants
    static unsigned char lds_set[64];
    // the attribute is set once per device, so to the largest size the search above can return, not to this request's
    if (const int e = ensure_dynamic_lds((const void*)block_c64_kernel, kLds, lds_set)) return e;
    const int ncu = device_cus();
    const int grid = a.nstrips < ncu ? a.nstrips : ncu;
    hipLaunchKernelGGL(block_c64_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
