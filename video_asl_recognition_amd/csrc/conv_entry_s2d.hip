// Entry of a stride-2 residual block (gfx950): the block's first convolution, 3x3 / stride 2 / pad 1, and its 1x1 / stride 2
// downsample, both from ONE launch over a space-to-depth ("cell") copy of the block's input.
//
// Cell tensor: x2[n][Hc][Wc][4][C], Hc = (H+2)/2, Wc = (W+2)/2; pixel (Y, X) of the halo'd (halo 1) input sits in cell
// (Y >> 1, X >> 1), plane (Y & 1) * 2 + (X & 1).  The kernel that produces the block's input writes this layout from its
// own epilogue (out_s2d of asl_conv2d_nhwc_bf16 / asl_basic_block_c64_bf16), so no repacking pass exists.
//
// On cells the stride-2 convolution is a stride-1 one: output pixel (oy, ox) reads cells (oy + a, ox + b), a, b in {0, 1},
//   plane 0 (even row, even column): taps (ky, kx) = (0,0) (0,2) (2,0) (2,2) at cell offsets 0, 1, Wc, Wc + 1
//   plane 1 (even row, odd column):  taps (0,1) (2,1)                        at 0, Wc
//   plane 2 (odd row, even column):  taps (1,0) (1,2)                        at 0, 1
//   plane 3 (odd row, odd column):   tap  (1,1)                              at 0      (also the downsample's only input)
// so, as in conv3x3_pp.hip, the cells a run of consecutive output pixels needs are one consecutive range, they sit in
// LDS chunk-major, and every tap reads them at a constant offset.  K advances in steps of (32-channel group, tap), nine
// per group; the 64-byte-per-cell patch of a (group, plane) streams into one of THREE buffers two planes ahead of its use
// (plane 3 is read for a single step, so two buffers would leave no time to refill), one or two 1 KiB LDS-DMA pieces per
// wave and step.  The weights do not pass through LDS: the host packs them in MFMA-fragment order, and every wave loads
// its four 1 KiB fragments of the next step straight into the registers the current step's MFMAs have just read
// (measured equal to an LDS-DMA weight ring in this loop, and it frees the ring's LDS for the third patch buffer).
// Workgroup = 8 waves, 256 output pixels x 128 channels, two per CU, the halves one phase apart as in conv3x3_pp.hip.
// Downsample tiles are extra workgroups of the same launch (tile index >= tiles_n): one step per channel group.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "asl_hip.h"
#include "common.h"
#include "patch_args.h"

#ifndef ASL_ENTRY_EXP
#define ASL_ENTRY_EXP 0      // diagnostic builds (timing only, results wrong): 1 no patch DMA, 2 no weight loads, 3 no K loop
#endif

namespace asl_detail {

struct EntryArgs {
    const char* x2;
    const char* w;          // conv: [tiles_n][9 * C/32 steps][8 fragments][1 KiB], fragment order
    const char* wds;        // downsample: [tiles_n][C/32 steps][8 fragments][1 KiB] or null
    const float* scale;
    const float* bias;
    const float* scale_ds;
    const float* bias_ds;
    char* y;
    char* yds;
    int M, N, OH, OW, Hc, Wc, C;
    int bmv;                // valid output pixels per 256-slot tile
    long long x_bytes, w_bytes, wds_bytes, y_bytes, yds_bytes;
    long long y_sn, yds_sn;
    int y_sh, y_sw, y_off0, yds_sh, yds_sw, yds_off0;
    int relu;
    FastDiv dOW, dOHW;
    int tiles_m, tiles_n, has_ds;
};

constexpr int kEntryPcap = 384;                      // cells per patch buffer
constexpr int kEntryPatchBytes = kEntryPcap * 64;
constexpr int kEntryLds = 3 * kEntryPatchBytes;      // 72 KiB: two workgroups per CU

__global__ __launch_bounds__(512, 4) void conv_entry_s2d_kernel(const EntryArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int FM = 4, FN = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wc = wave & 1;
    const bool late = wave >= 4;

    int lbid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lbid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ntn = a.tiles_n * (1 + a.has_ds);
    const int tn2 = lbid % ntn, tm = lbid / ntn;
    const bool ds = tn2 >= a.tiles_n;
    const int tn = ds ? tn2 - a.tiles_n : tn2;
    const int m0 = tm * a.bmv;
    const int OHW = a.OH * a.OW;
    auto centre = [&](int m) -> long long {            // cell of output pixel m (its top-left tap cell)
        const int n = (int)fdiv(m, a.dOHW), r1 = m - n * OHW;
        const int oy = (int)fdiv(r1, a.dOW), ox = r1 - oy * a.OW;
        return ((long long)n * a.Hc + oy) * a.Wc + ox;
    };
    const int mend = (m0 + a.bmv < a.M ? m0 + a.bmv : a.M);
    const long long P0 = centre(m0);
    const int NP = (int)(centre(mend - 1) + a.Wc + 1 - P0) + 1;        // <= kEntryPcap (checked on the host)
    const int ncg = ASL_ENTRY_EXP == 3 ? 0 : a.C >> 5;

    const long long cell_bytes = 8LL * a.C;
    const char* xbase = a.x2 + P0 * cell_bytes;
    const long long xrem = a.x_bytes - P0 * cell_bytes;
    const unsigned xrec = (unsigned)(xrem > 0x7fffffffLL ? 0x7fffffffLL : xrem);
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, xrec, 0x00020000);
    const int nsteps = ds ? ncg : 9 * ncg;
    const long long wtile = 8192LL * nsteps;
    const char* wbase = (ds ? a.wds : a.w) + tn * wtile;
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, (unsigned)wtile, 0x00020000);
    constexpr unsigned kOob = 0x80000000u;

    // patch piece j of (group cg, plane): cells 16j .. 16j+15, chunk-major in LDS -- byte (cell p, chunk c) at
    // (p >> 4) * 1024 + c * 256 + (p & 15) * 16; DMA lane -> cell 16j + lane % 16, chunk lane / 16
    const unsigned pvoff = (unsigned)((lane & 15) * cell_bytes + (lane >> 4) * 16);
    auto dma_piece = [&](int cg, int plane, int buf, int k) {
        const int j = wave + 8 * k;
        const bool live = ASL_ENTRY_EXP != 1 && cg < ncg && 16 * j < NP;     // else zeros into the piece
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(16LL * j * cell_bytes + plane * 2 * a.C + cg * 64));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(smem + buf * kEntryPatchBytes + j * 1024),
                                                 16, live ? pvoff : kOob, live ? so : 0u, 0, 0);
    };
    // weight fragment j of this wave's 64 channels for step s: 1 KiB at ((s * 2 + wc) * 4 + j) * 1024
    const unsigned wfv = (unsigned)(wc * 4096 + lane * 16);
    bf16x8 wf[FN];
    auto load_weights = [&](int s) {
        // (the load behind the last step re-reads step 0: in range, never used)
        const unsigned so = __builtin_amdgcn_readfirstlane(s < nsteps ? (unsigned)(s * 8192) : 0u);
        if (ASL_ENTRY_EXP == 2 && s > 0) return;
#pragma unroll
        for (int j = 0; j < FN; ++j)
            wf[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wfv + j * 1024, so, 0));
    };

    const int fr = lane & 15, fq = lane >> 4;
    // u[i]: image address of (top-left tap cell, chunk 3) of this lane's pixel in fragment i; adding a tap's
    // (cell >> 4) << 10 | (cell & 15) << 4 carries from the low cell bits through the all-ones chunk field
    int u[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        int m = m0 + wp * (FM * 16) + i * 16 + fr;
        m = m < mend ? m : mend - 1;                    // idle slots recompute the tile's last pixel; never stored
        const int tl = (int)(centre(m) - P0);
        u[i] = ((tl >> 4) << 10) | 0x300 | ((tl & 15) << 4);
    }
    const int chunk_bits = fq << 8;
    constexpr int chunk_mask = 0x300;
    const int lds0 = (int)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one K step: `issue` patch pieces were put in flight by the caller in this load phase; read the four pixel
    // fragments of buffer `buf` at cell offset `o`, wait for everything older than this phase's pieces (the step's
    // weights among it), swap roles, multiply, fetch the next step's weights, swap roles
    auto step_body = [&](auto issuec, int buf, int o, int s) {
        constexpr int issue = decltype(issuec)::value;
        const int so = __builtin_amdgcn_readfirstlane((((o >> 4) << 10) | ((o & 15) << 4)) + lds0 + buf * kEntryPatchBytes);
        bf16x8 xf[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            int so_i = so;
            asm volatile("" : "+s"(so_i));              // keeps the per-tap addresses out of long-lived registers
            const int r = u[i] + so_i;
            const int ad = (r & ~chunk_mask) | chunk_bits;
            xf[i] = *(const __attribute__((address_space(3))) bf16x8*)(unsigned long long)(unsigned)ad;
        }
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(issue) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        load_weights(s + 1);
        asm volatile("s_barrier" ::: "memory");
    };
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    if (!ds) {
        // ---- prologue: plane 0 of group 0, the first piece of plane 1, the weights of step 0 ----------------------
#pragma unroll
        for (int k = 0; k < 3; ++k) dma_piece(0, 0, 0, k);
        dma_piece(0, 1, 1, 0);
        load_weights(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (late) asm volatile("s_barrier" ::: "memory");                 // run one phase behind
        // ---- K loop: buffers of (cg, plane 0..3) are b0, b1, b2, b0; of (cg + 1, plane 0 / 1) b1 / b2.  A plane's pieces
        // go out two or more steps before its first read and after the last read of the buffer's previous plane.
        int b0 = 0, b1 = 1, b2 = 2;
        const int Wc = a.Wc;
#pragma unroll 1
        for (int cg = 0; cg < ncg; ++cg) {
            const int s = 9 * cg;
            dma_piece(cg, 1, b1, 1);                                   step_body(I1{}, b0, 0, s);
            dma_piece(cg, 1, b1, 2);                                   step_body(I1{}, b0, 1, s + 1);
            dma_piece(cg, 2, b2, 0);                                   step_body(I1{}, b0, Wc, s + 2);
            dma_piece(cg, 2, b2, 1);                                   step_body(I1{}, b0, Wc + 1, s + 3);
            dma_piece(cg, 2, b2, 2); dma_piece(cg, 3, b0, 0);          step_body(I2{}, b1, 0, s + 4);
            dma_piece(cg, 3, b0, 1);                                   step_body(I1{}, b1, Wc, s + 5);
            dma_piece(cg, 3, b0, 2); dma_piece(cg + 1, 0, b1, 0);      step_body(I2{}, b2, 0, s + 6);
            dma_piece(cg + 1, 0, b1, 1); dma_piece(cg + 1, 0, b1, 2);  step_body(I2{}, b2, 1, s + 7);
            dma_piece(cg + 1, 1, b2, 0);                               step_body(I1{}, b0, 0, s + 8);
            const int t = b0; b0 = b1; b1 = b2; b2 = t;
        }
    } else {
        // ---- downsample tile: one step per channel group on plane 3; the patch of group cg sits in buffer cg % 3 and
        // is put in flight two steps ahead
#pragma unroll
        for (int k = 0; k < 3; ++k) dma_piece(0, 3, 0, k);
#pragma unroll
        for (int k = 0; k < 3; ++k) dma_piece(1, 3, 1, k);
        load_weights(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (late) asm volatile("s_barrier" ::: "memory");
        int b0 = 0, b1 = 1, b2 = 2;
#pragma unroll 1
        for (int cg = 0; cg < ncg; ++cg) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dma_piece(cg + 2, 3, b2, k);
            step_body(I3{}, b0, 0, cg);
            const int t = b0; b0 = b1; b1 = b2; b2 = t;
        }
    }
    if (!late) asm volatile("s_barrier" ::: "memory");  // the early half waits out the late half's last phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // trailing zero-fill pieces and the unused last weight loads

    // ---- epilogue, straight from registers (the host packs weight rows so that a lane's accumulators of channel
    // fragments 2h, 2h+1 are eight consecutive channels): scale / bias, ReLU, one 16-byte store per (fragment, half)
    {
        constexpr unsigned kOobOff = 0xffffffffu;
        char* const yb = ds ? a.yds : a.y;
        const long long ybytes = ds ? a.yds_bytes : a.y_bytes;
        const long long sn = ds ? a.yds_sn : a.y_sn;
        const int sh = ds ? a.yds_sh : a.y_sh, sw = ds ? a.yds_sw : a.y_sw, off0 = ds ? a.yds_off0 : a.y_off0;
        const float* scale = ds ? a.scale_ds : a.scale;
        const float* bias = ds ? a.bias_ds : a.bias;
        __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)yb, 0, (unsigned)ybytes, 0x00020000);
        const float floor_v = (a.relu && !ds) ? 0.f : -__builtin_inff();
        const int ch0 = tn * 128 + wc * 64 + fq * 8;
        unsigned oe[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            int m = m0 + wp * (FM * 16) + i * 16 + fr;
            asm volatile("" : "+v"(m));
            const bool ok = m < mend;
            m = ok ? m : mend - 1;
            const int n = (int)fdiv(m, a.dOHW), r1 = m - n * OHW;
            const int oy = (int)fdiv(r1, a.dOW), ox = r1 - oy * a.OW;
            const long long o = (long long)n * sn + (long long)oy * sh + (long long)ox * sw + off0 + ch0;
            oe[i] = ok ? (unsigned)(2 * o) : kOobOff;
        }
#pragma unroll
        for (int h = 0; h < FN / 2; ++h) {
            f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, bi0 = {0.f, 0.f, 0.f, 0.f}, bi1 = bi0;
            if (scale) { sc0 = *(const f32x4*)(scale + ch0 + h * 32); sc1 = *(const f32x4*)(scale + ch0 + h * 32 + 4); }
            if (bias) { bi0 = *(const f32x4*)(bias + ch0 + h * 32); bi1 = *(const f32x4*)(bias + ch0 + h * 32 + 4); }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const f32x4 v0 = acc[i][2 * h] * sc0 + bi0;
                const f32x4 v1 = acc[i][2 * h + 1] * sc1 + bi1;
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ov[e] = f32_to_bf16(fmaxf(v0[e], floor_v));
                    ov[4 + e] = f32_to_bf16(fmaxf(v1[e], floor_v));
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, ov), yrs, oe[i] == kOobOff ? kOobOff : oe[i] + h * 64, 0, 0);
            }
        }
    }
#endif
}

// Largest cell range (cells, both ends included plus the taps' reach) any tile of bmv consecutive output pixels needs
static int entry_max_cells(int n, int oh, int ow, int hc, int wc, int bmv) {
    const long long M = (long long)n * oh * ow;
    auto centre = [&](long long m) -> long long {
        const long long img = m / (oh * ow), r1 = m - img * (oh * ow);
        const long long oy = r1 / ow, ox = r1 - oy * ow;
        return (img * hc + oy) * wc + ox;
    };
    // the pattern of tile starts repeats every lcm(bmv, oh*ow) pixels: checking that many (or all) tiles is exact
    long long period = (long long)oh * ow;
    {
        long long g = bmv, b = period;
        while (b) { const long long t = g % b; g = b; b = t; }
        period = period / g * bmv;
    }
    const long long lim = period < M ? period + bmv : M;
    long long worst = 0;
    for (long long m0 = 0; m0 < lim; m0 += bmv) {
        const long long me = m0 + bmv < M ? m0 + bmv : M;
        const long long np = centre(me - 1) + wc + 1 - centre(m0) + 1;
        worst = np > worst ? np : worst;
    }
    // the last tile of the whole tensor may be a short one at any phase: it is never longer than a full tile's range
    return (int)worst;
}

}  // namespace asl_detail
using namespace asl_detail;

extern "C" int asl_block_entry_s2_bf16(const asl_entry_params* p, const void* x2, const void* w, const float* scale,
                                       const float* bias, const void* w_ds, const float* scale_ds, const float* bias_ds,
                                       void* y, void* y_ds, asl_stream_t stream) {
    if (!p || !x2 || !w || !y) return ASL_EINVAL;
    if (p->n <= 0 || p->h <= 0 || p->w <= 0 || (p->h & 1) || (p->w & 1)) return ASL_EINVAL;
    if (p->cin <= 0 || (p->cin & 31) || p->cout <= 0 || p->out_halo < 0 || p->ds_halo < 0) return ASL_EINVAL;
    if ((w_ds != nullptr) != (y_ds != nullptr)) return ASL_EINVAL;
    if (((uintptr_t)x2 & 15) || ((uintptr_t)w & 15) || ((uintptr_t)w_ds & 15) || ((uintptr_t)y & 15) || ((uintptr_t)y_ds & 15))
        return ASL_EINVAL;
    if (((uintptr_t)scale & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)scale_ds & 15) || ((uintptr_t)bias_ds & 15)) return ASL_EINVAL;
    if (p->cout & 127) return ASL_ENOTSUP;               // whole 128-channel tiles only
    if (p->w_rows != p->cout) return ASL_EINVAL;
    EntryArgs a = {};
    a.x2 = (const char*)x2; a.w = (const char*)w; a.wds = (const char*)w_ds;
    a.scale = scale; a.bias = bias; a.scale_ds = scale_ds; a.bias_ds = bias_ds;
    a.y = (char*)y; a.yds = (char*)y_ds;
    a.OH = p->h / 2; a.OW = p->w / 2; a.Hc = (p->h + 2) / 2; a.Wc = (p->w + 2) / 2; a.C = p->cin; a.N = p->cout;
    const long long M = (long long)p->n * a.OH * a.OW;
    if (M > 0x7fffffffLL) return ASL_EINVAL;
    a.M = (int)M;
    a.x_bytes = 2LL * p->n * a.Hc * a.Wc * 4 * p->cin;
    const int steps = 9 * (p->cin / 32);
    a.tiles_n = p->cout / 128;
    a.w_bytes = 8192LL * steps * a.tiles_n;
    a.wds_bytes = 8192LL * (p->cin / 32) * a.tiles_n;
    if (a.w_bytes > 0x7fffffffLL) return ASL_EINVAL;
    auto geom = [&](int halo, long long& sn, int& sh, int& sw, int& off0, long long& bytes) {
        const int owp = a.OW + 2 * halo, ohp = a.OH + 2 * halo;
        sn = (long long)ohp * owp * p->cout; sh = owp * p->cout; sw = p->cout; off0 = (halo * owp + halo) * p->cout;
        bytes = 2LL * p->n * sn;
    };
    geom(p->out_halo, a.y_sn, a.y_sh, a.y_sw, a.y_off0, a.y_bytes);
    geom(p->ds_halo, a.yds_sn, a.yds_sh, a.yds_sw, a.yds_off0, a.yds_bytes);
    if (a.y_bytes >= 0xffffffffLL || a.yds_bytes >= 0xffffffffLL) return ASL_ENOTSUP;   // 32-bit store offsets
    a.relu = p->relu; a.has_ds = w_ds ? 1 : 0;
    a.dOW = make_fastdiv((unsigned)a.OW); a.dOHW = make_fastdiv((unsigned)(a.OH * a.OW));
    // the widest tile of consecutive output pixels whose cell range fits a patch buffer (found once per shape)
    {
        struct Key { int n, oh, ow, bmv; };
        static thread_local Key cache[8];
        static thread_local int next = 0;
        int bmv = 0;
        for (const Key& k : cache)
            if (k.bmv && k.n == p->n && k.oh == a.OH && k.ow == a.OW) bmv = k.bmv;
        if (!bmv) {
            for (int b = 256; b >= 64 && !bmv; b -= 16)
                if (entry_max_cells(p->n, a.OH, a.OW, a.Hc, a.Wc, b) <= kEntryPcap) bmv = b;
            if (!bmv) return ASL_ENOTSUP;
            cache[next] = Key{p->n, a.OH, a.OW, bmv};
            next = (next + 1) & 7;
        }
        a.bmv = bmv;
    }
    a.tiles_m = (a.M + a.bmv - 1) / a.bmv;
    const long long nblk = (long long)a.tiles_m * a.tiles_n * (1 + a.has_ds);
    if (nblk > 0x7fffffffLL) return ASL_EINVAL;
    static unsigned char lds_set[64];
    if (const int e = ensure_dynamic_lds((const void*)conv_entry_s2d_kernel, kEntryLds, lds_set)) return e;
    hipLaunchKernelGGL(conv_entry_s2d_kernel, dim3((unsigned)nblk), dim3(512), kEntryLds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
