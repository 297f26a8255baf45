// The I3D stem, nn.Conv3d(3, 64, 7, stride 2, pad 3) + BatchNorm3d + ReLU on packed RGB clips, as an LDS-resident-patch
// kernel (gfx950).
//
// The gather-GEMM stages every (kd, kh) filter row of every output pixel separately: an input element crosses the
// L2 -> LDS path once per window it belongs to (12 times at stride 2).  Here a workgroup owns 512 consecutive output
// pixels of ONE output frame and, per depth tap kd, holds the input rows they touch in LDS: the packed input is
// [n][frames + 6][size + 6][size + 6][4] with the zero padding in memory, so those rows are one contiguous byte range
// of plane 2t + kd, copied by linear 1 KiB LDS-DMA pieces, and each element is staged once per depth tap.  A K step is
// one filter row (kd, kh): 8 pixels x 4 channels = 32 contiguous elements per output pixel, read straight from the
// row-major patch -- neighbouring output pixels are 16 bytes apart, so a fragment read of 16 pixels covers 256
// consecutive bytes and the four k-groups of a lane quad overlap them by broadcast.
//
// Everything else is the scheme of conv3x3_pp.hip: 8 waves, each a 64-pixel x 64-channel register tile (all waves share
// the 64 output channels), weight stages (64 rows x 32 k) through a 3-deep LDS-DMA ring two stages ahead, the next
// depth tap's patch streaming in while the current one is consumed, counted s_waitcnt vmcnt, the two halves of the
// workgroup one phase apart, two workgroups per CU (76 KiB of LDS each), epilogue from registers through a buffer
// descriptor.
//
// DMA issue: waves 0-3 (the early half) carry the weight stages, one piece per wave and step; waves 4-7 (the late half)
// carry the next depth tap's patch, two pieces per wave in filter rows 0..3 (32 pieces) and none after, so its last
// piece is retired well before the first read.  Each half therefore has its own compile-time wait counts and needs
// only one per-lane DMA offset.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "asl_hip.h"
#include "common.h"
#include "patch_args.h"

namespace asl_detail {

struct Stem3dArgs {
    const char* x;
    const char* w;
    const float* scale;
    const float* bias;
    char* y;
    int D1, H1, W1;                 // output grid per clip
    int Dp, Hp, Wp;                 // halo'd input, pixels
    int wk;                         // packed K per weight row (49 * 32)
    long long x_bytes, y_bytes;
    unsigned w_bytes;
    long long out_sn;               // elements
    int out_sd, out_sh, out_sw, out_off0;
    int relu;
    int tiles_per_frame;
    FastDiv dW1, dTPF, dD1;
};

struct Stem3dShape {
    static constexpr int TP = 512;                      // pixel slots per tile
    static constexpr int PIECES = 32;                   // 1 KiB pieces per patch buffer
    static constexpr int PATCH_BYTES = PIECES * 1024;
    static constexpr int WST = 64 * 64;                 // one weight stage: 64 rows x 32 k
    static constexpr int RING = 3;
    static constexpr int LDS = RING * WST + 2 * PATCH_BYTES;
};

__global__ __launch_bounds__(512, 4) void stem3d_kernel(const Stem3dArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    using S = Stem3dShape;
    constexpr int FM = 4, FN = 4, WST = S::WST;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    char* const patches = smem + S::RING * WST;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = wave >= 4;                        // the half that runs one phase behind
    const bool wload = wave < 4;                        // waves that carry the weight stages

    int lbid;                                           // neighbouring tiles (overlapping input rows) on one XCD
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lbid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int frame = (int)fdiv((unsigned)lbid, a.dTPF), tt = lbid - frame * a.tiles_per_frame;
    const int clip = (int)fdiv((unsigned)frame, a.dD1), t = frame - clip * a.D1;
    const int HW = a.H1 * a.W1;
    const int m0 = tt * S::TP, mend = m0 + S::TP < HW ? m0 + S::TP : HW;
    const int y0 = (int)fdiv((unsigned)m0, a.dW1), y1 = (int)fdiv((unsigned)(mend - 1), a.dW1);
    const int row_bytes = a.Wp * 8;
    const unsigned npb = (unsigned)((2 * (y1 - y0) + 7) * row_bytes);      // patch bytes (<= PATCH_BYTES, host-checked)
    const unsigned plane_bytes = (unsigned)a.Hp * (unsigned)row_bytes;

    // ---- DMA descriptors ---------------------------------------------------------------------------
    const long long xoff = ((long long)(clip * a.Dp + 2 * t) * a.Hp + 2 * y0) * row_bytes;
    const long long xrem = a.x_bytes - xoff;
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + xoff), 0,
                                                                   (unsigned)(xrem > 0x7fffffffLL ? 0x7fffffffLL : xrem), 0x00020000);
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
    constexpr unsigned kOob = 0x80000000u;              // beyond num_records: the DMA writes zeros
    // The one per-lane DMA offset of the K loop.  Waves 0-3: weight piece w = rows 16w .. 16w+15 (64 B each) of a stage,
    // chunk swizzle (row & 8) ? 3 : 0.  Waves 4-7: 16 bytes per lane of a linear 1 KiB patch piece.
    unsigned dvoff;
    {
        const int row = (wave & 3) * 16 + (lane >> 2);
        dvoff = wload ? (unsigned)(2 * row * a.wk + 16 * ((lane & 3) ^ ((row & 8) ? 3 : 0))) : (unsigned)lane * 16u;
    }
    // piece p of depth tap kt (kt = 7: none left) into patch buffer kt & 1
    auto dma_patch_piece = [&](int kt, int p, unsigned voff) {
        const bool live = kt < 7 && (unsigned)p * 1024u < npb;
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)kt * plane_bytes + (unsigned)p * 1024u);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(patches + (kt & 1) * S::PATCH_BYTES + p * 1024),
                                                 16, live ? voff : kOob, live ? so : 0u, 0, 0);
    };
    auto dma_weights = [&](int s, int slot) {           // stage s = kt * 7 + kh, waves 0-3
        const bool live = s < 49;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(ring + slot * WST + (wave & 3) * 1024), 16,
                                                 live ? dvoff : kOob, live ? (unsigned)s * 64u : 0u, 0, 0);
    };

    // ---- per-lane fragment geometry ----------------------------------------------------------------
    const int fr = lane & 15, fq = lane >> 4;
    const int lds0 = (int)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    int u[FM];                                          // patch byte offset of the lane's k-group in filter row 0
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        int m = m0 + wave * (FM * 16) + i * 16 + fr;
        m = m < mend ? m : mend - 1;                    // idle slots recompute the tile's last pixel; never stored
        const int y = (int)fdiv((unsigned)m, a.dW1), x = m - y * a.W1;
        u[i] = (2 * (y - y0) * a.Wp + 2 * x) * 8 + fq * 16;
    }
    // weight rows in the permuted order of conv3x3_pp.hip: a lane's accumulators for channel fragments 2h, 2h+1 are
    // eight consecutive output channels of its pixel (fragment j reads row (j >> 1) * 32 + (j & 1) * 4 + row0; bit 3 of
    // the row, which picks the swizzle, is the same for all four, so they are constant offsets from one address)
    int wrd0;
    {
        const int row0 = (fr >> 2) * 8 + (fr & 3);
        wrd0 = row0 * 64 + ((fq ^ ((row0 & 8) ? 3 : 0)) << 4);
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: patch of depth tap 0 (four pieces per wave), weight stages 0 and 1 ------------------
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_patch_piece(0, wave * 4 + k, (unsigned)lane * 16u);
    if (wload) {
        dma_weights(0, 0);
        dma_weights(1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (late) asm volatile("s_barrier" ::: "memory");  // run one phase behind
    // Every per-lane address of the K loop is in its register from here on.  (A value the compiler had parked in scratch
    // around the prologue and reloaded just ahead of the loop made it guard the loop's first LDS read with
    // s_waitcnt vmcnt(0) -- on every iteration, draining the DMA queue once per depth tap.)
    asm volatile("" : "+v"(wrd0), "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(dvoff));

    // ---- K loop: depth taps x filter rows -----------------------------------------------------------
    int kt = 0, r3 = 0;                                 // r3 = kt % 3: stage kt * 7 + kh sits in ring slot (r3 + kh) % 3
    int pboff = 0;                                      // LDS address of the current patch
    auto step = [&](auto khc) {
        constexpr int kh = decltype(khc)::value;
        const int s = kt * 7 + kh;
        int slot = r3 + kh % 3, slot_next = r3 + (kh + 2) % 3;
        slot = slot >= 3 ? slot - 3 : slot;
        slot_next = slot_next >= 3 ? slot_next - 3 : slot_next;
        // load phase: this wave's DMA pieces, then the fragment reads of this step
        if (wload) {
            dma_weights(s + 2, slot_next);
        } else if constexpr (kh < 4) {
            dma_patch_piece(kt + 1, kh * 8 + (wave - 4) * 2, dvoff);
            dma_patch_piece(kt + 1, kh * 8 + (wave - 4) * 2 + 1, dvoff);
        }
        const char* wb = ring + slot * WST + wrd0;
        const int so = __builtin_amdgcn_readfirstlane(pboff + kh * row_bytes);
        bf16x8 xf[FM], wf[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = *(const bf16x8*)(wb + (j >> 1) * 2048 + (j & 1) * 256);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            int so_i = so;
            asm volatile("" : "+s"(so_i));              // keeps the 28 row addresses out of registers (one add per read)
            const int ad = u[i] + so_i;
            xf[i] = *(const __attribute__((address_space(3))) bf16x8*)(unsigned long long)(unsigned)ad;
        }
        // everything issued before this load phase has landed (the previous step's weight stage among it)
        if (wload) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kh < 4 ? 2 : 0) : "memory");
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
    };
#pragma unroll 1
    for (kt = 0; kt < 7; ++kt) {
        pboff = lds0 + S::RING * WST + (kt & 1) * S::PATCH_BYTES;
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        r3 = r3 == 2 ? 0 : r3 + 1;
    }
    if (!late) asm volatile("s_barrier" ::: "memory");  // the early half waits out the late half's last phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // trailing zero-fill pieces have landed

    // ---- epilogue from registers: BatchNorm, ReLU, 16 bytes per (pixel fragment, 32-channel half) ----
    {
        constexpr unsigned kOobOff = 0xffffffffu;
        __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_bytes, 0x00020000);
        const float floor_v = a.relu ? 0.f : -__builtin_inff();
        const int ch0 = fq * 8;
        unsigned oe[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            int m = m0 + wave * (FM * 16) + i * 16 + fr;
            asm volatile("" : "+v"(m));                  // recomputed here rather than carried across the K loop
            const bool ok = m < mend;
            m = ok ? m : mend - 1;
            const int y = (int)fdiv((unsigned)m, a.dW1), x = m - y * a.W1;
            const long long o = (long long)clip * a.out_sn + (long long)t * a.out_sd + (long long)y * a.out_sh +
                                (long long)x * a.out_sw + a.out_off0 + ch0;
            oe[i] = ok ? (unsigned)(2 * o) : kOobOff;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, bi0 = {0.f, 0.f, 0.f, 0.f}, bi1 = bi0;
            if (a.scale) { sc0 = *(const f32x4*)(a.scale + ch0 + h * 32); sc1 = *(const f32x4*)(a.scale + ch0 + h * 32 + 4); }
            if (a.bias) { bi0 = *(const f32x4*)(a.bias + ch0 + h * 32); bi1 = *(const f32x4*)(a.bias + ch0 + h * 32 + 4); }
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

}  // namespace asl_detail
using namespace asl_detail;

extern "C" int asl_stem3d_bf16(const asl_stem3d_params* p, const void* x, const void* w, const float* scale,
                               const float* bias, void* y, asl_stream_t stream) {
    using S = Stem3dShape;
    if (!p || !x || !w || !y) return ASL_EINVAL;
    if (p->n <= 0 || p->frames < 2 || p->size < 2 || (p->frames & 1) || (p->size & 1)) return ASL_EINVAL;
    if (p->out_halo_d < 0 || p->out_halo < 0 || (p->relu != 0 && p->relu != 1)) return ASL_EINVAL;
    if (p->cout != 64) return p->cout > 0 ? ASL_ENOTSUP : ASL_EINVAL;
    if (p->w_rows < 64 || (p->w_rows & 127)) return ASL_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)scale & 15) || ((uintptr_t)bias & 15))
        return ASL_EINVAL;
    Stem3dArgs a;
    a.x = (const char*)x; a.w = (const char*)w; a.scale = scale; a.bias = bias; a.y = (char*)y;
    a.D1 = p->frames / 2; a.H1 = a.W1 = p->size / 2;
    a.Dp = p->frames + 6; a.Hp = a.Wp = p->size + 6;
    a.wk = 49 * 32;
    a.x_bytes = 8LL * p->n * a.Dp * a.Hp * a.Wp;
    a.w_bytes = (unsigned)(2LL * 64 * a.wk);             // only the 64 live rows are read
    const long long owp = a.W1 + 2 * p->out_halo, odp = a.D1 + 2 * p->out_halo_d;
    a.out_sw = 64; a.out_sh = (int)(owp * 64);
    const long long plane = owp * owp * 64;
    if (plane > 0x7fffffffLL) return ASL_EINVAL;
    a.out_sd = (int)plane; a.out_sn = plane * odp;
    a.out_off0 = (int)(((long long)p->out_halo_d * owp + p->out_halo) * owp + p->out_halo) * 64;
    a.y_bytes = 2LL * p->n * a.out_sn;
    if (a.y_bytes >= 0xffffffffLL) return ASL_ENOTSUP;  // the epilogue addresses y through one buffer descriptor
    if (7LL * a.Hp * a.Wp * 8 >= 0x7fffffffLL) return ASL_ENOTSUP;     // seven planes as a 32-bit DMA offset
    a.relu = p->relu;
    const int hw = a.H1 * a.W1;
    a.tiles_per_frame = (hw + S::TP - 1) / S::TP;
    for (int tt = 0; tt < a.tiles_per_frame; ++tt) {     // every tile's input rows must fit a patch buffer
        const int m0 = tt * S::TP, mend = m0 + S::TP < hw ? m0 + S::TP : hw;
        const long long rows = 2 * ((mend - 1) / a.W1 - m0 / a.W1) + 7;
        if (rows * a.Wp * 8 > S::PATCH_BYTES) return ASL_ENOTSUP;
    }
    const long long tiles = (long long)p->n * a.D1 * a.tiles_per_frame;
    if (tiles > 0x7fffffffLL) return ASL_EINVAL;
    a.dW1 = make_fastdiv((unsigned)a.W1); a.dTPF = make_fastdiv((unsigned)a.tiles_per_frame); a.dD1 = make_fastdiv((unsigned)a.D1);
    static unsigned char lds_set[64];
    if (const int e = ensure_dynamic_lds((const void*)stem3d_kernel, S::LDS, lds_set)) return e;
    hipLaunchKernelGGL(stem3d_kernel, dim3((unsigned)tiles), dim3(512), S::LDS, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
