// Gather-GEMM for gfx950: convolution (2-D, 3-D, the packed 7x7 stem) and nn.Linear as one
// bf16 MFMA kernel.  Output row m is an output pixel; its K axis is "taps x contiguous
// channels", every tap a constant element offset from the pixel's first input element, so
// the zero halo of the activation layout supplies the padding and the activation operand
// is fetched by LDS-DMA (buffer_load ... lds) with one fixed per-lane offset for the whole
// K loop and one scalar offset per K step.
//
// MFMA orientation: the weight tile is the A operand (rows = output channels) and the
// pixel tile the B operand (columns = pixels), so each lane ends up holding 4 CONSECUTIVE
// channels of one pixel per 16x16 tile and the epilogue stores channels-last without a
// transpose.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asl_hip.h"
#include "common.h"
#include "epilogue.h"
#include "patch_args.h"
#include "gemm_args.h"
#include <stdlib.h>

namespace asl_detail {


// LDS chunk swizzle: slot (row, c') of a staged tile holds 16-byte chunk c' ^ swz(row) of that
// row.  Chosen so that a 16-row x 16-byte fragment read (ds_read_b128) is bank-conflict
// free under the instruction's own lane grouping: 128-byte rows need the 8 chunk slots
// of rows 2i, 2i+1 rotated per row pair, 64-byte rows need rows 8-15 flipped against 0-7.
template <int BK>
__device__ __forceinline__ int swz(int row) {
    if constexpr (BK == 64) return (row >> 1) & 7;
    else return (row & 8) ? 3 : 0;
}

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int BM, int BN, int BK, int WGM, int WGN, bool SPLIT = false, bool DUAL = false>
__global__ __launch_bounds__(WGM* WGN * 64) void gather_gemm_kernel(const GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub; the body uses device-only types
    constexpr int NW = WGM * WGN;
    constexpr int RB = BK * 2;        // bytes per staged row
    constexpr int CPR = RB / 16;      // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;     // rows per wave-wide LDS-DMA instruction (1 KiB)
    constexpr int X_BYTES = BM * RB, W_BYTES = BN * RB;
    constexpr int X_PW = X_BYTES / 1024 / NW, W_PW = W_BYTES / 1024 / NW;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;   // wave tile: pixels x channels
    constexpr int FM = WTM / 16, FN = WTN / 16;
    constexpr int KK = BK / 32;
    static_assert(X_PW >= 1 && W_PW >= 1, "tile too small for the wave count");
    static_assert(X_PW * NW * 1024 == X_BYTES && W_PW * NW * 1024 == W_BYTES, "tile/wave mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF = X_BYTES + W_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    // blocks that share an XCD (same blockIdx % 8) get neighbouring tiles: consecutive
    // logical ids walk the channel tiles of one pixel tile, then the next pixel tile.
    int lbid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lbid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tn = lbid % a.tiles_n, tm = lbid / a.tiles_n;
    const int m0 = tm * BM;
    const int n0img = (int)fdiv(m0, a.dODHW);      // first image this tile touches (block-uniform)

    // ---- LDS-DMA source offsets: fixed per lane for the whole K loop --------------------
    const char* xbase = a.x + 2 * (long long)n0img * a.in_sn;
    long long xrem = a.x_bytes - 2 * (long long)n0img * a.in_sn;
    const unsigned xrec = (unsigned)(xrem > 0x7fffffffLL ? 0x7fffffffLL : xrem);
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, xrec, 0x00020000);
    const char* wbase = a.w + 2 * (long long)(tn * BN) * a.wk;
    const unsigned wrec = (unsigned)(a.w_bytes - 2 * (long long)(tn * BN) * a.wk);
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, wrec, 0x00020000);

    unsigned xvoff[X_PW], wvoff[W_PW];
    unsigned xvoff2[DUAL ? X_PW : 1];   // DUAL: the same rows of the second activation operand
    __amdgpu_buffer_rsrc_t xrs2 = xrs;
    if constexpr (DUAL) {
        const long long x2rem = a.x2_bytes - 2 * (long long)n0img * a.in2_sn;
        xrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 + 2 * (long long)n0img * a.in2_sn), 0,
                                                 (unsigned)(x2rem > 0x7fffffffLL ? 0x7fffffffLL : x2rem), 0x00020000);
    }
#pragma unroll
    for (int i = 0; i < X_PW; ++i) {
        const int row = (wave * X_PW + i) * RPI + lane / CPR;
        const int chunk = (lane % CPR) ^ swz<BK>(row);
        const int m = m0 + row;
        if (m < a.M) {
            const int n = (int)fdiv(m, a.dODHW), r1 = m - n * (int)a.dODHW.d;
            const int d = (int)fdiv(r1, a.dOHW), r2 = r1 - d * (int)a.dOHW.d;
            const int h = (int)fdiv(r2, a.dOW), w_ = r2 - h * (int)a.dOW.d;
            const long long e = (long long)(n - n0img) * a.in_sn + (long long)d * a.in_sd +
                                (long long)h * a.in_sh + (long long)w_ * a.in_sw + a.in_off0;
            xvoff[i] = (unsigned)(2 * e + 16 * chunk);
            if constexpr (DUAL) {
                const long long e2 = (long long)(n - n0img) * a.in2_sn + (long long)h * a.in2_sh + (long long)w_ * a.in2_sw + a.in2_off0;
                xvoff2[i] = (unsigned)(2 * e2 + 16 * chunk);
            }
        } else {
            xvoff[i] = 0x80000000u;     // beyond num_records: the DMA writes zeros
            if constexpr (DUAL) xvoff2[i] = 0x80000000u;
        }
    }
#pragma unroll
    for (int i = 0; i < W_PW; ++i) {
        const int row = (wave * W_PW + i) * RPI + lane / CPR;
        const int chunk = (lane % CPR) ^ swz<BK>(row);
        wvoff[i] = (unsigned)(2 * row * a.wk + 16 * chunk);
    }

    // ---- K-step state (scalar): tap (d,r,q) and channel chunk -----------------------------
    int st_c = 0, st_q = 0, st_r = 0, st_d = 0;
    int st_i = 0;                            // DUAL: K steps issued so far
    unsigned st_xoff = 0, st_woff = 0;      // byte offsets for the NEXT stage() call
    auto stage = [&](int buf) {
        char* lx = smem + buf * BUF + wave * (X_PW * 1024);
        char* lw = smem + buf * BUF + X_BYTES + wave * (W_PW * 1024);
        // the offsets are wave-uniform by construction; say so, or hipcc wraps every DMA in a
        // readfirstlane loop
        const unsigned xo = __builtin_amdgcn_readfirstlane(st_xoff);
        const unsigned wo = __builtin_amdgcn_readfirstlane(st_woff);
        if constexpr (DUAL) {
            if (st_i < a.nsteps2) {          // a step of the second operand: its channels st_i * BK .. of every row
                const unsigned xo2 = __builtin_amdgcn_readfirstlane((unsigned)st_i * (unsigned)RB);
#pragma unroll
                for (int i = 0; i < X_PW; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs2, (__attribute__((address_space(3))) void*)(lx + i * 1024),
                                                             16, xvoff2[i], xo2, 0, 0);
#pragma unroll
                for (int i = 0; i < W_PW; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(lw + i * 1024),
                                                             16, wvoff[i], wo, 0, 0);
                st_woff += RB;
                ++st_i;
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < X_PW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(lx + i * 1024),
                                                     16, xvoff[i], xo, 0, 0);
#pragma unroll
        for (int i = 0; i < W_PW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(lw + i * 1024),
                                                     16, wvoff[i], wo, 0, 0);
        // advance to the next K step
        st_woff += RB;
        if (++st_c == a.ksteps_per_tap) {
            st_c = 0;
            if (++st_q == a.KW) {
                st_q = 0;
                if (++st_r == a.KH) { st_r = 0; ++st_d; }
            }
            st_xoff = 2u * (unsigned)(st_d * a.tap_sd + st_r * a.tap_sh + st_q * a.tap_sw);
        } else {
            st_xoff += RB;
        }
    };

    // ---- fragment read offsets ----------------------------------------------------------
    const int fr = lane & 15, fq = lane >> 4;
    int xrd[FM][KK], wrd[FN][KK];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int row = wm * WTM + i * 16 + fr;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) xrd[i][kk] = row * RB + (((kk * 4 + fq) ^ swz<BK>(row)) << 4);
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int row = wn * WTN + i * 16 + fr;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            wrd[i][kk] = X_BYTES + row * RB + (((kk * 4 + fq) ^ swz<BK>(row)) << 4);
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // output coordinates and residual loads first: they complete under the K loop
    OutGeom g;
    g.scale = a.scale; g.bias = a.bias; g.res = a.res; g.y = a.y; g.M = a.M; g.N = a.N;
    g.dOW = a.dOW; g.dOHW = a.dOHW; g.dODHW = a.dODHW;
    g.out_sn = a.out_sn; g.out_sd = a.out_sd; g.out_sh = a.out_sh; g.out_sw = a.out_sw; g.out_off0 = a.out_off0;
    g.res_sn = a.res_sn; g.res_sd = a.res_sd; g.res_sh = a.res_sh; g.res_sw = a.res_sw; g.res_off0 = a.res_off0;
    g.relu = a.relu; g.out_f32 = a.out_f32; g.res_f32 = a.res_f32; g.out_lo = a.out_lo; g.res_lo = a.res_lo;
    EpiState<FM, FN> est;
    epilogue_prepare<FM, FN, SPLIT>(est, g, m0 + wm * WTM, tn * BN + wn * WTN, lane);

    const int nsteps = a.ntaps * a.ksteps_per_tap + (DUAL ? a.nsteps2 : 0);
    stage(0);
    wait_vm0();
    __syncthreads();
    int buf = 0;
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) stage(buf ^ 1);
        const char* base = smem + buf * BUF;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            bf16x8 xf[FM], wf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) xf[i] = *(const bf16x8*)(base + xrd[i][kk]);
#pragma unroll
            for (int j = 0; j < FN; ++j) wf[j] = *(const bf16x8*)(base + wrd[j][kk]);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        if constexpr (DUAL) {
            if (s == a.nsteps2 - 1) {        // the second operand is done: its sum takes its own per-channel scale
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int ch = tn * BN + wn * WTN + j * 16 + fq * 4;
                    const f32x4 r = ch + 3 < a.N ? *(const f32x4*)(a.pre_scale + ch) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
                    for (int i = 0; i < FM; ++i) acc[i][j] *= r;
                }
            }
        }
        wait_vm0();
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue (the loop's last barrier guarantees every wave is done reading the stages) -----
    static_assert(NW * kEpiWaveBytes(WTN) <= 2 * BUF, "epilogue stage does not fit the K-loop LDS");
    epilogue_store<FM, FN, SPLIT>(acc, smem + wave * kEpiWaveBytes(WTN), g, est, lane);
#endif
}


template <int BM, int BN, int BK, int WGM, int WGN, bool SPLIT = false, bool DUAL = false>
int launch(const GemmArgs& a, hipStream_t stream) {
    constexpr int lds = 2 * (BM + BN) * BK * 2;
    static_assert(lds <= 65536, "tiles above 64 KiB of LDS need hipFuncSetAttribute");
    GemmArgs b = a;
    b.tiles_m = (a.M + BM - 1) / BM;
    b.tiles_n = (a.N + BN - 1) / BN;
    const long long nblk = (long long)b.tiles_m * b.tiles_n;
    if (nblk > 0x7fffffffLL) return ASL_EINVAL;
    hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, BK, WGM, WGN, SPLIT, DUAL>), dim3((unsigned)nblk), dim3(WGM * WGN * 64), lds, stream, b);
    return (int)hipGetLastError();
}

// largest element offset (exclusive) a geom can reach for the given output grid
long long geom_extent(const asl_geom& g, int nb, int od, int oh, int ow) {
    return (long long)(nb - 1) * g.sn + (long long)(od - 1) * g.sd + (long long)(oh - 1) * g.sh +
           (long long)(ow - 1) * g.sw + g.off0;
}

bool geom_nonneg(const asl_geom& g) { return g.sn >= 0 && g.sd >= 0 && g.sh >= 0 && g.sw >= 0 && g.off0 >= 0; }

}  // namespace asl_detail
using namespace asl_detail;

extern "C" int asl_gather_gemm_bf16(const asl_gemm_params* p, const void* x, const void* w, const float* scale,
                                    const float* bias, const void* res, void* y, asl_stream_t stream) {
    if (!p || !x || !w || !y) return ASL_EINVAL;
    if (p->variant < ASL_CONV_AUTO || p->variant > ASL_CONV_C64 || (p->out_s2d != 0 && p->out_s2d != 1)) return ASL_EINVAL;
    if (p->split_out != 0 && p->split_out != 1) return ASL_EINVAL;
    if (p->split_out) {
        // split-precision output (and residual): bf16 only, lo halves 16-byte aligned behind the hi halves, both in bounds
        if (p->out_f32 || p->res_f32 || p->out_s2d || p->out_lo < p->cout || (p->out_lo & 7)) return ASL_EINVAL;
        if (res && (p->res_lo < p->cout || (p->res_lo & 7))) return ASL_EINVAL;
        if (p->variant > ASL_CONV_GATHER) return ASL_ENOTSUP;               // only the gather-GEMM writes it
    }
    if (p->nb <= 0 || p->od <= 0 || p->oh <= 0 || p->ow <= 0 || p->cout <= 0) return ASL_EINVAL;
    if (p->kd <= 0 || p->kh <= 0 || p->kw <= 0 || p->ck <= 0 || (p->ck & 31)) return ASL_EINVAL;
    if (p->tap_sd < 0 || p->tap_sh < 0 || p->tap_sw < 0) return ASL_EINVAL;
    if (!geom_nonneg(p->in) || !geom_nonneg(p->out) || (res && !geom_nonneg(p->res))) return ASL_EINVAL;
    const long long M = (long long)p->nb * p->od * p->oh * p->ow;
    if (M > 0x7fffffffLL) return ASL_EINVAL;
    const long long wk = (long long)p->kd * p->kh * p->kw * p->ck;
    if (p->w_rows < p->cout || (p->w_rows & 127) || wk * p->w_rows * 2 > 0x7fffffffLL) return ASL_EINVAL;
    // 16-byte alignment of every LDS-DMA source and 8-byte alignment of epilogue vectors
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)res & 15)) return ASL_EINVAL;
    if (((uintptr_t)scale & 15) || ((uintptr_t)bias & 15)) return ASL_EINVAL;
    if ((p->in.sn & 7) || (p->in.sd & 7) || (p->in.sh & 7) || (p->in.sw & 7) || (p->in.off0 & 7)) return ASL_EINVAL;
    if ((p->tap_sd & 7) || (p->tap_sh & 7) || (p->tap_sw & 7)) return ASL_EINVAL;
    // epilogue vectors: 8 bf16 or 2 x 4 fp32 per lane, 16-byte aligned
    const int oal = p->out_f32 ? 3 : 7;
    if ((p->out.sn & oal) || (p->out.sd & oal) || (p->out.sh & oal) || (p->out.sw & oal) || (p->out.off0 & oal))
        return ASL_EINVAL;
    if (res && ((p->res.sn & 7) || (p->res.sd & 7) || (p->res.sh & 7) || (p->res.sw & 7) || (p->res.off0 & 7)))
        return ASL_EINVAL;
    // bounds: last input element read, last output / residual element touched
    const long long tap_ext = (long long)(p->kd - 1) * p->tap_sd + (long long)(p->kh - 1) * p->tap_sh +
                              (long long)(p->kw - 1) * p->tap_sw + p->ck;
    if (geom_extent(p->in, p->nb, p->od, p->oh, p->ow) + tap_ext > p->x_elems) return ASL_EINVAL;
    if (p->out_s2d) {
        // cell layout of the halo-1 output: the last pixel (oh-1, ow-1) sits in cell ((oh)>>1, (ow)>>1), plane <= 3
        if (p->od != 1 || p->out_f32 || p->out.sd <= 0 || (p->out.sd & 7)) return ASL_EINVAL;
        const long long last = (long long)(p->nb - 1) * p->out.sn + (long long)(p->oh >> 1) * p->out.sh +
                               (long long)(p->ow >> 1) * p->out.sw + 3LL * p->out.sd + p->out.off0 + p->cout;
        if (last > p->y_elems) return ASL_EINVAL;
    } else if (geom_extent(p->out, p->nb, p->od, p->oh, p->ow) + (p->split_out ? p->out_lo : 0) + p->cout > p->y_elems) return ASL_EINVAL;
    if (res && geom_extent(p->res, p->nb, p->od, p->oh, p->ow) + (p->split_out ? p->res_lo : 0) + p->cout > p->res_elems) return ASL_EINVAL;
    // one tile's rows must stay within a 2 GiB window of the input (32-bit DMA offsets)
    {
        const long long odhw = (long long)p->od * p->oh * p->ow;
        const long long imgs = 256 / odhw + 2;
        if (2 * (imgs * p->in.sn + (long long)p->od * p->in.sd + (long long)p->oh * p->in.sh +
                 (long long)p->ow * p->in.sw + p->in.off0 + tap_ext) > 0x7fffffffLL)
            return ASL_EINVAL;
    }

    GemmArgs a;
    a.x = (const char*)x; a.w = (const char*)w; a.scale = scale; a.bias = bias;
    a.res = (const char*)res; a.y = (char*)y;
    a.M = (int)M; a.N = p->cout;
    a.dOW = make_fastdiv(p->ow); a.dOHW = make_fastdiv(p->oh * p->ow); a.dODHW = make_fastdiv(p->od * p->oh * p->ow);
    a.in_sn = p->in.sn; a.in_sd = p->in.sd; a.in_sh = p->in.sh; a.in_sw = p->in.sw; a.in_off0 = p->in.off0;
    a.out_sn = p->out.sn; a.out_sd = p->out.sd; a.out_sh = p->out.sh; a.out_sw = p->out.sw; a.out_off0 = p->out.off0;
    a.res_sn = p->res.sn; a.res_sd = p->res.sd; a.res_sh = p->res.sh; a.res_sw = p->res.sw; a.res_off0 = p->res.off0;
    a.KH = p->kh; a.KW = p->kw; a.ntaps = p->kd * p->kh * p->kw;
    a.tap_sd = p->tap_sd; a.tap_sh = p->tap_sh; a.tap_sw = p->tap_sw;
    a.wk = (int)wk;
    a.x_bytes = 2 * p->x_elems; a.w_bytes = 2 * wk * p->w_rows;
    a.relu = p->relu; a.out_f32 = p->out_f32; a.res_f32 = p->res_f32;
    a.tiles_m = a.tiles_n = 0;
    a.out_lo = p->out_lo; a.res_lo = p->res_lo;

    hipStream_t s = (hipStream_t)stream;
    if (p->split_out) {                         // parity mode: one tile shape per K granule, split-precision epilogue
        if (p->ck % 64 == 0) { a.ksteps_per_tap = p->ck / 64; return launch<128, 128, 64, 2, 4, true>(a, s); }
        a.ksteps_per_tap = p->ck / 32;
        return launch<128, 128, 32, 2, 4, true>(a, s);
    }
    // 3x3 / stride 1 / pad 1 on a halo'd NHWC tensor with C % 64 == 0: the LDS-resident-patch kernel
    if (p->kd == 1 && p->kh == 3 && p->kw == 3 && p->od == 1 && (p->ck & 63) == 0 && !p->out_f32 && !p->res_f32 &&
        p->tap_sw == p->ck && p->in.sw == p->ck && p->in.sh == p->tap_sh && p->tap_sh % p->ck == 0 &&
        p->in.sn % p->tap_sh == 0 && p->variant != ASL_CONV_GATHER) {
        const int C = p->ck, Wp = p->tap_sh / C, Hp = (int)(p->in.sn / p->tap_sh);
        const int halo = (Wp - p->ow) / 2;
        if (halo >= 1 && Wp == p->ow + 2 * halo && Hp == p->oh + 2 * halo &&
            p->in.off0 == ((halo - 1) * Wp + (halo - 1)) * C) {
            PatchArgs b = {};
            b.x = a.x; b.w = a.w; b.scale = a.scale; b.bias = a.bias; b.res = a.res; b.y = a.y;
            b.M = a.M; b.N = a.N; b.H = p->oh; b.W = p->ow; b.Hp = Hp; b.Wp = Wp; b.in_halo = halo;
            b.C = C; b.nchunks = C / 64; b.bmv = 0; b.wk = a.wk; b.x_bytes = a.x_bytes; b.w_bytes = a.w_bytes;
            b.y_bytes = 2 * p->y_elems; b.res_bytes = res ? 2 * p->res_elems : 0;
            b.out_sn = a.out_sn; b.out_sh = a.out_sh; b.out_sw = a.out_sw; b.out_off0 = a.out_off0;
            b.res_sn = a.res_sn; b.res_sh = a.res_sh; b.res_sw = a.res_sw; b.res_off0 = a.res_off0;
            b.relu = a.relu; b.tiles_m = b.tiles_n = 0; b.stamps = nullptr; b.variant = p->variant;
            b.out_s2d = p->out_s2d ? p->out.sd : 0;
#if defined(ASL_STAMPS)
            if (const char* e = getenv("ASL_STAMP_PTR")) b.stamps = (unsigned long long*)strtoull(e, nullptr, 16);
#endif
            b.dOW = a.dOW; b.dOHW = a.dOHW; b.dODHW = a.dODHW;
            if (b.out_s2d) {                                       // only the 8-wave large-tile kernel writes cells
                const int r2 = try_launch_conv3x3_pp(b, p->nb, s);
                return r2 >= 0 ? r2 : ASL_ENOTSUP;
            }
            int r = try_launch_conv3x3_c64(b, p->nb, s);           // 64 -> 64: persistent, weights resident in LDS
            if (r >= 0) return r;
            r = try_launch_conv3x3_wide(b, p->nb, s);              // C >= 128: 16-wave workgroup, 512 x 128 tile
            if (r >= 0) return r;
            r = try_launch_conv3x3_pp(b, p->nb, s);                // C >= 128: 8-wave ping-pong workgroups
            if (r >= 0) return r;
            if ((p->variant == ASL_CONV_AUTO || p->variant == ASL_CONV_PATCH) &&
                (b.bmv = patch_conv_tile(p->nb, p->oh, p->ow, halo)) > 0)
                return launch_conv3x3_patch(b, s);
        }
    }
    // 3x3x3 / stride 1 / pad 1 on a halo'd NDHWC tensor: the 3-D instance of the 8-wave large-tile kernel
    if (p->kd == 3 && p->kh == 3 && p->kw == 3 && (p->ck & 31) == 0 && !p->out_f32 && !p->res_f32 && !res && !p->out_s2d &&
        p->tap_sw == p->ck && p->in.sw == p->ck && p->in.sh == p->tap_sh && p->in.sd == p->tap_sd && p->tap_sh % p->ck == 0 &&
        p->tap_sh > 0 && p->tap_sd % p->tap_sh == 0 && p->in.sn % p->tap_sd == 0 && p->variant != ASL_CONV_GATHER) {
        const int C = p->ck, Wp = p->tap_sh / C, Hp = p->tap_sd / p->tap_sh, Dp = (int)(p->in.sn / p->tap_sd);
        const int halo = (Wp - p->ow) / 2, hd = (Dp - p->od) / 2;
        if (halo >= 1 && hd >= 1 && Wp == p->ow + 2 * halo && Hp == p->oh + 2 * halo && Dp == p->od + 2 * hd &&
            p->in.off0 == (((hd - 1) * Hp + (halo - 1)) * Wp + (halo - 1)) * C) {
            PatchArgs b = {};
            b.x = a.x; b.w = a.w; b.scale = a.scale; b.bias = a.bias; b.res = nullptr; b.y = a.y;
            b.M = a.M; b.N = a.N; b.H = p->oh; b.W = p->ow; b.Hp = Hp; b.Wp = Wp; b.in_halo = halo;
            b.D = p->od; b.Dp = Dp; b.in_halo_d = hd; b.out_sd = a.out_sd;
            b.C = C; b.nchunks = C / 64; b.bmv = 0; b.wk = a.wk; b.x_bytes = a.x_bytes; b.w_bytes = a.w_bytes;
            b.y_bytes = 2 * p->y_elems; b.res_bytes = 0;
            b.out_sn = a.out_sn; b.out_sh = a.out_sh; b.out_sw = a.out_sw; b.out_off0 = a.out_off0;
            b.relu = a.relu; b.stamps = nullptr; b.variant = p->variant; b.out_s2d = 0;
            b.dOW = a.dOW; b.dOHW = a.dOHW; b.dODHW = a.dODHW;
            const int r3 = try_launch_conv3x3x3_pp(b, p->nb, s);
            if (r3 >= 0) return r3;
        }
    }
    if (p->variant > ASL_CONV_GATHER || p->out_s2d) return ASL_ENOTSUP;   // the forced kernel / the cell output does not fit this request
    if (p->ck % 64 == 0) {
        a.ksteps_per_tap = p->ck / 64;
        // eight waves per workgroup (two workgroups per CU as before): twice the waves issuing the stage's LDS-DMA pieces
        // and covering each other's waits; measured 3-8 % faster than four waves on every ResNet / I3D shape
        if (p->cout <= 64) return launch<128, 64, 64, 4, 2>(a, s);
        return launch<128, 128, 64, 2, 4>(a, s);
    }
    a.ksteps_per_tap = p->ck / 32;
    if (p->cout <= 64) return launch<128, 64, 32, 2, 2>(a, s);
    return launch<128, 128, 32, 2, 4>(a, s);
}

// One launch for the last two GEMMs of a ResNet bottleneck whose shortcut is a projection:
//     y = act( scale * ( W1 . t  +  pre_scale * ( W2 . x_strided ) ) + bias )
// with t the 1x1 / stride 1 operand and x the 1x1 / stride `stride2` one.  The shortcut's K steps run first, the
// accumulator is multiplied by pre_scale (fp32, one rounding), and the main branch accumulates on top: the shortcut
// tensor is neither written nor read back.
extern "C" int asl_conv1x1_dual_bf16(const asl_dual1x1_params* p, const void* t, const void* x, const void* w, const float* scale,
                                     const float* bias, const float* pre_scale, void* y, asl_stream_t stream) {
    if (!p || !t || !x || !w || !y || !pre_scale) return ASL_EINVAL;
    if (p->n <= 0 || p->h <= 0 || p->w <= 0 || p->cin <= 0 || p->cin2 <= 0 || p->cout <= 0 || p->stride2 <= 0) return ASL_EINVAL;
    if ((p->cin & 63) || (p->cin2 & 63) || (p->cout & 7)) return (p->cin & 31) || (p->cin2 & 31) || (p->cout & 7) ? ASL_EINVAL : ASL_ENOTSUP;
    if (p->in_halo < 0 || p->in2_halo < 0 || p->out_halo < 0 || p->out_coff < 0 || p->out_coff + p->cout > p->out_ctot) return ASL_EINVAL;
    if ((p->out_ctot & 7) || (p->out_coff & 7) || (p->relu != 0 && p->relu != 1)) return ASL_EINVAL;
    const long long wk = (long long)p->cin + p->cin2;
    if (p->w_rows < p->cout || (p->w_rows & 127) || wk * p->w_rows * 2 > 0x7fffffffLL) return ASL_EINVAL;
    if (((uintptr_t)t & 15) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)scale & 15) ||
        ((uintptr_t)bias & 15) || ((uintptr_t)pre_scale & 15))
        return ASL_EINVAL;
    const long long M = (long long)p->n * p->h * p->w;
    if (M > 0x7fffffffLL) return ASL_EINVAL;
    const int wp = p->w + 2 * p->in_halo, hp = p->h + 2 * p->in_halo;
    const int w2 = p->w * p->stride2, h2 = p->h * p->stride2;
    const int wp2 = w2 + 2 * p->in2_halo, hp2 = h2 + 2 * p->in2_halo;
    const int owp = p->w + 2 * p->out_halo, ohp = p->h + 2 * p->out_halo;
    GemmArgs a = {};
    a.x = (const char*)t; a.w = (const char*)w; a.scale = scale; a.bias = bias; a.res = nullptr; a.y = (char*)y;
    a.M = (int)M; a.N = p->cout;
    a.dOW = make_fastdiv(p->w); a.dOHW = make_fastdiv(p->h * p->w); a.dODHW = make_fastdiv(p->h * p->w);
    a.in_sn = (long long)hp * wp * p->cin; a.in_sd = 0; a.in_sh = wp * p->cin; a.in_sw = p->cin;
    a.in_off0 = (p->in_halo * wp + p->in_halo) * p->cin;
    a.out_sn = (long long)ohp * owp * p->out_ctot; a.out_sd = 0; a.out_sh = owp * p->out_ctot; a.out_sw = p->out_ctot;
    a.out_off0 = (p->out_halo * owp + p->out_halo) * p->out_ctot + p->out_coff;
    a.KH = 1; a.KW = 1; a.ntaps = 1; a.tap_sd = a.tap_sh = a.tap_sw = 0;
    a.ksteps_per_tap = p->cin / 64;
    a.wk = (int)wk;
    a.x_bytes = 2LL * p->n * a.in_sn; a.w_bytes = 2 * wk * p->w_rows;
    a.relu = p->relu; a.out_f32 = 0; a.res_f32 = 0; a.out_lo = a.res_lo = 0;
    a.x2 = (const char*)x; a.pre_scale = pre_scale;
    a.in2_sn = (long long)hp2 * wp2 * p->cin2; a.in2_sh = p->stride2 * wp2 * p->cin2; a.in2_sw = p->stride2 * p->cin2;
    a.in2_off0 = (p->in2_halo * wp2 + p->in2_halo) * p->cin2;
    a.x2_bytes = 2LL * p->n * a.in2_sn;
    a.nsteps2 = p->cin2 / 64;
    // 32-bit DMA offsets within a tile's window of images, as in asl_gather_gemm_bf16
    {
        const long long imgs = 256 / ((long long)p->h * p->w) + 2;
        if (2 * (imgs * a.in_sn + a.in_off0 + p->cin) > 0x7fffffffLL || 2 * (imgs * a.in2_sn + a.in2_off0 + p->cin2) > 0x7fffffffLL)
            return ASL_EINVAL;
    }
    // the weight columns of the primary operand follow the second operand's: the kernel's column offset runs on
    return launch<128, 128, 64, 2, 4, false, true>(a, (hipStream_t)stream);
}
