// Two chained 1x1 convolutions in one launch (gfx950): the exit of a ResNet bottleneck and the entry of the next one,
//     y  = relu( scale_a * (W_a . t) + bias_a + res )        64 -> 256 channels, written to memory (the next block's
//                                                             shortcut and, here, the second GEMM's operand)
//     y2 = relu( scale_b * (W_b . bf16(y)) + bias_b )        256 -> 64 / 128 / 256 channels
// A workgroup owns 64 consecutive output pixels.  It computes their 256-channel y tile, stores it, keeps the bf16 values
// in LDS (row-major, 528-byte rows: 16 pixels x 16 bytes of one fragment read fall into distinct banks) and multiplies
// that tile by W_b without y coming back from memory: the pair moves 1 280 bytes per pixel instead of 1 792.  The second
// GEMM consumes exactly the bf16 values the first one stores and accumulates its K steps in the same order as the
// gather-GEMM does, so both outputs equal those of two asl_conv2d_nhwc_bf16 launches bit for bit.
//
// MFMA orientation, LDS-DMA staging, chunk swizzle and epilogue are those of gather_gemm.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asl_hip.h"
#include "common.h"
#include "epilogue.h"
#include "patch_args.h"

namespace asl_detail {

struct ChainArgs {
    const char* x;           // t: halo'd NHWC, 64 channels
    const char* w1;          // [>= 256 rows][64]
    const char* res;         // halo'd NHWC, 256 channels
    const char* w2;          // [>= N2 rows][256]
    char* y;                 // halo'd NHWC, 256 channels
    char* y2;                // halo'd NHWC, N2 channels
    const float *scale1, *bias1, *scale2, *bias2;
    int M;
    FastDiv dOW, dOHW;
    long long in_sn, res_sn, out_sn, out2_sn;
    int in_sh, in_sw, in_off0;
    int res_sh, res_sw, res_off0;
    int out_sh, out_sw, out_off0;
    int out2_sh, out2_sw, out2_off0;
    long long x_bytes, w1_bytes, w2_bytes;
};

__device__ __forceinline__ int swz64(int row) { return (row >> 1) & 7; }     // 128-byte rows (gather_gemm.hip)

constexpr int kChainBM = 64, kChainK1 = 64, kChainN1 = 256;
constexpr int kChainStage = (kChainBM + kChainN1) * 128;        // t tile + W_a, later W_b's K chunks: 40 KiB
constexpr int kChainORow = kChainN1 * 2 + 16;                   // bytes per pixel row of the y tile in LDS
constexpr int kChainLds = kChainStage + kChainBM * kChainORow;  // 74 752 B: two workgroups per CU

template <int N2>
__global__ __launch_bounds__(512) void chain1x1_kernel(const ChainArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = kChainBM, N1 = kChainN1, RB = 128;
    constexpr int X_BYTES = BM * RB;
    constexpr int OROW = kChainORow, O_OFF = kChainStage;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    int lbid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lbid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = lbid * BM;
    const int n0img = (int)fdiv(m0, a.dOHW);

    // ---- first GEMM: stage the t tile (one 1 KiB piece per wave) and all of W_a (four per wave) ------------------
    {
        const char* xbase = a.x + 2 * (long long)n0img * a.in_sn;
        const long long xrem = a.x_bytes - 2 * (long long)n0img * a.in_sn;
        __amdgpu_buffer_rsrc_t xrs =
            __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, (unsigned)(xrem > 0x7fffffffLL ? 0x7fffffffLL : xrem), 0x00020000);
        __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1, 0, (unsigned)a.w1_bytes, 0x00020000);
        {
            const int row = wave * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ swz64(row);
            const int m = m0 + row;
            unsigned voff = 0x80000000u;            // beyond num_records: the DMA writes zeros
            if (m < a.M) {
                const int n = (int)fdiv(m, a.dOHW), r1 = m - n * (int)a.dOHW.d;
                const int h = (int)fdiv(r1, a.dOW), w_ = r1 - h * (int)a.dOW.d;
                const long long e = (long long)(n - n0img) * a.in_sn + (long long)h * a.in_sh + (long long)w_ * a.in_sw + a.in_off0;
                voff = (unsigned)(2 * e + 16 * chunk);
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ swz64(row);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(smem + X_BYTES + (wave * 4 + i) * 1024),
                                                     16, (unsigned)(2 * row * kChainK1 + 16 * chunk), 0, 0, 0);
        }
    }
    // waves 2 (pixels) x 4 (channels): 32 x 64 per wave
    const int wm = wave >> 2, wn = wave & 3;
    OutGeom g1;
    g1.scale = a.scale1; g1.bias = a.bias1; g1.res = a.res; g1.y = a.y; g1.M = a.M; g1.N = N1;
    g1.dOW = a.dOW; g1.dOHW = a.dOHW; g1.dODHW = a.dOHW;
    g1.out_sn = a.out_sn; g1.out_sd = 0; g1.out_sh = a.out_sh; g1.out_sw = a.out_sw; g1.out_off0 = a.out_off0;
    g1.res_sn = a.res_sn; g1.res_sd = 0; g1.res_sh = a.res_sh; g1.res_sw = a.res_sw; g1.res_off0 = a.res_off0;
    g1.relu = 1; g1.out_f32 = 0; g1.res_f32 = 0; g1.out_lo = 0; g1.res_lo = 0;
    EpiState<2, 4> est1;
    epilogue_prepare<2, 4>(est1, g1, m0 + wm * 32, wn * 64, lane);      // residual loads go out under the staging

    f32x4 acc1[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        bf16x8 xf[2], wf[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 32 + i * 16 + fr;
            xf[i] = *(const bf16x8*)(smem + row * RB + (((kk * 4 + fq) ^ swz64(row)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wn * 64 + j * 16 + fr;
            wf[j] = *(const bf16x8*)(smem + X_BYTES + row * RB + (((kk * 4 + fq) ^ swz64(row)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc1[i][j], 0, 0, 0);
    }
    __syncthreads();                                    // every wave is done reading the stage: the epilogue overlays it
    static_assert(8 * kEpiWaveBytes(64) <= kChainStage, "first epilogue's staging does not fit the stage region");
    epilogue_store<2, 4, false, true>(acc1, smem + wave * kEpiWaveBytes(64), g1, est1, lane, smem + O_OFF, wm * 32, OROW);
    __syncthreads();                                    // the y tile is complete, the stage region is free again

    // ---- second GEMM: W_b through the stage region in rounds of 32 KiB (K chunks of 64), the y tile from LDS ----
    constexpr int WGN2 = N2 / 32, WGM2 = 8 / WGN2, WTM2 = BM / WGM2, FM2 = WTM2 / 16;
    constexpr int PPC = N2 / 8;                         // 1 KiB pieces per K chunk (N2 rows x 128 B)
    constexpr int CPRD = 32 / PPC, ROUNDS = 4 / CPRD;   // K chunks per round, rounds
    static_assert(FM2 >= 1 && CPRD >= 1 && ROUNDS * CPRD == 4, "second GEMM's tiling");
    const int wm2 = wave / WGN2, wn2 = wave % WGN2;
    OutGeom g2;
    g2.scale = a.scale2; g2.bias = a.bias2; g2.res = nullptr; g2.y = a.y2; g2.M = a.M; g2.N = N2;
    g2.dOW = a.dOW; g2.dOHW = a.dOHW; g2.dODHW = a.dOHW;
    g2.out_sn = a.out2_sn; g2.out_sd = 0; g2.out_sh = a.out2_sh; g2.out_sw = a.out2_sw; g2.out_off0 = a.out2_off0;
    g2.res_sn = 0; g2.res_sd = 0; g2.res_sh = 0; g2.res_sw = 0; g2.res_off0 = 0;
    g2.relu = 1; g2.out_f32 = 0; g2.res_f32 = 0; g2.out_lo = 0; g2.res_lo = 0;
    EpiState<FM2, 2> est2;
    epilogue_prepare<FM2, 2>(est2, g2, m0 + wm2 * WTM2, wn2 * 32, lane);
    __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, (unsigned)a.w2_bytes, 0x00020000);
    unsigned w2voff[4];
    int w2dst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = wave * 4 + i, c = p / PPC, rp = p - c * PPC;
        const int row = rp * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz64(row);
        w2voff[i] = (unsigned)(2 * (row * N1 + c * 64) + 16 * chunk);
        w2dst[i] = c * (N2 * RB) + rp * 1024;
    }
    f32x4 acc2[FM2][2];
#pragma unroll
    for (int i = 0; i < FM2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        if (rd > 0) __syncthreads();                    // the previous round's fragment reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rs, (__attribute__((address_space(3))) void*)(smem + w2dst[i]), 16, w2voff[i],
                                                     (unsigned)(rd * CPRD * 128), 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CPRD; ++c) {
            const int kc = rd * CPRD + c;               // K chunk: y channels 64 kc .. 64 kc + 63
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 xf[FM2], wf[2];
#pragma unroll
                for (int i = 0; i < FM2; ++i)
                    xf[i] = *(const bf16x8*)(smem + O_OFF + (wm2 * WTM2 + i * 16 + fr) * OROW + (kc * 64 + kk * 32 + fq * 8) * 2);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wn2 * 32 + j * 16 + fr;
                    wf[j] = *(const bf16x8*)(smem + c * (N2 * RB) + row * RB + (((kk * 4 + fq) ^ swz64(row)) << 4));
                }
#pragma unroll
                for (int i = 0; i < FM2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc2[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                    // W_b's chunks are read: the second epilogue overlays them
    static_assert(8 * kEpiWaveBytes(32) <= kChainStage, "second epilogue's staging does not fit the stage region");
    epilogue_store<FM2, 2>(acc2, smem + wave * kEpiWaveBytes(32), g2, est2, lane);
#endif
}

template <int N2>
static int chain_launch(const ChainArgs& a, hipStream_t stream) {
    static unsigned char lds_set[64];
    if (const int e = ensure_dynamic_lds((const void*)chain1x1_kernel<N2>, kChainLds, lds_set)) return e;
    const long long nblk = ((long long)a.M + kChainBM - 1) / kChainBM;
    if (nblk > 0x7fffffffLL) return ASL_EINVAL;
    hipLaunchKernelGGL(chain1x1_kernel<N2>, dim3((unsigned)nblk), dim3(512), kChainLds, stream, a);
    return (int)hipGetLastError();
}

}  // namespace asl_detail
using namespace asl_detail;

extern "C" int asl_conv1x1_chain_bf16(const asl_chain1x1_params* p, const void* t, const void* w_a, const float* scale_a,
                                      const float* bias_a, const void* res, void* y, const void* w_b, const float* scale_b,
                                      const float* bias_b, void* y2, asl_stream_t stream) {
    if (!p || !t || !w_a || !scale_a || !bias_a || !res || !y || !w_b || !scale_b || !bias_b || !y2) return ASL_EINVAL;
    if (p->n <= 0 || p->h <= 0 || p->w <= 0 || p->cin <= 0 || p->cmid <= 0 || p->cout <= 0) return ASL_EINVAL;
    if ((p->cin & 31) || (p->cmid & 31) || (p->cout & 7)) return ASL_EINVAL;
    if (p->in_halo < 0 || p->res_halo < 0 || p->mid_halo < 0 || p->out_halo < 0) return ASL_EINVAL;
    if (p->w_rows_a < p->cmid || p->w_rows_b < p->cout || (p->w_rows_a & 127) || (p->w_rows_b & 127)) return ASL_EINVAL;
    // the shapes the kernel is built for: a 64 -> 256 exit chained with a 256 -> 64 / 128 / 256 entry
    if (p->cin != kChainK1 || p->cmid != kChainN1 || (p->cout != 64 && p->cout != 128 && p->cout != 256)) return ASL_ENOTSUP;
    if (((uintptr_t)t & 15) || ((uintptr_t)w_a & 15) || ((uintptr_t)res & 15) || ((uintptr_t)y & 15) || ((uintptr_t)w_b & 15) ||
        ((uintptr_t)y2 & 15) || ((uintptr_t)scale_a & 15) || ((uintptr_t)bias_a & 15) || ((uintptr_t)scale_b & 15) ||
        ((uintptr_t)bias_b & 15))
        return ASL_EINVAL;
    const long long M = (long long)p->n * p->h * p->w;
    if (M > 0x7fffffffLL) return ASL_EINVAL;
    ChainArgs a;
    a.x = (const char*)t; a.w1 = (const char*)w_a; a.res = (const char*)res; a.w2 = (const char*)w_b;
    a.y = (char*)y; a.y2 = (char*)y2;
    a.scale1 = scale_a; a.bias1 = bias_a; a.scale2 = scale_b; a.bias2 = bias_b;
    a.M = (int)M;
    a.dOW = make_fastdiv((unsigned)p->w);
    a.dOHW = make_fastdiv((unsigned)(p->h * p->w));
    auto geom = [&](int halo, int c, long long& sn, int& sh, int& sw, int& off0) -> bool {
        const long long wp = p->w + 2LL * halo, hp = p->h + 2LL * halo;
        if (hp * wp * c > 0x3fffffffLL) return false;               // one image's bytes as a 32-bit DMA offset
        sn = hp * wp * c; sh = (int)(wp * c); sw = c; off0 = (int)((halo * wp + halo) * c);
        return true;
    };
    if (!geom(p->in_halo, p->cin, a.in_sn, a.in_sh, a.in_sw, a.in_off0) ||
        !geom(p->res_halo, p->cmid, a.res_sn, a.res_sh, a.res_sw, a.res_off0) ||
        !geom(p->mid_halo, p->cmid, a.out_sn, a.out_sh, a.out_sw, a.out_off0) ||
        !geom(p->out_halo, p->cout, a.out2_sn, a.out2_sh, a.out2_sw, a.out2_off0))
        return ASL_EINVAL;
    a.x_bytes = 2LL * p->n * a.in_sn;
    a.w1_bytes = 2LL * p->w_rows_a * p->cin;
    a.w2_bytes = 2LL * p->w_rows_b * p->cmid;
    if (a.w1_bytes > 0x7fffffffLL || a.w2_bytes > 0x7fffffffLL) return ASL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (p->cout == 64) return chain_launch<64>(a, s);
    if (p->cout == 128) return chain_launch<128>(a, s);
    return chain_launch<256>(a, s);
}
