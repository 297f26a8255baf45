// 3x3 / stride 1 / pad 1 convolution with the input patch resident in LDS (gfx950).
//
// The gather-GEMM fetches a pixel's 64-channel slice once per filter tap -- nine times per
// K sweep -- and at 128-wide tiles the L2->LDS fabric, not the MFMA pipe, sets its speed.
// Here a workgroup owns 256 consecutive output pixels (row-major over image, row, column;
// tiles may straddle rows and images), copies the contiguous range of halo'd input pixels
// those outputs touch into LDS ONCE per 64-channel chunk, and reads all nine taps' pixel
// fragments from that patch at a per-tap constant offset.  Only the weight tile of each
// (chunk, tap) step still streams through a double-buffered LDS-DMA stage, and the next
// chunk's patch is fetched in ninths alongside it.
//
// MFMA orientation, epilogue and weight packing are those of gather_gemm.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asl_hip.h"
#include "common.h"
#include "epilogue.h"
#include "patch_args.h"

namespace asl_detail {

constexpr int kPatchPixels = 448;             // LDS patch capacity in pixels (x 128 B per 64-channel chunk)
constexpr int kPatchBytes = kPatchPixels * 128;

__device__ __forceinline__ void wait_vm0_p() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// BN: output channels per workgroup; WGN: waves along channels (4 waves along pixels always)
template <int BN, int WGN>
__global__ __launch_bounds__(4 * WGN * 64) void conv3x3_patch_kernel(const PatchArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 256, WGM = 4, NW = WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int FM = WTM / 16, FN = WTN / 16;
    constexpr int W_BYTES = BN * 128;                 // one (chunk, tap) weight stage: BN rows x 64 channels
    constexpr int W_PW = W_BYTES / 1024 / NW;         // LDS-DMA instructions per wave per stage
    static_assert(W_PW >= 1 && W_PW * NW * 1024 == W_BYTES, "weight stage / wave mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int npbuf = a.nchunks > 1 ? 2 : 1;
    char* const wst = smem + npbuf * kPatchBytes;      // weight stages follow the patch buffers

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    int lbid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lbid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tn = lbid % a.tiles_n, tm = lbid / a.tiles_n;
    const int m0 = tm * BM;
    const int HW = a.H * a.W;

    // centre pixel (in halo'd, image-major pixel units) of output pixel m
    auto centre = [&](int m) -> long long {
        const int n = m / HW, r1 = m - n * HW;
        const int y = r1 / a.W, x = r1 - y * a.W;
        return ((long long)n * a.Hp + y + a.in_halo) * a.Wp + x + a.in_halo;
    };
    const int mlast = (m0 + BM - 1 < a.M ? m0 + BM - 1 : a.M - 1);
    const long long P0 = centre(m0) - a.Wp - 1;                       // first pixel of the patch
    const int NP = (int)(centre(mlast) + a.Wp + 1 - P0) + 1;          // pixels in the patch (<= kPatchPixels)
    const int NPI = (NP + 7) >> 3;                                    // 8 pixels per DMA instruction

    // ---- DMA descriptors ---------------------------------------------------------------------
    const long long pix_bytes = 2LL * a.C;
    const char* xbase = a.x + P0 * pix_bytes;
    long long xrem = a.x_bytes - P0 * pix_bytes;
    const unsigned xrec = (unsigned)(xrem > 0x7fffffffLL ? 0x7fffffffLL : xrem);
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, xrec, 0x00020000);
    const char* wbase = a.w + 2LL * (tn * BN) * a.wk;
    const unsigned wrec = (unsigned)(a.w_bytes - 2LL * (tn * BN) * a.wk);
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, wrec, 0x00020000);

    // patch DMA: instruction i covers patch pixels 8i..8i+7; slot (p, c') holds chunk c' ^ ((p>>1)&7)
    // lane -> pixel 8i + lane/8, so (p>>1)&7 = ((lane>>4) + 4i) & 7: two per-lane variants (i even / odd)
    // (fq + 4*(i&1)) & 7 == fq | ((i&1) << 2): the odd-i variant is the even one with byte-offset bit 6 flipped
    const unsigned pvoff0 = (unsigned)((lane >> 3) * pix_bytes + 16 * ((lane & 7) ^ (lane >> 4)));
    unsigned wvoff[W_PW];
#pragma unroll
    for (int i = 0; i < W_PW; ++i) {
        const int row = (wave * W_PW + i) * 8 + (lane >> 3);
        wvoff[i] = (unsigned)(2 * row * a.wk + 16 * ((lane & 7) ^ ((row >> 1) & 7)));
    }

    auto dma_patch = [&](int i, int chunk, int pbuf) {      // i, chunk, pbuf wave-uniform
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(i * 8 * pix_bytes + chunk * 128));
        char* dst = smem + pbuf * kPatchBytes + i * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)dst, 16,
                                                 pvoff0 ^ ((unsigned)(i & 1) << 6), so, 0, 0);
    };
    auto dma_weights = [&](int chunk, int tap, int wbuf) {
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(2 * (tap * a.C + chunk * 64)));
        char* dst = wst + wbuf * W_BYTES + wave * (W_PW * 1024);
#pragma unroll
        for (int i = 0; i < W_PW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                                     wvoff[i], so, 0, 0);
    };

    // ---- per-lane fragment geometry ---------------------------------------------------------
    const int fr = lane & 15, fq = lane >> 4;
    int pc[FM];                                   // patch-local centre pixel of this lane's pixel, per fragment
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        int m = m0 + wm * WTM + i * 16 + fr;
        m = m < a.M ? m : a.M - 1;                // tail slots compute a valid pixel again; never stored
        pc[i] = (int)(centre(m) - P0);
    }
    int wrd[FN][2];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int row = wn * WTN + j * 16 + fr;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) wrd[j][kk] = row * 128 + (((kk * 4 + fq) ^ ((row >> 1) & 7)) << 4);
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: whole patch of chunk 0 + weights of (chunk 0, tap 0) -----------------------
    for (int i = wave; i < NPI; i += NW) dma_patch(i, 0, 0);
    dma_weights(0, 0, 0);
    wait_vm0_p();
    __syncthreads();

    int step = 0;
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        const int pbuf = chunk & 1;
        const char* patch = smem + pbuf * kPatchBytes;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++step) {
            // prefetch: next step's weights, and a ninth of the next chunk's patch
            const int ntap = tap == 8 ? 0 : tap + 1, nchunk = tap == 8 ? chunk + 1 : chunk;
#if !defined(ASL_ABL_NODMA)
            if (nchunk < a.nchunks) dma_weights(nchunk, ntap, (step + 1) & 1);
            if (chunk + 1 < a.nchunks) {
                const int i = tap * NW + wave;
                if (i < NPI) dma_patch(i, chunk + 1, pbuf ^ 1);
                if (tap == 8)                                  // NW * 9 >= 56 instructions except for 4-wave blocks
                    for (int i2 = 9 * NW + wave; i2 < NPI; i2 += NW) dma_patch(i2, chunk + 1, pbuf ^ 1);
            }
#endif
            const int r = tap / 3, q = tap - 3 * r;
            const int toff = (r - 1) * a.Wp + (q - 1);
            const char* wb = wst + (step & 1) * W_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 xf[FM], wf[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int p = pc[i] + toff;
#if defined(ASL_ABL_NOLDS)
                    xf[i] = bf16x8{(short)p, 1, 2, 3, 4, 5, 6, 7};
#else
                    xf[i] = *(const bf16x8*)(patch + p * 128 + (((kk * 4 + fq) ^ ((p >> 1) & 7)) << 4));
#endif
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
#if defined(ASL_ABL_NOLDS)
                    wf[j] = bf16x8{(short)wrd[j][kk], 1, 2, 3, 4, 5, 6, 7};
#else
                    wf[j] = *(const bf16x8*)(wb + wrd[j][kk]);
#endif
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
#if defined(ASL_ABL_NOMFMA)
                        asm volatile("" ::"v"(wf[j]), "v"(xf[i]));
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
#endif
                    }
            }
            wait_vm0_p();
#if !defined(ASL_ABL_NOBAR)
            __syncthreads();
#endif
        }
    }
#if defined(ASL_ABL_NOEPI)
    if (a.M > 0) { if (acc[0][0][0] == 123.456f) a.y[0] = 1; return; }
#endif

    // ---- epilogue: the staging LDS overlays the patch buffer; the loop's last barrier frees it -----
    OutGeom g;
    g.scale = a.scale; g.bias = a.bias; g.res = a.res; g.y = a.y; g.M = a.M; g.N = a.N;
    g.OW = a.OW; g.OHW = a.OHW; g.ODHW = a.ODHW;
    g.out_sn = a.out_sn; g.out_sd = 0; g.out_sh = a.out_sh; g.out_sw = a.out_sw; g.out_off0 = a.out_off0;
    g.res_sn = a.res_sn; g.res_sd = 0; g.res_sh = a.res_sh; g.res_sw = a.res_sw; g.res_off0 = a.res_off0;
    g.relu = a.relu; g.out_f32 = 0;
    static_assert(NW * kEpiWaveBytes(WTN) <= kPatchBytes, "epilogue stage does not fit the patch buffer");
    epilogue_store<FM, FN>(acc, smem + wave * kEpiWaveBytes(WTN), g, m0 + wm * WTM, tn * BN + wn * WTN, lane);
#endif
}

template <int BN, int WGN>
int launch_patch(PatchArgs a, hipStream_t stream) {
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.N + BN - 1) / BN;
    const int lds = (a.nchunks > 1 ? 2 : 1) * kPatchBytes + 2 * BN * 128;
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_patch_kernel<BN, WGN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv3x3_patch_kernel<BN, WGN>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(4 * WGN * 64), lds,
                       stream, a);
    return (int)hipGetLastError();
}

// Can the patch kernel run this conv?  (3x3, stride 1, pad 1, C % 64 == 0, every tile's patch fits.)
bool patch_conv_fits(int n, int h, int w, int in_halo) {
    const int hp = h + 2 * in_halo, wp = w + 2 * in_halo;
    const long long M = (long long)n * h * w, hw = (long long)h * w;
    auto centre = [&](long long m) {
        const long long img = m / hw, r1 = m - img * hw;
        const long long y = r1 / w, x = r1 - y * w;
        return (img * hp + y + in_halo) * wp + x + in_halo;
    };
    const long long tiles = (M + 255) / 256;
    const long long scan = tiles < hw ? tiles : hw;      // the tile/image phase repeats within h*w tiles
    for (long long t = 0; t < scan; ++t) {
        const long long m0 = t * 256, ml = (m0 + 255 < M ? m0 + 255 : M - 1);
        if (centre(ml) - centre(m0) + 2 * wp + 3 > kPatchPixels) return false;
    }
    return true;
}

int launch_conv3x3_patch(const PatchArgs& a, hipStream_t stream) {
    if (a.N <= 64) return launch_patch<64, 1>(a, stream);
    return launch_patch<128, 2>(a, stream);
}

}  // namespace asl_detail
