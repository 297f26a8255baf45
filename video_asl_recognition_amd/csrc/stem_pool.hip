// ResNet stem fused with its max-pool for gfx950:
//   nn.Conv2d(3, 64, 7, stride 2, pad 3, bias=False) + BatchNorm2d(eval) + ReLU + nn.MaxPool2d(3, 2, 1)
// in one kernel, so the 112x112x64 stem activation (1.6 MB per frame, the largest tensor of the
// network) never goes to HBM: only the packed input (0.4 MB) is read and the pooled 56x56x64
// output (0.4 MB) written.
//
// A workgroup (4 waves) works on one tile of 8 x 7 POOLED pixels at a time; wave w owns pooled rows
// 2w, 2w+1 of the tile.  Those need conv rows 4w-1 .. 4w+3 (5 rows; one row is shared with the next
// wave and simply recomputed) x conv columns 2*px0-1 .. 2*px0+14 (16 = one MFMA pixel fragment), and
// for all waves together a 39-row x 40-pixel patch of the packed input, copied to LDS once per tile.
// Conv pixel (cy, cx), filter row r reads 8 consecutive packed pixels (32 elements = one K step) at
// patch[2*cy + r][2*cx ..], a 16-byte-aligned LDS address; the 64 x 224 weight matrix lives in
// registers (28 fragments).  BN + ReLU and the pool stay in registers: the vertical max runs over
// the wave's conv rows as they are produced, the horizontal max is two DPP row shifts across the 16
// lanes that hold a conv row.  Conv pixels outside the image count as 0, which is neutral for a max
// over ReLU outputs, and rounding to bf16 commutes with max, so the result equals
// maxpool(bf16(stem)) bit for bit.
//
// Persistent: a workgroup loads the weights once and walks tiles blockIdx.x, blockIdx.x + gridDim.x,
// ...; the next tile's patch is fetched (LDS-DMA, second patch buffer) while the MFMAs run.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asl_hip.h"
#include "common.h"
#include "patch_args.h"

namespace asl_detail {

struct StemPoolArgs {
    const char* x;       // packed [n][H+6][W+6][4] bf16
    const char* w;       // [128][224] bf16 (pack_stem layout)
    const float* scale;
    const float* bias;
    char* y;             // [n][PH+2*oh][PW+2*oh][64] bf16
    int n, H, W;         // input size
    int CH, CW;          // conv output size (H/2, W/2)
    int PH, PW;          // pooled size
    int out_halo;
    int tiles_x, tiles_y;
    long long x_bytes;
};

constexpr int kPTY = 8, kPTX = 7;       // pooled tile
constexpr int kPR = 2 * (2 * kPTY) + 7; // patch rows: conv rows 0..16 of the tile -> 2*16 + 7 = 39
[[maybe_unused]] constexpr int kPC = 40;                 // patch pixels per row: 2*15 + 8 = 38, padded to 40
constexpr int kPatchChunks = kPR * 20;  // 16-byte chunks (780)
constexpr int kPatchInstr = (kPatchChunks + 63) / 64;   // 13 wave-wide DMA instructions
[[maybe_unused]] constexpr int kPatchStride = kPatchInstr * 1024;

// shift a value down the 16-lane row by N lanes (lane i receives lane i+N; the top lanes keep their own)
template <int N>
__device__ __forceinline__ float row_shl(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v),
                                                                 0x100 + N, 0xf, 0xf, false));
}

// max of a lane's value and those of the next two lanes of its 16-lane row, for values >= 0 (lanes past the row's
// end read as 0): the shifts ride on the max instructions' own operand, two instructions in all
__device__ __forceinline__ float row_max3(float v) {
    float m;
    // (two wait states between a vector write of a register and its use as a DPP operand: the compiler does not
    // see into the asm, so they are spelled out)
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(m) : "v"(v));
    asm("v_max_f32_dpp %0, %1, %0 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(m) : "v"(v));
    return m;
}

__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const StemPoolArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) char lds[2 * kPatchStride];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Wp = a.W + 6, Hp = a.H + 6;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int ntiles = a.n * tiles_per_img;

    // patch chunk geometry of this lane for each of its DMA instructions (i = wave, wave+4, ...)
    constexpr int kDmaPerWave = (kPatchInstr + 3) / 4;
    int prow[kDmaPerWave], pch[kDmaPerWave];
#pragma unroll
    for (int k = 0; k < kDmaPerWave; ++k) {
        const int c = (wave + 4 * k) * 64 + lane;
        prow[k] = c / 20;
        pch[k] = c - prow[k] * 20;
    }
    auto dma_patch = [&](int tile, int buf) {
        const int img = tile / tiles_per_img, rem = tile - img * tiles_per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int iy0 = 2 * (2 * ty * kPTY - 1), ix0 = 2 * (2 * tx * kPTX - 1);    // patch origin, halo'd coords (may be -2)
        const long long org = (long long)img * Hp * Wp * 8 + ((long long)iy0 * Wp + ix0) * 8;
        // descriptor rebased near the patch so lane offsets stay small; offsets that would fall before the
        // tensor (first image, top rows) or past its end read as zero
        const long long shift = org > 0 ? (org & ~15LL) : 0;
        const long long rem_b = a.x_bytes - shift;
        const unsigned rec = (unsigned)(rem_b > 0x7fffffffLL ? 0x7fffffffLL : rem_b);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)(a.x + shift)));
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)(a.x + shift) >> 32));
        const char* base = (const char*)(((unsigned long long)hi << 32) | lo);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, __builtin_amdgcn_readfirstlane(rec),
                                                                      0x00020000);
        const long long rel = org - shift;                                         // <= 15, or negative for the first rows
#pragma unroll
        for (int k = 0; k < kDmaPerWave; ++k) {
            const int i = wave + 4 * k;
            if (i < kPatchInstr) {
                const long long off = rel + ((long long)prow[k] * Wp) * 8 + pch[k] * 16;
                const unsigned vo = (prow[k] < kPR && off >= 0) ? (unsigned)off : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rs, (__attribute__((address_space(3))) void*)(lds + buf * kPatchStride + i * 1024), 16, vo, 0, 0, 0);
            }
        }
    };

    // ---- weights -> registers: A fragments, wf[j][r] = rows 16j.., K step r (filter row r) ----------
    const int fr = lane & 15, fq = lane >> 4;
    bf16x8 wf[4][7];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 7; ++r) wf[j][r] = *(const bf16x8*)(a.w + ((j * 16 + fr) * 224 + r * 32 + fq * 8) * 2);
    f32x4 sc[4], bi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sc[j] = a.scale ? *(const f32x4*)(a.scale + j * 16 + fq * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
        bi[j] = a.bias ? *(const f32x4*)(a.bias + j * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int owp = a.PW + 2 * a.out_halo, ohp = a.PH + 2 * a.out_halo;

    int buf = 0;
    if ((int)blockIdx.x < ntiles) dma_patch(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // patch(tile) has landed (own DMA waited, then the barrier covers the other waves'); the barrier
        // also ends the previous tile's reads of the other patch buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) dma_patch(tile + gridDim.x, buf ^ 1);
        const int img = tile / tiles_per_img, rem = tile - img * tiles_per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int py0 = ty * kPTY, px0 = tx * kPTX;        // first pooled pixel of the tile
        const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;    // first conv pixel of the tile (may be -1)
        const char* patch = lds + buf * kPatchStride;
        const int cx = cx0 + fr;
        const bool col_in = (cx >= 0 && cx < a.CW) || fr == 15;   // lane 15's column feeds no pooled pixel of this tile

        f32x4 pa[4], pb[4];                                 // running vertical max of pooled rows 2w, 2w+1
#pragma unroll
        for (int j = 0; j < 4; ++j) pa[j] = pb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 5; ++c) {                       // conv rows 4w + c of the tile
            const int cr = 4 * wave + c;
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // lane: conv column fr, K chunk fq = packed pixels 2*fq, 2*fq+1 of the 8-pixel window
            const char* pbp = patch + ((2 * cr) * kPC + 2 * fr + 2 * fq) * 8;
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const bf16x8 xf = *(const bf16x8*)(pbp + r * kPC * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][r], xf, acc[j], 0, 0, 0);
            }
            const int cy = cy0 + cr;
            const bool inside = col_in && cy >= 0 && cy < a.CH;
            // The running maxima start at 0, so max(p, v) already is max(p, relu(v)).  Only tiles on the image's top
            // and left edge hold conv pixels outside it (whose value must not count): every other tile takes the
            // branch-free path with no select per value.
            if (__builtin_amdgcn_ballot_w64(!inside) == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = acc[j] * sc[j] + bi[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (c <= 2) pa[j][e] = fmaxf(pa[j][e], v[e]);
                        if (c >= 2) pb[j][e] = fmaxf(pb[j][e], v[e]);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = inside ? acc[j][e] * sc[j][e] + bi[j][e] : 0.f;
                        if (c <= 2) pa[j][e] = fmaxf(pa[j][e], v);
                        if (c >= 2) pb[j][e] = fmaxf(pb[j][e], v);
                    }
            }
        }
        // ---- horizontal max over conv columns 2lx, 2lx+1, 2lx+2 -> even lanes fr = 2lx hold pooled column lx --
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int py = py0 + 2 * wave + half, px = px0 + (fr >> 1);
            const bool store = !(fr & 1) && fr < 2 * kPTX && py < a.PH && px < a.PW;
            char* dst = a.y + ((((long long)img * ohp + py + a.out_halo) * owp + px + a.out_halo) * 64 + fq * 4) * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = half ? pb[j][e] : pa[j][e];
                    o[e] = f32_to_bf16(row_max3(v));
                }
                if (store) *(bf16x4*)(dst + j * 32) = o;
            }
        }
        buf ^= 1;
    }
#endif
}

}  // namespace asl_detail
using namespace asl_detail;

extern "C" int asl_stem_pool_bf16(const asl_stem_pool_params* p, const void* x, const void* w, const float* scale,
                                  const float* bias, void* y, asl_stream_t stream) {
    if (!p || !x || !w || !y) return ASL_EINVAL;
    if (p->n <= 0 || p->h <= 0 || p->w <= 0 || (p->h & 3) || (p->w & 3) || p->out_halo < 0) return ASL_EINVAL;
    if (p->cout != 64 || p->w_rows < 64) return ASL_ENOTSUP;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)scale & 15) ||
        ((uintptr_t)bias & 15))
        return ASL_EINVAL;
    StemPoolArgs a;
    a.x = (const char*)x; a.w = (const char*)w; a.scale = scale; a.bias = bias; a.y = (char*)y;
    a.n = p->n; a.H = p->h; a.W = p->w; a.CH = p->h / 2; a.CW = p->w / 2; a.PH = p->h / 4; a.PW = p->w / 4;
    a.out_halo = p->out_halo;
    a.tiles_x = (a.PW + kPTX - 1) / kPTX; a.tiles_y = (a.PH + kPTY - 1) / kPTY;
    a.x_bytes = (long long)p->n * (p->h + 6) * (p->w + 6) * 8;
    const long long blocks = (long long)p->n * a.tiles_x * a.tiles_y;
    if (blocks > 0x7fffffffLL) return ASL_EINVAL;
    // persistent grid: two 4-wave workgroups per CU fit the register budget (weights live in VGPRs)
    const long long slots = 2LL * device_cus();
    const unsigned grid = (unsigned)(blocks < slots ? blocks : slots);
    hipLaunchKernelGGL(stem_pool_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
