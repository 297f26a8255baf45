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
#include <stdlib.h>

#include <type_traits>

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

// ---------------------------------------------------------------------------------------------------
// Column-strip form of the same fusion (the default).  The tile kernel above recomputes one conv row in
// five (every wave needs the row above its first pooled row) and fetches a 39-row patch per 32 conv rows.
// Here a WAVE is the unit: it owns a strip of 16 conv columns (7 pooled columns) and walks it top to bottom,
// one conv row per step, so every conv row is computed once; the vertical 3-max is carried in registers from
// row to row (pooled row py = conv rows 2py-1, 2py, 2py+1: the last of them also opens pooled row py+1).
// Input rows stream through a private LDS ring of 16 row pairs (1 KiB slots: 2 rows x 320 B + padding) per wave,
// one LDS-DMA instruction per step, issued kDist steps ahead and retired by counted s_waitcnt -- the waves of a
// workgroup share nothing, so the kernel has no barrier at all.  A task is (image, half of the pooled rows,
// strip); the four waves of a workgroup take four neighbouring strips, whose column overlap (40 packed pixels
// fetched per 28 new) then hits in the CU's vector cache.
//
// Vector work per conv row is what bounds this kernel next to its 28 MFMAs, so it is cut to the minimum:
//  * BatchNorm runs on POOLED values only.  s * a + b is monotonic in a once s >= 0, and a channel with s < 0 is
//    turned into one with s > 0 by negating its weight row (exact in bf16 and through the fp32 accumulation), done
//    on the fragments as they are loaded; then max commutes with BatchNorm bit for bit.
//  * Conv pixels outside the image are set to -inf once per pooled row (columns) or skipped (row -1).
//  * Weight rows are read in a permuted order so that a lane ends up with 16 consecutive output channels and
//    stores 32 B per pooled pixel straight from registers, through a buffer descriptor (idle lanes need no branch).
//  * The next row's first six fragments are read before the current row's MFMAs are issued.
[[maybe_unused]] constexpr int kSlots = 16;               // ring slots (row pairs) per wave
[[maybe_unused]] constexpr int kDist = 8;                 // DMA prefetch distance in conv rows (even); 4 + kDist <= kSlots
[[maybe_unused]] constexpr int kSlot = 1024;              // ring slot: one row pair (640 B) + the idle lanes' zeros
[[maybe_unused]] constexpr int kRingBytes = kSlots * kSlot;
static_assert(kDist % 2 == 0 && 4 + kDist <= kSlots, "prefetch distance");

// PLANAR form: the input is the clip tensor itself, bf16 [n][3][H][W] -- no packed copy, no halo (what lies outside the
// image is fetched with out-of-range offsets and arrives as zeros).  A row pair is 2 rows x 3 planes x 38 pixels:
// six 76-byte segments = 114 dwords = two dword-wide LDS-DMA instructions into a 512-byte slot.  K is ordered
// (filter row, channel, 8 taps): a lane's eight K elements are eight CONSECUTIVE pixels of one plane, so fragments
// come straight out of the planar rows, and the 21 (row, channel) groups fill 6 MFMAs per channel fragment where the
// packed form needs 7.  The eight taps of a group start one pixel left of the filter (an even pixel, so every
// fragment is dword-aligned); that first tap has weight zero.
[[maybe_unused]] constexpr int kSegDw = 19;               // dwords per segment: (2 * 15 + 8) pixels
[[maybe_unused]] constexpr int kPSlot = 512;              // planar ring slot
[[maybe_unused]] constexpr int kPRingBytes = kSlots * kPSlot;

struct StemStripArgs {
    const char* x;       // packed [n][H+6][W+6][4] bf16; PLANAR: [n][3][H][W] bf16
    const char* w;       // [128][224] bf16 (pack_stem layout); PLANAR: [128][192] (pack_stem_planar layout)
    const float* scale;
    const float* bias;
    char* y;             // [n][PH+2*oh][PW+2*oh][64] bf16
    int n, H, W, CH, CW, PH, PW, out_halo;
    int strips, strip_groups, segs, seg_rows;   // strips of 7 pooled columns in groups of 4; row segments
    long long x_bytes;
    int ngroups;         // n * segs * strip_groups workgroup tasks
};

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Horizontal 3-max (lane i <- max of lanes i, i+1, i+2 of its 16-lane row) of eight values in one block: the two
// wait states a DPP operand needs after a vector write are paid once per block, not per value.  Lanes past the row's
// end read 0; they feed no stored pixel.
__device__ __forceinline__ void row_max3_x8(const float (&v)[8], float (&m)[8]) {
#define D1(i, j) "v_max_f32_dpp %" #i ", %" #j ", %" #j " row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define D2(i, j) "v_max_f32_dpp %" #i ", %" #j ", %" #i " row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm("s_nop 1\n\t" D1(0, 8) D1(1, 9) D1(2, 10) D1(3, 11) D1(4, 12) D1(5, 13) D1(6, 14) D1(7, 15)
        D2(0, 8) D2(1, 9) D2(2, 10) D2(3, 11) D2(4, 12) D2(5, 13) D2(6, 14) D2(7, 15)
        : "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3]), "=&v"(m[4]), "=&v"(m[5]), "=&v"(m[6]), "=&v"(m[7])
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
#undef D1
#undef D2
}

template <class T, int kOff>
__device__ __forceinline__ T lds_read(unsigned ad) {
    T v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(ad), "n"(kOff) : "memory");
    return v;
}

typedef __attribute__((ext_vector_type(2))) int i32x2;
// Four consecutive dwords from the dword-aligned LDS address ad + kImm (two ds_read2_b32: ds_read_b128 wants 16-byte
// alignment, and the offset fields of ds_read2 reach 1 KiB only).  The addition sits inside the statement so that the
// compiler cannot hoist one address register per (ring slot, fragment) out of the unrolled loop.
template <int kImm>
__device__ __forceinline__ bf16x8 lds_read_4dw(unsigned ad) {
    i32x2 lo, hi;
    unsigned t;
    asm volatile("v_add_u32 %2, %4, %3\n\tds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3"
                 : "=v"(lo), "=v"(hi), "=&v"(t) : "v"(ad), "n"(kImm) : "memory");
    return __builtin_bit_cast(bf16x8, i32x4{lo[0], lo[1], hi[0], hi[1]});
}

template <bool PLANAR>
__global__ __launch_bounds__(256, 2) void stem_strip_kernel(const StemStripArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NQ = PLANAR ? 6 : 7;                    // K steps (MFMAs per channel fragment) of a conv row
    constexpr int WK = NQ * 32;                           // weight row length
    constexpr int SLOT = PLANAR ? kPSlot : kSlot;
    __shared__ __attribute__((aligned(16))) char lds[4 * kSlots * SLOT];
    __shared__ __attribute__((aligned(16))) float bnt[128];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int Wp = a.W + 6;
    char* const ring = lds + wave * (kSlots * SLOT);
    const unsigned ring_ad = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring;
    constexpr float kNegInf = -__builtin_inff();

    // ---- weights -> registers.  Fragment j, row rho holds channel (rho >> 2) * 16 + j * 4 + (rho & 3), so that the
    // lane's 16 results (j, e) are channels fq * 16 + j * 4 + e: 16 consecutive ones.  Rows of channels with a
    // negative BatchNorm scale are negated (sign bit of each bf16), and |scale| is used below.
    bf16x8 wf[4][NQ];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = (fr >> 2) * 16 + j * 4 + (fr & 3);
        const int flip = (a.scale && a.scale[ch] < 0.f) ? (int)0x80008000 : 0;
#pragma unroll
        for (int r = 0; r < NQ; ++r) {
            i32x4 t = *(const i32x4*)(a.w + (ch * WK + r * 32 + fq * 8) * 2);
            t ^= flip;
            wf[j][r] = __builtin_bit_cast(bf16x8, t);
        }
    }
    // |scale| and bias are needed once per pooled row only: they wait in LDS (a lane reads its 16 channels from there)
    // and leave the registers to the fragments
    if (tid < 64) {
        bnt[tid] = a.scale ? __builtin_fabsf(a.scale[tid]) : 1.f;
        bnt[64 + tid] = a.bias ? a.bias[tid] : 0.f;
    }
    __syncthreads();
    const unsigned bn_ad = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)(char*)bnt + fq * 64;
    const int owp = a.PW + 2 * a.out_halo, ohp = a.PH + 2 * a.out_halo;
    // lane's place in a DMA instruction: rows 0 / 1 of the pair, 16-byte chunk 0..19; lanes 40..63 stay out of range
    // (the DMA then writes zeros into the slot's padding) so the instruction needs no exec mask
    const int drow = lane >= 20 ? 1 : 0, dch = lane - 20 * drow;
    const bool dma_lane = lane < 40;
    const unsigned rd = ring_ad + (2 * fr + 2 * fq) * 8;  // LDS address of the lane's fragment in slot 0, row 0
    constexpr unsigned kOob = 0x80000000u;
    // PLANAR: lane's dwords of a row pair (instruction i: dword 64 i + lane of the 114) and its six fragment addresses.
    // Fragment q holds K group g = 4q + fq = (filter row g / 3, channel g % 3); groups 21..23 have zero weights and
    // re-read group 20.  Filter row r lies in pair (step + (r >> 1)), row r & 1 of it; within MFMAs 1 and 4 the lanes
    // fq >= 2 sit one pair further on than the lanes fq < 2 (+ kPSlot, or - 15 slots where the ring wraps).
    int pl_base[2], pl_x[2];
    unsigned pa[6], pa_wrap[2];
    constexpr int kNoSrc = (int)0x80000000;
    const bool second_row0 = lane >= 3 * kSegDw;          // instruction 0: dwords 57..63 already belong to the pair's second row
    if constexpr (PLANAR) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int d = 64 * i + lane, seg = d / kSegDw, dw = d - seg * kSegDw;
            // plane, row of the pair (input row = 2 * conv row - 3 + it), pixel pair; the strip's origin is added per task
            pl_base[i] = d < 6 * kSegDw ? ((seg % 3) * a.H * a.W + (seg / 3 - 3) * a.W + 2 * dw) * 2 : kNoSrc;
            pl_x[i] = 2 * dw;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            constexpr int kbase[6] = {0, 0, 1, 2, 2, 3};
            int g = 4 * q + fq;
            g = g > 20 ? 20 : g;
            const int r = g / 3, ch = g - 3 * r;
            pa[q] = ring_ad + ((r >> 1) - kbase[q]) * kPSlot + ((r & 1) * 3 + ch) * (kSegDw * 4) + 4 * fr;
        }
        pa_wrap[0] = pa[1] - (fq >= 2 ? kSlots * kPSlot : 0);
        pa_wrap[1] = pa[4] - (fq >= 2 ? kSlots * kPSlot : 0);
    }

    for (int grp = blockIdx.x; grp < a.ngroups; grp += gridDim.x) {
        const int per_img = a.segs * a.strip_groups;
        const int img = grp / per_img, rem = grp - img * per_img;
        const int seg = rem / a.strip_groups, sg = rem - seg * a.strip_groups;
        const int strip = sg * 4 + wave;
        if (strip >= a.strips) continue;                  // (wave-uniform: the waves share no barrier)
        const int py0 = seg * a.seg_rows;
        const int npy = (py0 + a.seg_rows <= a.PH ? a.seg_rows : a.PH - py0);     // pooled rows of this task
        const int cy0 = 2 * py0 - 1, cx0 = 14 * strip - 1;                        // first conv row / column (may be -1)
        const int last_pair = 2 * npy + 3;                // conv row 2 * npy (the last) reads pairs up to this one
        // Row pair k = halo'd input rows 2 * (cy0 + k), +1; columns from halo'd pixel 2 * cx0.  Descriptor rebased to
        // the task so lane offsets stay small; what falls before the tensor (cy0 = -1, cx0 = -1 on the first image)
        // reads as zero, and those conv pixels are masked anyway.
        // (PLANAR: unpadded rows 2 * (cy0 + k) - 3, -2 of the three planes, 38 pixels from 2 * cx0 - 4; descriptor on
        // the image; rows and pixel pairs outside it get out-of-range offsets.)
        const long long org = ((long long)img * (a.H + 6) * Wp + (long long)(2 * cy0) * Wp + 2 * cx0) * 8;
        const long long shift = PLANAR ? (long long)img * 3 * a.H * a.W * 2 : (org > 0 ? org : 0);   // multiples of 16
        const long long rem_b = a.x_bytes - shift;
        const unsigned rec = PLANAR ? (unsigned)(3 * a.H * a.W * 2) : (unsigned)(rem_b > 0x7fffffffLL ? 0x7fffffffLL : rem_b);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + shift), 0, rec, 0x00020000);
        const int rel = (int)(org - shift);               // 0, or negative on the first rows of the first image
        const int lane_src = dma_lane ? rel + drow * Wp * 8 + dch * 16 : -1;
        int psrc[2];
        if constexpr (PLANAR) {
            const int xs = 2 * cx0 - 4;                    // even: a pixel pair is inside the image or outside it as a whole
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = (pl_base[i] != kNoSrc) & ((unsigned)(xs + pl_x[i]) < (unsigned)a.W);
                psrc[i] = ok ? pl_base[i] + xs * 2 : kNoSrc;
            }
        }
        auto dma_pair = [&](int k) {
            // pairs past the task's last one move no data (out-of-range offset: the DMA writes zeros), so the count
            // of instructions in flight is the same on every step
            if constexpr (PLANAR) {
                const int y0 = 2 * (cy0 + k) - 3;          // first row of the pair
                const bool k_ok = k <= last_pair;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int y = y0 + ((i == 1) | (int)second_row0);
                    const bool ok = (psrc[i] != kNoSrc) & k_ok & ((unsigned)y < (unsigned)a.H);
                    const unsigned vo = ok ? (unsigned)(psrc[i] + 2 * (cy0 + k) * a.W * 2) : kOob;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rs, (__attribute__((address_space(3))) void*)(ring + (k & (kSlots - 1)) * kPSlot + i * 256), 4, vo, 0, 0, 0);
                    asm volatile("" ::: "memory");
                }
            } else {
                const int off = lane_src + k * (2 * Wp * 8);
                const unsigned vo = (k <= last_pair && lane_src != -1 && off >= 0) ? (unsigned)off : kOob;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rs, (__attribute__((address_space(3))) void*)(ring + (k & (kSlots - 1)) * kSlot), 16, vo, 0, 0, 0);
                asm volatile("" ::: "memory");            // the counted waits need DMAs and stores in program order
            }
        };
        // output through a descriptor on this image: lanes that store nothing get an out-of-range offset
        const long long img_bytes = (long long)ohp * owp * 128;
        __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + img * img_bytes), 0, (unsigned)img_bytes,
                                                                      0x00020000);
        const int px = 7 * strip + (fr >> 1);
        const bool store = !(fr & 1) && fr < 14 && px < a.PW;
        const unsigned out0 = store ? (unsigned)((((py0 + a.out_halo) * owp + px + a.out_halo) * 64 + fq * 16) * 2) : kOob;
        const int cx = cx0 + fr;
        const bool col_in = cx >= 0 && cx < a.CW;
        const bool edge_cols = cx0 < 0 || cx0 + 14 >= a.CW;      // lanes 0..14 feed pooled pixels; lane 15 none

        // Fragment r of the conv row in ring position C (= step & (kSlots - 1)): input row 2C + r = pair C + (r >> 1), row
        // r & 1 of it.  The reads are inline assembly on purpose: the compiler orders a visible LDS read behind EVERY
        // LDS-DMA in flight (s_waitcnt vmcnt(0)), which would undo the prefetch distance.
        auto read6 = [&](auto Cc, bf16x8 (&x)[NQ]) {
            constexpr int C = decltype(Cc)::value;
            if constexpr (PLANAR) {
                // fragments 0..3 lie in pairs C .. C + 2 (MFMA 1: the lanes fq >= 2 one pair beyond the others)
                x[0] = lds_read_4dw<((C + 0) & (kSlots - 1)) * kPSlot>(pa[0]);
                x[1] = lds_read_4dw<((C + 0) & (kSlots - 1)) * kPSlot>(((C + 0) & (kSlots - 1)) == kSlots - 1 ? pa_wrap[0] : pa[1]);
                x[2] = lds_read_4dw<((C + 1) & (kSlots - 1)) * kPSlot>(pa[2]);
                x[3] = lds_read_4dw<((C + 2) & (kSlots - 1)) * kPSlot>(pa[3]);
            } else {
                x[0] = lds_read<bf16x8, ((C + 0) & (kSlots - 1)) * kSlot>(rd);
                x[1] = lds_read<bf16x8, ((C + 0) & (kSlots - 1)) * kSlot + 320>(rd);
                x[2] = lds_read<bf16x8, ((C + 1) & (kSlots - 1)) * kSlot>(rd);
                x[3] = lds_read<bf16x8, ((C + 1) & (kSlots - 1)) * kSlot + 320>(rd);
                x[4] = lds_read<bf16x8, ((C + 2) & (kSlots - 1)) * kSlot>(rd);
                x[5] = lds_read<bf16x8, ((C + 2) & (kSlots - 1)) * kSlot + 320>(rd);
            }
        };
        auto read_last = [&](auto Cc, bf16x8 (&x)[NQ]) {
            constexpr int C = decltype(Cc)::value;
            if constexpr (PLANAR) {
                // fragments 4, 5 touch pair C + 3, the one that lands last
                x[4] = lds_read_4dw<((C + 2) & (kSlots - 1)) * kPSlot>(((C + 2) & (kSlots - 1)) == kSlots - 1 ? pa_wrap[1] : pa[4]);
                x[5] = lds_read_4dw<((C + 3) & (kSlots - 1)) * kPSlot>(pa[5]);
            } else {
                x[6] = lds_read<bf16x8, ((C + 3) & (kSlots - 1)) * kSlot>(rd);
            }
        };
        // the fragments are in their registers once at most the early reads issued since are still out (LDS operations
        // complete in order): six ds_read_b128, or four fragments of two ds_read2_b32 each
        auto frags_ready = [&](bf16x8 (&x)[NQ]) {
            if constexpr (PLANAR)
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]));
            else
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[NQ - 1]));
        };
        auto mfma_row = [&](const bf16x8 (&xf)[NQ], f32x4 (&acc)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < NQ; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][r], xf[r], acc[j], 0, 0, 0);
        };
        // pooled row from the vertical maxima of raw accumulators: mask columns, horizontal 3-max, BatchNorm, ReLU, store
        auto emit = [&](int p, const f32x4 (&m)[4]) {
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
            i32x4 o[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                // |scale| and bias of this half's 8 channels (read per half: the registers are scarce here)
                f32x4 s0, s1, b0, b1;
                if (half == 0) {
                    s0 = lds_read<f32x4, 0>(bn_ad);   s1 = lds_read<f32x4, 16>(bn_ad);
                    b0 = lds_read<f32x4, 256>(bn_ad); b1 = lds_read<f32x4, 272>(bn_ad);
                } else {
                    s0 = lds_read<f32x4, 32>(bn_ad);  s1 = lds_read<f32x4, 48>(bn_ad);
                    b0 = lds_read<f32x4, 288>(bn_ad); b1 = lds_read<f32x4, 304>(bn_ad);
                }
                float v[8], h[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = m[half * 2 + (i >> 2)][i & 3];
                if (edge_cols) {                           // (wave-uniform) columns outside the image count for nothing
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = col_in ? v[i] : kNegInf;
                }
                row_max3_x8(v, h);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1), "+v"(b0), "+v"(b1));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int q = 2 * i;                   // values q, q + 1 -> one packed pair
                    const float t0 = h[q] * (q < 4 ? s0[q & 3] : s1[q & 3]) + (q < 4 ? b0[q & 3] : b1[q & 3]);
                    const float t1 = h[q + 1] * (q < 4 ? s0[(q + 1) & 3] : s1[(q + 1) & 3]) + (q < 4 ? b0[(q + 1) & 3] : b1[(q + 1) & 3]);
                    unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{t0, t1}, bf16x2_t));
                    // as 16-bit integers, negative bf16 values are negative: ReLU of the rounded pair in one instruction
                    asm("v_pk_max_i16 %0, %1, 0" : "=v"(pk) : "v"(pk));
                    o[half][i] = (int)pk;
                }
            }
            const unsigned off = out0 == kOob ? kOob : out0 + (unsigned)(p * owp * 128);
            __builtin_amdgcn_raw_buffer_store_b128(o[0], ys, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(o[1], ys, off == kOob ? kOob : off + 16, 0, 0);
            asm volatile("" ::: "memory");
        };

        // One step = one conv row c: issue the next row's first six fragment reads, multiply this row, then -- pair
        // c + 4 (the next row's last fragment) being the only thing the next row still waits for -- retire its DMA,
        // issue the DMA kDist pairs further on, and read that last fragment.  Vector-memory operations complete in
        // order, so "pair c + 4 has landed" = at most kWait operations issued after its DMA are still out: the kDist - 1
        // DMAs of the steps since and two stores per pooled row closed in those kDist steps (every other one).  The
        // prologue issues out-of-range (dropped) stores where the steps before the first would have stored, so the count
        // holds from step 0 on.
        bf16x8 xa[NQ], xb[NQ];
        f32x4 ve[4], vo_[4], m[4];
#define STEM_STEP(CM, c, CUR, NXT, ACC)                            \
    {                                                              \
        read6(std::integral_constant<int, ((CM) + 1) & (kSlots - 1)>{}, NXT);     \
        frags_ready(CUR);                                          \
        mfma_row(CUR, ACC);                                        \
        wait_vm<kWait>();                                          \
        dma_pair((c) + 4 + kDist);                                 \
        read_last(std::integral_constant<int, ((CM) + 1) & (kSlots - 1)>{}, NXT); \
    }
        // the previous task's fragment reads are behind us (their MFMAs were issued); its trailing DMAs are older than
        // the ones issued now and land first
        // (PLANAR: two DMA instructions per step)
        constexpr int DPS = PLANAR ? 2 : 1;
        constexpr int kWait = (kDist - 1) * DPS + kDist;
        auto placeholder_stores = [&]() {
            __builtin_amdgcn_raw_buffer_store_b128(i32x4{0, 0, 0, 0}, ys, kOob, 0, 0);
            asm volatile("" ::: "memory");                // (two instructions: identical neighbours would be merged)
            __builtin_amdgcn_raw_buffer_store_b128(i32x4{0, 0, 0, 0}, ys, kOob + 16, 0, 0);
            asm volatile("" ::: "memory");
        };
#pragma unroll
        for (int k = 0; k < 4; ++k) dma_pair(k);
#pragma unroll
        for (int k = 0; k < kDist; ++k) {                 // what steps -kDist .. -1 would have issued
            dma_pair(4 + k);
            if (!(k & 1)) placeholder_stores();
        }
        wait_vm<kDist * DPS + kDist>();                   // pairs 0..3
        read6(std::integral_constant<int, 0>{}, xa);
        read_last(std::integral_constant<int, 0>{}, xa);
        // conv row 0 = row 2 * py0 - 1: outside the image for the first segment
        STEM_STEP(0, 0, xa, xb, ve)
        placeholder_stores();
        if (cy0 < 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ve[j] = f32x4{kNegInf, kNegInf, kNegInf, kNegInf};
        }
        // pooled row p: conv rows 2p + 1 (odd step) and 2p + 2 (even step, which closes it).  kSlots / 2 pooled rows per
        // trip keep the ring slot of every read a compile-time constant.
#define STEM_POOLED(U)                                             \
    {                                                              \
        const int p = p0 + (U);                                    \
        if (p >= npy) break;                                       \
        const int c = 2 * p + 1;                                   \
        STEM_STEP(2 * (U) + 1, c, xb, xa, vo_)                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)              \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) m[j][e] = fmaxf(ve[j][e], vo_[j][e]); \
        STEM_STEP(2 * (U) + 2, c + 1, xa, xb, ve)                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)              \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) m[j][e] = fmaxf(m[j][e], ve[j][e]); \
        emit(p, m);                                                \
    }
        static_assert(kSlots == 16, "eight pooled rows per trip below");
        for (int p0 = 0; p0 < npy; p0 += kSlots / 2) {
            STEM_POOLED(0) STEM_POOLED(1) STEM_POOLED(2) STEM_POOLED(3)
            STEM_POOLED(4) STEM_POOLED(5) STEM_POOLED(6) STEM_POOLED(7)
        }
#undef STEM_POOLED
#undef STEM_STEP
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
    if (p->planar && (p->planar != 1 || (long long)p->h * p->w * 6 > 0x7fffffffLL)) return ASL_EINVAL;
    static const bool use_tiles = getenv("ASL_STEM_TILES") && atoi(getenv("ASL_STEM_TILES"));   // A/B: the tile kernel
    if (!use_tiles || p->planar) {
        StemStripArgs s;
        s.x = (const char*)x; s.w = (const char*)w; s.scale = scale; s.bias = bias; s.y = (char*)y;
        s.n = p->n; s.H = p->h; s.W = p->w; s.CH = p->h / 2; s.CW = p->w / 2; s.PH = p->h / 4; s.PW = p->w / 4;
        s.out_halo = p->out_halo;
        s.strips = (s.PW + 6) / 7;
        s.strip_groups = (s.strips + 3) / 4;
        s.segs = s.PH >= 16 ? 2 : 1;                       // two tasks per strip: 10 240 of them at 640 frames of 224 x 224
        s.seg_rows = (s.PH + s.segs - 1) / s.segs;
        s.x_bytes = (long long)p->n * (p->h + 6) * (p->w + 6) * 8;
        const long long groups = (long long)p->n * s.segs * s.strip_groups;
        if (groups > 0x7fffffffLL) return ASL_EINVAL;
        if ((long long)(s.PH + 2 * s.out_halo) * (s.PW + 2 * s.out_halo) * 128 > 0x7fffffffLL) return ASL_EINVAL;
        s.ngroups = (int)groups;
        const long long slots = 2LL * device_cus();
        const dim3 grid((unsigned)(groups < slots ? groups : slots));
        if (p->planar) hipLaunchKernelGGL(stem_strip_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, s);
        else hipLaunchKernelGGL(stem_strip_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, s);
        return (int)hipGetLastError();
    }
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
