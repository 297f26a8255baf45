// HBM-bound layout, preprocess, pooling and head kernels for gfx950.  Every kernel moves
// 8-16 bytes per lane with consecutive lanes on consecutive addresses.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asl_hip.h"
#include "common.h"

namespace asl_detail {

// ---------------------------------------------------------------------------------------
// clips [n][3][s][s] -> packed [n][s+2*halo][s+2*halo][4] bf16; one lane per 2 output pixels (16 B)
// ---------------------------------------------------------------------------------------
// T > 0: the frames are clips of T frames and y is [clips][T + 2*hd][sp][sp][4]; frame f of a clip goes to plane hd + f
// (the 2*hd halo planes of every clip are not touched)
template <bool F32>
__global__ __launch_bounds__(256) void pack_clips_kernel(const void* __restrict__ clips, int n, int s,
                                                         int kHalo0, short* __restrict__ y, int T, int hd) {
    const int sp = s + 2 * kHalo0;          // even: s is even
    const int pairs_per_row = sp / 2;
    const long long total = (long long)n * sp * pairs_per_row;
    const long long plane = (long long)s * s;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int px = (int)(i % pairs_per_row) * 2;
        const long long t = i / pairs_per_row;
        const int py = (int)(t % sp);
        const long long img = t / sp;
        bf16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        const int yy = py - kHalo0;
        if (yy >= 0 && yy < s) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int xx = px + k - kHalo0;
                if (xx >= 0 && xx < s) {
                    const long long base = img * 3 * plane + (long long)yy * s + xx;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if constexpr (F32) o[4 * k + c] = f32_to_bf16(((const float*)clips)[base + c * plane]);
                        else o[4 * k + c] = ((const short*)clips)[base + c * plane];
                    }
                }
            }
        }
        const long long dimg = T ? img / T * (T + 2 * hd) + hd + img % T : img;
        *(bf16x8*)(y + 8 * ((dimg * sp + py) * pairs_per_row + px / 2)) = o;
    }
}

// bf16 clips, s % 4 == 0, 8-byte aligned: one lane per 8 output pixels (64 B).  Per channel the lane reads
// three aligned 8-byte words around its 8 source pixels (the halo shifts a row by an odd number of pixels,
// so the sources are only 2-byte aligned) and funnel-shifts them into place; 9 loads and 4 stores per 64 B
// against 24 two-byte loads for the same bytes in the generic kernel.
__global__ __launch_bounds__(256) void pack_clips_bf16x8_kernel(const short* __restrict__ clips, int n, int s, int kHalo0,
                                                                short* __restrict__ y, int T, int hd) {
    const int sp = s + 2 * kHalo0;
    const int groups = (sp + 7) >> 3;
    const long long total = (long long)n * sp * groups;
    const long long plane = (long long)s * s;
    const int e_off = ((-kHalo0) % 4 + 4) % 4;          // first source pixel of a group relative to its aligned word
    const int shr = 16 * e_off;
    typedef unsigned long long u64;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % groups);
        const long long t = i / groups;
        const int py = (int)(t % sp);
        const long long img = t / sp;
        const int px0 = g * 8, yy = py - kHalo0;
        u64 v[3][2] = {{0, 0}, {0, 0}, {0, 0}};           // [channel][pixels 0-3 | 4-7]
        if (yy >= 0 && yy < s) {
            const int e0 = px0 - kHalo0 - e_off;         // multiple of 4 (may be negative)
            const short* row = clips + img * 3 * plane + (long long)yy * s;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                u64 w[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int e = e0 + 4 * k;
                    w[k] = (e >= 0 && e + 4 <= s) ? *(const u64*)(row + c * plane + e) : 0ull;
                }
                v[c][0] = shr ? (w[0] >> shr) | (w[1] << (64 - shr)) : w[0];
                v[c][1] = shr ? (w[1] >> shr) | (w[2] << (64 - shr)) : w[1];
            }
        }
        const long long dimg = T ? img / T * (T + 2 * hd) + hd + img % T : img;
        short* dst = y + (((dimg * sp + py) * sp) + px0) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                    // two pixels per 16-byte store
            if (px0 + 2 * q < sp) {
                const int h = q >> 1, sh0 = 32 * (q & 1);
                const unsigned a0 = (unsigned)(v[0][h] >> sh0), a1 = (unsigned)(v[1][h] >> sh0), a2 = (unsigned)(v[2][h] >> sh0);
                i32x4 o;
                o[0] = (int)((a0 & 0xffffu) | (a1 << 16));
                o[1] = (int)(a2 & 0xffffu);
                o[2] = (int)((a0 >> 16) | (a1 & 0xffff0000u));
                o[3] = (int)(a2 >> 16);
                *(i32x4*)(dst + 8 * q) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// uint8 frames [n][h0][w0][3] -> bilinear resize -> /255 -> (x-mean)/std -> packed bf16
// Source index and weights follow ATen's area_pixel_compute_source_index /
// guard_index_and_lambda for align_corners=False (scale = in/out in fp32).
// ---------------------------------------------------------------------------------------
// PLANAR: y is the bf16 clip tensor [n][3][s][s] itself (what the planar-input stem reads in place; kHalo0 = 0): 6 bytes
// written per pixel instead of 8 plus halo, and no packed copy for the stem to re-read.
#ifndef ASL_PRE_ROWS
#define ASL_PRE_ROWS 4
#endif
constexpr int kPreRows = ASL_PRE_ROWS;     // output rows per workgroup pass

template <bool PLANAR>
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ frames, int n, int h0, int w0,
                                                         int s, int kHalo0, float m0, float m1, float m2, float i0,
                                                         float i1, float i2, short* __restrict__ y,
                                                         float* __restrict__ clips_f32) {
    // fused multiply-adds are written out (and the compiler's own fusing is off): which products it fuses differs
    // between the two instantiations, and the two output forms are specified to hold the same values bit for bit
#pragma clang fp contract(off)
    const int sp = s + 2 * kHalo0;
    const float sy = (float)h0 / (float)s, sx = (float)w0 / (float)s;
    // 1/255 and 1/std as multiplications: a true fp32 division is ~10 instructions and the kernel does six per
    // pixel; the result differs from (v / 255 - mean) / std by at most 2 ulp of fp32
    const float mean[3] = {m0, m1, m2}, isd[3] = {1.f / i0, 1.f / i1, 1.f / i2};
    const float inv255 = 1.f / 255.f;
    const long long frame_bytes = (long long)n * h0 * w0 * 3;
    // A workgroup takes kRows consecutive output rows (row index and vertical weights are wave-uniform), lanes
    // run along the row; a lane first issues the source loads of all its rows, then blends and stores: one load
    // round trip per kRows pixels instead of one per pixel (the kernel is bound by that latency, not by bytes).
    constexpr int kRows = kPreRows;
    const unsigned rows = (unsigned)n * (unsigned)sp;       // < 2^31: checked on the host
    const unsigned groups = (rows + kRows - 1) / kRows;
    for (unsigned grp = blockIdx.x; grp < groups; grp += gridDim.x)
    for (int px = threadIdx.x; px < sp; px += 256) {
        const int ox = px - kHalo0;
        const bool col_in = ox >= 0 && ox < s;
        float fx = __builtin_fmaf(sx, (float)ox + 0.5f, -0.5f);      // (fused, as the CPU library computes it)
        fx = fx < 0.f ? 0.f : fx;
        int x0 = (int)fx;
        x0 = x0 < w0 - 1 ? x0 : w0 - 1;
        const int x1 = x0 + (x0 < w0 - 1 ? 1 : 0);
        const float lx = fminf(fmaxf(fx - (float)x0, 0.f), 1.f), hx = 1.f - lx;
        unsigned long long q0[kRows], q1[kRows];
        float ly[kRows];
        bool live[kRows];
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const unsigned row = grp * kRows + r;
            const unsigned img = row / (unsigned)sp;
            const int oy = (int)(row - img * sp) - kHalo0;
            live[r] = row < rows && col_in && oy >= 0 && oy < s;
            q0[r] = q1[r] = 0;
            ly[r] = 0.f;
            if (live[r]) {
                float fy = __builtin_fmaf(sy, (float)oy + 0.5f, -0.5f);
                fy = fy < 0.f ? 0.f : fy;
                int y0 = (int)fy;
                y0 = y0 < h0 - 1 ? y0 : h0 - 1;
                const int y1 = y0 + (y0 < h0 - 1 ? 1 : 0);
                ly[r] = fminf(fmaxf(fy - (float)y0, 0.f), 1.f);
                const long long fo = (long long)img * h0 * w0 * 3;
                const long long o0 = fo + ((long long)y0 * w0 + x0) * 3, o1 = fo + ((long long)y1 * w0 + x0) * 3;
                // the two source pixels of a row are 6 consecutive bytes: one unaligned 8-byte load per row instead
                // of six byte loads (not in the last column, where x1 == x0, nor within 8 bytes of the tensor's end)
                if (x1 == x0 + 1 && o1 + 8 <= frame_bytes) {
                    struct __attribute__((packed)) U64 { unsigned long long v; };
                    q0[r] = ((const U64*)(frames + o0))->v;
                    q1[r] = ((const U64*)(frames + o1))->v;
                } else {
                    const uint8_t* a0 = frames + o0;
                    const uint8_t* a1 = frames + o1;
                    const int d = (x1 - x0) * 3;
                    q0[r] = (unsigned long long)a0[0] | ((unsigned long long)a0[1] << 8) | ((unsigned long long)a0[2] << 16) |
                            ((unsigned long long)a0[d] << 24) | ((unsigned long long)a0[d + 1] << 32) | ((unsigned long long)a0[d + 2] << 40);
                    q1[r] = (unsigned long long)a1[0] | ((unsigned long long)a1[1] << 8) | ((unsigned long long)a1[2] << 16) |
                            ((unsigned long long)a1[d] << 24) | ((unsigned long long)a1[d + 1] << 32) | ((unsigned long long)a1[d + 2] << 40);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const unsigned row = grp * kRows + r;
            if (row >= rows) break;
            const unsigned img = row / (unsigned)sp;
            const int oy = (int)(row - img * sp) - kHalo0;
            bf16x4 o = {0, 0, 0, 0};
            if (live[r]) {
                const float hy = 1.f - ly[r];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float v00 = (float)(unsigned)((q0[r] >> (8 * c)) & 0xff), v01 = (float)(unsigned)((q0[r] >> (8 * c + 24)) & 0xff);
                    const float v10 = (float)(unsigned)((q1[r] >> (8 * c)) & 0xff), v11 = (float)(unsigned)((q1[r] >> (8 * c + 24)) & 0xff);
                    const float top = __builtin_fmaf(lx, v01, hx * v00), bot = __builtin_fmaf(lx, v11, hx * v10);
                    const float v = __builtin_fmaf(ly[r], bot, hy * top);
                    const float z = __builtin_fmaf(v, inv255, -mean[c]) * isd[c];
                    o[c] = f32_to_bf16(z);
                    if (PLANAR) y[(((long long)img * 3 + c) * s + oy) * (long long)s + ox] = o[c];
                    else if (clips_f32) clips_f32[(((long long)img * 3 + c) * s + oy) * (long long)s + ox] = z;
                }
            }
            if (!PLANAR) *(bf16x4*)(y + 4 * ((long long)row * sp + px)) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same resize through LDS, for frames whose source rows of a group of output rows fit there (any upscale, and
// downscales up to about 1.3x at 320-pixel rows): the rows a group of kStageRows output rows blends from are one
// contiguous byte range of the frame, copied into LDS with aligned 16-byte loads; the six bytes of a source pixel pair
// then come from three aligned LDS words and two byte-alignment instructions.  The direct kernel's unaligned 8-byte
// global loads run at a fraction of the aligned rate and bound it; the arithmetic here is the same, operation for
// operation, so the two kernels' outputs are identical.
// ---------------------------------------------------------------------------------------
#ifndef ASL_STAGE_ROWS
#define ASL_STAGE_ROWS 8
#endif
constexpr int kStageRows = ASL_STAGE_ROWS;
constexpr int kStageMaxBytes = 32768;

// ATen's area_pixel_compute_source_index + guard_index_and_lambda (align_corners=False)
__host__ __device__ inline void resize_src_index(float scale, int o, int lim, int& i0, int& i1, float& l) {
    float f = __builtin_fmaf(scale, (float)o + 0.5f, -0.5f);
    f = f < 0.f ? 0.f : f;
    i0 = (int)f;
    i0 = i0 < lim - 1 ? i0 : lim - 1;
    i1 = i0 + (i0 < lim - 1 ? 1 : 0);
    l = f - (float)i0;
    l = l < 0.f ? 0.f : (l > 1.f ? 1.f : l);
}

// bytes [first source row of the group, end of its last) of image `img`, for padded output rows [r0, r0 + kStageRows)
__host__ __device__ inline bool stage_range(int r0, int halo, int s, int h0, int w0, float sy, int& ylo, int& yhi) {
    const int oy_lo = r0 - halo > 0 ? r0 - halo : 0;
    const int oy_hi = r0 + kStageRows - 1 - halo < s - 1 ? r0 + kStageRows - 1 - halo : s - 1;
    if (oy_lo > oy_hi) return false;
    int t;
    float l;
    resize_src_index(sy, oy_lo, h0, ylo, t, l);
    resize_src_index(sy, oy_hi, h0, t, yhi, l);
    return true;
}

template <bool PLANAR>
__global__ __launch_bounds__(256) void preprocess_staged_kernel(const uint8_t* __restrict__ frames, int n, int h0, int w0,
                                                                int s, int halo, float m0, float m1, float m2, float i0,
                                                                float i1, float i2, short* __restrict__ y,
                                                                float* __restrict__ clips_f32, int groups_per_img) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char stage_lds[];
    const int sp = s + 2 * halo;
    const float sy = (float)h0 / (float)s, sx = (float)w0 / (float)s;
    const float mean[3] = {m0, m1, m2}, isd[3] = {1.f / i0, 1.f / i1, 1.f / i2};
    const float inv255 = 1.f / 255.f;
    const int img = blockIdx.x / groups_per_img, r0 = (blockIdx.x - img * groups_per_img) * kStageRows;
    const int r_end = r0 + kStageRows < sp ? r0 + kStageRows : sp;
    const int tid = threadIdx.x;
    const long long total = (long long)n * h0 * w0 * 3;
    const int row_bytes = w0 * 3;
    int ylo = 0, yhi = 0;
    unsigned shift = 0;
    if (stage_range(r0, halo, s, h0, w0, sy, ylo, yhi)) {
        const long long start = ((long long)img * h0 + ylo) * row_bytes, end = ((long long)img * h0 + yhi + 1) * row_bytes;
        const long long astart = start & ~15LL;             // (frames is 16-byte aligned: checked on the host)
        shift = (unsigned)(start - astart);
        const int n16 = (int)((end - astart + 15) >> 4);
        for (int base = 0; base < n16; base += 1024) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = base + k * 256 + tid;
                const long long o = astart + 16LL * i;
                v[k] = uint4{0, 0, 0, 0};
                if (i < n16) {
                    if (o + 16 <= total) v[k] = *(const uint4*)(frames + o);
                    else {                                   // the tensor's last, partial 16 bytes
                        unsigned w[4] = {0, 0, 0, 0};
                        for (int b = 0; b < 16 && o + b < total; ++b) w[b >> 2] |= (unsigned)frames[o + b] << (8 * (b & 3));
                        v[k] = uint4{w[0], w[1], w[2], w[3]};
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = base + k * 256 + tid;
                if (i < n16) ((uint4*)stage_lds)[i] = v[k];
            }
        }
    }
    __syncthreads();
    const unsigned* lds32 = (const unsigned*)stage_lds;
    for (int px = tid; px < sp; px += 256) {
        const int ox = px - halo;
        const bool col_in = ox >= 0 && ox < s;
        int x0, x1;
        float lx;
        resize_src_index(sx, ox, w0, x0, x1, lx);
        const float hx = 1.f - lx;
        const unsigned bx = shift + 3u * (unsigned)x0;
        const bool pair = x1 != x0;
        for (int r = r0; r < r_end; ++r) {
            const int oy = r - halo;
            const unsigned row = (unsigned)img * (unsigned)sp + (unsigned)r;
            bf16x4 o = {0, 0, 0, 0};
            if (col_in && oy >= 0 && oy < s) {
                int y0, y1;
                float ly;
                resize_src_index(sy, oy, h0, y0, y1, ly);
                const float hy = 1.f - ly;
                unsigned d[2][2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const unsigned b = (unsigned)((t ? y1 : y0) - ylo) * (unsigned)row_bytes + bx;
                    const unsigned wi = b >> 2, sh = b & 3;
                    const unsigned w_0 = lds32[wi], w_1 = lds32[wi + 1], w_2 = lds32[wi + 2];
                    d[t][0] = __builtin_amdgcn_alignbyte(w_1, w_0, sh);
                    d[t][1] = __builtin_amdgcn_alignbyte(w_2, w_1, sh);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const unsigned long long q0 = ((unsigned long long)d[0][1] << 32) | d[0][0];
                    const unsigned long long q1 = ((unsigned long long)d[1][1] << 32) | d[1][0];
                    const int s1 = pair ? 8 * c + 24 : 8 * c;
                    const float v00 = (float)(unsigned)((q0 >> (8 * c)) & 0xff), v01 = (float)(unsigned)((q0 >> s1) & 0xff);
                    const float v10 = (float)(unsigned)((q1 >> (8 * c)) & 0xff), v11 = (float)(unsigned)((q1 >> s1) & 0xff);
                    const float top = __builtin_fmaf(lx, v01, hx * v00), bot = __builtin_fmaf(lx, v11, hx * v10);
                    const float v = __builtin_fmaf(ly, bot, hy * top);
                    const float z = __builtin_fmaf(v, inv255, -mean[c]) * isd[c];
                    o[c] = f32_to_bf16(z);
                    if (PLANAR) y[(((long long)img * 3 + c) * s + oy) * (long long)s + ox] = o[c];
                    else if (clips_f32) clips_f32[(((long long)img * 3 + c) * s + oy) * (long long)s + ox] = z;
                }
            }
            if (!PLANAR) *(bf16x4*)(y + 4 * ((long long)row * sp + px)) = o;
        }
    }
}

// LDS bytes the staged kernel needs for this geometry, or 0 when a group's source rows do not fit (or the frames are
// not 16-byte aligned): the direct kernel then runs
static int preprocess_stage_bytes(const void* frames, int h0, int w0, int s, int halo) {
    if (((uintptr_t)frames & 15) || getenv("ASL_PREPROCESS_DIRECT")) return 0;
    const float sy = (float)h0 / (float)s;
    const int sp = s + 2 * halo;
    int rows = 0;
    for (int r0 = 0; r0 < sp; r0 += kStageRows) {
        int ylo, yhi;
        if (stage_range(r0, halo, s, h0, w0, sy, ylo, yhi) && yhi - ylo + 1 > rows) rows = yhi - ylo + 1;
    }
    // + 15 bytes of alignment shift, rounded up to whole 16-byte pieces, + the two words a pixel read may run past
    const long long need = (((long long)rows * w0 * 3 + 15 + 15) & ~15LL) + 16;
    return need <= kStageMaxBytes ? (int)need : 0;
}

template <bool PLANAR>
static int preprocess_launch(const uint8_t* frames, int n, int h0, int w0, int s, int halo, const float* mean, const float* std_,
                             void* y, float* clips_f32, hipStream_t st) {
    const int sp = s + 2 * halo;
    const int lds = preprocess_stage_bytes(frames, h0, w0, s, halo);
    const long long groups = (long long)n * ((sp + kStageRows - 1) / kStageRows);
    if (lds > 0 && groups <= 0x7fffffffLL) {
        hipLaunchKernelGGL(preprocess_staged_kernel<PLANAR>, dim3((unsigned)groups), dim3(256), lds, st, frames, n, h0, w0, s, halo,
                           mean[0], mean[1], mean[2], std_[0], std_[1], std_[2], (short*)y, clips_f32,
                           (sp + kStageRows - 1) / kStageRows);
        return (int)hipGetLastError();
    }
    const long long rows = ((long long)n * sp + kPreRows - 1) / kPreRows;
    hipLaunchKernelGGL(preprocess_kernel<PLANAR>, dim3((unsigned)(rows < (1 << 20) ? rows : (1 << 20))), dim3(256), 0, st, frames, n, h0, w0,
                       s, halo, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2], (short*)y, clips_f32);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// max pooling over halo'd channels-last tensors; one lane per 8 channels of an output pixel
// ---------------------------------------------------------------------------------------
struct PoolArgs {
    const short* x;
    short* y;
    int n, od, oh, ow, c8;          // c8 = channels / 8
    int kd, kh, kw, sd, sh, sw;
    long long in_sn, out_sn;        // element strides
    int in_sd, in_sh, in_sw, in_off0;
    int out_sd, out_sh, out_sw, out_off0;
};

__global__ __launch_bounds__(256) void maxpool_kernel(const PoolArgs a) {
    const long long total = (long long)a.n * a.od * a.oh * a.ow * a.c8;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % a.c8);
        long long t = i / a.c8;
        const int w = (int)(t % a.ow); t /= a.ow;
        const int h = (int)(t % a.oh); t /= a.oh;
        const int d = (int)(t % a.od);
        const long long n = t / a.od;
        const short* src = a.x + n * a.in_sn + (long long)d * a.sd * a.in_sd + (long long)h * a.sh * a.in_sh +
                           (long long)w * a.sw * a.in_sw + a.in_off0 + 8 * c;
        // inputs are >= 0 (post-ReLU) and every window holds a real cell: non-negative bf16 values order like their
        // bit patterns read as signed 16-bit integers (and -0.0 reads as the most negative one), so the max runs
        // on the packed pairs directly (v_pk_max_i16: two channels per instruction, no unpacking), starting from 0
        bf16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int kd = 0; kd < a.kd; ++kd)
            for (int kh = 0; kh < a.kh; ++kh) {
                const short* row = src + (long long)kd * a.in_sd + (long long)kh * a.in_sh;
                for (int kw = 0; kw < a.kw; ++kw) o = __builtin_elementwise_max(o, *(const bf16x8*)(row + (long long)kw * a.in_sw));
            }
        *(bf16x8*)(a.y + n * a.out_sn + (long long)d * a.out_sd + (long long)h * a.out_sh + (long long)w * a.out_sw +
                   a.out_off0 + 8 * c) = o;
    }
}

int grid_for(long long total);

// The same for the window sizes the models use, known at compile time: a lane produces WPT neighbouring outputs of
// a row for its 8 channels.  Every input column the lane needs is reduced over the window's (kd, kh) rows once and
// shared by the outputs it belongs to (3x3x3 / stride 1, four outputs: 54 loads instead of 108), and the output
// index is split with multiply-high divisions instead of 64-bit ones.
template <int KD, int KH, int KW, int SW, int WPT>
__global__ __launch_bounds__(256) void maxpool_fixed_kernel(const PoolArgs a, const FastDiv dC8, const FastDiv dWG, const FastDiv dOH,
                                                            const FastDiv dOD, const int wgroups) {
    constexpr int NCOL = (WPT - 1) * SW + KW;
    const unsigned total = (unsigned)a.n * a.od * a.oh * wgroups * a.c8;       // < 2^31, checked by the host
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        unsigned t = fdiv(i, dC8);
        const int c = (int)(i - t * a.c8);
        unsigned q = fdiv(t, dWG);
        const int w0 = (int)(t - q * wgroups) * WPT;
        t = q; q = fdiv(t, dOH);
        const int h = (int)(t - q * a.oh);
        t = q; q = fdiv(t, dOD);
        const int d = (int)(t - q * a.od);
        const long long n = q;
        const short* src = a.x + n * a.in_sn + (long long)d * a.sd * a.in_sd + (long long)h * a.sh * a.in_sh +
                           (long long)w0 * SW * a.in_sw + a.in_off0 + 8 * c;
        const int nout = a.ow - w0 < WPT ? a.ow - w0 : WPT;                    // outputs of this lane that exist
        const int ncol = (nout - 1) * SW + KW;                                 // columns they read
        bf16x8 col[NCOL];
#pragma unroll
        for (int j = 0; j < NCOL; ++j) col[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int kd = 0; kd < KD; ++kd)
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                const short* row = src + (long long)kd * a.in_sd + (long long)kh * a.in_sh;
#pragma unroll
                for (int j = 0; j < NCOL; ++j)
                    if (j < KW || j < ncol) col[j] = __builtin_elementwise_max(col[j], *(const bf16x8*)(row + (long long)j * a.in_sw));
            }
        short* dst = a.y + n * a.out_sn + (long long)d * a.out_sd + (long long)h * a.out_sh + (long long)w0 * a.out_sw + a.out_off0 + 8 * c;
#pragma unroll
        for (int o = 0; o < WPT; ++o) {
            bf16x8 m = col[o * SW];
#pragma unroll
            for (int kw = 1; kw < KW; ++kw) m = __builtin_elementwise_max(m, col[o * SW + kw]);
            if (o < nout) *(bf16x8*)(dst + (long long)o * a.out_sw) = m;
        }
    }
}

// 3x3x3 / stride 1 (the pooling branch of every Inception block): a lane walks DS output planes for its WPT outputs of a
// row and 8 channels.  Each input plane is reduced once, over its three rows and then over the three columns of every
// output, to a per-plane maximum that the three output planes around it share (kept in registers, rolled from plane to
// plane): 18 loads per plane for four outputs -- 5.6 loads per output with DS = 8, against 13.5 when every output plane
// starts over.
template <int WPT, int DS>
__global__ __launch_bounds__(256) void maxpool333_roll_kernel(const PoolArgs a, const FastDiv dC8, const FastDiv dWG, const FastDiv dOH,
                                                              const FastDiv dSEG, const int wgroups, const int nseg) {
    constexpr int NCOL = WPT + 2;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned total = (unsigned)a.n * nseg * a.oh * wgroups * a.c8;       // < 2^31, checked by the host
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        unsigned t = fdiv(i, dC8);
        const int c = (int)(i - t * a.c8);
        unsigned q = fdiv(t, dWG);
        const int w0 = (int)(t - q * wgroups) * WPT;
        t = q; q = fdiv(t, dOH);
        const int h = (int)(t - q * a.oh);
        t = q; q = fdiv(t, dSEG);
        const int d0 = (int)(t - q * nseg) * DS;
        const long long n = q;
        const int nd = a.od - d0 < DS ? a.od - d0 : DS;                        // output planes of this lane
        const int nout = a.ow - w0 < WPT ? a.ow - w0 : WPT;                    // outputs per row that exist
        const int ncol = nout + 2;
        const short* src = a.x + n * a.in_sn + (long long)d0 * a.in_sd + (long long)h * a.in_sh + (long long)w0 * a.in_sw +
                           a.in_off0 + 8 * c;                                  // input plane d0 - 1 (tap 0 of output plane d0)
        short* dst = a.y + n * a.out_sn + (long long)d0 * a.out_sd + (long long)h * a.out_sh + (long long)w0 * a.out_sw + a.out_off0 + 8 * c;
        bf16x8 pa[WPT], pb[WPT];                                               // maxima of the two planes before the current one
#pragma unroll
        for (int o = 0; o < WPT; ++o) pa[o] = pb[o] = zero;
        for (int p = 0; p < nd + 2; ++p) {
            const short* pl = src + (long long)p * a.in_sd;
            bf16x8 col[NCOL];
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                // columns past the row's last output are not needed: they re-read the last needed one, so that all 18 loads
                // of the plane go out together (a predicate per column would put a branch and a wait between them)
                const int jj = j < ncol ? j : ncol - 1;
                const bf16x8 r0 = *(const bf16x8*)(pl + jj * a.in_sw);
                const bf16x8 r1 = *(const bf16x8*)(pl + a.in_sh + jj * a.in_sw);
                const bf16x8 r2 = *(const bf16x8*)(pl + 2 * a.in_sh + jj * a.in_sw);
                col[j] = __builtin_elementwise_max(__builtin_elementwise_max(r0, r1), r2);
            }
#pragma unroll
            for (int o = 0; o < WPT; ++o) {
                const bf16x8 pm = __builtin_elementwise_max(__builtin_elementwise_max(col[o], col[o + 1]), col[o + 2]);
                if (p >= 2 && o < nout)
                    *(bf16x8*)(dst + (long long)(p - 2) * a.out_sd + (long long)o * a.out_sw) =
                        __builtin_elementwise_max(__builtin_elementwise_max(pa[o], pb[o]), pm);
                pa[o] = pb[o];
                pb[o] = pm;
            }
        }
    }
}

template <int WPT, int DS>
static int launch_maxpool333_roll(const PoolArgs& a, hipStream_t stream) {
    const int wgroups = (a.ow + WPT - 1) / WPT, nseg = (a.od + DS - 1) / DS;
    const long long total = (long long)a.n * nseg * a.oh * wgroups * a.c8;
    if (total >= 0x7fffffffLL) return -1;
    hipLaunchKernelGGL((maxpool333_roll_kernel<WPT, DS>), dim3(grid_for(total)), dim3(256), 0, stream, a,
                       make_fastdiv((unsigned)a.c8), make_fastdiv((unsigned)wgroups), make_fastdiv((unsigned)a.oh),
                       make_fastdiv((unsigned)nseg), wgroups, nseg);
    return (int)hipGetLastError();
}

template <int KD, int KH, int KW, int SW, int WPT>
static int launch_maxpool_fixed(const PoolArgs& a, hipStream_t stream) {
    const int wgroups = (a.ow + WPT - 1) / WPT;
    const long long total = (long long)a.n * a.od * a.oh * wgroups * a.c8;
    if (total >= 0x7fffffffLL) return -1;
    hipLaunchKernelGGL((maxpool_fixed_kernel<KD, KH, KW, SW, WPT>), dim3(grid_for(total)), dim3(256), 0, stream, a,
                       make_fastdiv((unsigned)a.c8), make_fastdiv((unsigned)wgroups), make_fastdiv((unsigned)a.oh),
                       make_fastdiv((unsigned)a.od), wgroups);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// per-frame spatial mean: halo'd NHWC bf16 -> fp32 [n][c] (+ bf16 copy)
// block = one frame; lane layout (pixel slice, 8-channel chunk)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_kernel(const short* __restrict__ x, int h, int w, int c, int halo,
                                                      float* __restrict__ yf, short* __restrict__ yb, int ldb,
                                                      int lo_off) {
    __shared__ float red[256 * 8];
    const int c8 = c >> 3;
    const int n = blockIdx.x;
    const int wp = w + 2 * halo, hp = h + 2 * halo;
    const short* img = x + (long long)n * hp * wp * c;
    const float inv = 1.0f / (float)(h * w);
    for (int cb = 0; cb < c8; cb += 256) {           // channel chunks handled per pass
        const int chunks = (c8 - cb) < 256 ? (c8 - cb) : 256;
        const int slices = 256 / chunks;             // pixel slices per pass (chunks is a power of two or >= 256)
        const int ch = threadIdx.x % chunks, sl = threadIdx.x / chunks;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (sl < slices) {
            for (int p = sl; p < h * w; p += slices) {
                const int py = p / w, px = p - py * w;
                const bf16x8 v = *(const bf16x8*)(img + ((long long)(py + halo) * wp + px + halo) * c + 8 * (cb + ch));
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += bf16_to_f32(v[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = acc[e];
        __syncthreads();
        if (sl == 0) {
            for (int s2 = 1; s2 < slices; ++s2)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += red[(s2 * chunks + ch) * 8 + e];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = acc[e] * inv;
                yf[(long long)n * c + 8 * (cb + ch) + e] = v;
                if (yb) {
                    const short hi = f32_to_bf16(v);
                    yb[(long long)n * ldb + 8 * (cb + ch) + e] = hi;
                    if (lo_off > 0) yb[(long long)n * ldb + lo_off + 8 * (cb + ch) + e] = f32_to_bf16(v - bf16_to_f32(hi));
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void mean_t_kernel(const float* __restrict__ x, int rows, int t, int c,
                                                     float* __restrict__ y) {
    const long long total = (long long)rows * c;
    const float inv = 1.0f / (float)t;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / c;
        const int ch = (int)(i - r * c);
        const float* p = x + r * t * c + ch;
        // eight loads in flight per thread; the additions keep their sequential order, so the sum is the
        // same as a plain loop's bit for bit
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= t; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long long)(k + u) * c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < t; ++k) s += p[(long long)k * c];
        y[i] = s * inv;
    }
}

// F.avg_pool3d's temporal part followed by the mean over the remaining steps:
// y[r][c] = mean_{t'=0..t-window} mean_{k<window} x[r][t'+k][c]
__global__ __launch_bounds__(256) void window_mean_t_kernel(const float* __restrict__ x, int rows, int t, int c,
                                                            int window, float* __restrict__ y) {
    const long long total = (long long)rows * c;
    const int nout = t - window + 1;
    const float iw = 1.0f / (float)window, io = 1.0f / (float)nout;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / c;
        const int ch = (int)(i - r * c);
        const float* p = x + r * t * c + ch;
        float s = 0.f;
        for (int o = 0; o < nout; ++o) {
            float a = 0.f;
            for (int k = 0; k < window; ++k) a += p[(long long)(o + k) * c];
            s += a * iw;
        }
        y[i] = s * io;
    }
}

// one wave per output element: dot product over f with a shuffle reduction
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ x, int m, int f,
                                                   const float* __restrict__ w, const float* __restrict__ b, int c,
                                                   float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long long wid = (blockIdx.x * 256LL + threadIdx.x) >> 6;
    if (wid >= (long long)m * c) return;
    const int row = (int)(wid / c), col = (int)(wid % c);
    const float* xr = x + (long long)row * f;
    const float* wr = w + (long long)col * f;
    float s = 0.f;
    for (int k = lane; k < f; k += 64) s += xr[k] * wr[k];
    s = wave_sum(s);
    if (lane == 0) y[wid] = s + (b ? b[col] : 0.f);
}

// one block per row: softmax statistics, then k rounds of arg-max over the row kept in LDS
__global__ __launch_bounds__(256) void softmax_topk_kernel(const float* __restrict__ logits, int c, int k,
                                                           int64_t* __restrict__ idx, float* __restrict__ prob) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* v = (float*)smem_raw;              // [c]
    __shared__ float wred[4];
    __shared__ int wredi[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* lr = logits + (long long)row * c;
    // NaN logits rank last: they are staged as -inf (fmaxf already ignores them for the row maximum)
    float mx = -INFINITY;
    for (int i = tid; i < c; i += 256) {
        const float x = lr[i];
        v[i] = x == x ? x : -INFINITY;
        mx = fmaxf(mx, x);
    }
    mx = wave_max(mx);
    if (lane == 0) wred[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < c; i += 256) sum += expf(v[i] - mx);
    sum = wave_sum(sum);
    if (lane == 0) wred[wv] = sum;
    __syncthreads();
    sum = (wred[0] + wred[1]) + (wred[2] + wred[3]);
    __syncthreads();
    // k rounds of arg-max.  A picked entry is marked NaN (never a candidate again); among candidates an
    // index always wins over "none yet" (kNone), so rows of -inf still yield k distinct valid indices.
    constexpr int kNone = 0x7fffffff;
    for (int r = 0; r < k; ++r) {
        float best = -INFINITY;
        int bi = kNone;
        for (int i = tid; i < c; i += 256) {
            const float x = v[i];
            if (x == x && (bi == kNone || x > best)) { best = x; bi = i; }     // ascending i: first maximum wins
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi != kNone && (bi == kNone || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
        }
        if (lane == 0) { wred[wv] = best; wredi[wv] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int q = 1; q < 4; ++q)
                if (wredi[q] != kNone && (bi == kNone || wred[q] > best || (wred[q] == best && wredi[q] < bi))) {
                    best = wred[q];
                    bi = wredi[q];
                }
            if (bi < c) {                      // always true for k <= c; never index LDS with a sentinel
                idx[(long long)row * k + r] = bi;
                prob[(long long)row * k + r] = expf(best - mx) / sum;
                v[bi] = __builtin_nanf("");
            }
        }
        __syncthreads();
    }
}

int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace asl_detail
using namespace asl_detail;

static int pack_clips_launch(const void* clips, int in_f32, int n, int s, int halo, void* y, int T, int hd, hipStream_t st) {
    const long long total = (long long)n * (s + 2 * halo) * ((s + 2 * halo) / 2);
    if (in_f32) hipLaunchKernelGGL(pack_clips_kernel<true>, dim3(grid_for(total)), dim3(256), 0, st, clips, n, s, halo, (short*)y, T, hd);
    else if ((s & 3) == 0 && ((uintptr_t)clips & 7) == 0) {
        const long long items = (long long)n * (s + 2 * halo) * ((s + 2 * halo + 7) / 8);
        hipLaunchKernelGGL(pack_clips_bf16x8_kernel, dim3(grid_for(items)), dim3(256), 0, st, (const short*)clips, n, s, halo, (short*)y, T, hd);
    } else hipLaunchKernelGGL(pack_clips_kernel<false>, dim3(grid_for(total)), dim3(256), 0, st, clips, n, s, halo, (short*)y, T, hd);
    return (int)hipGetLastError();
}

extern "C" int asl_pack_clips(const void* clips, int in_f32, int n, int s, int halo, void* y, asl_stream_t stream) {
    if (!clips || !y || n <= 0 || s <= 0 || (s & 1) || halo < 0 || ((uintptr_t)y & 15)) return ASL_EINVAL;
    return pack_clips_launch(clips, in_f32, n, s, halo, y, 0, 0, (hipStream_t)stream);
}

extern "C" int asl_pack_clips_3d(const void* clips, int in_f32, int n_clips, int frames, int s, int halo, int halo_d, void* y,
                                 asl_stream_t stream) {
    if (!clips || !y || n_clips <= 0 || frames <= 0 || s <= 0 || (s & 1) || halo < 0 || halo_d < 0 || ((uintptr_t)y & 15)) return ASL_EINVAL;
    if ((long long)n_clips * frames > 0x7fffffffLL) return ASL_EINVAL;
    return pack_clips_launch(clips, in_f32, n_clips * frames, s, halo, y, frames, halo_d, (hipStream_t)stream);
}

extern "C" int asl_preprocess_u8(const uint8_t* frames, int n, int h0, int w0, int s, int halo, const float mean[3],
                                 const float std_[3], void* y, float* clips_f32, asl_stream_t stream) {
    if (!frames || !y || !mean || !std_ || n <= 0 || h0 <= 0 || w0 <= 0 || s <= 0 || halo < 0 || ((uintptr_t)y & 7))
        return ASL_EINVAL;
    for (int c = 0; c < 3; ++c)
        if (!(std_[c] > 0.f)) return ASL_EINVAL;
    if ((long long)n * (s + 2 * halo) >= 0x7fffffffLL) return ASL_EINVAL;
    return preprocess_launch<false>(frames, n, h0, w0, s, halo, mean, std_, y, clips_f32, (hipStream_t)stream);
}

extern "C" int asl_preprocess_u8_planar(const uint8_t* frames, int n, int h0, int w0, int s, const float mean[3],
                                        const float std_[3], void* clips_bf16, asl_stream_t stream) {
    if (!frames || !clips_bf16 || !mean || !std_ || n <= 0 || h0 <= 0 || w0 <= 0 || s <= 0 || ((uintptr_t)clips_bf16 & 1))
        return ASL_EINVAL;
    for (int c = 0; c < 3; ++c)
        if (!(std_[c] > 0.f)) return ASL_EINVAL;
    if ((long long)n * s >= 0x7fffffffLL) return ASL_EINVAL;
    return preprocess_launch<true>(frames, n, h0, w0, s, 0, mean, std_, clips_bf16, nullptr, (hipStream_t)stream);
}

extern "C" int asl_maxpool3d_ndhwc(const asl_pool3d_params* p, const void* x, void* y, asl_stream_t stream) {
    if (!p || !x || !y) return ASL_EINVAL;
    if (p->n <= 0 || p->d <= 0 || p->h <= 0 || p->w <= 0 || p->c <= 0 || (p->c & 7)) return ASL_EINVAL;
    if (p->kd <= 0 || p->kh <= 0 || p->kw <= 0 || p->sd <= 0 || p->sh <= 0 || p->sw <= 0) return ASL_EINVAL;
    if (p->pd < 0 || p->ph < 0 || p->pw < 0 || p->pd > p->in_halo_d || p->ph > p->in_halo || p->pw > p->in_halo)
        return ASL_EINVAL;
    // every window must hold at least one real cell for the zero halo to stand in for -inf
    if (p->pd >= p->kd || p->ph >= p->kh || p->pw >= p->kw) return ASL_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return ASL_EINVAL;
    PoolArgs a;
    a.x = (const short*)x; a.y = (short*)y;
    a.n = p->n; a.c8 = p->c / 8;
    a.od = (p->d + 2 * p->pd - p->kd) / p->sd + 1;
    a.oh = (p->h + 2 * p->ph - p->kh) / p->sh + 1;
    a.ow = (p->w + 2 * p->pw - p->kw) / p->sw + 1;
    if (a.od <= 0 || a.oh <= 0 || a.ow <= 0) return ASL_EINVAL;
    // the last window must not run past the halo
    if ((a.od - 1) * p->sd - p->pd + p->kd > p->d + p->in_halo_d) return ASL_EINVAL;
    if ((a.oh - 1) * p->sh - p->ph + p->kh > p->h + p->in_halo) return ASL_EINVAL;
    if ((a.ow - 1) * p->sw - p->pw + p->kw > p->w + p->in_halo) return ASL_EINVAL;
    a.kd = p->kd; a.kh = p->kh; a.kw = p->kw; a.sd = p->sd; a.sh = p->sh; a.sw = p->sw;
    const int wp = p->w + 2 * p->in_halo, hp = p->h + 2 * p->in_halo, dp = p->d + 2 * p->in_halo_d;
    a.in_sw = p->c; a.in_sh = wp * p->c; a.in_sd = hp * wp * p->c; a.in_sn = (long long)dp * hp * wp * p->c;
    a.in_off0 = (p->in_halo_d - p->pd) * a.in_sd + (p->in_halo - p->ph) * a.in_sh + (p->in_halo - p->pw) * a.in_sw;
    const int owp = a.ow + 2 * p->out_halo, ohp = a.oh + 2 * p->out_halo, odp = a.od + 2 * p->out_halo_d;
    a.out_sw = p->c; a.out_sh = owp * p->c; a.out_sd = ohp * owp * p->c;
    a.out_sn = (long long)odp * ohp * owp * p->c;
    a.out_off0 = p->out_halo_d * a.out_sd + p->out_halo * a.out_sh + p->out_halo * a.out_sw;
    int e = -1;
    // (rolling over depth pays once there are lanes enough to hide its serial walk: below ~64 k lanes the per-output kernel)
    if (a.kd == 3 && a.kh == 3 && a.kw == 3 && a.sd == 1 && a.sh == 1 && a.sw == 1 &&
        (long long)a.n * ((a.od + 7) / 8) * a.oh * ((a.ow + 3) / 4) * a.c8 >= 65536)
        // seven outputs per lane where rows divide by seven (28, 14: 27 loads per plane for seven outputs instead of 18 for
        // four; 29-33 us against 36-43 at 14 x 14, 98 / 129 against 104 / 145 at 28 x 28; fourteen per lane is slower again)
        e = a.ow % 7 == 0 ? launch_maxpool333_roll<7, 8>(a, (hipStream_t)stream) : launch_maxpool333_roll<4, 8>(a, (hipStream_t)stream);
    else if (a.kd == 3 && a.kh == 3 && a.kw == 3 && a.sw == 1) e = launch_maxpool_fixed<3, 3, 3, 1, 4>(a, (hipStream_t)stream);
    else if (a.kd == 3 && a.kh == 3 && a.kw == 3 && a.sw == 2) e = launch_maxpool_fixed<3, 3, 3, 2, 2>(a, (hipStream_t)stream);
    else if (a.kd == 1 && a.kh == 3 && a.kw == 3 && a.sw == 2) e = launch_maxpool_fixed<1, 3, 3, 2, 2>(a, (hipStream_t)stream);
    else if (a.kd == 2 && a.kh == 2 && a.kw == 2 && a.sw == 2) e = launch_maxpool_fixed<2, 2, 2, 2, 2>(a, (hipStream_t)stream);
    if (e != -1) return e;
    const long long total = (long long)a.n * a.od * a.oh * a.ow * a.c8;
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// nn.MaxPool2d(3, 2, 1) on a split-precision tensor (parity mode): [n][h+2*ih][w+2*ih][2c], channels [0, c) the hi halves
// and [c, 2c) the lo halves of values >= 0.  A thread takes 8 channels of one output pixel: max over the window of hi + lo in
// fp32 (the zero halo is neutral), written back as a hi / lo pair.
__global__ __launch_bounds__(256) void maxpool3x3s2_split_kernel(const short* __restrict__ x, short* __restrict__ y, int n, int h,
                                                                 int w, int c, int ih, int oh_) {
    const int c8 = c >> 3, ohh = (h + 2 - 3) / 2 + 1, oww = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)n * ohh * oww * c8;
    const int wp = w + 2 * ih, hp = h + 2 * ih, owp = oww + 2 * oh_, ohp = ohh + 2 * oh_;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % c8);
        long long r = i / c8;
        const int ox = (int)(r % oww); r /= oww;
        const int oy = (int)(r % ohh);
        const long long img = r / ohh;
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = 0.f;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const long long px = (img * hp + (2 * oy - 1 + ky + ih)) * wp + (2 * ox - 1 + kx + ih);
                const bf16x8 hv = *(const bf16x8*)(x + px * 2 * c + g * 8);
                const bf16x8 lv = *(const bf16x8*)(x + px * 2 * c + c + g * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], bf16_to_f32(hv[e]) + bf16_to_f32(lv[e]));
            }
        bf16x8 ho, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ho[e] = f32_to_bf16(m[e]);
            lo[e] = f32_to_bf16(m[e] - bf16_to_f32(ho[e]));
        }
        const long long opx = (img * ohp + oy + oh_) * owp + ox + oh_;
        *(bf16x8*)(y + opx * 2 * c + g * 8) = ho;
        *(bf16x8*)(y + opx * 2 * c + c + g * 8) = lo;
    }
}

extern "C" int asl_maxpool3x3s2_split_nhwc(const void* x, int n, int h, int w, int c, int in_halo, void* y, int out_halo,
                                           asl_stream_t stream) {
    if (!x || !y || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 7) || in_halo < 1 || out_halo < 0) return ASL_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return ASL_EINVAL;
    const long long total = (long long)n * ((h - 1) / 2 + 1) * ((w - 1) / 2 + 1) * (c / 8);
    hipLaunchKernelGGL(maxpool3x3s2_split_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const short*)x,
                       (short*)y, n, h, w, c, in_halo, out_halo);
    return (int)hipGetLastError();
}

extern "C" int asl_maxpool3x3s2_nhwc(const void* x, int n, int h, int w, int c, int in_halo, void* y, int out_halo,
                                     asl_stream_t stream) {
    asl_pool3d_params p;
    p.n = n; p.d = 1; p.h = h; p.w = w; p.c = c; p.in_halo_d = 0; p.in_halo = in_halo;
    p.kd = 1; p.kh = 3; p.kw = 3; p.sd = 1; p.sh = 2; p.sw = 2; p.pd = 0; p.ph = 1; p.pw = 1;
    p.out_halo_d = 0; p.out_halo = out_halo;
    if (in_halo < 1 || out_halo < 0) return ASL_EINVAL;
    return asl_maxpool3d_ndhwc(&p, x, y, stream);
}

extern "C" int asl_avgpool_nhwc(const void* x, int n, int h, int w, int c, int halo, float* y_f32, void* y_bf16,
                                int ld_bf16, int lo_off, asl_stream_t stream) {
    if (!x || !y_f32 || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 7) || halo < 0) return ASL_EINVAL;
    const int c8 = c / 8;
    if (c8 < 256 && (c8 & (c8 - 1))) return ASL_EINVAL;       // lane layout needs 256 % chunks == 0
    if (c8 > 256 && (c8 % 256)) return ASL_EINVAL;
    if (y_bf16 && (ld_bf16 < c || lo_off < 0 || (lo_off > 0 && (lo_off < c || lo_off + c > ld_bf16)))) return ASL_EINVAL;
    if ((uintptr_t)x & 15) return ASL_EINVAL;
    hipLaunchKernelGGL(avgpool_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const short*)x, h, w, c, halo,
                       y_f32, (short*)y_bf16, ld_bf16, lo_off);
    return (int)hipGetLastError();
}

extern "C" int asl_mean_t_f32(const float* x, int rows, int t, int c, float* y, asl_stream_t stream) {
    if (!x || !y || rows <= 0 || t <= 0 || c <= 0) return ASL_EINVAL;
    hipLaunchKernelGGL(mean_t_kernel, dim3(grid_for((long long)rows * c)), dim3(256), 0, (hipStream_t)stream, x, rows,
                       t, c, y);
    return (int)hipGetLastError();
}

extern "C" int asl_window_mean_t_f32(const float* x, int rows, int t, int c, int window, float* y, asl_stream_t stream) {
    if (!x || !y || rows <= 0 || t <= 0 || c <= 0 || window <= 0 || window > t) return ASL_EINVAL;
    hipLaunchKernelGGL(window_mean_t_kernel, dim3(grid_for((long long)rows * c)), dim3(256), 0, (hipStream_t)stream, x,
                       rows, t, c, window, y);
    return (int)hipGetLastError();
}

extern "C" int asl_head_f32(const float* x, int m, int f, const float* w, const float* b, int c, float* y,
                            asl_stream_t stream) {
    if (!x || !w || !y || m <= 0 || f <= 0 || c <= 0) return ASL_EINVAL;
    const long long waves = (long long)m * c;
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, m, f, w,
                       b, c, y);
    return (int)hipGetLastError();
}

extern "C" int asl_softmax_topk(const float* logits, int m, int c, int k, int64_t* idx, float* prob,
                                asl_stream_t stream) {
    if (!logits || !idx || !prob || m <= 0 || c <= 0 || k <= 0 || k > 16 || k > c || c > 12288) return ASL_EINVAL;
    hipLaunchKernelGGL(softmax_topk_kernel, dim3(m), dim3(256), (size_t)c * 4, (hipStream_t)stream, logits, c, k, idx,
                       prob);
    return (int)hipGetLastError();
}
