// Shared epilogue of the MFMA kernels: scale/bias (BatchNorm), residual add, ReLU, bf16 or fp32
// store, with the accumulator transposed through a small wave-private LDS stage so that every
// global access is 16 bytes per lane and 8 consecutive lanes cover one pixel's 64 contiguous
// channels (full 128-byte lines) instead of the accumulator's native 4-channel granules.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace asl_detail {

struct OutGeom {
    const float* scale;
    const float* bias;
    const char* res;
    char* y;
    int M, N;
    FastDiv dOW, dOHW, dODHW;          // output grid decomposition of row m (divisors OW, OH*OW, OD*OH*OW)
    long long out_sn, res_sn;
    int out_sd, out_sh, out_sw, out_off0;
    int res_sd, res_sh, res_sw, res_off0;
    int relu, out_f32;
    int res_f32;                       // residual is fp32 instead of bf16 (same geometry)
    int out_lo, res_lo;                // split-precision epilogue: element offsets of the lo halves (see epilogue_store)
};

constexpr int kEpiRowBytes(int wtn) { return wtn * 4 + 16; }          // fp32 row + 16 B pad
constexpr int kEpiWaveBytes(int wtn) { return 16 * kEpiRowBytes(wtn); }

// Per-lane output coordinates and the residual values of an epilogue, computed (and the residual loads
// issued) BEFORE the K loop so their latency hides under it.  Lane (g8, pr0) of a wave handles channels
// ch..ch+7 of pixels pr0, pr0 + PPI, ... of every 16-pixel fragment.
template <int FM, int FN>
struct EpiState {
    static constexpr int WTN = FN * 16;
    static constexpr int G8 = WTN / 8;            // 8-channel groups per pixel row of the wave tile
    static constexpr int PPI = 64 / G8;           // pixels covered per pass of 64 lanes
    static constexpr int PASSES = 16 / PPI;
    long long oe[FM][PASSES];                     // output element offset (channel ch of the pixel)
    long long re[FM][PASSES];                     // residual element offset
    bf16x8 r8[FM][PASSES];                        // bf16 residual values (zeros when there is no residual)
    unsigned okmask;                              // bit i*PASSES+ps: this lane stores that pixel
    int ch;
    bool full;                                    // ch..ch+7 all < N
};

template <int FM, int FN, bool SPLIT = false>
__device__ __forceinline__ void epilogue_prepare(EpiState<FM, FN>& st, const OutGeom& g, int m_wave0, int ch_wave0,
                                                 int lane) {
    using S = EpiState<FM, FN>;
    const int g8 = lane % S::G8, pr0 = lane / S::G8;
    st.ch = ch_wave0 + g8 * 8;
    st.full = st.ch + 7 < g.N;
    st.okmask = 0;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int ps = 0; ps < S::PASSES; ++ps) {
            const int m = m_wave0 + i * 16 + ps * S::PPI + pr0;
            const bool ok = m < g.M && st.ch < g.N;
            if (ok) st.okmask |= 1u << (i * S::PASSES + ps);
            const int mm = m < g.M ? m : g.M - 1;
            const int n = (int)fdiv(mm, g.dODHW), r1 = mm - n * (int)g.dODHW.d;
            const int d = (int)fdiv(r1, g.dOHW), r2 = r1 - d * (int)g.dOHW.d;
            const int h = (int)fdiv(r2, g.dOW), w_ = r2 - h * (int)g.dOW.d;
            st.oe[i][ps] = (long long)n * g.out_sn + (long long)d * g.out_sd + (long long)h * g.out_sh +
                           (long long)w_ * g.out_sw + g.out_off0 + st.ch;
            st.re[i][ps] = 0;
            st.r8[i][ps] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (g.res) {
                st.re[i][ps] = (long long)n * g.res_sn + (long long)d * g.res_sd + (long long)h * g.res_sh +
                               (long long)w_ * g.res_sw + g.res_off0 + st.ch;
                if (!SPLIT && ok && st.full && !g.res_f32) st.r8[i][ps] = *(const bf16x8*)(g.res + 2 * st.re[i][ps]);
            }
        }
}

// acc[i][j]: 16x16 tile (pixel fragment i, channel fragment j); lane holds channels 4*(lane>>4)..+3
// of pixel (lane & 15).  `stage` is this wave's private LDS region of kEpiWaveBytes(FN*16) bytes; the
// caller guarantees no other wave still reads the LDS it overlays.
// SPLIT (the parity mode's split-precision activations): the bf16 output is written as two halves, hi = bf16(v) at the
// usual place and lo = bf16(v - hi) out_lo elements further on, so hi + lo carries v to ~16 mantissa bits; a residual is
// read the same way (its lo half res_lo elements further on) when it is used, not ahead of the K loop.
// TOLDS: every 8-channel bf16 vector stored to memory is also written to a row-major LDS tile (row = the pixel's index
// within the workgroup's tile, lds_row0 = this wave's first row, lds_row_bytes per row, channel ch at byte 2 * ch), where
// a second GEMM of the same launch reads it as its operand; whole 8-channel groups only.
template <int FM, int FN, bool SPLIT = false, bool TOLDS = false>
__device__ __forceinline__ void epilogue_store(const f32x4 (&acc)[FM][FN], char* stage, const OutGeom& g,
                                               const EpiState<FM, FN>& st, int lane, char* lds_tile = nullptr,
                                               int lds_row0 = 0, int lds_row_bytes = 0) {
    using S = EpiState<FM, FN>;
    constexpr int ROWB = kEpiRowBytes(S::WTN);
    const int fr = lane & 15, fq = lane >> 4;
    const int g8 = lane % S::G8, pr0 = lane / S::G8;
    const int ch = st.ch;
    const bool full = st.full;
    f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (full) {
        if (g.scale) { s0 = *(const f32x4*)(g.scale + ch); s1 = *(const f32x4*)(g.scale + ch + 4); }
        if (g.bias) { b0 = *(const f32x4*)(g.bias + ch); b1 = *(const f32x4*)(g.bias + ch + 4); }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j) *(f32x4*)(stage + fr * ROWB + (j * 16 + fq * 4) * 4) = acc[i][j];
#pragma unroll
        for (int ps = 0; ps < S::PASSES; ++ps) {
            const int pr = ps * S::PPI + pr0;
            f32x4 v0 = *(const f32x4*)(stage + pr * ROWB + g8 * 32);
            f32x4 v1 = *(const f32x4*)(stage + pr * ROWB + g8 * 32 + 16);
            if (!((st.okmask >> (i * S::PASSES + ps)) & 1)) continue;
            const long long o = st.oe[i][ps];
            if (full) {
                v0 = v0 * s0 + b0;
                v1 = v1 * s1 + b1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] += bf16_to_f32(st.r8[i][ps][e]);
                    v1[e] += bf16_to_f32(st.r8[i][ps][4 + e]);
                }
                if (g.res && g.res_f32) {          // fp32 residual stream (small GEMMs only): loaded in place
                    v0 += *(const f32x4*)(g.res + 4 * st.re[i][ps]);
                    v1 += *(const f32x4*)(g.res + 4 * st.re[i][ps] + 16);
                }
                if constexpr (SPLIT) {
                    if (g.res && !g.res_f32) {
                        const bf16x8 rh = *(const bf16x8*)(g.res + 2 * st.re[i][ps]);
                        const bf16x8 rl = *(const bf16x8*)(g.res + 2 * (st.re[i][ps] + g.res_lo));
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v0[e] += bf16_to_f32(rh[e]) + bf16_to_f32(rl[e]);
                            v1[e] += bf16_to_f32(rh[4 + e]) + bf16_to_f32(rl[4 + e]);
                        }
                    }
                }
                if (g.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                }
                if (g.out_f32) {
                    *(f32x4*)(g.y + 4 * o) = v0;
                    *(f32x4*)(g.y + 4 * o + 16) = v1;
                } else {
                    bf16x8 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ov[e] = f32_to_bf16(v0[e]); ov[4 + e] = f32_to_bf16(v1[e]); }
                    *(bf16x8*)(g.y + 2 * o) = ov;
                    if constexpr (TOLDS) *(bf16x8*)(lds_tile + (lds_row0 + i * 16 + pr) * lds_row_bytes + 2 * ch) = ov;
                    if constexpr (SPLIT) {
                        bf16x8 lv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            lv[e] = f32_to_bf16(v0[e] - bf16_to_f32(ov[e]));
                            lv[4 + e] = f32_to_bf16(v1[e] - bf16_to_f32(ov[4 + e]));
                        }
                        *(bf16x8*)(g.y + 2 * (o + g.out_lo)) = lv;
                    }
                }
            } else {
                for (int e = 0; e < 8 && ch + e < g.N; ++e) {
                    float t = e < 4 ? v0[e & 3] : v1[e & 3];
                    if (g.scale) t *= g.scale[ch + e];
                    if (g.bias) t += g.bias[ch + e];
                    if (g.res) t += g.res_f32 ? *(const float*)(g.res + 4 * (st.re[i][ps] + e))
                                              : bf16_to_f32(*(const short*)(g.res + 2 * (st.re[i][ps] + e)));
                    if (SPLIT && g.res && !g.res_f32) t += bf16_to_f32(*(const short*)(g.res + 2 * (st.re[i][ps] + g.res_lo + e)));
                    if (g.relu) t = fmaxf(t, 0.f);
                    if (g.out_f32) *(float*)(g.y + 4 * (o + e)) = t;
                    else {
                        const short hv = f32_to_bf16(t);
                        *(short*)(g.y + 2 * (o + e)) = hv;
                        if (SPLIT) *(short*)(g.y + 2 * (o + g.out_lo + e)) = f32_to_bf16(t - bf16_to_f32(hv));
                    }
                }
            }
        }
    }
}

}  // namespace asl_detail
