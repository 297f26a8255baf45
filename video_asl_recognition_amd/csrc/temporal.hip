// Temporal-encoder kernels for gfx950: fused BiLSTM recurrence, LayerNorm, fused softmax
// attention (MFMA QK^T / PV, shuffle reductions), and the small row-wise helpers around them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "asl_hip.h"
#include "common.h"

namespace asl_detail {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }
__device__ __forceinline__ float tanhf_(float v) {
    // tanh(v) = 1 - 2 / (exp(2v) + 1); exact limits at +-inf, no cancellation near 0 beyond fp32 ulp
    const float e = __expf(2.0f * v);
    return 1.0f - 2.0f / (e + 1.0f);
}

// ---------------------------------------------------------------------------------------------
// One time step of one bidirectional LSTM layer, both directions, on the matrix cores (hidden % 32 == 0):
// wave = (direction, four consecutive hidden units).
// The A operand is the 16 x K slab of W_hh holding the four gates of those units, rows permuted so that MFMA
// row 4u + g is gate g of unit u; the B operand is h_prev of 16 batch rows, kept as a bf16 pair hi + lo
// (h to ~16 mantissa bits), so every product is exact and the sums are fp32.  The accumulator layout then puts
// all four gate pre-activations of (unit fq, batch row fr) in ONE lane: gate math and state update need no
// cross-lane traffic at all, and a step is one round of 16-byte loads, 2 * K/32 MFMAs and a few stores.
// ---------------------------------------------------------------------------------------------
struct LstmMfmaArgs {
    const float* gx;       // [B*T][ld_gx]
    const short* whh;      // [2][4H][H] bf16
    const short* h_prev;   // [2 (hi, lo)][2 (direction)][B][H] bf16
    short* h_next;         // same layout
    float* c;              // [2][B][H]
    short* y;
    float* ysum;
    int B, T, H, ld_gx, ld_y, step, split_y;
    float ysum_scale;
};

__global__ __launch_bounds__(64) void lstm_step_mfma_kernel(const LstmMfmaArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = threadIdx.x, fr = lane & 15, fq = lane >> 4;
    const int H = a.H, groups = H >> 2;
    const int d = blockIdx.x / groups, j0 = (blockIdx.x - d * groups) * 4;
    const int t = d == 0 ? a.step : a.T - 1 - a.step;
    // A fragment of k-step ks: row fr = (unit fr >> 2, gate fr & 3), elements ks*32 + fq*8 .. +7
    const short* wrow = a.whh + ((long long)d * 4 * H + (long long)(fr & 3) * H + j0 + (fr >> 2)) * H + fq * 8;
    const long long lo_off = 2LL * a.B * H;                     // hi plane -> lo plane
    const int ksteps = H >> 5;
    for (int b0 = 0; b0 < a.B; b0 += 16) {
        const int bcol = b0 + fr;                               // this lane's batch row as B-operand column
        const bool colok = bcol < a.B;
        const short* hrow = a.h_prev + ((long long)d * a.B + (colok ? bcol : 0)) * H + fq * 8;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // the input projection's four gate terms and the cell state of this lane's (unit, batch row) are asked for first:
        // loaded after the MFMAs they would be one more memory round trip on the step's dependency chain
        const int bq = colok ? bcol : 0;
        const float* gxp = a.gx + ((long long)bq * a.T + t) * a.ld_gx + d * 4 * H + j0 + fq;
        const long long si = ((long long)d * a.B + bq) * H + j0 + fq;
        const float gx0 = gxp[0], gx1 = gxp[H], gx2 = gxp[2 * H], gx3 = gxp[3 * H];
        const float c_prev = a.c[si];
        // eight k-steps of loads (24 x 16 bytes per lane) are issued before their MFMAs: the step is a latency
        // chain, so it wants every load of a batch in flight at once
        for (int ks0 = 0; ks0 < ksteps; ks0 += 8) {
            bf16x8 wf[8], hh[8], hl[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ks = ks0 + u < ksteps ? ks0 + u : ksteps - 1;     // clamp: in-range, result unused
                wf[u] = *(const bf16x8*)(wrow + ks * 32);
                hh[u] = *(const bf16x8*)(hrow + ks * 32);
                hl[u] = *(const bf16x8*)(hrow + lo_off + ks * 32);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (ks0 + u < ksteps) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], hh[u], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], hl[u], acc, 0, 0, 0);
                }
            }
        }
        // lane (fr, fq) holds gates i, f, g, o of unit j0 + fq for batch row b0 + fr
        if (colok) {
            const int b = bcol, j = j0 + fq;
            const float gi = sigmoidf_(acc[0] + gx0);
            const float gf = sigmoidf_(acc[1] + gx1);
            const float gg = tanhf_(acc[2] + gx2);
            const float go = sigmoidf_(acc[3] + gx3);
            const float cn = gf * c_prev + gi * gg;
            const float hn = go * tanhf_(cn);
            a.c[si] = cn;
            const short hi = f32_to_bf16(hn);
            const short lo = f32_to_bf16(hn - bf16_to_f32(hi));
            a.h_next[si] = hi;
            a.h_next[si + lo_off] = lo;
            short* yr = a.y + ((long long)b * a.T + t) * a.ld_y + d * H + j;
            yr[0] = hi;
            if (a.split_y) yr[2 * H] = lo;
            if (a.ysum) a.ysum[(long long)b * 2 * H + d * H + j] += hn * a.ysum_scale;
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dimension, one wave per row, fp32 in, bf16 and/or fp32 out.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int rows, int d,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        short* __restrict__ yb, int ldb, float* __restrict__ yf) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (long long)row * d;
    float s = 0.f;
    for (int k = lane * 4; k < d; k += 256) {
        const f32x4 v = *(const f32x4*)(xr + k);
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int k = lane * 4; k < d; k += 256) {
        const f32x4 v = *(const f32x4*)(xr + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) q += (v[e] - mean) * (v[e] - mean);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
    for (int k = lane * 4; k < d; k += 256) {
        const f32x4 v = *(const f32x4*)(xr + k);
        const f32x4 g = *(const f32x4*)(gamma + k), b = *(const f32x4*)(beta + k);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * g[e] + b[e];
        if (yf) *(f32x4*)(yf + (long long)row * d + k) = o;
        if (yb) {
            bf16x4 ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) ob[e] = f32_to_bf16(o[e]);
            *(bf16x4*)(yb + (long long)row * ldb + k) = ob;
        }
    }
}

// y = x + add[row % period]  (fp32), optional bf16 copy: the positional-encoding add
__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ x, int rows, int d,
                                                       const float* __restrict__ add, int period,
                                                       float* __restrict__ yf, short* __restrict__ yb, int ldb) {
    const long long total = (long long)rows * (d / 4);
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / (d / 4);
        const int k = (int)(i - row * (d / 4)) * 4;
        f32x4 v = *(const f32x4*)(x + row * d + k);
        const f32x4 p = *(const f32x4*)(add + (row % period) * d + k);
        v += p;
        if (yf) *(f32x4*)(yf + row * d + k) = v;
        if (yb) {
            bf16x4 ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) ob[e] = f32_to_bf16(v[e]);
            *(bf16x4*)(yb + row * ldb + k) = ob;
        }
    }
}

// fp32 [rows][k] -> bf16 [rows][ld] with columns >= k zeroed (GEMM K padding)
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ x, int rows, int k, short* __restrict__ y,
                                                       int ld) {
    const long long total = (long long)rows * ld;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / ld;
        const int c = (int)(i - row * ld);
        y[i] = c < k ? f32_to_bf16(x[row * k + c]) : (short)0;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused softmax attention for short sequences (T <= 128, head_dim 64): one workgroup of 4 waves per
// (batch, head); wave w owns query rows [32w, 32w+32).  S = Q K^T by MFMA with keys on the
// accumulator rows, so a lane holds 4 consecutive keys of one query per tile and the softmax
// row reductions are two lane shuffles (xor 16, xor 32) plus in-register work.  P goes through LDS
// (bf16) to become the A operand of O = P V; V is staged transposed so its fragments are 16-byte
// reads as well.
// ---------------------------------------------------------------------------------------------
struct AttnArgs {
    const short* qkv;   // [B*T][ld_qkv] bf16: q | k | v, each heads*64 wide
    short* out;         // [B*T][ld_out] bf16
    int B, T, heads, ld_qkv, ld_out;
    float scale;
};

constexpr int kAttnT = 128, kAttnD = 64;

__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    // LDS: K [128][64] (swizzled 16-B chunks), V^T [64][128+8], P [4 waves][32][128+8]
    __shared__ __attribute__((aligned(16))) short k_lds[kAttnT * kAttnD];
    __shared__ __attribute__((aligned(16))) short vt_lds[kAttnD * (kAttnT + 8)];
    __shared__ __attribute__((aligned(16))) short p_lds[4 * 32 * (kAttnT + 8)];
    constexpr int VTP = kAttnT + 8, PP = kAttnT + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int D = a.heads * kAttnD;
    const short* qb = a.qkv + (long long)b * a.T * a.ld_qkv + h * kAttnD;
    const short* kb = qb + D;
    const short* vb = qb + 2 * D;
    const int T = a.T;

    // stage K (row-major, chunk-swizzled) and V^T; rows >= T are zero
    for (int i = tid; i < kAttnT * 8; i += 256) {
        const int row = i >> 3, ch = i & 7;
        bf16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = kv;
        if (row < T) {
            kv = *(const bf16x8*)(kb + (long long)row * a.ld_qkv + ch * 8);
            vv = *(const bf16x8*)(vb + (long long)row * a.ld_qkv + ch * 8);
        }
        *(bf16x8*)(k_lds + row * kAttnD + ((ch ^ ((row >> 1) & 7)) << 3)) = kv;
#pragma unroll
        for (int e = 0; e < 8; ++e) vt_lds[(ch * 8 + e) * VTP + row] = vv[e];
    }
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    // Q fragments straight from global: B operand (columns = queries): lane -> query fr, k chunk fq
    bf16x8 qf[2][2];                            // [query tile][k step]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = wave * 32 + qt * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[qt][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (q < T) qf[qt][ks] = *(const bf16x8*)(qb + (long long)q * a.ld_qkv + (ks * 4 + fq) * 8);
        }
    }
    // S^T tiles: rows = keys (8 tiles of 16), columns = queries (2 tiles of 16)
    f32x4 s[8][2];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        const int row = kt * 16 + fr;
        bf16x8 kf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            kf[ks] = *(const bf16x8*)(k_lds + row * kAttnD + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 3));
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[qt][ks], acc, 0, 0, 0);
            s[kt][qt] = acc;
        }
    }
    // lane holds, for query (qt, fr), keys kt*16 + fq*4 + e.  Softmax over keys:
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = kt * 16 + fq * 4 + e;
                const float v = key < T ? s[kt][qt][e] * a.scale : -INFINITY;
                s[kt][qt][e] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = __expf(s[kt][qt][e] - mx);
                s[kt][qt][e] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        // P[q][key] (bf16) into this wave's LDS slab: 4 consecutive keys per store
        short* prow = p_lds + (wave * 32 + qt * 16 + fr) * PP;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            bf16x4 pv;
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[e] = f32_to_bf16(s[kt][qt][e] * inv);
            *(bf16x4*)(prow + kt * 16 + fq * 4) = pv;
        }
    }
    // O^T = V^T P^T: A operand = V^T rows (head dims), B operand = P rows (queries); k = keys
    f32x4 o[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {             // 4 steps of 32 keys
        bf16x8 pf[2], vf[4];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
            pf[qt] = *(const bf16x8*)(p_lds + (wave * 32 + qt * 16 + fr) * PP + ks * 32 + fq * 8);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vf[dt] = *(const bf16x8*)(vt_lds + (dt * 16 + fr) * VTP + ks * 32 + fq * 8);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
                o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt], pf[qt], o[dt][qt], 0, 0, 0);
    }
    // lane holds head dims dt*16 + fq*4 + e of query (qt, fr): 8-byte stores
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = wave * 32 + qt * 16 + fr;
        if (q >= T) continue;
        short* orow = a.out + ((long long)b * T + q) * a.ld_out + h * kAttnD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = f32_to_bf16(o[dt][qt][e]);
            *(bf16x4*)(orow + dt * 16 + fq * 4) = ov;
        }
    }
#endif
}

int grid_rows(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace asl_detail
using namespace asl_detail;

// Zero-fill of the recurrent state and the running sum as a kernel of the library, not hipMemsetAsync: a memset node in
// a captured HIP graph zeroes on the first replay only on this ROCm release (later replays write an arbitrary byte
// pattern), and the LSTM entry points are meant to be capturable.  p: n4 16-byte units; q: m floats (any alignment).
namespace asl_detail {
__global__ __launch_bounds__(256) void zero_fill_kernel(f32x4* __restrict__ p, long long n4, float* __restrict__ q, long long m) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) p[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    else if (i - n4 < m) q[i - n4] = 0.f;
}
}  // namespace asl_detail

// Shared body of the two LSTM entry points: validates, zeroes the state, issues one launch per step.
static int lstm_run(const asl_lstm_params* p, const float* gx, const void* w_hh, void* y_bf16, float* ysum,
                    float* state, hipStream_t s) {
    if (!p || !gx || !w_hh || !y_bf16 || !state) return ASL_EINVAL;
    if (p->batch <= 0 || p->t <= 0 || p->hidden <= 0 || (p->hidden & 7)) return ASL_EINVAL;
    if (p->hidden & 31) return ASL_ENOTSUP;      // a valid layer, but the recurrence kernel steps K by 32
    if (p->ld_gx < 8 * p->hidden || p->ld_y < (p->split_y ? 4 : 2) * p->hidden) return ASL_EINVAL;
    if (((uintptr_t)w_hh & 15) || ((uintptr_t)state & 15)) return ASL_EINVAL;
    const long long n = 2LL * p->batch * p->hidden;           // elements of one [2][B][H] state array
    // state = h ping | h pong | c ; zero initial state, and the running sum if requested (n is a multiple of 64)
    {
        const long long n4 = 3 * n / 4, m = ysum ? n : 0;
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n4 + m + 255) / 256)), dim3(256), 0, s, (f32x4*)state, n4, ysum, m);
    }
    {
        // the two h arrays of `state` hold bf16 (hi | lo) planes of [2][B][H] each: 2*B*H floats' worth of bytes
        LstmMfmaArgs m;
        m.gx = gx; m.whh = (const short*)w_hh; m.c = state + 2 * n; m.y = (short*)y_bf16; m.ysum = ysum;
        m.B = p->batch; m.T = p->t; m.H = p->hidden; m.ld_gx = p->ld_gx; m.ld_y = p->ld_y;
        m.ysum_scale = p->ysum_scale; m.split_y = p->split_y;
        const unsigned blocks = (unsigned)(2 * (p->hidden / 4));       // one wave per (direction, four units)
        for (int step = 0; step < p->t; ++step) {
            m.h_prev = (const short*)(state + (step & 1) * n);
            m.h_next = (short*)(state + ((step + 1) & 1) * n);
            m.step = step;
            hipLaunchKernelGGL(lstm_step_mfma_kernel, dim3(blocks), dim3(64), 0, s, m);
        }
    }
    return (int)hipGetLastError();
}

extern "C" int asl_lstm_bidir_layer(const asl_lstm_params* p, const float* gx, const void* w_hh, void* y_bf16,
                                    float* ysum, float* state, asl_stream_t stream) {
    return lstm_run(p, gx, w_hh, y_bf16, ysum, state, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// Allocation-free variant: the recurrent state (h ping | h pong | c) lives in a device array that is
// part of the library's code object, so the caller brings no scratch and nothing here allocates.
// Every GPU the library runs on holds its own copy of the array (1.5 MiB) at its own address; the
// address is looked up once per device, on the first call made with that device current, and only
// that looked-up pointer is ever handed to a kernel -- the array's name in host
// code is not a device address.
// ---------------------------------------------------------------------------------------------
static __device__ __align__(256) float scratch[ASL_LSTM_STATIC_STATE_FLOATS];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
float* g_scratch_ptr[kMaxDevices];        // per device, null until that device's first call

int scratch_of_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= kMaxDevices) return ASL_EINVAL;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        float* ptr = nullptr;
        e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!ptr) return ASL_EINVAL;
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return ASL_OK;
}
}  // namespace

extern "C" int asl_lstm_bidir_layer_static(const asl_lstm_params* p, const float* gx, const void* w_hh,
                                           void* y_bf16, float* ysum, asl_stream_t stream) {
    if (!p) return ASL_EINVAL;
    if (p->batch <= 0 || p->hidden <= 0) return ASL_EINVAL;
    if (6LL * p->batch * p->hidden > (long long)ASL_LSTM_STATIC_STATE_FLOATS) return ASL_ETOOBIG;
    float* state = nullptr;
    const int rc = scratch_of_current_device(&state);
    if (rc != ASL_OK) return rc;
    return lstm_run(p, gx, w_hh, y_bf16, ysum, state, (hipStream_t)stream);
}

extern "C" int asl_layernorm_f32(const float* x, int rows, int d, const float* gamma, const float* beta, float eps,
                                 void* y_bf16, int ld_bf16, float* y_f32, asl_stream_t stream) {
    if (!x || !gamma || !beta || rows <= 0 || d <= 0 || (d & 3) || (!y_bf16 && !y_f32)) return ASL_EINVAL;
    if (y_bf16 && (ld_bf16 < d || (ld_bf16 & 3))) return ASL_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15) || ((uintptr_t)y_f32 & 15) ||
        ((uintptr_t)y_bf16 & 7))
        return ASL_EINVAL;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, rows, d,
                       gamma, beta, eps, (short*)y_bf16, ld_bf16, y_f32);
    return (int)hipGetLastError();
}

extern "C" int asl_add_rows_f32(const float* x, int rows, int d, const float* add, int period, float* y_f32,
                                void* y_bf16, int ld_bf16, asl_stream_t stream) {
    if (!x || !add || rows <= 0 || d <= 0 || (d & 3) || period <= 0 || (!y_bf16 && !y_f32)) return ASL_EINVAL;
    if (y_bf16 && (ld_bf16 < d || (ld_bf16 & 3))) return ASL_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)add & 15) || ((uintptr_t)y_f32 & 15) || ((uintptr_t)y_bf16 & 7)) return ASL_EINVAL;
    hipLaunchKernelGGL(add_rows_kernel, dim3(grid_rows((long long)rows * (d / 4))), dim3(256), 0, (hipStream_t)stream, x,
                       rows, d, add, period, y_f32, (short*)y_bf16, ld_bf16);
    return (int)hipGetLastError();
}

extern "C" int asl_cast_pad_bf16(const float* x, int rows, int k, void* y_bf16, int ld, asl_stream_t stream) {
    if (!x || !y_bf16 || rows <= 0 || k <= 0 || ld < k) return ASL_EINVAL;
    hipLaunchKernelGGL(cast_pad_kernel, dim3(grid_rows((long long)rows * ld)), dim3(256), 0, (hipStream_t)stream, x, rows,
                       k, (short*)y_bf16, ld);
    return (int)hipGetLastError();
}

extern "C" int asl_attention_fwd(const asl_attn_params* p, const void* qkv, void* out, asl_stream_t stream) {
    if (!p || !qkv || !out) return ASL_EINVAL;
    if (p->batch <= 0 || p->heads <= 0 || p->t <= 0 || p->t > kAttnT || p->head_dim != kAttnD) return ASL_EINVAL;
    if (p->ld_qkv < 3 * p->heads * kAttnD || (p->ld_qkv & 7) || p->ld_out < p->heads * kAttnD || (p->ld_out & 3))
        return ASL_EINVAL;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 7)) return ASL_EINVAL;
    AttnArgs a;
    a.qkv = (const short*)qkv; a.out = (short*)out; a.B = p->batch; a.T = p->t; a.heads = p->heads;
    a.ld_qkv = p->ld_qkv; a.ld_out = p->ld_out; a.scale = 1.0f / sqrtf((float)kAttnD);
    hipLaunchKernelGGL(attention_kernel, dim3((unsigned)(p->batch * p->heads)), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
