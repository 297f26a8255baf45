// Shared device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(2))) short bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float bf16_to_f32(short v) {
    return __builtin_bit_cast(float, (unsigned)(unsigned short)v << 16);
}

// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ short f32_to_bf16(float f) {
    return __builtin_bit_cast(short, (__bf16)f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Division of a 32-bit unsigned by a divisor fixed at launch: multiply-high + shifts instead of the
// ~40-instruction hardware-less integer division (Granlund-Montgomery round-up method, exact for all n).
struct FastDiv {
    unsigned d, mul, s1, s2;
};

inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (unsigned)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
    f.s1 = l < 1 ? l : 1;
    f.s2 = l < 1 ? 0 : l - 1;
    return f;
}

__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
    const unsigned t = __umulhi(f.mul, n);
    return (t + ((n - t) >> f.s1)) >> f.s2;
}
