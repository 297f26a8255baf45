// Pieces shared by the fused 64-channel BasicBlock kernels (block_c64.hip, block_c64_pair.hip): the chunk-major
// LDS image address, the hand-written MFMA step with its wait states, and the fragment read.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace asl_detail {

// byte (pixel p, 16-byte chunk c) of a chunk-major image lives at (p >> 4) * 2048 + c * 256 + (p & 15) * 16
__device__ __forceinline__ int img_addr(int p) { return ((p >> 4) << 11) | ((p & 15) << 4); }

// One 32-pixel x 32-channel tile of a 3x3 convolution over a chunk-major LDS image: 36 K steps of 16
// channels.  `u` is the image pixel of the lane's top-left tap, `lds` the image's LDS byte address plus
// the lane's k-half * 256.
// The K loop is written as inline assembly: the wave is alone on its SIMD, so nothing else hides an LDS read's
// latency or an instruction's issue slot, and at one wave per SIMD every vector instruction that does not fit
// beside an MFMA's 32 cycles is paid in full.  Fragments are read kAhead steps ahead of the MFMA that consumes
// them (LDS operations return in order, so "all but the N youngest" is the wait); the second filter bank is
// addressed in the accumulation registers directly (`a` operands) instead of being copied out before every use.
// The assembler-level hazards the compiler cannot see are covered by hand and sit inside the MFMA's own asm
// statement, where no compiler-inserted copy can slip past them: kEdge 1 = first MFMA of a tile (two wait
// states after the accumulator was zeroed), 2 = last (18 wait states before its result may be read).
#define ASL_MFMA_FIRST "s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0"
#define ASL_MFMA_BODY "s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
#define ASL_MFMA_LAST "s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\ts_nop 15\n\ts_nop 3"
// kEdge 1: first MFMA of a chain (accumulates onto the constant 0: no register to clear), 2: last MFMA of the tile
template <bool kBankInAgprs, int kWait, int kEdge>
__device__ __forceinline__ void mfma_step(f32x16& acc, f32x16& other, const bf16x8& w, const bf16x8& x) {
    if constexpr (kBankInAgprs) {
        if constexpr (kEdge == 1) asm volatile(ASL_MFMA_FIRST : "=&v"(acc) : "a"(w), "v"(x), "n"(kWait));
        else if constexpr (kEdge == 2) asm volatile(ASL_MFMA_LAST : "+v"(acc), "+v"(other) : "a"(w), "v"(x), "n"(kWait));
        else asm volatile(ASL_MFMA_BODY : "+v"(acc) : "a"(w), "v"(x), "n"(kWait));
    } else {
        if constexpr (kEdge == 1) asm volatile(ASL_MFMA_FIRST : "=&v"(acc) : "v"(w), "v"(x), "n"(kWait));
        else if constexpr (kEdge == 2) asm volatile(ASL_MFMA_LAST : "+v"(acc), "+v"(other) : "v"(w), "v"(x), "n"(kWait));
        else asm volatile(ASL_MFMA_BODY : "+v"(acc) : "v"(w), "v"(x), "n"(kWait));
    }
}

template <int kOff>
__device__ __forceinline__ bf16x8 lds_frag(unsigned ad) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(ad), "n"(kOff));
    return v;
}

}  // namespace asl_detail
