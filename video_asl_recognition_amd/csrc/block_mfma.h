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
//
// What the compiler cannot see inside these statements, and what each rule below is there for:
//  * An MFMA's result may be read by a vector instruction only 18 wait states after a 16-pass MFMA issued.  The
//    compiler's hazard pass inserts those for MFMAs it emitted itself; for an MFMA inside an asm string it inserts
//    nothing.  So the tile's LAST statement carries the wait states itself and names, as read-write operands, EVERY
//    accumulator that has an MFMA in flight: whatever the compiler does with them afterwards (the sum of the two
//    chains, a copy) is then ordered behind the wait.
//  * The same accumulator must never be bound to two operands of one statement.  `"+v"(acc), "+v"(acc)` -- which is
//    what the two-chain last statement becomes when a one-chain loop passes its only accumulator as both -- makes
//    the compiler materialise a second copy of the variable: sixteen v_mov right after the previous MFMA, with no
//    wait states (partly written accumulators are copied), and the copy, which the MFMA never touches, is what the
//    code after the statement reads.  That is the whole cause of the "single accumulation chain gives wrong sums"
//    met while this kernel was written (disassembly: v_mfma; s_nop 0; 8 x v_mov_b64 v[0:15] <- v[16:31]; v_mfma
//    v[16:31]; ...; store from v[0:15]); a dependent chain through SrcC = vDst is handled by the hardware and is
//    as fast as two chains.  The API below makes the mistake unrepresentable: mfma_last2 takes the two-element
//    accumulator array (two distinct objects by construction), mfma_last1 takes one accumulator and names it once.
//  * An MFMA that finds the matrix pipe busy starts up to 32 cycles after it issues and reads its A/B operands
//    when it starts.  A fragment register must therefore not be refilled by the read issued right behind the MFMA
//    that consumes it (an LDS return can overtake the operand read).  lds_frag_after() keeps the fragment of the
//    MFMA just issued alive across the next read and marks the read's destination early-clobber, so the register
//    allocator cannot hand that register to the read: the ring of kAhead + 1 slots exists in registers, not only
//    in the source.
// kEdge 1: first MFMA of a chain (accumulates onto the constant 0: no register to clear); 0: any other but the last
template <bool kBankInAgprs, int kWait, int kEdge>
__device__ __forceinline__ void mfma_step(f32x16& acc, const bf16x8& w, const bf16x8& x) {
    static_assert(kEdge == 0 || kEdge == 1, "the last step of a tile goes through mfma_last1 / mfma_last2");
    if constexpr (kBankInAgprs) {
        if constexpr (kEdge == 1) asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(x), "n"(kWait));
        else asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x), "n"(kWait));
    } else {
        if constexpr (kEdge == 1) asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(x), "n"(kWait));
        else asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x), "n"(kWait));
    }
}

#define ASL_MFMA_LAST2 "s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\ts_nop 15\n\ts_nop 3"
#define ASL_MFMA_LAST1 "s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3"
// Last MFMA of a tile on two alternating chains: accumulates into acc[kChain]; both chains are fenced.
template <bool kBankInAgprs, int kWait, int kChain>
__device__ __forceinline__ void mfma_last2(f32x16 (&acc)[2], const bf16x8& w, const bf16x8& x) {
    static_assert(kChain == 0 || kChain == 1, "chain index");
    if constexpr (kBankInAgprs) asm volatile(ASL_MFMA_LAST2 : "+v"(acc[kChain]), "+v"(acc[kChain ^ 1]) : "a"(w), "v"(x), "n"(kWait));
    else asm volatile(ASL_MFMA_LAST2 : "+v"(acc[kChain]), "+v"(acc[kChain ^ 1]) : "v"(w), "v"(x), "n"(kWait));
}
// Last MFMA of a tile on ONE chain: the accumulator is named once.
template <bool kBankInAgprs, int kWait>
__device__ __forceinline__ void mfma_last1(f32x16& acc, const bf16x8& w, const bf16x8& x) {
    if constexpr (kBankInAgprs) asm volatile(ASL_MFMA_LAST1 : "+v"(acc) : "a"(w), "v"(x), "n"(kWait));
    else asm volatile(ASL_MFMA_LAST1 : "+v"(acc) : "v"(w), "v"(x), "n"(kWait));
}

template <int kOff>
__device__ __forceinline__ bf16x8 lds_frag(unsigned ad) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(ad), "n"(kOff));
    return v;
}

// The same read issued right behind an MFMA whose B fragment is `busy`: its destination is a different register.
template <int kOff>
__device__ __forceinline__ bf16x8 lds_frag_after(unsigned ad, const bf16x8& busy) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v) : "v"(ad), "n"(kOff), "v"(busy));
    return v;
}

}  // namespace asl_detail
