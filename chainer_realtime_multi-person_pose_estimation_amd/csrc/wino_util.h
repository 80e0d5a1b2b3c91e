// wino_util.h -- what the Winograd kernels share (conv_wino.hip: the 3x3 / 7x7 layers; conv1_wino.hip: conv1_1 + conv1_2): vector types,
// packed-fp32 adds with the subtrahend negated by the source modifier (same rounding as v_sub_f32), the transposed MFMA orientation.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_add2(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_sub2(f32x2 a, f32x2 b)
{
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x4 pk_add4(f32x4 a, f32x4 b)
{
    const f32x2 lo = pk_add2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_add2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 pk_sub4(f32x4 a, f32x4 b)
{
    const f32x2 lo = pk_sub2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_sub2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

// MFMA with the WEIGHT fragment as the matrix core's row operand: the accumulator tile comes out transposed -- lane li (+ 32 kh) holds
// Winograd tile / pixel li, register reg holds channel 8 (reg >> 2) + 4 kh + (reg & 3) of the wave's 32: four CONSECUTIVE channels per
// register quad, i.e. 16-byte channel runs per lane (conv1_wino.hip: conv1_1's tile goes to LDS that way, conv1_2's pooled outputs to
// memory).  The products and the order in which the two k's are added are those of the other orientation.  (For the big un-pooled outputs
// of conv_wino_kernel this orientation measured 1 - 3 % SLOWER: a store instruction then touches 32 pixels x 32 bytes instead of two
// 128-byte runs -- profiles/r05_store_ablation.json.)
__device__ __forceinline__ f32x16 wino_mfma(float tiles, float wts, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(wts, tiles, c, 0, 0, 0);
}
