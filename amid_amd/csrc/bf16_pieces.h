// An fp32 value as three bf16 pieces whose sum is the value exactly: the arithmetic behind the fp32 products on the bf16 matrix cores
// (csrc/wgrad_split.h: weight gradients; csrc/seqn_parts.h: the one-launch forward's projections).
#pragma once
#include "common.h"

namespace amid {

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned wg_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned wg_pack2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(wg_f32x2{a, b}, wg_bf16x2));       // v_cvt_pk_bf16_f32: round to nearest even
}
// x = hi + mid + lo, each piece rounded to nearest even: 3 x 8 significand bits plus the remainders' signs cover fp32's 24
struct WgSplit2 { unsigned hi, mid, lo; };          // two elements, packed bf16 pairs (first element in the low half)
__device__ __forceinline__ WgSplit2 wg_split3(float a, float b) {
    WgSplit2 s;
    s.hi = wg_pack2(a, b);
    const float a1 = a - __uint_as_float(s.hi << 16), b1 = b - __uint_as_float(s.hi & 0xffff0000u);       // exact
    s.mid = wg_pack2(a1, b1);
    const float a2 = a1 - __uint_as_float(s.mid << 16), b2 = b1 - __uint_as_float(s.mid & 0xffff0000u);   // exact
    s.lo = wg_pack2(a2, b2);
    return s;
}

}  // namespace amid
