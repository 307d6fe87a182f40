// fp32 operands on the bf16 matrix pipe: the exact 3-way split x = hi + mid + lo into bf16 pieces (gemm_wres.hip explains the
// arithmetic; gemm_wres.hip and gemm_dw4.hip use it), and a compile-time loop for hand-dealt instruction schedules.
#pragma once
#include <utility>

#include "common.h"

namespace refil {

// f(integral_constant<int, 0>) .. f(integral_constant<int, N - 1>): a loop whose index is a constant expression in the body
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

typedef __bf16 wr_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wr_bf16x8 __attribute__((ext_vector_type(8)));
typedef float wr_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned wr_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline unsigned wr_pk(float x, float y) { wr_f32x2 v = {x, y}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wr_bf16x2)); }
// a - b as ONE scalar v_sub_f32: left to the compiler, the two residuals of a pair are SLP-packed into a v_pk_add_f32, which costs
// ~13 matrix-pipe cycles beside an MFMA where a plain VALU operation costs none (MI355X_MICROARCH.md, "price of one filler")
__device__ inline float wr_sub(float a, float b) { float r; asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// (x, y) -> packed bf16 pairs hi / mid / lo with x = hi + mid + lo (+ < 2^-25 |x|)
__device__ inline void wr_split(float x, float y, unsigned& h, unsigned& m, unsigned& l) {
    h = wr_pk(x, y);
    x -= __uint_as_float(h << 16); y -= __uint_as_float(h & 0xFFFF0000u);
    m = wr_pk(x, y);
    x -= __uint_as_float(m << 16); y -= __uint_as_float(m & 0xFFFF0000u);
    l = wr_pk(x, y);
}

}  // namespace refil
