// Exact three-way bf16 operand split for the fp32 kernels that run on the bf16 matrix pipe (wino_f32.hip, wino_wgrad_f32.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace mgu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// exact three-way split of two fp32 values into packed bf16 pieces (low half: a, high half: b)
__device__ __forceinline__ void split3_pack(const float a, const float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  p0 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
  const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
  p1 = __builtin_amdgcn_perm(vb, va, 0x07060302u);
  const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);
  p2 = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}
// One-instruction fp32 arithmetic the backend cannot pair into v_pk_add_f32 / v_pk_fma_f32: beside a dense MFMA stream the packed
// forms are slow (wino_f32.hip: +2.5 % on the whole forward; the Winograd weight gradient ran 2 x slower with a packed split).
__device__ __forceinline__ float x3_add(float a, float b) {
  float d;
  asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float x3_sub(float a, float b) {
  float d;
  asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float x3_fma(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// split3_pack with those subtractions
__device__ __forceinline__ void split3_pack_s(const float a, const float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  p0 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  const float ra = x3_sub(a, __uint_as_float(ua & 0xffff0000u)), rb = x3_sub(b, __uint_as_float(ub & 0xffff0000u));
  const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
  p1 = __builtin_amdgcn_perm(vb, va, 0x07060302u);
  const float sa = x3_sub(ra, __uint_as_float(va & 0xffff0000u)), sb = x3_sub(rb, __uint_as_float(vb & 0xffff0000u));
  p2 = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}

template <int I, int N, class F>
__device__ __forceinline__ void x3_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    x3_static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

}  // namespace mgu
