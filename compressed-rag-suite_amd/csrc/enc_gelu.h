// erf-form GELU for the GEMM epilogues (BERT's hidden_act = "gelu": x Phi(x) = 0.5 x (1 + erf(x / sqrt 2))).
//
// erf(z) = z P(z^2) / Q(z^2) on z clamped to [-4, 4]: the odd rational minimax fit (numerator degree 13, denominator
// degree 8 in z) that single-precision math libraries use for erff; measured against scipy's erf over [-6, 6] in fp32:
// |erf error| <= 4.5e-7, |gelu error| <= 1.4e-6 -- three orders below the fp16 rounding of the stored activation.
// No exponential and a single v_rcp_f32; all the rest is fused multiply-adds, done on TWO values per instruction
// (v_pk_fma_f32 / v_pk_mul_f32): ~9 full-rate operations and one quarter-rate per value.  The FFN-up epilogue applies
// this to 100 M values per MiniLM layer at index-build size, which with the previous form (Abramowitz & Stegun 7.1.26:
// exp + rcp + 13 scalar operations per value) took more VALU time than the layer's MFMAs; libm's erff is ~50
// instructions per value.
#pragma once
#include <hip/hip_runtime.h>

namespace crs {

typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ gelu_f32x2 gelu_erf2(gelu_f32x2 x) {
  const gelu_f32x2 zs = x * 0.70710678118654752440f;
  const gelu_f32x2 z = {__builtin_amdgcn_fmed3f(zs[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(zs[1], -4.0f, 4.0f)};
  const gelu_f32x2 z2 = z * z;
  gelu_f32x2 p = z2 * -2.72614225801306e-10f + 2.77068142495902e-08f;
  p = p * z2 + -2.10102402082508e-06f;
  p = p * z2 + -5.69250639462346e-05f;
  p = p * z2 + -7.34990630326855e-04f;
  p = p * z2 + -2.95459980854025e-03f;
  p = p * z2 + -1.60960333262415e-02f;
  p = p * z;
  gelu_f32x2 q = z2 * -1.45660718464996e-05f + -2.13374055278905e-04f;
  q = q * z2 + -1.68282697438203e-03f;
  q = q * z2 + -7.37332916720468e-03f;
  q = q * z2 + -1.42647390514189e-02f;
  const gelu_f32x2 r = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
  const gelu_f32x2 hx = x * 0.5f;
  return hx * (p * r) + hx;
}

__device__ __forceinline__ float gelu_erf(float x) {
  const gelu_f32x2 r = gelu_erf2(gelu_f32x2{x, x});
  return r[0];
}

}  // namespace crs
