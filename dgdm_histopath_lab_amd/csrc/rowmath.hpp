// Activation functions, their derivatives and the counter-based dropout mask shared by the fused
// row kernels (rownorm.hip, elementwise.hip).
#pragma once
#include "common.hpp"

__device__ __forceinline__ uint32_t hash32(uint32_t x) {  // lowbias32
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// keep-scale of 4 consecutive elements starting at element index e (multiple of 4)
__device__ __forceinline__ float4 dropout_scale4(uint32_t seed, uint64_t e, uint32_t thresh16, float keep_scale) {
  const uint32_t lo = (uint32_t)(e >> 2), hi = (uint32_t)(e >> 34);
  const uint32_t a = hash32(lo * 0x9E3779B1U ^ seed ^ (hi * 0x85EBCA6BU));
  const uint32_t b = hash32(a ^ 0x68E31DA4U);
  float4 r;
  r.x = (a & 0xFFFFu) >= thresh16 ? keep_scale : 0.f;
  r.y = (a >> 16) >= thresh16 ? keep_scale : 0.f;
  r.z = (b & 0xFFFFu) >= thresh16 ? keep_scale : 0.f;
  r.w = (b >> 16) >= thresh16 ? keep_scale : 0.f;
  return r;
}

// ---- GELU (exact-erf form, nn.GELU() of the reference: encoders.py:60,205, graph_layers.py:158) without a library erf.
// ocml's erff is two polynomial ranges behind a divergent branch: ~45 vector instructions per value once both sides run, which made
// every GELU site of the path VALU-bound (k_act_dropout: 11 us for 40k x 128 values that HBM moves in 5).  GELU only needs
//   Phi(z) = 0.5 erfc(-z / sqrt 2),   and with t = |z| / sqrt 2:   0.5 erfc(t) = exp2(-(1 + t q(t)))
// where t q(t) = -log2(erfc(t)) is smooth; q is a degree-8 polynomial fitted on [0, 4] (weighted for the absolute error of erf;
// approximation error 2e-9, fp32 evaluation error 8e-8 = one ulp of erf near 1; t is clamped at 4, where 0.5 erfc = 8e-9).
// One range, no branch: 8 FMAs, one v_exp_f32.  The NEGATIVE side uses h = 0.5 erfc(t) directly instead of 1 - (1 - h), so
// gelu(z) for z < 0 keeps full relative accuracy (torch's fp32 GELU loses it there).  Against float64 over [-12, 12]
// (tools/check_fast_gelu.py): gelu max abs error 2.4e-7 (torch fp32: 1.2e-6), gelu' 1.4e-7 (torch: 2.8e-7).
__device__ __forceinline__ float half_erfc_abs(float z) {          // 0.5 * erfc(|z| / sqrt(2)); exactly 0 from |z| = 4 sqrt(2) on
  const float ta = fabsf(z) * 0.70710678118654752f;
  const float t = fminf(ta, 4.0f);
  float r = -1.1622888450801838e-05f;
  r = fmaf(r, t, 0.00015313828771468252f);
  r = fmaf(r, t, -0.0008489217725582421f);
  r = fmaf(r, t, 0.00227622059173882f);
  r = fmaf(r, t, -8.649988012621179e-05f);
  r = fmaf(r, t, -0.02772335335612297f);
  r = fmaf(r, t, 0.14830751717090607f);
  r = fmaf(r, t, 0.918442964553833f);
  r = fmaf(r, t, 1.6279072761535645f);
  // past the clamp the polynomial's value would stay at 0.5 erfc(4) = 7.7e-9 for ever (gelu(-1e4) = -7.7e-5, gelu' with a floor):
  // h is 0 there, a step of 4e-8 at |z| = 5.66.  !(ta < 4) also takes NaN through as 0 -- the callers' products keep it.
  return ta < 4.0f ? __builtin_amdgcn_exp2f(-fmaf(t, r, 1.0f)) : 0.0f;
}
__device__ __forceinline__ float gelu_f(float z) {                 // z >= 0: z - z h;  z < 0: z h   (h = 0.5 erfc(|z| / sqrt 2))
  // one rounding on the positive side (fma); with h = 0 past the clamp: gelu(1e4) = 1e4, gelu(-1e4) = -0, and +-inf give NaN (inf * 0),
  // which is what torch's fp32 GELU returns for them as well (tools/check_fast_gelu.py prints the table)
  const float h = half_erfc_abs(z);
  return z >= 0.f ? fmaf(-z, h, z) : z * h;
}
__device__ __forceinline__ float gelu_df(float z) {                // Phi(z) + z phi(z)
  const float h = half_erfc_abs(z);
  const float Phi = z >= 0.f ? 1.0f - h : h;
  return fmaf(z * 0.3989422804014327f, __builtin_amdgcn_exp2f(z * z * -0.72134752044448170f), Phi);
}

#ifdef DGDM_GELU_ERFF   // diagnostic builds only (tools/build_variant_lib.sh erff -DDGDM_GELU_ERFF): the library erf of rounds 1-4
#define gelu_f(z) (0.5f * (z) * (1.0f + erff((z) * 0.70710678118654752f)))
#define gelu_df(z) (0.5f * (1.0f + erff((z) * 0.70710678118654752f)) + (z) * 0.3989422804014327f * __expf(-0.5f * (z) * (z)))
#endif

template <int ACT>
__device__ __forceinline__ float act_f(float z) {
  if (ACT == DGDM_ACT_GELU) return gelu_f(z);
  if (ACT == DGDM_ACT_RELU) return fmaxf(z, 0.f);
  if (ACT == DGDM_ACT_SILU) return z / (1.0f + __expf(-z));
  if (ACT == DGDM_ACT_ELU) return z > 0.f ? z : expm1f(z);
  return z;
}
template <int ACT>
__device__ __forceinline__ float act_df(float z) {
  if (ACT == DGDM_ACT_GELU) return gelu_df(z);
  if (ACT == DGDM_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (ACT == DGDM_ACT_SILU) {
    const float s = 1.0f / (1.0f + __expf(-z));
    return s * (1.0f + z * (1.0f - s));
  }
  if (ACT == DGDM_ACT_ELU) return z > 0.f ? 1.f : __expf(z);
  return 1.f;
}
