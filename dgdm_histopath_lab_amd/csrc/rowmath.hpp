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

template <int ACT>
__device__ __forceinline__ float act_f(float z) {
  if (ACT == DGDM_ACT_GELU) return 0.5f * z * (1.0f + erff(z * 0.70710678118654752f));
  if (ACT == DGDM_ACT_RELU) return fmaxf(z, 0.f);
  if (ACT == DGDM_ACT_SILU) return z / (1.0f + __expf(-z));
  return z;
}
template <int ACT>
__device__ __forceinline__ float act_df(float z) {
  if (ACT == DGDM_ACT_GELU)
    return 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
  if (ACT == DGDM_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (ACT == DGDM_ACT_SILU) {
    const float s = 1.0f / (1.0f + __expf(-z));
    return s * (1.0f + z * (1.0f - s));
  }
  return 1.f;
}

