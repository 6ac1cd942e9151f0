// Shared host/device helpers for libdgdm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dgdm_hip.h"

#define DGDM_WAVE 64

#define DGDM_REQUIRE(cond)                      \
  do {                                          \
    if (!(cond)) return DGDM_ERR_INVALID_ARG;   \
  } while (0)

static inline int dgdm_launch_status() {
  return hipGetLastError() == hipSuccess ? DGDM_OK : DGDM_ERR_LAUNCH;
}

// Device-side fill used instead of hipMemsetAsync: memset nodes recorded by stream capture did not
// re-execute on graph replay on this runtime (counters kept their previous values and the dependent
// scatter ran out of bounds), a kernel node always does.  `bytes` must be a multiple of 4.
static __global__ void k_dgdm_fill32(uint32_t* __restrict__ p, size_t n, uint32_t v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
static inline void dgdm_fill_async(void* p, uint8_t byte, size_t bytes, hipStream_t s) {
  const size_t n = bytes / 4;
  if (n == 0) return;
  const size_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_dgdm_fill32, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s,
                     static_cast<uint32_t*>(p), n, 0x01010101u * byte);
}

// rows x cols floats of a matrix with leading dimension ld (floats) <- 0
static __global__ void k_dgdm_zero2d(float* __restrict__ p, int64_t ld, int cols, int64_t rows) {
  const int64_t n = rows * cols, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) p[(i / cols) * ld + i % cols] = 0.f;
}
static inline void dgdm_fill2d_async(float* p, int64_t ld, int cols, int64_t rows, hipStream_t s) {
  const int64_t n = rows * cols;
  if (n <= 0) return;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_dgdm_zero2d, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, p, ld, cols, rows);
}

static inline size_t dgdm_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline bool dgdm_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float4 f4_fma(float s, const float4 v, float4 a) {
  a.x = fmaf(s, v.x, a.x); a.y = fmaf(s, v.y, a.y); a.z = fmaf(s, v.z, a.z); a.w = fmaf(s, v.w, a.w);
  return a;
}

// wave64 reductions over the full wave (xor butterflies; result in every lane)
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

// ---- operand maxima for the fp16 hi+lo GEMMs (include/dgdm_hip.h: "amax slot", csrc/gemm_h.hip).  A kernel that produces a
// GEMM operand keeps max|x| as the float bits of its outputs (non-negative floats order like unsigned integers): per-thread
// running maximum, then ONE atomic max per workgroup into way (blockIdx.x % WAYS) of the caller's slot group.  Integer max is
// order-free: the result is bitwise reproducible.  Every thread of the workgroup must reach dgdm_amax_commit.
__device__ __forceinline__ unsigned dgdm_absbits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned dgdm_amax4(unsigned m, const float4 v) {
  return max(max(m, dgdm_absbits(v.x)), max(max(dgdm_absbits(v.y), dgdm_absbits(v.z)), dgdm_absbits(v.w)));
}
__device__ __forceinline__ void dgdm_amax_commit(unsigned m, unsigned* __restrict__ group) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  __shared__ unsigned dgdm_amax_sm[16];
  if ((threadIdx.x & 63) == 0) dgdm_amax_sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) t = max(t, dgdm_amax_sm[w]);
    if (t) atomicMax(group + (blockIdx.x % DGDM_AMAX_WAYS) * DGDM_AMAX_STRIDE, t);
  }
}

// Dropout seeds.  Every dropout site passes its seed by value (so a forward kernel and the backward
// kernels that recompute its mask agree by construction).  A kernel launch recorded in a HIP graph
// replays with the recorded value; to give each replay fresh masks the kernels fold in a per-device
// step counter ("seed epoch") that lives in device memory and is advanced by a kernel of its own
// (dgdm_seed_epoch_advance, once per training step, before the forward).  Epoch 0 (the default, and the
// only value outside graph replay) leaves the seed unchanged.
// Workgroups a split-M dW GEMM aims for (row chunks x 128 x 128 tiles): ~2 per CU, all resident in one round, when the problem
// runs alone (gemm3.hip, gemm_h.hip: the same rule, dgdm_gemm_tn_chunks reports it).
#ifndef DGDM_TN_WANT
#define DGDM_TN_WANT 496
#endif
// ... and inside a many-problem launch (dgdm_gemm_tn_partial_many_f16x2), where two dozen problems fill the chip together: fewer,
// longer chunks -- a quarter of the partial sums to write and to reduce (same-box sweep 496 / 248 / 124 / 62 / 31:
// 278.9 / 281.5 / 283.2 = 300.2 / 300.8 / 277.0 slides/s).
#ifndef DGDM_TN_WANT_GROUPED
#define DGDM_TN_WANT_GROUPED 96
#endif

struct DgdmSeed {
  uint32_t base;
  const uint32_t* epoch;
  __device__ __forceinline__ uint32_t value() const { return base ^ (epoch[0] * 0x9E3779B9U); }
};
const uint32_t* dgdm_seed_epoch_ptr();   // api.hip: address of the current device's counter
static inline DgdmSeed dgdm_seed_arg(uint32_t seed) { return DgdmSeed{seed, dgdm_seed_epoch_ptr()}; }
