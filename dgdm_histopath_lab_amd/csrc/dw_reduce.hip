// Deferred reduction of the split-M weight-gradient GEMMs (gemm3.hip / gemm_h.hip): every dW GEMM of a backward pass leaves its
// chunk partials [slots][N*K (+N)] in its workspace and the reductions of MANY such GEMMs run in one launch at the end of the pass
// (ops.py queues them; ~50 k_gemm*_tn_final launches of ~7 us each per training step otherwise).  Same arithmetic as
// k_gemm3_tn_final: thread (column, part) adds slots part, part + PARTS, ... in order, the PARTS partial sums are added in index
// order -- bitwise identical to the immediate reduction.
#include "common.hpp"

namespace {

__host__ __device__ inline int parts_of(int slots) { return slots > 32 ? 16 : 4; }

struct Batch {
  DgdmTnReduce d[DGDM_TN_REDUCE_MAX];
  int first_block[DGDM_TN_REDUCE_MAX + 1];
  int count;
};

// Four columns per thread (one 16-byte load per slot; the widths are multiples of 4: N, K, K0 % 4 == 0), four loads in flight.  Per
// COLUMN the arithmetic is the scalar kernels': a0 takes slots part, part + 2 PARTS, ..., a1 takes part + PARTS, part + 3 PARTS, ...,
// then a0 + a1, then the PARTS partial sums in index order -- bit for bit the immediate reduction (k_gemm*_tn_final), at twice the
// rate (the scalar form moved 64-byte pieces, two loads in flight per thread: 2.6 TB/s over ~1 GB of partials per step).
__device__ __forceinline__ void add4(float4& a, const float4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }

__global__ __launch_bounds__(256) void k_tn_reduce_many(const Batch b) {
  int i = 0;
  while (i + 1 < b.count && (int)blockIdx.x >= b.first_block[i + 1]) ++i;      // wave-uniform scan of <= 24 entries
  const DgdmTnReduce& d = b.d[i];
  const int PARTS = parts_of(d.slots), QUADS = 256 / PARTS;
  const int c = threadIdx.x % QUADS, part = threadIdx.x / QUADS;
  const int64_t col = 4 * ((int64_t)(blockIdx.x - b.first_block[i]) * QUADS + c);
  const int64_t width = (int64_t)d.N * d.K + (d.db ? d.N : 0);
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  if (col < width) {
    const float* base = d.partial + col;
    int s = part;
    for (; s + 3 * PARTS < d.slots; s += 4 * PARTS) {      // four independent loads in flight, added in slot order
      const float4 v0 = *reinterpret_cast<const float4*>(base + (int64_t)s * width);
      const float4 v1 = *reinterpret_cast<const float4*>(base + (int64_t)(s + PARTS) * width);
      const float4 v2 = *reinterpret_cast<const float4*>(base + (int64_t)(s + 2 * PARTS) * width);
      const float4 v3 = *reinterpret_cast<const float4*>(base + (int64_t)(s + 3 * PARTS) * width);
      add4(a0, v0); add4(a1, v1); add4(a0, v2); add4(a1, v3);
    }
    for (; s + PARTS < d.slots; s += 2 * PARTS) {
      add4(a0, *reinterpret_cast<const float4*>(base + (int64_t)s * width));
      add4(a1, *reinterpret_cast<const float4*>(base + (int64_t)(s + PARTS) * width));
    }
    if (s < d.slots) add4(a0, *reinterpret_cast<const float4*>(base + (int64_t)s * width));
  }
  __shared__ float4 sm[256];
  sm[part * QUADS + c] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
  __syncthreads();
  if (part == 0 && col < width) {
    float4 t = sm[c];
    for (int p = 1; p < PARTS; ++p) add4(t, sm[p * QUADS + c]);
    const int64_t nk = (int64_t)d.N * d.K;
    if (col < nk) {      // columns [0, K0) of dW go to dW0, [K0, K) to dW1 (two parameters behind one contraction); K, K0 % 4 == 0
      const int64_t n = col / d.K;
      const int k = (int)(col % d.K);
      float* dst = k < d.K0 ? d.dW0 + n * d.ld0 + k : d.dW1 + n * d.ld1 + (k - d.K0);
      dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;      // destination rows need not be 16-byte aligned
    } else {
      float* dst = d.db + (col - nk);
      dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
    }
  }
}

}  // namespace

extern "C" int dgdm_gemm_tn_reduce_many(const DgdmTnReduce* descs, int32_t count, void* stream) {
  if (count < 0 || count > DGDM_TN_REDUCE_MAX || (count > 0 && !descs)) return DGDM_ERR_INVALID_ARG;
  if (count == 0) return DGDM_OK;
  Batch b;
  b.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const DgdmTnReduce& d = descs[i];
    if (!d.partial || d.slots <= 0 || d.N <= 0 || d.K <= 0 || d.K0 < 0 || d.K0 > d.K || (d.K0 > 0 && !d.dW0) || (d.K0 < d.K && !d.dW1))
      return DGDM_ERR_INVALID_ARG;
    // what the dW GEMMs require anyway; N only counts through the bias columns behind the N*K block
    if ((d.K & 3) || (d.K0 & 3) || (d.db && (d.N & 3)) || !dgdm_aligned16(d.partial)) return DGDM_ERR_UNSUPPORTED;
    b.d[i] = d;
    b.first_block[i] = blocks;
    const int64_t width = (int64_t)d.N * d.K + (d.db ? d.N : 0);
    const int cols = 4 * (256 / parts_of(d.slots));
    blocks += (int)((width + cols - 1) / cols);
  }
  b.first_block[count] = blocks;
  hipLaunchKernelGGL(k_tn_reduce_many, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), b);
  return dgdm_launch_status();
}
