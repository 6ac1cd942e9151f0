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

__global__ __launch_bounds__(256) void k_tn_reduce_many(const Batch b) {
  int i = 0;
  while (i + 1 < b.count && (int)blockIdx.x >= b.first_block[i + 1]) ++i;      // wave-uniform scan of <= 24 entries
  const DgdmTnReduce& d = b.d[i];
  const int PARTS = parts_of(d.slots), COLS = 256 / PARTS;
  const int c = threadIdx.x % COLS, part = threadIdx.x / COLS;
  const int64_t col = (int64_t)(blockIdx.x - b.first_block[i]) * COLS + c;
  const int64_t width = (int64_t)d.N * d.K + (d.db ? d.N : 0);
  float a0 = 0.f, a1 = 0.f;
  if (col < width) {
    int s = part;
    for (; s + PARTS < d.slots; s += 2 * PARTS) {       // two independent loads in flight
      a0 += d.partial[(int64_t)s * width + col];
      a1 += d.partial[(int64_t)(s + PARTS) * width + col];
    }
    if (s < d.slots) a0 += d.partial[(int64_t)s * width + col];
  }
  __shared__ float sm[256];
  sm[part * COLS + c] = a0 + a1;
  __syncthreads();
  if (part == 0 && col < width) {
    float t = sm[c];
    for (int p = 1; p < PARTS; ++p) t += sm[p * COLS + c];
    const int64_t nk = (int64_t)d.N * d.K;
    if (col < nk) {      // columns [0, K0) of dW go to dW0, [K0, K) to dW1 (two parameters behind one contraction)
      const int64_t n = col / d.K;
      const int k = (int)(col % d.K);
      if (k < d.K0) d.dW0[n * d.ld0 + k] = t; else d.dW1[n * d.ld1 + (k - d.K0)] = t;
    } else {
      d.db[col - nk] = t;
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
    b.d[i] = d;
    b.first_block[i] = blocks;
    const int64_t width = (int64_t)d.N * d.K + (d.db ? d.N : 0);
    const int cols = 256 / parts_of(d.slots);
    blocks += (int)((width + cols - 1) / cols);
  }
  b.first_block[count] = blocks;
  hipLaunchKernelGGL(k_tn_reduce_many, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), b);
  return dgdm_launch_status();
}
