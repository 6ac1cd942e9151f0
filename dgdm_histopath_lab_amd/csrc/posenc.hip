// K5: SpatialAttention.get_positional_encoding (core/attention.py:225-259) fused with the
// `x + pos_enc` add (attention.py:306) for a whole batch.
//   per graph: lo = min(pos), hi = max(pos) over BOTH coordinates (one global min/max, :238-240)
//   pn = (pos - lo) / (hi - lo + 1e-8)
//   pe[n, 4k+0] = sin(pn_x * f_k), [4k+1] = cos(pn_x * f_k), [4k+2] = sin(pn_y * f_k), [4k+3] = cos(pn_y * f_k)
//   f_k = exp(-(2k) * ln(1e4) / (C/2)),  k = 0 .. C/4-1
// Elementwise / HBM-bound; the gradient wrt x is the identity, so no backward kernel exists.
#include "common.hpp"

namespace {

// one workgroup of 1024 threads per graph, 16 B per lane and iteration (a 256-thread block with scalar loads walked a 10k-node
// graph in 78 dependent steps: 34 us, more than the encoding kernel itself); min / max do not depend on the order
__global__ __launch_bounds__(1024) void k_pos_minmax(const float* __restrict__ pos, const int32_t* __restrict__ ptr,
                                                     float* __restrict__ minmax) {
  const int g = blockIdx.x;
  const int64_t a = 2 * (int64_t)ptr[g], b = 2 * (int64_t)ptr[g + 1];
  float lo = INFINITY, hi = -INFINITY;
  const int64_t off = (int64_t)((reinterpret_cast<uintptr_t>(pos) >> 2) & 3);       // float index of pos[0] inside its 16-byte line
  const int64_t a4 = a + ((4 - ((off + a) & 3)) & 3), b4 = b - ((off + b) & 3);     // [a4, b4): the 16-byte aligned middle
  if (a4 < b4) {
    for (int64_t i = a4 + 4 * (int64_t)threadIdx.x; i < b4; i += 4 * (int64_t)blockDim.x) {
      const float4 v = *reinterpret_cast<const float4*>(pos + i);
      lo = fminf(fminf(lo, v.x), fminf(fminf(v.y, v.z), v.w));
      hi = fmaxf(fmaxf(hi, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
    }
    for (int64_t i = a + threadIdx.x; i < a4; i += blockDim.x) { lo = fminf(lo, pos[i]); hi = fmaxf(hi, pos[i]); }
    for (int64_t i = b4 + threadIdx.x; i < b; i += blockDim.x) { lo = fminf(lo, pos[i]); hi = fmaxf(hi, pos[i]); }
  } else {
    for (int64_t i = a + threadIdx.x; i < b; i += blockDim.x) { lo = fminf(lo, pos[i]); hi = fmaxf(hi, pos[i]); }
  }
  __shared__ float slo[16], shi[16];
  lo = -wave_max(-lo);
  hi = wave_max(hi);
  if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) { lo = fminf(lo, slo[w]); hi = fmaxf(hi, shi[w]); }
    minmax[2 * g] = lo;
    minmax[2 * g + 1] = hi;
  }
}

// one thread per (node, frequency k): writes 4 consecutive channels (16 B)
__global__ __launch_bounds__(256) void k_add_posenc(const float* __restrict__ x, int64_t ldx, const float* __restrict__ pos,
                                                    const int32_t* __restrict__ ptr, int B, const float* __restrict__ minmax,
                                                    int N, int C, float* __restrict__ out, int64_t ldo, unsigned* __restrict__ amax) {
  const int nk = C >> 2;
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  // ONE call of dgdm_amax_commit, reached by every thread without divergence: it contains a workgroup barrier, and a wave whose
  // lanes sit on both sides of `idx < N * nk` (N * C / 4 not a multiple of 64: any odd N at C = 128) would otherwise run it once per
  // side -- the barrier count goes wrong and thread 0 reads maxima the other waves have not stored yet (whatever the LDS held).
  unsigned am = 0u;
  if (idx < (int64_t)N * nk) {
    const int n = (int)(idx / nk), k = (int)(idx % nk);
    int g = 0;
    while (g + 1 < B && ptr[g + 1] <= n) ++g;  // B is small
    const float lo = minmax[2 * g], hi = minmax[2 * g + 1];
    const float inv = 1.0f / (hi - lo + 1e-8f);
    const float px = (pos[2 * (int64_t)n] - lo) * inv, py = (pos[2 * (int64_t)n + 1] - lo) * inv;
    const float f = expf((float)(2 * k) * (-9.210340371976184f / (float)(C / 2)));  // ln(1e4)
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x) v = *reinterpret_cast<const float4*>(x + (int64_t)n * ldx + 4 * k);
    v.x += sinf(px * f); v.y += cosf(px * f); v.z += sinf(py * f); v.w += cosf(py * f);
    *reinterpret_cast<float4*>(out + (int64_t)n * ldo + 4 * k) = v;
    am = dgdm_amax4(0u, v);
  }
  if (amax) dgdm_amax_commit(am, amax);
}

}  // namespace

extern "C" int dgdm_add_posenc(const float* x, int64_t ldx, const float* pos, const int32_t* ptr, int32_t B, int32_t N,
                               int32_t C, float* minmax_ws, float* out, int64_t ldo, uint32_t* amax, void* stream_) {
  DGDM_REQUIRE(B >= 0 && N >= 0 && C > 0);
  if (N == 0 || B == 0) return DGDM_OK;
  DGDM_REQUIRE(pos && ptr && minmax_ws && out);
  if ((C & 3) || (ldo & 3) || (x && (ldx & 3)) || !dgdm_aligned16(out) || (x && !dgdm_aligned16(x))) return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(k_pos_minmax, dim3(B), dim3(1024), 0, s, pos, ptr, minmax_ws);
  const int64_t total = (int64_t)N * (C >> 2);
  hipLaunchKernelGGL(k_add_posenc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, ldx, pos, ptr, B, minmax_ws, N, C, out,
                     ldo, amax);
  return dgdm_launch_status();
}
