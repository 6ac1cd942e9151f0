// Element-wise pieces of the diffusion objective and of entity masking, batched over the graphs of a batch -- the reference
// runs them as chains of framework ops inside Python loops over graphs (models/dgdm_model.py:405-445, 482-506;
// core/diffusion.py:123-145):
//   dgdm_qsample        x_t[n] = sqrt(ac[t_g]) x0[n] + sqrt(1 - ac[t_g]) eps[n]        (add_noise; t_g = timestep of n's graph)
//                       with eps = NULL: out[n] = sqrt(ac[t_g]) g[n]                    (its backward w.r.t. x0)
//   dgdm_segment_mse_*  loss = mean over graphs of mse(pred_g, target_g) = sum_n w_g(n) sum_c (pred - target)^2,
//                       w_g = 1 / (B n_g C)  (dgdm_model.py:430-433); forward in two fixed-order stages, backward
//                       dpred = 2 w_g gloss (pred - target)
//   dgdm_mask_rows      out[n] = node_map[n] >= 0 ? token : x[n]                        (entity masking with the top-k kernel's map)
#include "common.hpp"

namespace {

__device__ __forceinline__ int graph_of(const int32_t* __restrict__ ptr, int B, int n) {
  int g = 0;
  while (g + 1 < B && ptr[g + 1] <= n) ++g;
  return g;
}

__global__ __launch_bounds__(256) void k_qsample(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ tab_a,
                                                 const float* __restrict__ tab_b, const int64_t* __restrict__ t, const int32_t* __restrict__ ptr,
                                                 int B, int64_t n4total, int c4, float* __restrict__ out, unsigned* __restrict__ amax) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned am = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4total; i += stride) {
    const int g = graph_of(ptr, B, (int)(i / c4));
    const int64_t tg = t[g];
    const float a = tab_a[tg];
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x *= a; v.y *= a; v.z *= a; v.w *= a;
    if (eps) {
      const float b = tab_b[tg];
      const float4 e = reinterpret_cast<const float4*>(eps)[i];
      v.x = fmaf(b, e.x, v.x); v.y = fmaf(b, e.y, v.y); v.z = fmaf(b, e.z, v.z); v.w = fmaf(b, e.w, v.w);
    }
    reinterpret_cast<float4*>(out)[i] = v;
    if (amax) am = dgdm_amax4(am, v);
  }
  if (amax) dgdm_amax_commit(am, amax);
}

constexpr int MSE_CHUNKS = 64;

// stage 1: block (chunk, g): sum of squared differences over the chunk's rows -> part[g][chunk]; fixed order inside the block
__global__ __launch_bounds__(256) void k_mse_stage1(const float* __restrict__ pred, const float* __restrict__ target,
                                                    const int32_t* __restrict__ ptr, int c4, float* __restrict__ part) {
  const int g = blockIdx.y, chunk = blockIdx.x;
  const int a = ptr[g], b = ptr[g + 1];
  const int per = (b - a + MSE_CHUNKS - 1) / MSE_CHUNKS;
  const int r0 = a + chunk * per, r1 = min(b, r0 + per);
  const int64_t i0 = (int64_t)r0 * c4, i1 = (int64_t)max(r1, r0) * c4;
  float acc = 0.f;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const float4 p = reinterpret_cast<const float4*>(pred)[i], q = reinterpret_cast<const float4*>(target)[i];
    const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z, dw = p.w - q.w;
    acc += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  acc = wave_sum(acc);
  __shared__ float sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[g * MSE_CHUNKS + chunk] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// stage 2: one wave: loss = sum_g w_g sum_chunk part[g][chunk], chunks and graphs in index order
__global__ __launch_bounds__(64) void k_mse_stage2(const float* __restrict__ part, const int32_t* __restrict__ ptr, int B, int C,
                                                   float* __restrict__ loss) {
  float total = 0.f;
  for (int g = 0; g < B; ++g) {
    float v = part[g * MSE_CHUNKS + threadIdx.x];
    v = wave_sum(v);
    const int n = ptr[g + 1] - ptr[g];
    if (n > 0) total += v / ((float)B * (float)n * (float)C);
  }
  if (threadIdx.x == 0) loss[0] = total;
}

__global__ __launch_bounds__(256) void k_mse_bwd(const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ gloss,
                                                 const int32_t* __restrict__ ptr, int B, int C, int64_t n4total, int c4,
                                                 float* __restrict__ dpred, unsigned* __restrict__ amax) {
  const float gl = gloss[0];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned am = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4total; i += stride) {
    const int g = graph_of(ptr, B, (int)(i / c4));
    const int n = ptr[g + 1] - ptr[g];
    const float w = 2.f * gl / ((float)B * (float)n * (float)C);
    const float4 p = reinterpret_cast<const float4*>(pred)[i], q = reinterpret_cast<const float4*>(target)[i];
    const float4 d = make_float4(w * (p.x - q.x), w * (p.y - q.y), w * (p.z - q.z), w * (p.w - q.w));
    reinterpret_cast<float4*>(dpred)[i] = d;
    am = dgdm_amax4(am, d);
  }
  if (amax) dgdm_amax_commit(am, amax);     // the gradient is the operand of the denoiser's last dX / dW GEMMs
}

__global__ __launch_bounds__(256) void k_mask_rows(const float* __restrict__ x, const int32_t* __restrict__ node_map, const float* __restrict__ token,
                                                   int64_t n4total, int f4, float* __restrict__ out, unsigned* __restrict__ amax) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned am = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4total; i += stride) {
    const int64_t n = i / f4;
    const int k = (int)(i % f4);
    const float4 v = node_map[n] >= 0 ? reinterpret_cast<const float4*>(token)[k] : reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<float4*>(out)[i] = v;
    if (amax) am = dgdm_amax4(am, v);
  }
  if (amax) dgdm_amax_commit(am, amax);
}

inline unsigned stream_blocks(int64_t n4) {
  int64_t b = (n4 + 255) / 256;
  return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int dgdm_qsample(const float* x, const float* eps, const float* tab_a, const float* tab_b, const int64_t* timesteps,
                            const int32_t* ptr, int32_t B, int32_t N, int32_t C, float* out, uint32_t* amax, void* stream) {
  DGDM_REQUIRE(B >= 0 && N >= 0 && C > 0);
  if (N == 0 || B == 0) return DGDM_OK;
  DGDM_REQUIRE(x && tab_a && timesteps && ptr && out && (!eps || tab_b));
  if ((C & 3) || !dgdm_aligned16(x) || !dgdm_aligned16(out) || (eps && !dgdm_aligned16(eps))) return DGDM_ERR_UNSUPPORTED;
  const int64_t n4 = (int64_t)N * (C >> 2);
  hipLaunchKernelGGL(k_qsample, dim3(stream_blocks(n4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, eps, tab_a, tab_b, timesteps, ptr,
                     B, n4, C >> 2, out, amax);
  return dgdm_launch_status();
}

extern "C" size_t dgdm_segment_mse_workspace_bytes(int32_t B) { return B <= 0 ? 0 : (size_t)B * MSE_CHUNKS * sizeof(float); }

extern "C" int dgdm_segment_mse_fwd(const float* pred, const float* target, const int32_t* ptr, int32_t B, int32_t N, int32_t C, float* loss,
                                    void* workspace, size_t workspace_bytes, void* stream_) {
  DGDM_REQUIRE(B > 0 && N >= 0 && C > 0 && loss && ptr && workspace);
  DGDM_REQUIRE(N == 0 || (pred && target));
  if ((C & 3) || (N > 0 && (!dgdm_aligned16(pred) || !dgdm_aligned16(target)))) return DGDM_ERR_UNSUPPORTED;
  if (workspace_bytes < dgdm_segment_mse_workspace_bytes(B)) return DGDM_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(k_mse_stage1, dim3(MSE_CHUNKS, B), dim3(256), 0, s, pred, target, ptr, C >> 2, part);
  hipLaunchKernelGGL(k_mse_stage2, dim3(1), dim3(64), 0, s, part, ptr, B, C, loss);
  return dgdm_launch_status();
}

extern "C" int dgdm_segment_mse_bwd(const float* pred, const float* target, const float* gloss, const int32_t* ptr, int32_t B, int32_t N,
                                    int32_t C, float* dpred, uint32_t* amax, void* stream) {
  DGDM_REQUIRE(B > 0 && N >= 0 && C > 0);
  if (N == 0) return DGDM_OK;
  DGDM_REQUIRE(pred && target && gloss && ptr && dpred);
  if ((C & 3) || !dgdm_aligned16(pred) || !dgdm_aligned16(target) || !dgdm_aligned16(dpred)) return DGDM_ERR_UNSUPPORTED;
  const int64_t n4 = (int64_t)N * (C >> 2);
  hipLaunchKernelGGL(k_mse_bwd, dim3(stream_blocks(n4)), dim3(256), 0, static_cast<hipStream_t>(stream), pred, target, gloss, ptr, B, C, n4,
                     C >> 2, dpred, amax);
  return dgdm_launch_status();
}

extern "C" int dgdm_mask_rows(const float* x, const int32_t* node_map, const float* token, int32_t N, int32_t F, float* out, uint32_t* amax,
                              void* stream) {
  DGDM_REQUIRE(N >= 0 && F > 0);
  if (N == 0) return DGDM_OK;
  DGDM_REQUIRE(x && node_map && token && out);
  if ((F & 3) || !dgdm_aligned16(x) || !dgdm_aligned16(token) || !dgdm_aligned16(out)) return DGDM_ERR_UNSUPPORTED;
  const int64_t n4 = (int64_t)N * (F >> 2);
  hipLaunchKernelGGL(k_mask_rows, dim3(stream_blocks(n4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, node_map, token, n4, F >> 2, out, amax);
  return dgdm_launch_status();
}
