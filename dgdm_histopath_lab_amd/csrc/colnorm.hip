// BatchNorm1d over the NODES of a batch, fused with the activation and dropout that follow it:
//   y = dropout(act( (x - mean_c) * rstd_c * gamma_c + beta_c ))        per channel c, statistics over the N rows
// The reference builds nn.BatchNorm1d(dim) when a FeatureEncoder / GraphEncoder is constructed with normalization="batch"
// (models/encoders.py:95-100,211-219) and applies it to the 2-D [N, dim] node matrix: training mode normalises with the biased
// variance of the batch of nodes and updates the running averages (momentum 0.1, unbiased variance), eval mode uses the running
// averages.  Not DGDMModel's default (LayerNorm), so this is a plain, deterministic implementation:
//   statistics  two passes over x (mean, then centred squares) as chunked column sums in FIXED order (256-row chunks, then the chunks in index order) -- no E[x^2] - E[x]^2;
//   apply       float4 streaming kernel (activation / dropout code of rowmath.hpp: the mask is the same function of
//               (seed, element index) as in k_act_dropout, so forward and backward regenerate it);
//   backward    g' = dy * mask * act'(z);  dbeta = sum g', dgamma = sum g' xhat (chunked column sums);
//               dx = gamma rstd (g' - dbeta / N - xhat dgamma / N)   (training)   |   gamma rstd g'   (eval).
#include "common.hpp"
#include "rowmath.hpp"

namespace {

constexpr int CN_CHUNK = 256;       // rows per partial-sum chunk

// partial[chunk][C] = sum over the chunk's rows of f(x): MODE 0: x;  1: (x - mean)^2
template <int MODE>
__global__ __launch_bounds__(256) void k_colnorm_partial(const float* __restrict__ x, int64_t ldx, int N, int C, const float* __restrict__ mean,
                                                         float* __restrict__ part) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), lane_r = threadIdx.x >> 6;
  const int r0 = blockIdx.y * CN_CHUNK, r1 = min(N, r0 + CN_CHUNK);
  float acc = 0.f;
  if (col < C) {
    const float mu = MODE == 1 ? mean[col] : 0.f;
    for (int r = r0 + lane_r; r < r1; r += 4) {
      const float v = x[(int64_t)r * ldx + col] - mu;
      acc += MODE == 1 ? v * v : v;
    }
  }
  __shared__ float sm[4][64];
  sm[lane_r][threadIdx.x & 63] = acc;
  __syncthreads();
  if (lane_r == 0 && col < C)
    part[(int64_t)blockIdx.y * C + col] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// column c: sums the `chunks` partial rows in order; MODE 0: mean = sum / N;  1: var = sum / N -> rstd, running averages
template <int MODE>
__global__ __launch_bounds__(256) void k_colnorm_finish(const float* __restrict__ part, int chunks, int C, int N, float eps, float momentum,
                                                        float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                                        float* __restrict__ run_var) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float t = 0.f;
  for (int k = 0; k < chunks; ++k) t += part[(int64_t)k * C + c];
  if (MODE == 0) {
    mean[c] = t / (float)N;
  } else {
    const float var = t / (float)N;
    rstd[c] = 1.0f / sqrtf(var + eps);
    if (run_mean) {
      run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean[c];
      run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (N > 1 ? t / (float)(N - 1) : var);
    }
  }
}

// eval mode: the "statistics" are the running averages
__global__ __launch_bounds__(256) void k_colnorm_from_running(const float* __restrict__ run_mean, const float* __restrict__ run_var, int C, float eps,
                                                              float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) { mean[c] = run_mean[c]; rstd[c] = 1.0f / sqrtf(run_var[c] + eps); }
}

template <int ACT>
__global__ __launch_bounds__(256) void k_colnorm_apply(const float* __restrict__ x, int64_t n4, int c4, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float drop_p, DgdmSeed seed_in, float* __restrict__ y,
                                                       unsigned* __restrict__ amax) {
  unsigned am = 0;
  const uint32_t seed = seed_in.value();
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int k = (int)(i % c4);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 mu = reinterpret_cast<const float4*>(mean)[k], rs = reinterpret_cast<const float4*>(rstd)[k];
    const float4 g = reinterpret_cast<const float4*>(gamma)[k], b = reinterpret_cast<const float4*>(beta)[k];
    float4 o = make_float4(act_f<ACT>((v.x - mu.x) * rs.x * g.x + b.x), act_f<ACT>((v.y - mu.y) * rs.y * g.y + b.y),
                           act_f<ACT>((v.z - mu.z) * rs.z * g.z + b.z), act_f<ACT>((v.w - mu.w) * rs.w * g.w + b.w));
    if (drop_p > 0.f) {
      const float4 m = dropout_scale4(seed, (uint64_t)i * 4, thresh, keep_scale);
      o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
    }
    reinterpret_cast<float4*>(y)[i] = o;
    if (amax) am = dgdm_amax4(am, o);
  }
  if (amax) dgdm_amax_commit(am, amax);
}

// backward stage 1: g' = dy * mask * act'(z) written to gp, partial[chunk][0:C) = sum g', [C:2C) = sum g' xhat
template <int ACT>
__global__ __launch_bounds__(256) void k_colnorm_bwd_partial(const float* __restrict__ x, const float* __restrict__ dy, int N, int C,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float drop_p,
                                                             DgdmSeed seed_in, float* __restrict__ gp, float* __restrict__ part) {
  const uint32_t seed = seed_in.value();
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), lane_r = threadIdx.x >> 6;
  const int r0 = blockIdx.y * CN_CHUNK, r1 = min(N, r0 + CN_CHUNK);
  float s1 = 0.f, s2 = 0.f;
  if (col < C) {
    const float mu = mean[col], rs = rstd[col], g = gamma[col], b = beta[col];
    for (int r = r0 + lane_r; r < r1; r += 4) {
      const int64_t e = (int64_t)r * C + col;
      const float xh = (x[e] - mu) * rs;
      float gv = dy[e] * act_df<ACT>(xh * g + b);
      if (drop_p > 0.f) {
        const float4 m = dropout_scale4(seed, (uint64_t)(e & ~(int64_t)3), thresh, keep_scale);
        const int w = (int)(e & 3);
        gv *= w == 0 ? m.x : w == 1 ? m.y : w == 2 ? m.z : m.w;
      }
      gp[e] = gv;
      s1 += gv; s2 += gv * xh;
    }
  }
  __shared__ float sa[4][64], sb[4][64];
  sa[lane_r][threadIdx.x & 63] = s1; sb[lane_r][threadIdx.x & 63] = s2;
  __syncthreads();
  if (lane_r == 0 && col < C) {
    part[(int64_t)blockIdx.y * 2 * C + col] = (sa[0][threadIdx.x] + sa[1][threadIdx.x]) + (sa[2][threadIdx.x] + sa[3][threadIdx.x]);
    part[(int64_t)blockIdx.y * 2 * C + C + col] = (sb[0][threadIdx.x] + sb[1][threadIdx.x]) + (sb[2][threadIdx.x] + sb[3][threadIdx.x]);
  }
}

__global__ __launch_bounds__(256) void k_colnorm_bwd_finish(const float* __restrict__ part, int chunks, int C, float* __restrict__ dbeta,
                                                            float* __restrict__ dgamma) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int k = 0; k < chunks; ++k) { a += part[(int64_t)k * 2 * C + c]; b += part[(int64_t)k * 2 * C + C + c]; }
  dbeta[c] = a; dgamma[c] = b;
}

// dx (in place over gp): training: gamma rstd (g' - dbeta / N - xhat dgamma / N);  eval: gamma rstd g'
__global__ __launch_bounds__(256) void k_colnorm_bwd_apply(const float* __restrict__ x, int64_t n4, int c4, int N, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ dbeta, const float* __restrict__ dgamma, int training,
                                                           float* __restrict__ gp, unsigned* __restrict__ amax) {
  unsigned am = 0;
  const float invN = training ? 1.0f / (float)N : 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int k = (int)(i % c4);
    const float4 v = reinterpret_cast<const float4*>(x)[i], gv = reinterpret_cast<const float4*>(gp)[i];
    const float4 mu = reinterpret_cast<const float4*>(mean)[k], rs = reinterpret_cast<const float4*>(rstd)[k];
    const float4 g = reinterpret_cast<const float4*>(gamma)[k];
    const float4 db = reinterpret_cast<const float4*>(dbeta)[k], dg = reinterpret_cast<const float4*>(dgamma)[k];
    float4 o;
    o.x = g.x * rs.x * (gv.x - db.x * invN - (v.x - mu.x) * rs.x * dg.x * invN);
    o.y = g.y * rs.y * (gv.y - db.y * invN - (v.y - mu.y) * rs.y * dg.y * invN);
    o.z = g.z * rs.z * (gv.z - db.z * invN - (v.z - mu.z) * rs.z * dg.z * invN);
    o.w = g.w * rs.w * (gv.w - db.w * invN - (v.w - mu.w) * rs.w * dg.w * invN);
    reinterpret_cast<float4*>(gp)[i] = o;
    if (amax) am = dgdm_amax4(am, o);
  }
  if (amax) dgdm_amax_commit(am, amax);
}

int cn_chunks(int N) { return N <= 0 ? 1 : (int)(((int64_t)N + CN_CHUNK - 1) / CN_CHUNK); }
int cn_blocks(int64_t n4) { int64_t b = (n4 + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

}  // namespace

extern "C" size_t dgdm_colnorm_workspace_bytes(int32_t N, int32_t C) {
  return (N <= 0 || C <= 0) ? 0 : (size_t)cn_chunks(N) * 2 * (size_t)C * sizeof(float);
}

extern "C" int dgdm_colnorm_fwd(const float* x, int32_t N, int32_t C, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, int32_t training, float momentum, float eps, int32_t act, float drop_p, uint32_t seed,
                                float* y, float* mean, float* rstd, void* workspace, size_t workspace_bytes, uint32_t* amax, void* stream_) {
  if (N < 0 || C <= 0 || act < 0 || act > DGDM_ACT_ELU || !(drop_p >= 0.f && drop_p < 1.f) || !(momentum >= 0.f && momentum <= 1.f))
    return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_OK;
  if (!x || !gamma || !beta || !y || !mean || !rstd || !workspace || (!training && (!running_mean || !running_var))) return DGDM_ERR_INVALID_ARG;
  if ((C & 3) || !dgdm_aligned16(x) || !dgdm_aligned16(y) || !dgdm_aligned16(gamma) || !dgdm_aligned16(beta) || !dgdm_aligned16(mean) ||
      !dgdm_aligned16(rstd))
    return DGDM_ERR_UNSUPPORTED;
  if (workspace_bytes < dgdm_colnorm_workspace_bytes(N, C)) return DGDM_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  float* part = static_cast<float*>(workspace);
  const int chunks = cn_chunks(N);
  const dim3 pg((C + 63) / 64, chunks), fg((C + 255) / 256);
  if (training) {
    hipLaunchKernelGGL(k_colnorm_partial<0>, pg, dim3(256), 0, s, x, (int64_t)C, N, C, nullptr, part);
    hipLaunchKernelGGL(k_colnorm_finish<0>, fg, dim3(256), 0, s, part, chunks, C, N, eps, momentum, mean, rstd, nullptr, nullptr);
    hipLaunchKernelGGL(k_colnorm_partial<1>, pg, dim3(256), 0, s, x, (int64_t)C, N, C, mean, part);
    hipLaunchKernelGGL(k_colnorm_finish<1>, fg, dim3(256), 0, s, part, chunks, C, N, eps, momentum, mean, rstd, running_mean, running_var);
  } else {
    hipLaunchKernelGGL(k_colnorm_from_running, fg, dim3(256), 0, s, running_mean, running_var, C, eps, mean, rstd);
  }
  const int64_t n4 = (int64_t)N * C / 4;
#define GO(A) hipLaunchKernelGGL((k_colnorm_apply<A>), dim3(cn_blocks(n4)), dim3(256), 0, s, x, n4, C / 4, mean, rstd, gamma, beta, drop_p, \
                                 dgdm_seed_arg(seed), y, amax)
  switch (act) {
    case DGDM_ACT_GELU: GO(DGDM_ACT_GELU); break;
    case DGDM_ACT_RELU: GO(DGDM_ACT_RELU); break;
    case DGDM_ACT_SILU: GO(DGDM_ACT_SILU); break;
    case DGDM_ACT_ELU: GO(DGDM_ACT_ELU); break;
    default: GO(DGDM_ACT_NONE); break;
  }
#undef GO
  return dgdm_launch_status();
}

extern "C" int dgdm_colnorm_bwd(const float* x, const float* dy, int32_t N, int32_t C, const float* gamma, const float* beta, const float* mean,
                                const float* rstd, int32_t training, int32_t act, float drop_p, uint32_t seed, float* dx, float* dgamma,
                                float* dbeta, void* workspace, size_t workspace_bytes, uint32_t* amax, void* stream_) {
  if (N < 0 || C <= 0 || act < 0 || act > DGDM_ACT_ELU || !(drop_p >= 0.f && drop_p < 1.f)) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_OK;
  if (!x || !dy || !gamma || !beta || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace) return DGDM_ERR_INVALID_ARG;
  if ((C & 3) || !dgdm_aligned16(x) || !dgdm_aligned16(dx) || !dgdm_aligned16(gamma) || !dgdm_aligned16(mean) || !dgdm_aligned16(rstd) ||
      !dgdm_aligned16(dgamma) || !dgdm_aligned16(dbeta))
    return DGDM_ERR_UNSUPPORTED;
  if (workspace_bytes < dgdm_colnorm_workspace_bytes(N, C)) return DGDM_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  float* part = static_cast<float*>(workspace);
  const int chunks = cn_chunks(N);
  const dim3 pg((C + 63) / 64, chunks);
#define GO(A) hipLaunchKernelGGL((k_colnorm_bwd_partial<A>), pg, dim3(256), 0, s, x, dy, N, C, mean, rstd, gamma, beta, drop_p, dgdm_seed_arg(seed), dx, part)
  switch (act) {
    case DGDM_ACT_GELU: GO(DGDM_ACT_GELU); break;
    case DGDM_ACT_RELU: GO(DGDM_ACT_RELU); break;
    case DGDM_ACT_SILU: GO(DGDM_ACT_SILU); break;
    case DGDM_ACT_ELU: GO(DGDM_ACT_ELU); break;
    default: GO(DGDM_ACT_NONE); break;
  }
#undef GO
  hipLaunchKernelGGL(k_colnorm_bwd_finish, dim3((C + 255) / 256), dim3(256), 0, s, part, chunks, C, dbeta, dgamma);
  const int64_t n4 = (int64_t)N * C / 4;
  hipLaunchKernelGGL(k_colnorm_bwd_apply, dim3(cn_blocks(n4)), dim3(256), 0, s, x, n4, C / 4, N, mean, rstd, gamma, dbeta, dgamma, training, dx, amax);
  return dgdm_launch_status();
}
