// Head-mean attention weights  W_g[q][k] = mean_h softmax_k(S_h)[q][k]  per graph -- the tensor
// MultiHeadAttention returns with need_weights/average_attn_weights (core/attention.py:171-173)
// and DGDMModel exposes under return_attention (dgdm_model.py:360-361,400-401).  O(N_g^2) output
// by definition, so it is only produced on request, from Q, K, pos and the forward's lse2
// (P = exp2(S' - lse2), no second softmax pass).  One launch per head group; groups after the
// first accumulate into W (launches on one stream serialise, so no atomics are needed).
#include "attn_common.hpp"

namespace {

constexpr int QB = 64;
constexpr int KB = 64;

template <int HG>
__global__ __launch_bounds__(256) void k_attn_mean_weights(const float* __restrict__ Q, const float* __restrict__ K, int64_t ld,
                                                           const float* __restrict__ pos, const int32_t* __restrict__ ptr, int B,
                                                           const float* __restrict__ L2, float qscale, float bscale, int head0,
                                                           float inv_heads, int accumulate, float* __restrict__ W,
                                                           const int64_t* __restrict__ w_off, int N_tot) {
  using T = AttnTile<KB>;
  constexpr int NT = KB / 16;
  constexpr int F4 = KB * HG * 4 / 256;
  __shared__ __attribute__((aligned(16))) float smem[HG * T::HS + 2 * KB];
  float* Ks = smem;
  float* Ps = smem + HG * T::HS;

  int n0 = 0, n1 = 0, ltile = 0, g = 0;
  {
    int tile = blockIdx.x;
    bool found = false;
    for (g = 0; g < B; ++g) {
      n0 = ptr[g]; n1 = ptr[g + 1];
      const int nt = (n1 - n0 + QB - 1) / QB;
      if (tile < nt) { ltile = tile; found = true; break; }
      tile -= nt;
    }
    if (!found) return;
  }
  const int ng = n1 - n0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, G = lane >> 4;
  const int q_local = ltile * QB + wave * 16 + j;
  const bool q_ok = q_local < ng;
  const int q_row = n0 + (q_ok ? q_local : ng - 1);
  f32x4 qf[HG];
  float l2[HG];
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    const float4 t = *reinterpret_cast<const float4*>(Q + (int64_t)q_row * ld + (head0 + h) * 16 + 4 * G);
    qf[h] = f32x4{t.x * qscale, t.y * qscale, t.z * qscale, t.w * qscale};
    l2[h] = L2[(int64_t)(head0 + h) * N_tot + q_row];
  }
  const float2 pq = *reinterpret_cast<const float2*>(pos + 2 * (int64_t)q_row);
  float* Wg = W + w_off[g];

  for (int kb0 = 0; kb0 < ng; kb0 += KB) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < F4; ++i) {
      const int idx = tid + 256 * i;
      const int key = idx / (HG * 4), c = idx % (HG * 4);
      const int kl = kb0 + key;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kl < ng) v = *reinterpret_cast<const float4*>(K + (int64_t)(n0 + kl) * ld + head0 * 16 + c * 4);
      *reinterpret_cast<float4*>(&Ks[T::rm(c >> 2, key, (c & 3) * 4)]) = v;
    }
    if (tid < KB) {
      const int kl = kb0 + tid;
      float2 p = make_float2(0.f, 0.f);
      if (kl < ng) p = *reinterpret_cast<const float2*>(pos + 2 * (int64_t)(n0 + kl));
      *reinterpret_cast<float2*>(&Ps[2 * tid]) = p;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 pa = *reinterpret_cast<const float4*>(&Ps[2 * (16 * t + 4 * G)]);
      const float4 pb = *reinterpret_cast<const float4*>(&Ps[2 * (16 * t + 4 * G) + 4]);
      const float kx[4] = {pa.x, pa.z, pb.x, pb.z}, ky[4] = {pa.y, pa.w, pb.y, pb.w};
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < HG; ++h) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[T::rm(h, 16 * t + j, 4 * G)]);
        const f32x4 s = mfma16_k16(kf, qf[h], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dx = pq.x - kx[r], dy = pq.y - ky[r];
          const float d = __builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy)) * bscale;
          acc[r] += __builtin_amdgcn_exp2f(s[r] - d - l2[h]);
        }
      }
      const int k_local = kb0 + 16 * t + 4 * G;
      if (q_ok) {
        float* dst = Wg + (int64_t)q_local * ng + k_local;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (k_local + r < ng) dst[r] = (accumulate ? dst[r] : 0.f) + acc[r] * inv_heads;
      }
    }
  }
}

}  // namespace

extern "C" int dgdm_spatial_attn_mean_weights(const float* Q, const float* K, int64_t ld, const float* pos, const int32_t* ptr,
                                              int32_t B, int32_t num_q_tiles, int32_t N_tot, int32_t H, float scale,
                                              float inv_tau, const float* lse2, float* W, const int64_t* w_offsets,
                                              void* stream_) {
  DGDM_REQUIRE(B >= 0 && N_tot >= 0 && H > 0 && num_q_tiles >= 0);
  if (N_tot == 0 || num_q_tiles == 0) return DGDM_OK;
  DGDM_REQUIRE(Q && K && pos && ptr && lse2 && W && w_offsets);
  if ((ld & 3) || ld < H * 16 || !dgdm_aligned16(Q) || !dgdm_aligned16(K)) return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const float qscale = scale * DGDM_LOG2E, bscale = inv_tau * DGDM_LOG2E, inv_heads = 1.0f / (float)H;
#define GO(HG)                                                                                                     \
  for (int h0 = 0; h0 < H; h0 += HG)                                                                               \
  hipLaunchKernelGGL((k_attn_mean_weights<HG>), dim3(num_q_tiles), dim3(256), 0, s, Q, K, ld, pos, ptr, B, lse2, qscale, \
                     bscale, h0, inv_heads, h0 > 0 ? 1 : 0, W, w_offsets, N_tot)
  if (H % 4 == 0) { GO(4); }
  else if (H % 2 == 0) { GO(2); }
  else { GO(1); }
#undef GO
  return dgdm_launch_status();
}
