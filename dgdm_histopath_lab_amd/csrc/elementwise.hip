// Fused activation + dropout for the convolution outputs of DynamicGraphLayer
// (core/graph_layers.py:233-239: h = dropout(GELU(conv(x)))) and the U-Net's ReLU sites.
//   fwd: y  = dropout(act(x))            bwd: dx = dy * mask * act'(x)
// `decide` (ReLU only, nullable, one byte per element): the side of the kink every element takes is read from it instead
// of from the sign of x -- the parity tests hand the reference's decisions to the kernels so that both sides differentiate the
// same piecewise-linear function (an element within rounding of zero may otherwise fall on either side).
// Pure HBM streaming: 16 B per lane, grid-stride; the dropout mask is recomputed from
// (seed, element index) in the backward instead of being stored.
#include "common.hpp"
#include "rowmath.hpp"

namespace {

template <int ACT, bool BWD>
__global__ __launch_bounds__(256) void k_act_dropout(const float* __restrict__ x, const float* __restrict__ dy, int64_t n4,
                                                     float drop_p, DgdmSeed seed_in, float* __restrict__ out,
                                                     const uint8_t* __restrict__ decide, unsigned* __restrict__ amax) {
  unsigned am = 0;
  const uint32_t seed = seed_in.value();
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // U float4 per thread and round, every load issued before the first use (round 5: one float4 per thread left the kernel latency-
  // bound at ~3.9 TB/s -- 10.6 us for 41 MB; the loads of a round now overlap)
  constexpr int U = 4;
  for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i0 < n4; i0 += U * stride) {
    float4 v[U], g[U];
    uchar4 d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      const bool ok = i < n4;
      v[u] = ok ? reinterpret_cast<const float4*>(x)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (BWD) g[u] = ok ? reinterpret_cast<const float4*>(dy)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (ACT == DGDM_ACT_RELU && decide) d[u] = ok ? reinterpret_cast<const uchar4*>(decide)[i] : make_uchar4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i >= n4) break;
      float4 o;
      if (BWD) o = make_float4(g[u].x * act_df<ACT>(v[u].x), g[u].y * act_df<ACT>(v[u].y), g[u].z * act_df<ACT>(v[u].z), g[u].w * act_df<ACT>(v[u].w));
      else o = make_float4(act_f<ACT>(v[u].x), act_f<ACT>(v[u].y), act_f<ACT>(v[u].z), act_f<ACT>(v[u].w));
      if (ACT == DGDM_ACT_RELU && decide) {
        const float4 s = BWD ? g[u] : v[u];
        o = make_float4(d[u].x ? s.x : 0.f, d[u].y ? s.y : 0.f, d[u].z ? s.z : 0.f, d[u].w ? s.w : 0.f);
      }
      if (drop_p > 0.f) {
        const float4 m = dropout_scale4(seed, (uint64_t)i * 4, thresh, keep_scale);
        o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
      }
      reinterpret_cast<float4*>(out)[i] = o;
      if (amax) am = dgdm_amax4(am, o);
    }
  }
  if (amax) dgdm_amax_commit(am, amax);     // wave-uniform: every thread of the block gets here
}

template <bool BWD>
int launch(const float* x, const float* dy, int64_t n, int32_t act, float drop_p, uint32_t seed, float* out, const uint8_t* decide,
           uint32_t* amax, hipStream_t s) {
  if (n < 0 || act < 0 || act > DGDM_ACT_ELU || !(drop_p >= 0.f && drop_p < 1.f)) return DGDM_ERR_INVALID_ARG;
  if (decide && act != DGDM_ACT_RELU) return DGDM_ERR_INVALID_ARG;
  if (n == 0) return DGDM_OK;
  if (!x || !out || (BWD && !dy)) return DGDM_ERR_INVALID_ARG;
  if ((n & 3) || !dgdm_aligned16(x) || !dgdm_aligned16(out) || (BWD && !dgdm_aligned16(dy))) return DGDM_ERR_UNSUPPORTED;
  const int64_t n4 = n >> 2;
  int64_t blocks = (n4 + 4 * 256 - 1) / (4 * 256);      // four float4 per thread and round (k_act_dropout: U)
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
#define GO(A) hipLaunchKernelGGL((k_act_dropout<A, BWD>), dim3((unsigned)blocks), dim3(256), 0, s, x, dy, n4, drop_p, dgdm_seed_arg(seed), out, decide, amax)
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

}  // namespace

extern "C" int dgdm_act_dropout_fwd(const float* x, int64_t n, int32_t act, float drop_p, uint32_t seed, float* y, const uint8_t* decide,
                                    uint32_t* amax, void* stream) {
  return launch<false>(x, nullptr, n, act, drop_p, seed, y, decide, amax, static_cast<hipStream_t>(stream));
}

extern "C" int dgdm_act_dropout_bwd(const float* x, const float* dy, int64_t n, int32_t act, float drop_p, uint32_t seed,
                                    float* dx, const uint8_t* decide, uint32_t* amax, void* stream) {
  return launch<true>(x, dy, n, act, drop_p, seed, dx, decide, amax, static_cast<hipStream_t>(stream));
}
