// K4 forward: fused variable-length spatial attention
//   O[q] = softmax_k( Q[q].K[k]/sqrt(d) - |pos_q - pos_k|/tau ) V[k]     per graph, per head
// Replaces SpatialAttention.compute_spatial_bias + MultiHeadAttention.forward's
// matmul / softmax / matmul (core/attention.py:261-283, 135-157) without ever materialising the
// [H, N, N] score tensor (3.2 GB per 10k-node slide in the reference).
//
// Structure: one workgroup = 4 waves = 64 queries of one graph x HG heads; each wave owns 16
// queries.  K/V blocks of KB keys are staged through LDS (K row-major per head, V transposed per
// head).  S^T = K Q^T is computed with the key on the MFMA row and the query on the lane, so the
// accumulator of S^T is directly the B operand of O^T += V^T P^T: P never leaves registers.  The
// distance bias is computed once per (key, query) pair and shared by all heads of the group.
// Online softmax in the log2 domain (Q is pre-scaled by log2(e)/sqrt(d), the bias by log2(e)/tau).
// fp32 MFMA (v_mfma_f32_16x16x4_f32): every product runs at the fp32 matrix peak (64 FLOP/clk/SIMD).
#include "attn_common.hpp"

namespace {

constexpr int QB = 64;   // queries per workgroup (4 waves x 16)
constexpr float NEG_BIG = -1.0e30f;

template <int HG, int KB, bool DROP>
__global__ __launch_bounds__(256, 2) void k_attn_fwd(const float* __restrict__ Q, const float* __restrict__ K,
                                                     const float* __restrict__ V, int64_t ld,
                                                     const float* __restrict__ pos, const int32_t* __restrict__ ptr, int B,
                                                     float qscale, float bscale, float* __restrict__ O, int64_t ldo,
                                                     float* __restrict__ L2, int N_tot, float drop_p, DgdmSeed seed_in) {
  const uint32_t seed = seed_in.value();
  using T = AttnTile<KB>;
  constexpr int NT = KB / 16;                     // 16-key tiles per block
  const DropCfg dc(drop_p);
  constexpr int F4_PER_THREAD = KB * HG * 4 / 256;  // float4 staged per thread per tensor
  static_assert(KB % 16 == 0 && (KB * HG * 4) % 256 == 0, "staging must divide evenly");
  __shared__ __attribute__((aligned(16))) float smem[2 * HG * T::HS + 2 * KB];
  float* Ks = smem;
  float* Vt = smem + HG * T::HS;
  float* Ps = smem + 2 * HG * T::HS;  // [KB][2] key positions

  int n0, n1, ltile;
  if (!find_graph(ptr, B, QB, blockIdx.x, &n0, &n1, &ltile)) return;
  const int ng = n1 - n0;
  const int head0 = blockIdx.y * HG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, G = lane >> 4;
  const int q_local = ltile * QB + wave * 16 + j;          // this lane's query (column of S^T)
  const int q_row = n0 + (q_local < ng ? q_local : ng - 1);  // clamped for loads
  const bool q_ok = q_local < ng;

  // Q fragments: lane (q=j, g=G) holds Q[q][h*16 + 4G .. +3], pre-scaled
  f32x4 qf[HG];
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    const float4 t = *reinterpret_cast<const float4*>(Q + (int64_t)q_row * ld + (head0 + h) * 16 + 4 * G);
    qf[h] = f32x4{t.x * qscale, t.y * qscale, t.z * qscale, t.w * qscale};
  }
  const float2 pq = *reinterpret_cast<const float2*>(pos + 2 * (int64_t)q_row);

  f32x4 oacc[HG], oacc2[HG];
  float m[HG], l[HG];
  uint32_t hq[HG];  // per-(lane, head) part of the dropout hash, hoisted out of the key loop
  DropHead dh[HG];
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    oacc[h] = f32x4{0.f, 0.f, 0.f, 0.f}; oacc2[h] = f32x4{0.f, 0.f, 0.f, 0.f}; m[h] = NEG_BIG; l[h] = 0.f;
    dh[h] = DropHead(seed, n0, head0 + h);
    hq[h] = attn_hq(dh[h], q_local);
  }

  // staging map: idx -> (key, h, part), part fastest => HG*64 contiguous bytes per key row
  float4 kreg[F4_PER_THREAD], vreg[F4_PER_THREAD];
  auto issue_loads = [&](int kb0) {
#pragma unroll
    for (int i = 0; i < F4_PER_THREAD; ++i) {
      const int idx = tid + 256 * i;
      const int key = idx / (HG * 4), c = idx % (HG * 4);
      const int kl = kb0 + key;
      if (kl < ng) {
        const int64_t off = (int64_t)(n0 + kl) * ld + head0 * 16 + c * 4;
        kreg[i] = *reinterpret_cast<const float4*>(K + off);
        vreg[i] = *reinterpret_cast<const float4*>(V + off);
      } else {  // zero-fill: 0 * garbage must not poison the accumulators
        kreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        vreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto write_lds = [&](int kb0) {
#pragma unroll
    for (int i = 0; i < F4_PER_THREAD; ++i) {
      const int idx = tid + 256 * i;
      const int key = idx / (HG * 4), c = idx % (HG * 4);
      const int h = c >> 2, part = c & 3;
      *reinterpret_cast<float4*>(&Ks[T::rm(h, key, part * 4)]) = kreg[i];
      Vt[T::tr(h, key, part * 4 + 0)] = vreg[i].x;
      Vt[T::tr(h, key, part * 4 + 1)] = vreg[i].y;
      Vt[T::tr(h, key, part * 4 + 2)] = vreg[i].z;
      Vt[T::tr(h, key, part * 4 + 3)] = vreg[i].w;
    }
    if (tid < KB) {
      const int kl = kb0 + tid;
      float2 p = make_float2(0.f, 0.f);
      if (kl < ng) p = *reinterpret_cast<const float2*>(pos + 2 * (int64_t)(n0 + kl));
      *reinterpret_cast<float2*>(&Ps[2 * tid]) = p;
    }
  };

  issue_loads(0);
  for (int kb0 = 0; kb0 < ng; kb0 += KB) {
    __syncthreads();  // previous block fully consumed
    write_lds(kb0);
    __syncthreads();
    if (kb0 + KB < ng) issue_loads(kb0 + KB);  // prefetch next block under this block's math

    // distance bias for this lane's (keys 16t+4G+r, query j), shared by every head
    float bias[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 pa = *reinterpret_cast<const float4*>(&Ps[2 * (16 * t + 4 * G)]);
      const float4 pb = *reinterpret_cast<const float4*>(&Ps[2 * (16 * t + 4 * G) + 4]);
      const float kx[4] = {pa.x, pa.z, pb.x, pb.z}, ky[4] = {pa.y, pa.w, pb.y, pb.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dx = pq.x - kx[r], dy = pq.y - ky[r];
        const float d = __builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy)) * bscale;
        bias[t][r] = (kb0 + 16 * t + 4 * G + r < ng) ? d : -NEG_BIG;  // masked keys: score -> -1e30
      }
    }

    // Software pipeline over the heads of the group: the S^T MFMAs of head h+1 are issued before the
    // softmax VALU work of head h, so the matrix pipe has work in flight while the VALU runs (MFMA
    // and VALU are separate pipes; in program order they only overlap when independent
    // instructions sit next to each other).  Independent accumulator chains are interleaved
    // (16x16x4: 32-cycle issue, 40-cycle dependent latency): the 4 key tiles of S^T, and two
    // alternating accumulators (oacc / oacc2) for O^T.
    f32x4 sbuf[2][NT];
    auto qk = [&](int h, int buf) {
      f32x4 kf[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        kf[t] = *reinterpret_cast<const f32x4*>(&Ks[T::rm(h, 16 * t + j, 4 * G)]);
        sbuf[buf][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t) sbuf[buf][t] = mfma16(kf[t][e], qf[h][e], sbuf[buf][t]);
    };
    qk(0, 0);
#pragma unroll
    for (int h = 0; h < HG; ++h) {
      const int cur = h & 1;
      if (h + 1 < HG) qk(h + 1, cur ^ 1);
      f32x4 (&s)[NT] = sbuf[cur];
      float mx = NEG_BIG;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[t][r] -= bias[t][r];
        mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
      }
      mx = group_max4(mx);
      const float m_new = fmaxf(m[h], mx);
      const float alpha = __builtin_amdgcn_exp2f(m[h] - m_new);
      m[h] = m_new;
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[t][r] = __builtin_amdgcn_exp2f(s[t][r] - m_new);
          psum += s[t][r];
        }
      l[h] = fmaf(l[h], alpha, psum);
      oacc[h] *= alpha;
      oacc2[h] *= alpha;
      if (DROP) {  // the row sum above uses the un-dropped weights; only the P.V product sees the mask
#pragma unroll
        for (int t = 0; t < NT; ++t) s[t] *= drop_factors_qmajor(hq[h], dh[h], q_local, kb0 + 16 * t + 4 * G, dc);
      }
      f32x4 vf[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) vf[t] = *reinterpret_cast<const f32x4*>(&Vt[T::tr(h, 16 * t + 4 * G, j)]);
      // O^T[d=j][q] += sum_r V^T[d][key 4G+r] * P^T[key 4G+r][q]; even tiles -> oacc, odd -> oacc2
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t & 1) oacc2[h] = mfma16(vf[t][r], s[t][r], oacc2[h]);
          else oacc[h] = mfma16(vf[t][r], s[t][r], oacc[h]);
        }
    }
  }

  // epilogue: lane (q=j, G) holds O[q][h*16 + 4G + r]
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    const float lt = group_sum4(l[h]);
    const float inv = 1.0f / lt;
    if (q_ok) {
      const f32x4 os = oacc[h] + oacc2[h];
      float4 o = make_float4(os[0] * inv, os[1] * inv, os[2] * inv, os[3] * inv);
      *reinterpret_cast<float4*>(O + (int64_t)(n0 + q_local) * ldo + (head0 + h) * 16 + 4 * G) = o;
      if (G == 0) L2[(int64_t)(head0 + h) * N_tot + n0 + q_local] = m[h] + log2f(lt);
    }
  }
}

}  // namespace

// variant: 0 = default tiling; other values select alternative (HG, KB) tilings for tuning runs.
extern "C" int dgdm_spatial_attn_fwd_variant(const float* Q, const float* K, const float* V, int64_t ld, const float* pos,
                                             const int32_t* ptr, int32_t B, int32_t num_q_tiles, int32_t N_tot, int32_t H,
                                             float scale, float inv_tau, float drop_p, uint32_t seed, float* O, int64_t ldo,
                                             float* lse2, int32_t variant, void* stream_) {
  DGDM_REQUIRE(B >= 0 && N_tot >= 0 && H > 0 && num_q_tiles >= 0 && drop_p >= 0.f && drop_p < 1.f);
  if (N_tot == 0 || num_q_tiles == 0) return DGDM_OK;
  DGDM_REQUIRE(Q && K && V && pos && ptr && O && lse2);
  if ((ld & 3) || (ldo & 3) || ld < H * 16 || ldo < H * 16) return DGDM_ERR_UNSUPPORTED;
  if (!dgdm_aligned16(Q) || !dgdm_aligned16(K) || !dgdm_aligned16(V) || !dgdm_aligned16(O) ||
      (reinterpret_cast<uintptr_t>(pos) & 7u))
    return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const float qscale = scale * DGDM_LOG2E, bscale = inv_tau * DGDM_LOG2E;
#define GO(HG, KB)                                                                                                     \
  do {                                                                                                                 \
    if (drop_p > 0.f)                                                                                                  \
      hipLaunchKernelGGL((k_attn_fwd<HG, KB, true>), dim3(num_q_tiles, H / HG), dim3(256), 0, s, Q, K, V, ld, pos, ptr, \
                         B, qscale, bscale, O, ldo, lse2, N_tot, drop_p, dgdm_seed_arg(seed));                                        \
    else                                                                                                               \
      hipLaunchKernelGGL((k_attn_fwd<HG, KB, false>), dim3(num_q_tiles, H / HG), dim3(256), 0, s, Q, K, V, ld, pos,    \
                         ptr, B, qscale, bscale, O, ldo, lse2, N_tot, 0.f, dgdm_seed_arg(0u));                                        \
  } while (0)
  if (H % 8 == 0 && variant == 1) GO(8, 32);
  else if (H % 8 == 0 && variant == 2) GO(8, 64);
  else if (H % 4 == 0 && variant == 3) GO(4, 32);
  else if (H % 4 == 0) GO(4, 64);
  else if (H % 2 == 0) GO(2, 64);
  else GO(1, 64);
#undef GO
  return dgdm_launch_status();
}

extern "C" int dgdm_spatial_attn_fwd(const float* Q, const float* K, const float* V, int64_t ld, const float* pos,
                                     const int32_t* ptr, int32_t B, int32_t num_q_tiles, int32_t N_tot, int32_t H,
                                     float scale, float inv_tau, float drop_p, uint32_t seed, float* O, int64_t ldo,
                                     float* lse2, void* stream) {
  return dgdm_spatial_attn_fwd_variant(Q, K, V, ld, pos, ptr, B, num_q_tiles, N_tot, H, scale, inv_tau, drop_p, seed, O, ldo, lse2,
                                       0, stream);
}

extern "C" int32_t dgdm_spatial_attn_q_tile_rows(void) { return QB; }
