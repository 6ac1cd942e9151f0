// K4 backward: gradients of the fused spatial attention wrt Q, K, V (the distance bias has no
// learnable parameter: core/attention.py:274-281, `spatial_proj`/`pos_encoding` are unused).
// Autograd of the reference materialises P [H,N,N] and dP; here P is recomputed per tile from
// Q, K, pos and the forward's log2-sum-exp.
//
// Two launches, no float atomics (bitwise reproducible):
//   k_attn_bwd_dq : one workgroup per 64-query tile sweeps the keys:   dQ = scale * dS K
//                   (also writes delta[h][q] = rowsum(dO * O))
//   k_attn_bwd_dkv: one workgroup per 64-key tile sweeps the queries:  dV = P^T dO,
//                   dK = scale * dS^T Q
// with P = exp2(S' - lse2), dP = dO V^T, dS = P * (dP - delta).  Each kernel orients its first
// products so that the accumulator holding P / dS is directly the B operand of the following
// product (attn_common.hpp), so neither ever crosses LDS.
#include "attn_common.hpp"

namespace {

constexpr int TB = 64;  // rows owned by a workgroup (4 waves x 16): queries (dq) / keys (dkv)
constexpr float BIG = 1.0e30f;

// ---------------------------------------------------------------------------------------- dQ
template <int HG, int KB, bool DROP>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dq(const float* __restrict__ Q, const float* __restrict__ K,
                                                        const float* __restrict__ V, int64_t ld,
                                                        const float* __restrict__ Oa, const float* __restrict__ dO, int64_t ldo,
                                                        const float* __restrict__ pos, const int32_t* __restrict__ ptr, int B,
                                                        const float* __restrict__ L2, float qscale, float bscale, float scale,
                                                        float* __restrict__ dQ, int64_t ldg, float* __restrict__ delta,
                                                        int N_tot, float drop_p, DgdmSeed seed_in) {
  const uint32_t seed = seed_in.value();
  using T = AttnTile<KB>;
  const DropCfg dc(drop_p);
  constexpr int NT = KB / 16;
  constexpr int F4 = KB * HG * 4 / 256;
  static_assert(KB % 16 == 0 && (KB * HG * 4) % 256 == 0, "staging must divide evenly");
  __shared__ __attribute__((aligned(16))) float smem[3 * HG * T::HS + 2 * KB];
  float* Ks = smem;                    // K row-major  (A of S^T = K Q^T)
  float* Kt = smem + HG * T::HS;       // K transposed (A of dQ^T += K^T dS^T)
  float* Vs = smem + 2 * HG * T::HS;   // V row-major  (A of dP^T = V dO^T)
  float* Ps = smem + 3 * HG * T::HS;

  int n0, n1, ltile;
  if (!find_graph(ptr, B, TB, blockIdx.x, &n0, &n1, &ltile)) return;
  const int ng = n1 - n0;
  const int head0 = blockIdx.y * HG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, G = lane >> 4;
  const int q_local = ltile * TB + wave * 16 + j;
  const bool q_ok = q_local < ng;
  const int q_row = n0 + (q_ok ? q_local : ng - 1);

  f32x4 qf[HG], dof[HG], dq[HG];
  float l2[HG], dl[HG];
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    const int col = (head0 + h) * 16 + 4 * G;
    const float4 a = *reinterpret_cast<const float4*>(Q + (int64_t)q_row * ld + col);
    const float4 g = *reinterpret_cast<const float4*>(dO + (int64_t)q_row * ldo + col);
    const float4 o = *reinterpret_cast<const float4*>(Oa + (int64_t)q_row * ldo + col);
    qf[h] = f32x4{a.x * qscale, a.y * qscale, a.z * qscale, a.w * qscale};
    dof[h] = f32x4{g.x, g.y, g.z, g.w};
    dl[h] = group_sum4(g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w);
    l2[h] = L2[(int64_t)(head0 + h) * N_tot + q_row];
    dq[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q_ok && G == 0) delta[(int64_t)(head0 + h) * N_tot + q_row] = dl[h];
  }
  const float2 pq = *reinterpret_cast<const float2*>(pos + 2 * (int64_t)q_row);

  float4 kreg[F4], vreg[F4];
  auto issue_loads = [&](int kb0) {
#pragma unroll
    for (int i = 0; i < F4; ++i) {
      const int idx = tid + 256 * i;
      const int key = idx / (HG * 4), c = idx % (HG * 4);
      const int kl = kb0 + key;
      if (kl < ng) {
        const int64_t off = (int64_t)(n0 + kl) * ld + head0 * 16 + c * 4;
        kreg[i] = *reinterpret_cast<const float4*>(K + off);
        vreg[i] = *reinterpret_cast<const float4*>(V + off);
      } else {
        kreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        vreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto write_lds = [&](int kb0) {
#pragma unroll
    for (int i = 0; i < F4; ++i) {
      const int idx = tid + 256 * i;
      const int key = idx / (HG * 4), c = idx % (HG * 4);
      const int h = c >> 2, part = c & 3;
      *reinterpret_cast<float4*>(&Ks[T::rm(h, key, part * 4)]) = kreg[i];
      *reinterpret_cast<float4*>(&Vs[T::rm(h, key, part * 4)]) = vreg[i];
      Kt[T::tr(h, key, part * 4 + 0)] = kreg[i].x;
      Kt[T::tr(h, key, part * 4 + 1)] = kreg[i].y;
      Kt[T::tr(h, key, part * 4 + 2)] = kreg[i].z;
      Kt[T::tr(h, key, part * 4 + 3)] = kreg[i].w;
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
    __syncthreads();
    write_lds(kb0);
    __syncthreads();
    if (kb0 + KB < ng) issue_loads(kb0 + KB);

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
        bias[t][r] = (kb0 + 16 * t + 4 * G + r < ng) ? d : BIG;
      }
    }

#pragma unroll
    for (int h = 0; h < HG; ++h) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[T::rm(h, 16 * t + j, 4 * G)]);
        const f32x4 vf = *reinterpret_cast<const f32x4*>(&Vs[T::rm(h, 16 * t + j, 4 * G)]);
        f32x4 s = mfma16_k16(kf, qf[h], f32x4{0.f, 0.f, 0.f, 0.f});    // S'^T[key][q]
        f32x4 dp = mfma16_k16(vf, dof[h], f32x4{0.f, 0.f, 0.f, 0.f});  // dP^T[key][q]
        f32x4 ds;
        if (DROP) {
          const DropHead dhh(seed, n0, head0 + h);
          dp *= drop_factors_qmajor(attn_hq(dhh, q_local), dhh, q_local, kb0 + 16 * t + 4 * G, dc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[r] - bias[t][r] - l2[h]);
          ds[r] = p * (dp[r] - dl[h]);
        }
        const f32x4 ktf = *reinterpret_cast<const f32x4*>(&Kt[T::tr(h, 16 * t + 4 * G, j)]);
        dq[h] = mfma16(ktf[0], ds[0], dq[h]);  // dQ^T[d=j][q] += K^T[d][key 4G+r] dS^T[key][q]
        dq[h] = mfma16(ktf[1], ds[1], dq[h]);
        dq[h] = mfma16(ktf[2], ds[2], dq[h]);
        dq[h] = mfma16(ktf[3], ds[3], dq[h]);
      }
    }
  }

  if (q_ok) {
#pragma unroll
    for (int h = 0; h < HG; ++h) {
      const float4 o = make_float4(dq[h][0] * scale, dq[h][1] * scale, dq[h][2] * scale, dq[h][3] * scale);
      *reinterpret_cast<float4*>(dQ + (int64_t)(n0 + q_local) * ldg + (head0 + h) * 16 + 4 * G) = o;
    }
  }
}

// ------------------------------------------------------------------------------------- dK, dV
template <int HG, int QBK, bool DROP>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dkv(const float* __restrict__ Q, const float* __restrict__ K,
                                                         const float* __restrict__ V, int64_t ld,
                                                         const float* __restrict__ dO, int64_t ldo,
                                                         const float* __restrict__ pos, const int32_t* __restrict__ ptr, int B,
                                                         const float* __restrict__ L2, const float* __restrict__ delta,
                                                         float qscale, float bscale, float scale, float* __restrict__ dK,
                                                         float* __restrict__ dV, int64_t ldg, int N_tot, float drop_p,
                                                         DgdmSeed seed_in) {
  const uint32_t seed = seed_in.value();
  using T = AttnTile<QBK>;
  const DropCfg dc(drop_p);
  constexpr int NT = QBK / 16;
  constexpr int F4 = QBK * HG * 4 / 256;
  static_assert(QBK % 16 == 0 && (QBK * HG * 4) % 256 == 0, "staging must divide evenly");
  constexpr int IMG = HG * T::HS;
  __shared__ __attribute__((aligned(16))) float smem[4 * IMG + 2 * HG * QBK + 2 * QBK];
  float* Qs = smem;             // Q  row-major  (A of S = Q K^T)
  float* Qt = smem + IMG;       // Q  transposed (A of dK^T += Q^T dS)
  float* Gs = smem + 2 * IMG;   // dO row-major  (A of dP = dO V^T)
  float* Gt = smem + 3 * IMG;   // dO transposed (A of dV^T += dO^T P)
  float* Ls = smem + 4 * IMG;             // lse2  [HG][QBK]
  float* Ds = Ls + HG * QBK;              // delta [HG][QBK]
  float* Ps = Ds + HG * QBK;              // query positions [QBK][2]

  int n0, n1, ltile;
  if (!find_graph(ptr, B, TB, blockIdx.x, &n0, &n1, &ltile)) return;
  const int ng = n1 - n0;
  const int head0 = blockIdx.y * HG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, G = lane >> 4;
  const int k_local = ltile * TB + wave * 16 + j;  // this lane's key (column of S)
  const bool k_ok = k_local < ng;
  const int k_row = n0 + (k_ok ? k_local : ng - 1);

  f32x4 kf[HG], vf[HG], dk[HG], dv[HG];
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    const int col = (head0 + h) * 16 + 4 * G;
    const float4 a = *reinterpret_cast<const float4*>(K + (int64_t)k_row * ld + col);
    const float4 b = *reinterpret_cast<const float4*>(V + (int64_t)k_row * ld + col);
    kf[h] = f32x4{a.x * qscale, a.y * qscale, a.z * qscale, a.w * qscale};
    vf[h] = f32x4{b.x, b.y, b.z, b.w};
    dk[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[h] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float2 pk = *reinterpret_cast<const float2*>(pos + 2 * (int64_t)k_row);

  float4 qreg[F4], greg[F4];
  auto issue_loads = [&](int qb0) {
#pragma unroll
    for (int i = 0; i < F4; ++i) {
      const int idx = tid + 256 * i;
      const int q = idx / (HG * 4), c = idx % (HG * 4);
      const int ql = qb0 + q;
      if (ql < ng) {
        qreg[i] = *reinterpret_cast<const float4*>(Q + (int64_t)(n0 + ql) * ld + head0 * 16 + c * 4);
        greg[i] = *reinterpret_cast<const float4*>(dO + (int64_t)(n0 + ql) * ldo + head0 * 16 + c * 4);
      } else {  // masked queries: dO = 0 and (below) score = -1e30 => they contribute nothing
        qreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        greg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto write_lds = [&](int qb0) {
#pragma unroll
    for (int i = 0; i < F4; ++i) {
      const int idx = tid + 256 * i;
      const int q = idx / (HG * 4), c = idx % (HG * 4);
      const int h = c >> 2, part = c & 3;
      *reinterpret_cast<float4*>(&Qs[T::rm(h, q, part * 4)]) = qreg[i];
      *reinterpret_cast<float4*>(&Gs[T::rm(h, q, part * 4)]) = greg[i];
      Qt[T::tr(h, q, part * 4 + 0)] = qreg[i].x; Gt[T::tr(h, q, part * 4 + 0)] = greg[i].x;
      Qt[T::tr(h, q, part * 4 + 1)] = qreg[i].y; Gt[T::tr(h, q, part * 4 + 1)] = greg[i].y;
      Qt[T::tr(h, q, part * 4 + 2)] = qreg[i].z; Gt[T::tr(h, q, part * 4 + 2)] = greg[i].z;
      Qt[T::tr(h, q, part * 4 + 3)] = qreg[i].w; Gt[T::tr(h, q, part * 4 + 3)] = greg[i].w;
    }
    for (int idx = tid; idx < HG * QBK; idx += 256) {
      const int h = idx / QBK, q = idx % QBK;
      const int ql = qb0 + q;
      const int64_t o = (int64_t)(head0 + h) * N_tot + n0 + ql;
      Ls[idx] = ql < ng ? L2[o] : 0.f;
      Ds[idx] = ql < ng ? delta[o] : 0.f;
    }
    if (tid < QBK) {
      const int ql = qb0 + tid;
      float2 p = make_float2(0.f, 0.f);
      if (ql < ng) p = *reinterpret_cast<const float2*>(pos + 2 * (int64_t)(n0 + ql));
      *reinterpret_cast<float2*>(&Ps[2 * tid]) = p;
    }
  };

  issue_loads(0);
  for (int qb0 = 0; qb0 < ng; qb0 += QBK) {
    __syncthreads();
    write_lds(qb0);
    __syncthreads();
    if (qb0 + QBK < ng) issue_loads(qb0 + QBK);

    // lane (key=j, G), reg r <-> query 16t + 4G + r
    float bias[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 pa = *reinterpret_cast<const float4*>(&Ps[2 * (16 * t + 4 * G)]);
      const float4 pb = *reinterpret_cast<const float4*>(&Ps[2 * (16 * t + 4 * G) + 4]);
      const float qx[4] = {pa.x, pa.z, pb.x, pb.z}, qy[4] = {pa.y, pa.w, pb.y, pb.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dx = pk.x - qx[r], dy = pk.y - qy[r];
        const float d = __builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy)) * bscale;
        bias[t][r] = (qb0 + 16 * t + 4 * G + r < ng) ? d : BIG;
      }
    }

#pragma unroll
    for (int h = 0; h < HG; ++h) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 qa = *reinterpret_cast<const f32x4*>(&Qs[T::rm(h, 16 * t + j, 4 * G)]);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(&Gs[T::rm(h, 16 * t + j, 4 * G)]);
        const f32x4 lq = *reinterpret_cast<const f32x4*>(&Ls[h * QBK + 16 * t + 4 * G]);
        const f32x4 dq = *reinterpret_cast<const f32x4*>(&Ds[h * QBK + 16 * t + 4 * G]);
        f32x4 s = mfma16_k16(qa, kf[h], f32x4{0.f, 0.f, 0.f, 0.f});   // S'[q][key]
        f32x4 dp = mfma16_k16(ga, vf[h], f32x4{0.f, 0.f, 0.f, 0.f});  // dP[q][key]
        f32x4 p, ds, f = f32x4{1.f, 1.f, 1.f, 1.f};
        if (DROP) f = drop_factors_kmajor(DropHead(seed, n0, head0 + h), k_local, qb0 + 16 * t + 4 * G, dc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = __builtin_amdgcn_exp2f(s[r] - bias[t][r] - lq[r]);
          ds[r] = p[r] * (f[r] * dp[r] - dq[r]);
          p[r] *= f[r];  // dV sees the dropped weights
        }
        const f32x4 gt = *reinterpret_cast<const f32x4*>(&Gt[T::tr(h, 16 * t + 4 * G, j)]);
        const f32x4 qt = *reinterpret_cast<const f32x4*>(&Qt[T::tr(h, 16 * t + 4 * G, j)]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dv[h] = mfma16(gt[r], p[r], dv[h]);   // dV^T[d=j][key] += dO^T[d][q 4G+r] P[q][key]
          dk[h] = mfma16(qt[r], ds[r], dk[h]);  // dK^T[d=j][key] += Q^T[d][q 4G+r] dS[q][key]
        }
      }
    }
  }

  if (k_ok) {
#pragma unroll
    for (int h = 0; h < HG; ++h) {
      const int64_t off = (int64_t)(n0 + k_local) * ldg + (head0 + h) * 16 + 4 * G;
      *reinterpret_cast<float4*>(dK + off) = make_float4(dk[h][0] * scale, dk[h][1] * scale, dk[h][2] * scale, dk[h][3] * scale);
      *reinterpret_cast<float4*>(dV + off) = make_float4(dv[h][0], dv[h][1], dv[h][2], dv[h][3]);
    }
  }
}

}  // namespace

static int check_bwd_common(float drop_p, const void* Q, const void* K, const void* V, const void* dO, const void* pos, const void* ptr,
                            const void* lse2, int64_t ld, int64_t ldo, int64_t ldg, int32_t B, int32_t N_tot, int32_t H,
                            int32_t num_q_tiles) {
  if (B < 0 || N_tot < 0 || H <= 0 || num_q_tiles < 0 || !(drop_p >= 0.f && drop_p < 1.f)) return DGDM_ERR_INVALID_ARG;
  if (N_tot == 0 || num_q_tiles == 0) return 1;  // nothing to do
  if (!Q || !K || !V || !dO || !pos || !ptr || !lse2) return DGDM_ERR_INVALID_ARG;
  if ((ld & 3) || (ldo & 3) || (ldg & 3) || ld < H * 16 || ldo < H * 16 || ldg < H * 16) return DGDM_ERR_UNSUPPORTED;
  if (!dgdm_aligned16(Q) || !dgdm_aligned16(K) || !dgdm_aligned16(V) || !dgdm_aligned16(dO) ||
      (reinterpret_cast<uintptr_t>(pos) & 7u))
    return DGDM_ERR_UNSUPPORTED;
  return DGDM_OK;
}

// pass 1: dQ (+ delta = rowsum(dO*O) into delta_ws)
extern "C" int dgdm_spatial_attn_bwd_dq(const float* Q, const float* K, const float* V, int64_t ld, const float* O,
                                        const float* dO, int64_t ldo, const float* pos, const int32_t* ptr, int32_t B,
                                        int32_t num_q_tiles, int32_t N_tot, int32_t H, float scale, float inv_tau,
                                        const float* lse2, float drop_p, uint32_t seed, float* dQ, int64_t ldg, float* delta_ws,
                                        void* stream_) {
  int rc = check_bwd_common(drop_p, Q, K, V, dO, pos, ptr, lse2, ld, ldo, ldg, B, N_tot, H, num_q_tiles);
  if (rc != DGDM_OK) return rc > 0 ? DGDM_OK : rc;
  DGDM_REQUIRE(O && dQ && delta_ws);
  if (!dgdm_aligned16(O) || !dgdm_aligned16(dQ)) return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const float qscale = scale * DGDM_LOG2E, bscale = inv_tau * DGDM_LOG2E;
#define GO(HG, KB)                                                                                                        \
  do {                                                                                                                    \
    if (drop_p > 0.f)                                                                                                     \
      hipLaunchKernelGGL((k_attn_bwd_dq<HG, KB, true>), dim3(num_q_tiles, H / HG), dim3(256), 0, s, Q, K, V, ld, O, dO,   \
                         ldo, pos, ptr, B, lse2, qscale, bscale, scale, dQ, ldg, delta_ws, N_tot, drop_p, dgdm_seed_arg(seed));         \
    else                                                                                                                  \
      hipLaunchKernelGGL((k_attn_bwd_dq<HG, KB, false>), dim3(num_q_tiles, H / HG), dim3(256), 0, s, Q, K, V, ld, O, dO,  \
                         ldo, pos, ptr, B, lse2, qscale, bscale, scale, dQ, ldg, delta_ws, N_tot, 0.f, dgdm_seed_arg(0u));              \
  } while (0)
  if (H % 4 == 0) GO(4, 64);
  else if (H % 2 == 0) GO(2, 64);
  else GO(1, 64);
#undef GO
  return dgdm_launch_status();
}

// pass 2: dK, dV (reads delta_ws written by pass 1 on the same stream)
extern "C" int dgdm_spatial_attn_bwd_dkv(const float* Q, const float* K, const float* V, int64_t ld, const float* dO,
                                         int64_t ldo, const float* pos, const int32_t* ptr, int32_t B, int32_t num_q_tiles,
                                         int32_t N_tot, int32_t H, float scale, float inv_tau, const float* lse2,
                                         const float* delta_ws, float drop_p, uint32_t seed, float* dK, float* dV, int64_t ldg,
                                         void* stream_) {
  int rc = check_bwd_common(drop_p, Q, K, V, dO, pos, ptr, lse2, ld, ldo, ldg, B, N_tot, H, num_q_tiles);
  if (rc != DGDM_OK) return rc > 0 ? DGDM_OK : rc;
  DGDM_REQUIRE(delta_ws && dK && dV);
  if (!dgdm_aligned16(dK) || !dgdm_aligned16(dV)) return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const float qscale = scale * DGDM_LOG2E, bscale = inv_tau * DGDM_LOG2E;
#define GO(HG, QBK)                                                                                                       \
  do {                                                                                                                    \
    if (drop_p > 0.f)                                                                                                     \
      hipLaunchKernelGGL((k_attn_bwd_dkv<HG, QBK, true>), dim3(num_q_tiles, H / HG), dim3(256), 0, s, Q, K, V, ld, dO,    \
                         ldo, pos, ptr, B, lse2, delta_ws, qscale, bscale, scale, dK, dV, ldg, N_tot, drop_p, dgdm_seed_arg(seed));     \
    else                                                                                                                  \
      hipLaunchKernelGGL((k_attn_bwd_dkv<HG, QBK, false>), dim3(num_q_tiles, H / HG), dim3(256), 0, s, Q, K, V, ld, dO,   \
                         ldo, pos, ptr, B, lse2, delta_ws, qscale, bscale, scale, dK, dV, ldg, N_tot, 0.f, dgdm_seed_arg(0u));          \
  } while (0)
  if (H % 4 == 0) GO(4, 32);
  else if (H % 2 == 0) GO(2, 64);
  else GO(1, 64);
#undef GO
  return dgdm_launch_status();
}

extern "C" int dgdm_spatial_attn_bwd(const float* Q, const float* K, const float* V, int64_t ld, const float* O,
                                     const float* dO, int64_t ldo, const float* pos, const int32_t* ptr, int32_t B,
                                     int32_t num_q_tiles, int32_t N_tot, int32_t H, float scale, float inv_tau,
                                     const float* lse2, float drop_p, uint32_t seed, float* dQ, float* dK, float* dV, int64_t ldg,
                                     float* delta_ws, void* stream) {
  int rc = dgdm_spatial_attn_bwd_dq(Q, K, V, ld, O, dO, ldo, pos, ptr, B, num_q_tiles, N_tot, H, scale, inv_tau, lse2, drop_p, seed,
                                    dQ, ldg, delta_ws, stream);
  if (rc != DGDM_OK) return rc;
  return dgdm_spatial_attn_bwd_dkv(Q, K, V, ld, dO, ldo, pos, ptr, B, num_q_tiles, N_tot, H, scale, inv_tau, lse2, delta_ws, drop_p,
                                   seed, dK, dV, ldg, stream);
}
