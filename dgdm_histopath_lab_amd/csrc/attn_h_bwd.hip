// K4 backward, split-fp16 path (see attn_h.hpp / attn_h_fwd.hip).  Same two-pass, atomic-free
// structure as attn_bwd.hip:
//   k_attn_h_bwd_dq : per 64-query tile, sweeps key blocks:   dQ = scale * dS K
//   k_attn_h_bwd_dkv: per 64-key tile, sweeps query blocks:   dV = (P*F)^T dO,  dK = scale * dS^T Q
// with P = exp2(S' - lse2), dP = dO V^T, dS = P * (F*dP - delta)  (F = dropout keep factors: 0 or 1/(1-p)).
// What the VALU does per score is kept to: one exp2, the dropout word (attn_common.hpp, 2.5 instructions), one compare, one or
// two selects, one multiply, and the hi+lo split of what feeds the next product.  Everything else rides on the matrix pipe or
// is prepared per workgroup:
//   * S' - dist - (lse2 - 8) comes out of the S MFMAs: C = (8 - lse2) - dist, one packed add per two scores (positions are
//     pre-scaled, 8 - lse2 is what the pack kernel wrote), P' = exp2(.) = 2^8 P;
//   * keep * dP - delta comes out of the dP MFMAs: the per-workgroup operand (dO in dQ, V in dK/dV) is multiplied by 1/(1-p)
//     once, -delta is the C input; a dropped element then needs -delta itself:  dS' = P' * (kept ? keep dP - delta : -delta);
//   * P and dS are carried as fp16 hi+lo pairs (three MFMAs per product, lo.lo dropped) like every other operand.
// Operands come from the packed images written by dgdm_attn_pack (Q', K, V from the forward's pack; dO, -delta and lse - 8 from
// a second pack call) and are staged by direct-to-LDS DMA.
#include "attn_h.hpp"

// In-kernel time stamps for tools/attn_stamps.py (a diagnostic build with -DDGDM_ATTN_STAMPS; in the library no stamp executes):
// one workgroup (blockIdx.x == 40, blockIdx.y == 0), thread 0, iteration 40 of the block loop writes (s_memtime, s_memrealtime) pairs.
#ifdef DGDM_ATTN_STAMPS
__device__ unsigned long long* dgdm_attn_stamps;
extern "C" __attribute__((visibility("default"))) int dgdm_debug_set_attn_stamps(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(dgdm_attn_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define DGDM_ASTAMP(it_, i_)                                                                              \
  if (blockIdx.x == 40 && blockIdx.y == 0 && threadIdx.x == 0 && (it_) == 40) {                           \
    dgdm_attn_stamps[2 * (i_)] = __builtin_amdgcn_s_memtime();                                            \
    dgdm_attn_stamps[2 * (i_) + 1] = __builtin_amdgcn_s_memrealtime();                                    \
  }
#else
#define DGDM_ASTAMP(it_, i_)
#endif

namespace {

constexpr float NEG_BIG = -1.0e30f;

// ---------------------------------------------------------------------------------------- dQ
// LDS per buffer: Rk | Rv | pos  (K^T of the dQ product is read transposed out of the row image Rk: load_tr_pair)
template <int HG, int NBUF, bool DROP, int WPE = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k_attn_h_bwd_dq(const _Float16* __restrict__ Rq, const _Float16* __restrict__ Rk,
                                                       const _Float16* __restrict__ Rv,
                                                       const _Float16* __restrict__ Rg, const float* __restrict__ pos_b,
                                                       const float* __restrict__ lse_b, const float* __restrict__ ndelta_b, int H,
                                                       const int32_t* __restrict__ ptr, int B, float scale,
                                                       const float* __restrict__ unscale_dev, float* __restrict__ dQ, int64_t ldg,
                                                       float drop_p, DgdmSeed seed_in) {
  const uint32_t seed = seed_in.value();
  constexpr int NT = HB / 16;
  constexpr int R_BYTES = HG * R_HEAD * 2, POS_BYTES = HB * 8;
  constexpr int BUF_BYTES = 2 * R_BYTES + POS_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[NBUF * BUF_BYTES];
  const DropCfg dc(drop_p);

  int n0, ng, lblk, blk0;
  if (!find_block(ptr, B, blockIdx.x, &n0, &ng, &lblk, &blk0)) return;
  const int nbg = (ng + HB - 1) / HB;
  const int head0 = blockIdx.y * HG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, G = lane >> 4;
  const int q_in_blk = wave * 16 + j;
  const int q_local = lblk * HB + q_in_blk;
  const bool q_ok = q_local < ng;

  auto stage = [&](int kb, int buf) {
    const int64_t gb = (int64_t)(blk0 + kb) * H + head0;
    char* base = smem + buf * BUF_BYTES;
    dma_to_lds<R_BYTES>(Rk + gb * R_HEAD, base, tid);
    dma_to_lds<R_BYTES>(Rv + gb * R_HEAD, base + R_BYTES, tid);
    dma_to_lds<POS_BYTES>(pos_b + (int64_t)(blk0 + kb) * HB * 2, base + 2 * R_BYTES, tid);
  };
  stage(0, 0);

  f16x8 qb1[HG], qb2[HG], gb1[HG], gb2[HG];   // Q' and (keep *) dO of the lane's query
  f32x4 dq[HG];
  float nl2[HG], ndl[HG];                     // 8 - lse2, -delta
  DropLaneQ dl[HG];
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    const int64_t rowoff = (((int64_t)blockIdx.x * H + head0 + h) * HB + q_in_blk);
    const int64_t imgoff = ((int64_t)blockIdx.x * H + head0 + h) * R_HEAD;
    load_b_pair(Rq + imgoff, q_in_blk, G, &qb1[h], &qb2[h]);
    load_b_pair(Rg + imgoff, q_in_blk, G, &gb1[h], &gb2[h]);
    if (DROP) scale_b_pair(&gb1[h], &gb2[h], dc.keep);
    nl2[h] = lse_b[rowoff];
    ndl[h] = ndelta_b[rowoff];
    dq[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (DROP) dl[h] = DropLaneQ(DropHead(seed, n0, head0 + h), q_local);
  }
  const uint32_t lck = __umul24(2u * (uint32_t)G, DROP_CK);
  const int aoff = r_lane_off(j, G);
  const float px = pos_b[((int64_t)blockIdx.x * 2 + 0) * HB + q_in_blk], py = pos_b[((int64_t)blockIdx.x * 2 + 1) * HB + q_in_blk];
  __syncthreads();

  for (int kb = 0; kb < nbg; ++kb) {
    const int buf = NBUF == 2 ? (kb & 1) : 0;
    if (NBUF == 2 && kb + 1 < nbg) stage(kb + 1, buf ^ 1);
    const _Float16* Kimg = reinterpret_cast<const _Float16*>(smem + buf * BUF_BYTES);
    const _Float16* Vimg = reinterpret_cast<const _Float16*>(smem + buf * BUF_BYTES + R_BYTES);
    const float* Ps = reinterpret_cast<const float*>(smem + buf * BUF_BYTES + 2 * R_BYTES);
    const int kb0 = kb * HB;

    f32x4 dist[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      dist[t] = dist4(&Ps[16 * t + 4 * G], &Ps[HB + 16 * t + 4 * G], px, py);
    }
    if (kb0 + HB > ng) {   // only the last key block of a graph has keys to mask (wave-uniform branch): P' = exp2(-1e30) = 0
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kb0 + 16 * t + 4 * G + r >= ng) dist[t][r] = -NEG_BIG;
    }

#pragma unroll
    for (int h = 0; h < HG; ++h) {
      f32x4 ds[NT];
      const f32x4 cdl = {ndl[h], ndl[h], ndl[h], ndl[h]};
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(Kimg + h * R_HEAD + t * 512 + aoff);
        const f16x8 vf = *reinterpret_cast<const f16x8*>(Vimg + h * R_HEAD + t * 512 + aoff);
        f32x4 s = mfma_h(kf, qb1[h], sub4(nl2[h], dist[t]));  // S'^T - dist - lse2 + 8
        s = mfma_h(kf, qb2[h], s);
        f32x4 dp = mfma_h(vf, gb1[h], cdl);              // keep * dP^T - delta
        dp = mfma_h(vf, gb2[h], dp);
        if (DROP) {
          uint32_t e[4];
          drop_words_q(dl[h], (uint32_t)(kb0 + 16 * t), lck, e);
#pragma unroll
          for (int r = 0; r < 4; ++r) dp[r] = (int)e[r] >= dc.ts32 ? dp[r] : ndl[h];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[t][r] = __builtin_amdgcn_exp2f(s[r]) * dp[r];
      }
      const _Float16* kr = Kimg + h * R_HEAD;
#pragma unroll
      for (int tp = 0; tp < NT / 2; ++tp) {
        f16x8 sh, sl;
        split8(ds[2 * tp], ds[2 * tp + 1], &sh, &sl);
        const f16x8 khi = load_tr_pair(kr, 0, 2 * tp, lane);
        dq[h] = mfma_h(khi, sh, dq[h]);                                   // dQ^T[d=j][q] += K^T[d][key] dS^T[key][q]
        dq[h] = mfma_h(load_tr_pair(kr, 1, 2 * tp, lane), sh, dq[h]);     // lo part of K: same accumulator (back-to-back
                                                                           // MFMAs on one accumulator issue at full rate)
        dq[h] = mfma_h(khi, sl, dq[h]);                                   // lo part of dS
      }
    }
    __syncthreads();
    if (NBUF == 1 && kb + 1 < nbg) {
      stage(kb + 1, 0);
      __syncthreads();
    }
  }

  if (q_ok) {
    const float un = scale * unscale_dev[1] * exp2f(-DGDM_ATTN_P_SHIFT);
#pragma unroll
    for (int h = 0; h < HG; ++h) {
      const f32x4 o = dq[h] * un;
      *reinterpret_cast<float4*>(dQ + (int64_t)(n0 + q_local) * ldg + (head0 + h) * 16 + 4 * G) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// ------------------------------------------------------------------------------------- dK, dV
// LDS per buffer: Rq | Rg | 8 - lse2 [HG][64] | -delta [HG][64] | pos.  The transposed operands of the dV / dK products (dO^T, Q'^T)
// are read out of the ROW images with transposed LDS reads (load_tr_pair): no transposed image is staged.
template <int HG, int NBUF, bool DROP, int WPE = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k_attn_h_bwd_dkv(const _Float16* __restrict__ Rq,
                                                        const _Float16* __restrict__ Rk, const _Float16* __restrict__ Rv,
                                                        const _Float16* __restrict__ Rg,
                                                        const float* __restrict__ pos_b, const float* __restrict__ lse_b,
                                                        const float* __restrict__ ndelta_b, int H, const int32_t* __restrict__ ptr,
                                                        int B, float kscale, const float* __restrict__ unscale_dev,
                                                        float* __restrict__ dK, float* __restrict__ dV, int64_t ldg, float drop_p,
                                                        DgdmSeed seed_in) {
  const uint32_t seed = seed_in.value();
  constexpr int NT = HB / 16;
  constexpr int R_BYTES = HG * R_HEAD * 2, SC_BYTES = HG * HB * 4, POS_BYTES = HB * 8;
  constexpr int BUF_BYTES = 2 * R_BYTES + 2 * SC_BYTES + POS_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[NBUF * BUF_BYTES];
  const DropCfg dc(drop_p);

  int n0, ng, lblk, blk0;
  if (!find_block(ptr, B, blockIdx.x, &n0, &ng, &lblk, &blk0)) return;
  const int nbg = (ng + HB - 1) / HB;
  const int head0 = blockIdx.y * HG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, G = lane >> 4;
  const int k_in_blk = wave * 16 + j;
  const int k_local = lblk * HB + k_in_blk;
  const bool k_ok = k_local < ng;

  auto stage = [&](int qb, int buf) {
    const int64_t gb = (int64_t)(blk0 + qb) * H + head0;
    char* base = smem + buf * BUF_BYTES;
    dma_to_lds<R_BYTES>(Rq + gb * R_HEAD, base, tid);
    dma_to_lds<R_BYTES>(Rg + gb * R_HEAD, base + R_BYTES, tid);
    dma_to_lds<SC_BYTES>(lse_b + gb * HB, base + 2 * R_BYTES, tid);
    dma_to_lds<SC_BYTES>(ndelta_b + gb * HB, base + 2 * R_BYTES + SC_BYTES, tid);
    dma_to_lds<POS_BYTES>(pos_b + (int64_t)(blk0 + qb) * HB * 2, base + 2 * R_BYTES + 2 * SC_BYTES, tid);
  };
  stage(0, 0);

  f16x8 kb1[HG], kb2[HG], vb1[HG], vb2[HG];   // K and (keep *) V of the lane's key
  f32x4 dk[HG], dv[HG];
  DropLaneK dl[HG];
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    const int64_t imgoff = ((int64_t)blockIdx.x * H + head0 + h) * R_HEAD;
    load_b_pair(Rk + imgoff, k_in_blk, G, &kb1[h], &kb2[h]);
    load_b_pair(Rv + imgoff, k_in_blk, G, &vb1[h], &vb2[h]);
    if (DROP) scale_b_pair(&vb1[h], &vb2[h], dc.keep);
    dk[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (DROP) dl[h] = DropLaneK(DropHead(seed, n0, head0 + h), k_local);
  }
  const uint32_t lcq = __umul24(2u * (uint32_t)G, DROP_CQ);
  const int aoff = r_lane_off(j, G);
  const float px = pos_b[((int64_t)blockIdx.x * 2 + 0) * HB + k_in_blk], py = pos_b[((int64_t)blockIdx.x * 2 + 1) * HB + k_in_blk];
  __syncthreads();

  for (int qb = 0; qb < nbg; ++qb) {
    const int buf = NBUF == 2 ? (qb & 1) : 0;
    if (NBUF == 2 && qb + 1 < nbg) stage(qb + 1, buf ^ 1);
    const char* base = smem + buf * BUF_BYTES;
    const _Float16* Qimg = reinterpret_cast<const _Float16*>(base);
    const _Float16* Gimg = reinterpret_cast<const _Float16*>(base + R_BYTES);
    const float* Ls = reinterpret_cast<const float*>(base + 2 * R_BYTES);
    const float* Ds = Ls + HG * HB;
    const float* Ps = Ds + HG * HB;
    const int qb0 = qb * HB;
    DGDM_ASTAMP(qb, 0)

    // lane (key=j, G), reg r <-> query 16t + 4G + r
    f32x4 dist[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      dist[t] = dist4(&Ps[16 * t + 4 * G], &Ps[HB + 16 * t + 4 * G], px, py);
    }
    if (qb0 + HB > ng) {   // only the last query block of a graph has rows to mask: P' = 0, they contribute nothing
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (qb0 + 16 * t + 4 * G + r >= ng) dist[t][r] = -NEG_BIG;
    }
    DGDM_ASTAMP(qb, 1)

#pragma unroll
    for (int h = 0; h < HG; ++h) {
      f32x4 p[NT], ds[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f16x8 qa = *reinterpret_cast<const f16x8*>(Qimg + h * R_HEAD + t * 512 + aoff);
        const f16x8 ga = *reinterpret_cast<const f16x8*>(Gimg + h * R_HEAD + t * 512 + aoff);
        const f32x4 lq = *reinterpret_cast<const f32x4*>(&Ls[h * HB + 16 * t + 4 * G]);
        const f32x4 nd = *reinterpret_cast<const f32x4*>(&Ds[h * HB + 16 * t + 4 * G]);
        f32x4 s = mfma_h(qa, kb1[h], sub4(lq, dist[t]));   // S'[q][key] - dist - lse2[q] + 8   (lq = 8 - lse2)
        s = mfma_h(qa, kb2[h], s);
        f32x4 dpv = mfma_h(ga, vb1[h], nd);            // keep * dP[q][key] - delta[q]
        dpv = mfma_h(ga, vb2[h], dpv);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[t][r] = __builtin_amdgcn_exp2f(s[r]);
        if (DROP) {
          uint32_t e[4];
          drop_words_k(dl[h], (uint32_t)(qb0 + 16 * t), lcq, e);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool kept = (int)e[r] >= dc.ts32;
            ds[t][r] = p[t][r] * (kept ? dpv[r] : nd[r]);
            p[t][r] = kept ? p[t][r] : 0.f;           // dV sees the dropped weights (1/(1-p) multiplies the finished rows)
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[t][r] = p[t][r] * dpv[r];
        }
      }
      DGDM_ASTAMP(qb, 2 + 2 * h)
      const _Float16* gr = Gimg + h * R_HEAD;
      const _Float16* qr = Qimg + h * R_HEAD;
#pragma unroll
      for (int tp = 0; tp < NT / 2; ++tp) {
        f16x8 ph, pl, sh, sl;
        split8(p[2 * tp], p[2 * tp + 1], &ph, &pl);
        split8(ds[2 * tp], ds[2 * tp + 1], &sh, &sl);
        const f16x8 ghi = load_tr_pair(gr, 0, 2 * tp, lane), qhi = load_tr_pair(qr, 0, 2 * tp, lane);
        dv[h] = mfma_h(ghi, ph, dv[h]);                                       // dV^T[d=j][key] += dO^T[d][q] P[q][key]
        dv[h] = mfma_h(load_tr_pair(gr, 1, 2 * tp, lane), ph, dv[h]);
        dv[h] = mfma_h(ghi, pl, dv[h]);
        dk[h] = mfma_h(qhi, sh, dk[h]);                                       // dK^T[d=j][key] += Q'^T[d][q] dS[q][key]
        dk[h] = mfma_h(load_tr_pair(qr, 1, 2 * tp, lane), sh, dk[h]);
        dk[h] = mfma_h(qhi, sl, dk[h]);
      }
      DGDM_ASTAMP(qb, 3 + 2 * h)
    }
    DGDM_ASTAMP(qb, 10)
    __syncthreads();
    DGDM_ASTAMP(qb, 11)
    if (NBUF == 1 && qb + 1 < nbg) {
      stage(qb + 1, 0);
      DGDM_ASTAMP(qb, 12)
      __syncthreads();
      DGDM_ASTAMP(qb, 13)
    }
  }

  if (k_ok) {
    const float un = unscale_dev[1] * exp2f(-DGDM_ATTN_P_SHIFT);
    const float unk = kscale * un, unv = DROP ? dc.keep * un : un;
#pragma unroll
    for (int h = 0; h < HG; ++h) {
      const int64_t off = (int64_t)(n0 + k_local) * ldg + (head0 + h) * 16 + 4 * G;
      const f32x4 a = dk[h] * unk, b = dv[h] * unv;
      *reinterpret_cast<float4*>(dK + off) = make_float4(a[0], a[1], a[2], a[3]);
      *reinterpret_cast<float4*>(dV + off) = make_float4(b[0], b[1], b[2], b[3]);
    }
  }
}

}  // namespace

extern "C" int dgdm_spatial_attn_h_bwd_dq(const void* Rq, const void* Rk, const void* Rv, const void* Rg,
                                          const float* pos_b, const float* lse_adj_b, const float* ndelta_b, const int32_t* ptr, int32_t B,
                                          int32_t num_blocks, int32_t H, float scale, float drop_p, uint32_t seed,
                                          const float* grad_scale2, float* dQ, int64_t ldg, int32_t variant, void* stream_) {
  DGDM_REQUIRE(B >= 0 && H > 0 && num_blocks >= 0 && drop_p >= 0.f && drop_p < 1.f);
  if (num_blocks == 0 || B == 0) return DGDM_OK;
  DGDM_REQUIRE(Rq && Rk && Rv && Rg && pos_b && lse_adj_b && ndelta_b && ptr && dQ && grad_scale2);
  if ((ldg & 3) || ldg < H * 16 || !dgdm_aligned16(dQ)) return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  auto h16 = [](const void* p) { return static_cast<const _Float16*>(p); };
#define GOW(HG, NBUF, WPE)                                                                                                        \
  do {                                                                                                                           \
    if (drop_p > 0.f)                                                                                                            \
      hipLaunchKernelGGL((k_attn_h_bwd_dq<HG, NBUF, true, WPE>), dim3(num_blocks, H / HG), dim3(256), 0, s, h16(Rq), h16(Rk), h16(Rv), \
                         h16(Rg), pos_b, lse_adj_b, ndelta_b, H, ptr, B, scale, grad_scale2, dQ, ldg, drop_p, dgdm_seed_arg(seed));             \
    else                                                                                                                         \
      hipLaunchKernelGGL((k_attn_h_bwd_dq<HG, NBUF, false, WPE>), dim3(num_blocks, H / HG), dim3(256), 0, s, h16(Rq), h16(Rk),        \
                         h16(Rv), h16(Rg), pos_b, lse_adj_b, ndelta_b, H, ptr, B, scale, grad_scale2, dQ, ldg, 0.f, dgdm_seed_arg(0u));         \
  } while (0)
  if (H % 2 == 0 && variant == 1) GOW(2, 2, 1);
  else if (H % 4 == 0 && variant == 2) GOW(4, 2, 1);
  else if (H % 2 == 0 && variant == 3) GOW(2, 1, 1);
  else if (H % 2 == 0 && variant == 4) GOW(2, 1, 4);
  else if (H % 4 == 0 && variant == 5) GOW(4, 1, 3);
  else if (H % 4 == 0) GOW(4, 1, 1);
  else if (H % 2 == 0) GOW(2, 1, 1);
  else GOW(1, 2, 1);
#undef GOW
  return dgdm_launch_status();
}

extern "C" int dgdm_spatial_attn_h_bwd_dkv(const void* Rq, const void* Rk, const void* Rv, const void* Rg,
                                           const float* pos_b, const float* lse_adj_b, const float* ndelta_b,
                                           const int32_t* ptr, int32_t B, int32_t num_blocks, int32_t H, float drop_p,
                                           uint32_t seed, const float* grad_scale2, float* dK, float* dV, int64_t ldg, int32_t variant,
                                           void* stream_) {
  DGDM_REQUIRE(B >= 0 && H > 0 && num_blocks >= 0 && drop_p >= 0.f && drop_p < 1.f);
  if (num_blocks == 0 || B == 0) return DGDM_OK;
  DGDM_REQUIRE(Rq && Rk && Rv && Rg && pos_b && lse_adj_b && ndelta_b && ptr && dK && dV && grad_scale2);
  if ((ldg & 3) || ldg < H * 16 || !dgdm_aligned16(dK) || !dgdm_aligned16(dV)) return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const float kscale = 0.6931471805599453f;  // Q' carries scale*log2(e): dK = sum dS Q' / log2(e)
  auto h16 = [](const void* p) { return static_cast<const _Float16*>(p); };
#define GOW(HG, NBUF, WPE)                                                                                                          \
  do {                                                                                                                             \
    if (drop_p > 0.f)                                                                                                              \
      hipLaunchKernelGGL((k_attn_h_bwd_dkv<HG, NBUF, true, WPE>), dim3(num_blocks, H / HG), dim3(256), 0, s, h16(Rq), h16(Rk),  \
                         h16(Rv), h16(Rg), pos_b, lse_adj_b, ndelta_b, H, ptr, B, kscale, grad_scale2, dK, dV, ldg, drop_p, dgdm_seed_arg(seed)); \
    else                                                                                                                           \
      hipLaunchKernelGGL((k_attn_h_bwd_dkv<HG, NBUF, false, WPE>), dim3(num_blocks, H / HG), dim3(256), 0, s, h16(Rq), h16(Rk), \
                         h16(Rv), h16(Rg), pos_b, lse_adj_b, ndelta_b, H, ptr, B, kscale, grad_scale2, dK, dV, ldg, 0.f, dgdm_seed_arg(0u));      \
  } while (0)
  if (H % 4 == 0 && variant == 1) GOW(4, 1, 1);
  else if (H % 4 == 0 && variant == 2) GOW(4, 2, 1);
  else if (H % 2 == 0 && variant == 3) GOW(2, 1, 1);
  else if (H % 2 == 0 && variant == 4) GOW(2, 1, 3);   // 3 workgroups per CU (registers capped at 168)
  else if (H % 2 == 0 && variant == 5) GOW(2, 2, 3);
  else if (H % 2 == 0) GOW(2, 1, 3);  // default: 2 heads per group, one staging buffer, registers capped at 168 => three workgroups
  else GOW(1, 2, 1);                  // per CU (2.11 vs 2.22 ms for the double-buffered two-per-CU form at 4 x 10k nodes)
#undef GOW
  return dgdm_launch_status();
}
