// K4 backward, split-fp16 path, ONE pass for dQ, dK and dV (round 4).
//
// attn_h_bwd.hip recomputes S, dP, exp2, the distance and the dropout word twice -- once per query tile for dQ (lane = query) and
// once per key tile for dK / dV (lane = key) -- because an MFMA product reduces over the index its operands hold in REGISTERS:
// dK / dV reduce over queries (scores as [query rows in registers][key in lanes]), dQ reduces over keys (the transposed
// arrangement).  Here the key-stationary pass of k_attn_h_bwd_dkv also produces dQ:
//   * per (head, query block) every wave has dS'[64 q][its 16 keys] in registers, already split into fp16 hi / lo for the dK
//     product.  The same halfs go through a wave-private LDS tile T[part][key][q] (8-byte writes) and come back TRANSPOSED
//     (ds_read_b64_tr_b16: 4 keys x 16 queries per lane group) as the B operand of
//         dQ^T[d][q] += K^T[d][key] dS'^T[key][q]          (reduction over the wave's 16 keys)
//     with the 32 reduction slots of v_mfma_f32_16x16x32_f16 = 16 keys x {hi, lo}:  A1 = [K_hi | K_hi], A2 = [K_lo | 0],
//     B = [dS_hi ; dS_lo]  =>  A1.B + A2.B = K_hi (dS_hi + dS_lo) + K_lo dS_hi   (lo.lo dropped, as everywhere else).
//     No vector instruction is added: the transposition is LDS traffic, the product two MFMAs per 16 x 16 queries x keys;
//   * the four waves' partial dQ tiles (their 16 keys each) meet in LDS between the two barriers the single-buffered loop has
//     anyway (write before "everyone is done with the block", sum + store while the next block's DMA is in flight), and leave as
//     ONE 64 x 16 fp32 tile per (key block, query block, head) into a partial buffer;
//   * k_attn_dq_reduce sums the key blocks' partials of a query block in key-block order (fixed order: bitwise repeatable, no
//     float atomics) and applies the final scale.
// Cost of the second stage: N^2 H / 64 * 64 B of partials written and read once (3.2 GB at 4 x 10k nodes, 8 heads), against a
// whole second evaluation of every score.  The host cuts the key blocks into groups so that the buffer stays within a budget
// (ops.py); the reduction then accumulates group after group.
#include "attn_h.hpp"

// diagnostic builds only (tools/build_variant_lib.sh -DDGDM_FUSED_SKIP=n): 1 = no partial store, 2 = no cross-wave stage either,
// 4 = no transposition / dQ product at all (what is left is k_attn_h_bwd_dkv at this kernel's occupancy)
#ifndef DGDM_FUSED_SKIP
#define DGDM_FUSED_SKIP 0
#endif

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr int T_LD = 72;                       // halfs per key row of the transposition tile (64 queries + 8: 144-byte rows spread the
                                               // four rows of a transposed read over distinct banks)
constexpr int T_PART = 16 * T_LD;              // halfs per part (hi / lo)
constexpr int T_WAVE = 2 * T_PART;             // halfs per wave

// 4 rows x 16 halfs, transposed: lane (j, G) receives column j of rows 4G .. 4G+3 of a [row][T_LD] tile starting at column c0
__device__ __forceinline__ f16x4 tr4_tile(const _Float16* __restrict__ tile, int c0, int lane) {
  typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int j = lane & 15, G = lane >> 4;
  const _Float16* p = tile + (4 * G + (j >> 2)) * T_LD + c0 + 4 * (j & 3);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  return __builtin_bit_cast(f16x4, a);
}
// the same out of a packed row image (attn_h.hpp): rows 16 t + 4G .. +3 of `part`, column j
__device__ __forceinline__ f16x4 tr4_image(const _Float16* __restrict__ rimg_head, int part, int t, int lane) {
  typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int j = lane & 15, G = lane >> 4;
  const _Float16* p = rimg_head + t * 512 + r_off(4 * G + (j >> 2), 2 * part + ((j & 3) >> 1)) + 4 * (j & 1);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  return __builtin_bit_cast(f16x4, a);
}
__device__ __forceinline__ f16x8 cat4(f16x4 a, f16x4 b) { return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// find_block plus the graph's first PAIR slot: slots number the (key block, query block) pairs of all graphs, graph by graph, key
// block major: slot(kb, qb) = pair0(graph) + (kb - blk0) * nbg + qb.  One 64 x 16 fp32 tile per (slot, head).
__device__ __forceinline__ bool find_block_pairs(const int32_t* __restrict__ ptr, int B, int blk, int* n0, int* ng, int* lblk, int* blk0,
                                                 int64_t* pair0) {
  int base = 0;
  int64_t pairs = 0;
  for (int g = 0; g < B; ++g) {
    const int a = ptr[g], b = ptr[g + 1];
    const int nb = (b - a + HB - 1) / HB;
    if (blk < base + nb) { *n0 = a; *ng = b - a; *lblk = blk - base; *blk0 = base; *pair0 = pairs; return true; }
    base += nb;
    pairs += (int64_t)nb * nb;
  }
  return false;
}

// LDS: stage buffer (Rq | Rg | 8 - lse2 | -delta | pos, as k_attn_h_bwd_dkv) | T (4 waves) | X[HG][4 waves][64 q][16 d] fp32
template <int HG, bool DROP, int WPE = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k_attn_h_bwd_fused(
    const _Float16* __restrict__ Rq, const _Float16* __restrict__ Rk, const _Float16* __restrict__ Rv, const _Float16* __restrict__ Rg,
    const float* __restrict__ pos_b, const float* __restrict__ lse_b, const float* __restrict__ ndelta_b, int H,
    const int32_t* __restrict__ ptr, int B, float kscale, const float* __restrict__ unscale_dev, float* __restrict__ dK,
    float* __restrict__ dV, int64_t ldg, float* __restrict__ dq_part, int kb_first, int kb_count, int64_t slot_first, float drop_p,
    DgdmSeed seed_in) {
  const uint32_t seed = seed_in.value();
  constexpr int NT = HB / 16;
  constexpr int R_BYTES = HG * R_HEAD * 2, SC_BYTES = HG * HB * 4, POS_BYTES = HB * 8;
  constexpr int BUF_BYTES = 2 * R_BYTES + 2 * SC_BYTES + POS_BYTES;
  constexpr int T_BYTES = 4 * T_WAVE * 2;
  constexpr int X_BYTES = HG * 4 * HB * 16 * 4;
  __shared__ __attribute__((aligned(16))) char smem[BUF_BYTES + T_BYTES + X_BYTES];
  const DropCfg dc(drop_p);

  // blockIdx.x counts the key blocks of this launch's group: global packed block = kb_first + blockIdx.x
  int n0, ng, lblk, blk0;
  int64_t pair0;
  if ((int)blockIdx.x >= kb_count || !find_block_pairs(ptr, B, kb_first + blockIdx.x, &n0, &ng, &lblk, &blk0, &pair0)) return;
  const int blk = kb_first + blockIdx.x;
  const int nbg = (ng + HB - 1) / HB;
  const int head0 = blockIdx.y * HG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, G = lane >> 4;
  const int k_in_blk = wave * 16 + j;
  const int k_local = lblk * HB + k_in_blk;
  const bool k_ok = k_local < ng;
  const bool pad_blk = (lblk + 1) * HB > ng;       // block-uniform: this key block runs past the end of its graph

  auto stage = [&](int qb, int buf) {
    (void)buf;
    const int64_t gb = (int64_t)(blk0 + qb) * H + head0;
    char* base = smem;
    dma_to_lds<R_BYTES>(Rq + gb * R_HEAD, base, tid);
    dma_to_lds<R_BYTES>(Rg + gb * R_HEAD, base + R_BYTES, tid);
    dma_to_lds<SC_BYTES>(lse_b + gb * HB, base + 2 * R_BYTES, tid);
    dma_to_lds<SC_BYTES>(ndelta_b + gb * HB, base + 2 * R_BYTES + SC_BYTES, tid);
    dma_to_lds<POS_BYTES>(pos_b + (int64_t)(blk0 + qb) * HB * 2, base + 2 * R_BYTES + 2 * SC_BYTES, tid);
  };
  _Float16* Tw = reinterpret_cast<_Float16*>(smem + BUF_BYTES) + wave * T_WAVE;
  float* X = reinterpret_cast<float*>(smem + BUF_BYTES + T_BYTES);

  // ---- the workgroup's own key block: K^T fragments of the dQ product (loop invariant), out of the K row image staged once
  dma_to_lds<R_BYTES>(Rk + ((int64_t)blk * H + head0) * R_HEAD, smem, tid);
  __syncthreads();
  f16x8 ka1[HG], ka2[HG];
  {
    const _Float16 z = (_Float16)0.0f;
    const f16x4 zero4 = {z, z, z, z};
#pragma unroll
    for (int h = 0; h < HG; ++h) {
      const _Float16* kr = reinterpret_cast<const _Float16*>(smem) + h * R_HEAD;
      const f16x4 khi = tr4_image(kr, 0, wave, lane), klo = tr4_image(kr, 1, wave, lane);
      ka1[h] = cat4(khi, khi);
      ka2[h] = cat4(klo, zero4);
    }
  }
  __syncthreads();
  stage(0, 0);

  f16x8 kb1[HG], kb2[HG], vb1[HG], vb2[HG];   // K and (keep *) V of the lane's key
  f32x4 dk[HG], dv[HG];
  DropLaneK dl[HG];
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    const int64_t imgoff = ((int64_t)blk * H + head0 + h) * R_HEAD;
    load_b_pair(Rk + imgoff, k_in_blk, G, &kb1[h], &kb2[h]);
    load_b_pair(Rv + imgoff, k_in_blk, G, &vb1[h], &vb2[h]);
    if (DROP) scale_b_pair(&vb1[h], &vb2[h], dc.keep);
    dk[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (DROP) dl[h] = DropLaneK(DropHead(seed, n0, head0 + h), k_local);
  }
  const uint32_t lcq = __umul24(2u * (uint32_t)G, DROP_CQ);
  const int aoff = r_lane_off(j, G);
  const float px = pos_b[((int64_t)blk * 2 + 0) * HB + k_in_blk], py = pos_b[((int64_t)blk * 2 + 1) * HB + k_in_blk];
  const int64_t slot0 = pair0 + (int64_t)lblk * nbg - slot_first;     // this key block's first slot inside the launch's scratch
  __syncthreads();

  for (int qb = 0; qb < nbg; ++qb) {
    const char* base = smem;
    const _Float16* Qimg = reinterpret_cast<const _Float16*>(base);
    const _Float16* Gimg = reinterpret_cast<const _Float16*>(base + R_BYTES);
    const float* Ls = reinterpret_cast<const float*>(base + 2 * R_BYTES);
    const float* Ds = Ls + HG * HB;
    const float* Ps = Ds + HG * HB;
    const int qb0 = qb * HB;

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
    if (pad_blk) {         // the graph's last key block: lanes past the graph end hold zero K / V rows.  k_attn_h_bwd_dkv lets them run
                           // (their dK / dV rows are never stored); here their dS' feeds dQ, and a row whose real scores are all far
                           // below zero has P' = exp2(0 - lse2 + 8) = inf there (inf x K = 0 is NaN): P' = 0 for those lanes
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dist[t][r] = k_ok ? dist[t][r] : -NEG_BIG;
    }

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
      const _Float16* gr = Gimg + h * R_HEAD;
      const _Float16* qr = Qimg + h * R_HEAD;
#pragma unroll
      for (int tp = 0; tp < NT / 2; ++tp) {
        f16x8 ph, pl, sh, sl;
        split8(p[2 * tp], p[2 * tp + 1], &ph, &pl);
        split8(ds[2 * tp], ds[2 * tp + 1], &sh, &sl);
        // dS' of the wave's key j for queries 32 tp + 4G .. +3 and 32 tp + 16 + 4G .. +3 -> the transposition tile [part][key][q]
        if (!(DGDM_FUSED_SKIP & 4)) {
          typedef _Float16 h4 __attribute__((ext_vector_type(4)));
          _Float16* th = Tw + j * T_LD + 32 * tp + 4 * G;
          *reinterpret_cast<h4*>(th) = h4{sh[0], sh[1], sh[2], sh[3]};
          *reinterpret_cast<h4*>(th + 16) = h4{sh[4], sh[5], sh[6], sh[7]};
          *reinterpret_cast<h4*>(th + T_PART) = h4{sl[0], sl[1], sl[2], sl[3]};
          *reinterpret_cast<h4*>(th + T_PART + 16) = h4{sl[4], sl[5], sl[6], sl[7]};
        }
        const f16x8 ghi = load_tr_pair(gr, 0, 2 * tp, lane), qhi = load_tr_pair(qr, 0, 2 * tp, lane);
        dv[h] = mfma_h(ghi, ph, dv[h]);                                       // dV^T[d=j][key] += dO^T[d][q] P[q][key]
        dv[h] = mfma_h(load_tr_pair(gr, 1, 2 * tp, lane), ph, dv[h]);
        dv[h] = mfma_h(ghi, pl, dv[h]);
        dk[h] = mfma_h(qhi, sh, dk[h]);                                       // dK^T[d=j][key] += Q'^T[d][q] dS[q][key]
        dk[h] = mfma_h(load_tr_pair(qr, 1, 2 * tp, lane), sh, dk[h]);
        dk[h] = mfma_h(qhi, sl, dk[h]);
      }
      // dQ^T[d][q] of this wave's 16 keys: the tile is this wave's own (its writes above are complete in program order once
      // lgkmcnt has drained: hipcc waits before the dependent reads); lane (q = j, G) supplies reduction slots = keys 4G .. 4G+3, hi | lo
      float* Xw = X + ((h * 4 + wave) * HB) * 16;
#pragma unroll
      for (int t = 0; t < NT && !(DGDM_FUSED_SKIP & 4); ++t) {
        const f16x8 b = cat4(tr4_tile(Tw, 16 * t, lane), tr4_tile(Tw + T_PART, 16 * t, lane));
        f32x4 dqp = mfma_h(ka1[h], b, f32x4{0.f, 0.f, 0.f, 0.f});
        dqp = mfma_h(ka2[h], b, dqp);
        if (!(DGDM_FUSED_SKIP & 2)) *reinterpret_cast<f32x4*>(Xw + (16 * t + j) * 16 + 4 * G) = dqp;      // X[h][wave][q = 16 t + j][d = 4G .. 4G+3]
        else if (dqp[0] == 123.456f) Xw[0] = dqp[1];      // (diagnostic build: keep the product alive)
      }
    }
    __syncthreads();      // everyone is done with the staged block; every wave's dQ tiles are in X
    if (qb + 1 < nbg) stage(qb + 1, 0);
    if (!(DGDM_FUSED_SKIP & 6)) {   // ... and while the next block's DMA is in flight: the four waves' tiles summed (fixed order) and stored as one partial tile
      const int q = tid >> 2, d4 = tid & 3;
      const int64_t slot = (slot0 + qb) * H + head0;                          // [pair slot][head][64 q][16 d]
#pragma unroll
      for (int h = 0; h < HG; ++h) {
        const float* xs = X + (h * 4 * HB + q) * 16 + 4 * d4;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(xs), a1 = *reinterpret_cast<const f32x4*>(xs + HB * 16);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(xs + 2 * HB * 16), a3 = *reinterpret_cast<const f32x4*>(xs + 3 * HB * 16);
        const f32x4 sum = (a0 + a1) + (a2 + a3);
        if (!(DGDM_FUSED_SKIP & 1)) *reinterpret_cast<f32x4*>(dq_part + ((slot + h) * HB + q) * 16 + 4 * d4) = sum;
        else if (sum[0] == 123.456f) dq_part[0] = sum[1];
      }
    }
    __syncthreads();      // the DMA has landed; X and T may be rewritten
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

// dQ[q block][head] (+)= scale * sum over the launch's key blocks of that graph, in key-block order (fixed order: bitwise repeatable).
// grid (all query blocks, H); a thread owns one float4 of the 64 x 16 tile.  A query block whose graph has key blocks in an EARLIER
// launch (blk0 < kb_first) adds to what that launch left in dQ; one whose graph has none in this launch is left alone.
__global__ __launch_bounds__(256) void k_attn_dq_reduce(const float* __restrict__ dq_part, const int32_t* __restrict__ ptr, int B, int H,
                                                        int kb_first, int kb_count, int64_t slot_first, float scale,
                                                        const float* __restrict__ unscale_dev, float* __restrict__ dQ, int64_t ldg) {
  int n0, ng, lblk, blk0;
  int64_t pair0;
  if (!find_block_pairs(ptr, B, blockIdx.x, &n0, &ng, &lblk, &blk0, &pair0)) return;
  const int nbg = (ng + HB - 1) / HB;
  const int h = blockIdx.y, tid = threadIdx.x, q = tid >> 2, d4 = tid & 3;
  const int lo = max(blk0, kb_first), hi = min(blk0 + nbg, kb_first + kb_count);      // this graph's key blocks inside the launch
  if (hi <= lo) return;
  f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  // slot(kb, this query block) = pair0 + (kb - blk0) * nbg + lblk - slot_first: consecutive key blocks are nbg slots apart
  const float* base = dq_part + (((pair0 + lblk - slot_first) * H + h) * HB + q) * 16 + 4 * d4;
  const int64_t stride = (int64_t)nbg * H * HB * 16;
  int kb = lo;
  for (; kb + 3 < hi; kb += 4) {        // four loads in flight; fixed association: ((k0 + k4 + ..) + (k1 + k5 + ..)) + ((k2 ..) + (k3 ..))
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] += *reinterpret_cast<const f32x4*>(base + (int64_t)(kb + u - blk0) * stride);
  }
  for (; kb < hi; ++kb) acc[0] += *reinterpret_cast<const f32x4*>(base + (int64_t)(kb - blk0) * stride);
  const f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  const int q_local = lblk * HB + q;
  if (q_local < ng) {
    const float un = scale * unscale_dev[1] * exp2f(-DGDM_ATTN_P_SHIFT);
    float* o = dQ + (int64_t)(n0 + q_local) * ldg + h * 16 + 4 * d4;
    f32x4 r = sum * un;
    if (blk0 < kb_first) {          // an earlier launch covered the graph's first key blocks
      const float4 old = *reinterpret_cast<const float4*>(o);
      r += f32x4{old.x, old.y, old.z, old.w};
    }
    *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
  }
}

}  // namespace

// First pair slot of packed block `blk` (query block 0) and the slot count of [kb_first, kb_first + kb_count): host mirror of
// find_block_pairs over the HOST copy of the graph offsets.
static bool pair_slots_host(const int32_t* ptr_host, int32_t B, int64_t kb_first, int64_t kb_count, int64_t* first, int64_t* count) {
  int64_t base = 0, pairs = 0, f = -1, l = -1;
  const int64_t kb_last = kb_first + kb_count - 1;
  for (int g = 0; g < B; ++g) {
    const int64_t nb = ((int64_t)ptr_host[g + 1] - ptr_host[g] + HB - 1) / HB;
    if (nb < 0) return false;
    if (f < 0 && kb_first < base + nb) f = pairs + (kb_first - base) * nb;
    if (l < 0 && kb_last < base + nb) l = pairs + (kb_last - base) * nb + nb;     // one past the last slot of the last key block
    base += nb;
    pairs += nb * nb;
  }
  if (f < 0 || l < 0) return false;
  *first = f; *count = l - f;
  return true;
}

// bytes of partial-tile scratch the key blocks [kb_first, kb_first + kb_count) need (one 64 x 16 fp32 tile per (key block, query
// block of its graph, head)); 0 for an empty / out-of-range request.  ptr_host: the B + 1 graph offsets ON THE HOST.
extern "C" size_t dgdm_spatial_attn_h_bwd_fused_workspace_bytes(const int32_t* ptr_host, int32_t B, int32_t H, int32_t kb_first,
                                                                int32_t kb_count) {
  if (!ptr_host || B <= 0 || H <= 0 || kb_first < 0 || kb_count <= 0) return 0;
  int64_t first, count;
  if (!pair_slots_host(ptr_host, B, kb_first, kb_count, &first, &count)) return 0;
  return (size_t)count * (size_t)H * HB * 16 * sizeof(float);
}

extern "C" int dgdm_spatial_attn_h_bwd_fused(const void* Rq, const void* Rk, const void* Rv, const void* Rg, const float* pos_b,
                                             const float* lse_adj_b, const float* ndelta_b, const int32_t* ptr, const int32_t* ptr_host,
                                             int32_t B, int32_t num_blocks, int32_t H, float scale, float drop_p, uint32_t seed,
                                             const float* grad_scale2, float* dQ, float* dK, float* dV, int64_t ldg,
                                             int32_t kb_first, int32_t kb_count, void* workspace, size_t workspace_bytes, void* stream_) {
  DGDM_REQUIRE(B >= 0 && H > 0 && num_blocks >= 0 && drop_p >= 0.f && drop_p < 1.f && kb_first >= 0 && kb_count >= 0);
  if (num_blocks == 0 || B == 0 || kb_count == 0) return DGDM_OK;
  DGDM_REQUIRE(kb_first + (int64_t)kb_count <= num_blocks);
  DGDM_REQUIRE(Rq && Rk && Rv && Rg && pos_b && lse_adj_b && ndelta_b && ptr && ptr_host && dQ && dK && dV && grad_scale2 && workspace);
  if ((ldg & 3) || ldg < H * 16 || !dgdm_aligned16(dQ) || !dgdm_aligned16(dK) || !dgdm_aligned16(dV) || !dgdm_aligned16(workspace))
    return DGDM_ERR_UNSUPPORTED;
  if (H % 2) return DGDM_ERR_UNSUPPORTED;      // two heads per workgroup (odd head counts: the two-pass kernels)
  int64_t slot_first, slots;
  if (!pair_slots_host(ptr_host, B, kb_first, kb_count, &slot_first, &slots)) return DGDM_ERR_INVALID_ARG;
  if (workspace_bytes < (size_t)slots * (size_t)H * HB * 16 * sizeof(float)) return DGDM_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const float kscale = 0.6931471805599453f;  // Q' carries scale*log2(e): dK = sum dS Q' / log2(e)
  auto h16 = [](const void* p) { return static_cast<const _Float16*>(p); };
  float* part = static_cast<float*>(workspace);
  if (drop_p > 0.f)
    hipLaunchKernelGGL((k_attn_h_bwd_fused<2, true, 2>), dim3(kb_count, H / 2), dim3(256), 0, s, h16(Rq), h16(Rk), h16(Rv), h16(Rg), pos_b,
                       lse_adj_b, ndelta_b, H, ptr, B, kscale, grad_scale2, dK, dV, ldg, part, kb_first, kb_count, slot_first, drop_p,
                       dgdm_seed_arg(seed));
  else
    hipLaunchKernelGGL((k_attn_h_bwd_fused<2, false, 2>), dim3(kb_count, H / 2), dim3(256), 0, s, h16(Rq), h16(Rk), h16(Rv), h16(Rg), pos_b,
                       lse_adj_b, ndelta_b, H, ptr, B, kscale, grad_scale2, dK, dV, ldg, part, kb_first, kb_count, slot_first, 0.f,
                       dgdm_seed_arg(0u));
  hipLaunchKernelGGL(k_attn_dq_reduce, dim3(num_blocks, H), dim3(256), 0, s, part, ptr, B, H, kb_first, kb_count, slot_first, scale,
                     grad_scale2, dQ, ldg);
  return dgdm_launch_status();
}
