// K4-dense: MultiHeadAttention.forward as the reference exposes it (core/attention.py:73-181): a dense batch [B, Lq] of queries
// against [B, Lk] keys / values with everything its score tensor may receive before the softmax --
//   S[b,h,q,k] = Q[b,q,h].K[b,k,h] * scale  +  attn_mask (float: added; bool: -inf where set; any shape that broadcasts to
//                [B, H, Lq, Lk], given here by its four strides)  -  |pos_q - pos_k| * inv_tau (SpatialAttention.forward with a
//                mask: attention.py:311-314)  ;  -inf where key_padding_mask[b, k]          (attention.py:129-142)
//   O = dropout(softmax_k S) V                                                              (attention.py:145-149)
// The DGDM model itself never comes here: its two uses of the class run on the fused variable-length kernels (attn_h_*.hip,
// segment.hip).  This file closes the class's own forward for callers that use it directly: plain fp32 on the vector units,
// flash-style (no [Lq, Lk] tensor but the optional weights output), one thread (two at head_dim 128) per query row (forward, dQ,
// weights) or per key row (dK, dV); the streamed side goes through LDS in tiles and the additive term of a 64 x KT tile is STAGED once (coalesced reads of the
// mask along the keys, masks folded in as -inf), so the inner loops see one LDS word per score.  head_dim in {16, 32, 64, 128}
// (narrower heads zero-padded by the caller).  A row whose every key is masked is NaN in the output and in the gradients, as
// softmax(-inf, ..., -inf) is in the reference.
#include "attn_common.hpp"
#include "rowmath.hpp"

namespace {

constexpr int TB = 64;                                   // threads per workgroup (one wave)
// A row (query or key) is owned by S lanes, each holding DP = D / S of its head_dim values in registers (S = 2 at head_dim 128: 2 x 64
// registers per lane for the row and its accumulator is what fits without scratch; dot products then take one cross-lane add).
template <int D> struct Geo {
  static constexpr int S = D > 64 ? 2 : 1;
  static constexpr int DP = D / S;
  static constexpr int RW = TB / S;                      // rows owned by a workgroup
  static constexpr int KT = D <= 64 ? 64 : 32;           // streamed rows per stage (2 x KT x D floats of LDS)
};
template <int S> __device__ __forceinline__ float row_sum(float v) { return S == 2 ? v + __shfl_xor(v, 1, 64) : v; }

struct DenseAdd {            // what is added to the scaled dot product; every pointer may be null
  const float* bias;         // float attn_mask
  const uint8_t* bmask;      // bool attn_mask (nonzero = masked)
  int64_t sb, sh, sq, sk;    // strides (elements) of the mask's [B, H, Lq, Lk] broadcast view
  const uint8_t* kpm;        // key_padding_mask [B, Lk] (nonzero = ignored)
  const float* posq;         // [B * Lq, 2]
  const float* posk;         // [B * Lk, 2]
  float inv_tau;
  __host__ __device__ bool any() const { return bias || bmask || kpm || posq; }
};

__device__ __forceinline__ float dense_keep(uint32_t seed, int b, int head, int q, int k, uint32_t thresh16, float keep) {
  const uint32_t w = hash32((uint32_t)q * 0x9E3779B1U ^ hash32((uint32_t)k * 0x85EBCA6BU ^ seed ^ ((uint32_t)head * 0xC2B2AE35U) ^
                                                               ((uint32_t)b * 0x27D4EB2FU)));
  return (w >> 16) >= thresh16 ? keep : 0.f;
}

template <int D>
__device__ __forceinline__ void load_row(float (&r)[D], const float* __restrict__ p, float mul) {
#pragma unroll
  for (int i = 0; i < D; i += 4) {
    const float4 t = *reinterpret_cast<const float4*>(p + i);
    r[i] = t.x * mul; r[i + 1] = t.y * mul; r[i + 2] = t.z * mul; r[i + 3] = t.w * mul;
  }
}

template <int D>
__device__ __forceinline__ float dot_lds(const float (&a)[D], const float* __restrict__ row) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < D; i += 4) {
    const float4 t = *reinterpret_cast<const float4*>(row + i);
    s = fmaf(a[i], t.x, s); s = fmaf(a[i + 1], t.y, s); s = fmaf(a[i + 2], t.z, s); s = fmaf(a[i + 3], t.w, s);
  }
  return s;
}

// rows [r0, r0 + R) of a [*, ld] matrix (columns c0 .. c0 + D) -> LDS tile[R][D]; rows >= r1 as zeros
template <int D, int R>
__device__ __forceinline__ void stage_rows(float* __restrict__ tile, const float* __restrict__ src, int64_t ld, int c0, int64_t base, int r0, int r1) {
  for (int i = threadIdx.x; i < R * (D / 4); i += TB) {
    const int r = i / (D / 4), c = i % (D / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < r1) v = *reinterpret_cast<const float4*>(src + (base + r0 + r) * ld + c0 + 4 * c);
    *reinterpret_cast<float4*>(tile + r * D + 4 * c) = v;
  }
}

// the additive term of query rows [q0, q0 + QR) x key rows [k0, k0 + KR) -> A[q - q0][k - k0] (row pitch KR + 1); lane = key (or query
// when the tile is narrower along the keys than a wave), so the mask is read along its contiguous index.  Entries outside
// [0, Lq) x [0, Lk) are left alone: the callers never read them.
template <int QR, int KR>
__device__ __forceinline__ void stage_add(float* __restrict__ A, const DenseAdd& m, int b, int head, int q0, int k0, int Lq, int Lk) {
  for (int i = threadIdx.x; i < QR * KR; i += TB) {
    const int r = i / KR, c = i % KR;
    const int q = q0 + r, k = k0 + c;
    if (q >= Lq || k >= Lk) continue;
    float a = 0.f;
    const int64_t off = (int64_t)b * m.sb + (int64_t)head * m.sh + (int64_t)q * m.sq + (int64_t)k * m.sk;
    if (m.bias) a = m.bias[off];
    if (m.posq) {
      const float dx = m.posq[2 * ((int64_t)b * Lq + q)] - m.posk[2 * ((int64_t)b * Lk + k)];
      const float dy = m.posq[2 * ((int64_t)b * Lq + q) + 1] - m.posk[2 * ((int64_t)b * Lk + k) + 1];
      a -= sqrtf(fmaf(dx, dx, dy * dy)) * m.inv_tau;
    }
    if (m.bmask && m.bmask[off]) a = -INFINITY;
    if (m.kpm && m.kpm[(int64_t)b * Lk + k]) a = -INFINITY;
    A[r * (KR + 1) + c] = a;
  }
}

// ---- forward: thread = query row ------------------------------------------------------------------------------------------------
template <int D, bool ADD>
__global__ __launch_bounds__(TB) void k_attn_dense_fwd(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ K,
                                                       const float* __restrict__ V, int64_t ldk, int Lq, int Lk, float scale, DenseAdd add,
                                                       float drop_p, DgdmSeed seed_in, float* __restrict__ O, int64_t ldo,
                                                       float* __restrict__ lse) {
  constexpr int KT = Geo<D>::KT, S = Geo<D>::S, DP = Geo<D>::DP, RW = Geo<D>::RW;
  __shared__ __attribute__((aligned(16))) float Ks[KT * D];
  __shared__ __attribute__((aligned(16))) float Vs[KT * D];
  __shared__ float As[ADD ? RW * (KT + 1) : 1];
  const int b = blockIdx.z, head = blockIdx.y, H = gridDim.y;
  const int r = threadIdx.x / S, part = threadIdx.x % S;
  const int q0 = blockIdx.x * RW, ql = q0 + r;
  const bool ok = ql < Lq;
  const int64_t qrow = (int64_t)b * Lq + (ok ? ql : Lq - 1);
  const uint32_t seed = seed_in.value();
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  float q[DP], o[DP];
  load_row<DP>(q, Q + qrow * ldq + head * D + part * DP, scale);
#pragma unroll
  for (int i = 0; i < DP; ++i) o[i] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < Lk; k0 += KT) {
    __syncthreads();
    stage_rows<D, KT>(Ks, K, ldk, head * D, (int64_t)b * Lk, k0, Lk);
    stage_rows<D, KT>(Vs, V, ldk, head * D, (int64_t)b * Lk, k0, Lk);
    if (ADD) stage_add<RW, KT>(As, add, b, head, q0, k0, Lq, Lk);
    __syncthreads();
    const int nk = min(KT, Lk - k0);
    for (int jb = 0; jb < nk; jb += 4) {
      float s[4];
      float mx = m;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        __builtin_amdgcn_sched_barrier(0);      // one key's row in flight at a time
        const int j = jb + u, jc = j < nk ? j : 0;
        float v = row_sum<S>(dot_lds<DP>(q, Ks + jc * D + part * DP));
        if (ADD) v += As[r * (KT + 1) + jc];
        s[u] = (j < nk && ok) ? v : -INFINITY;
        mx = fmaxf(mx, s[u]);
      }
      const float alpha = mx == -INFINITY ? 1.0f : __expf(m - mx);       // nothing unmasked yet: keep the zeros
      m = mx;
      l *= alpha;
#pragma unroll
      for (int i = 0; i < DP; ++i) o[i] *= alpha;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        const int j = jb + u, jc = j < nk ? j : 0;
        float p = s[u] == -INFINITY ? 0.f : __expf(s[u] - m);
        l += p;
        if (drop_p > 0.f) p *= dense_keep(seed, b, head, ql, k0 + j, thresh, keep);
        const float* vr = Vs + jc * D + part * DP;
#pragma unroll
        for (int i = 0; i < DP; i += 4) {
          const float4 t = *reinterpret_cast<const float4*>(vr + i);
          o[i] = fmaf(p, t.x, o[i]); o[i + 1] = fmaf(p, t.y, o[i + 1]); o[i + 2] = fmaf(p, t.z, o[i + 2]); o[i + 3] = fmaf(p, t.w, o[i + 3]);
        }
      }
    }
  }
  if (ok) {
    const float inv = l > 0.f ? 1.0f / l : __int_as_float(0x7fc00000);      // every key masked: NaN, as softmax of all -inf is
    float* orow = O + ((int64_t)b * Lq + ql) * ldo + head * D + part * DP;
#pragma unroll
    for (int i = 0; i < DP; i += 4) *reinterpret_cast<float4*>(orow + i) = make_float4(o[i] * inv, o[i + 1] * inv, o[i + 2] * inv, o[i + 3] * inv);
    if (part == 0) lse[((int64_t)b * H + head) * Lq + ql] = l > 0.f ? m + __logf(l) : __int_as_float(0x7fc00000);
  }
}

// ---- backward, dQ: thread = query row; also writes delta[b][h][q] = sum_d dO O for the dK / dV passes ---------------------------
template <int D, bool ADD>
__global__ __launch_bounds__(TB) void k_attn_dense_bwd_dq(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ K,
                                                          const float* __restrict__ V, int64_t ldk, int Lq, int Lk, float scale, DenseAdd add,
                                                          float drop_p, DgdmSeed seed_in, const float* __restrict__ O,
                                                          const float* __restrict__ dO, int64_t ldo, const float* __restrict__ lse,
                                                          float* __restrict__ delta, float* __restrict__ dQ, int64_t ldgq) {
  constexpr int KT = Geo<D>::KT, S = Geo<D>::S, DP = Geo<D>::DP, RW = Geo<D>::RW;
  __shared__ __attribute__((aligned(16))) float Ks[KT * D];
  __shared__ __attribute__((aligned(16))) float Vs[KT * D];
  __shared__ float As[ADD ? RW * (KT + 1) : 1];
  const int b = blockIdx.z, head = blockIdx.y, H = gridDim.y;
  const int r = threadIdx.x / S, part = threadIdx.x % S;
  const int q0 = blockIdx.x * RW, ql = q0 + r;
  const bool ok = ql < Lq;
  const int64_t qrow = (int64_t)b * Lq + (ok ? ql : Lq - 1);
  const uint32_t seed = seed_in.value();
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  float q[DP], go[DP], dq[DP];
  load_row<DP>(q, Q + qrow * ldq + head * D + part * DP, scale);
  load_row<DP>(go, dO + qrow * ldo + head * D + part * DP, 1.0f);
  float dl = 0.f;
  {
    const float* orow = O + qrow * ldo + head * D + part * DP;
#pragma unroll
    for (int i = 0; i < DP; ++i) { dl = fmaf(go[i], orow[i], dl); dq[i] = 0.f; }
  }
  dl = row_sum<S>(dl);
  const int64_t stat = ((int64_t)b * H + head) * Lq + (ok ? ql : Lq - 1);
  const float L = lse[stat];
  if (ok && part == 0) delta[stat] = dl;
  for (int k0 = 0; k0 < Lk; k0 += KT) {
    __syncthreads();
    stage_rows<D, KT>(Ks, K, ldk, head * D, (int64_t)b * Lk, k0, Lk);
    stage_rows<D, KT>(Vs, V, ldk, head * D, (int64_t)b * Lk, k0, Lk);
    if (ADD) stage_add<RW, KT>(As, add, b, head, q0, k0, Lq, Lk);
    __syncthreads();
    const int nk = min(KT, Lk - k0);
    {                                                 // (rows past Lq run on a clamped row: the lanes of a row pair stay together)
      for (int j = 0; j < nk; ++j) {
        float s = row_sum<S>(dot_lds<DP>(q, Ks + j * D + part * DP));
        if (ADD) s += As[(ok ? r : 0) * (KT + 1) + j];
        const float p = __expf(s - L);                // masked key: exp(-inf) = 0; a fully masked row has L = NaN
        float dp = row_sum<S>(dot_lds<DP>(go, Vs + j * D + part * DP));
        if (drop_p > 0.f) dp *= dense_keep(seed, b, head, ql, k0 + j, thresh, keep);
        const float ds = p * (dp - dl) * scale;
        const float* kr = Ks + j * D + part * DP;
#pragma unroll
        for (int i = 0; i < DP; i += 4) {
          const float4 t = *reinterpret_cast<const float4*>(kr + i);
          dq[i] = fmaf(ds, t.x, dq[i]); dq[i + 1] = fmaf(ds, t.y, dq[i + 1]); dq[i + 2] = fmaf(ds, t.z, dq[i + 2]); dq[i + 3] = fmaf(ds, t.w, dq[i + 3]);
        }
      }
    }
  }
  if (ok) {
    float* drow = dQ + ((int64_t)b * Lq + ql) * ldgq + head * D + part * DP;
#pragma unroll
    for (int i = 0; i < DP; i += 4) *reinterpret_cast<float4*>(drow + i) = make_float4(dq[i], dq[i + 1], dq[i + 2], dq[i + 3]);
  }
}

// ---- backward, dK (WHICH = 0) or dV (WHICH = 1): thread = key row, query tiles streamed through LDS ------------------------------
template <int D, int WHICH, bool ADD>
__global__ __launch_bounds__(TB) void k_attn_dense_bwd_kv(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ K,
                                                          const float* __restrict__ V, int64_t ldk, int Lq, int Lk, float scale, DenseAdd add,
                                                          float drop_p, DgdmSeed seed_in, const float* __restrict__ dO, int64_t ldo,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          float* __restrict__ dOut, int64_t ldgk) {
  constexpr int QT = Geo<D>::KT, S = Geo<D>::S, DP = Geo<D>::DP, RW = Geo<D>::RW;
  __shared__ __attribute__((aligned(16))) float Qs[QT * D];
  __shared__ __attribute__((aligned(16))) float Gs[QT * D];
  __shared__ float Ls[QT], Ds[QT];
  __shared__ float As[ADD ? QT * (RW + 1) : 1];
  const int b = blockIdx.z, head = blockIdx.y, H = gridDim.y;
  const int r = threadIdx.x / S, part = threadIdx.x % S;
  const int kb = blockIdx.x * RW, kl = kb + r;
  const bool ok = kl < Lk;
  const int64_t krow = (int64_t)b * Lk + (ok ? kl : Lk - 1);
  const uint32_t seed = seed_in.value();
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  float k[DP], v[DP], acc[DP];                                    // v: used by the dK pass only (dead code otherwise)
  load_row<DP>(k, K + krow * ldk + head * D + part * DP, scale);  // s = q . (scale k)
  if constexpr (WHICH == 0) load_row<DP>(v, V + krow * ldk + head * D + part * DP, 1.0f);
#pragma unroll
  for (int i = 0; i < DP; ++i) acc[i] = 0.f;
  for (int q0 = 0; q0 < Lq; q0 += QT) {
    __syncthreads();
    stage_rows<D, QT>(Qs, Q, ldq, head * D, (int64_t)b * Lq, q0, Lq);
    stage_rows<D, QT>(Gs, dO, ldo, head * D, (int64_t)b * Lq, q0, Lq);
    if ((int)threadIdx.x < QT && q0 + (int)threadIdx.x < Lq) {
      const int64_t st = ((int64_t)b * H + head) * Lq + q0 + threadIdx.x;
      Ls[threadIdx.x] = lse[st]; Ds[threadIdx.x] = delta[st];
    }
    if (ADD) stage_add<QT, RW>(As, add, b, head, q0, kb, Lq, Lk);
    __syncthreads();
    const int nq = min(QT, Lq - q0);
    {                                                 // (rows past Lk run on a clamped row: the lanes of a row pair stay together)
      for (int j = 0; j < nq; ++j) {
        float s = row_sum<S>(dot_lds<DP>(k, Qs + j * D + part * DP));
        if (ADD) s += As[j * (RW + 1) + (ok ? r : 0)];
        const float p = __expf(s - Ls[j]);
        const float mk = drop_p > 0.f ? dense_keep(seed, b, head, q0 + j, kl, thresh, keep) : 1.0f;
        float c;
        const float* row;
        if constexpr (WHICH == 1) { c = p * mk; row = Gs + j * D + part * DP; }                                        // dV += (m p) dO
        else { c = p * (mk * row_sum<S>(dot_lds<DP>(v, Gs + j * D + part * DP)) - Ds[j]) * scale; row = Qs + j * D + part * DP; }   // dK += ds scale q
#pragma unroll
        for (int i = 0; i < DP; i += 4) {
          const float4 t = *reinterpret_cast<const float4*>(row + i);
          acc[i] = fmaf(c, t.x, acc[i]); acc[i + 1] = fmaf(c, t.y, acc[i + 1]); acc[i + 2] = fmaf(c, t.z, acc[i + 2]); acc[i + 3] = fmaf(c, t.w, acc[i + 3]);
        }
      }
    }
  }
  if (ok) {
    float* drow = dOut + ((int64_t)b * Lk + kl) * ldgk + head * D + part * DP;
#pragma unroll
    for (int i = 0; i < DP; i += 4) *reinterpret_cast<float4*>(drow + i) = make_float4(acc[i], acc[i + 1], acc[i + 2], acc[i + 3]);
  }
}

// ---- attention weights AFTER dropout (attention.py:145-146,171-178): head mean W[b][q][k] (per_head = 0: head 0 writes, the others
// add -- a diagnostic output, Lq Lk H read-modify-writes) or W[b * H + h][q][k]; thread = query row ------------------------------------
template <int D, bool ADD>
__global__ __launch_bounds__(TB) void k_attn_dense_weights(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ K, int64_t ldk,
                                                           int Lq, int Lk, int H, float scale, DenseAdd add, float drop_p, DgdmSeed seed_in,
                                                           const float* __restrict__ lse, int per_head, float* __restrict__ W) {
  constexpr int KT = Geo<D>::KT, S = Geo<D>::S, DP = Geo<D>::DP, RW = Geo<D>::RW;
  __shared__ __attribute__((aligned(16))) float Ks[KT * D];
  __shared__ float As[ADD ? RW * (KT + 1) : 1];
  const int b = blockIdx.z;
  const int r = threadIdx.x / S, part = threadIdx.x % S;
  const int q0 = blockIdx.x * RW, ql = q0 + r;
  const bool ok = ql < Lq;
  const int64_t qrow = (int64_t)b * Lq + (ok ? ql : Lq - 1);
  const uint32_t seed = seed_in.value();
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  const float wmul = per_head ? 1.0f : 1.0f / (float)H;
  for (int k0 = 0; k0 < Lk; k0 += KT) {
    const int nk = min(KT, Lk - k0);
    for (int h = 0; h < H; ++h) {
      __syncthreads();
      stage_rows<D, KT>(Ks, K, ldk, h * D, (int64_t)b * Lk, k0, Lk);
      if (ADD) stage_add<RW, KT>(As, add, b, h, q0, k0, Lq, Lk);
      __syncthreads();
      float q[DP];
      load_row<DP>(q, Q + qrow * ldq + h * D + part * DP, scale);
      const float L = lse[((int64_t)b * H + h) * Lq + (ok ? ql : Lq - 1)];
      float* wrow = W + ((per_head ? (int64_t)b * H + h : (int64_t)b) * Lq + ql) * Lk;
      for (int j = 0; j < nk; ++j) {
        float s = row_sum<S>(dot_lds<DP>(q, Ks + j * D + part * DP));
        if (ADD) s += As[(ok ? r : 0) * (KT + 1) + j];
        float p = __expf(s - L) * wmul;
        if (drop_p > 0.f) p *= dense_keep(seed, b, h, ql, k0 + j, thresh, keep);
        if (ok && part == 0) wrow[k0 + j] = (per_head || h == 0) ? p : wrow[k0 + j] + p;
      }
    }
  }
}

bool dense_shape_ok(int B, int Lq, int Lk, int H, int D) {
  return B >= 0 && Lq >= 0 && Lk >= 0 && H > 0 && H <= 65535 && B <= 65535 && (D == 16 || D == 32 || D == 64 || D == 128);
}

DenseAdd make_add(const float* bias, const uint8_t* bmask, int64_t sb, int64_t sh, int64_t sq, int64_t sk, const uint8_t* kpm, const float* posq,
                  const float* posk, float inv_tau) {
  return DenseAdd{bias, bmask, sb, sh, sq, sk, kpm, posq, posk, inv_tau};
}

}  // namespace

#define DENSE_DISPATCH(D_, ADD_, GO)                                                                     \
  switch (D_) {                                                                                          \
    case 16: if (ADD_) { GO(16, true); } else { GO(16, false); } break;                                  \
    case 32: if (ADD_) { GO(32, true); } else { GO(32, false); } break;                                  \
    case 64: if (ADD_) { GO(64, true); } else { GO(64, false); } break;                                  \
    default: if (ADD_) { GO(128, true); } else { GO(128, false); } break;                                \
  }

#define DENSE_COMMON_CHECKS()                                                                                                              \
  if (B < 0 || Lq < 0 || Lk < 0 || H <= 0 || !(drop_p >= 0.f && drop_p < 1.f) || (bias && bmask) || (!posq != !posk)) return DGDM_ERR_INVALID_ARG; \
  if (!dense_shape_ok(B, Lq, Lk, H, D)) return (D == 16 || D == 32 || D == 64 || D == 128) ? DGDM_ERR_INVALID_ARG : DGDM_ERR_UNSUPPORTED;          \
  if (B == 0 || Lq == 0) return DGDM_OK;                                                                                                  \
  if (Lk == 0) return DGDM_ERR_INVALID_ARG

extern "C" int dgdm_attn_dense_fwd(const float* Q, int64_t ldq, const float* K, const float* V, int64_t ldk, int32_t B, int32_t Lq, int32_t Lk,
                                   int32_t H, int32_t D, float scale, const float* bias, const uint8_t* bmask, int64_t sb, int64_t sh, int64_t sq,
                                   int64_t sk, const uint8_t* kpm, const float* posq, const float* posk, float inv_tau, float drop_p,
                                   uint32_t seed, float* O, int64_t ldo, float* lse, void* stream_) {
  DENSE_COMMON_CHECKS();
  if (!Q || !K || !V || !O || !lse) return DGDM_ERR_INVALID_ARG;
  if ((ldq & 3) || (ldk & 3) || (ldo & 3) || ldq < (int64_t)H * D || ldk < (int64_t)H * D || ldo < (int64_t)H * D || !dgdm_aligned16(Q) ||
      !dgdm_aligned16(K) || !dgdm_aligned16(V) || !dgdm_aligned16(O))
    return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const DenseAdd add = make_add(bias, bmask, sb, sh, sq, sk, kpm, posq, posk, inv_tau);
  const int rw = D > 64 ? TB / 2 : TB;      // Geo<D>::RW
  const dim3 grid((Lq + rw - 1) / rw, H, B);
#define GO(DD, AA) hipLaunchKernelGGL((k_attn_dense_fwd<DD, AA>), grid, dim3(TB), 0, s, Q, ldq, K, V, ldk, Lq, Lk, scale, add, drop_p, \
                                      dgdm_seed_arg(seed), O, ldo, lse)
  DENSE_DISPATCH(D, add.any(), GO)
#undef GO
  return dgdm_launch_status();
}

extern "C" int dgdm_attn_dense_bwd(const float* Q, int64_t ldq, const float* K, const float* V, int64_t ldk, int32_t B, int32_t Lq, int32_t Lk,
                                   int32_t H, int32_t D, float scale, const float* bias, const uint8_t* bmask, int64_t sb, int64_t sh, int64_t sq,
                                   int64_t sk, const uint8_t* kpm, const float* posq, const float* posk, float inv_tau, float drop_p,
                                   uint32_t seed, const float* O, const float* dO, int64_t ldo, const float* lse, float* delta, float* dQ,
                                   int64_t ldgq, float* dK, float* dV, int64_t ldgk, void* stream_) {
  DENSE_COMMON_CHECKS();
  if (!Q || !K || !V || !O || !dO || !lse || !delta || !dQ || !dK || !dV) return DGDM_ERR_INVALID_ARG;
  if ((ldq & 3) || (ldk & 3) || (ldo & 3) || (ldgq & 3) || (ldgk & 3) || ldq < (int64_t)H * D || ldk < (int64_t)H * D || ldo < (int64_t)H * D ||
      ldgq < (int64_t)H * D || ldgk < (int64_t)H * D || !dgdm_aligned16(Q) || !dgdm_aligned16(K) || !dgdm_aligned16(V) || !dgdm_aligned16(O) ||
      !dgdm_aligned16(dO) || !dgdm_aligned16(dQ) || !dgdm_aligned16(dK) || !dgdm_aligned16(dV))
    return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const DenseAdd add = make_add(bias, bmask, sb, sh, sq, sk, kpm, posq, posk, inv_tau);
  const DgdmSeed sd = dgdm_seed_arg(seed);
  const int rw = D > 64 ? TB / 2 : TB;      // Geo<D>::RW
  const dim3 gq((Lq + rw - 1) / rw, H, B), gk((Lk + rw - 1) / rw, H, B);
#define GO(DD, AA)                                                                                                                           \
  hipLaunchKernelGGL((k_attn_dense_bwd_dq<DD, AA>), gq, dim3(TB), 0, s, Q, ldq, K, V, ldk, Lq, Lk, scale, add, drop_p, sd, O, dO, ldo, lse, delta, \
                     dQ, ldgq);                                                                                                              \
  hipLaunchKernelGGL((k_attn_dense_bwd_kv<DD, 0, AA>), gk, dim3(TB), 0, s, Q, ldq, K, V, ldk, Lq, Lk, scale, add, drop_p, sd, dO, ldo, lse, delta, \
                     dK, ldgk);                                                                                                              \
  hipLaunchKernelGGL((k_attn_dense_bwd_kv<DD, 1, AA>), gk, dim3(TB), 0, s, Q, ldq, K, V, ldk, Lq, Lk, scale, add, drop_p, sd, dO, ldo, lse, delta, \
                     dV, ldgk)
  DENSE_DISPATCH(D, add.any(), GO)
#undef GO
  return dgdm_launch_status();
}

extern "C" int dgdm_attn_dense_weights(const float* Q, int64_t ldq, const float* K, int64_t ldk, int32_t B, int32_t Lq, int32_t Lk, int32_t H,
                                       int32_t D, float scale, const float* bias, const uint8_t* bmask, int64_t sb, int64_t sh, int64_t sq,
                                       int64_t sk, const uint8_t* kpm, const float* posq, const float* posk, float inv_tau, float drop_p,
                                       uint32_t seed, const float* lse, int32_t per_head, float* W, void* stream_) {
  DENSE_COMMON_CHECKS();
  if (!Q || !K || !lse || !W) return DGDM_ERR_INVALID_ARG;
  if ((ldq & 3) || (ldk & 3) || ldq < (int64_t)H * D || ldk < (int64_t)H * D || !dgdm_aligned16(Q) || !dgdm_aligned16(K)) return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const DenseAdd add = make_add(bias, bmask, sb, sh, sq, sk, kpm, posq, posk, inv_tau);
  const int rw = D > 64 ? TB / 2 : TB;      // Geo<D>::RW
  const dim3 grid((Lq + rw - 1) / rw, 1, B);
#define GO(DD, AA) hipLaunchKernelGGL((k_attn_dense_weights<DD, AA>), grid, dim3(TB), 0, s, Q, ldq, K, ldk, Lq, Lk, H, scale, add, drop_p, \
                                      dgdm_seed_arg(seed), lse, per_head, W)
  DENSE_DISPATCH(D, add.any(), GO)
#undef GO
  return dgdm_launch_status();
}
