// K4-gen: spatial attention for head dims the MFMA kernels are not tiled for (16 < head_dim <= 64).
//   O[q] = sum_k dropout(softmax_k( Q[q].K[k] * scale - |pos_q - pos_k| * inv_tau )) V[k]     per graph, per head
// The reference accepts any embed_dim % num_heads == 0 (core/attention.py:36-40); DGDMModel's defaults give head_dim 16, the
// shape csrc/attn_h_*.hip / attn_*.hip are built around (one MFMA tile per head).  Valid configurations such as
// hidden_dims[-1] = 128 with attention_heads in {2, 4} (head_dim 64 / 32) run HERE: plain fp32 on the vector units, flash-style
// (no [N, N] tensor), one thread per query row (forward, dQ) or per key row (dK / dV), the streamed side of a 64-row tile staged
// through LDS and read as broadcasts.  Correct and self-contained rather than fast: ~N^2 * H * 4 D FMAs per pass on the VALU
// (a 10 000-node graph at H = 2, D = 64: ~0.1 TFLOP forward + backward, milliseconds) -- the default configurations never come here.
// Dropout on the attention weights (core/attention.py:154): one hash word per (head, query, key), the same function in every
// kernel of this file, row sums taken before the mask as in the MFMA kernels.
#include "attn_common.hpp"
#include "rowmath.hpp"

namespace {

constexpr int TB = 64;      // rows per tile = threads per workgroup (one wave)

__device__ __forceinline__ float gen_keep(uint32_t seed, int graph_row0, int head, int q, int k, uint32_t thresh16, float keep) {
  // (q, k): rows local to the graph; graph_row0 separates the graphs of a batch
  const uint32_t w = hash32((uint32_t)q * 0x9E3779B1U ^ hash32((uint32_t)k * 0x85EBCA6BU ^ seed ^ ((uint32_t)head * 0xC2B2AE35U) ^
                                                               ((uint32_t)graph_row0 * 0x27D4EB2FU)));
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

// stage rows [r0, r0 + TB) of a [*, ld] matrix (columns c0 .. c0 + D) into LDS tile[TB][D]; rows >= r1 as zeros
template <int D>
__device__ __forceinline__ void stage(float* __restrict__ tile, const float* __restrict__ src, int64_t ld, int c0, int r0, int r1) {
  for (int i = threadIdx.x; i < TB * (D / 4); i += TB) {
    const int r = i / (D / 4), c = i % (D / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < r1) v = *reinterpret_cast<const float4*>(src + (int64_t)(r0 + r) * ld + c0 + 4 * c);
    *reinterpret_cast<float4*>(tile + r * D + 4 * c) = v;
  }
}

// ---- forward: thread = query row ------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(TB) void k_attn_gen_fwd(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                     int64_t ld, const float* __restrict__ pos, const int32_t* __restrict__ ptr, int B,
                                                     float scale, float inv_tau, float drop_p, DgdmSeed seed_in, float* __restrict__ O,
                                                     int64_t ldo, float* __restrict__ lse, int N_tot) {
  __shared__ __attribute__((aligned(16))) float Ks[TB * D];
  __shared__ __attribute__((aligned(16))) float Vs[TB * D];
  __shared__ float Px[TB], Py[TB];
  int n0, n1, lt;
  if (!find_graph(ptr, B, TB, blockIdx.x, &n0, &n1, &lt)) return;
  const int head = blockIdx.y, ng = n1 - n0;
  const int ql = lt * TB + threadIdx.x;
  const bool ok = ql < ng;
  const int qrow = n0 + (ok ? ql : ng - 1);
  const uint32_t seed = seed_in.value();
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  float q[D], o[D];
  load_row<D>(q, Q + (int64_t)qrow * ld + head * D, scale);
#pragma unroll
  for (int i = 0; i < D; ++i) o[i] = 0.f;
  const float qx = pos[2 * (int64_t)qrow], qy = pos[2 * (int64_t)qrow + 1];
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < ng; k0 += TB) {
    __syncthreads();
    stage<D>(Ks, K, ld, head * D, n0 + k0, n1);
    stage<D>(Vs, V, ld, head * D, n0 + k0, n1);
    if (k0 + (int)threadIdx.x < ng) { Px[threadIdx.x] = pos[2 * (int64_t)(n0 + k0 + threadIdx.x)]; Py[threadIdx.x] = pos[2 * (int64_t)(n0 + k0 + threadIdx.x) + 1]; }
    __syncthreads();
    const int nk = min(TB, ng - k0);
    // four keys at a time (static register indices): scores, the running maximum, one rescale, then the weighted V rows
    for (int jb = 0; jb < nk; jb += 4) {
      float s[4];
      float mx = m;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        __builtin_amdgcn_sched_barrier(0);      // one key's row in flight at a time (the compiler otherwise hoists every LDS read)
        const int j = jb + u, jc = j < nk ? j : 0;
        const float dx = qx - Px[jc], dy = qy - Py[jc];
        const float v = dot_lds<D>(q, Ks + jc * D) - sqrtf(fmaf(dx, dx, dy * dy)) * inv_tau;
        s[u] = j < nk ? v : -INFINITY;
        mx = fmaxf(mx, s[u]);
      }
      const float alpha = __expf(m - mx);        // first block: exp(-inf) = 0
      m = mx;
      l *= alpha;
#pragma unroll
      for (int i = 0; i < D; ++i) o[i] *= alpha;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        const int j = jb + u, jc = j < nk ? j : 0;
        float p = __expf(s[u] - m);               // masked keys: exp(-inf) = 0
        l += p;
        if (drop_p > 0.f) p *= gen_keep(seed, n0, head, ql, k0 + j, thresh, keep);
        const float* vr = Vs + jc * D;
#pragma unroll
        for (int i = 0; i < D; i += 4) {
          const float4 t = *reinterpret_cast<const float4*>(vr + i);
          o[i] = fmaf(p, t.x, o[i]); o[i + 1] = fmaf(p, t.y, o[i + 1]); o[i + 2] = fmaf(p, t.z, o[i + 2]); o[i + 3] = fmaf(p, t.w, o[i + 3]);
        }
      }
    }
  }
  if (ok) {
    const float inv = 1.0f / l;
    float* orow = O + (int64_t)(n0 + ql) * ldo + head * D;
#pragma unroll
    for (int i = 0; i < D; i += 4) *reinterpret_cast<float4*>(orow + i) = make_float4(o[i] * inv, o[i + 1] * inv, o[i + 2] * inv, o[i + 3] * inv);
    lse[(int64_t)head * N_tot + n0 + ql] = m + __logf(l);
  }
}

// ---- backward, dQ: thread = query row; also writes delta[h][q] = sum_d dO O for the dK / dV passes ---------------------------
template <int D>
__global__ __launch_bounds__(TB) void k_attn_gen_bwd_dq(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                        int64_t ld, const float* __restrict__ pos, const int32_t* __restrict__ ptr, int B,
                                                        float scale, float inv_tau, float drop_p, DgdmSeed seed_in,
                                                        const float* __restrict__ O, const float* __restrict__ dO, int64_t ldo,
                                                        const float* __restrict__ lse, float* __restrict__ delta, float* __restrict__ dQ,
                                                        int64_t ldg, int N_tot) {
  __shared__ __attribute__((aligned(16))) float Ks[TB * D];
  __shared__ __attribute__((aligned(16))) float Vs[TB * D];
  __shared__ float Px[TB], Py[TB];
  int n0, n1, lt;
  if (!find_graph(ptr, B, TB, blockIdx.x, &n0, &n1, &lt)) return;
  const int head = blockIdx.y, ng = n1 - n0;
  const int ql = lt * TB + threadIdx.x;
  const bool ok = ql < ng;
  const int qrow = n0 + (ok ? ql : ng - 1);
  const uint32_t seed = seed_in.value();
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  float q[D], go[D], dq[D];
  load_row<D>(q, Q + (int64_t)qrow * ld + head * D, scale);
  load_row<D>(go, dO + (int64_t)qrow * ldo + head * D, 1.0f);
  float dl = 0.f;
  {
    const float* orow = O + (int64_t)qrow * ldo + head * D;
#pragma unroll
    for (int i = 0; i < D; ++i) { dl = fmaf(go[i], orow[i], dl); dq[i] = 0.f; }
  }
  const float L = lse[(int64_t)head * N_tot + qrow];
  if (ok) delta[(int64_t)head * N_tot + qrow] = dl;
  const float qx = pos[2 * (int64_t)qrow], qy = pos[2 * (int64_t)qrow + 1];
  for (int k0 = 0; k0 < ng; k0 += TB) {
    __syncthreads();
    stage<D>(Ks, K, ld, head * D, n0 + k0, n1);
    stage<D>(Vs, V, ld, head * D, n0 + k0, n1);
    if (k0 + (int)threadIdx.x < ng) { Px[threadIdx.x] = pos[2 * (int64_t)(n0 + k0 + threadIdx.x)]; Py[threadIdx.x] = pos[2 * (int64_t)(n0 + k0 + threadIdx.x) + 1]; }
    __syncthreads();
    const int nk = min(TB, ng - k0);
    for (int j = 0; j < nk; ++j) {
      const float dx = qx - Px[j], dy = qy - Py[j];
      const float s = dot_lds<D>(q, Ks + j * D) - sqrtf(fmaf(dx, dx, dy * dy)) * inv_tau;
      const float p = __expf(s - L);
      float dp = dot_lds<D>(go, Vs + j * D);
      if (drop_p > 0.f) dp *= gen_keep(seed, n0, head, ql, k0 + j, thresh, keep);
      const float ds = p * (dp - dl) * scale;
      const float* kr = Ks + j * D;
#pragma unroll
      for (int i = 0; i < D; i += 4) {
        const float4 t = *reinterpret_cast<const float4*>(kr + i);
        dq[i] = fmaf(ds, t.x, dq[i]); dq[i + 1] = fmaf(ds, t.y, dq[i + 1]); dq[i + 2] = fmaf(ds, t.z, dq[i + 2]); dq[i + 3] = fmaf(ds, t.w, dq[i + 3]);
      }
    }
  }
  if (ok) {
    float* drow = dQ + (int64_t)(n0 + ql) * ldg + head * D;
#pragma unroll
    for (int i = 0; i < D; i += 4) *reinterpret_cast<float4*>(drow + i) = make_float4(dq[i], dq[i + 1], dq[i + 2], dq[i + 3]);
  }
}

// ---- backward, dK (WHICH = 0) or dV (WHICH = 1): thread = key row, query tiles streamed through LDS ---------------------------
template <int D, int WHICH>
__global__ __launch_bounds__(TB) void k_attn_gen_bwd_kv(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                        int64_t ld, const float* __restrict__ pos, const int32_t* __restrict__ ptr, int B,
                                                        float scale, float inv_tau, float drop_p, DgdmSeed seed_in,
                                                        const float* __restrict__ dO, int64_t ldo, const float* __restrict__ lse,
                                                        const float* __restrict__ delta, float* __restrict__ dOut, int64_t ldg, int N_tot) {
  __shared__ __attribute__((aligned(16))) float Qs[TB * D];
  __shared__ __attribute__((aligned(16))) float Gs[TB * D];
  __shared__ float Px[TB], Py[TB], Ls[TB], Ds[TB];
  int n0, n1, lt;
  if (!find_graph(ptr, B, TB, blockIdx.x, &n0, &n1, &lt)) return;
  const int head = blockIdx.y, ng = n1 - n0;
  const int kl = lt * TB + threadIdx.x;
  const bool ok = kl < ng;
  const int krow = n0 + (ok ? kl : ng - 1);
  const uint32_t seed = seed_in.value();
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  float k[D], v[D], acc[D];                                       // v: used by the dK pass only (dead code otherwise)
  load_row<D>(k, K + (int64_t)krow * ld + head * D, scale);      // s = q . (scale k)
  if constexpr (WHICH == 0) load_row<D>(v, V + (int64_t)krow * ld + head * D, 1.0f);
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] = 0.f;
  const float kx = pos[2 * (int64_t)krow], ky = pos[2 * (int64_t)krow + 1];
  for (int q0 = 0; q0 < ng; q0 += TB) {
    __syncthreads();
    stage<D>(Qs, Q, ld, head * D, n0 + q0, n1);
    stage<D>(Gs, dO, ldo, head * D, n0 + q0, n1);
    if (q0 + (int)threadIdx.x < ng) {
      const int r = n0 + q0 + threadIdx.x;
      Px[threadIdx.x] = pos[2 * (int64_t)r]; Py[threadIdx.x] = pos[2 * (int64_t)r + 1];
      Ls[threadIdx.x] = lse[(int64_t)head * N_tot + r]; Ds[threadIdx.x] = delta[(int64_t)head * N_tot + r];
    }
    __syncthreads();
    const int nq = min(TB, ng - q0);
    for (int j = 0; j < nq; ++j) {
      const float dx = kx - Px[j], dy = ky - Py[j];
      const float s = dot_lds<D>(k, Qs + j * D) - sqrtf(fmaf(dx, dx, dy * dy)) * inv_tau;
      const float p = __expf(s - Ls[j]);
      const float mk = drop_p > 0.f ? gen_keep(seed, n0, head, q0 + j, kl, thresh, keep) : 1.0f;
      float c;
      const float* row;
      if constexpr (WHICH == 1) { c = p * mk; row = Gs + j * D; }                                  // dV += (m p) dO
      else { c = p * (mk * dot_lds<D>(v, Gs + j * D) - Ds[j]) * scale; row = Qs + j * D; }       // dK += ds scale q
#pragma unroll
      for (int i = 0; i < D; i += 4) {
        const float4 t = *reinterpret_cast<const float4*>(row + i);
        acc[i] = fmaf(c, t.x, acc[i]); acc[i + 1] = fmaf(c, t.y, acc[i + 1]); acc[i + 2] = fmaf(c, t.z, acc[i + 2]); acc[i + 3] = fmaf(c, t.w, acc[i + 3]);
      }
    }
  }
  if (ok) {
    float* drow = dOut + (int64_t)(n0 + kl) * ldg + head * D;
#pragma unroll
    for (int i = 0; i < D; i += 4) *reinterpret_cast<float4*>(drow + i) = make_float4(acc[i], acc[i + 1], acc[i + 2], acc[i + 3]);
  }
}

// ---- head-mean attention weights (eval mode), W[off[g] + i * n_g + j]: thread = query row -----------------------------------------
template <int D>
__global__ __launch_bounds__(TB) void k_attn_gen_weights(const float* __restrict__ Q, const float* __restrict__ K, int64_t ld,
                                                         const float* __restrict__ pos, const int32_t* __restrict__ ptr, int B, int H,
                                                         float scale, float inv_tau, const float* __restrict__ lse, float* __restrict__ W,
                                                         const int64_t* __restrict__ off, int N_tot) {
  __shared__ __attribute__((aligned(16))) float Ks[TB * D];
  __shared__ float Px[TB], Py[TB];
  int n0 = 0, n1 = 0, lt = 0, g = 0, tile = blockIdx.x;
  bool found = false;
  for (; g < B; ++g) {
    const int a = ptr[g], b = ptr[g + 1], nt = (b - a + TB - 1) / TB;
    if (tile < nt) { n0 = a; n1 = b; lt = tile; found = true; break; }
    tile -= nt;
  }
  if (!found) return;
  const int ng = n1 - n0;
  const int ql = lt * TB + threadIdx.x;
  const bool ok = ql < ng;
  const int qrow = n0 + (ok ? ql : ng - 1);
  const float qx = pos[2 * (int64_t)qrow], qy = pos[2 * (int64_t)qrow + 1];
  const float invH = 1.0f / (float)H;
  float* wrow = W + off[g] + (int64_t)ql * ng;      // this thread's row of the graph's [n_g, n_g] matrix: nobody else writes it
  for (int k0 = 0; k0 < ng; k0 += TB) {
    const int nk = min(TB, ng - k0);
    __syncthreads();
    if (k0 + (int)threadIdx.x < ng) { Px[threadIdx.x] = pos[2 * (int64_t)(n0 + k0 + threadIdx.x)]; Py[threadIdx.x] = pos[2 * (int64_t)(n0 + k0 + threadIdx.x) + 1]; }
    for (int h = 0; h < H; ++h) {
      __syncthreads();
      stage<D>(Ks, K, ld, h * D, n0 + k0, n1);
      __syncthreads();
      float q[D];
      load_row<D>(q, Q + (int64_t)qrow * ld + h * D, scale);
      const float L = lse[(int64_t)h * N_tot + qrow];
      if (ok)
        for (int j = 0; j < nk; ++j) {              // head 0 writes, the others add (a diagnostic output: N^2 H read-modify-writes)
          const float dx = qx - Px[j], dy = qy - Py[j];
          const float p = __expf(dot_lds<D>(q, Ks + j * D) - sqrtf(fmaf(dx, dx, dy * dy)) * inv_tau - L) * invH;
          wrow[k0 + j] = h == 0 ? p : wrow[k0 + j] + p;
        }
    }
  }
}

bool gen_args_ok(const void* Q, const void* K, const void* V, int64_t ld, const void* pos, const void* ptr, int B, int ntiles, int N_tot, int H,
                 int D) {
  return Q && K && V && pos && ptr && B >= 0 && ntiles >= 0 && N_tot >= 0 && H > 0 && (D == 32 || D == 64) && ld >= (int64_t)H * D;
}

}  // namespace

#define GEN_DISPATCH(D_, ...)      \
  if ((D_) == 32) { __VA_ARGS__(32); } else { __VA_ARGS__(64); }

extern "C" int dgdm_spatial_attn_gen_fwd(const float* Q, const float* K, const float* V, int64_t ld, const float* pos, const int32_t* ptr,
                                         int32_t B, int32_t num_tiles, int32_t N_tot, int32_t H, int32_t D, float scale, float inv_tau,
                                         float drop_p, uint32_t seed, float* O, int64_t ldo, float* lse, void* stream_) {
  if (B < 0 || num_tiles < 0 || N_tot < 0 || H <= 0 || !(drop_p >= 0.f && drop_p < 1.f)) return DGDM_ERR_INVALID_ARG;
  if (N_tot == 0 || num_tiles == 0) return DGDM_OK;
  if (!gen_args_ok(Q, K, V, ld, pos, ptr, B, num_tiles, N_tot, H, D) || !O || !lse) return (D == 32 || D == 64) ? DGDM_ERR_INVALID_ARG : DGDM_ERR_UNSUPPORTED;
  if ((ld & 3) || (ldo & 3) || ldo < (int64_t)H * D || !dgdm_aligned16(Q) || !dgdm_aligned16(K) || !dgdm_aligned16(V) || !dgdm_aligned16(O))
    return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
#define GO(DD) hipLaunchKernelGGL((k_attn_gen_fwd<DD>), dim3(num_tiles, H), dim3(TB), 0, s, Q, K, V, ld, pos, ptr, B, scale, inv_tau, drop_p, \
                                  dgdm_seed_arg(seed), O, ldo, lse, N_tot)
  GEN_DISPATCH(D, GO)
#undef GO
  return dgdm_launch_status();
}

extern "C" int dgdm_spatial_attn_gen_bwd(const float* Q, const float* K, const float* V, int64_t ld, const float* pos, const int32_t* ptr,
                                         int32_t B, int32_t num_tiles, int32_t N_tot, int32_t H, int32_t D, float scale, float inv_tau,
                                         float drop_p, uint32_t seed, const float* O, const float* dO, int64_t ldo, const float* lse,
                                         float* delta, float* dQ, float* dK, float* dV, int64_t ldg, void* stream_) {
  if (B < 0 || num_tiles < 0 || N_tot < 0 || H <= 0 || !(drop_p >= 0.f && drop_p < 1.f)) return DGDM_ERR_INVALID_ARG;
  if (N_tot == 0 || num_tiles == 0) return DGDM_OK;
  if (!gen_args_ok(Q, K, V, ld, pos, ptr, B, num_tiles, N_tot, H, D) || !O || !dO || !lse || !delta || !dQ || !dK || !dV)
    return (D == 32 || D == 64) ? DGDM_ERR_INVALID_ARG : DGDM_ERR_UNSUPPORTED;
  if ((ld & 3) || (ldo & 3) || (ldg & 3) || ldo < (int64_t)H * D || ldg < (int64_t)H * D || !dgdm_aligned16(Q) || !dgdm_aligned16(K) ||
      !dgdm_aligned16(V) || !dgdm_aligned16(O) || !dgdm_aligned16(dO) || !dgdm_aligned16(dQ) || !dgdm_aligned16(dK) || !dgdm_aligned16(dV))
    return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const DgdmSeed sd = dgdm_seed_arg(seed);
#define GO(DD)                                                                                                                          \
  hipLaunchKernelGGL((k_attn_gen_bwd_dq<DD>), dim3(num_tiles, H), dim3(TB), 0, s, Q, K, V, ld, pos, ptr, B, scale, inv_tau, drop_p, sd, O, \
                     dO, ldo, lse, delta, dQ, ldg, N_tot);                                                                              \
  hipLaunchKernelGGL((k_attn_gen_bwd_kv<DD, 0>), dim3(num_tiles, H), dim3(TB), 0, s, Q, K, V, ld, pos, ptr, B, scale, inv_tau, drop_p, sd, \
                     dO, ldo, lse, delta, dK, ldg, N_tot);                                                                              \
  hipLaunchKernelGGL((k_attn_gen_bwd_kv<DD, 1>), dim3(num_tiles, H), dim3(TB), 0, s, Q, K, V, ld, pos, ptr, B, scale, inv_tau, drop_p, sd, \
                     dO, ldo, lse, delta, dV, ldg, N_tot)
  GEN_DISPATCH(D, GO)
#undef GO
  return dgdm_launch_status();
}

extern "C" int dgdm_spatial_attn_gen_mean_weights(const float* Q, const float* K, int64_t ld, const float* pos, const int32_t* ptr, int32_t B,
                                                  int32_t num_tiles, int32_t N_tot, int32_t H, int32_t D, float scale, float inv_tau,
                                                  const float* lse, float* W, const int64_t* offsets, void* stream_) {
  if (B < 0 || num_tiles < 0 || N_tot < 0 || H <= 0) return DGDM_ERR_INVALID_ARG;
  if (N_tot == 0 || num_tiles == 0) return DGDM_OK;
  if (!Q || !K || !pos || !ptr || !lse || !W || !offsets) return DGDM_ERR_INVALID_ARG;
  if (D != 32 && D != 64) return DGDM_ERR_UNSUPPORTED;
  if ((ld & 3) || ld < (int64_t)H * D || !dgdm_aligned16(Q) || !dgdm_aligned16(K)) return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
#define GO(DD) hipLaunchKernelGGL((k_attn_gen_weights<DD>), dim3(num_tiles), dim3(TB), 0, s, Q, K, ld, pos, ptr, B, H, scale, inv_tau, lse, W, \
                                  offsets, N_tot)
  GEN_DISPATCH(D, GO)
#undef GO
  return dgdm_launch_status();
}
