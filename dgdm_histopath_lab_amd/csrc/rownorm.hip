// K6/K7 fused row normalisation:  y = dropout( act( norm_G(x [+ res]) * gamma + beta ) )
// One kernel family covers every norm site of the hot path:
//   LayerNorm (+GELU+dropout)         models/encoders.py:73-83,267-269   (G = 1)
//   LayerNorm(out + residual)         core/graph_layers.py:245, core/attention.py:325  (G = 1, res)
//   GroupNorm(8) + SiLU + dropout     core/diffusion.py:96-102 on 2-D [N, C] rows (G = 8)
// A row of C channels is split into G groups of L = C/G channels; every (row, group) pair is a
// "pseudo-row" of L contiguous floats with its own mean / rstd and affine parameters at channel
// offset (pseudo_row % G) * L.  HBM-bound: 16 B per lane, one pass over x in the forward (values
// stay in registers between the statistics and the output), x re-read once in the backward.
// Backward writes per-lane-group partial dgamma/dbeta slices and a second kernel sums them in a
// fixed order: no float atomics, bitwise reproducible.
// Dropout is a counter-based hash of (seed, element index): the mask is recomputed in backward.
#include <numeric>
#include <type_traits>

#include "common.hpp"
#include "rowmath.hpp"
#include "colsum.hpp"

namespace {

template <int LPR>
__device__ __forceinline__ float seg_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

#define F4_ZERO make_float4(0.f, 0.f, 0.f, 0.f)

template <int LPR, int R, int ACT>
__global__ __launch_bounds__(256) void k_rownorm_fwd(const float* __restrict__ x, const float* __restrict__ res,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int64_t rows, int L, int G, float eps, float drop_p, DgdmSeed seed_in,
                                                     float* __restrict__ y, float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                     unsigned* __restrict__ amax) {
  const uint32_t seed = seed_in.value();
  unsigned am = 0;
  constexpr int GPW = 64 / LPR;  // pseudo-rows per wave
  const int lane = threadIdx.x & 63, sub = lane / LPR, lir = lane % LPR;
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x >> 6) * GPW;
  const int64_t gid = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * GPW + sub;
  const int l4 = L >> 2;
  const float invL = 1.0f / (float)L;
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  // every lane group must run the same number of iterations (shuffles): iterate to the common bound
  const int64_t iters = (rows + ngroups - 1) / ngroups;
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t row = gid + it * ngroups;
    const bool ok = row < rows;
    const float4* xr = reinterpret_cast<const float4*>(x + row * L);
    const float4* rr = res ? reinterpret_cast<const float4*>(res + row * L) : nullptr;
    float4 v[R];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int k = lir + r * LPR;
      v[r] = F4_ZERO;
      if (ok && k < l4) {
        v[r] = xr[k];
        if (rr) { const float4 t = rr[k]; v[r].x += t.x; v[r].y += t.y; v[r].z += t.z; v[r].w += t.w; }
      }
      s += (v[r].x + v[r].y) + (v[r].z + v[r].w);
    }
    const float mu = seg_sum<LPR>(s) * invL;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int k = lir + r * LPR;
      if (k < l4) {
        const float a = v[r].x - mu, b = v[r].y - mu, c = v[r].z - mu, d = v[r].w - mu;
        q += (a * a + b * b) + (c * c + d * d);
      }
    }
    const float rs = 1.0f / sqrtf(seg_sum<LPR>(q) * invL + eps);
    if (!ok) continue;
    const int coff = (int)(row % G) * L;
    float4* yr = reinterpret_cast<float4*>(y + row * L);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int k = lir + r * LPR;
      if (k < l4) {
        const float4 g4 = reinterpret_cast<const float4*>(gamma + coff)[k];
        const float4 b4 = reinterpret_cast<const float4*>(beta + coff)[k];
        float4 o;
        o.x = act_f<ACT>((v[r].x - mu) * rs * g4.x + b4.x);
        o.y = act_f<ACT>((v[r].y - mu) * rs * g4.y + b4.y);
        o.z = act_f<ACT>((v[r].z - mu) * rs * g4.z + b4.z);
        o.w = act_f<ACT>((v[r].w - mu) * rs * g4.w + b4.w);
        if (drop_p > 0.f) {
          const float4 m = dropout_scale4(seed, (uint64_t)row * L + 4 * k, thresh, keep_scale);
          o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
        }
        yr[k] = o;
        if (amax) am = dgdm_amax4(am, o);
      }
    }
    if (lir == 0) { mean_o[row] = mu; rstd_o[row] = rs; }
  }
  if (amax) dgdm_amax_commit(am, amax);
}

template <int LPR, int R, int ACT>
__global__ __launch_bounds__(256) void k_rownorm_bwd(const float* __restrict__ x, const float* __restrict__ res,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                     const float* __restrict__ dy, int64_t rows, int L, int G, float drop_p,
                                                     DgdmSeed seed_in, float* __restrict__ dx, float* __restrict__ partial, int C,
                                                     unsigned* __restrict__ tickets, int ntickets, int block_slots,
                                                     unsigned* __restrict__ amax) {
  const uint32_t seed = seed_in.value();
  unsigned am = 0;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntickets) tickets[threadIdx.x] = 0u;   // for the column-sum kernel that follows
  constexpr int GPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane / LPR, lir = lane % LPR;
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x >> 6) * GPW;  // multiple of G (host guarantees)
  const int64_t gid = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * GPW + sub;
  const int l4 = L >> 2;
  const float invL = 1.0f / (float)L;
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  const int coff = (int)(gid % G) * L;  // fixed for this lane group: rows gid, gid+ngroups, ... share row % G
  float4 g4[R], b4[R], dg[R], db[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int k = lir + r * LPR;
    g4[r] = k < l4 ? reinterpret_cast<const float4*>(gamma + coff)[k] : F4_ZERO;
    b4[r] = k < l4 ? reinterpret_cast<const float4*>(beta + coff)[k] : F4_ZERO;
    dg[r] = F4_ZERO; db[r] = F4_ZERO;
  }
  const int64_t iters = (rows + ngroups - 1) / ngroups;
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t row = gid + it * ngroups;
    const bool ok = row < rows;
    const float mu = ok ? mean_i[row] : 0.f, rs = ok ? rstd_i[row] : 0.f;
    float4 xh[R], gz[R];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int k = lir + r * LPR;
      xh[r] = F4_ZERO; gz[r] = F4_ZERO;
      if (ok && k < l4) {
        float4 v = reinterpret_cast<const float4*>(x + row * L)[k];
        if (res) { const float4 t = reinterpret_cast<const float4*>(res + row * L)[k]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        float4 g = reinterpret_cast<const float4*>(dy + row * L)[k];
        if (drop_p > 0.f) {
          const float4 m = dropout_scale4(seed, (uint64_t)row * L + 4 * k, thresh, keep_scale);
          g.x *= m.x; g.y *= m.y; g.z *= m.z; g.w *= m.w;
        }
        xh[r] = make_float4((v.x - mu) * rs, (v.y - mu) * rs, (v.z - mu) * rs, (v.w - mu) * rs);
        g.x *= act_df<ACT>(xh[r].x * g4[r].x + b4[r].x);
        g.y *= act_df<ACT>(xh[r].y * g4[r].y + b4[r].y);
        g.z *= act_df<ACT>(xh[r].z * g4[r].z + b4[r].z);
        g.w *= act_df<ACT>(xh[r].w * g4[r].w + b4[r].w);
        db[r].x += g.x; db[r].y += g.y; db[r].z += g.z; db[r].w += g.w;
        dg[r].x += g.x * xh[r].x; dg[r].y += g.y * xh[r].y; dg[r].z += g.z * xh[r].z; dg[r].w += g.w * xh[r].w;
        gz[r] = make_float4(g.x * g4[r].x, g.y * g4[r].y, g.z * g4[r].z, g.w * g4[r].w);  // d/d xhat
        s1 += (gz[r].x + gz[r].y) + (gz[r].z + gz[r].w);
        s2 += (gz[r].x * xh[r].x + gz[r].y * xh[r].y) + (gz[r].z * xh[r].z + gz[r].w * xh[r].w);
      }
    }
    const float m1 = seg_sum<LPR>(s1) * invL, m2 = seg_sum<LPR>(s2) * invL;
    if (!ok) continue;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int k = lir + r * LPR;
      if (k < l4) {
        float4 o;
        o.x = rs * (gz[r].x - m1 - xh[r].x * m2);
        o.y = rs * (gz[r].y - m1 - xh[r].y * m2);
        o.z = rs * (gz[r].z - m1 - xh[r].z * m2);
        o.w = rs * (gz[r].w - m1 - xh[r].w * m2);
        reinterpret_cast<float4*>(dx + row * L)[k] = o;
        if (amax) am = dgdm_amax4(am, o);
      }
    }
  }
  if (amax) dgdm_amax_commit(am, amax);
  // partial[slot][0..C) = dgamma slice, partial[slot][C..2C) = dbeta slice.
  // block_slots > 0 (the block's lane groups cover block_slots whole rows of 2C, i.e. groups-per-block % G == 0): the rows
  // are first added in LDS in index order and the block writes ONE row, slot = blockIdx.x -- 4x (LayerNorm) fewer partial
  // rows for the column-sum kernel to read.  Otherwise every lane group writes its own slice, slot = gid / G.
  extern __shared__ float red[];
  const bool in_block = block_slots > 0;
  const int gl = (int)(gid - (int64_t)blockIdx.x * (blockDim.x >> 6) * GPW);   // lane group inside the block
  float* pg = in_block ? red + (gl / G) * (2 * C) + coff : partial + (gid / G) * (int64_t)(2 * C) + coff;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int k = lir + r * LPR;
    if (k < l4) {
      reinterpret_cast<float4*>(pg)[k] = dg[r];
      reinterpret_cast<float4*>(pg + C)[k] = db[r];
    }
  }
  if (in_block) {
    __syncthreads();
    float* prow = partial + (int64_t)blockIdx.x * (2 * C);
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) {
      float t = red[c];
      for (int sl = 1; sl < block_slots; ++sl) t += red[sl * 2 * C + c];
      prow[c] = t;
    }
  }
}

constexpr int REDUCE_CHUNKS = 32;

struct Geo { int lpr, r; };
bool geometry(int L, Geo* g) {
  if (L <= 0 || (L & 3)) return false;
  const int l4 = L >> 2;
  if (l4 > 256) return false;
  if (l4 > 64) { g->lpr = 64; g->r = (l4 + 63) / 64; return true; }
  int p = 1;
  while (p < l4) p <<= 1;
  g->lpr = p; g->r = 1;
  return true;
}

int64_t bwd_lane_groups(int64_t rows, int G, const Geo& geo) {
  // lane groups in the backward grid: enough to fill the chip, a multiple of G and of the
  // groups-per-block, capped so the partial buffer stays small
  const int gpw = 64 / geo.lpr, gpb = gpw * 4;
  int64_t want = 4096 > 1024 * gpb ? 4096 : 1024 * gpb;   // >= 1024 workgroups also when a block packs 16 or 32 narrow rows
  if (want > rows) want = rows;
  int64_t unit = (int64_t)gpb * G / std::gcd((int64_t)gpb, (int64_t)G);
  int64_t n = (want + unit - 1) / unit * unit;
  return n < unit ? unit : n;
}

// rows of the partial buffer: one per workgroup when its lane groups cover whole rows of 2C (reduced in LDS), else one per G groups
int bwd_block_slots(int G, const Geo& geo) {
  const int gpb = (64 / geo.lpr) * 4;
  return gpb % G == 0 ? gpb / G : 0;
}
int64_t bwd_slots(int64_t ng, int G, const Geo& geo) {
  const int gpb = (64 / geo.lpr) * 4;
  return bwd_block_slots(G, geo) > 0 ? ng / gpb : ng / G;
}

}  // namespace

#define ROWNORM_DISPATCH(KERNEL, ...)                                                      \
  do {                                                                                     \
    bool done = false;                                                                     \
    auto go = [&](auto lpr_c, auto r_c) {                                                  \
      constexpr int LPR_ = decltype(lpr_c)::value, R_ = decltype(r_c)::value;              \
      if (geo.lpr == LPR_ && geo.r == R_ && !done) {                                       \
        done = true;                                                                       \
        switch (act) {                                                                     \
          case DGDM_ACT_GELU: KERNEL(LPR_, R_, DGDM_ACT_GELU, __VA_ARGS__); break;         \
          case DGDM_ACT_RELU: KERNEL(LPR_, R_, DGDM_ACT_RELU, __VA_ARGS__); break;         \
          case DGDM_ACT_SILU: KERNEL(LPR_, R_, DGDM_ACT_SILU, __VA_ARGS__); break;         \
          case DGDM_ACT_ELU: KERNEL(LPR_, R_, DGDM_ACT_ELU, __VA_ARGS__); break;           \
          default: KERNEL(LPR_, R_, DGDM_ACT_NONE, __VA_ARGS__); break;                    \
        }                                                                                  \
      }                                                                                    \
    };                                                                                     \
    using std::integral_constant;                                                          \
    go(integral_constant<int, 1>{}, integral_constant<int, 1>{});                          \
    go(integral_constant<int, 2>{}, integral_constant<int, 1>{});                          \
    go(integral_constant<int, 4>{}, integral_constant<int, 1>{});                          \
    go(integral_constant<int, 8>{}, integral_constant<int, 1>{});                          \
    go(integral_constant<int, 16>{}, integral_constant<int, 1>{});                         \
    go(integral_constant<int, 32>{}, integral_constant<int, 1>{});                         \
    go(integral_constant<int, 64>{}, integral_constant<int, 1>{});                         \
    go(integral_constant<int, 64>{}, integral_constant<int, 2>{});                         \
    go(integral_constant<int, 64>{}, integral_constant<int, 3>{});                         \
    go(integral_constant<int, 64>{}, integral_constant<int, 4>{});                         \
  } while (0)

static int check_common(const float* x, const float* gamma, const float* beta, int32_t N, int32_t C, int32_t G, int32_t act,
                        float drop_p) {
  if (N < 0 || C <= 0 || G <= 0 || act < 0 || act > DGDM_ACT_ELU || !(drop_p >= 0.f && drop_p < 1.f)) return DGDM_ERR_INVALID_ARG;
  if (N > 0 && (!x || !gamma || !beta)) return DGDM_ERR_INVALID_ARG;
  if (C % G) return DGDM_ERR_UNSUPPORTED;
  return DGDM_OK;
}

extern "C" int dgdm_rownorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, int32_t N, int32_t C,
                                int32_t G, float eps, int32_t act, float drop_p, uint32_t seed, float* y, float* mean,
                                float* rstd, uint32_t* amax, void* stream_) {
  int rc = check_common(x, gamma, beta, N, C, G, act, drop_p);
  if (rc != DGDM_OK) return rc;
  if (N == 0) return DGDM_OK;
  DGDM_REQUIRE(y && mean && rstd);
  Geo geo;
  if (!geometry(C / G, &geo)) return DGDM_ERR_UNSUPPORTED;
  if (!dgdm_aligned16(x) || !dgdm_aligned16(y) || (res && !dgdm_aligned16(res)) || !dgdm_aligned16(gamma) || !dgdm_aligned16(beta))
    return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const int64_t rows = (int64_t)N * G;
  const int L = C / G;
  const int gpb = (64 / geo.lpr) * 4;
  int64_t blocks = (rows + gpb - 1) / gpb;
  if (blocks > 16384) blocks = 16384;
#define FWD(LPR_, R_, ACT_, ...) \
  hipLaunchKernelGGL((k_rownorm_fwd<LPR_, R_, ACT_>), dim3((unsigned)blocks), dim3(256), 0, s, __VA_ARGS__)
  ROWNORM_DISPATCH(FWD, x, res, gamma, beta, rows, L, G, eps, drop_p, dgdm_seed_arg(seed), y, mean, rstd, amax);
#undef FWD
  return dgdm_launch_status();
}

extern "C" size_t dgdm_rownorm_bwd_workspace_bytes(int32_t N, int32_t C, int32_t G) {
  Geo geo;
  if (N <= 0 || C <= 0 || G <= 0 || C % G || !geometry(C / G, &geo)) return 0;
  const int64_t ng = bwd_lane_groups((int64_t)N * G, G, geo);
  return (size_t)(bwd_slots(ng, G, geo) + REDUCE_CHUNKS + 1) * 2 * C * sizeof(float);  // partials + stage-1 sums + ticket row
}

extern "C" int64_t dgdm_rownorm_bwd_slots(int32_t N, int32_t C, int32_t G) {
  Geo geo;
  if (N <= 0 || C <= 0 || G <= 0 || C % G || !geometry(C / G, &geo)) return 0;
  return bwd_slots(bwd_lane_groups((int64_t)N * G, G, geo), G, geo);
}

extern "C" int dgdm_rownorm_bwd(const float* x, const float* res, const float* gamma, const float* beta, const float* mean,
                                const float* rstd, const float* dy, int32_t N, int32_t C, int32_t G, int32_t act, float drop_p,
                                uint32_t seed, float* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                uint32_t* amax, void* stream_) {
  int rc = check_common(x, gamma, beta, N, C, G, act, drop_p);
  if (rc != DGDM_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  if (N == 0) {
    if (dgamma) dgdm_fill_async(dgamma, 0, sizeof(float) * C, s);
    if (dbeta) dgdm_fill_async(dbeta, 0, sizeof(float) * C, s);
    return dgdm_launch_status();
  }
  DGDM_REQUIRE(mean && rstd && dy && dx && workspace);
  if (!dgamma != !dbeta) return DGDM_ERR_INVALID_ARG;      // both, or neither (the caller reduces the partials: dgdm_rownorm_bwd_slots)
  Geo geo;
  if (!geometry(C / G, &geo)) return DGDM_ERR_UNSUPPORTED;
  if (!dgdm_aligned16(x) || !dgdm_aligned16(dy) || !dgdm_aligned16(dx) || (res && !dgdm_aligned16(res)) ||
      !dgdm_aligned16(gamma) || !dgdm_aligned16(beta) || !dgdm_aligned16(workspace))
    return DGDM_ERR_UNSUPPORTED;
  const int64_t rows = (int64_t)N * G;
  const int L = C / G;
  const int64_t ng = bwd_lane_groups(rows, G, geo);
  const int64_t slots = bwd_slots(ng, G, geo);
  const int block_slots = bwd_block_slots(G, geo);
  const size_t lds = (size_t)block_slots * 2 * C * sizeof(float);
  if (lds > 64 * 1024) return DGDM_ERR_UNSUPPORTED;
  if (workspace_bytes < (size_t)(slots + REDUCE_CHUNKS + 1) * 2 * C * sizeof(float)) return DGDM_ERR_WORKSPACE;
  const int gpb = (64 / geo.lpr) * 4;
  const int64_t blocks = ng / gpb;
  float* partial = static_cast<float*>(workspace);
#define BWD(LPR_, R_, ACT_, ...) \
  hipLaunchKernelGGL((k_rownorm_bwd<LPR_, R_, ACT_>), dim3((unsigned)blocks), dim3(256), lds, s, __VA_ARGS__)
  // dgamma | dbeta = column sums of partial [slots][2C]: two fixed-order stages in one launch (tickets zeroed above)
  float* stage1 = partial + slots * 2 * C;
  unsigned* tickets = reinterpret_cast<unsigned*>(stage1 + (int64_t)REDUCE_CHUNKS * 2 * C);
  const int ntickets = (2 * C + 63) / 64;
  if (ntickets > 256) return DGDM_ERR_UNSUPPORTED;
  ROWNORM_DISPATCH(BWD, x, res, gamma, beta, mean, rstd, dy, rows, L, G, drop_p, dgdm_seed_arg(seed), dx, partial, C, tickets, ntickets, block_slots, amax);
#undef BWD
  if (!dgamma) return dgdm_launch_status();
  const int64_t chunk = (slots + REDUCE_CHUNKS - 1) / REDUCE_CHUNKS;
  const int nch = (int)((slots + chunk - 1) / chunk);
  hipLaunchKernelGGL(k_colsum_ticket<REDUCE_CHUNKS>, dim3(ntickets, nch), dim3(256), 0, s, partial, slots, 2 * C, chunk, stage1, tickets, dgamma, C, dbeta);
  return dgdm_launch_status();
}
