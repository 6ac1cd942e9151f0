// K9: top-k node pooling and its inverse for the graph U-Net (reference: AdaptiveGraphPooling.forward
// core/graph_layers.py:285-329, unpooling core/graph_layers.py:441-448).  The reference runs it as a
// chain of framework ops (MLP, tanh, topk, mask.nonzero, fancy indexing, edge filter + relabel);
// here it is five small kernels with no host synchronisation and no data-dependent shapes:
//
//   dgdm_pool_score_fwd/bwd  s[i] = tanh(w2 . relu(h[i]) + b2)       (h = first score layer, a GEMM)
//   dgdm_topk_perm           exact top-k of s: perm (ascending node ids, as mask.nonzero() gives
//                            them) and node_map (new id or -1).  Radix select of the k-th largest
//                            key in four 8-bit passes, then an ordered stream compaction.  Ties
//                            at the threshold take the lowest node ids (torch.topk leaves the choice
//                            open; graph_layers.py:308-310).  Integer work only: bit-exact.
//   dgdm_pool_gather_fwd/bwd out[j] = x[perm[j]] * s[perm[j]] * mult; backward by node_map (a
//                            gather per source row: no scatter, no zero fill, no atomics)
//   dgdm_edge_relabel        edge (u, v) -> (node_map[u], node_map[v]) or (-1, -1) when either end
//                            was dropped (now or at an earlier level); E stays fixed, the CSR
//                            builder skips negative ids
//   dgdm_unpool_add_relu_fwd/bwd   out[i] = relu(skip[i] + (node_map[i] >= 0 ? x[node_map[i]] : 0))
#include "common.hpp"
#include "colsum.hpp"

namespace {

// ------------------------------------------------------------------ score
// 16 lanes per row, float4 per lane per 64-column chunk; 16 rows per 256-thread block.
__global__ __launch_bounds__(256) void k_pool_score_fwd(const float* __restrict__ h, int64_t ldh, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, int N, int C, float* __restrict__ s,
                                                        const uint8_t* __restrict__ decide, int nl) {
  const int sub = threadIdx.x & 15, row = blockIdx.x * 16 + (threadIdx.x >> 4);
  float acc = 0.f;
  if (row < N)
    for (int c = 4 * sub; c < C; c += 64) {
      const float4 v = *reinterpret_cast<const float4*>(h + (int64_t)row * ldh + c);
      const float4 w = *reinterpret_cast<const float4*>(w2 + c);
      if (decide) {   // kink decisions supplied by the caller (parity tests), see dgdm_hip.h
        const uchar4 d = *reinterpret_cast<const uchar4*>(decide + (int64_t)row * C + c);
        acc += (d.x ? v.x : 0.f) * w.x + (d.y ? v.y : 0.f) * w.y + (d.z ? v.z : 0.f) * w.z + (d.w ? v.w : 0.f) * w.w;
      } else {
        acc += fmaxf(v.x, 0.f) * w.x + fmaxf(v.y, 0.f) * w.y + fmaxf(v.z, 0.f) * w.z + fmaxf(v.w, 0.f) * w.w;
      }
    }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (sub == 0 && row < N) {
    const float z = acc + b2[0];     // nl: 0 tanh (DGDMModel's pools), 1 sigmoid, 2 none (the logit; softmax over all nodes follows)
    s[row] = nl == 0 ? tanhf(z) : nl == 1 ? 1.f / (1.f + __expf(-z)) : z;
  }
}

// dh[i] = t_i * w2 * [h > 0], t_i = ds_i (1 - s_i^2); partial[b] = [sum_i t_i relu(h[i]) (C) | sum_i t_i] over the block's rows.
// The sums are carried in float64 from the first add to the last (round 4): db2 = sum_i t_i is ONE sum over every node of the batch
// whose terms cancel (to 1e-4 of their magnitude at trained weights), so an fp32 accumulation puts its own rounding -- eps * sum|t_i|
// -- on top of whatever the terms carry.  With float64 partials the reduction adds nothing: the result is the sum of the fp32
// terms, correctly rounded once.  (N * C adds in float64: nothing beside the GEMM that produced h.)
__global__ __launch_bounds__(256) void k_pool_score_bwd(const float* __restrict__ h, int64_t ldh, const float* __restrict__ w2,
                                                        const float* __restrict__ s, const float* __restrict__ ds, int N, int C,
                                                        int rows_per_block, float* __restrict__ dh, int64_t lddh,
                                                        double* __restrict__ partial, const uint8_t* __restrict__ decide, int nl,
                                                        unsigned* __restrict__ amax) {
  // thread = (float4 column c4, row lane): C/4 <= 64 columns x (256 / cols) row lanes
  const int cols = C / 4;
  const int c4 = threadIdx.x % cols, rl = threadIdx.x / cols, nrl = 256 / cols;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
  double ax = 0., ay = 0., az = 0., aw = 0., tsum = 0.;
  unsigned am = 0;
  if (rl < nrl) {
    const float4 w = *reinterpret_cast<const float4*>(w2 + 4 * c4);
    for (int r = r0 + rl; r < r1; r += nrl) {
      const float sv = s[r];
      const float t = ds[r] * (nl == 0 ? 1.f - sv * sv : nl == 1 ? sv * (1.f - sv) : 1.f);
      const float4 v = *reinterpret_cast<const float4*>(h + (int64_t)r * ldh + 4 * c4);
      bool px = v.x > 0.f, py = v.y > 0.f, pz = v.z > 0.f, pw = v.w > 0.f;
      if (decide) {
        const uchar4 d = *reinterpret_cast<const uchar4*>(decide + (int64_t)r * C + 4 * c4);
        px = d.x; py = d.y; pz = d.z; pw = d.w;
      }
      float4 g;
      g.x = px ? t * w.x : 0.f; g.y = py ? t * w.y : 0.f;
      g.z = pz ? t * w.z : 0.f; g.w = pw ? t * w.w : 0.f;
      *reinterpret_cast<float4*>(dh + (int64_t)r * lddh + 4 * c4) = g;
      am = dgdm_amax4(am, g);
      const double td = (double)t;
      if (px) ax += td * (double)v.x;
      if (py) ay += td * (double)v.y;
      if (pz) az += td * (double)v.z;
      if (pw) aw += td * (double)v.w;
      if (c4 == 0) tsum += td;
    }
  }
  __shared__ double sm[256][4];
  __shared__ double st[256];
  sm[threadIdx.x][0] = ax; sm[threadIdx.x][1] = ay; sm[threadIdx.x][2] = az; sm[threadIdx.x][3] = aw;
  st[threadIdx.x] = tsum;
  __syncthreads();
  if (rl == 0) {
    for (int j = 1; j < nrl; ++j) {  // fixed order
      const double* t = sm[j * cols + c4];
      ax += t[0]; ay += t[1]; az += t[2]; aw += t[3];
      if (c4 == 0) tsum += st[j * cols];
    }
    double* P = partial + (int64_t)blockIdx.x * (C + 1);
    P[4 * c4 + 0] = ax; P[4 * c4 + 1] = ay; P[4 * c4 + 2] = az; P[4 * c4 + 3] = aw;
    if (c4 == 0) P[C] = tsum;
  }
  if (amax) dgdm_amax_commit(am, amax);     // dh is the operand of the score MLP's first-layer dX / dW GEMMs
}

// column sums of the float64 block partials, rounded to fp32 once: columns [0, C) -> dw2, column C -> db2.  One workgroup per column,
// one partial per thread (nb <= SCORE_BWD_BLOCKS = 256), a pairwise tree in LDS: the association is fixed by the thread index, so
// the sum is bitwise repeatable.  (A first version looped over the partials in one thread per column: 62 us of load latency.)
__global__ __launch_bounds__(256) void k_pool_score_bwd_final(const double* __restrict__ partial, int nb, int C, float* __restrict__ dw2,
                                                              float* __restrict__ db2) {
  __shared__ double sm[256];
  const int col = blockIdx.x, t = threadIdx.x;
  sm[t] = t < nb ? partial[(int64_t)t * (C + 1) + col] : 0.;
  __syncthreads();
#pragma unroll
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) sm[t] += sm[t + o];
    __syncthreads();
  }
  if (t == 0) {
    if (col < C) dw2[col] = (float)sm[0]; else db2[0] = (float)sm[0];
  }
}

constexpr int SCORE_BWD_BLOCKS = 256;

// ------------------------------------------------------------------ softmax over ALL nodes (nonlinearity='softmax',
// core/graph_layers.py:280-281: torch.softmax(scores, dim=0)) and the count behind min_score pooling.  One workgroup, fixed-order
// reductions: N is the node count of a batch, the vector a few hundred KB.
__device__ __forceinline__ float block_reduce_1024(float v, bool is_max, float* sm) {
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sm[0];
  for (int i = 1; i < 16; ++i) r = is_max ? fmaxf(r, sm[i]) : r + sm[i];
  return r;
}
__global__ __launch_bounds__(1024) void k_vec_softmax_fwd(const float* __restrict__ z, int N, float* __restrict__ s) {
  __shared__ float sm[16];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < N; i += 1024) m = fmaxf(m, z[i]);
  m = block_reduce_1024(m, true, sm);
  float l = 0.f;
  for (int i = threadIdx.x; i < N; i += 1024) l += __expf(z[i] - m);
  l = block_reduce_1024(l, false, sm);
  const float inv = 1.f / l;
  for (int i = threadIdx.x; i < N; i += 1024) s[i] = __expf(z[i] - m) * inv;
}
__global__ __launch_bounds__(1024) void k_vec_softmax_bwd(const float* __restrict__ s, const float* __restrict__ ds, int N, float* __restrict__ dz) {
  __shared__ float sm[16];
  float d = 0.f;
  for (int i = threadIdx.x; i < N; i += 1024) d += s[i] * ds[i];
  d = block_reduce_1024(d, false, sm);
  for (int i = threadIdx.x; i < N; i += 1024) dz[i] = s[i] * (ds[i] - d);
}
__global__ __launch_bounds__(1024) void k_count_ge(const float* __restrict__ s, int N, float thr, int32_t* __restrict__ out) {
  // integer count all the way (a float sum loses exactness beyond 2^24 nodes; the entry point admits N < 2^30)
  __shared__ int32_t sm[16];
  int32_t c = 0;
  for (int i = threadIdx.x; i < N; i += 1024) c += s[i] >= thr ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t r = 0;
    for (int i = 0; i < 16; ++i) r += sm[i];
    out[0] = r;
  }
}

// ------------------------------------------------------------------ top-k
// order-preserving map float -> uint32 (larger float <=> larger key); -0 == +0, NaNs by bit pattern
__device__ __forceinline__ uint32_t key_of(float f) {
  uint32_t u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0u;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// state[0] = key prefix found so far, state[1] = how many still to take among keys matching it
struct SelectState { uint32_t prefix; uint32_t need; };

// Pass p (0..3, most significant byte first): every block first resolves pass p-1 from its finished
// histogram (256 bins: redundant per block, but it saves a launch and a sync), then histograms
// byte (3 - p) of the keys that match the prefix so far.
__device__ __forceinline__ void resolve_pass(const uint32_t* __restrict__ hist, SelectState in, int pass, SelectState* out) {
  // executed by one wave: digit d = largest with count(digits > d) < need <= count(digits >= d)
  const int lane = threadIdx.x & 63;
  uint32_t c[4];
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { c[j] = hist[255 - (4 * lane + j)]; mine += c[j]; }   // lane 0 holds digits 255..252
  uint32_t incl = mine;   // inclusive scan over lanes (digits descending)
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  uint32_t above = incl - mine;
  int found = -1;
  uint32_t need_after = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (found < 0 && above + c[j] >= in.need && above < in.need) { found = 255 - (4 * lane + j); need_after = in.need - above; }
    above += c[j];
  }
  const unsigned long long who = __ballot(found >= 0);
  const int src = __ffsll((long long)who) - 1;   // exactly one lane when need <= matching count
  const int digit = __shfl(found, src < 0 ? 0 : src, 64);
  const uint32_t nd = __shfl(need_after, src < 0 ? 0 : src, 64);
  out->prefix = in.prefix | ((uint32_t)digit << (8 * (3 - pass)));
  out->need = nd;
}

__global__ __launch_bounds__(256) void k_topk_hist(const float* __restrict__ s, int N, int k, int pass, uint32_t* __restrict__ hist_all,
                                                   SelectState* __restrict__ states) {
  __shared__ uint32_t lh[256];
  __shared__ SelectState st;
  lh[threadIdx.x] = 0;
  if (threadIdx.x < 64) {
    SelectState cur;
    if (pass == 0) { cur.prefix = 0; cur.need = (uint32_t)k; }
    else resolve_pass(hist_all + 256 * (pass - 1), states[pass - 1], pass - 1, &cur);
    if (threadIdx.x == 0) { st = cur; if (blockIdx.x == 0) states[pass] = cur; }
  }
  __syncthreads();
  const uint32_t prefix = st.prefix;
  const uint32_t mask = pass == 0 ? 0u : (0xffffffffu << (8 * (4 - pass)));
  const int shift = 8 * (3 - pass);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    const uint32_t key = key_of(s[i]);
    if ((key & mask) == prefix) atomicAdd(&lh[(key >> shift) & 255u], 1u);
  }
  __syncthreads();
  const uint32_t v = lh[threadIdx.x];
  if (v) atomicAdd(&hist_all[256 * pass + threadIdx.x], v);   // integer counts: order-free, deterministic
}

constexpr int CP_ITEMS = 1024;   // nodes per compaction block (4 per thread)

// per block: number of keys above the threshold and equal to it
__global__ __launch_bounds__(256) void k_topk_count(const float* __restrict__ s, int N, const uint32_t* __restrict__ hist_all,
                                                    SelectState* __restrict__ states, uint32_t* __restrict__ counts) {
  __shared__ SelectState st;
  if (threadIdx.x < 64) {
    SelectState fin;
    resolve_pass(hist_all + 256 * 3, states[3], 3, &fin);
    if (threadIdx.x == 0) { st = fin; if (blockIdx.x == 0) states[4] = fin; }
  }
  __syncthreads();
  const uint32_t T = st.prefix;
  uint32_t gt = 0, eq = 0;
  const int base = blockIdx.x * CP_ITEMS;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = base + 4 * threadIdx.x + j;
    if (i < N) { const uint32_t key = key_of(s[i]); gt += key > T; eq += key == T; }
  }
  __shared__ uint32_t sg[256], se[256];
  sg[threadIdx.x] = gt; se[threadIdx.x] = eq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sg[threadIdx.x] += sg[threadIdx.x + o]; se[threadIdx.x] += se[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { counts[2 * blockIdx.x] = sg[0]; counts[2 * blockIdx.x + 1] = se[0]; }
}

// single block: exclusive scans over the compaction blocks -> bases[b] = {first output slot, equal-keys seen before}
__global__ __launch_bounds__(256) void k_topk_scan(const uint32_t* __restrict__ counts, int nb, const SelectState* __restrict__ states,
                                                   uint32_t* __restrict__ bases) {
  __shared__ uint32_t carry_sel, carry_eq;
  __shared__ uint32_t ssel[256], seq[256];
  const uint32_t need = states[4].need;   // equal keys to take
  if (threadIdx.x == 0) { carry_sel = 0; carry_eq = 0; }
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += 256) {
    const int b = b0 + threadIdx.x;
    const uint32_t eq = b < nb ? counts[2 * b + 1] : 0, gt = b < nb ? counts[2 * b] : 0;
    seq[threadIdx.x] = eq;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {   // inclusive scan of eq
      const uint32_t t = threadIdx.x >= o ? seq[threadIdx.x - o] : 0;
      __syncthreads();
      seq[threadIdx.x] += t;
      __syncthreads();
    }
    const uint32_t eq_before = carry_eq + seq[threadIdx.x] - eq;
    const uint32_t take = eq_before >= need ? 0u : min(eq, need - eq_before);
    ssel[threadIdx.x] = gt + take;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const uint32_t t = threadIdx.x >= o ? ssel[threadIdx.x - o] : 0;
      __syncthreads();
      ssel[threadIdx.x] += t;
      __syncthreads();
    }
    if (b < nb) { bases[2 * b] = carry_sel + ssel[threadIdx.x] - (gt + take); bases[2 * b + 1] = eq_before; }
    __syncthreads();
    if (threadIdx.x == 255) { carry_sel += ssel[255]; carry_eq += seq[255]; }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_topk_compact(const float* __restrict__ s, int N, int k, const SelectState* __restrict__ states,
                                                      const uint32_t* __restrict__ bases, int64_t* __restrict__ perm,
                                                      int32_t* __restrict__ node_map) {
  const uint32_t T = states[4].prefix, need = states[4].need;
  const int base = blockIdx.x * CP_ITEMS;
  uint32_t key[4];
  bool gt[4], eq[4];
  uint32_t neq = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = base + 4 * threadIdx.x + j;
    key[j] = i < N ? key_of(s[i]) : 0u;
    gt[j] = i < N && key[j] > T;
    eq[j] = i < N && key[j] == T;
    neq += eq[j];
  }
  __shared__ uint32_t sc[256];
  // exclusive scan of equal-key counts over the block's threads
  sc[threadIdx.x] = neq;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t t = threadIdx.x >= o ? sc[threadIdx.x - o] : 0;
    __syncthreads();
    sc[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t eq_rank = bases[2 * blockIdx.x + 1] + sc[threadIdx.x] - neq;
  bool sel[4];
  uint32_t nsel = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sel[j] = gt[j] || (eq[j] && eq_rank < need);
    eq_rank += eq[j];
    nsel += sel[j];
  }
  __syncthreads();
  sc[threadIdx.x] = nsel;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t t = threadIdx.x >= o ? sc[threadIdx.x - o] : 0;
    __syncthreads();
    sc[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t pos = bases[2 * blockIdx.x] + sc[threadIdx.x] - nsel;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = base + 4 * threadIdx.x + j;
    if (i >= N) break;
    if (sel[j]) {
      if ((int)pos < k) perm[pos] = i;
      node_map[i] = (int32_t)pos;
      ++pos;
    } else {
      node_map[i] = -1;
    }
  }
}

// ------------------------------------------------------------------ gather * score
// one 16-lane group per output row (float4 per lane per 64-column chunk)
__global__ __launch_bounds__(256) void k_pool_gather_fwd(const float* __restrict__ x, int64_t ldx, const float* __restrict__ s,
                                                         const int64_t* __restrict__ perm, int k, int C, float mult,
                                                         float* __restrict__ out, int64_t ldo) {
  const int sub = threadIdx.x & 15, j = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (j >= k) return;
  const int64_t src = perm[j];
  const float sc = s[src] * mult;
  for (int c = 4 * sub; c < C; c += 64) {
    float4 v = *reinterpret_cast<const float4*>(x + src * ldx + c);
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    *reinterpret_cast<float4*>(out + (int64_t)j * ldo + c) = v;
  }
}

// per source row i: dx[i] = g[node_map[i]] * s[i] * mult (0 when dropped), ds[i] = mult * <g[node_map[i]], x[i]>
__global__ __launch_bounds__(256) void k_pool_gather_bwd(const float* __restrict__ g, int64_t ldg, const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ s, const int32_t* __restrict__ node_map, int N, int C,
                                                         float mult, float* __restrict__ dx, int64_t lddx, float* __restrict__ ds) {
  const int sub = threadIdx.x & 15, i = blockIdx.x * 16 + (threadIdx.x >> 4);
  float dot = 0.f;
  if (i < N) {
    const int nm = node_map[i];
    const float sc = s[i] * mult;
    for (int c = 4 * sub; c < C; c += 64) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (nm >= 0) {
        const float4 gv = *reinterpret_cast<const float4*>(g + (int64_t)nm * ldg + c);
        const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)i * ldx + c);
        dot += gv.x * xv.x + gv.y * xv.y + gv.z * xv.z + gv.w * xv.w;
        o.x = gv.x * sc; o.y = gv.y * sc; o.z = gv.z * sc; o.w = gv.w * sc;
      }
      *reinterpret_cast<float4*>(dx + (int64_t)i * lddx + c) = o;
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  if (sub == 0 && i < N) ds[i] = dot * mult;
}

// ------------------------------------------------------------------ edge relabel
__global__ __launch_bounds__(256) void k_edge_relabel(const int64_t* __restrict__ ei, int64_t E, const int32_t* __restrict__ node_map, int N,
                                                      int64_t* __restrict__ out) {
  const int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (e >= E) return;
  const int64_t u = ei[e], v = ei[E + e];
  int64_t mu = -1, mv = -1;
  if (u >= 0 && u < N && v >= 0 && v < N) {
    const int a = node_map[u], b = node_map[v];
    if (a >= 0 && b >= 0) { mu = a; mv = b; }
  }
  out[e] = mu;
  out[E + e] = mv;
}

// ------------------------------------------------------------------ unpool + skip + relu
__global__ __launch_bounds__(256) void k_unpool_add_relu_fwd(const float* __restrict__ xc, int64_t ldxc, const float* __restrict__ skip,
                                                             int64_t lds, const int32_t* __restrict__ node_map, int N, int C,
                                                             float* __restrict__ out, int64_t ldo, const uint8_t* __restrict__ decide) {
  const int sub = threadIdx.x & 15, i = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (i >= N) return;
  const int nm = node_map[i];
  for (int c = 4 * sub; c < C; c += 64) {
    float4 v = *reinterpret_cast<const float4*>(skip + (int64_t)i * lds + c);
    if (nm >= 0) {
      const float4 u = *reinterpret_cast<const float4*>(xc + (int64_t)nm * ldxc + c);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (decide) {
      const uchar4 d = *reinterpret_cast<const uchar4*>(decide + (int64_t)i * C + c);
      v.x = d.x ? v.x : 0.f; v.y = d.y ? v.y : 0.f; v.z = d.z ? v.z : 0.f; v.w = d.w ? v.w : 0.f;
    } else {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    *reinterpret_cast<float4*>(out + (int64_t)i * ldo + c) = v;
  }
}

// dskip[i] = g[i] * [out[i] > 0]; dxc[node_map[i]] = dskip[i] for kept rows (each coarse row has exactly one source)
__global__ __launch_bounds__(256) void k_unpool_add_relu_bwd(const float* __restrict__ g, int64_t ldg, const float* __restrict__ out,
                                                             int64_t ldo, const int32_t* __restrict__ node_map, int N, int C,
                                                             float* __restrict__ dskip, int64_t ldds, float* __restrict__ dxc,
                                                             int64_t lddxc, const uint8_t* __restrict__ decide) {
  const int sub = threadIdx.x & 15, i = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (i >= N) return;
  const int nm = node_map[i];
  for (int c = 4 * sub; c < C; c += 64) {
    const float4 gv = *reinterpret_cast<const float4*>(g + (int64_t)i * ldg + c);
    const float4 ov = *reinterpret_cast<const float4*>(out + (int64_t)i * ldo + c);
    bool px = ov.x > 0.f, py = ov.y > 0.f, pz = ov.z > 0.f, pw = ov.w > 0.f;
    if (decide) {
      const uchar4 m = *reinterpret_cast<const uchar4*>(decide + (int64_t)i * C + c);
      px = m.x; py = m.y; pz = m.z; pw = m.w;
    }
    float4 d;
    d.x = px ? gv.x : 0.f; d.y = py ? gv.y : 0.f; d.z = pz ? gv.z : 0.f; d.w = pw ? gv.w : 0.f;
    *reinterpret_cast<float4*>(dskip + (int64_t)i * ldds + c) = d;
    if (nm >= 0) *reinterpret_cast<float4*>(dxc + (int64_t)nm * lddxc + c) = d;
  }
}

inline bool rows_ok(const float* p, int64_t ld, int C) { return p && (ld & 3) == 0 && ld >= C && dgdm_aligned16(p); }

}  // namespace

extern "C" int dgdm_vec_softmax_fwd(const float* z, int32_t N, float* s, void* stream) {
  if (N < 0) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_OK;
  if (!z || !s) return DGDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_vec_softmax_fwd, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), z, N, s);
  return dgdm_launch_status();
}
extern "C" int dgdm_vec_softmax_bwd(const float* s, const float* ds, int32_t N, float* dz, void* stream) {
  if (N < 0) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_OK;
  if (!s || !ds || !dz) return DGDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_vec_softmax_bwd, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), s, ds, N, dz);
  return dgdm_launch_status();
}
extern "C" int dgdm_count_ge(const float* s, int32_t N, float threshold, int32_t* out, void* stream) {
  if (N < 0 || !out) return DGDM_ERR_INVALID_ARG;
  if (N > 0 && !s) return DGDM_ERR_INVALID_ARG;
  if (N >= (1 << 24) * 64) return DGDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_count_ge, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), s, N, threshold, out);
  return dgdm_launch_status();
}

extern "C" int dgdm_pool_score_fwd(const float* h, int64_t ldh, const float* w2, const float* b2, int32_t N, int32_t C, float* s,
                                   const uint8_t* decide, int32_t nonlinearity, void* stream) {
  if (N < 0 || C <= 0 || nonlinearity < 0 || nonlinearity > 2) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_OK;
  if (!h || !w2 || !b2 || !s) return DGDM_ERR_INVALID_ARG;
  if ((C & 3) || !rows_ok(h, ldh, C) || !dgdm_aligned16(w2)) return DGDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_pool_score_fwd, dim3((N + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), h, ldh, w2, b2, N, C, s,
                     decide, nonlinearity);
  return dgdm_launch_status();
}

extern "C" size_t dgdm_pool_score_bwd_workspace_bytes(int32_t N, int32_t C) {
  if (N <= 0 || C <= 0) return 0;
  return (size_t)(SCORE_BWD_BLOCKS + 16) * ((size_t)C + 1) * sizeof(double);
}

extern "C" int dgdm_pool_score_bwd(const float* h, int64_t ldh, const float* w2, const float* s, const float* ds, int32_t N, int32_t C,
                                   float* dh, int64_t lddh, float* dw2, float* db2, const uint8_t* decide, int32_t nonlinearity,
                                   void* workspace, size_t workspace_bytes, uint32_t* amax, void* stream_) {
  if (N < 0 || C <= 0 || nonlinearity < 0 || nonlinearity > 2) return DGDM_ERR_INVALID_ARG;
  if (!dw2 || !db2) return DGDM_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (N == 0) {
    dgdm_fill_async(dw2, 0, sizeof(float) * C, st);
    dgdm_fill_async(db2, 0, sizeof(float), st);
    return dgdm_launch_status();
  }
  if (!h || !w2 || !s || !ds || !dh || !workspace) return DGDM_ERR_INVALID_ARG;
  if ((C & 3) || C > 256 || 256 % (C / 4) != 0 || !rows_ok(h, ldh, C) || !rows_ok(dh, lddh, C) || !dgdm_aligned16(w2)) return DGDM_ERR_UNSUPPORTED;
  if (workspace_bytes < dgdm_pool_score_bwd_workspace_bytes(N, C)) return DGDM_ERR_WORKSPACE;
  const int rpb = (N + SCORE_BWD_BLOCKS - 1) / SCORE_BWD_BLOCKS;
  const int nb = (N + rpb - 1) / rpb;
  double* partial = static_cast<double*>(workspace);       // torch allocations are 256-byte aligned; checked below
  if (reinterpret_cast<uintptr_t>(workspace) & 7) return DGDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_pool_score_bwd, dim3(nb), dim3(256), 0, st, h, ldh, w2, s, ds, N, C, rpb, dh, lddh, partial, decide, nonlinearity, amax);
  hipLaunchKernelGGL(k_pool_score_bwd_final, dim3(C + 1), dim3(256), 0, st, partial, nb, C, dw2, db2);
  return dgdm_launch_status();
}

extern "C" size_t dgdm_topk_perm_workspace_bytes(int32_t N) {
  if (N <= 0) return 0;
  const size_t nb = ((size_t)N + CP_ITEMS - 1) / CP_ITEMS;
  return 4 * 256 * sizeof(uint32_t) + 8 * sizeof(SelectState) + 4 * nb * sizeof(uint32_t);
}

extern "C" int dgdm_topk_perm(const float* s, int32_t N, int32_t k, int64_t* perm, int32_t* node_map, void* workspace,
                              size_t workspace_bytes, void* stream_) {
  if (N < 0 || k < 0 || k > N) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_OK;
  if (!s || !node_map || (k > 0 && !perm) || !workspace) return DGDM_ERR_INVALID_ARG;
  if (workspace_bytes < dgdm_topk_perm_workspace_bytes(N)) return DGDM_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (k == 0) {
    dgdm_fill_async(node_map, 0xff, sizeof(int32_t) * N, st);
    return dgdm_launch_status();
  }
  const int nb = (N + CP_ITEMS - 1) / CP_ITEMS;
  uint32_t* hist = static_cast<uint32_t*>(workspace);
  SelectState* states = reinterpret_cast<SelectState*>(hist + 4 * 256);
  uint32_t* counts = reinterpret_cast<uint32_t*>(states + 8);
  uint32_t* bases = counts + 2 * nb;
  dgdm_fill_async(hist, 0, 4 * 256 * sizeof(uint32_t), st);
  const int hb = min(nb * 4, 1024);
  for (int pass = 0; pass < 4; ++pass)
    hipLaunchKernelGGL(k_topk_hist, dim3(hb), dim3(256), 0, st, s, N, k, pass, hist, states);
  hipLaunchKernelGGL(k_topk_count, dim3(nb), dim3(256), 0, st, s, N, hist, states, counts);
  hipLaunchKernelGGL(k_topk_scan, dim3(1), dim3(256), 0, st, counts, nb, states, bases);
  hipLaunchKernelGGL(k_topk_compact, dim3(nb), dim3(256), 0, st, s, N, k, states, bases, perm, node_map);
  return dgdm_launch_status();
}

extern "C" int dgdm_pool_gather_fwd(const float* x, int64_t ldx, const float* s, const int64_t* perm, int32_t k, int32_t C, float mult,
                                    float* out, int64_t ldo, void* stream) {
  if (k < 0 || C <= 0) return DGDM_ERR_INVALID_ARG;
  if (k == 0) return DGDM_OK;
  if (!x || !s || !perm || !out) return DGDM_ERR_INVALID_ARG;
  if ((C & 3) || !rows_ok(x, ldx, C) || !rows_ok(out, ldo, C)) return DGDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_pool_gather_fwd, dim3((k + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, s, perm, k, C, mult,
                     out, ldo);
  return dgdm_launch_status();
}

extern "C" int dgdm_pool_gather_bwd(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* s, const int32_t* node_map,
                                    int32_t N, int32_t C, float mult, float* dx, int64_t lddx, float* ds, void* stream) {
  if (N < 0 || C <= 0) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_OK;
  if (!g || !x || !s || !node_map || !dx || !ds) return DGDM_ERR_INVALID_ARG;
  if ((C & 3) || !rows_ok(g, ldg, C) || !rows_ok(x, ldx, C) || !rows_ok(dx, lddx, C)) return DGDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_pool_gather_bwd, dim3((N + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), g, ldg, x, ldx, s, node_map, N,
                     C, mult, dx, lddx, ds);
  return dgdm_launch_status();
}

extern "C" int dgdm_edge_relabel(const int64_t* edge_index, int64_t E, const int32_t* node_map, int32_t N, int64_t* out, void* stream) {
  if (E < 0 || N < 0) return DGDM_ERR_INVALID_ARG;
  if (E == 0) return DGDM_OK;
  if (!edge_index || !node_map || !out) return DGDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_edge_relabel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), edge_index, E,
                     node_map, N, out);
  return dgdm_launch_status();
}

extern "C" int dgdm_unpool_add_relu_fwd(const float* xc, int64_t ldxc, const float* skip, int64_t lds, const int32_t* node_map, int32_t N,
                                        int32_t C, float* out, int64_t ldo, const uint8_t* decide, void* stream) {
  if (N < 0 || C <= 0) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_OK;
  if (!xc || !skip || !node_map || !out) return DGDM_ERR_INVALID_ARG;
  if ((C & 3) || !rows_ok(xc, ldxc, C) || !rows_ok(skip, lds, C) || !rows_ok(out, ldo, C)) return DGDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_unpool_add_relu_fwd, dim3((N + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), xc, ldxc, skip, lds,
                     node_map, N, C, out, ldo, decide);
  return dgdm_launch_status();
}

extern "C" int dgdm_unpool_add_relu_bwd(const float* g, int64_t ldg, const float* out, int64_t ldo, const int32_t* node_map, int32_t N,
                                        int32_t C, float* dskip, int64_t ldds, float* dxc, int64_t lddxc, const uint8_t* decide,
                                        void* stream) {
  if (N < 0 || C <= 0) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_OK;
  if (!g || !out || !node_map || !dskip || !dxc) return DGDM_ERR_INVALID_ARG;
  if ((C & 3) || !rows_ok(g, ldg, C) || !rows_ok(out, ldo, C) || !rows_ok(dskip, ldds, C) || !rows_ok(dxc, lddxc, C)) return DGDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_unpool_add_relu_bwd, dim3((N + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), g, ldg, out, ldo, node_map,
                     N, C, dskip, ldds, dxc, lddxc, decide);
  return dgdm_launch_status();
}
