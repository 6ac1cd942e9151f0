// K11: tissue-graph edge construction on the GPU -- the step immediately upstream of the model
// (reference: TissueGraphBuilder._create_edges / _remove_duplicate_edges / the edge part of
// _to_pytorch_geometric, preprocessing/tissue_graph_builder.py:269-414, which run scikit-learn
// NearestNeighbors + cosine_similarity and Python loops on the CPU).
//
//   dgdm_knn2d            K nearest points (self included) of every 2-D coordinate, brute force through
//                         LDS tiles, ascending by (distance, index); squared distances are formed as
//                         fl(fl(dx*dx) + fl(dy*dy)) (no FMA contraction), so a float32 CPU restatement
//                         reproduces indices and distances bit for bit
//   dgdm_row_sqnorm       ||x_i||^2
//   dgdm_knn_gram_partial / dgdm_knn_gram_merge
//                         K nearest rows in feature space from a Gram block G^T[j][q] = x_j . x_q
//                         (produced by the GEMM kernels): d^2 = |x_q|^2 + |x_j|^2 - 2 G, candidates
//                         split into segments (parallelism), merged in index order (same tie rule)
//   dgdm_edge_candidates  spatial (weight exp(-10 d) >= thr) and morphological (cosine >= thr)
//                         candidates in the reference's enumeration order
//   dgdm_edge_dedup_*     key = sorted (src, tgt): keep the heaviest candidate (earliest wins ties),
//                         output ordered by the key's first occurrence -- Python dict semantics --
//                         through an open-addressing table with integer atomics (max / min are
//                         order-free, so the result is deterministic)
//   dgdm_edge_emit        both directions of every kept edge as consecutive columns, attributes
//                         zero-padded to edge_dim (repair R6 of SURVEY.md D11)
#include "common.hpp"

namespace {

// branch-free stable insertion of (d, i, g) into an ascending list (strict <: equal keys keep scan order)
template <int KMAX, bool WITH_G>
__device__ __forceinline__ void list_insert(float (&bd)[KMAX], int (&bi)[KMAX], float (&bg)[KMAX], float d, int i, float g) {
  bool sw = false;   // once the insertion point is found every later slot just shifts (equal keys keep their order)
#pragma unroll
  for (int p = 0; p < KMAX; ++p) {
    sw = sw || d < bd[p];
    const float td = bd[p]; const int ti = bi[p];
    bd[p] = sw ? d : td; bi[p] = sw ? i : ti;
    d = sw ? td : d; i = sw ? ti : i;
    if (WITH_G) { const float tg = bg[p]; bg[p] = sw ? g : tg; g = sw ? tg : g; }
  }
}

constexpr int KNN_TILE = 1024;

template <int KMAX>
__global__ __launch_bounds__(128) void k_knn2d(const float* __restrict__ coords, int N, int K, int32_t* __restrict__ idx,
                                                float* __restrict__ dist) {
  __shared__ float2 tile[KNN_TILE];
  const int q = blockIdx.x * 128 + threadIdx.x;
  const float2 me = q < N ? reinterpret_cast<const float2*>(coords)[q] : make_float2(0.f, 0.f);
  float bd[KMAX], bg[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int p = 0; p < KMAX; ++p) { bd[p] = __builtin_inff(); bi[p] = -1; bg[p] = 0.f; }
  for (int t0 = 0; t0 < N; t0 += KNN_TILE) {
    __syncthreads();
    for (int t = threadIdx.x; t < KNN_TILE; t += 128)
      if (t0 + t < N) tile[t] = reinterpret_cast<const float2*>(coords)[t0 + t];
    __syncthreads();
    const int lim = min(KNN_TILE, N - t0);
    for (int t = 0; t < lim; ++t) {
      const float dx = __fsub_rn(me.x, tile[t].x), dy = __fsub_rn(me.y, tile[t].y);
      const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
      if (d2 < bd[KMAX - 1]) list_insert<KMAX, false>(bd, bi, bg, d2, t0 + t, 0.f);
    }
  }
  if (q < N) {
#pragma unroll
    for (int p = 0; p < KMAX; ++p)
      if (p < K) { idx[(int64_t)q * K + p] = bi[p]; dist[(int64_t)q * K + p] = sqrtf(bd[p]); }
  }
}

__global__ __launch_bounds__(256) void k_row_sqnorm(const float* __restrict__ X, int64_t ldx, int N, int F, float* __restrict__ sq) {
  const int sub = threadIdx.x & 15, row = blockIdx.x * 16 + (threadIdx.x >> 4);
  float acc = 0.f;
  if (row < N)
    for (int c = 4 * sub; c < F; c += 64) {
      const float4 v = *reinterpret_cast<const float4*>(X + (int64_t)row * ldx + c);
      acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (sub == 0 && row < N) sq[row] = acc;
}

// partial[(seg * B + q) * KMAX + p] = p-th nearest candidate of query q0 + q among rows [seg*per, (seg+1)*per)
struct KnnRec { float d2; int32_t idx; float g; };

template <int KMAX>
__global__ __launch_bounds__(128) void k_knn_gram_partial(const float* __restrict__ GT, int64_t ldg, const float* __restrict__ sq, int N,
                                                          int q0, int B, int per, KnnRec* __restrict__ partial) {
  const int q = blockIdx.x * 128 + threadIdx.x, seg = blockIdx.y;
  if (q >= B) return;
  const float sqq = sq[q0 + q];
  float bd[KMAX], bg[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int p = 0; p < KMAX; ++p) { bd[p] = __builtin_inff(); bi[p] = -1; bg[p] = 0.f; }
  const int j0 = seg * per, j1 = min(N, j0 + per);
  for (int j = j0; j < j1; ++j) {
    const float g = GT[(int64_t)j * ldg + q];          // consecutive threads = consecutive queries: coalesced
    float d2 = fmaxf(sqq + sq[j] - 2.f * g, 0.f);
    if (j == q0 + q) d2 = 0.f;                           // the point itself
    if (d2 < bd[KMAX - 1]) list_insert<KMAX, true>(bd, bi, bg, d2, j, g);
  }
  KnnRec* out = partial + ((int64_t)seg * B + q) * KMAX;
#pragma unroll
  for (int p = 0; p < KMAX; ++p) out[p] = KnnRec{bd[p], bi[p], bg[p]};
}

template <int KMAX>
__global__ __launch_bounds__(128) void k_knn_gram_merge(const KnnRec* __restrict__ partial, int nseg, int q0, int B, int K,
                                                        const float* __restrict__ sq, int32_t* __restrict__ idx,
                                                        float* __restrict__ sim) {
  const int q = blockIdx.x * 128 + threadIdx.x;
  if (q >= B) return;
  float bd[KMAX], bg[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int p = 0; p < KMAX; ++p) { bd[p] = __builtin_inff(); bi[p] = -1; bg[p] = 0.f; }
  for (int s = 0; s < nseg; ++s) {   // segments in index order, each list ascending: the (distance, index) order survives
    const KnnRec* in = partial + ((int64_t)s * B + q) * KMAX;
    for (int p = 0; p < KMAX; ++p) {
      const KnnRec r = in[p];
      if (r.idx < 0 || !(r.d2 < bd[KMAX - 1])) break;
      list_insert<KMAX, true>(bd, bi, bg, r.d2, r.idx, r.g);
    }
  }
  const float nq = sqrtf(sq[q0 + q]);
#pragma unroll
  for (int p = 0; p < KMAX; ++p)
    if (p < K) {
      const int j = bi[p];
      idx[(int64_t)(q0 + q) * K + p] = j;
      const float nj = j >= 0 ? sqrtf(sq[j]) : 0.f;
      sim[(int64_t)(q0 + q) * K + p] = (nq > 0.f && nj > 0.f) ? bg[p] / (nq * nj) : 0.f;
    }
}

// cosine similarity of every (row, neighbour) pair from a direct dot product in a fixed order: the
// value of (i, j) and of (j, i) is the same bit pattern (the Gram GEMM's is not: its accumulation
// order depends on which operand a row is), which the duplicate rule "heavier wins, first on ties"
// needs in order to behave as it does on the reference's symmetric similarity matrix.
__global__ __launch_bounds__(256) void k_pair_cosine(const float* __restrict__ X, int64_t ldx, const float* __restrict__ sq,
                                                     const int32_t* __restrict__ idx, int64_t pairs, int K, int F,
                                                     float* __restrict__ sim) {
  const int sub = threadIdx.x & 15;
  const int64_t pr = blockIdx.x * (int64_t)16 + (threadIdx.x >> 4);
  float acc = 0.f;
  int i = 0, j = -1;
  if (pr < pairs) {
    i = (int)(pr / K);
    j = idx[pr];
    if (j >= 0)
      for (int c = 4 * sub; c < F; c += 64) {
        const float4 a = *reinterpret_cast<const float4*>(X + (int64_t)i * ldx + c);
        const float4 b = *reinterpret_cast<const float4*>(X + (int64_t)j * ldx + c);
        acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
      }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (sub == 0 && pr < pairs) {
    const float ni = sqrtf(sq[i]), nj = j >= 0 ? sqrtf(sq[j]) : 0.f;
    sim[pr] = (ni > 0.f && nj > 0.f) ? acc / (ni * nj) : 0.f;
  }
}

// ---------------------------------------------------------------- candidates
// seq < N*ks: spatial (i = seq / ks, neighbour column 1 + seq % ks); otherwise morphological
struct Cand { int32_t src, tgt; float w, f0, f1; int32_t type; };

__device__ __forceinline__ bool make_cand(int64_t seq, int N, int ks, int km, int Ks1, int Km1, const int32_t* __restrict__ sidx,
                                          const float* __restrict__ sdist, const int32_t* __restrict__ midx,
                                          const float* __restrict__ msim, float thr, Cand* c) {
  const int64_t nsp = (int64_t)N * ks;
  if (seq < nsp) {
    const int i = (int)(seq / ks), r = 1 + (int)(seq % ks);
    const int j = sidx[(int64_t)i * Ks1 + r];
    const float d = sdist[(int64_t)i * Ks1 + r];
    const float w = expf(-d * 10.f);
    *c = Cand{i, j, w, d, w, 0};
    return j >= 0 && w >= thr;
  }
  const int64_t s2 = seq - nsp;
  const int i = (int)(s2 / km), r = 1 + (int)(s2 % km);
  const int j = midx[(int64_t)i * Km1 + r];
  const float s = msim[(int64_t)i * Km1 + r];
  *c = Cand{i, j, s, s, 0.f, 1};
  return j >= 0 && s >= thr;
}

__device__ __forceinline__ uint32_t wkey(float f) {   // order-preserving float -> uint
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ uint64_t hash64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

constexpr uint64_t EMPTY_KEY = ~0ULL;

struct Table { unsigned long long* keys; unsigned long long* best; uint32_t* first; uint64_t mask; };

__global__ __launch_bounds__(256) void k_table_init(Table t, int64_t cap, int32_t* __restrict__ winner_at, int64_t L) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i < cap) { t.keys[i] = EMPTY_KEY; t.best[i] = 0ULL; t.first[i] = 0xffffffffu; }
  if (i < L) winner_at[i] = -1;
}

__global__ __launch_bounds__(256) void k_table_insert(Table t, int64_t L, int N, int ks, int km, int Ks1, int Km1,
                                                      const int32_t* __restrict__ sidx, const float* __restrict__ sdist,
                                                      const int32_t* __restrict__ midx, const float* __restrict__ msim, float thr) {
  const int64_t seq = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (seq >= L) return;
  Cand c;
  if (!make_cand(seq, N, ks, km, Ks1, Km1, sidx, sdist, midx, msim, thr, &c)) return;
  const uint64_t key = ((uint64_t)(uint32_t)min(c.src, c.tgt) << 32) | (uint32_t)max(c.src, c.tgt);
  uint64_t slot = hash64(key) & t.mask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&t.keys[slot], EMPTY_KEY, (unsigned long long)key);
    if (prev == EMPTY_KEY || prev == key) break;
    slot = (slot + 1) & t.mask;
  }
  // heaviest first, then earliest: max over (weight key, ~seq)
  atomicMax(&t.best[slot], ((unsigned long long)wkey(c.w) << 32) | (uint32_t)(0xffffffffu - (uint32_t)seq));
  atomicMin(&t.first[slot], (uint32_t)seq);
}

// winner_at[first occurrence of the key] = seq of the candidate that survives
__global__ __launch_bounds__(256) void k_table_winners(Table t, int64_t L, int N, int ks, int km, int Ks1, int Km1,
                                                       const int32_t* __restrict__ sidx, const float* __restrict__ sdist,
                                                       const int32_t* __restrict__ midx, const float* __restrict__ msim, float thr,
                                                       int32_t* __restrict__ winner_at) {
  const int64_t seq = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (seq >= L) return;
  Cand c;
  if (!make_cand(seq, N, ks, km, Ks1, Km1, sidx, sdist, midx, msim, thr, &c)) return;
  const uint64_t key = ((uint64_t)(uint32_t)min(c.src, c.tgt) << 32) | (uint32_t)max(c.src, c.tgt);
  uint64_t slot = hash64(key) & t.mask;
  while (t.keys[slot] != key) slot = (slot + 1) & t.mask;
  const uint32_t best_seq = 0xffffffffu - (uint32_t)(t.best[slot] & 0xffffffffu);
  if (best_seq == (uint32_t)seq) winner_at[t.first[slot]] = (int32_t)seq;
}

constexpr int SC_ITEMS = 1024;

__global__ __launch_bounds__(256) void k_flag_count(const int32_t* __restrict__ winner_at, int64_t L, uint32_t* __restrict__ counts) {
  uint32_t c = 0;
  const int64_t base = blockIdx.x * (int64_t)SC_ITEMS;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = base + 4 * threadIdx.x + j;
    c += (i < L && winner_at[i] >= 0);
  }
  __shared__ uint32_t sm[256];
  sm[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[blockIdx.x] = sm[0];
}

__global__ __launch_bounds__(256) void k_block_scan(const uint32_t* __restrict__ counts, int nb, uint32_t* __restrict__ bases,
                                                    int64_t* __restrict__ total) {
  __shared__ uint32_t sm[256];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += 256) {
    const int b = b0 + threadIdx.x;
    const uint32_t v = b < nb ? counts[b] : 0;
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const uint32_t t = threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
      __syncthreads();
      sm[threadIdx.x] += t;
      __syncthreads();
    }
    if (b < nb) bases[b] = carry + sm[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 255) carry += sm[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k_edge_emit(const int32_t* __restrict__ winner_at, int64_t L, const uint32_t* __restrict__ bases,
                                                   int N, int ks, int km, int Ks1, int Km1, const int32_t* __restrict__ sidx,
                                                   const float* __restrict__ sdist, const int32_t* __restrict__ midx,
                                                   const float* __restrict__ msim, float thr, int64_t U, int edge_dim,
                                                   int64_t* __restrict__ edge_index, float* __restrict__ edge_attr,
                                                   int64_t* __restrict__ edge_type, float* __restrict__ edge_weight) {
  const int64_t base = blockIdx.x * (int64_t)SC_ITEMS;
  int32_t w[4];
  uint32_t n = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = base + 4 * threadIdx.x + j;
    w[j] = i < L ? winner_at[i] : -1;
    n += w[j] >= 0;
  }
  __shared__ uint32_t sm[256];
  sm[threadIdx.x] = n;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t t = threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  int64_t pos = (int64_t)bases[blockIdx.x] + sm[threadIdx.x] - n;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (w[j] < 0) continue;
    Cand c;
    make_cand(w[j], N, ks, km, Ks1, Km1, sidx, sdist, midx, msim, thr, &c);
    if (pos < U) {
      const int64_t e = 2 * pos;
      edge_index[e] = c.src; edge_index[2 * U + e] = c.tgt;
      edge_index[e + 1] = c.tgt; edge_index[2 * U + e + 1] = c.src;
      edge_type[e] = c.type; edge_type[e + 1] = c.type;
      if (edge_weight) { edge_weight[e] = c.w; edge_weight[e + 1] = c.w; }
      for (int d = 0; d < edge_dim; ++d) {
        const float v = d == 0 ? c.f0 : (d == 1 ? c.f1 : 0.f);
        edge_attr[e * edge_dim + d] = v;
        edge_attr[(e + 1) * edge_dim + d] = v;
      }
    }
    ++pos;
  }
}

int kmax_of(int K) { return K <= 9 ? 9 : (K <= 17 ? 17 : (K <= 33 ? 33 : 0)); }

inline int64_t table_capacity(int64_t L) {
  int64_t cap = 1024;
  while (cap < 2 * L) cap <<= 1;
  return cap;
}

struct DedupLayout { size_t keys, best, first, winner, counts, bases, total, bytes; int64_t cap; int nb; };

inline DedupLayout dedup_layout(int64_t L) {
  DedupLayout l;
  l.cap = table_capacity(L);
  l.nb = (int)((L + SC_ITEMS - 1) / SC_ITEMS);
  size_t off = 0;
  l.keys = off; off += (size_t)l.cap * 8;
  l.best = off; off += (size_t)l.cap * 8;
  l.first = off; off += dgdm_align_up((size_t)l.cap * 4, 16);
  l.winner = off; off += dgdm_align_up((size_t)L * 4, 16);
  l.counts = off; off += dgdm_align_up((size_t)l.nb * 4, 16);
  l.bases = off; off += dgdm_align_up((size_t)l.nb * 4, 16);
  l.total = off; off += 16;
  l.bytes = off;
  return l;
}

}  // namespace

extern "C" int dgdm_knn2d(const float* coords, int32_t N, int32_t K, int32_t* idx, float* dist, void* stream) {
  if (N < 0 || K < 1) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_OK;
  if (!coords || !idx || !dist) return DGDM_ERR_INVALID_ARG;
  if (K > N) return DGDM_ERR_INVALID_ARG;
  const int km = kmax_of(K);
  if (!km || (reinterpret_cast<uintptr_t>(coords) & 7u)) return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((N + 127) / 128);
  if (km == 9) hipLaunchKernelGGL(k_knn2d<9>, grid, dim3(128), 0, s, coords, N, K, idx, dist);
  else if (km == 17) hipLaunchKernelGGL(k_knn2d<17>, grid, dim3(128), 0, s, coords, N, K, idx, dist);
  else hipLaunchKernelGGL(k_knn2d<33>, grid, dim3(128), 0, s, coords, N, K, idx, dist);
  return dgdm_launch_status();
}

extern "C" int dgdm_row_sqnorm(const float* X, int64_t ldx, int32_t N, int32_t F, float* sq, void* stream) {
  if (N < 0 || F <= 0) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_OK;
  if (!X || !sq) return DGDM_ERR_INVALID_ARG;
  if ((F & 3) || (ldx & 3) || ldx < F || !dgdm_aligned16(X)) return DGDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_row_sqnorm, dim3((N + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), X, ldx, N, F, sq);
  return dgdm_launch_status();
}

constexpr int GRAM_SEGMENTS = 16;

extern "C" size_t dgdm_knn_gram_workspace_bytes(int32_t B, int32_t K) {
  const int km = kmax_of(K);
  if (B <= 0 || !km) return 0;
  return (size_t)GRAM_SEGMENTS * B * km * sizeof(KnnRec);
}

extern "C" int dgdm_knn_gram(const float* GT, int64_t ldg, const float* sq, int32_t N, int32_t q0, int32_t B, int32_t K, int32_t* idx,
                             float* sim, void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0 || B < 0 || q0 < 0 || K < 1 || q0 + B > N) return DGDM_ERR_INVALID_ARG;
  if (B == 0) return DGDM_OK;
  if (!GT || !sq || !idx || !sim || !workspace || ldg < B) return DGDM_ERR_INVALID_ARG;
  if (K > N) return DGDM_ERR_INVALID_ARG;
  const int km = kmax_of(K);
  if (!km) return DGDM_ERR_UNSUPPORTED;
  if (workspace_bytes < dgdm_knn_gram_workspace_bytes(B, K)) return DGDM_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  KnnRec* partial = static_cast<KnnRec*>(workspace);
  const int per = (N + GRAM_SEGMENTS - 1) / GRAM_SEGMENTS;
  const int nseg = (N + per - 1) / per;
  const dim3 g1((B + 127) / 128, nseg), g2((B + 127) / 128);
#define DGDM_KNN_LAUNCH(KM)                                                                                             \
  hipLaunchKernelGGL(k_knn_gram_partial<KM>, g1, dim3(128), 0, s, GT, ldg, sq, N, q0, B, per, partial);                 \
  hipLaunchKernelGGL(k_knn_gram_merge<KM>, g2, dim3(128), 0, s, partial, nseg, q0, B, K, sq, idx, sim);
  if (km == 9) { DGDM_KNN_LAUNCH(9) } else if (km == 17) { DGDM_KNN_LAUNCH(17) } else { DGDM_KNN_LAUNCH(33) }
#undef DGDM_KNN_LAUNCH
  return dgdm_launch_status();
}

extern "C" int dgdm_pair_cosine(const float* X, int64_t ldx, const float* sq, const int32_t* idx, int32_t N, int32_t K, int32_t F,
                                float* sim, void* stream) {
  if (N < 0 || K < 1 || F <= 0) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_OK;
  if (!X || !sq || !idx || !sim) return DGDM_ERR_INVALID_ARG;
  if ((F & 3) || (ldx & 3) || ldx < F || !dgdm_aligned16(X)) return DGDM_ERR_UNSUPPORTED;
  const int64_t pairs = (int64_t)N * K;
  hipLaunchKernelGGL(k_pair_cosine, dim3((unsigned)((pairs + 15) / 16)), dim3(256), 0, static_cast<hipStream_t>(stream), X, ldx, sq, idx,
                     pairs, K, F, sim);
  return dgdm_launch_status();
}

extern "C" size_t dgdm_edge_dedup_workspace_bytes(int32_t N, int32_t Ks1, int32_t Km1) {
  if (N <= 0 || Ks1 < 1 || Km1 < 1) return 0;
  const int64_t L = (int64_t)N * ((Ks1 - 1) + (Km1 - 1));
  return L > 0 ? dedup_layout(L).bytes : 16;
}

// Phase 1: candidates -> table -> winners -> *n_edges (device int64) = number of kept undirected edges.
extern "C" int dgdm_edge_dedup_count(const int32_t* sidx, const float* sdist, int32_t Ks1, const int32_t* midx, const float* msim,
                                     int32_t Km1, int32_t N, float threshold, void* workspace, size_t workspace_bytes, int64_t* n_edges,
                                     void* stream) {
  if (N < 0 || Ks1 < 1 || Km1 < 1 || !n_edges) return DGDM_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int ks = Ks1 - 1, km = Km1 - 1;
  const int64_t L = (int64_t)N * (ks + km);
  if (L == 0) { (void)hipMemsetAsync(n_edges, 0, sizeof(int64_t), s); return dgdm_launch_status(); }
  if (!sidx || !sdist || !midx || !msim || !workspace) return DGDM_ERR_INVALID_ARG;
  if (L >= 0x7fffffffLL) return DGDM_ERR_UNSUPPORTED;
  const DedupLayout l = dedup_layout(L);
  if (workspace_bytes < l.bytes) return DGDM_ERR_WORKSPACE;
  char* w = static_cast<char*>(workspace);
  Table t{reinterpret_cast<unsigned long long*>(w + l.keys), reinterpret_cast<unsigned long long*>(w + l.best),
          reinterpret_cast<uint32_t*>(w + l.first), (uint64_t)(l.cap - 1)};
  int32_t* winner_at = reinterpret_cast<int32_t*>(w + l.winner);
  uint32_t* counts = reinterpret_cast<uint32_t*>(w + l.counts);
  uint32_t* bases = reinterpret_cast<uint32_t*>(w + l.bases);
  const int64_t init_n = l.cap > L ? l.cap : L;
  hipLaunchKernelGGL(k_table_init, dim3((unsigned)((init_n + 255) / 256)), dim3(256), 0, s, t, l.cap, winner_at, L);
  const dim3 gl((unsigned)((L + 255) / 256));
  hipLaunchKernelGGL(k_table_insert, gl, dim3(256), 0, s, t, L, N, ks, km, Ks1, Km1, sidx, sdist, midx, msim, threshold);
  hipLaunchKernelGGL(k_table_winners, gl, dim3(256), 0, s, t, L, N, ks, km, Ks1, Km1, sidx, sdist, midx, msim, threshold, winner_at);
  hipLaunchKernelGGL(k_flag_count, dim3(l.nb), dim3(256), 0, s, winner_at, L, counts);
  hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(256), 0, s, counts, l.nb, bases, n_edges);
  return dgdm_launch_status();
}

// Phase 2 (same workspace, after the caller has read *n_edges = U and allocated the outputs):
// edge_index int64 [2, 2U], edge_attr float [2U, edge_dim], edge_type int64 [2U], edge_weight float [2U] (nullable).
extern "C" int dgdm_edge_emit(const int32_t* sidx, const float* sdist, int32_t Ks1, const int32_t* midx, const float* msim, int32_t Km1,
                              int32_t N, float threshold, const void* workspace, int64_t U, int32_t edge_dim, int64_t* edge_index,
                              float* edge_attr, int64_t* edge_type, float* edge_weight, void* stream) {
  if (N < 0 || Ks1 < 1 || Km1 < 1 || U < 0 || edge_dim < 2) return DGDM_ERR_INVALID_ARG;
  if (U == 0) return DGDM_OK;
  if (!sidx || !sdist || !midx || !msim || !workspace || !edge_index || !edge_attr || !edge_type) return DGDM_ERR_INVALID_ARG;
  const int ks = Ks1 - 1, km = Km1 - 1;
  const int64_t L = (int64_t)N * (ks + km);
  const DedupLayout l = dedup_layout(L);
  const char* w = static_cast<const char*>(workspace);
  hipLaunchKernelGGL(k_edge_emit, dim3(l.nb), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const int32_t*>(w + l.winner), L, reinterpret_cast<const uint32_t*>(w + l.bases), N, ks, km, Ks1, Km1,
                     sidx, sdist, midx, msim, threshold, U, edge_dim, edge_index, edge_attr, edge_type, edge_weight);
  return dgdm_launch_status();
}
