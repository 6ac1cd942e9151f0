// K11: tissue-graph edge construction on the GPU -- the step immediately upstream of the model
// (reference: TissueGraphBuilder._create_edges / _remove_duplicate_edges / the edge part of
// _to_pytorch_geometric, preprocessing/tissue_graph_builder.py:269-414, which run scikit-learn
// NearestNeighbors + cosine_similarity and Python loops on the CPU).
//
//   dgdm_knn2d            K nearest points (self included) of every 2-D coordinate, brute force, one wave per
//                         query with the running top-K spread over its lanes, ascending by (distance, index);
//                         squared distances are formed as fl(fl(dx*dx) + fl(dy*dy)) (no FMA contraction), so a
//                         float32 CPU restatement reproduces indices and distances bit for bit
//   dgdm_row_sqnorm       ||x_i||^2
//   dgdm_knn_gram         K nearest rows in feature space from Gram rows G[q][j] = x_q . x_j (produced by the
//                         GEMM kernels): d^2 = |x_q|^2 + |x_j|^2 - 2 G, same wave-per-query selection
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

// Wave-cooperative top-K: one wave per query.  The current K best (d2, index, g) live one per lane (lane i = i-th
// best, lanes >= K hold +inf), so the threshold is a readlane away.  Each iteration the 64 lanes score 64
// consecutive candidates; a ballot finds the few that beat the threshold (K ln(N/K) of N over the whole scan),
// and those are inserted one at a time in index order: position = number of kept entries <= the candidate
// (existing equals stay in front: ties keep ascending index), the tail shifts one lane up.
struct TopK {
  float d;
  int i;
  float g;
  __device__ __forceinline__ void init() { d = __builtin_inff(); i = -1; g = 0.f; }
  __device__ __forceinline__ float threshold(int K) const { return __shfl(d, K - 1, 64); }
  // all lanes call; (cd, ci, cg) are wave-uniform
  __device__ __forceinline__ void insert(float cd, int ci, float cg, int lane) {
    const unsigned long long le = __ballot(d <= cd);
    const int p = __popcll(le);                       // kept entries that stay in front (the list is sorted: a prefix)
    const float ud = __shfl_up(d, 1, 64), ug = __shfl_up(g, 1, 64);
    const int ui = __shfl_up(i, 1, 64);
    if (lane == p) { d = cd; i = ci; g = cg; }
    else if (lane > p) { d = ud; i = ui; g = ug; }
  }
  __device__ __forceinline__ void offer(float cd, int ci, float cg, int K, int lane) {
    float thr = threshold(K);
    unsigned long long m = __ballot(cd < thr);
    while (m) {                                        // wave-uniform loop over the candidates that qualify, lowest index first
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const float sd = __shfl(cd, src, 64);
      if (sd < thr) {                                  // the threshold may have dropped since the ballot
        insert(sd, __shfl(ci, src, 64), __shfl(cg, src, 64), lane);
        thr = threshold(K);
      }
    }
  }
};

__global__ __launch_bounds__(256) void k_knn2d(const float* __restrict__ coords, int N, int K, int32_t* __restrict__ idx,
                                                float* __restrict__ dist) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= N) return;                                  // whole wave
  const float2 me = reinterpret_cast<const float2*>(coords)[q];
  TopK best;
  best.init();
  for (int j0 = 0; j0 < N; j0 += 64) {
    const int j = j0 + lane;
    float d2 = __builtin_inff();
    if (j < N) {
      const float2 c = reinterpret_cast<const float2*>(coords)[j];
      const float dx = __fsub_rn(me.x, c.x), dy = __fsub_rn(me.y, c.y);
      d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
    }
    best.offer(d2, j, 0.f, K, lane);
  }
  if (lane < K) { idx[(int64_t)q * K + lane] = best.i; dist[(int64_t)q * K + lane] = sqrtf(best.d); }
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

// feature-space kNN of query q0 + q from its Gram row G[q][j] = x_(q0+q) . x_j (row-major, contiguous over j)
__global__ __launch_bounds__(256) void k_knn_gram(const float* __restrict__ G, int64_t ldg, const float* __restrict__ sq, int N, int q0,
                                                   int B, int K, int32_t* __restrict__ idx, float* __restrict__ sim) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= B) return;
  const float sqq = sq[q0 + q];
  const float* row = G + (int64_t)q * ldg;
  TopK best;
  best.init();
  for (int j0 = 0; j0 < N; j0 += 64) {
    const int j = j0 + lane;
    float d2 = __builtin_inff(), g = 0.f;
    if (j < N) {
      g = row[j];
      d2 = fmaxf(sqq + sq[j] - 2.f * g, 0.f);
      if (j == q0 + q) d2 = 0.f;                         // the point itself
    }
    best.offer(d2, j, g, K, lane);
  }
  if (lane < K) {
    const int j = best.i;
    const float nq = sqrtf(sqq), nj = j >= 0 ? sqrtf(sq[j]) : 0.f;
    idx[(int64_t)(q0 + q) * K + lane] = j;
    sim[(int64_t)(q0 + q) * K + lane] = (nq > 0.f && nj > 0.f) ? best.g / (nq * nj) : 0.f;
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
  if (K > 64 || (reinterpret_cast<uintptr_t>(coords) & 7u)) return DGDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_knn2d, dim3((N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), coords, N, K, idx, dist);
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

extern "C" size_t dgdm_knn_gram_workspace_bytes(int32_t B, int32_t K) {
  (void)B; (void)K;
  return 0;   // the wave-cooperative selection needs no scratch (kept in the ABI for callers sized against it)
}

extern "C" int dgdm_knn_gram(const float* G, int64_t ldg, const float* sq, int32_t N, int32_t q0, int32_t B, int32_t K, int32_t* idx,
                             float* sim, void* workspace, size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes;
  if (N < 0 || B < 0 || q0 < 0 || K < 1 || (int64_t)q0 + B > N) return DGDM_ERR_INVALID_ARG;
  if (B == 0) return DGDM_OK;
  if (!G || !sq || !idx || !sim || ldg < N) return DGDM_ERR_INVALID_ARG;
  if (K > N) return DGDM_ERR_INVALID_ARG;
  if (K > 64) return DGDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_knn_gram, dim3((B + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), G, ldg, sq, N, q0, B, K, idx, sim);
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
  const int64_t per = ((int64_t)Ks1 - 1) + ((int64_t)Km1 - 1);     // candidate pairs per node
  if (per > ((int64_t)1 << 24)) return 0;                           // not a neighbour table any more (and N * per must fit 64 bits)
  const int64_t L = (int64_t)N * per;
  return L > 0 ? dedup_layout(L).bytes : 16;
}

// Phase 1: candidates -> table -> winners -> *n_edges (device int64) = number of kept undirected edges.
extern "C" int dgdm_edge_dedup_count(const int32_t* sidx, const float* sdist, int32_t Ks1, const int32_t* midx, const float* msim,
                                     int32_t Km1, int32_t N, float threshold, void* workspace, size_t workspace_bytes, int64_t* n_edges,
                                     void* stream) {
  if (N < 0 || Ks1 < 1 || Km1 < 1 || !n_edges) return DGDM_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int ks = Ks1 - 1, km = Km1 - 1;
  if ((int64_t)ks + km > ((int64_t)1 << 24)) return DGDM_ERR_UNSUPPORTED;      // N * (ks + km) must fit 64 bits
  const int64_t L = (int64_t)N * ((int64_t)ks + km);
  if (L == 0) { dgdm_fill_async(n_edges, 0, sizeof(int64_t), s); return dgdm_launch_status(); }
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
  if ((int64_t)ks + km > ((int64_t)1 << 24)) return DGDM_ERR_UNSUPPORTED;
  const int64_t L = (int64_t)N * ((int64_t)ks + km);
  if (L >= 0x7fffffffLL) return DGDM_ERR_UNSUPPORTED;                          // as dgdm_edge_dedup_count refuses it
  const DedupLayout l = dedup_layout(L);
  const char* w = static_cast<const char*>(workspace);
  hipLaunchKernelGGL(k_edge_emit, dim3(l.nb), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const int32_t*>(w + l.winner), L, reinterpret_cast<const uint32_t*>(w + l.bases), N, ks, km, Ks1, Km1,
                     sidx, sdist, midx, msim, threshold, U, edge_dim, edge_index, edge_attr, edge_type, edge_weight);
  return dgdm_launch_status();
}
