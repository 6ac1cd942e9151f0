// K1: edge list -> CSR (stable by edge id), GCN normalisation.  Integer work, bit-exact vs
// oracle/csr_oracle.py.  Replaces core/graph_layers.py:76-84 index preparation.
//
// Pipeline (all on the caller's stream, no host sync):
//   memset counts -> count keys (int atomics: order-free, exact) -> exclusive scan (+1 per row
//   when loops are appended) -> unordered fill of a scratch CSR (atomic cursor) -> rank pass:
//   every scratch entry counts the entries of its row with a smaller edge id and is written to
//   that rank, which yields the stable (ascending edge id) order deterministically.
// The self-loop entry of row i has the largest edge id of the row (E+i), so it is written
// straight to the last slot and takes no part in the rank pass.
#include "common.hpp"

namespace {

constexpr int SCAN_BLOCK = 1024;  // threads; one item per thread

// an edge takes part only if both endpoints are in [0, N): out-of-range ids mark dropped edges
// (the sync-free top-k pooling keeps its edge arrays un-compacted) and can never index memory.
__device__ __forceinline__ bool edge_ok(int64_t k, int64_t v, int32_t N) {
  return (uint64_t)k < (uint64_t)N && (uint64_t)v < (uint64_t)N;
}

__global__ void k_count(const int64_t* __restrict__ keys, const int64_t* __restrict__ vals, int64_t E, int32_t N,
                        int32_t* __restrict__ cnt) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < E; i += stride)
    if (edge_ok(keys[i], vals[i], N)) atomicAdd(&cnt[keys[i]], 1);
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
  // 1024 threads = 16 waves; wave-level inclusive scan then scan of wave totals in LDS
  __shared__ int wsum[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  if (wid == 0) {
    int s = lane < 16 ? wsum[lane] : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      int t = __shfl_up(s, o, 64);
      if (lane >= o) s += t;
    }
    if (lane < 16) wsum[lane] = s;  // inclusive over waves
  }
  __syncthreads();
  const int base = wid ? wsum[wid - 1] : 0;
  *total = wsum[15];
  __syncthreads();
  return base + inc - v;
}

// pass 1: per-block totals of (cnt[i] + extra)
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_block_totals(const int32_t* __restrict__ cnt, int32_t N, int extra,
                                                                  int32_t* __restrict__ block_tot) {
  const int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  int v = i < N ? cnt[i] + extra : 0, tot;
  block_exclusive_scan(v, &tot);
  if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}

// pass 2: one block scans the block totals in place (exclusive), chunk by chunk
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_totals(int32_t* __restrict__ block_tot, int nblocks) {
  int carry = 0;
  for (int base = 0; base < nblocks; base += SCAN_BLOCK) {
    const int i = base + threadIdx.x;
    int v = i < nblocks ? block_tot[i] : 0, tot;
    int ex = block_exclusive_scan(v, &tot);
    if (i < nblocks) block_tot[i] = carry + ex;
    carry += tot;
  }
}

// pass 3: rowptr[i] = block offset + in-block exclusive scan; rowptr[N] = total
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_write(const int32_t* __restrict__ cnt, int32_t N, int extra,
                                                           const int32_t* __restrict__ block_tot, int32_t* __restrict__ rowptr) {
  const int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  int v = i < N ? cnt[i] + extra : 0, tot;
  int ex = block_exclusive_scan(v, &tot) + block_tot[blockIdx.x];
  if (i < N) rowptr[i] = ex;
  if (i == N - 1) rowptr[N] = ex + v;
}

__global__ void k_fill(const int64_t* __restrict__ keys, const int64_t* __restrict__ vals, int64_t E, int32_t N,
                       const int32_t* __restrict__ rowptr, int32_t* __restrict__ cursor,
                       int32_t* __restrict__ tcol, int32_t* __restrict__ teid) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; e < E; e += stride) {
    if (!edge_ok(keys[e], vals[e], N)) continue;
    const int32_t k = (int32_t)keys[e];
    const int32_t slot = rowptr[k] + atomicAdd(&cursor[k], 1);
    tcol[slot] = (int32_t)vals[e];
    teid[slot] = (int32_t)e;
  }
}

// one thread per row: places the row's loop entry (if any) and ranks the scratch entries.
// rows are short on tissue graphs (kNN, degree ~ 5-16); a long row costs O(deg^2) reads from L2.
__global__ void k_rank(const int32_t* __restrict__ rowptr, int32_t N, int32_t E32, int add_loops,
                       const int32_t* __restrict__ tcol, const int32_t* __restrict__ teid,
                       int32_t* __restrict__ col, int32_t* __restrict__ eid) {
  // thread per scratch entry would need the entry's row; instead a wave cooperates on rows:
  // lane l of a wave takes entries l, l+64, ... of the row and counts smaller ids.
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < N; r += nwaves) {
    const int s = rowptr[r];
    const int n = rowptr[r + 1] - s - (add_loops ? 1 : 0);  // non-loop entries
    for (int i = lane; i < n; i += 64) {
      const int my = teid[s + i];
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += (teid[s + j] < my) ? 1 : 0;
      col[s + rank] = tcol[s + i];
      eid[s + rank] = my;
    }
    if (add_loops && lane == 0) {
      col[s + n] = r;
      eid[s + n] = E32 + r;
    }
  }
}

__global__ void k_dinv(const int32_t* __restrict__ rowptr, int32_t N, float* __restrict__ dinv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    const float deg = (float)(rowptr[i + 1] - rowptr[i]);
    dinv[i] = deg > 0.f ? 1.0f / sqrtf(deg) : 0.f;  // deg^-1/2, inf -> 0 (graph_layers.py:82-83)
  }
}

__global__ void k_edge_weights(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                               const float* __restrict__ dinv, int32_t N, float* __restrict__ w) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < N; r += nwaves) {
    const int s = rowptr[r], e = rowptr[r + 1];
    const float dr = dinv[r];
    for (int p = s + lane; p < e; p += 64) w[p] = dinv[col[p]] * dr;
  }
}

// ---- both orientations of one edge list in one pipeline (5 launches instead of 17) ------------------
// o = 0: rows are destinations (forward), o = 1: rows are sources (backward).  Same results, bit for bit, as two
// dgdm_csr_build calls + dgdm_gcn_dinv + two dgdm_csr_edge_weights calls.
struct PairArrays {
  int32_t *cnt[2], *cursor[2], *tcol[2], *teid[2];
  int32_t *rowptr[2], *col[2], *eid[2];
  float* w[2];
  int32_t* status;      // one word in the workspace (zeroed with the counters): bit 0 = a scatter slot fell outside its row,
                        // bit 1 = a row's extent fell outside the arrays.  The offending write is SKIPPED: stale or corrupted
                        // counters (the hipMemsetAsync-in-graph bug of round 1) become a readable flag, not a memory fault.
  int32_t n_entries;
  int32_t* long_table[2];   // nullable: per orientation the list of rows longer than DGDM_SPMM_LONG_ROW (include/dgdm_hip.h)
  int32_t long_item_cap;
};

__global__ void k_count_pair(const int64_t* __restrict__ ei, int64_t E, int32_t N, int32_t* __restrict__ cnt_dst,
                             int32_t* __restrict__ cnt_src) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; e < E; e += stride) {
    const int64_t s = ei[e], d = ei[E + e];
    if (edge_ok(s, d, N)) {
      atomicAdd(&cnt_dst[d], 1);
      atomicAdd(&cnt_src[s], 1);
    }
  }
}

// Exclusive scan of the row counts (+1 per row with loops), orientation blockIdx.y, without a second launch or a
// look-back chain: block b owns rows [b*4096, (b+1)*4096), first adds up every count in front of its chunk (the whole
// array is a few hundred KB in L2), then scans its own chunk.  Block (b, 0) also writes dinv from the in-degrees it is
// looking at (k_dinv's formula).
constexpr int SCAN_CHUNK = 4 * SCAN_BLOCK;
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_pair(PairArrays a, int32_t N, int extra, float* __restrict__ dinv) {
  const int o = blockIdx.y;
  const int32_t* __restrict__ cnt = a.cnt[o];
  int32_t* __restrict__ rowptr = a.rowptr[o];
  const int base = blockIdx.x * SCAN_CHUNK;
  int before = 0;
  for (int i = 4 * threadIdx.x; i < base; i += SCAN_CHUNK) {   // base is a multiple of 4: whole int4s (arrays are 256-B aligned)
    const int4 c = *reinterpret_cast<const int4*>(cnt + i);
    before += (c.x + c.y) + (c.z + c.w);
  }
  int tot;
  block_exclusive_scan(before, &tot);
  const int carry = tot + extra * base;
  const int i = base + 4 * threadIdx.x;
  int v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = i + j < N ? cnt[i + j] + extra : 0;
  int ex = carry + block_exclusive_scan(v[0] + v[1] + v[2] + v[3], &tot);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (i + j < N) {
      rowptr[i + j] = ex;
      if (o == 0) dinv[i + j] = v[j] > 0 ? 1.0f / sqrtf((float)v[j]) : 0.f;
    }
    ex += v[j];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) rowptr[N] = carry + tot;
  if (blockIdx.x == 0 && a.long_table[o]) {     // header (count, slots) and the arrival counters behind the items start at zero
    int32_t* t = a.long_table[o];
    for (int i = threadIdx.x; i < 2; i += SCAN_BLOCK) t[i] = 0;
    for (int i = threadIdx.x; i < a.long_item_cap; i += SCAN_BLOCK) t[2 + 2 * a.long_item_cap + i] = 0;
  }
}

__global__ void k_fill_pair(const int64_t* __restrict__ ei, int64_t E, int32_t N, PairArrays a) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; e < E; e += stride) {
    const int64_t s64 = ei[e], d64 = ei[E + e];
    if (!edge_ok(s64, d64, N)) continue;
    const int32_t s = (int32_t)s64, d = (int32_t)d64;
    const int32_t cd = atomicAdd(&a.cursor[0][d], 1), cs = atomicAdd(&a.cursor[1][s], 1);
    const int32_t slot_d = a.rowptr[0][d] + cd, slot_s = a.rowptr[1][s] + cs;
    // a slot is valid while the cursor stays below the row's count AND inside the scratch arrays
    const bool ok_d = cd >= 0 && cd < a.cnt[0][d] && slot_d >= 0 && slot_d < a.n_entries;
    const bool ok_s = cs >= 0 && cs < a.cnt[1][s] && slot_s >= 0 && slot_s < a.n_entries;
    if (ok_d) { a.tcol[0][slot_d] = s; a.teid[0][slot_d] = (int32_t)e; }
    if (ok_s) { a.tcol[1][slot_s] = d; a.teid[1][slot_s] = (int32_t)e; }
    if (!(ok_d && ok_s)) atomicOr(a.status, 1);
  }
}

// k_rank for both orientations (blockIdx.y) + the GCN weight of every entry, w[p] = dinv[col[p]] * dinv[row]
__global__ void k_rank_pair(PairArrays a, int32_t N, int32_t E32, int add_loops, const float* __restrict__ dinv) {
  const int o = blockIdx.y;
  const int32_t* __restrict__ rowptr = a.rowptr[o];
  const int32_t* __restrict__ tcol = a.tcol[o];
  const int32_t* __restrict__ teid = a.teid[o];
  int32_t* __restrict__ col = a.col[o];
  int32_t* __restrict__ eid = a.eid[o];
  float* __restrict__ w = a.w[o];
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < N; r += nwaves) {
    const int s = rowptr[r];
    const int n = rowptr[r + 1] - s - (add_loops ? 1 : 0);
    if (s < 0 || n < 0 || (int64_t)s + n + (add_loops ? 1 : 0) > a.n_entries) {   // corrupted offsets: flag, leave the row alone
      if (lane == 0) atomicOr(a.status, 2);
      continue;
    }
    const float dr = dinv[r];
    if (a.long_table[o] && n + (add_loops ? 1 : 0) > DGDM_SPMM_LONG_ROW) {
      // a row one wave should not walk alone: dgdm_spmm* splits it into segments (csrc/spmm.hip), and its entries are ordered by
      // k_rank_long (this loop is quadratic in the row length: 17 ms for one row of 5000 entries)
      if (lane == 0) {
        int32_t* t = a.long_table[o];
        const int idx = atomicAdd(&t[0], 1);
        const int nseg = (n + (add_loops ? 1 : 0) + DGDM_SPMM_SEGMENT - 1) / DGDM_SPMM_SEGMENT;
        const int slot0 = atomicAdd(&t[1], nseg);
        if (idx < a.long_item_cap) { t[2 + 2 * idx] = r; t[2 + 2 * idx + 1] = slot0; }
        if (add_loops) { col[s + n] = r; eid[s + n] = E32 + r; w[s + n] = dr * dr; }
      }
      continue;
    }
    for (int i = lane; i < n; i += 64) {
      const int my = teid[s + i];
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += (teid[s + j] < my) ? 1 : 0;
      const int c = tcol[s + i];
      col[s + rank] = c;
      eid[s + rank] = my;
      w[s + rank] = ((unsigned)c < (unsigned)N ? dinv[c] : 0.f) * dr;
    }
    if (add_loops && lane == 0) {
      col[s + n] = r;
      eid[s + n] = E32 + r;
      w[s + n] = dr * dr;
    }
  }
}

// Ordering of the LONG rows' entries (rank by edge id inside the row): segment `slot` of the table = 64 entries of one row, one
// entry per lane, which counts the row's entries with a smaller edge id -- 64 at a time out of a register (v_readlane), so a row
// of n entries costs n / 64 waves x n comparisons running side by side instead of one wave x n^2 / 64.
static_assert(DGDM_SPMM_SEGMENT == 64, "k_rank_long hands one table segment to one wavefront");
__global__ void k_rank_long(PairArrays a, int32_t N, const float* __restrict__ dinv, int add_loops) {
  const int o = blockIdx.y;
  const int32_t* __restrict__ t = a.long_table[o];
  const int count = min(t[0], a.long_item_cap), slots = t[1];
  const int slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (slot >= slots) return;
  const int32_t* __restrict__ rowptr = a.rowptr[o];
  int row = -1, slot0 = 0;
  for (int i = 0; i < count; ++i) {
    const int r_ = t[2 + 2 * i], s0 = t[2 + 2 * i + 1];
    const int ns = (rowptr[r_ + 1] - rowptr[r_] + DGDM_SPMM_SEGMENT - 1) / DGDM_SPMM_SEGMENT;
    if (slot >= s0 && slot < s0 + ns) { row = r_; slot0 = s0; }
  }
  if (row < 0) return;
  const int s = rowptr[row];
  const int n = rowptr[row + 1] - s - (add_loops ? 1 : 0);
  if (s < 0 || n < 0 || (int64_t)s + n + (add_loops ? 1 : 0) > a.n_entries) return;      // k_rank_pair flagged it
  const int32_t* __restrict__ teid = a.teid[o];
  const int i = (slot - slot0) * 64 + lane;
  const int my = i < n ? teid[s + i] : 0x7fffffff;
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int v = j0 + lane < n ? teid[s + j0 + lane] : 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 64; ++k) rank += (__builtin_amdgcn_readlane(v, k) < my) ? 1 : 0;
  }
  if (i < n) {
    const int c = a.tcol[o][s + i];
    a.col[o][s + rank] = c;
    a.eid[o][s + rank] = my;
    a.w[o][s + rank] = ((unsigned)c < (unsigned)N ? dinv[c] : 0.f) * dinv[row];
  }
}

struct PairWorkspace {
  int32_t *cnt[2], *cursor[2], *tcol[2], *teid[2], *block_tot[2], *status;
  size_t counters_bytes, status_offset, bytes;
};

PairWorkspace carve_pair(void* base, int64_t E, int32_t N, int32_t add_loops) {
  PairWorkspace w;
  const size_t n_entries = (size_t)E + (add_loops ? (size_t)N : 0);
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t count) {
    int32_t* r = p ? reinterpret_cast<int32_t*>(p + off) : nullptr;      // size queries carve a null base: no arithmetic on it
    off += dgdm_align_up(count * sizeof(int32_t), 256);
    return r;
  };
  w.cnt[0] = take(N); w.cnt[1] = take(N); w.cursor[0] = take(N); w.cursor[1] = take(N);
  w.status_offset = off;
  w.status = take(1);
  w.counters_bytes = off;  // the four counter arrays and the status word are adjacent: one fill
  w.tcol[0] = take(n_entries); w.tcol[1] = take(n_entries); w.teid[0] = take(n_entries); w.teid[1] = take(n_entries);
  const size_t nblk = ((size_t)N + SCAN_BLOCK - 1) / SCAN_BLOCK + 1;
  w.block_tot[0] = take(nblk); w.block_tot[1] = take(nblk);
  w.bytes = off;
  return w;
}

struct Workspace {
  int32_t *cnt, *cursor, *block_tot, *tcol, *teid;
  size_t bytes;
};

Workspace carve(void* base, int64_t E, int32_t N, int32_t add_loops) {
  Workspace w;
  const size_t n_entries = (size_t)E + (add_loops ? (size_t)N : 0);
  const size_t nblk = ((size_t)N + SCAN_BLOCK - 1) / SCAN_BLOCK + 1;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t count) {
    int32_t* r = p ? reinterpret_cast<int32_t*>(p + off) : nullptr;      // size queries carve a null base: no arithmetic on it
    off += dgdm_align_up(count * sizeof(int32_t), 256);
    return r;
  };
  w.cnt = take(N);
  w.cursor = take(N);
  w.block_tot = take(nblk);
  w.tcol = take(n_entries);
  w.teid = take(n_entries);
  w.bytes = off;
  return w;
}


// The index set of the SAME edge list over MORE nodes (add_loops = 1): every edge id is < n_old, so rows < n_old keep their
// entries (edges in ascending id, the loop last) and their degrees; a node in [n_old, n_new) has its self loop only (degree 1,
// dinv 1, weight 1).  The extended arrays are therefore the old ones with (n_new - n_old) one-entry rows appended -- a copy,
// not a build.  One orientation per blockIdx.y; blockIdx.y == 2 pads the aggregated edge attributes with zero rows.
struct ExtendArrays {
  const int32_t *rowptr[2], *col[2], *eid[2];
  const float *w[2], *dinv, *ea;
  int32_t *rowptr_o[2], *col_o[2], *eid_o[2];
  float *w_o[2], *dinv_o, *ea_o;
  int32_t n_old, n_new, E, ea_dim;
};
__global__ __launch_bounds__(256) void k_csr_extend(const ExtendArrays a) {
  const int o = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (o == 2) {
    const int64_t keep = (int64_t)a.n_old * a.ea_dim, total = (int64_t)a.n_new * a.ea_dim;
    for (int64_t i = t0; i < total; i += stride) a.ea_o[i] = i < keep ? a.ea[i] : 0.f;
    return;
  }
  const int32_t end = a.rowptr[o][a.n_old];        // entries in use (the arrays may be longer: edges marked -1 take no entry)
  for (int64_t i = t0; i <= a.n_new; i += stride) a.rowptr_o[o][i] = i <= a.n_old ? a.rowptr[o][i] : end + (int32_t)(i - a.n_old);
  for (int64_t p = t0; p < end; p += stride) {
    a.col_o[o][p] = a.col[o][p];
    a.eid_o[o][p] = a.eid[o][p];
    a.w_o[o][p] = a.w[o][p];
  }
  for (int64_t i = a.n_old + t0; i < a.n_new; i += stride) {
    const int64_t p = end + (i - a.n_old);
    a.col_o[o][p] = (int32_t)i;
    a.eid_o[o][p] = a.E + (int32_t)i;
    a.w_o[o][p] = 1.0f;
  }
  if (o == 0)
    for (int64_t i = t0; i < a.n_new; i += stride) a.dinv_o[i] = i < a.n_old ? a.dinv[i] : 1.0f;
}

}  // namespace

extern "C" int32_t dgdm_spmm_long_item_cap(int64_t n_entries) { return (int32_t)(n_entries / DGDM_SPMM_LONG_ROW + 1); }
extern "C" int32_t dgdm_spmm_long_slot_cap(int64_t n_entries) {
  return (int32_t)(n_entries / DGDM_SPMM_SEGMENT + n_entries / DGDM_SPMM_LONG_ROW + 2);
}
extern "C" size_t dgdm_spmm_long_table_words(int64_t n_entries) { return 2 + 3 * (size_t)dgdm_spmm_long_item_cap(n_entries); }

extern "C" size_t dgdm_csr_build_workspace_bytes(int64_t E, int32_t N, int32_t add_loops) {
  if (E < 0 || N < 0) return 0;
  return carve(nullptr, E, N, add_loops).bytes;
}

extern "C" int dgdm_csr_build(const int64_t* edge_index, int64_t E, int32_t N, int32_t add_loops, int32_t by_src,
                              int32_t* rowptr, int32_t* col, int32_t* eid,
                              void* workspace, size_t workspace_bytes, void* stream_) {
  DGDM_REQUIRE(E >= 0 && N >= 0 && rowptr);
  DGDM_REQUIRE(E == 0 || edge_index);
  const int64_t n_entries = E + (add_loops ? N : 0);
  DGDM_REQUIRE(n_entries == 0 || (col && eid));
  if (n_entries > 0x7fffffffLL) return DGDM_ERR_UNSUPPORTED;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (N == 0) {
    dgdm_fill_async(rowptr, 0, sizeof(int32_t), stream);
    return dgdm_launch_status();
  }
  DGDM_REQUIRE(workspace);
  Workspace w = carve(workspace, E, N, add_loops);
  if (workspace_bytes < w.bytes) return DGDM_ERR_WORKSPACE;
  const int64_t* keys = edge_index + (by_src ? 0 : E);
  const int64_t* vals = edge_index + (by_src ? E : 0);
  const int extra = add_loops ? 1 : 0;
  const int nblk = (N + SCAN_BLOCK - 1) / SCAN_BLOCK;
  // cnt and cursor are adjacent (both 256-B padded): one memset
  dgdm_fill_async(w.cnt, 0, (size_t)((char*)w.block_tot - (char*)w.cnt), stream);
  const int eb = (int)((E + 255) / 256 < 4096 ? (E + 255) / 256 : 4096);
  if (E > 0) hipLaunchKernelGGL(k_count, dim3(eb), dim3(256), 0, stream, keys, vals, E, N, w.cnt);
  hipLaunchKernelGGL(k_scan_block_totals, dim3(nblk), dim3(SCAN_BLOCK), 0, stream, w.cnt, N, extra, w.block_tot);
  hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(SCAN_BLOCK), 0, stream, w.block_tot, nblk);
  hipLaunchKernelGGL(k_scan_write, dim3(nblk), dim3(SCAN_BLOCK), 0, stream, w.cnt, N, extra, w.block_tot, rowptr);
  if (E > 0) hipLaunchKernelGGL(k_fill, dim3(eb), dim3(256), 0, stream, keys, vals, E, N, rowptr, w.cursor, w.tcol, w.teid);
  if (n_entries > 0) {
    const int rb = (N + 3) / 4 < 8192 ? (N + 3) / 4 : 8192;  // 4 waves per block, one row per wave
    hipLaunchKernelGGL(k_rank, dim3(rb), dim3(256), 0, stream, rowptr, N, (int32_t)E, extra, w.tcol, w.teid, col, eid);
  }
  return dgdm_launch_status();
}

extern "C" int dgdm_gcn_dinv(const int32_t* rowptr_dst, int32_t N, float* dinv, void* stream) {
  DGDM_REQUIRE(N >= 0);
  if (N == 0) return DGDM_OK;
  DGDM_REQUIRE(rowptr_dst && dinv);
  hipLaunchKernelGGL(k_dinv, dim3((N + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), rowptr_dst, N, dinv);
  return dgdm_launch_status();
}

extern "C" int dgdm_csr_edge_weights(const int32_t* rowptr, const int32_t* col, const float* dinv, int32_t N,
                                     float* w, void* stream) {
  DGDM_REQUIRE(N >= 0);
  if (N == 0) return DGDM_OK;
  DGDM_REQUIRE(rowptr && col && dinv && w);
  const int rb = (N + 3) / 4 < 8192 ? (N + 3) / 4 : 8192;
  hipLaunchKernelGGL(k_edge_weights, dim3(rb), dim3(256), 0, static_cast<hipStream_t>(stream), rowptr, col, dinv, N, w);
  return dgdm_launch_status();
}

extern "C" size_t dgdm_csr_build_pair_workspace_bytes(int64_t E, int32_t N, int32_t add_loops) {
  if (E < 0 || N < 0) return 0;
  return carve_pair(nullptr, E, N, add_loops).bytes;
}

extern "C" size_t dgdm_csr_build_pair_status_offset(int64_t E, int32_t N, int32_t add_loops) {
  if (E < 0 || N < 0) return 0;
  return carve_pair(nullptr, E, N, add_loops).status_offset;
}

extern "C" int dgdm_csr_build_pair(const int64_t* edge_index, int64_t E, int32_t N, int32_t add_loops,
                                   int32_t* rowptr_dst, int32_t* col_dst, int32_t* eid_dst, float* w_dst,
                                   int32_t* rowptr_src, int32_t* col_src, int32_t* eid_src, float* w_src, float* dinv,
                                   void* workspace, size_t workspace_bytes, int32_t* long_table_dst, int32_t* long_table_src,
                                   int32_t long_item_cap, void* stream_) {
  DGDM_REQUIRE(E >= 0 && N >= 0 && rowptr_dst && rowptr_src);
  DGDM_REQUIRE(E == 0 || edge_index);
  const int64_t n_entries = E + (add_loops ? N : 0);
  DGDM_REQUIRE(n_entries == 0 || (col_dst && eid_dst && w_dst && col_src && eid_src && w_src));
  if (n_entries > 0x7fffffffLL) return DGDM_ERR_UNSUPPORTED;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (N == 0) {
    dgdm_fill_async(rowptr_dst, 0, sizeof(int32_t), stream);
    dgdm_fill_async(rowptr_src, 0, sizeof(int32_t), stream);
    return dgdm_launch_status();
  }
  DGDM_REQUIRE(workspace && dinv);
  const PairWorkspace ws = carve_pair(workspace, E, N, add_loops);
  if (workspace_bytes < ws.bytes) return DGDM_ERR_WORKSPACE;
  PairArrays a;
  for (int o = 0; o < 2; ++o) { a.cnt[o] = ws.cnt[o]; a.cursor[o] = ws.cursor[o]; a.tcol[o] = ws.tcol[o]; a.teid[o] = ws.teid[o]; }
  a.rowptr[0] = rowptr_dst; a.col[0] = col_dst; a.eid[0] = eid_dst; a.w[0] = w_dst;
  a.rowptr[1] = rowptr_src; a.col[1] = col_src; a.eid[1] = eid_src; a.w[1] = w_src;
  a.status = ws.status;
  a.n_entries = (int32_t)n_entries;
  DGDM_REQUIRE((long_table_dst == nullptr) == (long_table_src == nullptr));
  if (long_table_dst && long_item_cap < dgdm_spmm_long_item_cap(n_entries)) return DGDM_ERR_WORKSPACE;
  if (long_table_dst && N > (1 << 20)) return DGDM_ERR_UNSUPPORTED;    // the tables are zeroed by the one-pass scan kernel
  a.long_table[0] = long_table_dst; a.long_table[1] = long_table_src; a.long_item_cap = long_item_cap;
  const int extra = add_loops ? 1 : 0;
  dgdm_fill_async(ws.cnt[0], 0, ws.counters_bytes, stream);
  const int eb = (int)((E + 255) / 256 < 4096 ? (E + 255) / 256 : 4096);
  if (E > 0) hipLaunchKernelGGL(k_count_pair, dim3(eb), dim3(256), 0, stream, edge_index, E, N, ws.cnt[0], ws.cnt[1]);
  if (N <= (1 << 20)) {
    hipLaunchKernelGGL(k_scan_pair, dim3((N + SCAN_CHUNK - 1) / SCAN_CHUNK, 2), dim3(SCAN_BLOCK), 0, stream, a, N, extra, dinv);
  } else {   // k_scan_pair reads O(N^2 / 4096) counts: beyond ~1M rows the three-pass scan is cheaper
    const int nblk = (N + SCAN_BLOCK - 1) / SCAN_BLOCK;
    for (int o = 0; o < 2; ++o) {
      int32_t* block_tot = ws.block_tot[o];
      hipLaunchKernelGGL(k_scan_block_totals, dim3(nblk), dim3(SCAN_BLOCK), 0, stream, ws.cnt[o], N, extra, block_tot);
      hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(SCAN_BLOCK), 0, stream, block_tot, nblk);
      hipLaunchKernelGGL(k_scan_write, dim3(nblk), dim3(SCAN_BLOCK), 0, stream, ws.cnt[o], N, extra, block_tot, a.rowptr[o]);
    }
    hipLaunchKernelGGL(k_dinv, dim3((N + 255) / 256), dim3(256), 0, stream, a.rowptr[0], N, dinv);
  }
  if (E > 0) hipLaunchKernelGGL(k_fill_pair, dim3(eb), dim3(256), 0, stream, edge_index, E, N, a);
  if (n_entries > 0) {
    const int rb = (N + 3) / 4 < 8192 ? (N + 3) / 4 : 8192;
    hipLaunchKernelGGL(k_rank_pair, dim3(rb, 2), dim3(256), 0, stream, a, N, (int32_t)E, extra, dinv);
    if (long_table_dst) {   // one wave per possible segment; waves beyond the tables' slot counts leave at once
      const int lb = (dgdm_spmm_long_slot_cap(n_entries) + 3) / 4;
      hipLaunchKernelGGL(k_rank_long, dim3(lb, 2), dim3(256), 0, stream, a, N, dinv, extra);
    }
  }
  return dgdm_launch_status();
}

extern "C" int dgdm_csr_extend(const int32_t* rowptr_dst, const int32_t* col_dst, const int32_t* eid_dst, const float* w_dst,
                               const int32_t* rowptr_src, const int32_t* col_src, const int32_t* eid_src, const float* w_src,
                               const float* dinv, const float* ea_hat, int32_t ea_dim, int64_t E, int32_t n_old, int32_t n_new,
                               int64_t entry_capacity, int32_t* rowptr_dst_out, int32_t* col_dst_out, int32_t* eid_dst_out,
                               float* w_dst_out, int32_t* rowptr_src_out, int32_t* col_src_out, int32_t* eid_src_out,
                               float* w_src_out, float* dinv_out, float* ea_hat_out, void* stream) {
  DGDM_REQUIRE(E >= 0 && n_old > 0 && n_new >= n_old && ea_dim >= 0 && entry_capacity >= 0);
  DGDM_REQUIRE(rowptr_dst && col_dst && eid_dst && w_dst && rowptr_src && col_src && eid_src && w_src && dinv);
  DGDM_REQUIRE(rowptr_dst_out && col_dst_out && eid_dst_out && w_dst_out && rowptr_src_out && col_src_out && eid_src_out &&
               w_src_out && dinv_out);
  DGDM_REQUIRE((ea_hat == nullptr) == (ea_hat_out == nullptr) && (ea_hat == nullptr || ea_dim > 0));
  if (E + (int64_t)n_new > 0x7fffffffLL) return DGDM_ERR_UNSUPPORTED;
  if (entry_capacity < E + (int64_t)n_new) return DGDM_ERR_WORKSPACE;     // the output arrays hold E + n_new entries
  ExtendArrays a;
  a.rowptr[0] = rowptr_dst; a.col[0] = col_dst; a.eid[0] = eid_dst; a.w[0] = w_dst;
  a.rowptr[1] = rowptr_src; a.col[1] = col_src; a.eid[1] = eid_src; a.w[1] = w_src;
  a.rowptr_o[0] = rowptr_dst_out; a.col_o[0] = col_dst_out; a.eid_o[0] = eid_dst_out; a.w_o[0] = w_dst_out;
  a.rowptr_o[1] = rowptr_src_out; a.col_o[1] = col_src_out; a.eid_o[1] = eid_src_out; a.w_o[1] = w_src_out;
  a.dinv = dinv; a.dinv_o = dinv_out; a.ea = ea_hat; a.ea_o = ea_hat_out;
  a.n_old = n_old; a.n_new = n_new; a.E = (int32_t)E; a.ea_dim = ea_dim;
  const int64_t work = E + (int64_t)n_new > (int64_t)n_new * (ea_dim > 0 ? ea_dim : 1) ? E + (int64_t)n_new : (int64_t)n_new * ea_dim;
  const int blocks = (int)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_csr_extend, dim3(blocks > 0 ? blocks : 1, ea_hat ? 3 : 2), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return dgdm_launch_status();
}
