// K2: CSR segmented gather-reduce  Y[r,:] = sum_{p in row r} w[p] * X[col[p],:] (+ bias).
// Replaces MessagePassing.propagate's gather / norm-scale / scatter-add
// (core/graph_layers.py:92,99-110).  HBM/L2-bandwidth bound: 16 B per lane coalesced row reads,
// UNROLL source rows in flight per lane group, no atomics (a lane group owns its output row and
// adds the entries in CSR order => bitwise reproducible).
//
// Geometry: a row of C floats is covered by LPR lanes x R float4 per lane (C = 4*LPR*R for the
// exact fits; a tail predicate handles other C % 4 == 0 widths).  LPR < 64 packs 64/LPR rows
// into one wavefront so narrow rows (C = 32, 128) still issue full 1-KiB wave loads.
#include "common.hpp"

namespace {

struct LongArgs {          // device view of DgdmLongRows (null table: no long-row handling)
  int32_t* table;
  float* partial;
  int64_t ld;
  int32_t item_cap, slot_cap;
  int32_t first_block;     // blocks [first_block, gridDim.x) work on segments of long rows
};

// acc[r] += sum over entries [p0, p1) of w[p] * X[col[p], :]  (this lane's float4 columns), entries in order
template <int LPR, int R, int UNROLL>
__device__ __forceinline__ void gather_range(const int32_t* __restrict__ col, const float* __restrict__ w, const float* __restrict__ X,
                                             int64_t ldx, int32_t table_rows, int c4, int lir, int p0, int p1, float4 (&acc)[R]) {
  for (int p = p0; p < p1; p += UNROLL) {
    int cc[UNROLL];
    float ww[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      const int q = p + j < p1 ? p + j : p1 - 1;  // clamp: tail entries get weight 0
      const int c = col[q];
      const bool ok = (p + j < p1) && (c < table_rows);
      cc[j] = ok ? c : 0;
      ww[j] = ok ? w[q] : 0.f;
    }
    float4 v[UNROLL][R];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      const float4* src = reinterpret_cast<const float4*>(X + (int64_t)cc[j] * ldx);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int k = lir + r * LPR;
        v[j][r] = (k < c4) ? src[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < UNROLL; ++j)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = f4_fma(ww[j], v[j][r], acc[r]);
  }
}

template <int LPR, int R, int UNROLL, bool AMAX>
__global__ __launch_bounds__(256) void k_spmm(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                              const float* __restrict__ w, const float* __restrict__ X, int64_t ldx,
                                              int32_t table_rows, float* __restrict__ Y, int64_t ldy, int32_t N, int32_t C,
                                              const float* __restrict__ bias, int accumulate,
                                              const float* __restrict__ tail, int64_t ldt, int tail_c4,
                                              const float* __restrict__ addend, int64_t lda, unsigned* __restrict__ amax, const LongArgs lg) {
  constexpr int RPW = 64 / LPR;                 // rows per wave
  unsigned am = 0;                              // max |Y| over what this thread writes (GEMM operand scale, see common.hpp)
  const int lane = threadIdx.x & 63;
  const int c4 = C >> 2;                        // float4 per row

  // the row's epilogue: bias / accumulate / addend, store, the concatenated tail, the operand maximum
  auto finish_row = [&](int row, int lir, int lpr, float4 (&acc)[R]) {
    float4* dst = reinterpret_cast<float4*>(Y + (int64_t)row * ldy);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int k = lir + r * lpr;
      if (k < c4) {
        float4 o = acc[r];
        if (bias) {
          const float4 b = reinterpret_cast<const float4*>(bias)[k];
          o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
        }
        if (accumulate) {
          const float4 y0 = dst[k];
          o.x += y0.x; o.y += y0.y; o.z += y0.z; o.w += y0.w;
        }
        if (addend) {  // Y = result + addend: a second gradient arriving at the same tensor (residual connection), no extra pass
          const float4 a0 = reinterpret_cast<const float4*>(addend + (int64_t)row * lda)[k];
          o.x += a0.x; o.y += a0.y; o.z += a0.z; o.w += a0.w;
        }
        dst[k] = o;
        if (AMAX) am = dgdm_amax4(am, o);
      }
    }
    if (tail) {  // Y[row, C : C + 4*tail_c4) = tail[row, :]  (the [A_hat x | EA_hat] operand of a graph convolution)
      const float4* t = reinterpret_cast<const float4*>(tail + (int64_t)row * ldt);
      for (int k = lir; k < tail_c4; k += lpr) {
        const float4 tv = t[k];
        dst[c4 + k] = tv;
        if (AMAX) am = dgdm_amax4(am, tv);
      }
    }
  };

  if (lg.table == nullptr || (int)blockIdx.x < lg.first_block) {
    const int sub = lane / LPR, lir = lane % LPR;  // which row of the wave, lane inside the row
    const int nblocks = lg.table ? lg.first_block : (int)gridDim.x;
    const int wave_global = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    const int nwaves = nblocks * (blockDim.x >> 6);
    for (int row0 = wave_global * RPW; row0 < N; row0 += nwaves * RPW) {
      const int row = row0 + sub;
      if (row >= N) continue;
      const int start = rowptr[row], end = rowptr[row + 1];
      if (lg.table && end - start > DGDM_SPMM_LONG_ROW) continue;      // a long row: the segment waves below own it
      float4 acc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      gather_range<LPR, R, UNROLL>(col, w, X, ldx, table_rows, c4, lir, start, end, acc);
      finish_row(row, lir, LPR, acc);
    }
  } else {
    // ---- long rows: one LANE GROUP (the LPR lanes that own a row in the pass above) per segment of DGDM_SPMM_SEGMENT entries --
    // same register footprint as the row pass.  The group that arrives last at the row's counter adds the partial sums in
    // segment order and runs the row's epilogue.
    const int sub = lane / LPR, lir = lane % LPR;
    const int32_t* __restrict__ t = lg.table;
    const int count = min(t[0], lg.item_cap), slots = min(t[1], lg.slot_cap);
    const int slot = ((((int)blockIdx.x - lg.first_block) * (blockDim.x >> 6)) + (threadIdx.x >> 6)) * RPW + sub;
    if (slot < slots) {
      int item = -1, row = 0, slot0 = 0, nseg = 0;
      for (int i = 0; i < count; ++i) {        // items are few (<= n_entries / DGDM_SPMM_LONG_ROW); every lane of the group walks them
        const int r_ = t[2 + 2 * i], s0 = t[2 + 2 * i + 1];
        const int ns = (rowptr[r_ + 1] - rowptr[r_] + DGDM_SPMM_SEGMENT - 1) / DGDM_SPMM_SEGMENT;
        if (slot >= s0 && slot < s0 + ns) { item = i; row = r_; slot0 = s0; nseg = ns; }
      }
      if (item >= 0) {
        const int start = rowptr[row], end = rowptr[row + 1];
        const int p0 = start + (slot - slot0) * DGDM_SPMM_SEGMENT, p1 = min(end, p0 + DGDM_SPMM_SEGMENT);
        float4 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        gather_range<LPR, R, UNROLL>(col, w, X, ldx, table_rows, c4, lir, p0, p1, acc);
        float4* part = reinterpret_cast<float4*>(lg.partial + (int64_t)slot * lg.ld);
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (lir + r * LPR < c4) part[lir + r * LPR] = acc[r];
        __threadfence();                                   // partial sums visible device-wide before the arrival is counted
        int32_t* sem = lg.table + 2 + 2 * lg.item_cap + item;
        int arrived = 0;
        if (lir == 0) arrived = atomicAdd(sem, 1);
        arrived = __shfl(arrived, sub * LPR, 64);
        if (arrived == nseg - 1) {                         // last segment of this row to finish: reduce in segment order
          __threadfence();
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
          typedef float f32x4_t __attribute__((ext_vector_type(4)));
          for (int sg0 = 0; sg0 < nseg; sg0 += 4) {       // four partial rows in flight; added in segment order
            f32x4_t v[4][R];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int sgm = sg0 + u < nseg ? sg0 + u : nseg - 1;
              const f32x4_t* ps = reinterpret_cast<const f32x4_t*>(lg.partial + (int64_t)(slot0 + sgm) * lg.ld);
#pragma unroll
              for (int r = 0; r < R; ++r) {
                const int k = lir + r * LPR;
                v[u][r] = k < c4 ? __builtin_nontemporal_load(ps + k) : f32x4_t{0.f, 0.f, 0.f, 0.f};   // written by other CUs:
              }                                                                                          // no cached copy wanted
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (sg0 + u < nseg) {
#pragma unroll
                for (int r = 0; r < R; ++r) { acc[r].x += v[u][r][0]; acc[r].y += v[u][r][1]; acc[r].z += v[u][r][2]; acc[r].w += v[u][r][3]; }
              }
          }
          finish_row(row, lir, LPR, acc);
          if (lir == 0) *sem = 0;                           // ready for the next launch that uses this table
        }
      }
    }
  }
  if (AMAX) dgdm_amax_commit(am, amax);
}

template <int LPR, int R, int UNROLL>
int launch(const int32_t* rowptr, const int32_t* col, const float* w, const float* X, int64_t ldx, int32_t table_rows,
           float* Y, int64_t ldy, int32_t N, int32_t C, const float* bias, int accumulate, const float* tail, int64_t ldt,
           int tail_c4, const float* addend, int64_t lda, uint32_t* amax, const DgdmLongRows* lr, hipStream_t stream) {
  constexpr int RPW = 64 / LPR;
  const int64_t waves = ((int64_t)N + RPW - 1) / RPW;
  int64_t blocks = (waves + 3) / 4;
  if (blocks > 256 * 64) blocks = 256 * 64;  // grid-stride beyond that
  LongArgs lg = {nullptr, nullptr, 0, 0, 0, 0};
  int64_t grid = blocks;
  if (lr) {
    lg.table = lr->table; lg.partial = lr->partial; lg.ld = lr->ld; lg.item_cap = lr->item_cap; lg.slot_cap = lr->slot_cap;
    lg.first_block = (int32_t)blocks;
    grid = blocks + (lr->slot_cap + 4 * RPW - 1) / (4 * RPW);      // a lane group per possible segment; groups beyond the table's
                                                                    // slot count leave at once
  }
  if (amax)   // the variant that also keeps max|Y| (GEMM operand scale); the plain kernel carries none of that code
    hipLaunchKernelGGL((k_spmm<LPR, R, UNROLL, true>), dim3((unsigned)grid), dim3(256), 0, stream, rowptr, col, w, X, ldx,
                       table_rows, Y, ldy, N, C, bias, accumulate, tail, ldt, tail_c4, addend, lda, amax, lg);
  else
    hipLaunchKernelGGL((k_spmm<LPR, R, UNROLL, false>), dim3((unsigned)grid), dim3(256), 0, stream, rowptr, col, w, X, ldx,
                       table_rows, Y, ldy, N, C, bias, accumulate, tail, ldt, tail_c4, addend, lda, amax, lg);
  return dgdm_launch_status();
}

}  // namespace

static int spmm_dispatch(const int32_t* rowptr, const int32_t* col, const float* w, const float* X, int64_t ldx,
                         int32_t table_rows, float* Y, int64_t ldy, int32_t N, int32_t C, const float* bias,
                         int32_t accumulate, const float* tail, int64_t ldt, int32_t Ct, const float* addend, int64_t lda,
                         uint32_t* amax, const DgdmLongRows* lr, void* stream_) {
  DGDM_REQUIRE(N >= 0 && C > 0 && table_rows >= 0 && Ct >= 0);
  if (N == 0) return DGDM_OK;
  DGDM_REQUIRE(rowptr && col && w && Y);
  DGDM_REQUIRE(table_rows == 0 || X);
  DGDM_REQUIRE(Ct == 0 || tail);
  if ((C & 3) || C > 1024 || (ldx & 3) || (ldy & 3) || ldx < C || ldy < C + Ct) return DGDM_ERR_UNSUPPORTED;
  if (!dgdm_aligned16(X) || !dgdm_aligned16(Y) || (bias && !dgdm_aligned16(bias))) return DGDM_ERR_UNSUPPORTED;
  if (Ct && ((Ct & 3) || (ldt & 3) || ldt < Ct || !dgdm_aligned16(tail) || accumulate)) return DGDM_ERR_UNSUPPORTED;
  if (addend && ((lda & 3) || lda < C || !dgdm_aligned16(addend))) return DGDM_ERR_UNSUPPORTED;
  if (lr) {
    DGDM_REQUIRE(lr->table && lr->partial && lr->item_cap > 0 && lr->slot_cap > 0);
    if (lr->ld < C || (lr->ld & 3) || !dgdm_aligned16(lr->partial)) return DGDM_ERR_WORKSPACE;
  }
  hipStream_t s = static_cast<hipStream_t>(stream_);
  if (table_rows == 0) {  // nothing to gather: Y = 0 (or unchanged when accumulating)
    if (bias || Ct || addend) return DGDM_ERR_UNSUPPORTED;
    if (!accumulate) dgdm_fill2d_async(Y, ldy, C, N, s);
    return dgdm_launch_status();
  }
  const int c4 = C >> 2;
  if (!Ct) tail = nullptr;
#define GO(LPR, R, U) return launch<LPR, R, U>(rowptr, col, w, X, ldx, table_rows, Y, ldy, N, C, bias, accumulate, tail, ldt, Ct >> 2, addend, lda, amax, lr, s)
  if (c4 <= 8) GO(8, 1, 4);
  if (c4 <= 16) GO(16, 1, 4);
  if (c4 <= 32) GO(32, 1, 4);
  if (c4 <= 64) GO(64, 1, 4);
  if (c4 <= 128) GO(64, 2, 4);
  if (c4 <= 192) GO(64, 3, 4);
  GO(64, 4, 2);
#undef GO
}

extern "C" int dgdm_spmm(const int32_t* rowptr, const int32_t* col, const float* w, const float* X, int64_t ldx,
                         int32_t table_rows, float* Y, int64_t ldy, int32_t N, int32_t C, const float* bias,
                         int32_t accumulate, const DgdmLongRows* long_rows, void* stream) {
  return spmm_dispatch(rowptr, col, w, X, ldx, table_rows, Y, ldy, N, C, bias, accumulate, nullptr, 0, 0, nullptr, 0, nullptr, long_rows, stream);
}

extern "C" int dgdm_spmm_concat(const int32_t* rowptr, const int32_t* col, const float* w, const float* X, int64_t ldx,
                                int32_t table_rows, const float* tail, int64_t ldt, int32_t Ct, float* Y, int64_t ldy,
                                int32_t N, int32_t C, uint32_t* amax, const DgdmLongRows* long_rows, void* stream) {
  return spmm_dispatch(rowptr, col, w, X, ldx, table_rows, Y, ldy, N, C, nullptr, 0, tail, ldt, Ct, nullptr, 0, amax, long_rows, stream);
}

extern "C" int dgdm_spmm_add(const int32_t* rowptr, const int32_t* col, const float* w, const float* X, int64_t ldx,
                             int32_t table_rows, const float* addend, int64_t lda, float* Y, int64_t ldy, int32_t N, int32_t C,
                             const DgdmLongRows* long_rows, void* stream) {
  DGDM_REQUIRE(addend);
  return spmm_dispatch(rowptr, col, w, X, ldx, table_rows, Y, ldy, N, C, nullptr, 0, nullptr, 0, 0, addend, lda, nullptr, long_rows, stream);
}
