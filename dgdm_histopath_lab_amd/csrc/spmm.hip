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

template <int LPR, int R, int UNROLL, bool AMAX>
__global__ __launch_bounds__(256) void k_spmm(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                              const float* __restrict__ w, const float* __restrict__ X, int64_t ldx,
                                              int32_t table_rows, float* __restrict__ Y, int64_t ldy, int32_t N, int32_t C,
                                              const float* __restrict__ bias, int accumulate,
                                              const float* __restrict__ tail, int64_t ldt, int tail_c4,
                                              const float* __restrict__ addend, int64_t lda, unsigned* __restrict__ amax) {
  constexpr int RPW = 64 / LPR;                 // rows per wave
  unsigned am = 0;                              // max |Y| over what this thread writes (GEMM operand scale, see common.hpp)
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR, lir = lane % LPR;  // which row of the wave, lane inside the row
  const int wave_global = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  const int c4 = C >> 2;                        // float4 per row

  for (int row0 = wave_global * RPW; row0 < N; row0 += nwaves * RPW) {
    const int row = row0 + sub;
    if (row >= N) continue;
    const int start = rowptr[row], end = rowptr[row + 1];
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int p = start; p < end; p += UNROLL) {
      int cc[UNROLL];
      float ww[UNROLL];
#pragma unroll
      for (int j = 0; j < UNROLL; ++j) {
        const int q = p + j < end ? p + j : end - 1;  // clamp: tail entries get weight 0
        const int c = col[q];
        const bool ok = (p + j < end) && (c < table_rows);
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

    float4* dst = reinterpret_cast<float4*>(Y + (int64_t)row * ldy);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int k = lir + r * LPR;
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
      for (int k = lir; k < tail_c4; k += LPR) {
        const float4 tv = t[k];
        dst[c4 + k] = tv;
        if (AMAX) am = dgdm_amax4(am, tv);
      }
    }
  }
  if (AMAX) dgdm_amax_commit(am, amax);
}

template <int LPR, int R, int UNROLL>
int launch(const int32_t* rowptr, const int32_t* col, const float* w, const float* X, int64_t ldx, int32_t table_rows,
           float* Y, int64_t ldy, int32_t N, int32_t C, const float* bias, int accumulate, const float* tail, int64_t ldt,
           int tail_c4, const float* addend, int64_t lda, uint32_t* amax, hipStream_t stream) {
  constexpr int RPW = 64 / LPR;
  const int64_t waves = ((int64_t)N + RPW - 1) / RPW;
  int64_t blocks = (waves + 3) / 4;
  if (blocks > 256 * 64) blocks = 256 * 64;  // grid-stride beyond that
  if (amax)   // the variant that also keeps max|Y| (GEMM operand scale); the plain kernel carries none of that code
    hipLaunchKernelGGL((k_spmm<LPR, R, UNROLL, true>), dim3((unsigned)blocks), dim3(256), 0, stream, rowptr, col, w, X, ldx,
                       table_rows, Y, ldy, N, C, bias, accumulate, tail, ldt, tail_c4, addend, lda, amax);
  else
    hipLaunchKernelGGL((k_spmm<LPR, R, UNROLL, false>), dim3((unsigned)blocks), dim3(256), 0, stream, rowptr, col, w, X, ldx,
                       table_rows, Y, ldy, N, C, bias, accumulate, tail, ldt, tail_c4, addend, lda, amax);
  return dgdm_launch_status();
}

}  // namespace

static int spmm_dispatch(const int32_t* rowptr, const int32_t* col, const float* w, const float* X, int64_t ldx,
                         int32_t table_rows, float* Y, int64_t ldy, int32_t N, int32_t C, const float* bias,
                         int32_t accumulate, const float* tail, int64_t ldt, int32_t Ct, const float* addend, int64_t lda,
                         uint32_t* amax, void* stream_) {
  DGDM_REQUIRE(N >= 0 && C > 0 && table_rows >= 0 && Ct >= 0);
  if (N == 0) return DGDM_OK;
  DGDM_REQUIRE(rowptr && col && w && Y);
  DGDM_REQUIRE(table_rows == 0 || X);
  DGDM_REQUIRE(Ct == 0 || tail);
  if ((C & 3) || C > 1024 || (ldx & 3) || (ldy & 3) || ldx < C || ldy < C + Ct) return DGDM_ERR_UNSUPPORTED;
  if (!dgdm_aligned16(X) || !dgdm_aligned16(Y) || (bias && !dgdm_aligned16(bias))) return DGDM_ERR_UNSUPPORTED;
  if (Ct && ((Ct & 3) || (ldt & 3) || ldt < Ct || !dgdm_aligned16(tail) || accumulate)) return DGDM_ERR_UNSUPPORTED;
  if (addend && ((lda & 3) || lda < C || !dgdm_aligned16(addend))) return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  if (table_rows == 0) {  // nothing to gather: Y = 0 (or unchanged when accumulating)
    if (bias || Ct || addend) return DGDM_ERR_UNSUPPORTED;
    if (!accumulate) dgdm_fill2d_async(Y, ldy, C, N, s);
    return dgdm_launch_status();
  }
  const int c4 = C >> 2;
  if (!Ct) tail = nullptr;
#define GO(LPR, R, U) return launch<LPR, R, U>(rowptr, col, w, X, ldx, table_rows, Y, ldy, N, C, bias, accumulate, tail, ldt, Ct >> 2, addend, lda, amax, s)
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
                         int32_t accumulate, void* stream) {
  return spmm_dispatch(rowptr, col, w, X, ldx, table_rows, Y, ldy, N, C, bias, accumulate, nullptr, 0, 0, nullptr, 0, nullptr, stream);
}

extern "C" int dgdm_spmm_concat(const int32_t* rowptr, const int32_t* col, const float* w, const float* X, int64_t ldx,
                                int32_t table_rows, const float* tail, int64_t ldt, int32_t Ct, float* Y, int64_t ldy,
                                int32_t N, int32_t C, uint32_t* amax, void* stream) {
  return spmm_dispatch(rowptr, col, w, X, ldx, table_rows, Y, ldy, N, C, nullptr, 0, tail, ldt, Ct, nullptr, 0, amax, stream);
}

extern "C" int dgdm_spmm_add(const int32_t* rowptr, const int32_t* col, const float* w, const float* X, int64_t ldx,
                             int32_t table_rows, const float* addend, int64_t lda, float* Y, int64_t ldy, int32_t N, int32_t C,
                             void* stream) {
  DGDM_REQUIRE(addend);
  return spmm_dispatch(rowptr, col, w, X, ldx, table_rows, Y, ldy, N, C, nullptr, 0, nullptr, 0, 0, addend, lda, nullptr, stream);
}
