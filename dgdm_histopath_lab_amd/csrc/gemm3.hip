// K3': the same three contractions as gemm.hip (forward, dX and dW/db of every nn.Linear of the hot
// path) with fp32-level accuracy on the 16-bit matrix pipe.
//
// Why: v_mfma_f32_32x32x2_f32 runs at 64 FLOP/clk/SIMD (157 TFLOP/s on the chip) and does not
// overlap with VALU work; v_mfma_f32_32x32x16_bf16 runs at 1024 FLOP/clk/SIMD and holds the
// issue port for only 8 of its 32 cycles.  Every fp32 operand x is split exactly into three bf16
// values, x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest:
// |m| <= 2^-9 |x|, |l| <= 2^-18 |x|; the three 8-bit significands cover the 24 bits of x), and a
// product is evaluated as the six terms that are not below 2^-26 of it:
//     a.b = ah.bh + ah.bm + am.bh + ah.bl + al.bh + am.bm   (+ am.bl + al.bm + al.bl, dropped)
// Each term is an exact bf16 x bf16 product accumulated in fp32 by the MFMA, so the result differs
// from an fp32-MFMA GEMM by a few 2^-26 relative per product -- below the fp32 rounding of the
// accumulation itself (tests/test_hip_gemm.py measures both against fp64).
// 6 MFMAs at 16x the fp32 rate = 2.67x the fp32 matrix peak (419 TFLOP/s-equivalent).
//
// Tiling as gemm.hip: 128x128 output tile per 256-thread workgroup, 4 waves x (2x2) 32x32 tiles,
// 32-deep k blocks, next block's global loads in registers under the current block's MFMAs.  The
// split happens on the way from registers to LDS (v_cvt_pk_bf16_f32 + packed subtract: 4.5 VALU
// per element, issued in the shadow of the other resident workgroup's MFMAs); the LDS image of an
// operand is three planes [3][128 rows][32 k] of bf16 with 80-byte rows, which makes the
// ds_read_b128 operand reads of all four lane groups conflict-free.
// Non-finite inputs: Inf - Inf in the residual turns an Inf operand into NaN (the model rejects
// non-finite features before any GEMM, models/dgdm_model.py:278-283 in the reference).
#include "common.hpp"
#include "colsum.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int RSB = 80;                 // bytes per LDS row: 32 bf16 + 16 B pad
constexpr int PLANE = BM * RSB;         // bytes per plane (10240)
constexpr int OPERAND = 3 * PLANE;      // bytes per operand image (30720)

__device__ __forceinline__ f32x16 mfma_bf(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {   // {bf16(a) in the low half, bf16(b) in the high half}
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

// exact three-way split of a pair of floats; each output dword holds the pair's bf16 values
__device__ __forceinline__ void split_pair(float a, float b, unsigned* h, unsigned* m, unsigned* l) {
  const unsigned hh = cvt_pk(a, b);
  const float ar = a - __uint_as_float(hh << 16), br = b - __uint_as_float(hh & 0xffff0000u);
  const unsigned mm = cvt_pk(ar, br);
  *h = hh;
  *m = mm;
  *l = cvt_pk(ar - __uint_as_float(mm << 16), br - __uint_as_float(mm & 0xffff0000u));
}

// one 32-deep block: acc[mt][nt] += A(wave rows, k) . B(wave cols, k), six bf16 products per term
__device__ __forceinline__ void mma_block3(const char* __restrict__ As, const char* __restrict__ Bs, int arow0, int brow0, int lane,
                                           f32x16 (&acc)[2][2], int live_m = 2, int live_n = 2) {
  if (live_m == 0 || live_n == 0) return;
  const int i = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int s = 0; s < BK / 16; ++s) {
    bf16x8 a[3][2], b[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[p][t] = *reinterpret_cast<const bf16x8*>(As + p * PLANE + (arow0 + t * 32 + i) * RSB + 32 * s + 16 * kh);
        b[p][t] = *reinterpret_cast<const bf16x8*>(Bs + p * PLANE + (brow0 + t * 32 + i) * RSB + 32 * s + 16 * kh);
      }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if (mt >= live_m) continue;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        if (nt >= live_n) continue;
        f32x16 c = acc[mt][nt];
        c = mfma_bf(a[1][mt], b[1][nt], c);   // smallest terms first
        c = mfma_bf(a[0][mt], b[2][nt], c);
        c = mfma_bf(a[2][mt], b[0][nt], c);
        c = mfma_bf(a[0][mt], b[1][nt], c);
        c = mfma_bf(a[1][mt], b[0][nt], c);
        c = mfma_bf(a[0][mt], b[0][nt], c);
        acc[mt][nt] = c;
      }
    }
  }
}

// ---- a [128 rows x 32 k] tile whose global layout has the reduction index contiguous
struct RowTile3 {
  float4 v[4];
  unsigned okbits;
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int row0, int nrows, int k0, int K, int tid) {
    okbits = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, r = idx >> 3, c4 = idx & 7;
      const int row = row0 + r, k = k0 + 4 * c4;
      const int rc = row < nrows ? row : nrows - 1, kc = k < K ? k : K - 4;   // clamped address, masked at store time
      v[i] = *reinterpret_cast<const float4*>(P + (int64_t)rc * ld + kc);
      okbits |= (row < nrows && k < K) ? (1u << i) : 0u;
    }
  }
  __device__ __forceinline__ void store(char* __restrict__ S, int tid) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, r = idx >> 3, c4 = idx & 7;
      const bool ok = (okbits >> i) & 1u;
      const float x = ok ? v[i].x : 0.f, y = ok ? v[i].y : 0.f, z = ok ? v[i].z : 0.f, w = ok ? v[i].w : 0.f;
      uint2 h, m, l;
      split_pair(x, y, &h.x, &m.x, &l.x);
      split_pair(z, w, &h.y, &m.y, &l.y);
      char* dst = S + r * RSB + 8 * c4;
      *reinterpret_cast<uint2*>(dst) = h;
      *reinterpret_cast<uint2*>(dst + PLANE) = m;
      *reinterpret_cast<uint2*>(dst + 2 * PLANE) = l;
    }
  }
};

// ---- a [32 k x 128 cols] tile whose global layout has the OUTPUT index contiguous: transposed into
// the [col][k] image.  A thread holds two pairs of adjacent k rows (i = 2q, 2q+1) of 4 columns, so a
// column's two k values pack into one dword: ds_write_b32 at [col][k pair], banks {pair} + {0,16}
// per 32-lane half = 2-way, which ds_write_b32 absorbs.
struct ColTile3 {
  __device__ static __forceinline__ void map(int tid, int i, int* kr, int* c4) {
    const int lane = tid & 63, wave = tid >> 6;
    *kr = 16 * (i >> 1) + 2 * (lane & 7) + (i & 1);
    *c4 = (lane >> 3) + 8 * wave;
  }
  float4 v[4];
  unsigned okbits;
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int k0, int K, int col0, int ncols, int tid) {
    okbits = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int kr, c4;
      map(tid, i, &kr, &c4);
      const int k = k0 + kr, col = col0 + 4 * c4;
      const int kc = k < K ? k : K - 1, cc = col < ncols ? col : ncols - 4;
      v[i] = *reinterpret_cast<const float4*>(P + (int64_t)kc * ld + cc);
      okbits |= (k < K && col < ncols) ? (1u << i) : 0u;
    }
  }
  // sum over this thread's valid k rows of each of its 4 columns (bias gradient of the dW kernel)
  __device__ __forceinline__ void add_colsum(float (&s)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = (okbits >> i) & 1u;
      s[0] += ok ? v[i].x : 0.f; s[1] += ok ? v[i].y : 0.f; s[2] += ok ? v[i].z : 0.f; s[3] += ok ? v[i].w : 0.f;
    }
  }
  __device__ __forceinline__ void store(char* __restrict__ S, int tid) const {
    const int lane = tid & 63, wave = tid >> 6;
    const int c4 = (lane >> 3) + 8 * wave;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const bool ok0 = (okbits >> (2 * q)) & 1u, ok1 = (okbits >> (2 * q + 1)) & 1u;
      const float4 a = v[2 * q], b = v[2 * q + 1];
      const float a4[4] = {a.x, a.y, a.z, a.w}, b4[4] = {b.x, b.y, b.z, b.w};
      const int kpair = 8 * q + (lane & 7);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned h, m, l;
        split_pair(ok0 ? a4[j] : 0.f, ok1 ? b4[j] : 0.f, &h, &m, &l);
        char* dst = S + (4 * c4 + j) * RSB + 4 * kpair;
        *reinterpret_cast<unsigned*>(dst) = h;
        *reinterpret_cast<unsigned*>(dst + PLANE) = m;
        *reinterpret_cast<unsigned*>(dst + 2 * PLANE) = l;
      }
    }
  }
};

template <bool B_KCONTIG, bool ACCUM>
__global__ __launch_bounds__(256, 2) void k_gemm3_rows(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                       int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                       const float* __restrict__ bias) {
  __shared__ __attribute__((aligned(16))) char smem[2 * OPERAND];
  char* As = smem;
  char* Bs = smem + OPERAND;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int wr = wave >> 1, wc = wave & 1;
  const int live_m = min(2, max(0, (M - (m0 + wr * 64) + 31) / 32)), live_n = min(2, max(0, (N - (n0 + wc * 64) + 31) / 32));
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  RowTile3 ta;
  RowTile3 tbr;
  ColTile3 tbc;
  ta.load(A, lda, m0, M, 0, K, tid);
  if (B_KCONTIG) tbr.load(B, ldb, n0, N, 0, K, tid); else tbc.load(B, ldb, 0, K, n0, N, tid);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
    ta.store(As, tid);
    if (B_KCONTIG) tbr.store(Bs, tid); else tbc.store(Bs, tid);
    __syncthreads();
    if (k0 + BK < K) {
      ta.load(A, lda, m0, M, k0 + BK, K, tid);
      if (B_KCONTIG) tbr.load(B, ldb, n0, N, k0 + BK, K, tid); else tbc.load(B, ldb, k0 + BK, K, n0, N, tid);
    }
    mma_block3(As, Bs, wr * 64, wc * 64, lane, acc, live_m, live_n);
  }

  const int j = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = n0 + wc * 64 + nt * 32 + j;
      const int colc = col < N ? col : N - 1;
      const float bv = bias ? bias[colc] : 0.f;
      float old[16];
      if (ACCUM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          old[r] = C[(int64_t)(row < M ? row : M - 1) * ldc + colc];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < M && col < N) C[(int64_t)row * ldc + col] = acc[mt][nt][r] + bv + (ACCUM ? old[r] : 0.f);
      }
    }
}

// dW partial: tile (n0, kk0) of [N x K], rows [mc*chunk, (mc+1)*chunk); partial row of chunk mc =
// [N*K dW elements | N bias sums (if with_bias)], as k_gemm_tn_partial of gemm.hip.
__global__ __launch_bounds__(256, 2) void k_gemm3_tn_partial(const float* __restrict__ dY, int64_t ldy, const float* __restrict__ X,
                                                             int64_t ldx, int M, int N, int K, int chunk, int with_bias,
                                                             float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) char smem[2 * OPERAND];
  char* Ys = smem;            // [n][m]
  char* Xs = smem + OPERAND;  // [k][m]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * BM, kk0 = blockIdx.y * BN, mc = blockIdx.z;
  const int mbeg = mc * chunk, mend = min(M, mbeg + chunk);
  const int wr = wave >> 1, wc = wave & 1;
  const int live_m = min(2, max(0, (N - (n0 + wr * 64) + 31) / 32)), live_n = min(2, max(0, (K - (kk0 + wc * 64) + 31) / 32));
  const bool do_bias = with_bias && blockIdx.y == 0;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  ColTile3 ty, tx;
  ty.load(dY, ldy, mbeg, mend, n0, N, tid);
  tx.load(X, ldx, mbeg, mend, kk0, K, tid);
  for (int m = mbeg; m < mend; m += BK) {
    __syncthreads();
    ty.store(Ys, tid);
    tx.store(Xs, tid);
    if (do_bias) ty.add_colsum(bs);
    __syncthreads();
    if (m + BK < mend) {
      ty.load(dY, ldy, m + BK, mend, n0, N, tid);
      tx.load(X, ldx, m + BK, mend, kk0, K, tid);
    }
    mma_block3(Ys, Xs, wr * 64, wc * 64, lane, acc, live_m, live_n);
  }
  const int j = lane & 31, hi = lane >> 5;
  const int64_t width = (int64_t)N * K + (with_bias ? N : 0);
  float* P = partial + (int64_t)mc * width;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = kk0 + wc * 64 + nt * 32 + j;
      if (col >= K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < N) P[(int64_t)row * K + col] = acc[mt][nt][r];
      }
    }
  if (do_bias) {  // the 8 lanes (lane & 7) of one c4 hold the k rows of the same 4 columns
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bs[q] += __shfl_xor(bs[q], 1, 64);
      bs[q] += __shfl_xor(bs[q], 2, 64);
      bs[q] += __shfl_xor(bs[q], 4, 64);
    }
    const int n = n0 + 4 * ((lane >> 3) + 8 * wave);
    if ((lane & 7) == 0 && n < N) *reinterpret_cast<float4*>(&P[(int64_t)N * K + n]) = make_float4(bs[0], bs[1], bs[2], bs[3]);
  }
}

// dW[n][k] / db[n] = sum over slots of in[slot][n*K + k] / in[slot][N*K + n]; fixed order.
__global__ __launch_bounds__(256) void k_gemm3_tn_final(const float* __restrict__ in, int slots, int64_t width, int N, int K,
                                                        float* __restrict__ dW, int64_t lddw, float* __restrict__ db) {
  const int64_t col = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  float acc = 0.f;
  if (col < width)
    for (int s = part; s < slots; s += 4) acc += in[(int64_t)s * width + col];
  __shared__ float sm[4][64];
  sm[part][threadIdx.x & 63] = acc;
  __syncthreads();
  if (part == 0 && col < width) {
    const float t = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
    const int64_t nk = (int64_t)N * K;
    if (col < nk) dW[(col / K) * lddw + col % K] = t; else db[col - nk] = t;
  }
}

constexpr int TN3_STAGE_SLOTS = 16;

int tn3_chunk_rows(int M, int N, int K) {
  const int tiles = ((N + BM - 1) / BM) * ((K + BN - 1) / BN);
  int want = (496 + tiles / 2) / tiles;   // ~2 workgroups per CU, all resident in one round
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  int chunk = (M + want - 1) / want;
  chunk = (chunk + BK - 1) / BK * BK;
  if (chunk < 4 * BK) chunk = 4 * BK;
  return chunk;
}

}  // namespace

extern "C" int dgdm_gemm_nt_bf16x3(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                                   int32_t M, int32_t N, int32_t K, int32_t accumulate, void* stream) {
  if (M < 0 || N < 0 || K < 0) return DGDM_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return DGDM_OK;
  if (!A || !W || !C) return DGDM_ERR_INVALID_ARG;
  if (K == 0) return DGDM_ERR_UNSUPPORTED;
  if ((K & 3) || (lda & 3) || (ldw & 3) || !dgdm_aligned16(A) || !dgdm_aligned16(W)) return DGDM_ERR_UNSUPPORTED;
  if (lda < K || ldw < K || ldc < N) return DGDM_ERR_INVALID_ARG;
  const dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (accumulate)
    hipLaunchKernelGGL((k_gemm3_rows<true, true>), grid, dim3(256), 0, s, A, lda, W, ldw, C, ldc, M, N, K, bias);
  else
    hipLaunchKernelGGL((k_gemm3_rows<true, false>), grid, dim3(256), 0, s, A, lda, W, ldw, C, ldc, M, N, K, bias);
  return dgdm_launch_status();
}

extern "C" int dgdm_gemm_nn_bf16x3(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int32_t M, int32_t N,
                                   int32_t Kout, int32_t accumulate, void* stream) {
  if (M < 0 || N < 0 || Kout < 0) return DGDM_ERR_INVALID_ARG;
  if (M == 0 || Kout == 0) return DGDM_OK;
  if (!A || !W || !C) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_ERR_UNSUPPORTED;
  if ((N & 3) || (Kout & 3) || (lda & 3) || (ldw & 3) || lda < N || ldw < Kout || ldc < Kout || !dgdm_aligned16(A) || !dgdm_aligned16(W))
    return DGDM_ERR_UNSUPPORTED;
  const dim3 grid((M + BM - 1) / BM, (Kout + BN - 1) / BN);
  const float* nobias = nullptr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (accumulate)
    hipLaunchKernelGGL((k_gemm3_rows<false, true>), grid, dim3(256), 0, s, A, lda, W, ldw, C, ldc, M, Kout, N, nobias);
  else
    hipLaunchKernelGGL((k_gemm3_rows<false, false>), grid, dim3(256), 0, s, A, lda, W, ldw, C, ldc, M, Kout, N, nobias);
  return dgdm_launch_status();
}

extern "C" size_t dgdm_gemm_tn_bf16x3_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t with_bias) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int chunk = tn3_chunk_rows(M, N, K);
  const int nchunks = (M + chunk - 1) / chunk;
  const size_t width = (size_t)N * K + (with_bias ? N : 0);
  return (size_t)(nchunks + (nchunks > 32 ? TN3_STAGE_SLOTS : 0)) * width * sizeof(float);
}

extern "C" int dgdm_gemm_tn_bf16x3(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW, int64_t lddw, float* db,
                                   int32_t M, int32_t N, int32_t K, void* workspace, size_t workspace_bytes, void* stream_) {
  if (M < 0 || N < 0 || K < 0) return DGDM_ERR_INVALID_ARG;
  if (N == 0 || K == 0) return DGDM_OK;
  if (!dW || lddw < K) return DGDM_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  if (M == 0) {
    (void)hipMemset2DAsync(dW, (size_t)lddw * sizeof(float), 0, (size_t)K * sizeof(float), (size_t)N, s);
    if (db) (void)hipMemsetAsync(db, 0, sizeof(float) * N, s);
    return dgdm_launch_status();
  }
  if (!dY || !X || !workspace) return DGDM_ERR_INVALID_ARG;
  if ((ldy & 3) || (ldx & 3) || (N & 3) || (K & 3) || ldy < N || ldx < K || !dgdm_aligned16(dY) || !dgdm_aligned16(X))
    return DGDM_ERR_UNSUPPORTED;
  const int chunk = tn3_chunk_rows(M, N, K);
  const int nchunks = (M + chunk - 1) / chunk;
  const int64_t width = (int64_t)N * K + (db ? N : 0);
  if (width > 0x7fffffffLL) return DGDM_ERR_UNSUPPORTED;
  const bool staged = nchunks > 32;
  if (workspace_bytes < (size_t)(nchunks + (staged ? TN3_STAGE_SLOTS : 0)) * width * sizeof(float)) return DGDM_ERR_WORKSPACE;
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(k_gemm3_tn_partial, dim3((N + BM - 1) / BM, (K + BN - 1) / BN, nchunks), dim3(256), 0, s, dY, ldy, X, ldx, M, N, K,
                     chunk, db ? 1 : 0, partial);
  const float* fin = partial;
  int slots = nchunks;
  if (staged) {
    float* stage = partial + (int64_t)nchunks * width;
    const int64_t per = (nchunks + TN3_STAGE_SLOTS - 1) / TN3_STAGE_SLOTS;
    slots = (int)((nchunks + per - 1) / per);
    hipLaunchKernelGGL(k_colsum, dim3((unsigned)((width + 63) / 64), slots), dim3(256), 0, s, partial, (int64_t)nchunks, (int)width, per,
                       stage, 0, (float*)nullptr);
    fin = stage;
  }
  hipLaunchKernelGGL(k_gemm3_tn_final, dim3((unsigned)((width + 63) / 64)), dim3(256), 0, s, fin, slots, width, N, K, dW, lddw, db);
  return dgdm_launch_status();
}
