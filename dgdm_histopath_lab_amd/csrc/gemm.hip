// K3: fp32 MFMA GEMMs for the dense node-feature x weight contractions of the hot path
// (every nn.Linear of models/encoders.py, core/graph_layers.py, core/attention.py,
// core/diffusion.py).  The shapes are tall-skinny: M = nodes of the batch (5k .. 400k), K and N
// <= 1024.  Exact fp32: v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD = the fp32 matrix peak).
//
//   dgdm_gemm_nt : C[M,N] (+)= A[M,K] . W[N,K]^T + bias      forward of nn.Linear
//   dgdm_gemm_nn : C[M,K] (+)= A[M,N] . W[N,K]               dX of nn.Linear
//   dgdm_gemm_tn : dW[N,K]  = dY[M,N]^T . X[M,K],  db[N] = colsum(dY)   (reduction over M)
//
// nt/nn share one kernel: a 128x128 output tile per 256-thread workgroup (4 waves, 64x64 each =
// 2x2 MFMA tiles), K swept in 32-deep blocks staged through LDS with the reduction index
// contiguous ([row][k], rows padded to 36 floats => conflict-free ds_read_b128), the next block's
// global loads held in registers under the current block's MFMAs.  Each lane reads 4 consecutive k
// per operand row with one b128 and feeds them to 4 MFMA steps; both operands use the same k
// permutation, so the sum is unchanged.
// tn: the reduction runs over M, so each workgroup takes one 128x128 tile of dW and one chunk of
// rows and writes a partial tile; a second kernel sums the partials in chunk order (no atomics,
// bitwise reproducible).  A virtual column of ones appended to X makes column K of the result the
// bias gradient.
#include "common.hpp"
#include "colsum.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4;  // LDS row = 36 floats (144 B)

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// one 32-deep block: acc[mt][nt] += As[wave rows][k] * Bs[wave cols][k].
// live_m / live_n (wave-uniform, 0..2): how many of the wave's two 32-row / 32-column sub-tiles
// intersect the matrix; empty ones are skipped (ragged edges of small outputs).
__device__ __forceinline__ void mma_block(const float* __restrict__ As, const float* __restrict__ Bs, int arow0, int brow0,
                                          int lane, f32x16 (&acc)[2][2], int live_m = 2, int live_n = 2) {
  const int i = lane & 31, kh = lane >> 5;
  if (live_m == 0 || live_n == 0) return;
#pragma unroll
  for (int s = 0; s < BK / 8; ++s) {
    f32x4v a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a[t] = *reinterpret_cast<const f32x4v*>(&As[(arow0 + t * 32 + i) * LDT + 8 * s + 4 * kh]);
      b[t] = *reinterpret_cast<const f32x4v*>(&Bs[(brow0 + t * 32 + i) * LDT + 8 * s + 4 * kh]);
    }
    if (live_m == 2 && live_n == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma32(a[mt][e], b[nt][e], acc[mt][nt]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][0] = mfma32(a[0][e], b[0][e], acc[0][0]);
        if (live_n == 2) acc[0][1] = mfma32(a[0][e], b[1][e], acc[0][1]);
        if (live_m == 2) acc[1][0] = mfma32(a[1][e], b[0][e], acc[1][0]);
        if (live_m == 2 && live_n == 2) acc[1][1] = mfma32(a[1][e], b[1][e], acc[1][1]);
      }
    }
  }
}

// ---- staging helpers: a [rows x 32] tile whose global layout has the reduction index contiguous
struct RowTile {  // 128 rows x 32 k, 1024 float4 -> 4 per thread
  float4 v[4];
  unsigned okbits;  // validity of v[i]; applied in store() so the loads stay in flight under the MFMAs
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int row0, int nrows, int k0, int K, int tid) {
    okbits = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, r = idx >> 3, c4 = idx & 7;
      const int row = row0 + r, k = k0 + 4 * c4;
      // unconditional load from a clamped (always valid) address + register select: a predicated
      // load makes hipcc branch around it and drain vmcnt(0) per element
      const int rc = row < nrows ? row : nrows - 1, kc = k < K ? k : K - 4;
      v[i] = *reinterpret_cast<const float4*>(P + (int64_t)rc * ld + kc);
      okbits |= (row < nrows && k < K) ? (1u << i) : 0u;
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, r = idx >> 3, c4 = idx & 7;
      const bool ok = (okbits >> i) & 1u;
      *reinterpret_cast<float4*>(&S[r * LDT + 4 * c4]) =
          make_float4(ok ? v[i].x : 0.f, ok ? v[i].y : 0.f, ok ? v[i].z : 0.f, ok ? v[i].w : 0.f);
    }
  }
};
// ---- a [32 k x 128 cols] tile whose global layout has the OUTPUT index contiguous (needs a
// transpose into the [col][k] LDS image).
// Thread map: per iteration a wave covers 8 k-rows x 8 float4 columns (lane = kr_lo + 8*c4_lo), so
// a global request is 8 rows x 128 B and the transposed ds_write_b32 of one component lands on
// banks {kr} + {0,16}: 2-way, which ds_write_b32 absorbs (the 2-rows-x-32-columns map was 16-way).
struct ColTile {
  __device__ static __forceinline__ void map(int tid, int i, int* kr, int* c4) {
    const int lane = tid & 63, wave = tid >> 6;
    *kr = 8 * i + (lane & 7);
    *c4 = (lane >> 3) + 8 * wave;
  }
  float4 v[4];
  unsigned okbits;  // bit i: v[i] valid
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int k0, int K, int col0, int ncols, int tid) {
    okbits = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int kr, c4;
      map(tid, i, &kr, &c4);
      const int k = k0 + kr, col = col0 + 4 * c4;
      // ncols % 4 == 0 and col % 4 == 0: a float4 is entirely inside or entirely outside.
      const int kc = k < K ? k : K - 1, cc = col < ncols ? col : ncols - 4;
      v[i] = *reinterpret_cast<const float4*>(P + (int64_t)kc * ld + cc);
      okbits |= (k < K && col < ncols) ? (1u << i) : 0u;
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int kr, c4;
      map(tid, i, &kr, &c4);
      const bool ok = (okbits >> i) & 1u;
      S[(4 * c4 + 0) * LDT + kr] = ok ? v[i].x : 0.f;
      S[(4 * c4 + 1) * LDT + kr] = ok ? v[i].y : 0.f;
      S[(4 * c4 + 2) * LDT + kr] = ok ? v[i].z : 0.f;
      S[(4 * c4 + 3) * LDT + kr] = ok ? v[i].w : 0.f;
    }
  }
};

// C/D map of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
template <bool B_KCONTIG, bool ACCUM>
__global__ __launch_bounds__(256, 2) void k_gemm_rows(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                      int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                      const float* __restrict__ bias) {
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDT];
  float* As = smem;
  float* Bs = smem + BM * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int wr = wave >> 1, wc = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  RowTile ta;
  RowTile tbr;
  ColTile tbc;
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
    mma_block(As, Bs, wr * 64, wc * 64, lane, acc);
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
      if (ACCUM) {  // all reads first (clamped addresses), then one wait
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          old[r] = C[(int64_t)(row < M ? row : M - 1) * ldc + colc];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = acc[mt][nt][r] + bv + (ACCUM ? old[r] : 0.f);
    }
  // the stores follow once every output is finished in place: a store whose data register is recycled for the next value makes
  // hipcc wait vmcnt(0) in front of every store (~7 us per launch, tools/ubench/gemm_img_stamps.hip)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = n0 + wc * 64 + nt * 32 + j;
      const int rbase = m0 + wr * 64 + mt * 32 + 4 * hi;
      if (col >= N) continue;
      float* p = C + (int64_t)rbase * ldc + col;
      if (m0 + wr * 64 + mt * 32 + 32 <= M) {       // wave-uniform: all 32 rows exist
#pragma unroll
        for (int r = 0; r < 16; ++r) p[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = acc[mt][nt][r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (rbase + (r & 3) + 8 * (r >> 2) < M) p[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = acc[mt][nt][r];
      }
    }
}

// dW partial: tile (n0, kk0) of [N x K], rows [mc*chunk, (mc+1)*chunk).  Partial row of chunk mc:
// [N*K dW elements | N bias sums (if with_bias)].  The k-tile-0 workgroups also sum dY over their
// rows -- the dY tile is already in LDS as [n][m] -- so the bias gradient costs no extra pass over
// dY and no extra MFMA tile.
__global__ __launch_bounds__(256, 2) void k_gemm_tn_partial(const float* __restrict__ dY, int64_t ldy, const float* __restrict__ X,
                                                            int64_t ldx, int M, int N, int K, int chunk, int with_bias,
                                                            float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDT];
  float* Ys = smem;             // [n][m]
  float* Xs = smem + BM * LDT;  // [k][m]
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
  float bsum = 0.f;  // threads (2n, 2n+1): sum of dY[m, n0 + n] over the even / odd 16-m halves of every block
  ColTile ty, tx;
  ty.load(dY, ldy, mbeg, mend, n0, N, tid);
  tx.load(X, ldx, mbeg, mend, kk0, K, tid);
  for (int m = mbeg; m < mend; m += BK) {
    __syncthreads();
    ty.store(Ys, tid);
    tx.store(Xs, tid);
    __syncthreads();
    if (m + BK < mend) {
      ty.load(dY, ldy, m + BK, mend, n0, N, tid);
      tx.load(X, ldx, m + BK, mend, kk0, K, tid);
    }
    if (do_bias) {
      const float* row = &Ys[(tid >> 1) * LDT + 16 * (tid & 1)];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(row + 4 * q);
        bsum += (v.x + v.y) + (v.z + v.w);
      }
    }
    mma_block(Ys, Xs, wr * 64, wc * 64, lane, acc, live_m, live_n);
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
  if (do_bias) {
    bsum += __shfl_xor(bsum, 1, 64);
    const int n = n0 + (tid >> 1);
    if ((tid & 1) == 0 && n < N) P[(int64_t)N * K + n] = bsum;
  }
}

// dW[n][k] / db[n] = sum over slots of in[slot][n*K + k] / in[slot][N*K + n]; fixed order.
__global__ __launch_bounds__(256) void k_gemm_tn_final(const float* __restrict__ in, int slots, int64_t width, int N, int K,
                                                       float* __restrict__ dW, int64_t lddw, float* __restrict__ db, int K0,
                                                       float* __restrict__ dW1, int64_t ld1) {
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
    if (col < nk) {      // columns [0, K0) of dW go to dW, [K0, K) to dW1 (two parameters behind one contraction)
      const int64_t n = col / K;
      const int k = (int)(col % K);
      if (k < K0) dW[n * lddw + k] = t; else dW1[n * ld1 + (k - K0)] = t;
    } else {
      db[col - nk] = t;
    }
  }
}

constexpr int TN_STAGE_SLOTS = 16;   // more than 32 chunks: first fold them into <= 16 partial sums

int tn_chunk_rows(int M, int N, int K) {
  // enough workgroups to fill 256 CUs ~2x, chunks a multiple of BK rows, at most 256 chunks
  // 64-bit throughout (sizes up to INT32_MAX must not overflow: see gemm3.hip, tn3_chunk_rows)
  const int64_t tiles = (((int64_t)N + BM - 1) / BM) * (((int64_t)K + BN - 1) / BN);
  int64_t want = (496 + tiles / 2) / tiles;   // ~2 workgroups per CU, all resident in one round
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  int64_t chunk = ((int64_t)M + want - 1) / want;
  chunk = (chunk + BK - 1) / BK * BK;
  if (chunk < 4 * BK) chunk = 4 * BK;
  return (int)(chunk > 0x7fffff00 ? 0x7fffff00 : chunk);
}

}  // namespace

static int gemm_rows_check(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N,
                           int32_t K, const float* bias, int kdim_a, int kdim_b_contig) {
  if (M < 0 || N < 0 || K < 0) return DGDM_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return 1;
  if (!A || !B || !C) return DGDM_ERR_INVALID_ARG;
  if (K == 0) return DGDM_ERR_UNSUPPORTED;
  if ((K & 3) || (lda & 3) || (ldb & 3) || lda < kdim_a || !dgdm_aligned16(A) || !dgdm_aligned16(B)) return DGDM_ERR_UNSUPPORTED;
  if (ldc < N) return DGDM_ERR_INVALID_ARG;
  (void)bias; (void)kdim_b_contig;
  return DGDM_OK;
}

extern "C" int dgdm_gemm_nt(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                            int32_t M, int32_t N, int32_t K, int32_t accumulate, void* stream) {
  int rc = gemm_rows_check(A, lda, W, ldw, C, ldc, M, N, K, bias, K, 1);
  if (rc != DGDM_OK) return rc > 0 ? DGDM_OK : rc;
  if (ldw < K) return DGDM_ERR_INVALID_ARG;
  const dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  if (accumulate)
    hipLaunchKernelGGL((k_gemm_rows<true, true>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), A, lda, W, ldw, C, ldc, M, N, K, bias);
  else
    hipLaunchKernelGGL((k_gemm_rows<true, false>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), A, lda, W, ldw, C, ldc, M, N, K, bias);
  return dgdm_launch_status();
}

// C[M, Kout] (+)= A[M, N] . W[N, Kout]   (W row-major with row stride ldw >= Kout)
extern "C" int dgdm_gemm_nn(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int32_t M, int32_t N,
                            int32_t Kout, int32_t accumulate, void* stream) {
  // reduction index = N; output columns = Kout (loaded 4 at a time along W rows)
  if (M < 0 || N < 0 || Kout < 0) return DGDM_ERR_INVALID_ARG;
  if (M == 0 || Kout == 0) return DGDM_OK;
  if (!A || !W || !C) return DGDM_ERR_INVALID_ARG;
  if ((N & 3) || (Kout & 3) || (lda & 3) || (ldw & 3) || lda < N || ldw < Kout || ldc < Kout || !dgdm_aligned16(A) || !dgdm_aligned16(W))
    return DGDM_ERR_UNSUPPORTED;
  const dim3 grid((M + BM - 1) / BM, (Kout + BN - 1) / BN);
  const float* nobias = nullptr;
  if (accumulate)
    hipLaunchKernelGGL((k_gemm_rows<false, true>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), A, lda, W, ldw, C, ldc, M, Kout, N, nobias);
  else
    hipLaunchKernelGGL((k_gemm_rows<false, false>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), A, lda, W, ldw, C, ldc, M, Kout, N, nobias);
  return dgdm_launch_status();
}

extern "C" size_t dgdm_gemm_tn_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t with_bias) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int chunk = tn_chunk_rows(M, N, K);
  const int nchunks = (int)(((int64_t)M + chunk - 1) / chunk);
  const size_t width = (size_t)N * K + (with_bias ? N : 0);
  return (size_t)(nchunks + (nchunks > 32 ? TN_STAGE_SLOTS : 0)) * width * sizeof(float);
}

static int tn_impl(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW, int64_t lddw, int32_t K0, float* dW1, int64_t ld1,
                   float* db, int32_t M, int32_t N, int32_t K, void* workspace, size_t workspace_bytes, void* stream_) {
  if (M < 0 || N < 0 || K < 0) return DGDM_ERR_INVALID_ARG;
  if (N == 0 || K == 0) return DGDM_OK;
  if (K0 < 0 || K0 > K || (K0 > 0 && (!dW || lddw < K0)) || (K0 < K && (!dW1 || ld1 < K - K0))) return DGDM_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  if (M == 0) {
    dgdm_fill2d_async(dW, lddw, K0, N, s);
    dgdm_fill2d_async(dW1, ld1, K - K0, N, s);
    if (db) dgdm_fill_async(db, 0, sizeof(float) * N, s);
    return dgdm_launch_status();
  }
  if (!dY || !X || !workspace) return DGDM_ERR_INVALID_ARG;
  if ((ldy & 3) || (ldx & 3) || (N & 3) || (K & 3) || ldy < N || ldx < K || !dgdm_aligned16(dY) || !dgdm_aligned16(X))
    return DGDM_ERR_UNSUPPORTED;
  const int chunk = tn_chunk_rows(M, N, K);
  const int nchunks = (int)(((int64_t)M + chunk - 1) / chunk);
  const int64_t width = (int64_t)N * K + (db ? N : 0);
  if (width > 0x7fffffffLL) return DGDM_ERR_UNSUPPORTED;
  const bool staged = nchunks > 32;
  if (workspace_bytes < (size_t)(nchunks + (staged ? TN_STAGE_SLOTS : 0)) * width * sizeof(float)) return DGDM_ERR_WORKSPACE;
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(k_gemm_tn_partial, dim3((N + BM - 1) / BM, (K + BN - 1) / BN, nchunks), dim3(256), 0, s, dY, ldy, X, ldx, M, N, K,
                     chunk, db ? 1 : 0, partial);
  const float* fin = partial;
  int slots = nchunks;
  if (staged) {
    float* stage = partial + (int64_t)nchunks * width;
    const int64_t per = (nchunks + TN_STAGE_SLOTS - 1) / TN_STAGE_SLOTS;
    slots = (int)((nchunks + per - 1) / per);
    hipLaunchKernelGGL(k_colsum, dim3((unsigned)((width + 63) / 64), slots), dim3(256), 0, s, partial, (int64_t)nchunks, (int)width, per,
                       stage, 0, (float*)nullptr);
    fin = stage;
  }
  hipLaunchKernelGGL(k_gemm_tn_final, dim3((unsigned)((width + 63) / 64)), dim3(256), 0, s, fin, slots, width, N, K, dW, lddw, db, K0, dW1, ld1);
  return dgdm_launch_status();
}

extern "C" int dgdm_gemm_tn(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW, int64_t lddw, float* db, int32_t M,
                            int32_t N, int32_t K, void* workspace, size_t workspace_bytes, void* stream) {
  return tn_impl(dY, ldy, X, ldx, dW, lddw, K, nullptr, 0, db, M, N, K, workspace, workspace_bytes, stream);
}

extern "C" int dgdm_gemm_tn_split(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW0, int64_t ld0, int32_t K0, float* dW1,
                            int64_t ld1, float* db, int32_t M, int32_t N, int32_t K, void* workspace, size_t workspace_bytes, void* stream) {
  return tn_impl(dY, ldy, X, ldx, dW0, ld0, K0, dW1, ld1, db, M, N, K, workspace, workspace_bytes, stream);
}
