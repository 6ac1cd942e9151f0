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
// Tiling: 128x128 output tile per 256-thread workgroup, 4 waves x (2x2) 32x32 tiles.  The reduction
// advances in 16-deep stages (one MFMA k step) through a double-buffered LDS image with ONE barrier
// per stage: while a wave issues the 24 MFMAs of stage s from LDS[s & 1] it splits stage s+1 from
// registers (v_cvt_pk_bf16_f32 + packed subtract, 4.5 VALU per element -- they issue in the 24 free
// cycles of each 32-cycle MFMA) into LDS[(s+1) & 1], and the global loads of stage s+3 are in flight
// in a third register set.  (A first version with one LDS buffer and a convert phase between two
// barriers ran the two resident workgroups of a CU in lockstep: PMC showed the matrix pipe 39 % busy.)
// LDS image of an operand stage: three planes [3][128 rows][16 k] of bf16 with 48-byte rows, which
// makes the ds_read_b128 operand reads of all four lane groups conflict-free; 72 KiB per workgroup,
// two workgroups per CU.
// Non-finite inputs: Inf - Inf in the residual turns an Inf operand into NaN (the model rejects
// non-finite features before any GEMM, models/dgdm_model.py:278-283 in the reference).
#include "common.hpp"
#include "colsum.hpp"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BM = 128, BN = 128, KS = 16;   // KS = reduction depth of a stage
constexpr int RSB = 48;                      // bytes per LDS row: 16 bf16 + 16 B pad
constexpr int PLANE = BM * RSB;              // 6144
constexpr int OPERAND = 3 * PLANE;           // 18432
constexpr int STAGE = 2 * OPERAND;           // 36864 (A image, then B image)

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

// ---- one stage (16 k) of an operand whose global layout has the reduction index contiguous:
// element (o, k) at P[(o0 + o) * ld + k], 128 output rows.  2 float4 per thread.
struct RowStage {
  float4 v[2];
  unsigned ok;
  template <bool BIAS>
  __device__ __forceinline__ void store_one(char* __restrict__ S, int tid, int j, float (&)[4]) const {
    const int r = (tid >> 2) + 64 * j, c4 = tid & 3;
    const bool g = (ok >> j) & 1u;
    uint2 h, m, l;
    split_pair(g ? v[j].x : 0.f, g ? v[j].y : 0.f, &h.x, &m.x, &l.x);
    split_pair(g ? v[j].z : 0.f, g ? v[j].w : 0.f, &h.y, &m.y, &l.y);
    char* dst = S + r * RSB + 8 * c4;
    *reinterpret_cast<uint2*>(dst) = h;
    *reinterpret_cast<uint2*>(dst + PLANE) = m;
    *reinterpret_cast<uint2*>(dst + 2 * PLANE) = l;
  }
};

// ---- one stage of an operand whose global layout has the OUTPUT index contiguous: element (o, k) at
// P[k * ld + o0 + o]; transposed into the [o][k] image.  A thread holds the two k rows of one k pair
// for 4 output columns, so a column's pair packs into one dword: ds_write_b32 at [col][pair], banks
// {pair} + {0,16} per 32-lane half = 2-way, which ds_write_b32 absorbs.
struct ColStage {
  float4 v[2];
  unsigned ok;
  // half: columns 2*half, 2*half + 1.  BIAS: also add the two (masked) values of each column to bs
  // (sum over this thread's valid k rows: the bias gradient of the dW kernel).
  template <bool BIAS>
  __device__ __forceinline__ void store_one(char* __restrict__ S, int tid, int half, float (&bs)[4]) const {
    const bool g0 = ok & 1u, g1 = (ok >> 1) & 1u;
    const float a4[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, b4[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * half + jj;
      unsigned h, m, l;
      const float x0 = g0 ? a4[j] : 0.f, x1 = g1 ? b4[j] : 0.f;
      if (BIAS) bs[j] += x0 + x1;
      split_pair(x0, x1, &h, &m, &l);
      char* dst = S + (4 * (tid >> 3) + j) * RSB + 4 * (tid & 7);
      *reinterpret_cast<unsigned*>(dst) = h;
      *reinterpret_cast<unsigned*>(dst + PLANE) = m;
      *reinterpret_cast<unsigned*>(dst + 2 * PLANE) = l;
    }
  }
};

// Stage loaders: per-thread addresses are set up once and advanced by a constant per stage (the
// straightforward index arithmetic costs ~10 VALU per load in 64-bit multiplies, and VALU slots are
// what this kernel lives on).  Stages are loaded strictly in order.  Loads past the end of the
// reduction range keep a valid address and are masked (ok = 0), so the main loop needs no branches.
struct RowLoader {
  const float* p[2];   // row base of this thread's two rows (clamped to the last valid row)
  int k, kend;         // k of this thread's float4 in the next stage to load
  unsigned rowok;
  __device__ __forceinline__ void init(const float* __restrict__ P, int64_t ld, int o0, int on, int kbeg, int kend_, int /*klim*/, int tid) {
    rowok = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = o0 + (tid >> 2) + 64 * j;
      p[j] = P + (int64_t)(row < on ? row : on - 1) * ld;
      rowok |= row < on ? (1u << j) : 0u;
    }
    k = kbeg + 4 * (tid & 3);
    kend = kend_;
  }
  __device__ __forceinline__ void next(RowStage& R) {
    const bool kin = k < kend;
    const int kc = kin ? k : kend - 4;
    R.v[0] = *reinterpret_cast<const float4*>(p[0] + kc);
    R.v[1] = *reinterpret_cast<const float4*>(p[1] + kc);
    R.ok = kin ? rowok : 0u;
    k += KS;
  }
};

// RowLoader over an operand whose reduction range is the concatenation of two matrices: k in [0, ksplit) comes from
// the first (P, ld), k in [ksplit, kend) from the second (P2, ld2).  ksplit is a multiple of 4 (a float4 never straddles).
struct RowLoaderSplit : RowLoader {
  const float* q[2];   // row base in the second matrix, shifted by -ksplit so that the same k indexes it
  int ksplit;
  __device__ __forceinline__ void init2(const float* __restrict__ P2, int64_t ld2, int o0, int on, int ksplit_, int tid) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = o0 + (tid >> 2) + 64 * j;
      q[j] = P2 + (int64_t)(row < on ? row : on - 1) * ld2 - ksplit_;
    }
    ksplit = ksplit_;
  }
  __device__ __forceinline__ void next(RowStage& R) {
    const bool kin = k < kend;
    const int kc = kin ? k : kend - 4;
    const bool first = kc < ksplit;
    R.v[0] = *reinterpret_cast<const float4*>((first ? p[0] : q[0]) + kc);
    R.v[1] = *reinterpret_cast<const float4*>((first ? p[1] : q[1]) + kc);
    R.ok = kin ? rowok : 0u;
    k += KS;
  }
};

struct ColLoader {
  const float* p[2];   // address of this thread's float4 in k rows k, k+1 of the next stage (clamped below klim)
  int k, kend, klim;   // klim: number of k rows that exist in memory (>= kend)
  int64_t step;
  bool colok;
  __device__ __forceinline__ void init(const float* __restrict__ P, int64_t ld, int o0, int on, int kbeg, int kend_, int klim_, int tid) {
    const int col = o0 + 4 * (tid >> 3);
    colok = col < on;
    k = kbeg + 2 * (tid & 7);
    kend = kend_;
    klim = klim_;
    step = (int64_t)KS * ld;
#pragma unroll
    for (int i = 0; i < 2; ++i) p[i] = P + (int64_t)(k + i < klim ? k + i : klim - 1) * ld + (colok ? col : on - 4);
  }
  __device__ __forceinline__ void next(ColStage& R) {
    R.v[0] = *reinterpret_cast<const float4*>(p[0]);
    R.v[1] = *reinterpret_cast<const float4*>(p[1]);
    R.ok = colok ? ((k < kend ? 1u : 0u) | (k + 1 < kend ? 2u : 0u)) : 0u;
    p[0] += (k + KS < klim) ? step : 0;       // stop advancing at the end of memory; those stages are masked anyway
    p[1] += (k + 1 + KS < klim) ? step : 0;
    k += KS;
  }
};

template <bool KC> struct StageOf { typedef RowStage type; typedef RowLoader loader; };
template <> struct StageOf<false> { typedef ColStage type; typedef ColLoader loader; };

// operand fragments of one stage: planes h, m, l of the wave's two 32-row sub-tiles
struct Frag {
  bf16x8 f[3][2];
  __device__ __forceinline__ void read(const char* __restrict__ img, int row0, int lane) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        f[p][t] = *reinterpret_cast<const bf16x8*>(img + p * PLANE + (row0 + t * 32 + (lane & 31)) * RSB + 16 * (lane >> 5));
  }
};

__device__ __forceinline__ f32x16 mma6(const Frag& a, const Frag& b, int mt, int nt, f32x16 c) {
  c = mfma_bf(a.f[1][mt], b.f[1][nt], c);   // smallest terms first
  c = mfma_bf(a.f[0][mt], b.f[2][nt], c);
  c = mfma_bf(a.f[2][mt], b.f[0][nt], c);
  c = mfma_bf(a.f[0][mt], b.f[1][nt], c);
  c = mfma_bf(a.f[1][mt], b.f[0][nt], c);
  c = mfma_bf(a.f[0][mt], b.f[0][nt], c);
  return c;
}

// The reduction loop shared by the three contractions.
//   acc[mt][nt] += sum over k in [kbeg, kend) of A(a0 + wave rows, k) * B(b0 + wave cols, k)
// A_KC / B_KC: the operand's global layout has k contiguous (RowStage) or the output index (ColStage).
// BIAS: also accumulate per-thread column sums of the A operand (ColStage only) into bs.
template <bool A_KC, bool B_KC, bool BIAS, bool B_SPLIT = false>
__device__ __forceinline__ void mainloop3(const float* __restrict__ A, int64_t lda, int a0, int an, const float* __restrict__ B,
                                          int64_t ldb, int b0, int bn, int kbeg, int kend, int klim, char* __restrict__ smem,
                                          f32x16 (&acc)[2][2], int live_m, int live_n, float (&bs)[4],
                                          const float* __restrict__ B2 = nullptr, int64_t ldb2 = 0, int bsplit = 0) {
  static_assert(!B_SPLIT || B_KC, "the two-matrix operand is implemented for the k-contiguous layout");
  typedef typename StageOf<A_KC>::type SA;
  typedef typename StageOf<B_KC>::type SB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int arow0 = (wave >> 1) * 64, brow0 = (wave & 1) * 64;
  const int nst = (kend - kbeg + KS - 1) / KS;
  const bool full = live_m == 2 && live_n == 2;
  SA ra0, ra1, ra2;
  SB rb0, rb1, rb2;
  typename StageOf<A_KC>::loader la;
  typename std::conditional<B_SPLIT, RowLoaderSplit, typename StageOf<B_KC>::loader>::type lb;
  la.init(A, lda, a0, an, kbeg, kend, klim, tid);
  lb.init(B, ldb, b0, bn, kbeg, kend, klim, tid);
  if constexpr (B_SPLIT) lb.init2(B2, ldb2, b0, bn, bsplit, tid);
  la.next(ra0); lb.next(rb0);
  la.next(ra1); lb.next(rb1);
  la.next(ra2); lb.next(rb2);
  float unused[4];
  ra0.template store_one<BIAS>(smem, tid, 0, bs); ra0.template store_one<BIAS>(smem, tid, 1, bs);
  rb0.template store_one<false>(smem + OPERAND, tid, 0, unused); rb0.template store_one<false>(smem + OPERAND, tid, 1, unused);
  __syncthreads();

  // Issue order of one stage.  A wave issues in order, and a wave stalled at an MFMA that waits for the
  // matrix pipe cannot issue the VALU work behind it -- so the split is placed in the issue slots
  // between consecutive MFMAs (each holds the port for 8 of the 32 pipe cycles): fragment reads
  // and the global loads first, a block of VALU under the LDS latency, then MFMA : VALU : DS-write
  // round robin.  Without this the compiler emits MFMA runs followed by VALU runs and the two
  // workgroups of a CU fall into lockstep (both in their MFMA run, then both in their VALU run).
#define DGDM_STAGE_SCHEDULE                                            \
  __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);                  \
  __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                   \
  __builtin_amdgcn_sched_group_barrier(0x002, 20, 0);                  \
  _Pragma("unroll") for (int g__ = 0; g__ < 24; ++g__) {               \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                 \
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                 \
  }

  // stage s: MFMAs from LDS[s & 1]; split set `cv` (stage s+1) into LDS[(s+1) & 1]; reload set `ld` with stage s+3
#define DGDM_STAGE(s_, ld_a, ld_b, cv_a, cv_b)                                                     \
  {                                                                                                \
    const int s__ = (s_);                                                                          \
    const char* cur = smem + (s__ & 1) * STAGE;                                                    \
    char* nxt = smem + ((s__ & 1) ^ 1) * STAGE;                                                    \
    la.next(ld_a);                                                                                 \
    lb.next(ld_b);                                                                                 \
    Frag fa, fb;                                                                                   \
    fa.read(cur, arow0, lane);                                                                     \
    fb.read(cur + OPERAND, brow0, lane);                                                           \
    if (full) {                                                                                    \
      acc[0][0] = mma6(fa, fb, 0, 0, acc[0][0]);                                                   \
      cv_a.template store_one<BIAS>(nxt, tid, 0, bs);                                              \
      acc[0][1] = mma6(fa, fb, 0, 1, acc[0][1]);                                                   \
      cv_a.template store_one<BIAS>(nxt, tid, 1, bs);                                              \
      acc[1][0] = mma6(fa, fb, 1, 0, acc[1][0]);                                                   \
      cv_b.template store_one<false>(nxt + OPERAND, tid, 0, unused);                               \
      acc[1][1] = mma6(fa, fb, 1, 1, acc[1][1]);                                                   \
      cv_b.template store_one<false>(nxt + OPERAND, tid, 1, unused);                               \
      DGDM_STAGE_SCHEDULE                                                                          \
    } else {                                                                                       \
      if (live_m > 0 && live_n > 0) acc[0][0] = mma6(fa, fb, 0, 0, acc[0][0]);                     \
      if (live_m > 0 && live_n > 1) acc[0][1] = mma6(fa, fb, 0, 1, acc[0][1]);                     \
      if (live_m > 1 && live_n > 0) acc[1][0] = mma6(fa, fb, 1, 0, acc[1][0]);                     \
      if (live_m > 1 && live_n > 1) acc[1][1] = mma6(fa, fb, 1, 1, acc[1][1]);                     \
      cv_a.template store_one<BIAS>(nxt, tid, 0, bs); cv_a.template store_one<BIAS>(nxt, tid, 1, bs);  \
      cv_b.template store_one<false>(nxt + OPERAND, tid, 0, unused);                               \
      cv_b.template store_one<false>(nxt + OPERAND, tid, 1, unused);                               \
    }                                                                                              \
    __syncthreads();                                                                               \
  }

  for (int s = 0; s < nst; s += 3) {
    DGDM_STAGE(s, ra0, rb0, ra1, rb1)
    if (s + 1 >= nst) break;
    DGDM_STAGE(s + 1, ra1, rb1, ra2, rb2)
    if (s + 2 >= nst) break;
    DGDM_STAGE(s + 2, ra2, rb2, ra0, rb0)
  }
#undef DGDM_STAGE
#undef DGDM_STAGE_SCHEDULE
}

template <bool B_KCONTIG, bool ACCUM, bool B_SPLIT = false>
__global__ __launch_bounds__(256, 2) void k_gemm3_rows(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                       int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                       const float* __restrict__ bias, const float* __restrict__ B2, int64_t ldb2,
                                                       int bsplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
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
  float nobs[4] = {0.f, 0.f, 0.f, 0.f};
  mainloop3<true, B_KCONTIG, false, B_SPLIT>(A, lda, m0, M, B, ldb, n0, N, 0, K, K, smem, acc, live_m, live_n, nobs, B2, ldb2, bsplit);

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

// dW partial: tile (n0, kk0) of [N x K], rows [mc*chunk, (mc+1)*chunk); partial row of chunk mc =
// [N*K dW elements | N bias sums (if with_bias)], as k_gemm_tn_partial of gemm.hip.
template <bool BIAS>
__global__ __launch_bounds__(256, 2) void k_gemm3_tn_partial(const float* __restrict__ dY, int64_t ldy, const float* __restrict__ X,
                                                             int64_t ldx, int M, int N, int K, int chunk, int with_bias,
                                                             float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * BM, kk0 = blockIdx.y * BN, mc = blockIdx.z;
  const int mbeg = mc * chunk, mend = min(M, mbeg + chunk);
  const int wr = wave >> 1, wc = wave & 1;
  const int live_m = min(2, max(0, (N - (n0 + wr * 64) + 31) / 32)), live_n = min(2, max(0, (K - (kk0 + wc * 64) + 31) / 32));
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  mainloop3<false, false, BIAS>(dY, ldy, n0, N, X, ldx, kk0, K, mbeg, mend, M, smem, acc, live_m, live_n, bs);

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
  if (BIAS && blockIdx.y == 0) {  // the 8 threads (tid & 7) of one column group hold the k pairs of the same 4 columns
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bs[q] += __shfl_xor(bs[q], 1, 64);
      bs[q] += __shfl_xor(bs[q], 2, 64);
      bs[q] += __shfl_xor(bs[q], 4, 64);
    }
    const int n = n0 + 4 * (tid >> 3);
    if ((tid & 7) == 0 && n < N) *reinterpret_cast<float4*>(&P[(int64_t)N * K + n]) = make_float4(bs[0], bs[1], bs[2], bs[3]);
  }
}

// dW[n][k] / db[n] = sum over slots of in[slot][n*K + k] / in[slot][N*K + n]; fixed order: thread (column, part)
// adds slots part, part + PARTS, ... in order, the PARTS partial sums are added in order.  256 / PARTS columns per
// block: 4 parts for a few chunks, 16 parts when a small output was cut into hundreds of row chunks (one launch
// instead of a staged column sum plus a final pass).
template <int PARTS>
__global__ __launch_bounds__(256) void k_gemm3_tn_final(const float* __restrict__ in, int slots, int64_t width, int N, int K,
                                                        float* __restrict__ dW, int64_t lddw, float* __restrict__ db, int K0,
                                                        float* __restrict__ dW1, int64_t ld1) {
  constexpr int COLS = 256 / PARTS;
  const int c = threadIdx.x % COLS, part = threadIdx.x / COLS;
  const int64_t col = (int64_t)blockIdx.x * COLS + c;
  float a0 = 0.f, a1 = 0.f;
  if (col < width) {
    int s = part;
    for (; s + PARTS < slots; s += 2 * PARTS) {       // two independent loads in flight
      a0 += in[(int64_t)s * width + col];
      a1 += in[(int64_t)(s + PARTS) * width + col];
    }
    if (s < slots) a0 += in[(int64_t)s * width + col];
  }
  __shared__ float sm[PARTS][COLS];
  sm[part][c] = a0 + a1;
  __syncthreads();
  if (part == 0 && col < width) {
    float t = sm[0][c];
#pragma unroll
    for (int p = 1; p < PARTS; ++p) t += sm[p][c];
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

constexpr int LDS_BYTES = 2 * STAGE;   // 73728: above the 64 KiB static limit, so dynamic

int tn3_chunk_rows(int M, int N, int K) {
  // 64-bit throughout: N, K, M up to INT32_MAX must not overflow (the tile count of a 2^31 x 2^31 problem does not fit an int;
  // found by the argument battery of tests/test_abi.py: a zero tile count divided)
  const int64_t tiles = (((int64_t)N + BM - 1) / BM) * (((int64_t)K + BN - 1) / BN);
  int64_t want = (DGDM_TN_WANT + tiles / 2) / tiles;   // workgroups per problem: see common.hpp
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  int64_t chunk = ((int64_t)M + want - 1) / want;
  chunk = (chunk + 2 * KS - 1) / (2 * KS) * (2 * KS);
  if (chunk < 8 * KS) chunk = 8 * KS;
  return (int)(chunk > 0x7fffff00 ? 0x7fffff00 : chunk);
}

// 72 KiB of dynamic LDS needs the opt-in once per kernel (per process; one device per process)
template <typename Kern>
int allow_big_lds(Kern kern) {
  static int status = 1;   // 1 = not asked yet
  if (status == 1)
    status = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess
                 ? DGDM_OK : DGDM_ERR_LAUNCH;
  return status;
}

}  // namespace

template <typename Kern>
static int launch_rows(Kern kern, dim3 grid, hipStream_t s, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                       int M, int N, int K, const float* bias, const float* B2 = nullptr, int64_t ldb2 = 0, int bsplit = 0) {
  if (allow_big_lds(kern) != DGDM_OK) return DGDM_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, s, A, lda, B, ldb, C, ldc, M, N, K, bias, B2, ldb2, bsplit);
  return dgdm_launch_status();
}

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
  return accumulate ? launch_rows(k_gemm3_rows<true, true>, grid, s, A, lda, W, ldw, C, ldc, M, N, K, bias)
                    : launch_rows(k_gemm3_rows<true, false>, grid, s, A, lda, W, ldw, C, ldc, M, N, K, bias);
}

// C = A . [W0 | W1]^T (+ bias): the weight's K columns come from two matrices, [0, K0) from W0 and [K0, K) from W1
extern "C" int dgdm_gemm_nt_split_bf16x3(const float* A, int64_t lda, const float* W0, int64_t ldw0, int32_t K0, const float* W1,
                                         int64_t ldw1, const float* bias, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                                         int32_t accumulate, void* stream) {
  if (M < 0 || N < 0 || K < 0 || K0 <= 0 || K0 >= K) return DGDM_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return DGDM_OK;
  if (!A || !W0 || !W1 || !C) return DGDM_ERR_INVALID_ARG;
  if ((K & 3) || (K0 & 3) || (lda & 3) || (ldw0 & 3) || (ldw1 & 3) || !dgdm_aligned16(A) || !dgdm_aligned16(W0) || !dgdm_aligned16(W1))
    return DGDM_ERR_UNSUPPORTED;
  if (lda < K || ldw0 < K0 || ldw1 < K - K0 || ldc < N) return DGDM_ERR_INVALID_ARG;
  const dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return accumulate ? launch_rows(k_gemm3_rows<true, true, true>, grid, s, A, lda, W0, ldw0, C, ldc, M, N, K, bias, W1, ldw1, K0)
                    : launch_rows(k_gemm3_rows<true, false, true>, grid, s, A, lda, W0, ldw0, C, ldc, M, N, K, bias, W1, ldw1, K0);
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
  return accumulate ? launch_rows(k_gemm3_rows<false, true>, grid, s, A, lda, W, ldw, C, ldc, M, Kout, N, nobias)
                    : launch_rows(k_gemm3_rows<false, false>, grid, s, A, lda, W, ldw, C, ldc, M, Kout, N, nobias);
}

extern "C" size_t dgdm_gemm_tn_bf16x3_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t with_bias) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int chunk = tn3_chunk_rows(M, N, K);
  const int nchunks = (int)(((int64_t)M + chunk - 1) / chunk);
  const size_t width = (size_t)N * K + (with_bias ? N : 0);
  return (size_t)nchunks * width * sizeof(float);
}

static int tn_impl(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW, int64_t lddw, int32_t K0, float* dW1, int64_t ld1,
                   float* db, int32_t M, int32_t N, int32_t K, void* workspace, size_t workspace_bytes, void* stream_,
                   bool partial_only = false) {
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
  const int chunk = tn3_chunk_rows(M, N, K);
  const int nchunks = (int)(((int64_t)M + chunk - 1) / chunk);
  const int64_t width = (int64_t)N * K + (db ? N : 0);
  if (width > 0x7fffffffLL) return DGDM_ERR_UNSUPPORTED;
  if (workspace_bytes < (size_t)nchunks * width * sizeof(float)) return DGDM_ERR_WORKSPACE;
  float* partial = static_cast<float*>(workspace);
  const dim3 grid((N + BM - 1) / BM, (K + BN - 1) / BN, nchunks);
  if (db) {
    if (allow_big_lds(k_gemm3_tn_partial<true>) != DGDM_OK) return DGDM_ERR_LAUNCH;
    hipLaunchKernelGGL(k_gemm3_tn_partial<true>, grid, dim3(256), LDS_BYTES, s, dY, ldy, X, ldx, M, N, K, chunk, 1, partial);
  } else {
    if (allow_big_lds(k_gemm3_tn_partial<false>) != DGDM_OK) return DGDM_ERR_LAUNCH;
    hipLaunchKernelGGL(k_gemm3_tn_partial<false>, grid, dim3(256), LDS_BYTES, s, dY, ldy, X, ldx, M, N, K, chunk, 0, partial);
  }
  if (partial_only) return dgdm_launch_status();   // the caller reduces the chunk partials later (dgdm_gemm_tn_reduce_many)
  if (nchunks > 32)
    hipLaunchKernelGGL(k_gemm3_tn_final<16>, dim3((unsigned)((width + 15) / 16)), dim3(256), 0, s, partial, nchunks, width, N, K, dW, lddw, db, K0, dW1, ld1);
  else
    hipLaunchKernelGGL(k_gemm3_tn_final<4>, dim3((unsigned)((width + 63) / 64)), dim3(256), 0, s, partial, nchunks, width, N, K, dW, lddw, db, K0, dW1, ld1);
  return dgdm_launch_status();
}

extern "C" int dgdm_gemm_tn_bf16x3(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW, int64_t lddw, float* db, int32_t M,
                                   int32_t N, int32_t K, void* workspace, size_t workspace_bytes, void* stream) {
  return tn_impl(dY, ldy, X, ldx, dW, lddw, K, nullptr, 0, db, M, N, K, workspace, workspace_bytes, stream);
}

extern "C" int dgdm_gemm_tn_partial_bf16x3(const float* dY, int64_t ldy, const float* X, int64_t ldx, int32_t with_bias, int32_t M, int32_t N,
                                           int32_t K, void* workspace, size_t workspace_bytes, void* stream) {
  float dummy;   // tn_impl only checks the destination pointers for presence when it skips the final reduction
  return tn_impl(dY, ldy, X, ldx, &dummy, K, K, nullptr, 0, with_bias ? &dummy : nullptr, M, N, K, workspace, workspace_bytes, stream, true);
}

// number of row chunks (= partial slots) dgdm_gemm_tn_*_bf16x3 / _f16x2 cut M into (same rule for both)
extern "C" int32_t dgdm_gemm_tn_chunks(int32_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int chunk = tn3_chunk_rows(M, N, K);
  return (int32_t)(((int64_t)M + chunk - 1) / chunk);
}

extern "C" int dgdm_gemm_tn_split_bf16x3(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW0, int64_t ld0, int32_t K0, float* dW1,
                                   int64_t ld1, float* db, int32_t M, int32_t N, int32_t K, void* workspace, size_t workspace_bytes, void* stream) {
  return tn_impl(dY, ldy, X, ldx, dW0, ld0, K0, dW1, ld1, db, M, N, K, workspace, workspace_bytes, stream);
}
