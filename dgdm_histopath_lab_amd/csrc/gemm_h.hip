// K3'': the three contractions of every nn.Linear of the hot path (forward, dX, dW/db) on the fp16 matrix pipe with
// fp32-level accuracy at HALF the matrix work of gemm3.hip's exact bf16 split.
//
// Every fp32 operand x is carried as two halfs, hi = fp16(x'), lo = fp16(x' - hi) with x' = x * 2^e (22 significand bits), and
// a product is the three terms a_hi.b_hi + a_hi.b_lo + a_lo.b_hi (the dropped a_lo.b_lo is below 2^-22 of the product), each an
// exact fp16 x fp16 product accumulated in fp32 by v_mfma_f32_32x32x16_f16: 3 MFMAs per product term instead of 6, two LDS planes
// instead of three.  What bf16 gives for free and fp16 does not is RANGE (gradients of 1e-7, un-normalised inputs above 65504), so
// every operand comes with its absolute maximum: a uint32 in device memory holding the float bits of max|x| (non-negative floats
// order like integers, so the kernels that PRODUCE the operand keep it with an integer atomic max -- order-free, hence bitwise
// reproducible; ops.py `AmaxArena`).  The kernel derives e = 14 - floor(log2(amax)) per operand (amax * 2^e in [2^14, 2^15): no
// overflow, and elements down to 2^-18 of the maximum keep their full 22 bits before fp16's subnormal floor is felt), scales on
// the way into LDS (a multiply by a power of two is exact) and multiplies the accumulators by 2^-(ea+eb) in the epilogue.
// An operand without a known maximum takes the bf16x3 kernels (ops.py decides): nothing here guesses a scale.
// Tiling, staging, the one-barrier-per-stage pipeline and the split-M weight-gradient scheme are gemm3.hip's.
#include "common.hpp"
#include "colsum.hpp"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BM = 128, BN = 128, KS = 16;   // KS = reduction depth of a stage
constexpr int RSB = 48;                      // bytes per LDS row: 16 bf16 + 16 B pad
constexpr int PLANE = BM * RSB;              // 6144
constexpr int OPERAND = 2 * PLANE;           // 12288 (hi plane, lo plane)
constexpr int STAGE = 2 * OPERAND;           // 24576 (A image, then B image)

__device__ __forceinline__ f32x16 mfma_hf(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// An amax "slot" is a GROUP of DGDM_AMAX_WAYS words, DGDM_AMAX_STRIDE words apart (one per 256-byte line): the thousands of
// workgroups of a producing kernel spread their atomic maxima over the ways (same-address atomics serialise in L2: 8192 of them
// cost ~50 us), a consumer takes the maximum of the ways.  Wave-uniform result.
__device__ __forceinline__ unsigned amax_group(const unsigned* __restrict__ g) {
  const int lane = threadIdx.x & 63;
  unsigned m = lane < DGDM_AMAX_WAYS ? g[lane * DGDM_AMAX_STRIDE] : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  return m;
}

// 2^e with amax * 2^e in [2^14, 2^15) from the float bits of amax; 1 for a zero / denormal maximum
__device__ __forceinline__ float scale_of(unsigned amax_bits) {
  const int E = (int)((amax_bits >> 23) & 0xffu) - 127;
  int e = (amax_bits & 0x7f800000u) ? 14 - E : 0;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return __uint_as_float((unsigned)(127 + e) << 23);
}

// hi + lo split of a pair of (already scaled) floats; each output dword holds the pair's halfs
__device__ __forceinline__ void split_pair(float a, float b, unsigned* h, unsigned* l) {
  const f16x2 hh = __builtin_convertvector(f32x2{a, b}, f16x2);
  const unsigned hb = __builtin_bit_cast(unsigned, hh);
  // lo = fp16(x - hi) by one mixed-precision FMA per value (reads hi as fp16, x as fp32; x - hi is exact in fp32, so the result is
  // bit for bit the two-step form's): 3 instructions per pair where hipcc's lowering of the plain expression takes 5
  unsigned lb;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(lb)
      : "v"(hb), "v"(a), "v"(b));
  *h = hb;
  *l = lb;
}

// ---- one stage (16 k) of an operand whose global layout has the reduction index contiguous:
// element (o, k) at P[(o0 + o) * ld + k], 128 output rows.  2 float4 per thread.
struct RowStage {
  float4 v[2];
  unsigned ok;
  template <bool BIAS>
  __device__ __forceinline__ void store_one(char* __restrict__ S, int tid, int j, float (&)[4], float sc) const {
    const int r = (tid >> 2) + 64 * j, c4 = tid & 3;
    const bool g = (ok >> j) & 1u;
    uint2 h, l;
    split_pair(g ? v[j].x * sc : 0.f, g ? v[j].y * sc : 0.f, &h.x, &l.x);
    split_pair(g ? v[j].z * sc : 0.f, g ? v[j].w * sc : 0.f, &h.y, &l.y);
    char* dst = S + r * RSB + 8 * c4;
    *reinterpret_cast<uint2*>(dst) = h;
    *reinterpret_cast<uint2*>(dst + PLANE) = l;
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
  __device__ __forceinline__ void store_one(char* __restrict__ S, int tid, int half, float (&bs)[4], float sc) const {
    const bool g0 = ok & 1u, g1 = (ok >> 1) & 1u;
    const float a4[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, b4[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * half + jj;
      unsigned h, l;
      const float x0 = g0 ? a4[j] : 0.f, x1 = g1 ? b4[j] : 0.f;
      if (BIAS) bs[j] += x0 + x1;           // bias gradient: column sums of the UNSCALED fp32 values
      split_pair(x0 * sc, x1 * sc, &h, &l);
      char* dst = S + (4 * (tid >> 3) + j) * RSB + 4 * (tid & 7);
      *reinterpret_cast<unsigned*>(dst) = h;
      *reinterpret_cast<unsigned*>(dst + PLANE) = l;
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

// operand fragments of one stage: planes hi, lo of the wave's two 32-row sub-tiles
struct Frag {
  f16x8 f[2][2];
  __device__ __forceinline__ void read(const char* __restrict__ img, int row0, int lane) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        f[p][t] = *reinterpret_cast<const f16x8*>(img + p * PLANE + (row0 + t * 32 + (lane & 31)) * RSB + 16 * (lane >> 5));
  }
};

__device__ __forceinline__ f32x16 mma3(const Frag& a, const Frag& b, int mt, int nt, f32x16 c) {
  c = mfma_hf(a.f[1][mt], b.f[0][nt], c);   // smaller terms first
  c = mfma_hf(a.f[0][mt], b.f[1][nt], c);
  c = mfma_hf(a.f[0][mt], b.f[0][nt], c);
  return c;
}

// The reduction loop shared by the three contractions.
//   acc[mt][nt] += sum over k in [kbeg, kend) of A(a0 + wave rows, k) * B(b0 + wave cols, k)
// A_KC / B_KC: the operand's global layout has k contiguous (RowStage) or the output index (ColStage).
// BIAS: also accumulate per-thread column sums of the A operand (ColStage only) into bs.
template <bool A_KC, bool B_KC, bool BIAS, bool B_SPLIT = false>
__device__ __forceinline__ void mainloop_h(const float* __restrict__ A, int64_t lda, int a0, int an, const float* __restrict__ B,
                                          int64_t ldb, int b0, int bn, int kbeg, int kend, int klim, char* __restrict__ smem,
                                          f32x16 (&acc)[2][2], int live_m, int live_n, float (&bs)[4], float sca, float scb,
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
  ra0.template store_one<BIAS>(smem, tid, 0, bs, sca); ra0.template store_one<BIAS>(smem, tid, 1, bs, sca);
  rb0.template store_one<false>(smem + OPERAND, tid, 0, unused, scb); rb0.template store_one<false>(smem + OPERAND, tid, 1, unused, scb);
  __syncthreads();

  // Issue order of one stage.  A wave issues in order, and a wave stalled at an MFMA that waits for the
  // matrix pipe cannot issue the VALU work behind it -- so the split is placed in the issue slots
  // between consecutive MFMAs (each holds the port for 8 of the 32 pipe cycles): fragment reads
  // and the global loads first, a block of VALU under the LDS latency, then MFMA : VALU : DS-write
  // round robin.  Without this the compiler emits MFMA runs followed by VALU runs and the two
  // workgroups of a CU fall into lockstep (both in their MFMA run, then both in their VALU run).
#define DGDM_STAGE_SCHEDULE                                            \
  __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                   \
  __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                   \
  __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);                  \
  _Pragma("unroll") for (int g__ = 0; g__ < 12; ++g__) {               \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                 \
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
      acc[0][0] = mma3(fa, fb, 0, 0, acc[0][0]);                                                   \
      cv_a.template store_one<BIAS>(nxt, tid, 0, bs, sca);                                              \
      acc[0][1] = mma3(fa, fb, 0, 1, acc[0][1]);                                                   \
      cv_a.template store_one<BIAS>(nxt, tid, 1, bs, sca);                                              \
      acc[1][0] = mma3(fa, fb, 1, 0, acc[1][0]);                                                   \
      cv_b.template store_one<false>(nxt + OPERAND, tid, 0, unused, scb);                               \
      acc[1][1] = mma3(fa, fb, 1, 1, acc[1][1]);                                                   \
      cv_b.template store_one<false>(nxt + OPERAND, tid, 1, unused, scb);                               \
      DGDM_STAGE_SCHEDULE                                                                          \
    } else {                                                                                       \
      if (live_m > 0 && live_n > 0) acc[0][0] = mma3(fa, fb, 0, 0, acc[0][0]);                     \
      if (live_m > 0 && live_n > 1) acc[0][1] = mma3(fa, fb, 0, 1, acc[0][1]);                     \
      if (live_m > 1 && live_n > 0) acc[1][0] = mma3(fa, fb, 1, 0, acc[1][0]);                     \
      if (live_m > 1 && live_n > 1) acc[1][1] = mma3(fa, fb, 1, 1, acc[1][1]);                     \
      cv_a.template store_one<BIAS>(nxt, tid, 0, bs, sca); cv_a.template store_one<BIAS>(nxt, tid, 1, bs, sca);  \
      cv_b.template store_one<false>(nxt + OPERAND, tid, 0, unused, scb);                               \
      cv_b.template store_one<false>(nxt + OPERAND, tid, 1, unused, scb);                               \
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
__global__ __launch_bounds__(256, 2) void k_gemmh_rows(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                       int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                       const float* __restrict__ bias, const float* __restrict__ B2, int64_t ldb2,
                                                       int bsplit, const unsigned* __restrict__ amax_a, const unsigned* __restrict__ amax_b,
                                                       const unsigned* __restrict__ amax_b2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int wr = wave >> 1, wc = wave & 1;
  const int live_m = min(2, max(0, (M - (m0 + wr * 64) + 31) / 32)), live_n = min(2, max(0, (N - (n0 + wc * 64) + 31) / 32));
  // one power-of-two scale per operand (wave-uniform; a two-matrix B operand shares the scale of the larger maximum)
  unsigned ub = amax_group(amax_b);
  if (B_SPLIT) { const unsigned u2 = amax_group(amax_b2); ub = ub > u2 ? ub : u2; }
  const float sca = scale_of(amax_group(amax_a)), scb = scale_of(ub);
  const float ia = 1.0f / sca, ib = 1.0f / scb;      // exact: powers of two
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float nobs[4] = {0.f, 0.f, 0.f, 0.f};
  mainloop_h<true, B_KCONTIG, false, B_SPLIT>(A, lda, m0, M, B, ldb, n0, N, 0, K, K, smem, acc, live_m, live_n, nobs, sca, scb, B2, ldb2, bsplit);

  // Epilogue: every output is finished IN PLACE first, the stores follow.  A store whose data register is recycled for the next
  // value makes hipcc wait vmcnt(0) in front of every store: 64 round trips = ~7 us per launch (tools/ubench/gemm_img_stamps.hip).
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
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = (acc[mt][nt][r] * ia) * ib + bv + (ACCUM ? old[r] : 0.f);
    }
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
// ---- dW partial, 32 m rows per stage, transposed LDS reads.
// Both operands of dW = dY^T X have the REDUCTION index (the node m) as their slow index, so an MFMA fragment (8 consecutive m of
// one column per lane) is a transpose of what a coalesced load delivers.  The bf16x3 kernel transposes on the way INTO LDS
// (gemm3.hip's ColStage: 16-m stages, ds_write_b32 per column pair, 8 vector instructions per MFMA); here the stage is stored as it arrives --
// row-major [m][column] halves, one ds_write_b64 per float4 -- and gfx950's ds_read_b64_tr_b16 delivers the column-major
// fragments (per 16 lanes a block of 4 m x 16 columns; lane 4q + p supplies the address of row q, columns 4p..4p+3, lane i
// receives column i of the 4 rows).  Twice the reduction depth per barrier, half the LDS store instructions, ~4.5 vector
// instructions per MFMA.  LDS rows are 320 B (256 B of halves + 64 B pad): the 4 rows x 64 B a 32-lane half reads sit on
// disjoint bank quarters.
namespace tn32 {
constexpr int KS2 = 32;                  // m rows per stage
constexpr int RS = 320;                  // bytes per LDS row
constexpr int PLANE2 = KS2 * RS;         // 10240
constexpr int OPER2 = 2 * PLANE2;        // hi plane, lo plane
constexpr int STAGE2 = 2 * OPER2;        // dY image, X image: 40960
constexpr int LDS2 = 2 * STAGE2;         // 81920: two workgroups per CU
#ifndef DGDM_TN_FLUSH
#define DGDM_TN_FLUSH 16
#endif
constexpr int TN_FLUSH = DGDM_TN_FLUSH;  // stages (of 2 k16-steps x 3 MFMAs per tile) between two flushes of the accumulators; a power of two

typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

// one operand's 32 x 128 stage in registers: thread (cg = tid & 31, mq = tid >> 5) holds rows 4 mq + e (e = 0..3), columns 4 cg .. +3
typedef float f32x4r __attribute__((ext_vector_type(4)));
struct Regs {
  f32x4r v[4];
  unsigned ok;        // bit e: row e exists and belongs to this chunk (and the columns exist)
};

struct Loader {
  const float* p[4];
  int64_t step;
  int m, mend, mlim;
  bool colok;
  __device__ __forceinline__ void init(const float* __restrict__ P, int64_t ld, int c0, int cn, int mbeg, int mend_, int mlim_, int tid) {
    const int col = c0 + 4 * (tid & 31);
    colok = col < cn;                                  // cn % 4 == 0: a float4 is inside or outside as a whole
    m = mbeg + 4 * (tid >> 5);
    mend = mend_; mlim = mlim_;
    step = (int64_t)KS2 * ld;
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = P + (int64_t)min(m + e, mlim - 1) * ld + (colok ? col : 0);
  }
  // The loads are inline asm: hipcc sinks a plain load to just in front of its first use -- here the conversion at the END of the
  // stage, with the whole memory latency exposed.  Completion is by hand (k_gemmh_tn32: one `s_waitcnt vmcnt(0)` tied to the eight
  // registers before the conversion); the register set is written by no one else in between (checked in the ISA: no spill, no
  // copy of these registers between the loads and the wait -- keep it so when this kernel's register budget changes).
  __device__ __forceinline__ void next(Regs& R) {
    unsigned ok = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) ok |= (colok && m + e < mend) ? (1u << e) : 0u;
#ifdef DGDM_STAGE_CANARY      // see csrc/gemm_img.hip: the registers hold NaN until the load lands
    {
      const float nan_ = __builtin_nanf("");
      R.v[0] = R.v[1] = R.v[2] = R.v[3] = f32x4r{nan_, nan_, nan_, nan_};
    }
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                 "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off"
                 : "+v"(R.v[0]), "+v"(R.v[1]), "+v"(R.v[2]), "+v"(R.v[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
#else
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                 "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off"
                 : "=&v"(R.v[0]), "=&v"(R.v[1]), "=&v"(R.v[2]), "=&v"(R.v[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] += (m + e + KS2 < mlim) ? step : 0;         // stop at the end of memory; those rows are masked anyway
    R.ok = ok;
    m += KS2;
  }
};

// split the stage and store it row-major: rows 4 mq + e, 8 bytes (4 halves) per plane at column 4 cg
template <bool BIAS>
__device__ __forceinline__ void store_stage(const Regs& R, char* __restrict__ img, int tid, float sc, float (&bs)[4]) {
  char* dst = img + (4 * (tid >> 5)) * RS + 8 * (tid & 31);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool g = (R.ok >> e) & 1u;
    const float x0 = g ? R.v[e][0] : 0.f, x1 = g ? R.v[e][1] : 0.f, x2 = g ? R.v[e][2] : 0.f, x3 = g ? R.v[e][3] : 0.f;
    if (BIAS) { bs[0] += x0; bs[1] += x1; bs[2] += x2; bs[3] += x3; }
    uint2 h, l;
    split_pair(x0 * sc, x1 * sc, &h.x, &l.x);
    split_pair(x2 * sc, x3 * sc, &h.y, &l.y);
    *reinterpret_cast<uint2*>(dst + e * RS) = h;
    *reinterpret_cast<uint2*>(dst + e * RS + PLANE2) = l;
  }
}

// MFMA fragment (8 consecutive m of one column per lane) of the 32-column tile at byte column offset `cb`, MFMA step j, from a plane
__device__ __forceinline__ f16x8 tr_frag(const char* __restrict__ plane_lane, int j, int cb) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane_lane + (16 * j) * RS + cb));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane_lane + (16 * j + 4) * RS + cb));
  typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));
  const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(f16x8, r);
}
}  // namespace tn32

// one workgroup's tile: (bx, by) of the [N x K] output, row chunk mc.  Shared by the one-problem launch and by the many-problem one.
template <bool BIAS>
__device__ __forceinline__ void tn32_tile(const float* __restrict__ dY, int64_t ldy, const float* __restrict__ X, int64_t ldx, int M, int N,
                                          int K, int chunk, int with_bias, float* __restrict__ partial,
                                          const unsigned* __restrict__ amax_dy, const unsigned* __restrict__ amax_x, int bx, int by, int mc,
                                          char* __restrict__ smem) {
  using namespace tn32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = bx * BM, kk0 = by * BN;
  const int mbeg = mc * chunk, mend = min(M, mbeg + chunk);
  const int wr = wave >> 1, wc = wave & 1;
  const int nst = (mend - mbeg + KS2 - 1) / KS2;

  Loader la, lb;
  la.init(dY, ldy, n0, N, mbeg, mend, M, tid);
  lb.init(X, ldx, kk0, K, mbeg, mend, M, tid);
  Regs ra, rb;
  la.next(ra); lb.next(rb);
  const float sca = scale_of(amax_group(amax_dy)), scb = scale_of(amax_group(amax_x));
  const float ia = 1.0f / sca, ib = 1.0f / scb;
#define DGDM_TN32_RETIRE()                                                                                       \
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra.v[0]), "+v"(ra.v[1]), "+v"(ra.v[2]), "+v"(ra.v[3]), "+v"(rb.v[0]), "+v"(rb.v[1]),       \
               "+v"(rb.v[2]), "+v"(rb.v[3]) :: "memory");

  // Two-level accumulation (round 6).  The matrix pipe adds a block of products into its fp32 accumulator with about TWICE the error of
  // one correctly rounded add (tools/ubench/mfma_rounding.hip: rms 1.76e-6 against 0.84e-6 over 2 500 accumulating MFMAs, in units of
  // sqrt(sum p^2)), and a dW chunk is a long chain: rows / 16 steps x 3 MFMAs (1 875 accumulations per tile at the headline batch).
  // Every TN_FLUSH stages (16 stages = 96 accumulations per tile) the accumulators are added into a second set by the vector unit
  // (round to nearest) and start again from zero: the same measurement gives 0.36e-6 against 1.64e-6 for a chain of 1 875 (0.30e-6
  // with 48, 0.26e-6 with 24: the interval is not critical).  64 v_add + 64 v_mov per 384 MFMAs of this wave: same-box A/B of the
  // step with a flush every 4 stages + 0.06 ms of 13.0, every 16 stages within the run-to-run spread (profiles/r06_tn_flush_ab.txt).
  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[a][b][r] = 0.f; tot[a][b][r] = 0.f; }
  float bs[4] = {0.f, 0.f, 0.f, 0.f}, nobs[4];

  // this lane's block address inside a 32-column tile: lane = 16 g + 4 q + p -> row 8 (g >> 1) + q, byte column 32 (g & 1) + 8 p
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int lane_off = (8 * (g >> 1) + q) * RS + 32 * (g & 1) + 8 * pp;

  DGDM_TN32_RETIRE()
  store_stage<BIAS>(ra, smem, tid, sca, bs);
  store_stage<false>(rb, smem + OPER2, tid, scb, nobs);

  // stage s: the loads of stage s + 1 go out first; MFMAs from LDS[s & 1]; then the loaded stage is split into LDS[(s + 1) & 1].
  // The conversion runs behind this wave's MFMAs, under those of the CU's other workgroup (two per CU, never in lockstep).
  for (int s = 0; s < nst; ++s) {
    __syncthreads();
    const char* cur = smem + (s & 1) * STAGE2 + lane_off;
    char* nxt = smem + ((s & 1) ^ 1) * STAGE2;
    la.next(ra);
    lb.next(rb);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f16x8 fa[2][2], fb[2][2];
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          fa[p][t] = tr_frag(cur + p * PLANE2, j, 2 * (wr * 64 + t * 32));
          fb[p][t] = tr_frag(cur + OPER2 + p * PLANE2, j, 2 * (wc * 64 + t * 32));
        }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          acc[mt][nt] = mfma_hf(fa[1][mt], fb[0][nt], acc[mt][nt]);   // smaller terms first
          acc[mt][nt] = mfma_hf(fa[0][mt], fb[1][nt], acc[mt][nt]);
          acc[mt][nt] = mfma_hf(fa[0][mt], fb[0][nt], acc[mt][nt]);
        }
    }
    if ((s & (TN_FLUSH - 1)) == TN_FLUSH - 1) {       // wave-uniform
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) { tot[mt][nt][r] += acc[mt][nt][r]; acc[mt][nt][r] = 0.f; }
    }
    DGDM_TN32_RETIRE()
    store_stage<BIAS>(ra, nxt, tid, sca, bs);
    store_stage<false>(rb, nxt + OPER2, tid, scb, nobs);
  }
#undef DGDM_TN32_RETIRE

  const int j = lane & 31, hi = lane >> 5;
  const int64_t width = (int64_t)N * K + (with_bias ? N : 0);
  float* P = partial + (int64_t)mc * width;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = ((acc[mt][nt][r] + tot[mt][nt][r]) * ia) * ib;     // in place first: see k_gemmh_rows
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = kk0 + wc * 64 + nt * 32 + j;
      if (col >= K) continue;
      const int rbase = n0 + wr * 64 + mt * 32 + 4 * hi;
      float* p = P + (int64_t)rbase * K + col;
      if (n0 + wr * 64 + mt * 32 + 32 <= N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) p[(int64_t)((r & 3) + 8 * (r >> 2)) * K] = acc[mt][nt][r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (rbase + (r & 3) + 8 * (r >> 2) < N) p[(int64_t)((r & 3) + 8 * (r >> 2)) * K] = acc[mt][nt][r];
      }
    }
  if (BIAS && with_bias && by == 0) {   // threads tid, tid + 32, ... (the 8 row groups) hold sums of the same 4 columns: add them through LDS
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    *reinterpret_cast<float4*>(&red[(tid >> 5) * 128 + 4 * (tid & 31)]) = make_float4(bs[0], bs[1], bs[2], bs[3]);
    __syncthreads();
    if (tid < 128) {
      float t = 0.f;
#pragma unroll
      for (int gq = 0; gq < 8; ++gq) t += red[gq * 128 + tid];
      if (n0 + tid < N) P[(int64_t)N * K + n0 + tid] = t;
    }
  }
}

template <bool BIAS>
__global__ __launch_bounds__(256, 2) void k_gemmh_tn32(const float* __restrict__ dY, int64_t ldy, const float* __restrict__ X, int64_t ldx,
                                                       int M, int N, int K, int chunk, int with_bias, float* __restrict__ partial,
                                                       const unsigned* __restrict__ amax_dy, const unsigned* __restrict__ amax_x) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  tn32_tile<BIAS>(dY, ldy, X, ldx, M, N, K, chunk, with_bias, partial, amax_dy, amax_x, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// MANY dW problems in one launch (the weight gradients of the pooled U-Net levels and of the 128-wide layers: 12-25 us each as
// launches of their own, a few dozen workgroups of four stages -- start-up, not work).  Workgroup b belongs to the last problem
// whose block0 <= b; inside it the tile order of the one-problem grid.  The descriptors travel in the kernel arguments.
struct TnProb {
  const float* dY; const float* X; float* partial; const unsigned* amax_dy; const unsigned* amax_x;
  int64_t ldy, ldx;
  int M, N, K, chunk, with_bias, gx, gy, block0, nchunks, cpad;
};
struct TnMany { TnProb p[DGDM_TN_PARTIAL_MAX]; int n; };

__global__ __launch_bounds__(256, 2) void k_gemmh_tn32_many(const TnMany b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int i = 0;
  for (int j = 1; j < b.n; ++j) i = ((int)blockIdx.x >= b.p[j].block0) ? j : i;        // wave-uniform: scalar loads of the arguments
  const TnProb& q = b.p[i];
  // XCD-aware order.  Workgroups go to the 8 XCDs round-robin and every XCD has an L2 of its own; the tiles of one ROW CHUNK read
  // the same rows of dY and X.  With the chunk as the fastest index (chunk counts are powers of two, block0 a multiple of 8) a
  // chunk's tiles all land on the XCD(s) chunk % 8, whose L2 then serves the re-reads: tile-fastest order spread them over all 8
  // and every L2 fetched the rows again (2.7x the operands' bytes over the fabric: profiles/r03_pmc_traffic.json).
  const int local = (int)blockIdx.x - q.block0;
  const int mc = local % q.cpad, t = local / q.cpad;
  if (mc >= q.nchunks || t >= q.gx * q.gy) return;
  tn32_tile<true>(q.dY, q.ldy, q.X, q.ldx, q.M, q.N, q.K, q.chunk, q.with_bias, q.partial, q.amax_dy, q.amax_x, t % q.gx, t / q.gx, mc, smem);
}

// dW[n][k] / db[n] = sum over slots of in[slot][n*K + k] / in[slot][N*K + n]; fixed order: thread (column, part)
// adds slots part, part + PARTS, ... in order, the PARTS partial sums are added in order.  256 / PARTS columns per
// block: 4 parts for a few chunks, 16 parts when a small output was cut into hundreds of row chunks (one launch
// instead of a staged column sum plus a final pass).
template <int PARTS>
__global__ __launch_bounds__(256) void k_gemmh_tn_final(const float* __restrict__ in, int slots, int64_t width, int N, int K,
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

constexpr int LDS_BYTES = 2 * STAGE;   // 49152

int tnh_chunk_rows(int M, int N, int K, int want_total = DGDM_TN_WANT) {
  // 64-bit throughout (see gemm3.hip, tn3_chunk_rows)
  const int64_t tiles = (((int64_t)N + BM - 1) / BM) * (((int64_t)K + BN - 1) / BN);
  int64_t want = (want_total + tiles / 2) / tiles;   // workgroups per problem: see common.hpp
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  int64_t chunk = ((int64_t)M + want - 1) / want;
  chunk = (chunk + 2 * KS - 1) / (2 * KS) * (2 * KS);
  if (chunk < 8 * KS) chunk = 8 * KS;
  return (int)(chunk > 0x7fffff00 ? 0x7fffff00 : chunk);
}

// dynamic LDS opt-in once per kernel (per process; one device per process)
template <typename Kern>
int allow_big_lds(Kern kern, int bytes = LDS_BYTES) {
  static int status = 1;   // 1 = not asked yet
  if (status == 1)
    status = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess
                 ? DGDM_OK : DGDM_ERR_LAUNCH;
  return status;
}

}  // namespace

static bool bad_amax(const void* a, const void* b) { return !a || !b; }

template <typename Kern>
static int launch_rows(Kern kern, dim3 grid, hipStream_t s, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                       int M, int N, int K, const float* bias, const unsigned* amax_a, const unsigned* amax_b, const float* B2 = nullptr,
                       int64_t ldb2 = 0, int bsplit = 0, const unsigned* amax_b2 = nullptr) {
  if (allow_big_lds(kern) != DGDM_OK) return DGDM_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, s, A, lda, B, ldb, C, ldc, M, N, K, bias, B2, ldb2, bsplit, amax_a, amax_b, amax_b2);
  return dgdm_launch_status();
}

extern "C" int dgdm_gemm_nt_f16x2(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                                  int32_t M, int32_t N, int32_t K, int32_t accumulate, const uint32_t* amax_a, const uint32_t* amax_w,
                                  void* stream) {
  if (M < 0 || N < 0 || K < 0) return DGDM_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return DGDM_OK;
  if (!A || !W || !C || bad_amax(amax_a, amax_w)) return DGDM_ERR_INVALID_ARG;
  if (K == 0) return DGDM_ERR_UNSUPPORTED;
  if ((K & 3) || (lda & 3) || (ldw & 3) || !dgdm_aligned16(A) || !dgdm_aligned16(W)) return DGDM_ERR_UNSUPPORTED;
  if (lda < K || ldw < K || ldc < N) return DGDM_ERR_INVALID_ARG;
  const dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return accumulate ? launch_rows(k_gemmh_rows<true, true>, grid, s, A, lda, W, ldw, C, ldc, M, N, K, bias, amax_a, amax_w)
                    : launch_rows(k_gemmh_rows<true, false>, grid, s, A, lda, W, ldw, C, ldc, M, N, K, bias, amax_a, amax_w);
}

// C = A . [W0 | W1]^T (+ bias): the weight's K columns come from two matrices, [0, K0) from W0 and [K0, K) from W1
extern "C" int dgdm_gemm_nt_split_f16x2(const float* A, int64_t lda, const float* W0, int64_t ldw0, int32_t K0, const float* W1,
                                        int64_t ldw1, const float* bias, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                                        int32_t accumulate, const uint32_t* amax_a, const uint32_t* amax_w0, const uint32_t* amax_w1,
                                        void* stream) {
  if (M < 0 || N < 0 || K < 0 || K0 <= 0 || K0 >= K) return DGDM_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return DGDM_OK;
  if (!A || !W0 || !W1 || !C || bad_amax(amax_a, amax_w0) || !amax_w1) return DGDM_ERR_INVALID_ARG;
  if ((K & 3) || (K0 & 3) || (lda & 3) || (ldw0 & 3) || (ldw1 & 3) || !dgdm_aligned16(A) || !dgdm_aligned16(W0) || !dgdm_aligned16(W1))
    return DGDM_ERR_UNSUPPORTED;
  if (lda < K || ldw0 < K0 || ldw1 < K - K0 || ldc < N) return DGDM_ERR_INVALID_ARG;
  const dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return accumulate ? launch_rows(k_gemmh_rows<true, true, true>, grid, s, A, lda, W0, ldw0, C, ldc, M, N, K, bias, amax_a, amax_w0, W1, ldw1, K0, amax_w1)
                    : launch_rows(k_gemmh_rows<true, false, true>, grid, s, A, lda, W0, ldw0, C, ldc, M, N, K, bias, amax_a, amax_w0, W1, ldw1, K0, amax_w1);
}

extern "C" int dgdm_gemm_nn_f16x2(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int32_t M, int32_t N,
                                  int32_t Kout, int32_t accumulate, const uint32_t* amax_a, const uint32_t* amax_w, void* stream) {
  if (M < 0 || N < 0 || Kout < 0) return DGDM_ERR_INVALID_ARG;
  if (M == 0 || Kout == 0) return DGDM_OK;
  if (!A || !W || !C || bad_amax(amax_a, amax_w)) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_ERR_UNSUPPORTED;
  if ((N & 3) || (Kout & 3) || (lda & 3) || (ldw & 3) || lda < N || ldw < Kout || ldc < Kout || !dgdm_aligned16(A) || !dgdm_aligned16(W))
    return DGDM_ERR_UNSUPPORTED;
  const dim3 grid((M + BM - 1) / BM, (Kout + BN - 1) / BN);
  const float* nobias = nullptr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return accumulate ? launch_rows(k_gemmh_rows<false, true>, grid, s, A, lda, W, ldw, C, ldc, M, Kout, N, nobias, amax_a, amax_w)
                    : launch_rows(k_gemmh_rows<false, false>, grid, s, A, lda, W, ldw, C, ldc, M, Kout, N, nobias, amax_a, amax_w);
}

extern "C" size_t dgdm_gemm_tn_f16x2_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t with_bias) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int chunk = tnh_chunk_rows(M, N, K);
  const int nchunks = (int)(((int64_t)M + chunk - 1) / chunk);
  const size_t width = (size_t)N * K + (with_bias ? N : 0);
  return (size_t)nchunks * width * sizeof(float);
}

static int tn_impl(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW, int64_t lddw, int32_t K0, float* dW1, int64_t ld1,
                   float* db, int32_t M, int32_t N, int32_t K, void* workspace, size_t workspace_bytes, const uint32_t* amax_dy,
                   const uint32_t* amax_x, void* stream_, bool partial_only = false) {
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
  if (!dY || !X || !workspace || bad_amax(amax_dy, amax_x)) return DGDM_ERR_INVALID_ARG;
  if ((ldy & 3) || (ldx & 3) || (N & 3) || (K & 3) || ldy < N || ldx < K || !dgdm_aligned16(dY) || !dgdm_aligned16(X))
    return DGDM_ERR_UNSUPPORTED;
  const int chunk = tnh_chunk_rows(M, N, K);
  const int nchunks = (int)(((int64_t)M + chunk - 1) / chunk);
  const int64_t width = (int64_t)N * K + (db ? N : 0);
  if (width > 0x7fffffffLL) return DGDM_ERR_UNSUPPORTED;
  if (workspace_bytes < (size_t)nchunks * width * sizeof(float)) return DGDM_ERR_WORKSPACE;
  float* partial = static_cast<float*>(workspace);
  const dim3 grid((N + BM - 1) / BM, (K + BN - 1) / BN, nchunks);
  if (db) {
    if (allow_big_lds(k_gemmh_tn32<true>, tn32::LDS2) != DGDM_OK) return DGDM_ERR_LAUNCH;
    hipLaunchKernelGGL(k_gemmh_tn32<true>, grid, dim3(256), tn32::LDS2, s, dY, ldy, X, ldx, M, N, K, chunk, 1, partial, amax_dy, amax_x);
  } else {
    if (allow_big_lds(k_gemmh_tn32<false>, tn32::LDS2) != DGDM_OK) return DGDM_ERR_LAUNCH;
    hipLaunchKernelGGL(k_gemmh_tn32<false>, grid, dim3(256), tn32::LDS2, s, dY, ldy, X, ldx, M, N, K, chunk, 0, partial, amax_dy, amax_x);
  }
  if (partial_only) return dgdm_launch_status();   // the caller reduces the chunk partials later (dgdm_gemm_tn_reduce_many)
  if (nchunks > 32)
    hipLaunchKernelGGL(k_gemmh_tn_final<16>, dim3((unsigned)((width + 15) / 16)), dim3(256), 0, s, partial, nchunks, width, N, K, dW, lddw, db, K0, dW1, ld1);
  else
    hipLaunchKernelGGL(k_gemmh_tn_final<4>, dim3((unsigned)((width + 63) / 64)), dim3(256), 0, s, partial, nchunks, width, N, K, dW, lddw, db, K0, dW1, ld1);
  return dgdm_launch_status();
}

extern "C" int dgdm_gemm_tn_f16x2(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW, int64_t lddw, float* db, int32_t M,
                                  int32_t N, int32_t K, void* workspace, size_t workspace_bytes, const uint32_t* amax_dy,
                                  const uint32_t* amax_x, void* stream) {
  return tn_impl(dY, ldy, X, ldx, dW, lddw, K, nullptr, 0, db, M, N, K, workspace, workspace_bytes, amax_dy, amax_x, stream);
}

extern "C" int dgdm_gemm_tn_partial_f16x2(const float* dY, int64_t ldy, const float* X, int64_t ldx, int32_t with_bias, int32_t M, int32_t N,
                                          int32_t K, void* workspace, size_t workspace_bytes, const uint32_t* amax_dy, const uint32_t* amax_x,
                                          void* stream) {
  float dummy;   // tn_impl only checks the destination pointers for presence when it skips the final reduction
  return tn_impl(dY, ldy, X, ldx, &dummy, K, K, nullptr, 0, with_bias ? &dummy : nullptr, M, N, K, workspace, workspace_bytes, amax_dy, amax_x,
                 stream, true);
}

// rows per chunk of one problem inside dgdm_gemm_tn_partial_many_f16x2: the chunk COUNT aimed for is a power of two (the XCD-aware
// workgroup order of k_gemmh_tn32_many wants chunk % 8 to name an XCD)
static int tnh_chunk_rows_grouped(int M, int N, int K) {
  const int64_t tiles = (((int64_t)N + BM - 1) / BM) * (((int64_t)K + BN - 1) / BN);     // 64-bit: see tnh_chunk_rows
  int64_t want = (DGDM_TN_WANT_GROUPED + tiles / 2) / tiles;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  int64_t p2 = 1;
  while (2 * p2 <= want) p2 *= 2;
  if (2 * p2 - want < want - p2) p2 *= 2;        // nearest power of two
  int64_t chunk = ((int64_t)M + p2 - 1) / p2;
  chunk = (chunk + 2 * KS - 1) / (2 * KS) * (2 * KS);
  if (chunk < 8 * KS) chunk = 8 * KS;
  return (int)(chunk > 0x7fffff00 ? 0x7fffff00 : chunk);
}

// row chunks (= partial slots) of one problem inside dgdm_gemm_tn_partial_many_f16x2
extern "C" int32_t dgdm_gemm_tn_chunks_grouped(int32_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int chunk = tnh_chunk_rows_grouped(M, N, K);
  return (int32_t)(((int64_t)M + chunk - 1) / chunk);
}

extern "C" int dgdm_gemm_tn_partial_many_f16x2(const DgdmTnPartial* descs, int32_t count, void* stream) {
  if (count < 0 || count > DGDM_TN_PARTIAL_MAX || (count && !descs)) return DGDM_ERR_INVALID_ARG;
  if (count == 0) return DGDM_OK;
  TnMany b;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const DgdmTnPartial& d = descs[i];
    if (d.M <= 0 || d.N <= 0 || d.K <= 0 || !d.dY || !d.X || !d.workspace || bad_amax(d.amax_dy, d.amax_x)) return DGDM_ERR_INVALID_ARG;
    if ((d.ldy & 3) || (d.ldx & 3) || (d.N & 3) || (d.K & 3) || d.ldy < d.N || d.ldx < d.K || !dgdm_aligned16(d.dY) || !dgdm_aligned16(d.X))
      return DGDM_ERR_UNSUPPORTED;
    const int chunk = tnh_chunk_rows_grouped(d.M, d.N, d.K);     // the problems fill the chip TOGETHER
    const int nchunks = (d.M + chunk - 1) / chunk;
    int cpad = 1;
    while (cpad < nchunks) cpad *= 2;
    const int64_t width = (int64_t)d.N * d.K + (d.with_bias ? d.N : 0);
    if (width > 0x7fffffffLL) return DGDM_ERR_UNSUPPORTED;
    if (d.workspace_bytes < (size_t)nchunks * width * sizeof(float)) return DGDM_ERR_WORKSPACE;
    TnProb& q = b.p[i];
    q.dY = d.dY; q.X = d.X; q.partial = static_cast<float*>(d.workspace); q.amax_dy = d.amax_dy; q.amax_x = d.amax_x;
    q.ldy = d.ldy; q.ldx = d.ldx; q.M = d.M; q.N = d.N; q.K = d.K; q.chunk = chunk; q.with_bias = d.with_bias ? 1 : 0;
    q.gx = (d.N + BM - 1) / BM; q.gy = (d.K + BN - 1) / BN; q.block0 = blocks; q.nchunks = nchunks; q.cpad = cpad;
    const int64_t nb = ((int64_t)q.gx * q.gy * cpad + 7) / 8 * 8;       // block0 of the next problem stays a multiple of 8
    if (nb + blocks > 0x3fffffff) return DGDM_ERR_UNSUPPORTED;
    blocks += (int)nb;
  }
  b.n = count;
  if (allow_big_lds(k_gemmh_tn32_many, tn32::LDS2) != DGDM_OK) return DGDM_ERR_LAUNCH;
  hipLaunchKernelGGL(k_gemmh_tn32_many, dim3((unsigned)blocks), dim3(256), tn32::LDS2, static_cast<hipStream_t>(stream), b);
  return dgdm_launch_status();
}

extern "C" int dgdm_gemm_tn_split_f16x2(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW0, int64_t ld0, int32_t K0,
                                        float* dW1, int64_t ld1, float* db, int32_t M, int32_t N, int32_t K, void* workspace,
                                        size_t workspace_bytes, const uint32_t* amax_dy, const uint32_t* amax_x, void* stream) {
  return tn_impl(dY, ldy, X, ldx, dW0, ld0, K0, dW1, ld1, db, M, N, K, workspace, workspace_bytes, amax_dy, amax_x, stream);
}

// ------------------------------------------------------------------------------------------------ absolute maxima
namespace {
// block-wide maximum of the float bits of |x|, then ONE atomic per block into way (block % WAYS) of the group
__device__ __forceinline__ void amax_commit(unsigned m, unsigned* __restrict__ group, int way) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  __shared__ unsigned sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
    if (t) atomicMax(group + (way % DGDM_AMAX_WAYS) * DGDM_AMAX_STRIDE, t);
  }
}

__global__ __launch_bounds__(256) void k_amax_bits(const float* __restrict__ x, int64_t ld, int cols4, int64_t rows, unsigned* __restrict__ group) {
  const int64_t n4 = rows * cols4, stride = (int64_t)gridDim.x * blockDim.x;
  unsigned m = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = *reinterpret_cast<const float4*>(x + (i / cols4) * ld + 4 * (i % cols4));
    const unsigned a = __float_as_uint(v.x) & 0x7fffffffu, b = __float_as_uint(v.y) & 0x7fffffffu;
    const unsigned c = __float_as_uint(v.z) & 0x7fffffffu, d = __float_as_uint(v.w) & 0x7fffffffu;
    m = max(m, max(max(a, b), max(c, d)));
  }
  amax_commit(m, group, blockIdx.x);
}

struct AmaxDesc { const float* p; long long n; long long group; };   // group: index of the tensor's slot group
// one launch for MANY tensors (all weights of a model): block b handles record b (a chunk of at most 64K floats of one tensor)
__global__ __launch_bounds__(256) void k_amax_table(const AmaxDesc* __restrict__ table, unsigned* __restrict__ groups) {
  const AmaxDesc d = table[blockIdx.x];
  unsigned m = 0;
  const long long n4 = d.n >> 2;
  for (long long i = threadIdx.x; i < n4; i += 256) {
    const float4 v = reinterpret_cast<const float4*>(d.p)[i];
    m = max(m, max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu),
                   max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
  }
  for (long long i = (n4 << 2) + threadIdx.x; i < d.n; i += 256) m = max(m, __float_as_uint(d.p[i]) & 0x7fffffffu);
  amax_commit(m, groups + d.group * (DGDM_AMAX_WAYS * DGDM_AMAX_STRIDE), blockIdx.x);
}
}  // namespace

extern "C" int dgdm_amax_bits(const float* x, int64_t ld, int64_t rows, int32_t cols, uint32_t* group, void* stream) {
  if (rows < 0 || cols < 0 || !group) return DGDM_ERR_INVALID_ARG;
  if (rows == 0 || cols == 0) return DGDM_OK;
  if (!x) return DGDM_ERR_INVALID_ARG;
  if ((cols & 3) || (ld & 3) || ld < cols || !dgdm_aligned16(x)) return DGDM_ERR_UNSUPPORTED;
  const int64_t n4 = rows * (cols >> 2);
  int64_t blocks = (n4 + 2047) / 2048;
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_amax_bits, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, ld, cols >> 2, rows, group);
  return dgdm_launch_status();
}

extern "C" int dgdm_amax_table(const void* table, int32_t count, uint32_t* groups, void* stream) {
  if (count < 0 || (count > 0 && (!table || !groups))) return DGDM_ERR_INVALID_ARG;
  if (count == 0) return DGDM_OK;
  hipLaunchKernelGGL(k_amax_table, dim3((unsigned)count), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const AmaxDesc*>(table), groups);
  return dgdm_launch_status();
}
