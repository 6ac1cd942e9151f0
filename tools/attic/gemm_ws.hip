// K3-ws: weight-STATIONARY row contraction for the narrow layers of the path (the graph U-Net and the last encoder layers:
// C = 128, K = 128 or 160 -- core/graph_layers.py:45-49,141-150 at hidden_channels = 128), with the fused epilogues of gemm_epi.hpp.
//
// Why a second kernel (profiles/r05a_*): at these widths k_gemm_img<4,1> is neither matrix- nor HBM-bound.  A 128-row workgroup
// re-stages the weight image stage by stage (DMA -> wait -> barrier, three times for K = 160), a launch over 40 000 rows is 313
// such workgroups on 256 CUs (1.2 per CU, 0.55 waves per SIMD over a step), and whatever follows the main loop -- an epilogue with
// an erf, a LayerNorm -- runs with nothing beside it: 16 us for 46 MB, 31 us with an activation fused.  Here
//   * the WHOLE image of the layer ([<= 128 columns] x [K <= 160]: 64 - 80 KiB) is copied into LDS ONCE per workgroup by LDS-DMA,
//     while the first activation rows are already on their way; ONE barrier per launch;
//   * after it the waves are independent: each takes 32-row tiles (tile = wave index + i * waves of the grid), loads ALL K / 32
//     chunks of a tile at once (five 64-byte loads per lane in flight, straight into MFMA fragment registers as in gemm_img.hip)
//     and refills a chunk's registers for its NEXT tile as soon as the chunk is converted -- a whole tile of prefetch distance;
//     the loads are unconditional straight-line code (see k_gemm_ws);
//   * accumulators are transposed (weight fragment as the MFMA's A operand), so every epilogue is the row-wise code of
//     gemm_epi.hpp: float4 accesses, four-element dropout words, in-lane row statistics.
// Arithmetic is gemm_img.hip's to the bit for the products (same fragments, same three MFMAs per term, same scales).
#include "gemm_epi.hpp"

namespace {

constexpr int WS_NT = 4;        // column tiles of 32: up to 128 output columns
constexpr int WS_MAXCH = 5;     // 32-k chunks: K <= 160

typedef float f32x4 __attribute__((ext_vector_type(4)));

// NCH = 32-k chunks of a row (K <= 32 NCH) is a template argument and every load below is UNCONDITIONAL: a load under a
// (wave-uniform) branch makes its destination registers a phi of "loaded" and "old" values, and hipcc is free to resolve such a phi
// with register copies placed BEFORE the hand-placed wait -- copies of registers whose data has not arrived (the staging canary of
// tests/test_hip_gemm_img.py caught exactly that in the first version of this kernel).  Straight-line code has no phis.  A prefetch
// for a tile past the end reads row M - 1 again (rows are clamped) and is never used.
template <int WAVES, int EPI, int NCH>
__global__ __launch_bounds__(64 * WAVES, 8 / WAVES) void k_gemm_ws(const float* __restrict__ A, int64_t lda, int M, int K,
                                                                   const char* __restrict__ img, int T_img, int t_begin, int Ncols,
                                                                   const float* __restrict__ bias, float* __restrict__ C, int64_t ldc,
                                                                   const unsigned* __restrict__ amax_a, const EpiArgs epi) {
  static_assert(NCH >= 1 && NCH <= WS_MAXCH, "K <= 160");
  extern __shared__ __attribute__((aligned(16))) char smem[];    // NCH x WS_NT x BLK: chunk c, tile t at (c * WS_NT + t) * BLK
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles = (Ncols + 31) >> 5;
  const int ntiles = (M + 31) >> 5;
  const int stride = gridDim.x * WAVES;
  int tile = blockIdx.x * WAVES + wave;
  // The two epilogues that hold the most registers (row statistics; pre-activations in flight) cannot also hold NCH chunks of the
  // next tile: for them the LAST chunk of a tile is loaded at the top of that tile (NCH - 1 chunks ahead of its use) instead of a
  // tile ahead, so its 16 registers are free during the epilogue.  These kernels must not spill (tests/test_abi.py).
  constexpr bool LATE = (EPI == EPI_NORM || EPI == EPI_ACTBWD) && NCH == WS_MAXCH;
  constexpr int AHEAD = LATE ? NCH - 1 : NCH;          // chunks in flight across a tile boundary: 0 .. AHEAD - 1

  // the image -> LDS, 1 KiB per wave instruction: chunk c = blocks [c * T_img + t_begin, + tiles), contiguous on both sides
  {
    const char* blocks = img + IMG_HDR + (size_t)t_begin * BLK + lane * 16;
    const int per_chunk = tiles * 4;
    for (int p = wave; p < NCH * per_chunk; p += WAVES) {
      const int c = p / per_chunk, pc = p - c * per_chunk;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(blocks + (size_t)c * T_img * BLK + pc * 1024),
                                       (__attribute__((address_space(3))) void*)(smem + c * WS_NT * BLK + pc * 1024), 16, 0, 0);
    }
  }

  // this lane's share of a tile's A rows: row (lane & 31), floats k = 32 c + 16 (lane >> 5) .. +15 of every chunk c.  The loads are
  // inline asm retired by ONE hand-placed s_waitcnt at the top of a tile (see gemm_img.hip: hipcc sinks a plain load to its first use);
  // the registers of chunk c are refilled for the wave's next tile right after chunk c has been converted.
  const int klane = 16 * (lane >> 5);
  f32x4 r0a, r0b, r0c, r0d, r1a, r1b, r1c, r1d, r2a, r2b, r2c, r2d, r3a, r3b, r3c, r3d, r4a, r4b, r4c, r4d;
#define WS_LOAD(tile_, c_, a_, b_, c4_, d_)                                                                         \
  {                                                                                                                 \
    const float* p__ = A + (int64_t)min(32 * (tile_) + (lane & 31), M - 1) * lda + min(32 * (c_) + klane, K - 16);  \
    DGDM_CANARY_POISON(a_, b_, c4_, d_)                                                                             \
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"                \
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"            \
                 : DGDM_CANARY_OUT(a_), DGDM_CANARY_OUT(b_), DGDM_CANARY_OUT(c4_), DGDM_CANARY_OUT(d_) : "v"(p__) : "memory"); \
  }
  // ONE wait statement ties every register set that is in flight across the tile boundary (a set that is not -- the late chunk,
  // chunk positions past NCH -- must not be named: tying it would keep it alive through the epilogue)
#define WS_T(x_) "+v"(x_##a), "+v"(x_##b), "+v"(x_##c), "+v"(x_##d)
#define WS_WAIT_AHEAD                                                                                               \
  {                                                                                                                 \
    if constexpr (AHEAD == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
    if constexpr (AHEAD == 1) asm volatile("s_waitcnt vmcnt(0)" : WS_T(r0) :: "memory");                            \
    if constexpr (AHEAD == 2) asm volatile("s_waitcnt vmcnt(0)" : WS_T(r0), WS_T(r1) :: "memory");                  \
    if constexpr (AHEAD == 3) asm volatile("s_waitcnt vmcnt(0)" : WS_T(r0), WS_T(r1), WS_T(r2) :: "memory");        \
    if constexpr (AHEAD == 4) asm volatile("s_waitcnt vmcnt(0)" : WS_T(r0), WS_T(r1), WS_T(r2), WS_T(r3) :: "memory");            \
    if constexpr (AHEAD == 5) asm volatile("s_waitcnt vmcnt(0)" : WS_T(r0), WS_T(r1), WS_T(r2), WS_T(r3), WS_T(r4) :: "memory");  \
  }
#define WS_LOAD_AHEAD(tile_)                                                                                        \
  {                                                                                                                 \
    if constexpr (AHEAD > 0) WS_LOAD(tile_, 0, r0a, r0b, r0c, r0d)                                                  \
    if constexpr (AHEAD > 1) WS_LOAD(tile_, 1, r1a, r1b, r1c, r1d)                                                  \
    if constexpr (AHEAD > 2) WS_LOAD(tile_, 2, r2a, r2b, r2c, r2d)                                                  \
    if constexpr (AHEAD > 3) WS_LOAD(tile_, 3, r3a, r3b, r3c, r3d)                                                  \
    if constexpr (AHEAD > 4) WS_LOAD(tile_, 4, r4a, r4b, r4c, r4d)                                                  \
  }

  WS_LOAD_AHEAD(tile)                                  // (a wave without a tile loads row M - 1: harmless, retired below)
  const float sca = scale_of(amax_group(amax_a));
  const float scb = *reinterpret_cast<const float*>(img);
  const float inv = (1.0f / sca) * (1.0f / scb);      // exact: powers of two
  // everything an epilogue needs that does not depend on the tile is fetched NOW, under the image copy: the dropout seeds, and the
  // per-column vectors four columns per lane (EpiVecLanes; lanes past the matrix hold zeros)
  uint32_t seed_v = 0, pseed_v = 0;
  if (EPI != EPI_NONE) { seed_v = epi.seed.value(); pseed_v = EPI == EPI_NORM ? epi.pre_seed.value() : 0u; }
  EpiVecLanes vbias{ld4_if(bias + 4 * lane, bias && 4 * lane < Ncols)}, vgamma{make_float4(0.f, 0.f, 0.f, 0.f)}, vbeta{make_float4(0.f, 0.f, 0.f, 0.f)};
  if (EPI == EPI_NORM) {
    vgamma.v = ld4_if(epi.gamma + 4 * lane, 4 * lane < Ncols);
    vbeta.v = ld4_if(epi.beta + 4 * lane, 4 * lane < Ncols);
  }
  WS_WAIT_AHEAD                                        // the image pieces this wave copied (and the first tile's rows)
  __syncthreads();                                     // ... and everybody else's: the only barrier before the end

  unsigned am = 0;
  const char* lbase = smem + lane * 16;
  while (tile < ntiles) {
    WS_WAIT_AHEAD                                      // this tile's rows (issued a whole tile ago) and the last epilogue's stores
    const int next = tile + stride;
    if constexpr (LATE) WS_LOAD(tile, 4, r4a, r4b, r4c, r4d)
    f32x16 acc[WS_NT];
#pragma unroll
    for (int t = 0; t < WS_NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // A tile is 4 NCH batches (chunk c, half j, column-tile pair p): 4 reads of the weight's fragments (2 column tiles x hi / lo) +
    // 6 MFMAs with the WEIGHT as the A operand (transposed accumulators).  The reads of the next batch are issued BEFORE the MFMAs of
    // this one and pinned there (sched_barrier): left alone, hipcc sinks every read to just in front of its MFMA and a wave that has a
    // SIMD to itself eats the LDS latency 8 NCH times per tile (the first version of this kernel: +3 us per launch against
    // k_gemm_img).  The split of the next 16-k half (and, behind a chunk's second half, the refill of its registers for the next
    // tile) sits among the MFMAs of the current half's first pair; ah / al have one slot per half j.
    f16x8 ah[2], al[2];                                // [half j]
    f16x8 bh[2][2], bl[2][2];                          // [pair parity][tile of the pair]
#define WS_SPLIT_HALF(c_, j_, x_, y_, a_, b_, c4_, d_)     /* floats 8 j .. 8 j + 7 of the chunk = registers (x_, y_) */ \
  {                                                                                                                 \
    if constexpr (LATE && (c_) == NCH - 1 && (j_) == 0)  /* loaded at the top of this tile; younger: the 16 loads of the next tile */ \
      asm volatile("s_waitcnt vmcnt(16)" : "+v"(a_), "+v"(b_), "+v"(c4_), "+v"(d_) :: "memory");                      \
    uint4 h__, l__;                                                                                                 \
    split_pair(x_[0] * sca, x_[1] * sca, &h__.x, &l__.x);                                                           \
    split_pair(x_[2] * sca, x_[3] * sca, &h__.y, &l__.y);                                                           \
    split_pair(y_[0] * sca, y_[1] * sca, &h__.z, &l__.z);                                                           \
    split_pair(y_[2] * sca, y_[3] * sca, &h__.w, &l__.w);                                                           \
    ah[j_] = __builtin_bit_cast(f16x8, h__);                                                                        \
    al[j_] = __builtin_bit_cast(f16x8, l__);                                                                        \
    if constexpr ((j_) == 1 && (c_) < AHEAD) WS_LOAD(next, c_, a_, b_, c4_, d_)   /* both halves taken: refill */     \
  }
#define WS_READ(c_, j_, p_)                                                                                         \
  {                                                                                                                 \
    const char* q__ = lbase + (c_) * WS_NT * BLK + (2 * (j_)) * 1024 + (2 * (p_)) * BLK;                            \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                 \
      bh[p_][u] = *reinterpret_cast<const f16x8*>(q__ + u * BLK);                                                   \
      bl[p_][u] = *reinterpret_cast<const f16x8*>(q__ + u * BLK + 1024);                                            \
    }                                                                                                               \
  }
#define WS_MFMA(j_, p_)                                                                                             \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                   \
    acc[2 * (p_) + u] = mfma_hf(bh[p_][u], al[j_], acc[2 * (p_) + u]);        /* smaller terms first */             \
    acc[2 * (p_) + u] = mfma_hf(bl[p_][u], ah[j_], acc[2 * (p_) + u]);                                              \
    acc[2 * (p_) + u] = mfma_hf(bh[p_][u], ah[j_], acc[2 * (p_) + u]);                                              \
  }
    // half (c, j): [reads of its pair 1] | MFMAs of pair 0 + split of the NEXT half | [reads of the next half's pair 0] | MFMAs of pair 1
#define WS_FENCE __builtin_amdgcn_sched_barrier(0);
    WS_SPLIT_HALF(0, 0, r0a, r0b, r0a, r0b, r0c, r0d)
    WS_READ(0, 0, 0)
    if constexpr (0 < NCH) {
      WS_FENCE WS_READ(0, 0, 1) WS_FENCE WS_MFMA(0, 0)
      WS_SPLIT_HALF(0, 1, r0c, r0d, r0a, r0b, r0c, r0d)
      WS_FENCE WS_READ(0, 1, 0) WS_FENCE WS_MFMA(0, 1)
      WS_FENCE WS_READ(0, 1, 1) WS_FENCE WS_MFMA(1, 0)
      if constexpr (1 < NCH) WS_SPLIT_HALF(1, 0, r1a, r1b, r1a, r1b, r1c, r1d)
      WS_FENCE
      if constexpr (1 < NCH) WS_READ(1, 0, 0)
      WS_FENCE WS_MFMA(1, 1)
    }
    if constexpr (1 < NCH) {
      WS_FENCE WS_READ(1, 0, 1) WS_FENCE WS_MFMA(0, 0)
      WS_SPLIT_HALF(1, 1, r1c, r1d, r1a, r1b, r1c, r1d)
      WS_FENCE WS_READ(1, 1, 0) WS_FENCE WS_MFMA(0, 1)
      WS_FENCE WS_READ(1, 1, 1) WS_FENCE WS_MFMA(1, 0)
      if constexpr (2 < NCH) WS_SPLIT_HALF(2, 0, r2a, r2b, r2a, r2b, r2c, r2d)
      WS_FENCE
      if constexpr (2 < NCH) WS_READ(2, 0, 0)
      WS_FENCE WS_MFMA(1, 1)
    }
    if constexpr (2 < NCH) {
      WS_FENCE WS_READ(2, 0, 1) WS_FENCE WS_MFMA(0, 0)
      WS_SPLIT_HALF(2, 1, r2c, r2d, r2a, r2b, r2c, r2d)
      WS_FENCE WS_READ(2, 1, 0) WS_FENCE WS_MFMA(0, 1)
      WS_FENCE WS_READ(2, 1, 1) WS_FENCE WS_MFMA(1, 0)
      if constexpr (3 < NCH) WS_SPLIT_HALF(3, 0, r3a, r3b, r3a, r3b, r3c, r3d)
      WS_FENCE
      if constexpr (3 < NCH) WS_READ(3, 0, 0)
      WS_FENCE WS_MFMA(1, 1)
    }
    if constexpr (3 < NCH) {
      WS_FENCE WS_READ(3, 0, 1) WS_FENCE WS_MFMA(0, 0)
      WS_SPLIT_HALF(3, 1, r3c, r3d, r3a, r3b, r3c, r3d)
      WS_FENCE WS_READ(3, 1, 0) WS_FENCE WS_MFMA(0, 1)
      WS_FENCE WS_READ(3, 1, 1) WS_FENCE WS_MFMA(1, 0)
      if constexpr (4 < NCH) WS_SPLIT_HALF(4, 0, r4a, r4b, r4a, r4b, r4c, r4d)
      WS_FENCE
      if constexpr (4 < NCH) WS_READ(4, 0, 0)
      WS_FENCE WS_MFMA(1, 1)
    }
    if constexpr (4 < NCH) {
      WS_FENCE WS_READ(4, 0, 1) WS_FENCE WS_MFMA(0, 0)
      WS_SPLIT_HALF(4, 1, r4c, r4d, r4a, r4b, r4c, r4d)
      WS_FENCE WS_READ(4, 1, 0) WS_FENCE WS_MFMA(0, 1)
      WS_FENCE WS_READ(4, 1, 1) WS_FENCE WS_MFMA(1, 0)
      WS_FENCE WS_MFMA(1, 1)
    }
#undef WS_FENCE
#undef WS_MFMA
#undef WS_READ
#undef WS_SPLIT_HALF
    epilogue_tr<WS_NT, EPI>(acc, inv, 32 * tile + (lane & 31), M, 0, Ncols, vbias, vgamma, vbeta, C, ldc, epi, lane >> 5, am, seed_v, pseed_v);
    tile = next;
  }
  WS_WAIT_AHEAD            // the prefetch of the tile past the wave's last one: retired before the registers are given back
#undef WS_LOAD
#undef WS_LOAD_AHEAD
#undef WS_WAIT_AHEAD
#undef WS_T
  if (epi.amax_out) {    // max|C| of the workgroup -> one atomic (the image's LDS is free once every wave has left the loop).  LDS
    // atomics instead of a shuffle butterfly: the butterfly's lane-index registers are common with amax_group's at the top of the
    // kernel, and hipcc kept them alive -- spilled -- across the whole tile loop.
    __syncthreads();
    unsigned* red = reinterpret_cast<unsigned*>(smem);
    // (the lane id taken afresh: threadIdx.x itself would be one more register kept -- spilled -- across the loop)
    const bool first = wave == 0 && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0;
    if (first) red[0] = 0u;
    __syncthreads();
    if (am) atomicMax(red, am);
    __syncthreads();
    if (first && red[0]) atomicMax(epi.amax_out + (blockIdx.x % DGDM_AMAX_WAYS) * DGDM_AMAX_STRIDE, red[0]);
  }
}

template <int WAVES, int EPI, int NCH>
int launch_ws(hipStream_t s, const float* A, int64_t lda, int M, int K, const char* img, int T_img, int t_begin, int Ncols,
              const float* bias, float* C, int64_t ldc, const unsigned* amax_a, const EpiArgs& epi, int num_cu) {
  constexpr int LDS = NCH * WS_NT * BLK;               // 16 KiB per chunk: 80 KiB at K = 160
  static int status = 1;
  auto kern = k_gemm_ws<WAVES, EPI, NCH>;
  if (status == 1)
    status = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess
                 ? DGDM_OK : DGDM_ERR_LAUNCH;
  if (status != DGDM_OK) return status;
  const int ntiles = (M + 31) / 32;
  const int per_cu = (160 * 1024) / LDS >= 2 && WAVES <= 4 ? 2 : 1;       // resident workgroups per CU (LDS, 8 waves)
  int grid = (ntiles + WAVES - 1) / WAVES;
  if (grid > per_cu * num_cu) grid = per_cu * num_cu;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * WAVES), LDS, s, A, lda, M, K, img, T_img, t_begin, Ncols, bias, C, ldc, amax_a, epi);
  return dgdm_launch_status();
}

template <int WAVES, int EPI>
int launch_ws_k(hipStream_t s, const float* A, int64_t lda, int M, int K, const char* img, int T_img, int t_begin, int Ncols,
                const float* bias, float* C, int64_t ldc, const unsigned* amax_a, const EpiArgs& epi, int num_cu) {
#define GO(N) return launch_ws<WAVES, EPI, N>(s, A, lda, M, K, img, T_img, t_begin, Ncols, bias, C, ldc, amax_a, epi, num_cu)
  switch ((K + 31) / 32) {
    case 1: GO(1);
    case 2: GO(2);
    case 3: GO(3);
    case 4: GO(4);
    default: GO(5);
  }
#undef GO
}

int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

}  // namespace

#ifndef DGDM_WS_WAVES
#define DGDM_WS_WAVES 4
#endif

int dgdm_gemm_ws_launch(int epi_kind, hipStream_t s, const float* A, int64_t lda, int M, int K, const char* img, int T_img, int t_begin,
                        int Ncols, const float* bias, float* C, int64_t ldc, const unsigned* amax_a, const EpiArgs& epi) {
  // MEASURED AND LEFT OFF (round 5, tools/microbench_epilogues.py, profiles/r05_epilogue_microbench.txt): against k_gemm_img<4,1> this
  // kernel is 1.3 - 3.2 us SLOWER per launch at every U-Net level (a wave owns a whole 32 x 128 output tile: one B-stage shared by
  // four waves and double-buffered by DMA already hides what the stationary image saves, and the image copy is a serial prologue).
  // Built only with -DDGDM_WS (tools/ab_ws.sh); the default library answers "unsupported" and the caller takes k_gemm_img.
#ifndef DGDM_WS
  return DGDM_ERR_UNSUPPORTED;
#else
  if (Ncols > 32 * WS_NT || K > 32 * WS_MAXCH || K < 16 || (K & 15) || (Ncols & 3) || (ldc & 3) || !dgdm_aligned16(C) || M <= 0 ||
      (bias && !dgdm_aligned16(bias)))
    return DGDM_ERR_UNSUPPORTED;
  if (epi_kind == EPI_NORM && epi.L > 32 * WS_NT) return DGDM_ERR_UNSUPPORTED;
  const int ncu = cu_count();
#define GO(E) return launch_ws_k<DGDM_WS_WAVES, E>(s, A, lda, M, K, img, T_img, t_begin, Ncols, bias, C, ldc, amax_a, epi, ncu)
  switch (epi_kind) {
    case EPI_NONE: GO(EPI_NONE);
    case EPI_ACT: GO(EPI_ACT);
    case EPI_ACTBWD: GO(EPI_ACTBWD);
    case EPI_NORM: GO(EPI_NORM);
    default: return DGDM_ERR_INVALID_ARG;
  }
#undef GO
#endif
}
