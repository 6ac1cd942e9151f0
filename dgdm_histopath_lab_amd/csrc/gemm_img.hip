// K3-img: the two "row" contractions of every nn.Linear of the hot path -- forward y = x W^T + b and dx = dy W -- with the WEIGHT
// operand pre-split once per step into an fp16 hi+lo "image" in MFMA fragment order, and the activation operand loaded straight
// from global memory into MFMA fragment registers.
//
// Why (profiles/r02f_pmc_valu.json): the register-staged kernels of gemm_h.hip convert BOTH operands fp32 -> fp16 hi+lo inside
// the main loop, once per 128 x 128 tile -- ~15 vector instructions per MFMA, the matrix pipe busy 11-15 % of the time, half of
// all wave cycles parked at the per-16-k barrier.  Here
//   * the weight side costs the main loop NOTHING: `dgdm_gemm_image_build` splits a weight (scaled by the power of two its amax
//     slot gives, exactly as gemm_h.hip does) into halfs laid out so that one 1 KiB block = one wave's B fragment of one MFMA;
//     a workgroup copies the blocks of its column range global -> LDS by LDS-DMA (no VGPRs, no VALU), 64 k per stage, double
//     buffered, ONE barrier per 64 k;
//   * the activation side is converted once per WAVE (32 rows x 32 k = 16 floats per lane, ~2.7 vector instructions per MFMA) and
//     never touches LDS: lane (row, kg) loads the 16 consecutive floats k = 32c + 16kg .. +15 of its row; MFMA j of the chunk
//     contracts floats 8j .. 8j+7.  The weight image uses the same (kg, j, i) -> k map, so the products pair up.
// Arithmetic is gemm_h.hip's: a.b = a_lo.b_hi + a_hi.b_lo + a_hi.b_hi, three v_mfma_f32_32x32x16_f16 per term, fp32 accumulate,
// operands scaled by 2^e with amax.2^e in [2^14, 2^15), accumulators unscaled in the epilogue.
//
// Image layout (bytes): [256-byte header: float scale_b] then blocks (c, t) of 4 KiB, c = 32-k chunk, t = 32-column tile, block
// index c * T + t (T = tiles of the whole image); inside a block four 1 KiB fragments (j, p), j = MFMA of the chunk, p = 0 hi /
// 1 lo, at (2j + p) KiB; lane l's 8 halfs at 16 l: B(col = 32t + (l & 31), k = 32c + 16 (l >> 5) + 8j + i), i = 0..7.
// The column count is zero-padded to a multiple of 32, K to a multiple of 64 (whole LDS stages).
#include "gemm_epi.hpp"
#include <type_traits>

namespace {


struct ImgDesc {
  const float* w0; const float* w1;     // sources (w1: second matrix of a column-concatenated weight, or null)
  long long ld0, ld1;
  const unsigned* amax0; const unsigned* amax1;
  char* img;
  int rows, cols0, cols1;               // source matrix [rows, cols0 (+ cols1)]
  int transposed;                       // 0: B(col, k) = W[col][k] (forward); 1: B(col, k) = W[k][col] (dx = dy . W)
  int block0;                           // first block of this image in the launch's block numbering
};

// one workgroup per image block (c, t): 32 columns x 32 k = 1024 elements, 4 per thread
__device__ __forceinline__ void image_block(const ImgDesc& d) {
  const int ncol = d.transposed ? d.cols0 : d.rows, nk = d.transposed ? d.rows : d.cols0 + d.cols1;
  const int T = (ncol + 31) >> 5;
  const int b = blockIdx.x - d.block0, c = b / T, t = b % T;
  unsigned u = amax_group(d.amax0);
  if (d.w1) { const unsigned u1 = amax_group(d.amax1); u = u > u1 ? u : u1; }
  const float sc = scale_of(u);
  if (b == 0 && threadIdx.x == 0) *reinterpret_cast<float*>(d.img) = sc;
  const int tid = threadIdx.x, j = tid >> 7, l = (tid >> 1) & 63, h = tid & 1;
  const int col = 32 * t + (l & 31), k0 = 32 * c + 16 * (l >> 5) + 8 * j + 4 * h;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (col < ncol) {
    if (!d.transposed) {
      if (k0 < d.cols0) {
        const float4 x = *reinterpret_cast<const float4*>(d.w0 + (long long)col * d.ld0 + k0);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
      } else if (k0 < nk) {
        const float4 x = *reinterpret_cast<const float4*>(d.w1 + (long long)col * d.ld1 + (k0 - d.cols0));
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (k0 + i < nk) v[i] = d.w0[(long long)(k0 + i) * d.ld0 + col];
    }
  }
  uint2 hi, lo;
  split_pair(v[0] * sc, v[1] * sc, &hi.x, &lo.x);
  split_pair(v[2] * sc, v[3] * sc, &hi.y, &lo.y);
  char* dst = d.img + IMG_HDR + (size_t)b * BLK + (2 * j) * 1024 + 16 * l + 8 * h;
  *reinterpret_cast<uint2*>(dst) = hi;
  *reinterpret_cast<uint2*>(dst + 1024) = lo;
}

__global__ __launch_bounds__(256) void k_image_build_many(const ImgDesc* __restrict__ table, int count) {
  int lo = 0, hi = count - 1;                    // last record with block0 <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ImgDesc d = table[lo];
  image_block(d);
}

__global__ __launch_bounds__(256) void k_image_build_one(const ImgDesc d) { image_block(d); }

// In-kernel time stamps for tools/ubench/gemm_img_stamps.hip (a diagnostic build of this file; in the library no stamp executes):
// workgroup 0, wave 0 writes (s_memtime, s_memrealtime) pairs to dgdm_stamp_buf.
#ifdef DGDM_GEMM_IMG_STAMPS
__device__ unsigned long long* dgdm_stamp_buf;
#define DGDM_STAMP(i_)                                                                                    \
  if (blockIdx.x == 0 && threadIdx.x == 0) {                                                              \
    dgdm_stamp_buf[2 * (i_)] = __builtin_amdgcn_s_memtime();                                              \
    dgdm_stamp_buf[2 * (i_) + 1] = __builtin_amdgcn_s_memrealtime();                                      \
  }
#else
#define DGDM_STAMP(i_)
#endif

// widest output the narrow kernel (wave = 32 x 128) takes; beyond it the wide one (wave = 32 x 256).  Round 4, same-box A/B
// (tools/build_variant_lib.sh -DDGDM_IMG_NARROW_MAX=128 vs 256, tools/microbench_gemm.py): at N = 256 and M = 40 000 the wide
// kernel has 313 workgroups for 512 resident slots, the narrow one 626 half-size ones: 54.8 -> 52.2, 36.0 -> 32.8 / 35.5,
// 48.3 -> 47.0 us (K = 544 / 288 / 512).  A small, consistent gain; beyond 256 columns the wide kernel wins by 25-40 %.
// register sets of activation rows in flight in the narrow kernel (stages of 64 k).  1 (shipped): stage s + 1 is loaded inside stage s.
// 2 (round 6, measured and left off): stage s + 2 -- see the kernel.
#ifndef DGDM_IMG_DEPTH
#define DGDM_IMG_DEPTH 1
#endif
#ifndef DGDM_IMG_NARROW_MAX
#define DGDM_IMG_NARROW_MAX 256
#endif
// narrow kernel's shape: WM x WN waves per workgroup, NTW 32-column tiles per wave
#ifndef DGDM_IMG_WM
#define DGDM_IMG_WM 4
#define DGDM_IMG_WN 1
#define DGDM_IMG_NTW 4
#endif
constexpr int CPS = 2;    // 32-k chunks per LDS stage (64 k); images are padded to whole stages
// 1 (round 6): every column tile's products of a stage / chunk form a block of their own that the vector unit adds to the running
// accumulator; 0: every MFMA accumulates into the running accumulator (rounds 2-5).  See DGDM_IMG8_INNER.
#ifndef DGDM_IMG_FRESH
#define DGDM_IMG_FRESH 1
#endif

// C[M, Ncols] (+)= A[M, K] . B + bias, B = image tiles [t_begin, t_begin + ceil(Ncols / 32)) of an image with T_img tiles per chunk.
// Workgroup = WM x WN waves; wave (wm, wn) owns rows 32 (WM rowtile + wm) .. +31 and columns 128 (WN colgroup + wn) .. +127.
// The main loop has no data-dependent branch: a wave always runs its four column tiles (tiles past the end of the matrix
// multiply whatever the LDS holds and are never stored), A loads past K re-read the row's last float4 (the image is zero there).
template <int WM, int WN, int NTW_, bool ACCUM, int EPI>
__global__ __launch_bounds__(64 * WM * WN, WM * WN > 4 ? 1 : 2) void k_gemm_img(const float* __restrict__ A, int64_t lda, int M, int K,
                                                            const char* __restrict__ img, int T_img, int t_begin, int Ncols,
                                                            const float* __restrict__ bias, float* __restrict__ C, int64_t ldc,
                                                            const unsigned* __restrict__ amax_a, const EpiArgs epi) {
  static_assert(EPI == EPI_NONE || !ACCUM, "the fused epilogues write C, they do not accumulate into it");
  constexpr bool TR = EPI != EPI_NONE;
  constexpr bool FRESH = DGDM_IMG_FRESH && EPI == EPI_NONE;      // the fused-epilogue variants (A/B only) keep round 5's accumulation
  constexpr int WAVES = WM * WN, NT_WG = NTW_ * WN, STAGE = CPS * NT_WG * BLK;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * STAGE
  const int tid = threadIdx.x, lane = tid & 63;
  DGDM_STAMP(0)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int gcol = (Ncols + 32 * NTW_ * WN - 1) / (32 * NTW_ * WN);          // column groups
  const int rowtile = blockIdx.x / gcol, colgroup = blockIdx.x % gcol;   // column groups of one row tile run side by side
  const int r0 = (rowtile * WM + wm) * 32;
  const int tiles = (Ncols + 31) >> 5;
  const int tg0 = colgroup * NT_WG;                               // first tile (relative to t_begin) of this workgroup
  const int live_wg = min(NT_WG, tiles - tg0);                    // tiles this workgroup stages
  const int nst = (K + 32 * CPS - 1) / (32 * CPS);
  const char* blocks = img + IMG_HDR + (size_t)(t_begin + tg0) * BLK;
  DGDM_STAMP(1)

  // LDS-DMA of stage s: chunk cc -> blocks [c * T_img + t_begin + tg0, + live_wg) -> smem[buf][cc][0 .. live_wg); 1 KiB per wave instruction
  auto stage_dma = [&](int s, int buf) {
#pragma unroll
    for (int cc = 0; cc < CPS; ++cc) {
      const char* src = blocks + (size_t)(s * CPS + cc) * T_img * BLK + lane * 16;
      char* dst = smem + buf * STAGE + cc * NT_WG * BLK;
#pragma unroll
      for (int p0 = 0; p0 < NT_WG * 4; p0 += WAVES) {
        const int p = p0 + wave;
        if (p < live_wg * 4)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * 1024),
                                           (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
      }
    }
  };

  // this lane's A row (clamped); its 16 floats of chunk c start at k = 32c + 16 (lane >> 5), clamped into the row (K % 16 == 0)
  const int row = min(r0 + (lane & 31), M - 1);
  const float* arow = A + (int64_t)row * lda;
  const int klane = 16 * (lane >> 5);
  // The A loads are inline asm: hipcc otherwise sinks them to the end of the loop body (right in front of the wait that
  // retires them) and, next to an LDS-DMA in flight, waits vmcnt(0) at every use.  Completion is by hand: the ONE
  // `s_waitcnt vmcnt(0)` at the top of a stage retires the DMA of that stage and both register sets (issued at least 24
  // MFMAs = 768 matrix-pipe cycles earlier); a set is only read after that wait and only overwritten after its conversion.
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  // DEPTH register sets of one stage (two chunks) each.  The in-kernel stamps (profiles/r06_gemm_img_stamps.txt) show every stage of this
  // kernel taking a memory round trip (1.6 us alone, 3.2 us with the chip loaded) around 0.8 us of arithmetic, so round 6 tried a second
  // set -- the loads of stage s + 2 issued inside stage s, the wait at the top of a stage COUNTED (the last 8 loads issued are always the
  // younger set's, behind the stage's LDS-DMA pieces: `vmcnt(8)` retires exactly this stage's DMA and set).  Correct (canary and parity
  // tests), 236 registers, and SLOWER: 17.6 against 15.9 us at 40 000 x 160 x 128 alone, 12.90 against 12.71 ms per step (five
  // alternating pairs, profiles/r06_img_depth_ab.txt).  With twice the loads in flight the first stage arrives later (5.4 against 3.7 us)
  // and the later ones no sooner: what a stage waits for under load is not the distance of its prefetch but the rate at which these
  // fragment-shaped loads (64 half-lines per wave instruction) are served.
  constexpr int DEPTH = DGDM_IMG_DEPTH;
  f32x4 ar[DEPTH][8];
#define DGDM_LOAD_CHUNK(c_, r0_, r1_, r2_, r3_)                                                                     \
  {                                                                                                                 \
    const float* p__ = arow + min(32 * (c_) + klane, K - 16);                                                       \
    DGDM_CANARY_POISON(r0_, r1_, r2_, r3_)                                                                            \
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"                \
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"            \
                 : DGDM_CANARY_OUT(r0_), DGDM_CANARY_OUT(r1_), DGDM_CANARY_OUT(r2_), DGDM_CANARY_OUT(r3_) : "v"(p__) : "memory"); \
  }
#define DGDM_CONVERT(cc_, r0_, r1_, r2_, r3_)                                                                       \
  {                                                                                                                 \
    uint4 h__, l__;                                                                                                 \
    split_pair(r0_[0] * sca, r0_[1] * sca, &h__.x, &l__.x);                                                         \
    split_pair(r0_[2] * sca, r0_[3] * sca, &h__.y, &l__.y);                                                         \
    split_pair(r1_[0] * sca, r1_[1] * sca, &h__.z, &l__.z);                                                         \
    split_pair(r1_[2] * sca, r1_[3] * sca, &h__.w, &l__.w);                                                         \
    ah[cc_][0] = __builtin_bit_cast(f16x8, h__);                                                                    \
    al[cc_][0] = __builtin_bit_cast(f16x8, l__);                                                                    \
    split_pair(r2_[0] * sca, r2_[1] * sca, &h__.x, &l__.x);                                                         \
    split_pair(r2_[2] * sca, r2_[3] * sca, &h__.y, &l__.y);                                                         \
    split_pair(r3_[0] * sca, r3_[1] * sca, &h__.z, &l__.z);                                                         \
    split_pair(r3_[2] * sca, r3_[3] * sca, &h__.w, &l__.w);                                                         \
    ah[cc_][1] = __builtin_bit_cast(f16x8, h__);                                                                    \
    al[cc_][1] = __builtin_bit_cast(f16x8, l__);                                                                    \
  }

  f32x16 acc[NTW_];
#pragma unroll
  for (int t = 0; t < NTW_; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  float sca, scb;
  if (ACCUM) {   // C += ...: the accumulators START from C (times the operand scales: powers of two, exact), no epilogue pass.
    // Done BEFORE any asm load is in flight: whatever the register allocator spills or copies here is complete data.
    sca = scale_of(amax_group(amax_a));
    scb = *reinterpret_cast<const float*>(img);
    const float sc2 = sca * scb;
#pragma unroll
    for (int t = 0; t < NTW_; ++t) {
      const float* p = C + min(32 * (tg0 + wn * NTW_ + t) + (lane & 31), Ncols - 1);
#pragma unroll
      for (int r = 0; r < 16; ++r)      // unconditional loads from clamped rows (rows past M are never stored): no branch per element
        acc[t][r] = p[(int64_t)min(r0 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2), M - 1) * ldc] * sc2;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  stage_dma(0, 0);
#pragma unroll
  for (int q = 0; q < DEPTH; ++q) {
    DGDM_LOAD_CHUNK(q * CPS, ar[q][0], ar[q][1], ar[q][2], ar[q][3])
    DGDM_LOAD_CHUNK(q * CPS + 1, ar[q][4], ar[q][5], ar[q][6], ar[q][7])
  }
  if (!ACCUM) {  // the operand scales are read AFTER the first stage and the first two chunks are on their way: one memory latency, not two
    sca = scale_of(amax_group(amax_a));
    scb = *reinterpret_cast<const float*>(img);
  }
  uint32_t seed_v = 0, pseed_v = 0;     // the fused epilogues' dropout seeds: read here, not behind the main loop
  if (EPI != EPI_NONE) { seed_v = epi.seed.value(); pseed_v = EPI == EPI_NORM ? epi.pre_seed.value() : 0u; }

  auto stage_body = [&](auto SET, const int s) {
    constexpr int q = decltype(SET)::value;
    // stage s has landed (this wave's pieces: vmcnt; everybody's: the barrier), and every wave is done reading the other buffer
    if constexpr (DEPTH == 1)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(ar[q][0]), "+v"(ar[q][1]), "+v"(ar[q][2]), "+v"(ar[q][3]), "+v"(ar[q][4]), "+v"(ar[q][5]), "+v"(ar[q][6]), "+v"(ar[q][7]) :: "memory");
    else
      asm volatile("s_waitcnt vmcnt(8)" : "+v"(ar[q][0]), "+v"(ar[q][1]), "+v"(ar[q][2]), "+v"(ar[q][3]), "+v"(ar[q][4]), "+v"(ar[q][5]), "+v"(ar[q][6]), "+v"(ar[q][7]) :: "memory");
    __builtin_amdgcn_s_barrier();
    DGDM_STAMP(2 + (s < 5 ? s : 5))
    if (s + 1 < nst) stage_dma(s + 1, (s + 1) & 1);
    const char* buf = smem + (s & 1) * STAGE + (wn * NTW_) * BLK + lane * 16;
    f16x8 ah[CPS][2], al[CPS][2];
    // Four batches b = (cc, j) of 8 fragment reads (4 column tiles x hi / lo) + 12 MFMAs.  The reads of batch b + 1 are issued
    // BEFORE the MFMAs of batch b and pinned there (sched_barrier): left alone, hipcc sinks every read to just in front of its
    // MFMA and the wave eats the LDS latency 16 times per stage.
    f16x8 bh[2][FRESH ? 4 : NTW_], bl[2][FRESH ? 4 : NTW_];      // FRESH: [slot][k16-step of the stage]
#define DGDM_READ_BATCH(b_, slot_)                                                                                  \
  {                                                                                                                 \
    const char* q__ = buf + ((b_) >> 1) * NT_WG * BLK + (2 * ((b_) & 1)) * 1024;                                    \
    _Pragma("unroll") for (int t = 0; t < NTW_; ++t) {                                                               \
      bh[slot_][t] = *reinterpret_cast<const f16x8*>(q__ + t * BLK);                                                \
      bl[slot_][t] = *reinterpret_cast<const f16x8*>(q__ + t * BLK + 1024);                                         \
    }                                                                                                               \
  }
#define DGDM_MFMA_BATCH(b_, slot_)                                                                                  \
  _Pragma("unroll") for (int t = 0; t < NTW_; ++t) {                                                                 \
    acc[t] = mfma_o<TR>(al[(b_) >> 1][(b_) & 1], bh[slot_][t], acc[t]);    /* smaller terms first */                 \
    acc[t] = mfma_o<TR>(ah[(b_) >> 1][(b_) & 1], bl[slot_][t], acc[t]);                                             \
    acc[t] = mfma_o<TR>(ah[(b_) >> 1][(b_) & 1], bh[slot_][t], acc[t]);                                             \
  }
    if constexpr (FRESH) {
    // Round 6 (the accumulate of the matrix pipe: tools/ubench/mfma_rounding.hip).  A column tile's products of the whole stage
    // (2 chunks x 2 k16-steps x {lo.hi, hi.lo, hi.hi} = 12 MFMAs) are chained from a ZERO accumulator into a block P of their own and
    // P is added to the running accumulator by the vector unit (round to nearest, one v_pk_add_f32 per two values, one tile late so
    // that it never waits for the matrix pipe): the long-lived accumulator sees K / 64 correctly rounded adds instead of 3 K / 16
    // matrix-pipe accumulations of twice the rounding error each.  Tile-major order; both chunks are converted before the first
    // tile (every tile needs all four A fragments).
    DGDM_CONVERT(0, ar[q][0], ar[q][1], ar[q][2], ar[q][3])
    DGDM_LOAD_CHUNK((s + DEPTH) * CPS, ar[q][0], ar[q][1], ar[q][2], ar[q][3])       // past the end: clamped re-reads, retired after the loop
    DGDM_CONVERT(1, ar[q][4], ar[q][5], ar[q][6], ar[q][7])
    DGDM_LOAD_CHUNK((s + DEPTH) * CPS + 1, ar[q][4], ar[q][5], ar[q][6], ar[q][7])
#define DGDM_READ_TILE(t_, slot_)                                                                                   \
  _Pragma("unroll") for (int st = 0; st < 4; ++st) {                                                                 \
    const char* q__ = buf + (st >> 1) * NT_WG * BLK + (2 * (st & 1)) * 1024 + (t_) * BLK;                            \
    bh[slot_][st] = *reinterpret_cast<const f16x8*>(q__);                                                           \
    bl[slot_][st] = *reinterpret_cast<const f16x8*>(q__ + 1024);                                                    \
  }
    f32x16 pblk[2];
    DGDM_READ_TILE(0, 0)
#pragma unroll
    for (int t = 0; t < NTW_; ++t) {
      __builtin_amdgcn_sched_barrier(0);
      f32x16 pb = mfma_o<TR>(al[0][0], bh[0][0], zero16);    /* smaller terms first */
      pb = mfma_o<TR>(ah[0][0], bl[0][0], pb);
      pb = mfma_o<TR>(al[0][1], bh[0][1], pb);
      pb = mfma_o<TR>(ah[0][1], bl[0][1], pb);
      pb = mfma_o<TR>(al[1][0], bh[0][2], pb);
      pb = mfma_o<TR>(ah[1][0], bl[0][2], pb);
      pb = mfma_o<TR>(al[1][1], bh[0][3], pb);
      pb = mfma_o<TR>(ah[1][1], bl[0][3], pb);
      pb = mfma_o<TR>(ah[0][0], bh[0][0], pb);
      pb = mfma_o<TR>(ah[0][1], bh[0][1], pb);
      pb = mfma_o<TR>(ah[1][0], bh[0][2], pb);
      pb = mfma_o<TR>(ah[1][1], bh[0][3], pb);
      pblk[t & 1] = pb;
      __builtin_amdgcn_sched_barrier(0);
      // the fragments are single-buffered: tile t + 1's eight are read right behind the ISSUE of tile t's MFMAs (which have taken
      // their operands by then) and land while those execute; tile t - 1's block -- complete since this chain started -- is added
      if (t + 1 < NTW_) DGDM_READ_TILE(t + 1, 0)
      if (t > 0) {
        acc[t - 1] += pblk[(t - 1) & 1];
        asm volatile("" : "+v"(acc[t - 1]));      // pins the add here (a pure node: it would float to the end of the stage and keep every block live)
      }
    }
    acc[NTW_ - 1] += pblk[(NTW_ - 1) & 1];
    asm volatile("" : "+v"(acc[NTW_ - 1]));
#undef DGDM_READ_TILE
    } else {
    DGDM_READ_BATCH(0, 0)
    DGDM_CONVERT(0, ar[q][0], ar[q][1], ar[q][2], ar[q][3])
    DGDM_LOAD_CHUNK((s + DEPTH) * CPS, ar[q][0], ar[q][1], ar[q][2], ar[q][3])       // past the end: clamped re-reads, retired after the loop
    __builtin_amdgcn_sched_barrier(0);
    DGDM_READ_BATCH(1, 1)
    __builtin_amdgcn_sched_barrier(0);
    DGDM_MFMA_BATCH(0, 0)
    DGDM_CONVERT(1, ar[q][4], ar[q][5], ar[q][6], ar[q][7])                      // scheduled among the MFMAs of batch 0
    DGDM_LOAD_CHUNK((s + DEPTH) * CPS + 1, ar[q][4], ar[q][5], ar[q][6], ar[q][7])
    __builtin_amdgcn_sched_barrier(0);
    DGDM_READ_BATCH(2, 0)
    __builtin_amdgcn_sched_barrier(0);
    DGDM_MFMA_BATCH(1, 1)
    __builtin_amdgcn_sched_barrier(0);
    DGDM_READ_BATCH(3, 1)
    __builtin_amdgcn_sched_barrier(0);
    DGDM_MFMA_BATCH(2, 0)
    __builtin_amdgcn_sched_barrier(0);
    DGDM_MFMA_BATCH(3, 1)
    }
  };
  for (int s = 0; s < nst; s += DEPTH) {
    stage_body(std::integral_constant<int, 0>{}, s);
    if constexpr (DEPTH == 2) {
      if (s + 1 < nst) stage_body(std::integral_constant<int, 1>{}, s + 1);
    }
  }
#undef DGDM_READ_BATCH
#undef DGDM_MFMA_BATCH
  // nothing may still be in flight into the a-registers when the epilogue reuses them (an asm load completes behind the
  // compiler's back: a late one would land in whatever the register holds by then -- a store address, for instance)
#pragma unroll
  for (int q = 0; q < DEPTH; ++q)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ar[q][0]), "+v"(ar[q][1]), "+v"(ar[q][2]), "+v"(ar[q][3]), "+v"(ar[q][4]), "+v"(ar[q][5]), "+v"(ar[q][6]), "+v"(ar[q][7]) :: "memory");
#undef DGDM_LOAD_CHUNK
#undef DGDM_CONVERT

  DGDM_STAMP(8)
  // Epilogue.  All 64 outputs of the wave are finished IN PLACE first and stored afterwards: a store whose data register is
  // recycled for the next value makes hipcc wait vmcnt(0) in front of every store (64 round trips = 7 us per launch, measured
  // with tools/ubench/gemm_img_stamps.hip -- more than the main loop of the U-Net's GEMMs).
  const float inv = (1.0f / sca) * (1.0f / scb);     // exact: powers of two
  if (TR) {
    unsigned am = 0;
    const EpiVecGlobal vb{bias, Ncols}, vg{epi.gamma, Ncols}, vbe{epi.beta, Ncols};
    epilogue_tr<NTW_, EPI>(acc, inv, r0 + (lane & 31), M, 32 * (tg0 + wn * NTW_), Ncols, vb, vg, vbe, C, ldc, epi, lane >> 5, am, seed_v, pseed_v);
    if (epi.amax_out) dgdm_amax_commit(am, epi.amax_out);   // workgroup-uniform condition: every thread reaches the barrier inside
    return;
  }
  const int jc = lane & 31, hi = lane >> 5;
  const int rbase = r0 + 4 * hi;
#pragma unroll
  for (int t = 0; t < NTW_; ++t) {
    const int col = 32 * (tg0 + wn * NTW_ + t) + jc;
    const float bv = (bias && col < Ncols) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = acc[t][r] * inv + bv;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < NTW_; ++t) {
    const int col = 32 * (tg0 + wn * NTW_ + t) + jc;
    float* p = C + (int64_t)rbase * ldc + col;
    if (col < Ncols) {
      if (r0 + 32 <= M) {          // wave-uniform: all 32 rows exist
#pragma unroll
        for (int r = 0; r < 16; ++r) p[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = acc[t][r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (rbase + (r & 3) + 8 * (r >> 2) < M) p[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = acc[t][r];
      }
    }
  }
  DGDM_STAMP(9)
}

// The inner loop of one 32-k chunk.  DGDM_IMG_FRESH (round 6, default): tile-major -- the six MFMAs of a column tile (2 k16-steps x
// {lo.hi, hi.lo, hi.hi}) are chained from a ZERO accumulator into a block of their own, which the vector unit adds to the running
// accumulator one tile later (round to nearest): K / 32 correctly rounded adds into the long-lived accumulator instead of 3 K / 16
// matrix-pipe accumulations with twice the rounding error each (tools/ubench/mfma_rounding.hip).  0: round 5's order (j-major, every
// MFMA accumulates into the tile's running accumulator).
#define DGDM_IMG8_INNER                                                                                             \
    if constexpr (FRESH) {                                                                                          \
    /* B fragments single-buffered: tile t + 1's four fragments are read right BEHIND the issue of tile t's six MFMAs (which  \
       have taken their operands by then) and land while those execute; the block of tile t - 1 -- complete since tile t's   \
       chain started -- is added meanwhile.  Two blocks in rotation: the add never waits for the matrix pipe. */              \
    f32x16 pblk[2];                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                 \
      bh[0][j] = *reinterpret_cast<const f16x8*>(buf + (2 * j) * 1024);                                             \
      bl[0][j] = *reinterpret_cast<const f16x8*>(buf + (2 * j) * 1024 + 1024);                                      \
    }                                                                                                               \
    _Pragma("unroll") for (int t = 0; t < NT8; ++t) {                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      f32x16 pb = mfma_o<TR>(al[0], bh[0][0], zero16);      /* smaller terms first */                                \
      pb = mfma_o<TR>(ah[0], bl[0][0], pb);                                                                         \
      pb = mfma_o<TR>(al[1], bh[0][1], pb);                                                                         \
      pb = mfma_o<TR>(ah[1], bl[0][1], pb);                                                                         \
      pb = mfma_o<TR>(ah[0], bh[0][0], pb);                                                                         \
      pb = mfma_o<TR>(ah[1], bh[0][1], pb);                                                                         \
      pblk[t & 1] = pb;                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      if (t + 1 < NT8) {                                                                                            \
        const char* q__ = buf + (t + 1) * BLK;                                                                      \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                             \
          bh[0][j] = *reinterpret_cast<const f16x8*>(q__ + (2 * j) * 1024);                                         \
          bl[0][j] = *reinterpret_cast<const f16x8*>(q__ + (2 * j) * 1024 + 1024);                                  \
        }                                                                                                           \
      }                                                                                                             \
      if (t > 0) {                                                                                                  \
        acc[t - 1] += pblk[(t - 1) & 1];                                                                            \
        asm volatile("" : "+v"(acc[t - 1]));      /* pins the add HERE (a pure node: it would float to the end of the chunk and keep every block live) */ \
      }                                                                                                             \
    }                                                                                                               \
    acc[NT8 - 1] += pblk[(NT8 - 1) & 1];                                                                            \
    asm volatile("" : "+v"(acc[NT8 - 1]));                                                                          \
    } else {                                                                                                        \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                 \
      bh[0][u] = *reinterpret_cast<const f16x8*>(buf + u * BLK);                                                    \
      bl[0][u] = *reinterpret_cast<const f16x8*>(buf + u * BLK + 1024);                                             \
    }                                                                                                               \
    _Pragma("unroll") for (int b = 0; b < 8; ++b) {          /* b = 4 j + tile pair */                               \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      if (b + 1 < 8) {                                                                                              \
        const char* q__ = buf + (2 * ((b + 1) & 3)) * BLK + (2 * ((b + 1) >> 2)) * 1024;                            \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                             \
          bh[(b + 1) & 1][u] = *reinterpret_cast<const f16x8*>(q__ + u * BLK);                                      \
          bl[(b + 1) & 1][u] = *reinterpret_cast<const f16x8*>(q__ + u * BLK + 1024);                               \
        }                                                                                                           \
      }                                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                               \
        const int t = 2 * (b & 3) + u;                                                                              \
        acc[t] = mfma_o<TR>(al[b >> 2], bh[b & 1][u], acc[t]);   /* smaller terms first */                          \
        acc[t] = mfma_o<TR>(ah[b >> 2], bl[b & 1][u], acc[t]);                                                      \
        acc[t] = mfma_o<TR>(ah[b >> 2], bh[b & 1][u], acc[t]);                                                      \
      }                                                                                                             \
    }                                                                                          \
    }

// ---- the wide variant: a wave owns 32 rows x 256 columns (8 accumulator tiles), a workgroup of four waves 128 x 256, ONE 32-k
// chunk per LDS stage (32 KiB, two slots) so that TWO workgroups share a CU.  Why two independent workgroups instead of one of
// eight waves: the two waves of a SIMD that belong to one workgroup run in lockstep behind the stage barrier -- both stage and
// convert, then both want the matrix pipe (stamps: 5,900 cycles per 64 k against 3,072 of MFMA) -- while waves of different
// workgroups drift apart and one's staging / conversion / LDS latency hides under the other's MFMAs.  Against the 4 x 2-wave form
// the conversion work per MFMA halves as well (16 floats per lane feed 48 MFMAs instead of 24).
constexpr int NT8 = 8;
template <bool ACCUM, int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_img8(const float* __restrict__ A, int64_t lda, int M, int K, const char* __restrict__ img,
                                                      int T_img, int t_begin, int Ncols, const float* __restrict__ bias,
                                                      float* __restrict__ C, int64_t ldc, const unsigned* __restrict__ amax_a,
                                                      const EpiArgs epi) {
  static_assert(EPI == EPI_NONE || !ACCUM, "the fused epilogues write C, they do not accumulate into it");
  constexpr bool TR = EPI != EPI_NONE;
  constexpr bool FRESH = DGDM_IMG_FRESH && EPI == EPI_NONE;      // the fused-epilogue variants (A/B only) keep round 5's accumulation
  constexpr int SLOT = NT8 * BLK;                                  // 32 KiB: one chunk of 8 column tiles
  extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 * SLOT
  const int tid = threadIdx.x, lane = tid & 63;
  DGDM_STAMP(0)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gcol = (Ncols + 255) / 256;
  const int rowtile = blockIdx.x / gcol, colgroup = blockIdx.x % gcol;
  const int r0 = (rowtile * 4 + wave) * 32;
  const int tiles = (Ncols + 31) >> 5;
  const int tg0 = colgroup * NT8;
  const int live_wg = min(NT8, tiles - tg0);
  const int nchunks = 2 * ((K + 63) / 64);                         // images are padded to whole 64-k stages: always even
  const char* blocks = img + IMG_HDR + (size_t)(t_begin + tg0) * BLK + lane * 16;

  auto chunk_dma = [&](int c, int slot) {
    const char* src = blocks + (size_t)c * T_img * BLK;
    char* dst = smem + slot * SLOT;
#pragma unroll
    for (int p0 = 0; p0 < NT8 * 4; p0 += 4) {
      const int p = p0 + wave;
      if (p < live_wg * 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
  };

  const int row = min(r0 + (lane & 31), M - 1);
  const float* arow = A + (int64_t)row * lda;
  const int klane = 16 * (lane >> 5);
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 a00, a01, a02, a03, a10, a11, a12, a13;
#define DGDM_LOAD_CHUNK(c_, r0_, r1_, r2_, r3_)                                                                     \
  {                                                                                                                 \
    const float* p__ = arow + min(32 * (c_) + klane, K - 16);                                                       \
    DGDM_CANARY_POISON(r0_, r1_, r2_, r3_)                                                                            \
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"                \
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"            \
                 : DGDM_CANARY_OUT(r0_), DGDM_CANARY_OUT(r1_), DGDM_CANARY_OUT(r2_), DGDM_CANARY_OUT(r3_) : "v"(p__) : "memory"); \
  }
  f32x16 acc[NT8];
#pragma unroll
  for (int t = 0; t < NT8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  float sca, scb;
  if (ACCUM) {   // C += ...: the accumulators START from C (times the operand scales: powers of two, exact), no epilogue pass.
    // Done BEFORE any asm load is in flight: whatever the register allocator spills or copies here is complete data.
    sca = scale_of(amax_group(amax_a));
    scb = *reinterpret_cast<const float*>(img);
    const float sc2 = sca * scb;
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      const float* p = C + min(32 * (tg0 + t) + (lane & 31), Ncols - 1);
#pragma unroll
      for (int r = 0; r < 16; ++r)      // unconditional loads from clamped rows (rows past M are never stored): no branch per element
        acc[t][r] = p[(int64_t)min(r0 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2), M - 1) * ldc] * sc2;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  chunk_dma(0, 0);
  DGDM_LOAD_CHUNK(0, a00, a01, a02, a03)
  DGDM_LOAD_CHUNK(1, a10, a11, a12, a13)
  if (!ACCUM) {  // the operand scales are read AFTER the first stage and the first two chunks are on their way: one memory latency, not two
    sca = scale_of(amax_group(amax_a));
    scb = *reinterpret_cast<const float*>(img);
  }
  uint32_t seed_v = 0, pseed_v = 0;     // the fused epilogues' dropout seeds: read here, not behind the main loop
  if (EPI != EPI_NONE) { seed_v = epi.seed.value(); pseed_v = EPI == EPI_NORM ? epi.pre_seed.value() : 0u; }

  // one chunk: wait for its B slot (and the A set), restage the other slot, convert the A set and refill it two chunks ahead,
  // then 8 batches (j, tile pair) of four fragment reads + six MFMAs, the reads one batch ahead
#define DGDM_CHUNK(c_, r0_, r1_, r2_, r3_)                                                                          \
  {                                                                                                                 \
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a00), "+v"(a01), "+v"(a02), "+v"(a03), "+v"(a10), "+v"(a11), "+v"(a12), "+v"(a13) :: "memory"); \
    __builtin_amdgcn_s_barrier();                                                                                   \
    DGDM_STAMP(2 + ((c_) < 5 ? (c_) : 5))                                                                           \
    if ((c_) + 1 < nchunks) chunk_dma((c_) + 1, ((c_) + 1) & 1);                                                    \
    const char* buf = smem + ((c_) & 1) * SLOT + lane * 16;                                                         \
    f16x8 ah[2], al[2];                                                                                             \
    {                                                                                                               \
      uint4 h__, l__;                                                                                               \
      split_pair(r0_[0] * sca, r0_[1] * sca, &h__.x, &l__.x);                                                       \
      split_pair(r0_[2] * sca, r0_[3] * sca, &h__.y, &l__.y);                                                       \
      split_pair(r1_[0] * sca, r1_[1] * sca, &h__.z, &l__.z);                                                       \
      split_pair(r1_[2] * sca, r1_[3] * sca, &h__.w, &l__.w);                                                       \
      ah[0] = __builtin_bit_cast(f16x8, h__);                                                                       \
      al[0] = __builtin_bit_cast(f16x8, l__);                                                                       \
      split_pair(r2_[0] * sca, r2_[1] * sca, &h__.x, &l__.x);                                                       \
      split_pair(r2_[2] * sca, r2_[3] * sca, &h__.y, &l__.y);                                                       \
      split_pair(r3_[0] * sca, r3_[1] * sca, &h__.z, &l__.z);                                                       \
      split_pair(r3_[2] * sca, r3_[3] * sca, &h__.w, &l__.w);                                                       \
      ah[1] = __builtin_bit_cast(f16x8, h__);                                                                       \
      al[1] = __builtin_bit_cast(f16x8, l__);                                                                       \
    }                                                                                                               \
    DGDM_LOAD_CHUNK((c_) + 2, r0_, r1_, r2_, r3_)      /* past the end: clamped re-reads, retired after the loop */  \
    f16x8 bh[2][2], bl[2][2];                                                                                       \
    DGDM_IMG8_INNER                                                                                                 \
  }
  for (int c = 0; c < nchunks; c += 2) {
    DGDM_CHUNK(c, a00, a01, a02, a03)
    DGDM_CHUNK(c + 1, a10, a11, a12, a13)
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a00), "+v"(a01), "+v"(a02), "+v"(a03), "+v"(a10), "+v"(a11), "+v"(a12), "+v"(a13) :: "memory");
#undef DGDM_CHUNK
#undef DGDM_LOAD_CHUNK
  DGDM_STAMP(8)

  const float inv = (1.0f / sca) * (1.0f / scb);
  if (TR) {
    unsigned am = 0;
    const EpiVecGlobal vb{bias, Ncols}, vg{epi.gamma, Ncols}, vbe{epi.beta, Ncols};
    epilogue_tr<NT8, EPI>(acc, inv, r0 + (lane & 31), M, 32 * tg0, Ncols, vb, vg, vbe, C, ldc, epi, lane >> 5, am, seed_v, pseed_v);
    if (epi.amax_out) dgdm_amax_commit(am, epi.amax_out);
    return;
  }
  const int jc = lane & 31, hi = lane >> 5;
  const int rbase = r0 + 4 * hi;
#pragma unroll
  for (int t = 0; t < NT8; ++t) {
    const int col = 32 * (tg0 + t) + jc;
    const float bv = (bias && col < Ncols) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = acc[t][r] * inv + bv;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < NT8; ++t) {
    const int col = 32 * (tg0 + t) + jc;
    float* p = C + (int64_t)rbase * ldc + col;
    if (col < Ncols) {
      if (r0 + 32 <= M) {
#pragma unroll
        for (int r = 0; r < 16; ++r) p[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = acc[t][r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (rbase + (r & 3) + 8 * (r >> 2) < M) p[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = acc[t][r];
      }
    }
  }
  DGDM_STAMP(9)
}

template <int EPI>
int launch_img8(hipStream_t s, const float* A, int64_t lda, int M, int K, const char* img, int T_img, int t_begin, int Ncols,
                const float* bias, float* C, int64_t ldc, int accumulate, const unsigned* amax_a, const EpiArgs& epi) {
  constexpr int LDS = 2 * NT8 * BLK;
  static int status[2] = {1, 1};
  auto kern = (EPI == EPI_NONE && accumulate) ? k_gemm_img8<EPI == EPI_NONE, EPI> : k_gemm_img8<false, EPI>;
  int& st = status[accumulate ? 1 : 0];
  if (st == 1)
    st = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess
             ? DGDM_OK : DGDM_ERR_LAUNCH;
  if (st != DGDM_OK) return st;
  const int gcol = (Ncols + 255) / 256, grow = (M + 127) / 128;
  hipLaunchKernelGGL(kern, dim3((unsigned)(gcol * grow)), dim3(256), LDS, s, A, lda, M, K, img, T_img, t_begin, Ncols, bias, C, ldc, amax_a,
                     epi);
  return dgdm_launch_status();
}

template <int WM, int WN, int NTW_, int EPI>
int launch_img(hipStream_t s, const float* A, int64_t lda, int M, int K, const char* img, int T_img, int t_begin, int Ncols,
               const float* bias, float* C, int64_t ldc, int accumulate, const unsigned* amax_a, const EpiArgs& epi) {
  constexpr int LDS = 2 * CPS * NTW_ * WN * BLK;
  static int status[2] = {1, 1};
  auto kern = (EPI == EPI_NONE && accumulate) ? k_gemm_img<WM, WN, NTW_, EPI == EPI_NONE, EPI> : k_gemm_img<WM, WN, NTW_, false, EPI>;
  int& st = status[accumulate ? 1 : 0];
  if (st == 1)
    st = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess
             ? DGDM_OK : DGDM_ERR_LAUNCH;
  if (st != DGDM_OK) return st;
  const int gcol = (Ncols + 32 * NTW_ * WN - 1) / (32 * NTW_ * WN), grow = (M + 32 * WM - 1) / (32 * WM);
  hipLaunchKernelGGL(kern, dim3((unsigned)(gcol * grow)), dim3(64 * WM * WN), LDS, s, A, lda, M, K, img, T_img, t_begin, Ncols, bias, C,
                     ldc, amax_a, epi);
  return dgdm_launch_status();
}

}  // namespace

extern "C" size_t dgdm_gemm_image_bytes(int32_t cols, int32_t k) {
  if (cols <= 0 || k <= 0) return 0;
  return (size_t)IMG_HDR + (size_t)(((int64_t)cols + 31) / 32) * (size_t)(2 * (((int64_t)k + 63) / 64)) * BLK;
}

extern "C" int32_t dgdm_gemm_image_blocks(int32_t cols, int32_t k) {
  if (cols <= 0 || k <= 0) return 0;
  const int64_t blocks = (((int64_t)cols + 31) / 32) * (2 * (((int64_t)k + 63) / 64));
  return blocks > 0x7fffffff ? 0 : (int32_t)blocks;       // an image that large cannot be built (0 = "none", as for empty shapes)
}

extern "C" int dgdm_gemm_image_build_many(const void* table, int32_t count, int32_t total_blocks, void* stream) {
  if (count < 0 || total_blocks < 0 || (count > 0 && !table)) return DGDM_ERR_INVALID_ARG;
  if (count == 0 || total_blocks == 0) return DGDM_OK;
  hipLaunchKernelGGL(k_image_build_many, dim3((unsigned)total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const ImgDesc*>(table), count);
  return dgdm_launch_status();
}

extern "C" int dgdm_gemm_image_build(const float* w0, int64_t ld0, const float* w1, int64_t ld1, const uint32_t* amax0,
                                     const uint32_t* amax1, void* image, int32_t rows, int32_t cols0, int32_t cols1, int32_t transposed,
                                     void* stream) {
  if (rows < 0 || cols0 < 0 || cols1 < 0) return DGDM_ERR_INVALID_ARG;
  if (rows == 0 || cols0 == 0) return DGDM_OK;
  if (!w0 || !amax0 || !image || (cols1 > 0 && (!w1 || !amax1 || transposed))) return DGDM_ERR_INVALID_ARG;
  if ((cols0 & 3) || (cols1 & 3) || (ld0 & 3) || (ld1 & 3) || ld0 < cols0 || (cols1 > 0 && ld1 < cols1) || !dgdm_aligned16(w0) ||
      (cols1 > 0 && !dgdm_aligned16(w1)) || !dgdm_aligned16(image))
    return DGDM_ERR_UNSUPPORTED;
  ImgDesc d;
  d.w0 = w0; d.w1 = cols1 > 0 ? w1 : nullptr; d.ld0 = ld0; d.ld1 = ld1; d.amax0 = amax0; d.amax1 = amax1;
  d.img = static_cast<char*>(image); d.rows = rows; d.cols0 = cols0; d.cols1 = cols1; d.transposed = transposed; d.block0 = 0;
  const int blocks = transposed ? dgdm_gemm_image_blocks(cols0, rows) : dgdm_gemm_image_blocks(rows, cols0 + cols1);
  hipLaunchKernelGGL(k_image_build_one, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d);
  return dgdm_launch_status();
}

extern "C" int dgdm_gemm_rows_img(const float* A, int64_t lda, int32_t M, int32_t K, const void* image, int32_t image_tiles,
                                  int32_t tile_begin, int32_t ncols, const float* bias, float* C, int64_t ldc, int32_t accumulate,
                                  const uint32_t* amax_a, void* stream) {
  if (M < 0 || K < 0 || ncols < 0 || image_tiles <= 0 || tile_begin < 0) return DGDM_ERR_INVALID_ARG;
  if (M == 0 || ncols == 0) return DGDM_OK;
  if (!A || !image || !C || !amax_a) return DGDM_ERR_INVALID_ARG;
  if (K == 0) return DGDM_ERR_UNSUPPORTED;
  if ((K & 15) || (lda & 3) || !dgdm_aligned16(A) || !dgdm_aligned16(image)) return DGDM_ERR_UNSUPPORTED;
  if (lda < K || ldc < ncols || tile_begin + (ncols + 31) / 32 > image_tiles) return DGDM_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const char* img = static_cast<const char*>(image);
  const EpiArgs none{};
  if (ncols <= DGDM_IMG_NARROW_MAX)
    return launch_img<DGDM_IMG_WM, DGDM_IMG_WN, DGDM_IMG_NTW, EPI_NONE>(s, A, lda, M, K, img, image_tiles, tile_begin, ncols, bias, C, ldc, accumulate, amax_a, none);
  return launch_img8<EPI_NONE>(s, A, lda, M, K, img, image_tiles, tile_begin, ncols, bias, C, ldc, accumulate, amax_a, none);
}

namespace {

int epi_common_checks(const float* A, int64_t lda, int32_t M, int32_t K, const void* image, int32_t image_tiles, int32_t tile_begin,
                      int32_t ncols, float* C, int64_t ldc, const uint32_t* amax_a, int32_t act, float drop_p) {
  if (M < 0 || K < 0 || ncols < 0 || image_tiles <= 0 || tile_begin < 0 || act < 0 || act > DGDM_ACT_ELU || !(drop_p >= 0.f && drop_p < 1.f))
    return DGDM_ERR_INVALID_ARG;
  if (M == 0 || ncols == 0) return 1;    // nothing to do
  if (!A || !image || !C || !amax_a) return DGDM_ERR_INVALID_ARG;
  if (K == 0) return DGDM_ERR_UNSUPPORTED;
  if ((K & 15) || (lda & 3) || (ncols & 3) || (ldc & 3) || !dgdm_aligned16(A) || !dgdm_aligned16(image) || !dgdm_aligned16(C))
    return DGDM_ERR_UNSUPPORTED;
  if (lda < K || ldc < ncols || tile_begin + (ncols + 31) / 32 > image_tiles) return DGDM_ERR_INVALID_ARG;
  return DGDM_OK;
}

}  // namespace

extern "C" int dgdm_gemm_rows_img_act(const float* A, int64_t lda, int32_t M, int32_t K, const void* image, int32_t image_tiles,
                                      int32_t tile_begin, int32_t ncols, const float* bias, float* pre, int64_t ldp, float* Y,
                                      int64_t ldy, int32_t act, float drop_p, uint32_t seed, const uint32_t* amax_a,
                                      uint32_t* amax_y, void* stream) {
  const int st = epi_common_checks(A, lda, M, K, image, image_tiles, tile_begin, ncols, Y, ldy, amax_a, act, drop_p);
  if (st != DGDM_OK) return st > 0 ? DGDM_OK : st;
  if (bias && !dgdm_aligned16(bias)) return DGDM_ERR_UNSUPPORTED;
  if (pre && ((ldp & 3) || ldp < ncols || !dgdm_aligned16(pre))) return DGDM_ERR_UNSUPPORTED;
  EpiArgs e{};
  e.pre_out = pre; e.ldp = ldp; e.act = act; e.drop_p = drop_p; e.seed = dgdm_seed_arg(seed); e.amax_out = amax_y;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const char* img = static_cast<const char*>(image);
  if (ncols <= DGDM_IMG_NARROW_MAX)
    return launch_img<DGDM_IMG_WM, DGDM_IMG_WN, DGDM_IMG_NTW, EPI_ACT>(s, A, lda, M, K, img, image_tiles, tile_begin, ncols, bias, Y, ldy, 0, amax_a, e);
  return launch_img8<EPI_ACT>(s, A, lda, M, K, img, image_tiles, tile_begin, ncols, bias, Y, ldy, 0, amax_a, e);
}

extern "C" int dgdm_gemm_rows_img_act_bwd(const float* A, int64_t lda, int32_t M, int32_t K, const void* image, int32_t image_tiles,
                                          int32_t tile_begin, int32_t ncols, const float* pre, int64_t ldp, float* G, int64_t ldg,
                                          int32_t act, float drop_p, uint32_t seed, const uint32_t* amax_a, uint32_t* amax_g,
                                          void* stream) {
  const int st = epi_common_checks(A, lda, M, K, image, image_tiles, tile_begin, ncols, G, ldg, amax_a, act, drop_p);
  if (st != DGDM_OK) return st > 0 ? DGDM_OK : st;
  if (!pre) return DGDM_ERR_INVALID_ARG;
  if ((ldp & 3) || ldp < ncols || !dgdm_aligned16(pre)) return DGDM_ERR_UNSUPPORTED;
  EpiArgs e{};
  e.pre_in = pre; e.ldp = ldp; e.act = act; e.drop_p = drop_p; e.seed = dgdm_seed_arg(seed); e.amax_out = amax_g;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const char* img = static_cast<const char*>(image);
  if (ncols <= DGDM_IMG_NARROW_MAX)
    return launch_img<DGDM_IMG_WM, DGDM_IMG_WN, DGDM_IMG_NTW, EPI_ACTBWD>(s, A, lda, M, K, img, image_tiles, tile_begin, ncols, nullptr, G, ldg, 0, amax_a, e);
  return launch_img8<EPI_ACTBWD>(s, A, lda, M, K, img, image_tiles, tile_begin, ncols, nullptr, G, ldg, 0, amax_a, e);
}

extern "C" int32_t dgdm_gemm_rows_img_norm_supported(int32_t ncols, int32_t groups) {
  if (ncols <= 0 || groups <= 0 || ncols % groups) return 0;
  const int L = ncols / groups;
  if (L % 32 || (L & (L - 1))) return 0;               // whole 32-column tiles, a power of two of them per group
  // a group must lie inside one wave's columns: 128 per wave on the narrow kernel (taken up to DGDM_IMG_NARROW_MAX columns), 256 on the wide one
  return (ncols <= 128 || (ncols <= DGDM_IMG_NARROW_MAX && L <= 128) || L <= 256) ? 1 : 0;
}

extern "C" int dgdm_gemm_rows_img_norm(const float* A, int64_t lda, int32_t M, int32_t K, const void* image, int32_t image_tiles,
                                       int32_t tile_begin, int32_t ncols, const float* bias, float pre_drop_p, uint32_t pre_seed,
                                       const float* res, int64_t ldr, const int32_t* res_ptr, int32_t res_segments,
                                       const float* gamma, const float* beta, int32_t groups, float eps, float* sum, int64_t lds,
                                       float* Y, int64_t ldy, float* mean, float* rstd, int32_t act, float drop_p, uint32_t seed,
                                       const uint32_t* amax_a, uint32_t* amax_y, void* stream) {
  const int st = epi_common_checks(A, lda, M, K, image, image_tiles, tile_begin, ncols, Y, ldy, amax_a, act, drop_p);
  if (st != DGDM_OK) return st > 0 ? DGDM_OK : st;
  if (!gamma || !beta || !mean || !rstd || groups <= 0 || !(pre_drop_p >= 0.f && pre_drop_p < 1.f)) return DGDM_ERR_INVALID_ARG;
  if (res_ptr && (!res || res_segments <= 0)) return DGDM_ERR_INVALID_ARG;
  if (!dgdm_gemm_rows_img_norm_supported(ncols, groups)) return DGDM_ERR_UNSUPPORTED;
  if ((bias && !dgdm_aligned16(bias)) || !dgdm_aligned16(gamma) || !dgdm_aligned16(beta)) return DGDM_ERR_UNSUPPORTED;
  if (res && ((ldr & 3) || ldr < ncols || !dgdm_aligned16(res))) return DGDM_ERR_UNSUPPORTED;
  if (sum && ((lds & 3) || lds < ncols || !dgdm_aligned16(sum))) return DGDM_ERR_UNSUPPORTED;
  EpiArgs e{};
  e.res = res; e.ldr = ldr; e.res_ptr = res ? res_ptr : nullptr; e.res_segments = res_segments;
  e.pre_drop_p = pre_drop_p; e.pre_seed = dgdm_seed_arg(pre_seed);
  e.gamma = gamma; e.beta = beta; e.sum_out = sum; e.lds = lds; e.mean = mean; e.rstd = rstd;
  e.eps = eps; e.L = ncols / groups; e.act = act; e.drop_p = drop_p; e.seed = dgdm_seed_arg(seed); e.amax_out = amax_y;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const char* img = static_cast<const char*>(image);
  if (ncols <= DGDM_IMG_NARROW_MAX && e.L <= 128)
    return launch_img<4, 1, 4, EPI_NORM>(s, A, lda, M, K, img, image_tiles, tile_begin, ncols, bias, Y, ldy, 0, amax_a, e);
  return launch_img8<EPI_NORM>(s, A, lda, M, K, img, image_tiles, tile_begin, ncols, bias, Y, ldy, 0, amax_a, e);
}
