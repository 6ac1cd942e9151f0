// K4 backward, split-fp16 path, ONE pass for dQ, dK and dV (round 4; the default backward).
//
// attn_h_bwd.hip evaluates S, dP, exp2, the distance and the dropout word twice -- once per query tile for dQ (lane = query) and
// once per key tile for dK / dV (lane = key) -- because an MFMA product reduces over the index its operands hold in REGISTERS:
// dK / dV reduce over queries (scores as [query rows in registers][key in lanes]), dQ reduces over keys (the transposed
// arrangement).  Here the key-stationary pass of k_attn_h_bwd_dkv also produces dQ:
//   * a workgroup owns a key SUPER-block -- 4 consecutive 64-key blocks of one graph, one per wave -- and ONE head; per staged
//     query block (64 queries) a wave runs its 4 key tiles of 16 keys one after the other, each exactly as k_attn_h_bwd_dkv does
//     (S' and keep.dP - delta out of the MFMAs, P' = exp2, dropout word, dS', fp16 hi / lo splits, dV and dK products);
//   * the dS' halfs of a key tile also go through a wave-private LDS tile T[part][key][q] (8-byte writes) and come back TRANSPOSED
//     (ds_read_b64_tr_b16: 4 keys x 16 queries per lane group) as the B operand of
//         dQ^T[d][q] += K^T[d][key] dS'^T[key][q]          (reduction over the tile's 16 keys)
//     with the 32 reduction slots of v_mfma_f32_16x16x32_f16 = 16 keys x {hi, lo}:  A1 = [K_hi | K_hi], A2 = [K_lo | 0],
//     B = [dS_hi ; dS_lo]  =>  A1.B + A2.B = K_hi (dS_hi + dS_lo) + K_lo dS_hi   (lo.lo dropped, as everywhere else).
//     No vector instruction is added: the transposition is LDS traffic, the product two MFMAs per 16 x 16 queries x keys, and the
//     wave's dQ tile (64 q x 16 d over its 64 keys) accumulates in registers across the four key tiles;
//   * the four waves' dQ tiles meet in LDS between the two barriers the single-buffered loop has anyway (written before "everyone
//     is done with the block", summed in fixed order + stored while the next block's DMA is in flight) and leave as ONE 64 x 16
//     fp32 tile per (super-block, query block, head) into a scratch buffer;
//   * k_attn_dq_reduce sums the super-blocks' tiles of a query block in super-block order (fixed association: bitwise repeatable,
//     no float atomics) and applies the final scale.
// Cost of the second stage: N^2 H / 256 * 64 B of partials written and read once (0.8 GB at 4 x 10k nodes, 8 heads), against a
// whole second evaluation of every score.  A first form with one 64-key block per workgroup (3.2 GB of partials, four times the
// cross-wave stages) was slower than the two passes; with 256 keys it is 3.18 against 3.59 ms.  The host cuts
// the super-blocks into groups so that the scratch stays within a budget (ops.py); the reduction accumulates group after group.
#include "attn_h.hpp"

// diagnostic builds only (tools/build_variant_lib.sh -DDGDM_FUSED_SKIP=n): 1 = no partial store, 2 = no cross-wave stage either,
// 4 = no transposition / dQ product at all (what is left is k_attn_h_bwd_dkv at this kernel's occupancy)
#ifndef DGDM_FUSED_SKIP
#define DGDM_FUSED_SKIP 0
#endif
// tuning switches of round 4, both measured and left OFF (same-box A/B through tools/build_variant_lib.sh, kernel alone, dropout on,
// 4 x 10k nodes): plain 3.02 ms | DEFER (the dQ product of key tile kt runs after the score phase of kt + 1, out of a second
// transposition tile) 3.05-3.08 | RAWBAR (the closing barrier of an iteration leaves the partial store in flight: s_waitcnt
// vmcnt(1) + s_barrier instead of __syncthreads) 3.19-3.20 | a second staging buffer for the query blocks 3.13.
#ifndef DGDM_FUSED_DEFER
#define DGDM_FUSED_DEFER 0
#endif
#ifndef DGDM_FUSED_RAWBAR
#define DGDM_FUSED_RAWBAR 0
#endif
#ifndef DGDM_FUSED_PAIRWISE
#define DGDM_FUSED_PAIRWISE 1
#endif
// NBUF 2: the next query block is staged into a second buffer at the top of the iteration (lands under the arithmetic): measured
// 3.12 against 3.00 ms with dropout on, 2.64 against 2.62 without (same box) -- left at 1
#ifndef DGDM_FUSED_NBUF
#define DGDM_FUSED_NBUF 1
#endif

// 64-key blocks (= waves) per workgroup.  4: 256 keys, 57 KB LDS, two workgroups per CU (round 4).  8 (round 5):
// 512 keys, 105 KB LDS, ONE workgroup of 8 waves per CU -- the same 2 waves per SIMD -- with HALF the partial dQ tiles to write and
// to reduce (one 64 x 16 tile per (super-block, query block, head): 0.43 instead of 0.87 GB at 4 x 10k nodes x 8 heads) and half
// the staging traffic per key; the price is a barrier of 8 waves per query block.  Measured: see DESIGN.md section 4.
// (the macro's default, 4, lives in attn_h.hpp: the zero-block map has one row per super-block)

namespace {

constexpr int SBW = ATTN_SBW;
constexpr float NEG_BIG = -1.0e30f;
// The transposition tile [key 0..15][query 0..63] of a wave (hi part, lo part).  It is WRITTEN by key rows (lane = key j, 8-byte
// chunks of 4 queries: a 16-lane store group holds ONE chunk column of 16 rows) and READ transposed (a 32-lane read group holds 8
// rows x 4 consecutive chunks).  DGDM_FUSED_SWIZZLE 1 (shipped): unpadded 128-byte rows, chunk c of row k stored at chunk
// c ^ g(k), g = the bit permutation (k2 k1 k3 k0) of the row number: the 16 rows of a store group land in 16 distinct bank pairs
// and the 8 x 4 chunks of a read group in 32 distinct ones -- no bank conflict on either side.  0: the round-4 layout, rows padded
// to 144 bytes: stores two-way conflicted (rows k and k + 8), transposed reads two-way on 4 of 64 banks (row 7 wraps onto row 0);
// together with the four-way conflicts of the X stores below that was 72 extra LDS cycles per key tile, 12.7 % of the CU cycles of
// the kernel (SQ_LDS_BANK_CONFLICT, profiles/r04_pmc_lds.json).
#ifndef DGDM_FUSED_SWIZZLE
#define DGDM_FUSED_SWIZZLE 1
#endif
constexpr int T_LD = DGDM_FUSED_SWIZZLE ? 64 : 72;      // halfs per key row
constexpr int T_PART = 16 * T_LD;              // halfs per part (hi / lo)
constexpr int T_WAVE = 2 * T_PART;             // halfs per wave
__device__ __forceinline__ int t_swz(int k) {   // g(k) in units of halfs (4 per chunk)
  return DGDM_FUSED_SWIZZLE ? 4 * ((((k >> 1) & 3) << 2) | (((k >> 3) & 1) << 1) | (k & 1)) : 0;
}

// 4 rows x 16 halfs, transposed: lane (j, G) receives column j of rows 4G .. 4G+3 of the tile, columns c0 .. c0 + 15
__device__ __forceinline__ f16x4 tr4_tile(const _Float16* __restrict__ tile, int c0, int lane) {
  typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int j = lane & 15, G = lane >> 4;
  const int k = 4 * G + (j >> 2);
  const _Float16* p = tile + k * T_LD + ((c0 + 4 * (j & 3)) ^ t_swz(k));
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  return __builtin_bit_cast(f16x4, a);
}
// the same out of a packed row image (attn_h.hpp): rows 16 t + 4G .. +3 of `part`, column j
__device__ __forceinline__ f16x4 tr4_image(const _Float16* __restrict__ rimg_head, int part, int t, int lane) {
  typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int j = lane & 15, G = lane >> 4;
  const _Float16* p = rimg_head + t * 512 + r_off(4 * G + (j >> 2), 2 * part + ((j & 3) >> 1)) + 4 * (j & 1);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  return __builtin_bit_cast(f16x4, a);
}
__device__ __forceinline__ int x_swz(int q) { return DGDM_FUSED_SWIZZLE ? (q >> 1) & 3 : 0; }
__device__ __forceinline__ f16x8 cat4(f16x4 a, f16x4 b) { return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// Key SUPER-blocks: 4 consecutive 64-key blocks of ONE graph (the last super-block of a graph may hold fewer); super-blocks are
// numbered graph by graph.  One partial dQ tile (64 x 16 fp32 per head) exists per (super-block, query block of its graph):
// slot(sb, qb) = spair0(graph) + (sb - sb0(graph)) * nbg + qb.
constexpr int KT = 4;                 // key tiles of 16 per wave = one 64-key block per wave; 4 waves = 256 keys per workgroup
__device__ __forceinline__ bool find_sblock(const int32_t* __restrict__ ptr, int B, int sb, int* n0, int* ng, int* sbl, int* blk0,
                                            int* sb0, int64_t* spair0) {
  int base = 0, sbase = 0;
  int64_t pairs = 0;
  for (int g = 0; g < B; ++g) {
    const int a = ptr[g], b = ptr[g + 1];
    const int nb = (b - a + HB - 1) / HB, nsb = (nb + SBW - 1) / SBW;
    if (sb < sbase + nsb) { *n0 = a; *ng = b - a; *sbl = sb - sbase; *blk0 = base; *sb0 = sbase; *spair0 = pairs; return true; }
    base += nb; sbase += nsb;
    pairs += (int64_t)nsb * nb;
  }
  return false;
}
// the same lookup by packed (64-row) block: used by the reduction, whose workgroups are query blocks
__device__ __forceinline__ bool find_block_s(const int32_t* __restrict__ ptr, int B, int blk, int* n0, int* ng, int* lblk, int* sb0,
                                             int64_t* spair0) {
  int base = 0, sbase = 0;
  int64_t pairs = 0;
  for (int g = 0; g < B; ++g) {
    const int a = ptr[g], b = ptr[g + 1];
    const int nb = (b - a + HB - 1) / HB, nsb = (nb + SBW - 1) / SBW;
    if (blk < base + nb) { *n0 = a; *ng = b - a; *lblk = blk - base; *sb0 = sbase; *spair0 = pairs; return true; }
    base += nb; sbase += nsb;
    pairs += (int64_t)nsb * nb;
  }
  return false;
}

// One head per workgroup, wave w = key block 4 sb + w of the graph (64 keys = KT key tiles, processed one after the other against
// the staged query block).  LDS: stage buffer (Rq | Rg | 8 - lse2 | -delta | pos of ONE head) | the four waves' own K row images |
// T (4 waves) | X[4 waves][64 q][16 d] fp32.
template <bool DROP, int WPE = 1>
__global__ __launch_bounds__(64 * SBW) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k_attn_h_bwd_fused(
    const _Float16* __restrict__ Rq, const _Float16* __restrict__ Rk, const _Float16* __restrict__ Rv, const _Float16* __restrict__ Rg,
    const float* __restrict__ pos_b, const float* __restrict__ lse_b, const float* __restrict__ ndelta_b, int H,
    const int32_t* __restrict__ ptr, int B, float kscale, const float* __restrict__ unscale_dev, float* __restrict__ dK,
    float* __restrict__ dV, int64_t ldg, float* __restrict__ dq_part, int sb_first, int sb_count, int64_t slot_first, float drop_p,
    DgdmSeed seed_in, const uint32_t* __restrict__ skip_map, int num_blocks, unsigned* __restrict__ amax_out) {
  const uint32_t seed = seed_in.value();
  constexpr int NT = HB / 16;
  constexpr int R_BYTES = R_HEAD * 2, SC_BYTES = HB * 4, POS_BYTES = HB * 8;
  constexpr int BUF_BYTES = 2 * R_BYTES + 2 * SC_BYTES + POS_BYTES;
  constexpr int KOWN_BYTES = SBW * R_BYTES;
  constexpr int T_BYTES = (DGDM_FUSED_DEFER ? 2 : 1) * SBW * T_WAVE * 2;      // two tiles per wave: the dQ product of key tile kt runs under the score phase of kt + 1
  constexpr int X_BYTES = SBW * HB * 16 * 4;
  constexpr int STG_BYTES = DGDM_FUSED_NBUF * BUF_BYTES;
#if DGDM_FUSED_SBW > 4
  extern __shared__ __attribute__((aligned(16))) char smem[];      // 105 KB: beyond the static limit, sized by the launch (fused_lds_bytes)
#else
  __shared__ __attribute__((aligned(16))) char smem[STG_BYTES + KOWN_BYTES + T_BYTES + X_BYTES];
#endif
  const DropCfg dc(drop_p);

  int n0, ng, sbl, blk0, sb0;
  int64_t spair0;
  // workgroup -> (head, super-block), head-major so that a head's super-blocks are contiguous in the logical order (attn_h.hpp)
  const int logical = attn_xcd_logical((int)blockIdx.x, sb_count * H);
  if (logical >= sb_count * H) return;
  const int head = __builtin_amdgcn_readfirstlane(logical / sb_count), sbx = logical - head * sb_count;      // (scalar registers)
  if (sbx >= sb_count || !find_sblock(ptr, B, sb_first + sbx, &n0, &ng, &sbl, &blk0, &sb0, &spair0)) return;
  const int nbg = (ng + HB - 1) / HB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, G = lane >> 4;
  const int lblk_w = SBW * sbl + wave;                       // this wave's key block inside the graph ...
  const bool blk_ok = lblk_w < nbg;                        // ... which the graph's last super-block may not have
  const int lblk = blk_ok ? lblk_w : nbg - 1;              // (such a wave runs on the last block with every key masked: it takes part
  const int blk = blk0 + lblk;                             //  in the barriers and the cross-wave stage, contributes zeros, stores nothing)

  auto stage = [&](int qb, int b) {
    if (SBW > 4 && wave >= 4) return;       // dma_to_lds hands the 1 KiB pieces to waves 0 .. 3
    const int64_t gb = (int64_t)(blk0 + qb) * H + head;
    char* base = smem + (DGDM_FUSED_NBUF == 2 ? (b & 1) * BUF_BYTES : 0);
    dma_to_lds<R_BYTES>(Rq + gb * R_HEAD, base, tid);
    dma_to_lds<R_BYTES>(Rg + gb * R_HEAD, base + R_BYTES, tid);
    dma_to_lds<SC_BYTES>(lse_b + gb * HB, base + 2 * R_BYTES, tid);
    dma_to_lds<SC_BYTES>(ndelta_b + gb * HB, base + 2 * R_BYTES + SC_BYTES, tid);
    dma_to_lds<POS_BYTES>(pos_b + (int64_t)(blk0 + qb) * HB * 2, base + 2 * R_BYTES + 2 * SC_BYTES, tid);
  };
  const _Float16* Kown = reinterpret_cast<const _Float16*>(smem + STG_BYTES) + wave * R_HEAD;      // this wave's K row image (64 keys)
  _Float16* Tw0 = reinterpret_cast<_Float16*>(smem + STG_BYTES + KOWN_BYTES) + wave * (DGDM_FUSED_DEFER ? 2 : 1) * T_WAVE;
  float* X = reinterpret_cast<float*>(smem + STG_BYTES + KOWN_BYTES + T_BYTES);

  // the four waves' K images: each wave copies its own (4 KiB = 4 pieces of 1 KiB), then the first query block
  {
    const char* g = reinterpret_cast<const char*>(Rk + ((int64_t)blk * H + head) * R_HEAD);
    char* l = smem + STG_BYTES + wave * R_BYTES;
#pragma unroll
    for (int p = 0; p < R_BYTES / 1024; ++p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + p * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(l + p * 1024), 16, 0, 0);
  }
  // query blocks whose weights are exactly zero against all of this super-block's keys (every head of the map's group) are walked
  // over: no staging, no arithmetic, no partial tile (k_attn_dq_reduce consults the same bits)
  const uint32_t* srow = skip_map ? skip_map + attn_map_row(1, head / attn_map_group(H), H, num_blocks, sb_first + sbx) : nullptr;
  LiveWalk live;
  live.init(srow, nbg);
  int qb = __builtin_amdgcn_readfirstlane(live.next());
  if (qb < nbg) stage(qb, 0);

  f16x8 vb1[KT], vb2[KT];                 // (keep *) V of the lane's key in key tile kt (K comes out of Kown at every use)
  f32x4 dk[KT], dv[KT];
  uint32_t dlF[KT];                       // dropout: folded hash of the lane's key pair per key tile; multipliers by key parity
  uint32_t dl_me = 0, dl_mo = 0;
  float px[KT], py[KT];
  bool k_ok[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int k_in_blk = 16 * kt + j;
    const int k_local = lblk * HB + k_in_blk;
    k_ok[kt] = blk_ok && k_local < ng;
    const int64_t imgoff = ((int64_t)blk * H + head) * R_HEAD;
    load_b_pair(Rv + imgoff, k_in_blk, G, &vb1[kt], &vb2[kt]);
    if (DROP) scale_b_pair(&vb1[kt], &vb2[kt], dc.keep);
    dk[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (DROP) {
      const DropLaneK d(DropHead(seed, n0, head), k_local);
      dlF[kt] = d.F; dl_me = d.me; dl_mo = d.mo;      // 16 kt is even: the key parity (and with it me / mo) is the same for every kt
    }
    px[kt] = pos_b[((int64_t)blk * 2 + 0) * HB + k_in_blk];
    py[kt] = pos_b[((int64_t)blk * 2 + 1) * HB + k_in_blk];
  }
  const uint32_t lcq = __umul24(2u * (uint32_t)G, DROP_CQ);
  const int aoff = r_lane_off(j, G);
  const int64_t slot0 = spair0 + (int64_t)sbl * nbg - slot_first;     // this super-block's first slot inside the launch's scratch
  const _Float16 zh = (_Float16)0.0f;
  const f16x4 zero4 = {zh, zh, zh, zh};
  __syncthreads();

  for (int it = 0; qb < nbg; ++it) {
    const int qb_next = __builtin_amdgcn_readfirstlane(live.next());
    if (DGDM_FUSED_NBUF == 2 && qb_next < nbg) stage(qb_next, it + 1);      // other buffer: every wave left it at the closing barrier of the iteration before
    const char* base = smem + (DGDM_FUSED_NBUF == 2 ? (it & 1) * BUF_BYTES : 0);
    const _Float16* Qimg = reinterpret_cast<const _Float16*>(base);
    const _Float16* Gimg = reinterpret_cast<const _Float16*>(base + R_BYTES);
    const float* Ls = reinterpret_cast<const float*>(base + 2 * R_BYTES);
    const float* Ds = Ls + HB;
    const float* Ps = Ds + HB;
    const int qb0 = qb * HB;
    const bool q_tail = qb0 + HB > ng;      // only the last query block of a graph has rows to mask
    f32x4 dqp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) dqp[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // dQ^T[d][q] += K^T[d][key] dS'^T[key][q] over key tile kt: lane (q = j, G) supplies reduction slots = keys 4G .. 4G+3, hi | lo,
    // read TRANSPOSED out of the tile the wave wrote during kt's score phase.  Called one key tile LATE (after the score phase of
    // kt + 1 has been issued), so that the LDS round trip of the tile is not on the critical path.
    auto dq_product = [&](int kt) {
      if (DGDM_FUSED_SKIP & 4) return;
      const _Float16* Tr = Tw0 + (DGDM_FUSED_DEFER ? (kt & 1) : 0) * T_WAVE;
      const f16x4 khi = tr4_image(Kown, 0, kt, lane), klo = tr4_image(Kown, 1, kt, lane);
      const f16x8 ka1 = cat4(khi, khi), ka2 = cat4(klo, zero4);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f16x8 b = cat4(tr4_tile(Tr, 16 * t, lane), tr4_tile(Tr + T_PART, 16 * t, lane));
        dqp[t] = mfma_h(ka1, b, dqp[t]);
        dqp[t] = mfma_h(ka2, b, dqp[t]);
      }
    };
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      _Float16* Tw = Tw0 + (DGDM_FUSED_DEFER ? (kt & 1) : 0) * T_WAVE;
      // lane (key = 16 kt + j, G), reg r <-> query 16t + 4G + r
      f32x4 dist[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) dist[t] = dist4(&Ps[16 * t + 4 * G], &Ps[HB + 16 * t + 4 * G], px[kt], py[kt]);
      if (q_tail) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (qb0 + 16 * t + 4 * G + r >= ng) dist[t][r] = -NEG_BIG;
      }
      // keys past the graph end (its last block; every key of a wave without a block): zero K / V rows.  Their dS' would feed dQ,
      // and a row whose real scores are all far below zero has P' = exp2(0 - lse2 + 8) = inf there (inf x K = 0 is NaN): P' = 0
      if (!__builtin_amdgcn_readfirstlane((int)__all(k_ok[kt]))) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) dist[t][r] = k_ok[kt] ? dist[t][r] : -NEG_BIG;
      }
      f16x8 kb1, kb2;
      load_b_pair(Kown, 16 * kt + j, G, &kb1, &kb2);
      DropLaneK dl;
      dl.F = dlF[kt]; dl.me = dl_me; dl.mo = dl_mo;

      if (DGDM_FUSED_DEFER && kt > 0) dq_product(kt - 1);
#if DGDM_FUSED_PAIRWISE
      // scores of ONE pair of query tiles at a time (32 queries = the reduction depth of one dV / dK MFMA), split and consumed
      // before the next pair is formed: half the live score registers of the all-tiles-first order
#pragma unroll
      for (int tp = 0; tp < NT / 2; ++tp) {
        f32x4 p[2], ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * tp + u;
          const f16x8 qa = *reinterpret_cast<const f16x8*>(Qimg + t * 512 + aoff);
          const f16x8 ga = *reinterpret_cast<const f16x8*>(Gimg + t * 512 + aoff);
          const f32x4 lq = *reinterpret_cast<const f32x4*>(&Ls[16 * t + 4 * G]);
          const f32x4 nd = *reinterpret_cast<const f32x4*>(&Ds[16 * t + 4 * G]);
          f32x4 s = mfma_h(qa, kb1, sub4(lq, dist[t]));
          s = mfma_h(qa, kb2, s);
          f32x4 dpv = mfma_h(ga, vb1[kt], nd);
          dpv = mfma_h(ga, vb2[kt], dpv);
#pragma unroll
          for (int r = 0; r < 4; ++r) p[u][r] = __builtin_amdgcn_exp2f(s[r]);
          if (DROP) {
            uint32_t e[4];
            drop_words_k(dl, (uint32_t)(qb0 + 16 * t), lcq, e);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool kept = (int)e[r] >= dc.ts32;
              ds[u][r] = p[u][r] * (kept ? dpv[r] : nd[r]);
              p[u][r] = kept ? p[u][r] : 0.f;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[u][r] = p[u][r] * dpv[r];
          }
        }
        f16x8 ph, pl, sh, sl;
        split8(p[0], p[1], &ph, &pl);
        split8(ds[0], ds[1], &sh, &sl);
#else
      f32x4 p[NT], ds[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f16x8 qa = *reinterpret_cast<const f16x8*>(Qimg + t * 512 + aoff);
        const f16x8 ga = *reinterpret_cast<const f16x8*>(Gimg + t * 512 + aoff);
        const f32x4 lq = *reinterpret_cast<const f32x4*>(&Ls[16 * t + 4 * G]);
        const f32x4 nd = *reinterpret_cast<const f32x4*>(&Ds[16 * t + 4 * G]);
        f32x4 s = mfma_h(qa, kb1, sub4(lq, dist[t]));   // S'[q][key] - dist - lse2[q] + 8   (lq = 8 - lse2)
        s = mfma_h(qa, kb2, s);
        f32x4 dpv = mfma_h(ga, vb1[kt], nd);            // keep * dP[q][key] - delta[q]
        dpv = mfma_h(ga, vb2[kt], dpv);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[t][r] = __builtin_amdgcn_exp2f(s[r]);
        if (DROP) {
          uint32_t e[4];
          drop_words_k(dl, (uint32_t)(qb0 + 16 * t), lcq, e);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool kept = (int)e[r] >= dc.ts32;
            ds[t][r] = p[t][r] * (kept ? dpv[r] : nd[r]);
            p[t][r] = kept ? p[t][r] : 0.f;           // dV sees the dropped weights (1/(1-p) multiplies the finished rows)
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[t][r] = p[t][r] * dpv[r];
        }
      }
#pragma unroll
      for (int tp = 0; tp < NT / 2; ++tp) {
        f16x8 ph, pl, sh, sl;
        split8(p[2 * tp], p[2 * tp + 1], &ph, &pl);
        split8(ds[2 * tp], ds[2 * tp + 1], &sh, &sl);
#endif
        // dS' of the lane's key for queries 32 tp + 4G .. +3 and 32 tp + 16 + 4G .. +3 -> the transposition tile [part][key][q]
        if (!(DGDM_FUSED_SKIP & 4)) {
          typedef _Float16 h4 __attribute__((ext_vector_type(4)));
          _Float16* tr = Tw + j * T_LD;
          const int c0 = (32 * tp + 4 * G) ^ t_swz(j), c1 = (32 * tp + 4 * G + 16) ^ t_swz(j);
          *reinterpret_cast<h4*>(tr + c0) = h4{sh[0], sh[1], sh[2], sh[3]};
          *reinterpret_cast<h4*>(tr + c1) = h4{sh[4], sh[5], sh[6], sh[7]};
          *reinterpret_cast<h4*>(tr + T_PART + c0) = h4{sl[0], sl[1], sl[2], sl[3]};
          *reinterpret_cast<h4*>(tr + T_PART + c1) = h4{sl[4], sl[5], sl[6], sl[7]};
        }
        const f16x8 ghi = load_tr_pair(Gimg, 0, 2 * tp, lane), qhi = load_tr_pair(Qimg, 0, 2 * tp, lane);
        dv[kt] = mfma_h(ghi, ph, dv[kt]);                                       // dV^T[d=j][key] += dO^T[d][q] P[q][key]
        dv[kt] = mfma_h(load_tr_pair(Gimg, 1, 2 * tp, lane), ph, dv[kt]);
        dv[kt] = mfma_h(ghi, pl, dv[kt]);
        dk[kt] = mfma_h(qhi, sh, dk[kt]);                                       // dK^T[d=j][key] += Q'^T[d][q] dS[q][key]
        dk[kt] = mfma_h(load_tr_pair(Qimg, 1, 2 * tp, lane), sh, dk[kt]);
        dk[kt] = mfma_h(qhi, sl, dk[kt]);
      }
      if (!DGDM_FUSED_DEFER) dq_product(kt);
    }
    if (DGDM_FUSED_DEFER) dq_product(KT - 1);
    if (!(DGDM_FUSED_SKIP & 6)) {
      float* Xw = X + (wave * HB) * 16;
#pragma unroll
      for (int t = 0; t < NT; ++t)      // X[wave][q = 16 t + j][d = 4G .. 4G+3]; the row's four 16-byte pieces rotated by (q >> 1) & 3:
        *reinterpret_cast<f32x4*>(Xw + (16 * t + j) * 16 + 4 * (G ^ x_swz(j))) = dqp[t];      // 8 rows of a store group -> 8 distinct bank quads
    } else if (dqp[0][0] == 123.456f) {
      X[0] = dqp[1][0] + dqp[2][1] + dqp[3][2];      // (diagnostic build: keep the product alive)
    }
    __syncthreads();      // everyone is done with the staged block; every wave's dQ tile (its 64 keys) is in X
    if (DGDM_FUSED_NBUF == 1 && qb_next < nbg) stage(qb_next, 0);
    if (!(DGDM_FUSED_SKIP & 6)) {   // ... and while the next block's DMA is in flight: the four waves' tiles summed (fixed order) and stored
      const int q = (tid & 255) >> 2, d4 = tid & 3;
      const float* xs = X + q * 16 + 4 * (d4 ^ x_swz(q));
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(xs), a1 = *reinterpret_cast<const f32x4*>(xs + HB * 16);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(xs + 2 * HB * 16), a3 = *reinterpret_cast<const f32x4*>(xs + 3 * HB * 16);
      f32x4 sum = (a0 + a1) + (a2 + a3);
      if (SBW == 8) {      // the second four waves' tiles, same association
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(xs + 4 * HB * 16), b1 = *reinterpret_cast<const f32x4*>(xs + 5 * HB * 16);
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(xs + 6 * HB * 16), b3 = *reinterpret_cast<const f32x4*>(xs + 7 * HB * 16);
        sum += (b0 + b1) + (b2 + b3);
      }
      float* o = dq_part + (((slot0 + qb) * H + head) * HB + q) * 16 + 4 * d4;        // [slot][head][64 q][16 d]
      if (!(DGDM_FUSED_SKIP & 1) && tid < 256) *reinterpret_cast<f32x4*>(o) = sum;
      else if (sum[0] == 123.456f) dq_part[0] = sum[1];
    }
    // closing barrier: the next block's DMA has landed and every wave has read X.  The partial-tile store was issued AFTER this wave's
    // DMA pieces, so waiting for all but the youngest vector-memory operation retires the DMA and leaves the store in flight (a full
    // __syncthreads waits for its acknowledgement as well: ~1-2 us per iteration)
#if DGDM_FUSED_RAWBAR
    if (!(DGDM_FUSED_SKIP & 7)) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#else
    __syncthreads();
#endif
    qb = qb_next;
  }

  unsigned am = 0;      // max |dK|, |dV| of the workgroup's rows -> the operand maximum of the QKV projection's input gradient
  if (blk_ok) {
    const float un = unscale_dev[1] * exp2f(-DGDM_ATTN_P_SHIFT);
    const float unk = kscale * un, unv = DROP ? dc.keep * un : un;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      if (k_ok[kt]) {
        const int64_t off = (int64_t)(n0 + lblk * HB + 16 * kt + j) * ldg + head * 16 + 4 * G;
        const f32x4 a = dk[kt] * unk, b = dv[kt] * unv;
        const float4 a4 = make_float4(a[0], a[1], a[2], a[3]), b4 = make_float4(b[0], b[1], b[2], b[3]);
        *reinterpret_cast<float4*>(dK + off) = a4;
        *reinterpret_cast<float4*>(dV + off) = b4;
        am = dgdm_amax4(dgdm_amax4(am, a4), b4);
      }
    }
  }
  if (amax_out) dgdm_amax_commit(am, amax_out);      // kernel argument: every thread of the workgroup is here
}

// dQ[q block][head] (+)= scale * sum over the launch's key super-blocks of that graph, in order (fixed order: bitwise repeatable).
// grid (all query blocks, H); a thread owns one float4 of the 64 x 16 tile.  A query block whose graph has super-blocks in an
// EARLIER launch (sb0 < sb_first) adds to what that launch left in dQ; one whose graph has none in this launch is left alone.
__global__ __launch_bounds__(256) void k_attn_dq_reduce(const float* __restrict__ dq_part, const int32_t* __restrict__ ptr, int B, int H,
                                                        int sb_first, int sb_count, int64_t slot_first, float scale,
                                                        const float* __restrict__ unscale_dev, float* __restrict__ dQ, int64_t ldg,
                                                        const uint32_t* __restrict__ skip_map, unsigned* __restrict__ amax_out) {
  int n0, ng, lblk, sb0;
  int64_t spair0;
  if (!find_block_s(ptr, B, blockIdx.x, &n0, &ng, &lblk, &sb0, &spair0)) return;
  const int nbg = (ng + HB - 1) / HB, nsb = (nbg + SBW - 1) / SBW;
  const int h = blockIdx.y, tid = threadIdx.x, q = tid >> 2, d4 = tid & 3;
  const int lo = max(sb0, sb_first), hi = min(sb0 + nsb, sb_first + sb_count);      // this graph's super-blocks inside the launch
  if (hi <= lo) return;
  f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  // slot(sb, this query block) = spair0 + (sb - sb0) * nbg + lblk - slot_first: consecutive super-blocks are nbg slots apart
  const float* base = dq_part + (((spair0 + lblk - slot_first) * H + h) * HB + q) * 16 + 4 * d4;
  const int64_t stride = (int64_t)nbg * H * HB * 16;
  int sb = lo;
  if (!skip_map) {
    for (; sb + 3 < hi; sb += 4) {        // four loads in flight; fixed association
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += *reinterpret_cast<const f32x4*>(base + (int64_t)(sb + u - sb0) * stride);
    }
    for (; sb < hi; ++sb) acc[0] += *reinterpret_cast<const f32x4*>(base + (int64_t)(sb - sb0) * stride);
  } else {
    // a super-block that walked over this query block (all SBW bits of its key blocks set in the block's row of the zero-block map)
    // wrote no tile; the others are added in the same order and association as above (a tile of zeros would change no bit)
    const uint32_t* frow = skip_map + attn_map_row(0, h / attn_map_group(H), H, gridDim.x, blockIdx.x);
    auto live = [&](int s) {
      const int k0 = SBW * (s - sb0);
      const uint32_t all = (1u << SBW) - 1u;
      return ((frow[k0 >> 5] >> (k0 & 31)) & all) != all;
    };
    for (; sb + 3 < hi; sb += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (live(sb + u)) acc[u] += *reinterpret_cast<const f32x4*>(base + (int64_t)(sb + u - sb0) * stride);
    }
    for (; sb < hi; ++sb)
      if (live(sb)) acc[0] += *reinterpret_cast<const f32x4*>(base + (int64_t)(sb - sb0) * stride);
  }
  const f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  const int q_local = lblk * HB + q;
  unsigned am = 0;
  if (q_local < ng) {
    const float un = scale * unscale_dev[1] * exp2f(-DGDM_ATTN_P_SHIFT);
    float* o = dQ + (int64_t)(n0 + q_local) * ldg + h * 16 + 4 * d4;
    f32x4 r = sum * un;
    if (sb0 < sb_first) {          // an earlier launch covered the graph's first super-blocks
      const float4 old = *reinterpret_cast<const float4*>(o);
      r += f32x4{old.x, old.y, old.z, old.w};
    }
    const float4 r4 = make_float4(r[0], r[1], r[2], r[3]);
    *reinterpret_cast<float4*>(o) = r4;
    am = dgdm_amax4(am, r4);
  }
  // max |dQ| (with several launches per batch: of every running sum, an upper bound of the final one -- all the consumer needs)
  if (amax_out) dgdm_amax_commit(am, amax_out);
}

}  // namespace

// Host mirror of find_sblock over the HOST copy of the graph offsets: total super-blocks; first slot and slot count of a range.
static bool sblock_slots_host(const int32_t* ptr_host, int32_t B, int64_t sb_first, int64_t sb_count, int64_t* first, int64_t* count,
                              int64_t* total_sb) {
  int64_t sbase = 0, pairs = 0, f = -1, l = -1;
  const int64_t sb_last = sb_first + sb_count - 1;
  for (int g = 0; g < B; ++g) {
    const int64_t nb = ((int64_t)ptr_host[g + 1] - ptr_host[g] + HB - 1) / HB;
    if (nb < 0) return false;
    const int64_t nsb = (nb + SBW - 1) / SBW;
    if (f < 0 && sb_first < sbase + nsb) f = pairs + (sb_first - sbase) * nb;
    if (l < 0 && sb_last < sbase + nsb) l = pairs + (sb_last - sbase) * nb + nb;     // one past the last slot of the last super-block
    sbase += nsb;
    pairs += nsb * nb;
  }
  *total_sb = sbase;
  if (sb_count <= 0) { *first = 0; *count = 0; return true; }
  if (f < 0 || l < 0) return false;
  *first = f; *count = l - f;
  return true;
}

// number of key super-blocks (4 consecutive 64-key blocks of one graph) of a batch: the unit the one-pass backward is launched over
extern "C" int32_t dgdm_spatial_attn_h_bwd_fused_superblocks(const int32_t* ptr_host, int32_t B) {
  if (!ptr_host || B <= 0) return 0;
  int64_t f, c, total;
  if (!sblock_slots_host(ptr_host, B, 0, 0, &f, &c, &total)) return 0;
  return total > 0x7fffffff ? 0 : (int32_t)total;
}

// bytes of partial-tile scratch the super-blocks [sb_first, sb_first + sb_count) need (one 64 x 16 fp32 tile per (super-block, query
// block of its graph, head)); 0 for an empty / out-of-range request.  ptr_host: the B + 1 graph offsets ON THE HOST.
extern "C" size_t dgdm_spatial_attn_h_bwd_fused_workspace_bytes(const int32_t* ptr_host, int32_t B, int32_t H, int32_t sb_first,
                                                                int32_t sb_count) {
  if (!ptr_host || B <= 0 || H <= 0 || sb_first < 0 || sb_count <= 0) return 0;
  int64_t first, count, total;
  if (!sblock_slots_host(ptr_host, B, sb_first, sb_count, &first, &count, &total)) return 0;
  return (size_t)count * (size_t)H * HB * 16 * sizeof(float);
}

// skip_map (nullable): the zero-block map of the forward (dgdm_attn_skip_map_build); the SAME pointer must go to the reduction.
// amax_out (nullable): a zeroed operand-maximum slot group; this call adds max |dK|, |dV|, the reduction max |dQ|
extern "C" int dgdm_spatial_attn_h_bwd_fused_sparse(const void* Rq, const void* Rk, const void* Rv, const void* Rg, const float* pos_b,
                                                    const float* lse_adj_b, const float* ndelta_b, const int32_t* ptr, const int32_t* ptr_host,
                                                    int32_t B, int32_t num_blocks, int32_t H, float drop_p, uint32_t seed,
                                                    const float* grad_scale2, float* dK, float* dV, int64_t ldg,
                                                    int32_t sb_first, int32_t sb_count, void* workspace, size_t workspace_bytes,
                                                    const uint32_t* skip_map, uint32_t* amax_out, void* stream_) {
  DGDM_REQUIRE(B >= 0 && H > 0 && num_blocks >= 0 && drop_p >= 0.f && drop_p < 1.f && sb_first >= 0 && sb_count >= 0);
  if (num_blocks == 0 || B == 0 || sb_count == 0) return DGDM_OK;
  DGDM_REQUIRE(Rq && Rk && Rv && Rg && pos_b && lse_adj_b && ndelta_b && ptr && ptr_host && dK && dV && grad_scale2 && workspace);
  if ((ldg & 3) || ldg < H * 16 || !dgdm_aligned16(dK) || !dgdm_aligned16(dV) || !dgdm_aligned16(workspace) ||
      (skip_map && !dgdm_aligned16(skip_map)))
    return DGDM_ERR_UNSUPPORTED;
  int64_t slot_first, slots, total_sb;
  if (!sblock_slots_host(ptr_host, B, sb_first, sb_count, &slot_first, &slots, &total_sb)) return DGDM_ERR_INVALID_ARG;
  DGDM_REQUIRE(sb_first + (int64_t)sb_count <= total_sb);
  if (workspace_bytes < (size_t)slots * (size_t)H * HB * 16 * sizeof(float)) return DGDM_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const float kscale = 0.6931471805599453f;  // Q' carries scale*log2(e): dK = sum dS Q' / log2(e)
  auto h16 = [](const void* p) { return static_cast<const _Float16*>(p); };
  float* part = static_cast<float*>(workspace);
  // dynamic LDS of the 8-wave form (the 4-wave form declares its 57 KB statically): stage | K images | T | X
  constexpr int LDS_DYN = SBW > 4 ? DGDM_FUSED_NBUF * (2 * R_HEAD * 2 + 2 * HB * 4 + HB * 8) + SBW * R_HEAD * 2 +
                                        (DGDM_FUSED_DEFER ? 2 : 1) * SBW * T_WAVE * 2 + SBW * HB * 16 * 4 : 0;
  if (LDS_DYN > 0) {
    static int attr = 1;
    if (attr == 1)
      attr = (hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_h_bwd_fused<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DYN) == hipSuccess &&
              hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_h_bwd_fused<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DYN) == hipSuccess)
                 ? DGDM_OK : DGDM_ERR_LAUNCH;
    if (attr != DGDM_OK) return attr;
  }
  if (drop_p > 0.f)
    hipLaunchKernelGGL((k_attn_h_bwd_fused<true, 2>), dim3(attn_xcd_grid(sb_count * H)), dim3(64 * SBW), LDS_DYN, s, h16(Rq), h16(Rk), h16(Rv), h16(Rg), pos_b,
                       lse_adj_b, ndelta_b, H, ptr, B, kscale, grad_scale2, dK, dV, ldg, part, sb_first, sb_count, slot_first, drop_p,
                       dgdm_seed_arg(seed), skip_map, num_blocks, amax_out);
  else
    hipLaunchKernelGGL((k_attn_h_bwd_fused<false, 2>), dim3(attn_xcd_grid(sb_count * H)), dim3(64 * SBW), LDS_DYN, s, h16(Rq), h16(Rk), h16(Rv), h16(Rg), pos_b,
                       lse_adj_b, ndelta_b, H, ptr, B, kscale, grad_scale2, dK, dV, ldg, part, sb_first, sb_count, slot_first, 0.f,
                       dgdm_seed_arg(0u), skip_map, num_blocks, amax_out);
  return dgdm_launch_status();
}

// second stage of the call above (same sb_first / sb_count / workspace): dQ rows of every query block whose graph has super-blocks in
// the range (+)= scale * sum of their partial tiles, in super-block order
extern "C" int dgdm_spatial_attn_h_bwd_fused_reduce_sparse(const int32_t* ptr, const int32_t* ptr_host, int32_t B, int32_t num_blocks,
                                                           int32_t H, float scale, const float* grad_scale2, float* dQ, int64_t ldg,
                                                           int32_t sb_first, int32_t sb_count, const void* workspace,
                                                           size_t workspace_bytes, const uint32_t* skip_map, uint32_t* amax_out,
                                                           void* stream_) {
  DGDM_REQUIRE(B >= 0 && H > 0 && num_blocks >= 0 && sb_first >= 0 && sb_count >= 0);
  if (num_blocks == 0 || B == 0 || sb_count == 0) return DGDM_OK;
  DGDM_REQUIRE(ptr && ptr_host && dQ && grad_scale2 && workspace);
  if ((ldg & 3) || ldg < H * 16 || !dgdm_aligned16(dQ) || !dgdm_aligned16(workspace)) return DGDM_ERR_UNSUPPORTED;
  int64_t slot_first, slots, total_sb;
  if (!sblock_slots_host(ptr_host, B, sb_first, sb_count, &slot_first, &slots, &total_sb)) return DGDM_ERR_INVALID_ARG;
  DGDM_REQUIRE(sb_first + (int64_t)sb_count <= total_sb);
  if (workspace_bytes < (size_t)slots * (size_t)H * HB * 16 * sizeof(float)) return DGDM_ERR_WORKSPACE;
  hipLaunchKernelGGL(k_attn_dq_reduce, dim3(num_blocks, H), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     static_cast<const float*>(workspace), ptr, B, H, sb_first, sb_count, slot_first, scale, grad_scale2, dQ, ldg, skip_map, amax_out);
  return dgdm_launch_status();
}

extern "C" int dgdm_spatial_attn_h_bwd_fused(const void* Rq, const void* Rk, const void* Rv, const void* Rg, const float* pos_b,
                                             const float* lse_adj_b, const float* ndelta_b, const int32_t* ptr, const int32_t* ptr_host,
                                             int32_t B, int32_t num_blocks, int32_t H, float drop_p, uint32_t seed,
                                             const float* grad_scale2, float* dK, float* dV, int64_t ldg,
                                             int32_t sb_first, int32_t sb_count, void* workspace, size_t workspace_bytes, void* stream_) {
  return dgdm_spatial_attn_h_bwd_fused_sparse(Rq, Rk, Rv, Rg, pos_b, lse_adj_b, ndelta_b, ptr, ptr_host, B, num_blocks, H, drop_p, seed,
                                              grad_scale2, dK, dV, ldg, sb_first, sb_count, workspace, workspace_bytes, nullptr, nullptr, stream_);
}

extern "C" int dgdm_spatial_attn_h_bwd_fused_reduce(const int32_t* ptr, const int32_t* ptr_host, int32_t B, int32_t num_blocks, int32_t H,
                                                    float scale, const float* grad_scale2, float* dQ, int64_t ldg, int32_t sb_first,
                                                    int32_t sb_count, const void* workspace, size_t workspace_bytes, void* stream_) {
  return dgdm_spatial_attn_h_bwd_fused_reduce_sparse(ptr, ptr_host, B, num_blocks, H, scale, grad_scale2, dQ, ldg, sb_first, sb_count,
                                                     workspace, workspace_bytes, nullptr, nullptr, stream_);
}
