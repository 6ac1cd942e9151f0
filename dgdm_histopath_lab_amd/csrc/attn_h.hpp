// Split-fp16 ("hi+lo") formulation of the spatial-attention products.
//
// Measured on MI355X (tools/ubench/mfma_valu_overlap.hip): the fp32 MFMA
// (v_mfma_f32_16x16x4_f32) does not overlap with fp32 VALU work -- it occupies the SIMD for its full
// 32 cycles, so an fp32-MFMA attention kernel costs MFMA + softmax VALU, not max().  The 16-bit
// MFMAs run on the separate matrix pipe and cost the wave only their ~8-cycle issue slot.
// To keep fp32-level accuracy every fp32 operand x is carried as two halfs, hi = fp16(x),
// lo = fp16(x - hi) (x = hi + lo up to 2^-22 relative), packed along the reduction index of
// v_mfma_f32_16x16x32_f16 (K = 32 = 16 "hi" slots + 16 "lo" slots of a 16-wide head):
//     A = [a_hi | a_lo],  B1 = [b_hi | b_hi],  B2 = [b_lo | b_lo]
//     A.B1 + A.B2 = a_hi b_hi + a_lo b_hi + a_hi b_lo + a_lo b_lo  = a.b   (2 MFMAs, fp32 accumulate)
// Operand maps (wave64, 16x16x32): lane l holds A[i = l&15][k = 8*(l>>4) + e], B[k = 8*(l>>4) + e][j = l&15],
// e = 0..7; C/D: lane l, reg r -> D[row = 4*(l>>4) + r][col = l&15]  (same C/D map as 16x16x4).
// A product whose accumulator feeds the next product as B operand pairs two 16-row tiles:
// reduction slot k = 8G + e  <->  row 16*t0 + 4G + e (e < 4), row 16*(t0+1) + 4G + e - 4 (e >= 4).
//
// Packed operand images (written once per launch by k_attn_pack, graph-block aligned: block b of
// graph g holds its rows 64*b .. 64*b+63, rows past the graph end are zero), all in halfs:
//   row image   R[blk][H][4 tiles][4 chunks][16][8]   per row 16 hi | 16 lo halfs, stored CHUNK-major inside a 16-row tile:
//                                          chunk 0 / 1 = hi[0..7] / hi[8..15], chunk 2 / 3 = lo[0..7] / lo[8..15]; the 16-byte
//                                          chunk of row j sits at position j (even chunks) or j ^ 12 (odd chunks) of its 256-byte
//                                          chunk block (r_off).  With this order both reads of the kernels are free of LDS bank
//                                          conflicts: the row read (ds_read_b128, lane (j, G) takes chunk G of row j -- A operand
//                                          with the row on the MFMA row) sees 16 distinct 16-byte slots in every lane group, and the
//                                          transposed read (ds_read_b64_tr_b16, load_tr_pair) finds the two chunks of a 32-lane half
//                                          on opposite halves of the banks.  Plain 64-byte rows made the first 2-way and the second
//                                          2-way conflicted (cdna_hip_programming.md T10): twice the LDS cycles for every fragment.
// A (block, head group) is one straight, contiguous byte range, so the attention kernels stage it with direct-to-LDS DMA
// (global_load_lds_dwordx4: no VGPRs, no VALU).  Operands whose reduction index is the ROW (V^T in the forward, K^T / Q'^T / dO^T
// in the backward) are read out of the same image with transposed LDS reads (load_tr_pair): there is no transposed image.
#pragma once
#include "attn_common.hpp"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int HB = 64;                 // rows per packed block
constexpr int R_HEAD = HB * 32;        // halfs per (block, head) in a row image      (4096 B)

__device__ __forceinline__ f32x4 mfma_h(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// offset (halfs) of chunk c (0..3) of row `row` (0..63) inside the row image of one (block, head)
__device__ __forceinline__ int r_off(int row, int c) { return (row >> 4) * 512 + c * 128 + 8 * ((row & 15) ^ (12 * (c & 1))); }
// the row read of lane (j, G) for tile t: chunk G of row 16 t + j  (add t * 512)
__device__ __forceinline__ int r_lane_off(int j, int G) { return G * 128 + 8 * (j ^ (12 * (G & 1))); }

// B-operand registers of row `row` of a (block, head) row image: lane group G takes hi[8*(G&1) .. +7] for B1 and lo[8*(G&1) .. +7] for B2.
__device__ __forceinline__ void load_b_pair(const _Float16* __restrict__ img_head, int row, int G, f16x8* b1, f16x8* b2) {
  *b1 = *reinterpret_cast<const f16x8*>(img_head + r_off(row, G & 1));
  *b2 = *reinterpret_cast<const f16x8*>(img_head + r_off(row, 2 + (G & 1)));
}

// 8 fp32 values (two accumulator quads) -> 8 halfs (B operand of the following product)
__device__ __forceinline__ f16x8 pack8(const f32x4 a, const f32x4 b) {
  f16x8 r;
  r[0] = (_Float16)a[0]; r[1] = (_Float16)a[1]; r[2] = (_Float16)a[2]; r[3] = (_Float16)a[3];
  r[4] = (_Float16)b[0]; r[5] = (_Float16)b[1]; r[6] = (_Float16)b[2]; r[7] = (_Float16)b[3];
  return r;
}

// 8 fp32 values -> fp16 hi+lo pairs (hi = fp16(x) by v_cvt_pk_f16_f32, lo = fp16(x - hi) by one v_fma_mixlo/hi_f16 per value:
// the mixed-precision FMA reads hi as fp16 and x as fp32 and rounds once -- 1.5 instructions per value; hipcc's own lowering of
// the same expression converts hi back to fp32 first, 2.5).  Probabilities and dS are carried like every other operand of the
// attention products: ~21 significand bits, not fp16's 11 (profiles/r03_sharp_rows_*: single-fp16 P / dS broke the 1e-3
// contract on trained-like attention rows).
__device__ __forceinline__ uint32_t pk_hi_f16(float x0, float x1) {
  typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
  const f16x2_t h = {(_Float16)x0, (_Float16)x1};
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ uint32_t pk_lo_f16(uint32_t hi_pk, float x0, float x1) {
  uint32_t r;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(r)
      : "v"(hi_pk), "v"(x0), "v"(x1));
  return r;
}
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, f16x8* hi, f16x8* lo) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t h, l;
  h[0] = pk_hi_f16(a[0], a[1]); h[1] = pk_hi_f16(a[2], a[3]); h[2] = pk_hi_f16(b[0], b[1]); h[3] = pk_hi_f16(b[2], b[3]);
  l[0] = pk_lo_f16(h[0], a[0], a[1]); l[1] = pk_lo_f16(h[1], a[2], a[3]); l[2] = pk_lo_f16(h[2], b[0], b[1]); l[3] = pk_lo_f16(h[3], b[2], b[3]);
  *hi = __builtin_bit_cast(f16x8, h);
  *lo = __builtin_bit_cast(f16x8, l);
}
// B-operand register pair (hi parts, lo parts of the same 8 values) times a constant, re-split (once per workgroup)
__device__ __forceinline__ void scale_b_pair(f16x8* b1, f16x8* b2, float c) {
  f32x4 a, b;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = ((float)(*b1)[e] + (float)(*b2)[e]) * c;
    b[e] = ((float)(*b1)[4 + e] + (float)(*b2)[4 + e]) * c;
  }
  split8(a, b, b1, b2);
}
// Packed fp32 arithmetic (v_pk_add/mul/fma_f32: two values per instruction) for the per-pair distance and the C inputs of the score
// products; written on 2-vectors because hipcc scalarises the same expressions on 4-vectors whose elements come out of v_sqrt.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// |p - k_r| for the four staged points k_r = (xs[r], ys[r]) (positions pre-scaled by log2(e)/tau in dgdm_attn_pack)
__device__ __forceinline__ f32x4 dist4(const float* __restrict__ xs, const float* __restrict__ ys, float px, float py) {
  const f32x4 kx = *reinterpret_cast<const f32x4*>(xs), ky = *reinterpret_cast<const f32x4*>(ys);
  const f32x2 px2 = {px, px}, py2 = {py, py};
  const f32x2 dxa = kx.xy - px2, dxb = kx.zw - px2, dya = ky.xy - py2, dyb = ky.zw - py2;
  const f32x2 da = __builtin_elementwise_fma(dya, dya, dxa * dxa), db = __builtin_elementwise_fma(dyb, dyb, dxb * dxb);
  return f32x4{__builtin_amdgcn_sqrtf(da.x), __builtin_amdgcn_sqrtf(da.y), __builtin_amdgcn_sqrtf(db.x), __builtin_amdgcn_sqrtf(db.y)};
}
// a - b on two values per instruction.  Inline asm: hipcc splits a 2-vector subtraction whose operand comes out of v_sqrt into two
// v_sub_f32 whatever way it is written (fsub, fadd of fneg).
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x4 sub4(const f32x4 a, const f32x4 b) {
  const f32x2 lo = pk_sub(a.xy, b.xy), hi = pk_sub(a.zw, b.zw);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 sub4(float a, const f32x4 b) {
  const f32x2 a2 = {a, a};
  const f32x2 lo = pk_sub(a2, b.xy), hi = pk_sub(a2, b.zw);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}

// log2 of the factor the backward kernels carry P with: P' = 2^P_SHIFT * P keeps the weights of a near-uniform row over 10^4..10^5
// keys (p ~ 1e-4..1e-5, at the bottom of fp16's normal range, where a lo part has nothing left) well inside it.  Folded into the
// log-sum-exp the kernels subtract, taken out again with the final scale: no instruction.
#define DGDM_ATTN_P_SHIFT 8.0f

// A operand for the tile pair (t0, t0+1) of a [d][row] matrix, lane (d = j, G) holding rows 16*t0 + 4G .. +3 and 16*(t0+1) + 4G .. +3,
// read TRANSPOSED out of the ROW image in LDS (chunk-major tiles, r_off; `part` 0 = hi, 1 = lo) with gfx950's
// ds_read_b64_tr_b16: per group of 16 lanes a block of 4 rows x 16 halfs, lane 4q + p of the group supplies the address of row q,
// halfs 4p .. 4p+3, lane i receives column i of the 4 rows.  A kernel that reads its transposed operands this way needs no
// transposed image staged at all: half the LDS-DMA pieces per block and half the LDS footprint.  Conflict-free on the chunk-major
// image (a 32-lane half reads rows 8n .. 8n+7 of two adjacent chunks: the odd chunk's rows sit 8 positions away).  EXEC must be
// all ones (no divergence around the call).
__device__ __forceinline__ f16x8 load_tr_pair(const _Float16* __restrict__ rimg_head, int part, int t0, int lane) {
  typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));
  const int j = lane & 15, G = lane >> 4;
  // lane 4q + p of a 16-lane group supplies the address of row 4G + q, halfs 4p .. 4p+3 of the 16-half part: chunk 2 part + (p >> 1)
  const _Float16* p = rimg_head + t0 * 512 + r_off(4 * G + (j >> 2), 2 * part + ((j & 3) >> 1)) + 4 * (j & 1);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 512));
  const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(f16x8, r);
}

// Workgroup order of the attention kernels.  The hardware deals consecutive workgroups to the 8 XCDs in turn (workgroup L runs on XCD
// L % 8), and the workgroups that stream the SAME operand tiles -- the query blocks of one (graph, head group) in the forward (they
// stream that group's K / V), the key super-blocks of one (graph, head) in the backward (they stream its Q' / dO) -- should meet in ONE
// L2 instead of fetching the stream into all eight.  The kernels number their work stream-major (`logical`) and take
//     logical(L) = (L % 8) * ceil(total / 8) + L / 8 :
// XCD r walks the contiguous eighth [r chunk, (r + 1) chunk) of the stream-major order, front to back, in step with the other XCDs
// (at the headline batch an eighth is exactly one (graph, head group) / one head).  The grid is 8 * chunk; ids >= total do nothing.
// DGDM_ATTN_XCD 0: logical = L (rounds 1-5: a stream spread over all XCDs).
#ifndef DGDM_ATTN_XCD
#define DGDM_ATTN_XCD 1
#endif
__host__ __device__ inline int attn_xcd_chunk(int total) { return (total + 7) >> 3; }
__host__ __device__ inline unsigned attn_xcd_grid(int total) { return DGDM_ATTN_XCD ? 8u * (unsigned)attn_xcd_chunk(total) : (unsigned)total; }
__device__ __forceinline__ int attn_xcd_logical(int L, int total) { return DGDM_ATTN_XCD ? (L & 7) * attn_xcd_chunk(total) + (L >> 3) : L; }

// ---- blocks whose weights are EXACTLY zero (round 5) ------------------------------------------------------------------------------
// The reference feeds raw slide coordinates (pixels) into -distance / temperature (preprocessing/tissue_graph_builder.py:381-384,
// core/attention.py:261-283): two patches a few hundred pixels apart get exp(-hundreds) = 0.0f in fp32, i.e. on real slides all but a
// band of (query block, key block) pairs contribute exact zeros.  dgdm_attn_skip_map_build marks a pair "zero" when an upper bound
// of every score in it (|q'|_max |k|_max - smallest distance between the blocks' bounding boxes) lies more than ZERO_MARGIN
// (log2 units) below a lower bound of every row maximum of the query block (the row's score with ITSELF: distance 0).  For such a
// pair every exp2 argument the kernels form is below -150: the weights, and with them every product, are 0.0f -- skipping the
// pair changes no bit of the result.  Bit k of a row = pair (row's block, k-th block of the same graph) is zero; bits past the
// graph's last block are set.  One map per FORWARD head group (a pair is skipped only if it is zero for every head of the group).
#ifndef DGDM_FUSED_SBW
#define DGDM_FUSED_SBW 4      // 64-key blocks per workgroup of the one-pass backward (attn_h_bwd_fused.hip)
#endif
constexpr int ATTN_SBW = DGDM_FUSED_SBW;
constexpr float ATTN_ZERO_MARGIN = 200.0f;
__host__ __device__ inline int attn_map_group(int H) { return H % 4 == 0 ? 4 : (H % 2 == 0 ? 2 : 1); }
__host__ __device__ inline int attn_map_words(int num_blocks) { return 2 * (int)(((int64_t)num_blocks + 63) / 64); }     // words per row (>= any graph's blocks / 32)
// rows of the map: [0] by query block (bits = key blocks), [1] by key super-block (bits = query blocks; set = zero for all its key blocks)
__host__ __device__ inline int64_t attn_map_row(int which, int group, int H, int num_blocks, int blk) {
  return (((int64_t)which * (H / attn_map_group(H)) + group) * num_blocks + blk) * attn_map_words(num_blocks);
}
// The live (unmarked) blocks of a row in ascending order, then n for good.  Wave-uniform state in scalar registers; one word of the
// row is loaded per 32 blocks (a load per block would put a scalar-memory wait -- which also waits for the LDS -- into every
// iteration of the kernels' loops).  row == nullptr: every block is live.
struct LiveWalk {
  const uint32_t* row;
  int n, base;
  uint32_t w;
  __device__ __forceinline__ void init(const uint32_t* __restrict__ r, int n_) { row = r; n = n_; base = 0; w = r ? ~r[0] : 0xffffffffu; }
  __device__ __forceinline__ int next() {
    while (w == 0u) {
      base += 32;
      if (base >= n) return n;
      w = row ? ~row[base >> 5] : 0xffffffffu;
    }
    const int b = base + __builtin_ctz(w);
    w &= w - 1u;
    return b < n ? b : n;
  }
};

// Asynchronous copy of BYTES contiguous bytes global -> LDS by a 256-thread workgroup: each wave
// instruction moves 1 KiB (lane l: 16 B at offset piece*1024 + l*16; LDS destination = wave-uniform
// base + l*16).  Completion: the issuing wave's vmcnt, then a workgroup barrier (hipcc emits
// vmcnt(0) in front of __syncthreads()).
template <int BYTES>
__device__ __forceinline__ void dma_to_lds(const void* __restrict__ gsrc, void* lds_dst, int tid) {
  static_assert(BYTES % 16 == 0, "16-byte granules");
  constexpr int PIECES = (BYTES + 1023) / 1024;
  const int lane = tid & 63, wave = tid >> 6;
  const char* g = static_cast<const char*>(gsrc);
  char* l = static_cast<char*>(lds_dst);
#pragma unroll
  for (int p0 = 0; p0 < PIECES; p0 += 4) {
    const int p = p0 + wave;
    if (p < PIECES && p * 1024 + lane * 16 < BYTES)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + p * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(l + p * 1024), 16, 0, 0);
  }
}

// which graph / local block does packed block `blk` belong to (blocks are numbered graph by graph)
__device__ __forceinline__ bool find_block(const int32_t* __restrict__ ptr, int B, int blk, int* n0, int* ng, int* lblk, int* blk0) {
  int base = 0;
  for (int g = 0; g < B; ++g) {
    const int a = ptr[g], b = ptr[g + 1];
    const int nb = (b - a + HB - 1) / HB;
    if (blk < base + nb) { *n0 = a; *ng = b - a; *lblk = blk - base; *blk0 = base; return true; }
    base += nb;
  }
  return false;
}
