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
#pragma once
#include "attn_common.hpp"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_h(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// B-operand registers of a 16-wide row x given as packed [hi16 | lo16] halfs (64 B):
// lane group G reads hi[8*(G&1) .. +7] for B1 and lo[8*(G&1) .. +7] for B2.
__device__ __forceinline__ void load_b_pair(const _Float16* __restrict__ packed_row, int G, f16x8* b1, f16x8* b2) {
  *b1 = *reinterpret_cast<const f16x8*>(packed_row + 8 * (G & 1));
  *b2 = *reinterpret_cast<const f16x8*>(packed_row + 16 + 8 * (G & 1));
}

// 8 fp32 values (two accumulator quads) -> 8 halfs (B operand of the following product)
__device__ __forceinline__ f16x8 pack8(const f32x4 a, const f32x4 b) {
  f16x8 r;
  r[0] = (_Float16)a[0]; r[1] = (_Float16)a[1]; r[2] = (_Float16)a[2]; r[3] = (_Float16)a[3];
  r[4] = (_Float16)b[0]; r[5] = (_Float16)b[1]; r[6] = (_Float16)b[2]; r[7] = (_Float16)b[3];
  return r;
}

// LDS images for a block of RB rows x HG heads (halfs):
//   row image   R[h][row][32]      = [hi16 | lo16] per row (64 B): A operand with the row on the MFMA row
//   transposed  T[h][d][RB + 8]    (one image for hi, one for lo): A operand with d on the MFMA row and
//               the block's rows as reduction index; row stride RB+8 halfs (144 B at RB = 64) keeps
//               the 8-byte reads of one 32-lane group on distinct banks.
template <int RB>
struct HTile {
  static constexpr int RS = RB * 32;             // halfs per head in the row image
  static constexpr int TS = RB + 8;              // row stride of a transposed image
  static constexpr int TH = 16 * TS;             // halfs per head in a transposed image
  __device__ static __forceinline__ int row(int h, int r, int e) { return h * RS + r * 32 + e; }
  __device__ static __forceinline__ int tr(int h, int d, int r) { return h * TH + d * TS + r; }
};
