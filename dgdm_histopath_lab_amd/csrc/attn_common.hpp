// Shared pieces of the fused spatial-attention kernels (K4): MFMA wrappers, LDS tile images,
// graph lookup.  Head dim is fixed at 16 (Base: 128/8, Large: 256/16).
//
// MFMA shape: v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles/SIMD).  Operand maps (wave64):
//   A: lane l holds A[i = l&15][k = l>>4]      B: lane l holds B[k = l>>4][j = l&15]
//   C/D: lane l, reg r holds D[row = 4*(l>>4) + r][col = l&15]
// A product X^T = A*B whose rows are the NEXT product's reduction index can be fed back as the
// B operand straight from its accumulator registers: in step r lane (j, G=l>>4) supplies
// X^T[4G + r][j], i.e. reduction index k = G <-> row 4G + r, so the A operand of that step must
// be  A2[i][4G + r]  -- one ds_read_b128 of 4 consecutive rows per lane.
#pragma once
#include "common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DGDM_HEAD_DIM 16
#define DGDM_LOG2E 1.4426950408889634f

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// D[16x16] += A[16x16] * B[16x16] where lane (i, g) holds A[i][4g..4g+3] in `a` and lane (j, g)
// holds B^T[j][4g..4g+3] in `b` (both reduce over the same permuted index 4g+s).
__device__ __forceinline__ f32x4 mfma16_k16(const f32x4 a, const f32x4 b, f32x4 c) {
  c = mfma16(a[0], b[0], c);
  c = mfma16(a[1], b[1], c);
  c = mfma16(a[2], b[2], c);
  c = mfma16(a[3], b[3], c);
  return c;
}

// LDS images of a block of RB rows x HG heads x 16 floats:
//  "row-major"  img[h][row][16]           -> lane (row&15, g) reads 16 B at [row][4g]: 1 KiB per wave
//  "transposed" img[h][row/16][d][row%16] -> lane (d, G) reads 16 B at [d][4G]: rows 4G..4G+3
// head blocks are padded by 16 floats so that the 8 head slots of one staging instruction do
// not share banks.
template <int RB>
struct AttnTile {
  static constexpr int HS = RB * 16 + 16;  // floats per head block
  __device__ static __forceinline__ int rm(int h, int row, int d) { return h * HS + row * 16 + d; }
  __device__ static __forceinline__ int tr(int h, int row, int d) { return h * HS + (row >> 4) * 256 + d * 16 + (row & 15); }
};

// which graph does q-tile `tile` (of `rows_per_tile` rows) belong to?  B is small (slides per
// batch), a linear walk over ptr is cheaper than shipping a per-tile map from the host.
__device__ __forceinline__ bool find_graph(const int32_t* __restrict__ ptr, int B, int rows_per_tile, int tile,
                                           int* n0, int* n1, int* local_tile) {
  for (int g = 0; g < B; ++g) {
    const int a = ptr[g], b = ptr[g + 1];
    const int nt = (b - a + rows_per_tile - 1) / rows_per_tile;
    if (tile < nt) { *n0 = a; *n1 = b; *local_tile = tile; return true; }
    tile -= nt;
  }
  return false;
}

__device__ __forceinline__ float group_max4(float v) {  // max over lanes l, l^16, l^32, l^48
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ float group_sum4(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
