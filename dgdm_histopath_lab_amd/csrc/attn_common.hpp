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

// ---- dropout on attention weights (core/attention.py:154: attn_dropout on softmax(S)) ----------
// Counter-based: element (graph, head, q, k) is kept iff a 16-bit slice of
// hash(seed, graph, head, q>>1, k) is >= drop_p * 65536; the low half serves even q, the high half
// odd q.  Pairing along q lets the k-major kernel (dK/dV: queries in registers) derive two
// elements per hash directly, and the q-major kernels (forward, dQ: keys in registers, queries on
// adjacent lanes) split the four keys of a register quad between the even/odd lane of a query
// pair and exchange the results with one DPP swap each -- 2 hashes per 4 elements everywhere.
// The backward kernels regenerate exactly the forward's mask from (seed, indices).
// hq = head_seed ^ (qp * 0x9E3779B1): the part that does not depend on the key (hoisted by callers)
// Counter hash of (seed, graph, head, query pair, key) -> 32 bits = two 16-bit uniform lanes (even / odd
// query of the pair).  The kernels are VALU-bound and the mask costs more than the softmax itself, so the
// per-element part is kept to 7 issue slots: the 32-bit v_mul_lo_u32 is quarter rate on gfx950, the 24-bit
// v_mul_u32_u24 is full rate, and node indices are < 2^24:
//     x = hs ^ mul24(q >> 1, C3) ^ mul24(k, C1);  x ^= x >> 16;  x = mul24(x, C2);  x ^= x >> 12
// with hs = a full 32-bit finaliser of (seed, graph offset, head), computed once per head.  Statistics
// (tools / tests): drop rate within 1e-4 of p, correlations between adjacent keys, queries, diagonals,
// heads, seeds all <= 5e-3 (the previous two-multiply hash: the same level), row / column rates binomial.
__device__ __forceinline__ uint32_t attn_fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6BU; x ^= x >> 13; x *= 0xC2B2AE35U; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t attn_head_seed(uint32_t seed, int n0, int head) {
  return attn_fmix32(seed ^ ((uint32_t)n0 * 0xC2B2AE35U) ^ ((uint32_t)(head + 1) * 0x27D4EB2FU));
}
// per-(head, query pair) part, hoisted out of the key loop where the query is fixed per lane
__device__ __forceinline__ uint32_t attn_hq(uint32_t hs, int q_local) { return hs ^ __umul24((uint32_t)q_local >> 1, 0x79B1A5U); }
__device__ __forceinline__ uint32_t attn_hash_hq(uint32_t hq, uint32_t k) {
  uint32_t x = hq ^ __umul24(k, 0x5BCA6BU);
  x ^= x >> 16; x = __umul24(x, 0x3C6D2BU); x ^= x >> 12;
  return x;
}
__device__ __forceinline__ uint32_t attn_hash(uint32_t hs, uint32_t qp, uint32_t k) {
  return attn_hash_hq(hs ^ __umul24(qp, 0x79B1A5U), k);
}
struct DropCfg {
  uint32_t thresh;
  float keep;
  __device__ __forceinline__ DropCfg(float p) : thresh((uint32_t)(p * 65536.0f)), keep(1.0f / (1.0f - (float)((uint32_t)(p * 65536.0f)) / 65536.0f)) {}
};
// q-major: this lane is query `q_local` (lane bit 0 == q_local & 1), its register quad holds keys
// k0 .. k0+3.  Returns the four keep-factors.
// `hq` = attn_hq(attn_head_seed(...), q_local), precomputed per (lane, head).
__device__ __forceinline__ f32x4 drop_factors_qmajor(uint32_t hq, int q_local, int k0, const DropCfg& c) {
  const bool odd = q_local & 1;
  const uint32_t kk = (uint32_t)k0 + (odd ? 2u : 0u);
  const uint32_t h0 = attn_hash_hq(hq, kk), h1 = attn_hash_hq(hq, kk + 1);
  const uint32_t o0 = __shfl_xor(h0, 1, 64), o1 = __shfl_xor(h1, 1, 64);
  const uint32_t u0 = odd ? o0 : h0, u1 = odd ? o1 : h1, u2 = odd ? h0 : o0, u3 = odd ? h1 : o1;
  const int sh = odd ? 16 : 0;
  f32x4 f;
  f[0] = ((u0 >> sh) & 0xFFFFu) >= c.thresh ? c.keep : 0.f;
  f[1] = ((u1 >> sh) & 0xFFFFu) >= c.thresh ? c.keep : 0.f;
  f[2] = ((u2 >> sh) & 0xFFFFu) >= c.thresh ? c.keep : 0.f;
  f[3] = ((u3 >> sh) & 0xFFFFu) >= c.thresh ? c.keep : 0.f;
  return f;
}
// k-major: this lane is key `k_local`, its register quad holds queries q0 .. q0+3 (q0 even).
__device__ __forceinline__ f32x4 drop_factors_kmajor(uint32_t hs, int k_local, int q0, const DropCfg& c) {
  const uint32_t h0 = attn_hash(hs, (uint32_t)q0 >> 1, (uint32_t)k_local), h1 = attn_hash(hs, ((uint32_t)q0 >> 1) + 1, (uint32_t)k_local);
  f32x4 f;
  f[0] = (h0 & 0xFFFFu) >= c.thresh ? c.keep : 0.f;
  f[1] = (h0 >> 16) >= c.thresh ? c.keep : 0.f;
  f[2] = (h1 & 0xFFFFu) >= c.thresh ? c.keep : 0.f;
  f[3] = (h1 >> 16) >= c.thresh ? c.keep : 0.f;
  return f;
}
