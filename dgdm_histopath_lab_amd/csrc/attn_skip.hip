// K4, zero-block map (round 5): which (query block, key block) pairs of the spatial attention contribute EXACT zeros (attn_h.hpp).
// Two small launches per attention forward, on the packed operands the attention kernels read anyway:
//   k_attn_block_stats   per (64-row block, head): max |q'|, max |k|, min over the block's rows of the row's score with itself
//                        (q'_i . k_i: its distance is 0, so every row maximum is at least that); per block: the bounding box of the
//                        (pre-scaled) positions;
//   k_attn_skip_map      bit (query block, key block) per forward head group, and the same bits gathered per key super-block of
//                        the one-pass backward (bit = zero for every key block of the super-block).
// The forward and the one-pass backward (attn_h_fwd.hip, attn_h_bwd_fused.hip) walk the set bits over; with positions in [0, 1)
// (BASELINE's synthetic slides) no bit is set and the kernels do what they did before.
#include "attn_h.hpp"

namespace {

__device__ __forceinline__ float wave_min(float v) { return -wave_max(-v); }

// stats[(blk * H + h) * 4 + {0: max |q'|, 1: max |k|, 2: min self score, 3: -}], bbox[blk * 4 + {xmin, xmax, ymin, ymax}]
__global__ __launch_bounds__(64) void k_attn_block_stats(const _Float16* __restrict__ Rq, const _Float16* __restrict__ Rk,
                                                         const float* __restrict__ pos_b, const int32_t* __restrict__ ptr, int B, int H,
                                                         float* __restrict__ stats, float* __restrict__ bbox) {
  const int blk = blockIdx.x, h = blockIdx.y, row = threadIdx.x;
  int n0, ng, lblk, blk0;
  if (!find_block(ptr, B, blk, &n0, &ng, &lblk, &blk0)) return;
  const bool valid = lblk * HB + row < ng;
  const _Float16* rq = Rq + ((int64_t)blk * H + h) * R_HEAD;
  const _Float16* rk = Rk + ((int64_t)blk * H + h) * R_HEAD;
  float qq = 0.f, kk = 0.f, qk = 0.f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const f16x8 qh = *reinterpret_cast<const f16x8*>(rq + r_off(row, c)), ql = *reinterpret_cast<const f16x8*>(rq + r_off(row, 2 + c));
    const f16x8 kh = *reinterpret_cast<const f16x8*>(rk + r_off(row, c)), kl = *reinterpret_cast<const f16x8*>(rk + r_off(row, 2 + c));
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float q = (float)qh[e] + (float)ql[e], k = (float)kh[e] + (float)kl[e];
      qq = fmaf(q, q, qq); kk = fmaf(k, k, kk); qk = fmaf(q, k, qk);
    }
  }
  const float mq = wave_max(valid ? qq : 0.f), mk = wave_max(valid ? kk : 0.f), ms = wave_min(valid ? qk : INFINITY);
  if (row == 0) {
    float* o = stats + ((int64_t)blk * H + h) * 4;
    o[0] = sqrtf(mq); o[1] = sqrtf(mk); o[2] = ms; o[3] = 0.f;
  }
  if (h == 0) {
    const float x = pos_b[((int64_t)blk * 2 + 0) * HB + row], y = pos_b[((int64_t)blk * 2 + 1) * HB + row];
    const float x0 = wave_min(valid ? x : INFINITY), x1 = wave_max(valid ? x : -INFINITY);
    const float y0 = wave_min(valid ? y : INFINITY), y1 = wave_max(valid ? y : -INFINITY);
    if (row == 0) {
      float* o = bbox + (int64_t)blk * 4;
      o[0] = x0; o[1] = x1; o[2] = y0; o[3] = y1;
    }
  }
}

// every weight of (query block qblk, key block kblk) is exactly zero for the heads [h0, h0 + hg): see attn_h.hpp.  The bounds are
// widened by 2^-10 relative (the kernels' scores carry ~2^-21; the box distance is formed like the kernels' distances); a NaN
// anywhere makes the comparison false (the pair is computed).
__device__ __forceinline__ bool pair_is_zero(const float* __restrict__ stats, const float* __restrict__ bbox, int H, int qblk, int kblk,
                                             int h0, int hg) {
  const float4 a = *reinterpret_cast<const float4*>(bbox + (int64_t)qblk * 4), b = *reinterpret_cast<const float4*>(bbox + (int64_t)kblk * 4);
  const float dx = fmaxf(0.f, fmaxf(a.x - b.y, b.x - a.y)), dy = fmaxf(0.f, fmaxf(a.z - b.w, b.z - a.w));
  const float mind = sqrtf(fmaf(dy, dy, dx * dx)) * (1.0f - 0x1p-10f);
  bool zero = true;
  for (int h = h0; h < h0 + hg; ++h) {
    const float* sq = stats + ((int64_t)qblk * H + h) * 4;
    const float kn = stats[((int64_t)kblk * H + h) * 4 + 1];
    const float ub = fmaf(sq[0] * kn, 1.0f + 0x1p-10f, 0x1p-10f);
    zero = zero && (ub - mind < sq[2] - ATTN_ZERO_MARGIN);
  }
  return zero;
}

// key SUPER-blocks as the one-pass backward numbers them (attn_h_bwd_fused.hip::find_sblock)
__device__ __forceinline__ bool find_sblock_m(const int32_t* __restrict__ ptr, int B, int sb, int* ng, int* sbl, int* blk0) {
  int base = 0, sbase = 0;
  for (int g = 0; g < B; ++g) {
    const int a = ptr[g], b = ptr[g + 1];
    const int nb = (b - a + HB - 1) / HB, nsb = (nb + ATTN_SBW - 1) / ATTN_SBW;
    if (sb < sbase + nsb) { *ng = b - a; *sbl = sb - sbase; *blk0 = base; return true; }
    base += nb; sbase += nsb;
  }
  return false;
}

// grid (num_blocks, H / group, 2).  z = 0: row of query block blockIdx.x; z = 1: row of key super-block blockIdx.x.
__global__ __launch_bounds__(256) void k_attn_skip_map(const float* __restrict__ stats, const float* __restrict__ bbox,
                                                       const int32_t* __restrict__ ptr, int B, int H, int num_blocks,
                                                       uint32_t* __restrict__ map) {
  const int hg = attn_map_group(H), g = blockIdx.y, W = attn_map_words(num_blocks);
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t* row = map + attn_map_row(blockIdx.z, g, H, num_blocks, blockIdx.x);
  if (blockIdx.z == 0) {
    int n0, ng, lblk, blk0;
    if (!find_block(ptr, B, blockIdx.x, &n0, &ng, &lblk, &blk0)) return;
    const int nbg = (ng + HB - 1) / HB;
    for (int b0 = 0; b0 < 32 * W; b0 += 256) {
      const int kb = b0 + tid;
      const bool z = kb >= nbg || pair_is_zero(stats, bbox, H, blockIdx.x, blk0 + kb, g * hg, hg);
      const unsigned long long m = __ballot(z);
      if (lane == 0 && kb < 32 * W) { row[kb >> 5] = (uint32_t)m; row[(kb >> 5) + 1] = (uint32_t)(m >> 32); }
    }
  } else {
    int ng, sbl, blk0;
    if (!find_sblock_m(ptr, B, blockIdx.x, &ng, &sbl, &blk0)) return;
    const int nbg = (ng + HB - 1) / HB;
    const int k0 = ATTN_SBW * sbl, k1 = min(k0 + ATTN_SBW, nbg);
    for (int b0 = 0; b0 < 32 * W; b0 += 256) {
      const int qb = b0 + tid;
      bool z = true;
      if (qb < nbg)
        for (int kb = k0; kb < k1; ++kb) z = z && pair_is_zero(stats, bbox, H, blk0 + qb, blk0 + kb, g * hg, hg);
      const unsigned long long m = __ballot(z);
      if (lane == 0 && qb < 32 * W) { row[qb >> 5] = (uint32_t)m; row[(qb >> 5) + 1] = (uint32_t)(m >> 32); }
    }
  }
}

// Scores the kernels EVALUATE under a map (measurement only: bench.py prices the attention kernels by the work they did, not by N^2):
// out[0] += rows(q block) x rows(k block) x heads for every unmarked (query block, key block) pair [the forward walks these],
// out[1] += rows(q block) x rows(key super-block) x heads for every unmarked (key super-block, query block) pair [the one-pass
// backward walks these: a super-block runs all its key tiles against a live query block], out[2] += N_g^2 x H (every pair).
// grid (num_blocks, H / group, 2) like k_attn_skip_map.
__global__ __launch_bounds__(256) void k_attn_map_count(const uint32_t* __restrict__ map, const int32_t* __restrict__ ptr, int B, int H,
                                                        int num_blocks, unsigned long long* __restrict__ out) {
  const int hg = attn_map_group(H), g = blockIdx.y;
  const uint32_t* row = map + attn_map_row(blockIdx.z, g, H, num_blocks, blockIdx.x);
  unsigned long long live = 0, all = 0;
  if (blockIdx.z == 0) {
    int n0, ng, lblk, blk0;
    if (!find_block(ptr, B, blockIdx.x, &n0, &ng, &lblk, &blk0)) return;
    const int nbg = (ng + HB - 1) / HB;
    const unsigned long long rq = (unsigned long long)min(HB, ng - lblk * HB);
    for (int kb = threadIdx.x; kb < nbg; kb += 256) {
      const unsigned long long rk = (unsigned long long)min(HB, ng - kb * HB);
      all += rq * rk * hg;
      if (!((row[kb >> 5] >> (kb & 31)) & 1u)) live += rq * rk * hg;
    }
  } else {
    int ng, sbl, blk0;
    if (!find_sblock_m(ptr, B, blockIdx.x, &ng, &sbl, &blk0)) return;
    const int nbg = (ng + HB - 1) / HB;
    const unsigned long long rk = (unsigned long long)(min(ng, (sbl + 1) * ATTN_SBW * HB) - sbl * ATTN_SBW * HB);
    for (int qb = threadIdx.x; qb < nbg; qb += 256) {
      const unsigned long long rq = (unsigned long long)min(HB, ng - qb * HB);
      if (!((row[qb >> 5] >> (qb & 31)) & 1u)) live += rq * rk * hg;
    }
  }
  __shared__ unsigned long long sm[2][256];
  sm[0][threadIdx.x] = live; sm[1][threadIdx.x] = all;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sm[0][threadIdx.x] += sm[0][threadIdx.x + o]; sm[1][threadIdx.x] += sm[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (sm[0][0]) atomicAdd(out + blockIdx.z, sm[0][0]);
    if (blockIdx.z == 0 && sm[1][0]) atomicAdd(out + 2, sm[1][0]);
  }
}

}  // namespace

// counts[0] = scores the forward evaluates under `map`, counts[1] = scores the one-pass backward evaluates, counts[2] = all scores
// (sum over graphs of n_g^2 x H); counts must be zero on entry (3 x uint64, device).  Measurement only (bench.py).
extern "C" int dgdm_attn_skip_map_count(const uint32_t* map, const int32_t* ptr, int32_t B, int32_t num_blocks, int32_t H, uint64_t* counts,
                                        void* stream_) {
  DGDM_REQUIRE(B >= 0 && H > 0 && num_blocks >= 0);
  if (num_blocks == 0 || B == 0) return DGDM_OK;
  DGDM_REQUIRE(map && ptr && counts);
  hipLaunchKernelGGL(k_attn_map_count, dim3(num_blocks, H / attn_map_group(H), 2), dim3(256), 0, static_cast<hipStream_t>(stream_), map, ptr, B, H,
                     num_blocks, reinterpret_cast<unsigned long long*>(counts));
  return dgdm_launch_status();
}

// bytes of the map (both sets of rows) and of the workspace (block statistics) for a batch of num_blocks packed blocks, H heads
extern "C" size_t dgdm_attn_skip_map_bytes(int32_t num_blocks, int32_t H) {
  if (num_blocks <= 0 || H <= 0) return 0;
  return (size_t)2 * (size_t)(H / attn_map_group(H)) * (size_t)num_blocks * (size_t)attn_map_words(num_blocks) * sizeof(uint32_t);
}
extern "C" size_t dgdm_attn_skip_map_workspace_bytes(int32_t num_blocks, int32_t H) {
  if (num_blocks <= 0 || H <= 0) return 0;
  return ((size_t)num_blocks * (size_t)H * 4 + (size_t)num_blocks * 4) * sizeof(float);
}

// Rq, Rk: the packed row images of Q' and K (dgdm_attn_pack), pos_b: the packed positions of the same call
extern "C" int dgdm_attn_skip_map_build(const void* Rq, const void* Rk, const float* pos_b, const int32_t* ptr, int32_t B,
                                        int32_t num_blocks, int32_t H, void* workspace, size_t workspace_bytes, uint32_t* map,
                                        size_t map_bytes, void* stream_) {
  DGDM_REQUIRE(B >= 0 && H > 0 && num_blocks >= 0);
  if (num_blocks == 0 || B == 0) return DGDM_OK;
  DGDM_REQUIRE(Rq && Rk && pos_b && ptr && workspace && map);
  if (!dgdm_aligned16(Rq) || !dgdm_aligned16(Rk) || !dgdm_aligned16(workspace) || !dgdm_aligned16(map)) return DGDM_ERR_UNSUPPORTED;
  if (workspace_bytes < dgdm_attn_skip_map_workspace_bytes(num_blocks, H) || map_bytes < dgdm_attn_skip_map_bytes(num_blocks, H))
    return DGDM_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  float* stats = static_cast<float*>(workspace);
  float* bbox = stats + (size_t)num_blocks * H * 4;
  hipLaunchKernelGGL(k_attn_block_stats, dim3(num_blocks, H), dim3(64), 0, s, static_cast<const _Float16*>(Rq),
                     static_cast<const _Float16*>(Rk), pos_b, ptr, B, H, stats, bbox);
  hipLaunchKernelGGL(k_attn_skip_map, dim3(num_blocks, H / attn_map_group(H), 2), dim3(256), 0, s, stats, bbox, ptr, B, H, num_blocks, map);
  return dgdm_launch_status();
}
