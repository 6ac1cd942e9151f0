// K4 (split-fp16 path): operand packing + forward of the fused spatial attention.
// Same algorithm and outputs as attn_fwd.hip (64 queries x HG heads per workgroup, 64-key blocks,
// online softmax in the log2 domain, distance bias shared by the head group, counter-hash dropout);
// differences that matter for speed on gfx950:
//   * both products run on v_mfma_f32_16x16x32_f16 with hi+lo operands (attn_h.hpp):
//       S'^T = [K_hi|K_lo].[Q'_hi|Q'_hi] + [K_hi|K_lo].[Q'_lo|Q'_lo]  with C-in = -distance (packed fp32 arithmetic on
//       positions pre-scaled by log2(e)/tau)
//       O^T += V^T_hi.P_hi + V^T_lo.P_hi + V^T_hi.P_lo     (P = exp2(S' - m) carried as fp16 hi+lo like every operand)
//     so the matrix work overlaps with the softmax VALU work (the fp32 MFMA cannot);
//   * K / V / key positions are pre-packed, block-aligned images staged by direct-to-LDS DMA: no staging VGPRs, no staging
//     VALU; V^T fragments are read out of V's ROW image with transposed LDS reads (ds_read_b64_tr_b16);
//   * lazy running maximum: the cross-lane max exchange (two LDS round trips) and the rescale of
//     O / l only happen when some lane's block maximum exceeds the running maximum by > 2^LAZY_THR.
#include "attn_h.hpp"

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float LAZY_THR = 6.0f;  // exp2 arguments stay <= 6 between rescales: P <= 64, safe in fp16/fp32

// grid (NB, H, ntensors).  Tensor z: columns [col0 + z*cstride, +H*16) of X, scaled by (z == 0 ? scale0 : 1).
// fp16 range guard for the backward: alpha = 2^k with alpha * max|x| in [target/2, target]; out = {alpha, 1/alpha}.
// Two launches (block maxima, then the final max), no host round trip.
__global__ __launch_bounds__(256) void k_amax_partial(const float* __restrict__ x, int64_t n4, float* __restrict__ part) {
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  m = wave_max(m);
  __shared__ float sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}
__global__ __launch_bounds__(256) void k_amax_final(const float* __restrict__ part, int nparts, float target, float* __restrict__ out2) {
  float m = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, part[i]);
  m = wave_max(m);
  __shared__ float sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float amax = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float e = (amax > 0.f && amax < INFINITY) ? floorf(log2f(target / amax)) : 0.f;
    e = fminf(fmaxf(e, -60.f), 60.f);
    out2[0] = exp2f(e);
    out2[1] = exp2f(-e);
  }
}

__global__ __launch_bounds__(256) void k_attn_pack(const float* __restrict__ X, int64_t ld, int col0, int cstride, float scale0,
                                                   const float* __restrict__ scale_dev, const int32_t* __restrict__ ptr, int B, int H, _Float16* __restrict__ R,
                                                   int64_t r_tensor_stride,
                                                   const float* __restrict__ pos, float pos_scale, float* __restrict__ pos_b,
                                                   const float* __restrict__ Oin, int64_t ldo, float* __restrict__ ndelta_b,
                                                   const float* __restrict__ lse_in, float* __restrict__ lse_out) {
  const int blk = blockIdx.x, h = blockIdx.y, z = blockIdx.z;
  int n0, ng, lblk, blk0;
  if (!find_block(ptr, B, blk, &n0, &ng, &lblk, &blk0)) return;
  const int tid = threadIdx.x, row = tid >> 2, part = tid & 3;
  const int rl = lblk * HB + row;
  const bool ok = rl < ng;
  const int64_t node = n0 + (ok ? rl : ng - 1);
  const float scale = (z == 0 ? scale0 : 1.0f) * (scale_dev ? scale_dev[0] : 1.0f);
  const float4 v = *reinterpret_cast<const float4*>(X + node * ld + col0 + z * cstride + h * 16 + part * 4);
  const float x[4] = {ok ? v.x * scale : 0.f, ok ? v.y * scale : 0.f, ok ? v.z * scale : 0.f, ok ? v.w * scale : 0.f};
  f16x4 hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (_Float16)x[e];
    lo[e] = (_Float16)(x[e] - (float)hi[e]);
  }
  _Float16* rimg = R + z * r_tensor_stride + ((int64_t)blk * H + h) * R_HEAD;      // chunk-major tiles (attn_h.hpp, r_off)
  *reinterpret_cast<f16x4*>(rimg + r_off(row, part >> 1) + 4 * (part & 1)) = hi;
  *reinterpret_cast<f16x4*>(rimg + r_off(row, 2 + (part >> 1)) + 4 * (part & 1)) = lo;
  if (Oin) {  // -delta = -rowsum(dO * O) (attention backward: C input of the dP products), block layout [blk][H][64]
    const float4 o = *reinterpret_cast<const float4*>(Oin + node * ldo + h * 16 + part * 4);
    float d = x[0] * o.x + x[1] * o.y + x[2] * o.z + x[3] * o.w;
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    if (part == 0) {
      const int64_t idx = ((int64_t)blk * H + h) * HB + row;
      ndelta_b[idx] = ok ? -d : 0.f;
      if (lse_out) lse_out[idx] = DGDM_ATTN_P_SHIFT - lse_in[idx];   // the backward kernels carry P' = 2^P_SHIFT * P (attn_h.hpp)
    }
  }
  if (pos_b && z == 0 && h == 0 && tid < HB) {   // planar, pre-scaled by log2(e)/tau: [blk][x | y][64]
    const int r2 = lblk * HB + tid;
    const float2 p = *reinterpret_cast<const float2*>(pos + 2 * (int64_t)(n0 + (r2 < ng ? r2 : ng - 1)));
    pos_b[((int64_t)blk * 2 + 0) * HB + tid] = r2 < ng ? p.x * pos_scale : 0.f;
    pos_b[((int64_t)blk * 2 + 1) * HB + tid] = r2 < ng ? p.y * pos_scale : 0.f;
  }
}

template <int HG, bool DROP, int NBUF = 2, int WPE = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k_attn_h_fwd(const _Float16* __restrict__ Rq, const _Float16* __restrict__ Rk,
                                                    const _Float16* __restrict__ Rv, const float* __restrict__ pos_b, int H,
                                                    const int32_t* __restrict__ ptr, int B, float* __restrict__ O,
                                                    int64_t ldo, float* __restrict__ lse2_b, float drop_p, DgdmSeed seed_in,
                                                    const uint32_t* __restrict__ skip_map, unsigned* __restrict__ amax_out, int num_blocks) {
  const uint32_t seed = seed_in.value();
  constexpr int NT = HB / 16;
  constexpr int RK_BYTES = HG * R_HEAD * 2, TV_BYTES = RK_BYTES, POS_BYTES = HB * 8;      // K and V row images (V^T is read transposed)
  constexpr int BUF_BYTES = RK_BYTES + TV_BYTES + POS_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[NBUF * BUF_BYTES];
  const DropCfg dc(drop_p);

  // workgroup -> (head group, query block), head-group-major: the query blocks of a (graph, head group) are contiguous (attn_h.hpp)
  const int logical = attn_xcd_logical((int)blockIdx.x, num_blocks * (H / HG));
  if (logical >= num_blocks * (H / HG)) return;
  const int hgrp = __builtin_amdgcn_readfirstlane(logical / num_blocks), bx = logical - hgrp * num_blocks;      // (scalar registers)
  int n0, ng, lblk, blk0;
  if (!find_block(ptr, B, bx, &n0, &ng, &lblk, &blk0)) return;
  const int nbg = (ng + HB - 1) / HB;
  const int head0 = hgrp * HG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, G = lane >> 4;
  const int q_in_blk = wave * 16 + j;
  const int q_local = lblk * HB + q_in_blk;
  const bool q_ok = q_local < ng;

  auto stage = [&](int kb, int buf) {
    const int64_t gb = (int64_t)(blk0 + kb) * H + head0;
    char* base = smem + buf * BUF_BYTES;
    dma_to_lds<RK_BYTES>(Rk + gb * R_HEAD, base, tid);
    dma_to_lds<TV_BYTES>(Rv + gb * R_HEAD, base + RK_BYTES, tid);
    dma_to_lds<POS_BYTES>(pos_b + (int64_t)(blk0 + kb) * HB * 2, base + RK_BYTES + TV_BYTES, tid);
  };
  // key blocks whose weights are exactly zero for this query block and every head of its group are walked over (attn_h.hpp;
  // the block that holds the rows themselves never is: kb < nbg below)
  const uint32_t* srow = skip_map ? skip_map + attn_map_row(0, head0 / attn_map_group(H), H, num_blocks, bx) : nullptr;
  LiveWalk live;
  live.init(srow, nbg);
  int kb = __builtin_amdgcn_readfirstlane(live.next());
  if (kb >= nbg) kb = lblk;      // (a map that marks the rows' own block cannot come out of dgdm_attn_skip_map_build)
  stage(kb, 0);

  f16x8 qb1[HG], qb2[HG];
  f32x4 oacc[HG], lacc[HG];  // lacc: every register = sum over keys of the weights (ones . [P_hi + P_lo])
  float m[HG];
  DropLaneQ dl[HG];
  const _Float16 one = (_Float16)1.0f;
  const f16x8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    load_b_pair(Rq + ((int64_t)bx * H + head0 + h) * R_HEAD, q_in_blk, G, &qb1[h], &qb2[h]);
    oacc[h] = f32x4{0.f, 0.f, 0.f, 0.f}; lacc[h] = f32x4{0.f, 0.f, 0.f, 0.f}; m[h] = NEG_BIG;
    if (DROP) dl[h] = DropLaneQ(DropHead(seed, n0, head0 + h), q_local);
  }
  const uint32_t lck = __umul24(2u * (uint32_t)G, DROP_CK);
  const int aoff = r_lane_off(j, G);
  const float px = pos_b[((int64_t)bx * 2 + 0) * HB + q_in_blk], py = pos_b[((int64_t)bx * 2 + 1) * HB + q_in_blk];
  __syncthreads();  // block 0 landed (vmcnt(0) + barrier)

  for (int it = 0; kb < nbg; ++it) {
    const int buf = NBUF == 2 ? (it & 1) : 0;
    const int kb_next = __builtin_amdgcn_readfirstlane(live.next());
    if (NBUF == 2 && kb_next < nbg) stage(kb_next, buf ^ 1);  // lands under this block's math; its buffer was released by the last barrier
    const _Float16* Kimg = reinterpret_cast<const _Float16*>(smem + buf * BUF_BYTES);
    const _Float16* Vimg = reinterpret_cast<const _Float16*>(smem + buf * BUF_BYTES + RK_BYTES);
    const float* Ps = reinterpret_cast<const float*>(smem + buf * BUF_BYTES + RK_BYTES + TV_BYTES);
    const int kb0 = kb * HB;

    f32x4 ndist[NT];  // -log2(e)/tau * |p_q - p_k| = C input of the first S MFMA (packed fp32 arithmetic); masked keys: -1e30
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      ndist[t] = sub4(0.f, dist4(&Ps[16 * t + 4 * G], &Ps[HB + 16 * t + 4 * G], px, py));
    }
    if (kb0 + HB > ng) {   // only the last key block of a graph has keys to mask (wave-uniform branch)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kb0 + 16 * t + 4 * G + r >= ng) ndist[t][r] = NEG_BIG;
    }

#pragma unroll
    for (int h = 0; h < HG; ++h) {
      f32x4 s[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(Kimg + h * R_HEAD + t * 512 + aoff);
        s[t] = mfma_h(kf, qb1[h], ndist[t]);
        s[t] = mfma_h(kf, qb2[h], s[t]);
      }
      float mloc = fmaxf(s[0][0], s[0][1]);   // a chain the compiler folds into v_max3_f32: two values per instruction
      mloc = fmaxf(fmaxf(mloc, s[0][2]), s[0][3]);
#pragma unroll
      for (int t = 1; t < NT; ++t) {
        mloc = fmaxf(fmaxf(mloc, s[t][0]), s[t][1]);
        mloc = fmaxf(fmaxf(mloc, s[t][2]), s[t][3]);
      }
      if (__any(mloc > m[h] + LAZY_THR)) {  // wave-uniform: rare after the first blocks
        const float m_new = fmaxf(m[h], group_max4(mloc));
        const float alpha = __builtin_amdgcn_exp2f(m[h] - m_new);
        m[h] = m_new;
        lacc[h] *= alpha;
        oacc[h] *= alpha;
      }
      const float nm = -m[h];
      const f32x4 mh = {nm, nm, nm, nm};
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        s[t] = s[t] + mh;   // vector form: two v_pk_add_f32 instead of four v_sub_f32
#pragma unroll
        for (int r = 0; r < 4; ++r) s[t][r] = __builtin_amdgcn_exp2f(s[t][r]);
      }
      const _Float16* vh = Vimg + h * R_HEAD;
#pragma unroll
      for (int tp = 0; tp < NT / 2; ++tp) {
        // the softmax denominator is summed from the same hi+lo weights that multiply V, on the matrix pipe (ones . P): no VALU
        // adds, and the result already covers all four lane groups of a query
        f16x8 ph, pl;
        split8(s[2 * tp], s[2 * tp + 1], &ph, &pl);
        lacc[h] = mfma_h(ones, ph, lacc[h]);
        lacc[h] = mfma_h(ones, pl, lacc[h]);
        if (DROP) {  // dropout applies to the normalised weights: mask only what multiplies V.  The packed halfs are masked
                     // in place (0xFFFF / 0 per half-word); the constant 1/(1-p) multiplies the finished row below.
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          u32x4 mk;
          uint32_t e[4];
          drop_words_q(dl[h], (uint32_t)(kb0 + 32 * tp), lck, e);
          mk[0] = drop_pair_mask(e[1], e[0], dc); mk[1] = drop_pair_mask(e[3], e[2], dc);
          drop_words_q(dl[h], (uint32_t)(kb0 + 32 * tp + 16), lck, e);
          mk[2] = drop_pair_mask(e[1], e[0], dc); mk[3] = drop_pair_mask(e[3], e[2], dc);
          ph = __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4, ph) & mk);
          pl = __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4, pl) & mk);
        }
        const f16x8 vhi = load_tr_pair(vh, 0, 2 * tp, lane);                     // V^T[d = j][keys of the tile pair], out of the row image
        oacc[h] = mfma_h(vhi, ph, oacc[h]);
        oacc[h] = mfma_h(load_tr_pair(vh, 1, 2 * tp, lane), ph, oacc[h]);        // lo part of V: same accumulator
        oacc[h] = mfma_h(vhi, pl, oacc[h]);                                      // lo part of P (V_lo . P_lo ~ 2^-22: dropped)
      }
    }
    __syncthreads();  // everyone is done with `buf`; the DMA of the next block has landed
    if (NBUF == 1 && kb_next < nbg) {   // single buffer: other workgroups of the CU (3 per CU) cover this load
      stage(kb_next, 0);
      __syncthreads();
    }
    kb = kb_next;
  }

  unsigned am = 0;      // max |O| of the workgroup's rows: the operand maximum of the output projection (no reduction launch)
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    const float lt = lacc[h][0];
    const float inv = (DROP ? dc.keep : 1.0f) / lt;
    if (q_ok) {
      const f32x4 os = oacc[h];
      const float4 o4 = make_float4(os[0] * inv, os[1] * inv, os[2] * inv, os[3] * inv);
      *reinterpret_cast<float4*>(O + (int64_t)(n0 + q_local) * ldo + (head0 + h) * 16 + 4 * G) = o4;
      am = dgdm_amax4(am, o4);
    }
    if (G == 0) lse2_b[((int64_t)bx * H + head0 + h) * HB + q_in_blk] = q_ok ? m[h] + log2f(lt) : 0.f;
  }
  if (amax_out) dgdm_amax_commit(am, amax_out);      // kernel argument: every thread of the workgroup is here
}

}  // namespace

extern "C" size_t dgdm_attn_pack_bytes(int32_t num_blocks, int32_t H, int32_t which) {
  // which: 0 = row image, 2 = [blk][2][64] positions, 3 = [blk][H][64] fp32 per-row scalars
  const size_t nb = num_blocks > 0 ? num_blocks : 0, h = H > 0 ? H : 0;
  switch (which) {
    case 0: return nb * h * R_HEAD * 2;
    case 2: return nb * HB * 2 * sizeof(float);
    case 3: return nb * h * HB * sizeof(float);
    default: return 0;
  }
}

// Packs `ntensors` column blocks of X (tensor z = columns [col0 + z*cstride, +H*16)) into row images; tensor 0 is scaled by scale0.  pos_b (nullable): block-aligned positions, planar, times
// pos_scale.  O (nullable): when given, ndelta_b[blk][H][64] = -rowsum(X_0 * O) (X_0 = dO in the backward) and, with lse_in /
// lse_out, lse_out = DGDM_ATTN_P_SHIFT - lse_in.
extern "C" size_t dgdm_amax_scale_workspace_bytes(void) { return 1024 * sizeof(float); }

// out2[0] = alpha = 2^k such that alpha * max|x| lies in (target/2, target]; out2[1] = 1/alpha.  x: n contiguous floats.
extern "C" int dgdm_amax_pow2_scale(const float* x, int64_t n, float target, float* out2, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  DGDM_REQUIRE(n >= 0 && out2 && workspace && target > 0.f);
  if (workspace_bytes < dgdm_amax_scale_workspace_bytes()) return DGDM_ERR_WORKSPACE;
  if (n > 0 && (!x || (n & 3) || !dgdm_aligned16(x))) return n > 0 && !x ? DGDM_ERR_INVALID_ARG : DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  const int64_t n4 = n >> 2;
  int blocks = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_amax_partial, dim3(blocks), dim3(256), 0, s, x, n4, part);
  hipLaunchKernelGGL(k_amax_final, dim3(1), dim3(256), 0, s, part, blocks, target, out2);
  return dgdm_launch_status();
}

extern "C" int dgdm_attn_pack(const float* X, int64_t ld, int32_t col0, int32_t cstride, int32_t ntensors, float scale0,
                              const float* scale_dev, const int32_t* ptr, int32_t B, int32_t num_blocks, int32_t H, void* R,
                              const float* pos, float pos_scale, float* pos_b, const float* O, int64_t ldo,
                              float* ndelta_b, const float* lse_in, float* lse_out, void* stream) {
  DGDM_REQUIRE(B >= 0 && num_blocks >= 0 && H > 0 && ntensors > 0 && ntensors <= 4);
  if (num_blocks == 0 || B == 0) return DGDM_OK;
  DGDM_REQUIRE(X && ptr && R);
  DGDM_REQUIRE((pos == nullptr) == (pos_b == nullptr) && (O == nullptr) == (ndelta_b == nullptr));
  DGDM_REQUIRE((lse_in == nullptr) == (lse_out == nullptr) && (lse_out == nullptr || O != nullptr));
  if ((ld & 3) || (col0 & 3) || (cstride & 3) || !dgdm_aligned16(X) || !dgdm_aligned16(R) ||
      (O && ((ldo & 3) || !dgdm_aligned16(O))))
    return DGDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_attn_pack, dim3(num_blocks, H, ntensors), dim3(256), 0, static_cast<hipStream_t>(stream), X, ld, col0, cstride,
                     scale0, scale_dev, ptr, B, H, static_cast<_Float16*>(R), (int64_t)num_blocks * H * R_HEAD,
                     pos, pos_scale, pos_b, O, ldo, ndelta_b, lse_in, lse_out);
  return dgdm_launch_status();
}

// skip_map (nullable): the zero-block map dgdm_attn_skip_map_build made from the same packed operands; amax_out (nullable): a zeroed
// operand-maximum slot group (dgdm_amax_bits' layout) that receives max |O|
extern "C" int dgdm_spatial_attn_h_fwd_sparse(const void* Rq, const void* Rk, const void* Rv, const float* pos_b, const int32_t* ptr,
                                              int32_t B, int32_t num_blocks, int32_t H, float drop_p, uint32_t seed, float* O,
                                              int64_t ldo, float* lse2_b, int32_t variant, const uint32_t* skip_map, uint32_t* amax_out,
                                              void* stream_) {
  DGDM_REQUIRE(B >= 0 && H > 0 && num_blocks >= 0 && drop_p >= 0.f && drop_p < 1.f);
  if (num_blocks == 0 || B == 0) return DGDM_OK;
  DGDM_REQUIRE(Rq && Rk && Rv && pos_b && ptr && O && lse2_b);
  if ((ldo & 3) || ldo < H * 16 || !dgdm_aligned16(Rq) || !dgdm_aligned16(Rk) || !dgdm_aligned16(Rv) || !dgdm_aligned16(O) ||
      !dgdm_aligned16(pos_b) || (skip_map && !dgdm_aligned16(skip_map)))
    return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const _Float16 *q = static_cast<const _Float16*>(Rq), *k = static_cast<const _Float16*>(Rk), *v = static_cast<const _Float16*>(Rv);
#define GO(HG, NBUF, WPE)                                                                                                   \
  do {                                                                                                                      \
    if (drop_p > 0.f)                                                                                                       \
      hipLaunchKernelGGL((k_attn_h_fwd<HG, true, NBUF, WPE>), dim3(attn_xcd_grid(num_blocks * (H / HG))), dim3(256), 0, s, q, k, v, pos_b, H, ptr, B,    \
                         O, ldo, lse2_b, drop_p, dgdm_seed_arg(seed), skip_map, amax_out, num_blocks);                                       \
    else                                                                                                                    \
      hipLaunchKernelGGL((k_attn_h_fwd<HG, false, NBUF, WPE>), dim3(attn_xcd_grid(num_blocks * (H / HG))), dim3(256), 0, s, q, k, v, pos_b, H, ptr, B,   \
                         O, ldo, lse2_b, 0.f, dgdm_seed_arg(0u), skip_map, amax_out, num_blocks);                                            \
  } while (0)
  // variant 0 = default; 1..3 select a tiling explicitly (tools/microbench_attn.py)
  if (H % 4 == 0 && variant == 1) GO(4, 2, 2);        // 4 heads, double-buffered, 2 workgroups per CU
  else if (H % 4 == 0 && variant == 2) GO(4, 1, 3);   // 4 heads, single buffer, 3 workgroups per CU
  else if (H % 2 == 0 && variant == 3) GO(2, 2, 4);   // 2 heads, double-buffered, 4 workgroups per CU
  // default: 4 heads share one distance bias, ONE staging buffer, three workgroups per CU -- the other workgroups' math covers
  // a workgroup's DMA wait better than its own second buffer did at two per CU (1.29 vs 1.46 ms at 4 x 10k nodes, dropout on)
  else if (H % 4 == 0) GO(4, 1, 3);
  else if (H % 2 == 0) GO(2, 2, 4);
  else GO(1, 2, 2);
#undef GO
  return dgdm_launch_status();
}

extern "C" int dgdm_spatial_attn_h_fwd(const void* Rq, const void* Rk, const void* Rv, const float* pos_b, const int32_t* ptr,
                                       int32_t B, int32_t num_blocks, int32_t H, float drop_p, uint32_t seed, float* O,
                                       int64_t ldo, float* lse2_b, int32_t variant, void* stream_) {
  return dgdm_spatial_attn_h_fwd_sparse(Rq, Rk, Rv, pos_b, ptr, B, num_blocks, H, drop_p, seed, O, ldo, lse2_b, variant, nullptr, nullptr, stream_);
}
