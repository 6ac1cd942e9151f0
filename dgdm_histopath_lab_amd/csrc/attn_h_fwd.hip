// K4 (split-fp16 path): operand packing + forward of the fused spatial attention.
// Same algorithm, tiling and outputs as attn_fwd.hip (64 queries x 4 heads per workgroup, 64-key
// blocks, online softmax in the log2 domain, distance bias shared by the head group, counter-hash
// dropout); the two products run on v_mfma_f32_16x16x32_f16 with hi+lo operands (attn_h.hpp):
//   S'^T = [K_hi|K_lo] . [Q'_hi|Q'_hi] + [K_hi|K_lo] . [Q'_lo|Q'_lo]   - bias   (C-in = -bias)
//   O^T += V^T_hi . P + V^T_lo . P           (P = fp16(exp2(S' - m)), pairs of 16-key tiles)
// which leaves the kernel bound by the softmax VALU work instead of MFMA + VALU.
#include "attn_h.hpp"

namespace {

constexpr int QB = 64;
constexpr float NEG_BIG = -1.0e30f;

// X [N, ncols] fp32 (row stride ld) -> out [N][ncols/16][32] halfs = [hi16 | lo16] of x * scale
__global__ __launch_bounds__(256) void k_split_pack(const float* __restrict__ X, int64_t ld, int64_t N, int ncols, float scale,
                                                    _Float16* __restrict__ out) {
  const int c4 = ncols >> 2;
  const int64_t total = N * c4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t n = i / c4;
    const int k = (int)(i % c4);           // float4 index inside the row
    const float4 v = *reinterpret_cast<const float4*>(X + n * ld + 4 * k);
    const float x[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = (_Float16)x[e];
      lo[e] = (_Float16)(x[e] - (float)hi[e]);
    }
    const int head = k >> 2, part = k & 3;  // 4 float4 per 16-wide head
    _Float16* dst = out + (n * (ncols >> 4) + head) * 32 + part * 4;
    *reinterpret_cast<f16x4*>(dst) = hi;
    *reinterpret_cast<f16x4*>(dst + 16) = lo;
  }
}

template <int HG, int KB, bool DROP>
__global__ __launch_bounds__(256) void k_attn_h_fwd(const _Float16* __restrict__ Qp, const _Float16* __restrict__ Kp,
                                                       const _Float16* __restrict__ Vp, int H, const float* __restrict__ pos,
                                                       const int32_t* __restrict__ ptr, int B, float bscale, float* __restrict__ O,
                                                       int64_t ldo, float* __restrict__ L2, int N_tot, float drop_p, uint32_t seed) {
  using T = HTile<KB>;
  constexpr int NT = KB / 16;
  constexpr int RS = T::RS + 16;                    // row-image head stride, +32 B so head slots differ in bank
  constexpr int CH = KB * HG * 4 / 256;              // 16-byte chunks staged per thread per tensor
  static_assert(NT % 2 == 0 && (KB * HG * 4) % 256 == 0, "tiling");
  __shared__ __attribute__((aligned(16))) _Float16 smem[HG * RS + 2 * HG * T::TH + 4 * KB];
  _Float16* Kimg = smem;
  _Float16* Vth = smem + HG * RS;
  _Float16* Vtl = Vth + HG * T::TH;
  float* Ps = reinterpret_cast<float*>(Vtl + HG * T::TH);  // [KB][2] key positions
  const DropCfg dc(drop_p);

  int n0, n1, ltile;
  if (!find_graph(ptr, B, QB, blockIdx.x, &n0, &n1, &ltile)) return;
  const int ng = n1 - n0;
  const int head0 = blockIdx.y * HG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, G = lane >> 4;
  const int q_local = ltile * QB + wave * 16 + j;
  const int q_row = n0 + (q_local < ng ? q_local : ng - 1);
  const bool q_ok = q_local < ng;

  f16x8 qb1[HG], qb2[HG];
  f32x4 oacc[HG], oacc2[HG];
  float m[HG], l[HG];
  uint32_t hq[HG];
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    load_b_pair(Qp + ((int64_t)q_row * H + head0 + h) * 32, G, &qb1[h], &qb2[h]);
    oacc[h] = f32x4{0.f, 0.f, 0.f, 0.f}; oacc2[h] = f32x4{0.f, 0.f, 0.f, 0.f}; m[h] = NEG_BIG; l[h] = 0.f;
    hq[h] = attn_head_seed(seed, n0, head0 + h) ^ (((uint32_t)q_local >> 1) * 0x9E3779B1U);
  }
  const float2 pq = *reinterpret_cast<const float2*>(pos + 2 * (int64_t)q_row);

  // staging: chunk idx -> (key, head, quarter); quarter 0/1 = hi[0..7]/hi[8..15], 2/3 = lo halves
  uint4 kreg[CH], vreg[CH];
  auto issue_loads = [&](int kb0) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int idx = tid + 256 * i;
      const int key = idx / (HG * 4), c = idx % (HG * 4);
      const int kl = kb0 + key;
      const int64_t off = ((int64_t)(n0 + (kl < ng ? kl : ng - 1)) * H + head0) * 32 + c * 8;  // clamped: always valid
      kreg[i] = *reinterpret_cast<const uint4*>(Kp + off);
      vreg[i] = *reinterpret_cast<const uint4*>(Vp + off);
    }
  };
  auto write_lds = [&](int kb0) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int idx = tid + 256 * i;
      const int key = idx / (HG * 4), c = idx % (HG * 4);
      const int h = c >> 2, quarter = c & 3;
      const bool ok = kb0 + key < ng;   // masked keys: zeros (0 * garbage must not poison the accumulators)
      uint4 kv = kreg[i], vv = vreg[i];
      if (!ok) { kv = make_uint4(0, 0, 0, 0); vv = make_uint4(0, 0, 0, 0); }
      *reinterpret_cast<uint4*>(&Kimg[h * RS + key * 32 + quarter * 8]) = kv;
      _Float16* vt = (quarter & 2) ? Vtl : Vth;
      const int d0 = (quarter & 1) * 8;
      const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint16_t bits = (uint16_t)(w[e >> 1] >> (16 * (e & 1)));
        reinterpret_cast<uint16_t*>(vt)[T::tr(h, d0 + e, key)] = bits;
      }
    }
    if (tid < KB) {
      const int kl = kb0 + tid;
      const float2 p = *reinterpret_cast<const float2*>(pos + 2 * (int64_t)(n0 + (kl < ng ? kl : ng - 1)));
      *reinterpret_cast<float2*>(&Ps[2 * tid]) = p;
    }
  };

  issue_loads(0);
  for (int kb0 = 0; kb0 < ng; kb0 += KB) {
    __syncthreads();
    write_lds(kb0);
    __syncthreads();
    if (kb0 + KB < ng) issue_loads(kb0 + KB);

    // minus the distance bias, as the C input of the first S MFMA (masked keys: -1e30)
    f32x4 nbias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 pa = *reinterpret_cast<const float4*>(&Ps[2 * (16 * t + 4 * G)]);
      const float4 pb = *reinterpret_cast<const float4*>(&Ps[2 * (16 * t + 4 * G) + 4]);
      const float kx[4] = {pa.x, pa.z, pb.x, pb.z}, ky[4] = {pa.y, pa.w, pb.y, pb.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dx = pq.x - kx[r], dy = pq.y - ky[r];
        const float d = __builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy)) * bscale;
        nbias[t][r] = (kb0 + 16 * t + 4 * G + r < ng) ? -d : NEG_BIG;
      }
    }

#pragma unroll
    for (int h = 0; h < HG; ++h) {
      f32x4 s[NT];
      float mx = NEG_BIG;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(&Kimg[h * RS + (16 * t + j) * 32 + 8 * G]);
        s[t] = mfma_h(kf, qb1[h], nbias[t]);
        s[t] = mfma_h(kf, qb2[h], s[t]);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
      mx = group_max4(mx);
      const float m_new = fmaxf(m[h], mx);
      const float alpha = __builtin_amdgcn_exp2f(m[h] - m_new);
      m[h] = m_new;
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[t][r] = __builtin_amdgcn_exp2f(s[t][r] - m_new);
          psum += s[t][r];
        }
      l[h] = fmaf(l[h], alpha, psum);
      oacc[h] *= alpha;
      oacc2[h] *= alpha;
      if (DROP) {
#pragma unroll
        for (int t = 0; t < NT; ++t) s[t] *= drop_factors_qmajor(hq[h], q_local, kb0 + 16 * t + 4 * G, dc);
      }
#pragma unroll
      for (int tp = 0; tp < NT / 2; ++tp) {
        const f16x8 pb = pack8(s[2 * tp], s[2 * tp + 1]);
        const f16x4 a0 = *reinterpret_cast<const f16x4*>(&Vth[T::tr(h, j, 32 * tp + 4 * G)]);
        const f16x4 a1 = *reinterpret_cast<const f16x4*>(&Vth[T::tr(h, j, 32 * tp + 16 + 4 * G)]);
        const f16x4 b0 = *reinterpret_cast<const f16x4*>(&Vtl[T::tr(h, j, 32 * tp + 4 * G)]);
        const f16x4 b1 = *reinterpret_cast<const f16x4*>(&Vtl[T::tr(h, j, 32 * tp + 16 + 4 * G)]);
        const f16x8 vh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const f16x8 vl = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        oacc[h] = mfma_h(vh, pb, oacc[h]);
        oacc2[h] = mfma_h(vl, pb, oacc2[h]);
      }
    }
  }

#pragma unroll
  for (int h = 0; h < HG; ++h) {
    const float lt = group_sum4(l[h]);
    const float inv = 1.0f / lt;
    if (q_ok) {
      const f32x4 os = oacc[h] + oacc2[h];
      *reinterpret_cast<float4*>(O + (int64_t)(n0 + q_local) * ldo + (head0 + h) * 16 + 4 * G) =
          make_float4(os[0] * inv, os[1] * inv, os[2] * inv, os[3] * inv);
      if (G == 0) L2[(int64_t)(head0 + h) * N_tot + n0 + q_local] = m[h] + log2f(lt);
    }
  }
}

}  // namespace

extern "C" int dgdm_attn_split_pack(const float* X, int64_t ld, int32_t N, int32_t ncols, float scale, void* out_halfs, void* stream) {
  DGDM_REQUIRE(N >= 0 && ncols > 0);
  if (N == 0) return DGDM_OK;
  DGDM_REQUIRE(X && out_halfs);
  if ((ncols & 15) || (ld & 3) || ld < ncols || !dgdm_aligned16(X) || !dgdm_aligned16(out_halfs)) return DGDM_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)N * (ncols >> 2);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_split_pack, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), X, ld, (int64_t)N, ncols,
                     scale, static_cast<_Float16*>(out_halfs));
  return dgdm_launch_status();
}

extern "C" int dgdm_spatial_attn_h_fwd(const void* Qp, const void* Kp, const void* Vp, const float* pos, const int32_t* ptr, int32_t B,
                                       int32_t num_q_tiles, int32_t N_tot, int32_t H, float inv_tau, float drop_p, uint32_t seed,
                                       float* O, int64_t ldo, float* lse2, int32_t variant, void* stream_) {
  DGDM_REQUIRE(B >= 0 && N_tot >= 0 && H > 0 && num_q_tiles >= 0 && drop_p >= 0.f && drop_p < 1.f);
  if (N_tot == 0 || num_q_tiles == 0) return DGDM_OK;
  DGDM_REQUIRE(Qp && Kp && Vp && pos && ptr && O && lse2);
  if ((ldo & 3) || ldo < H * 16 || !dgdm_aligned16(Qp) || !dgdm_aligned16(Kp) || !dgdm_aligned16(Vp) || !dgdm_aligned16(O) ||
      (reinterpret_cast<uintptr_t>(pos) & 7u))
    return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const float bscale = inv_tau * DGDM_LOG2E;
  const _Float16 *q = static_cast<const _Float16*>(Qp), *k = static_cast<const _Float16*>(Kp), *v = static_cast<const _Float16*>(Vp);
#define GO(HG, KB)                                                                                                          \
  do {                                                                                                                      \
    if (drop_p > 0.f)                                                                                                       \
      hipLaunchKernelGGL((k_attn_h_fwd<HG, KB, true>), dim3(num_q_tiles, H / HG), dim3(256), 0, s, q, k, v, H, pos, ptr, B,  \
                         bscale, O, ldo, lse2, N_tot, drop_p, seed);                                                        \
    else                                                                                                                    \
      hipLaunchKernelGGL((k_attn_h_fwd<HG, KB, false>), dim3(num_q_tiles, H / HG), dim3(256), 0, s, q, k, v, H, pos, ptr, B, \
                         bscale, O, ldo, lse2, N_tot, 0.f, 0u);                                                             \
  } while (0)
  if (H % 4 == 0 && variant == 1) GO(4, 32);
  else if (H % 2 == 0 && variant == 2) GO(2, 64);
  else if (H % 2 == 0 && variant == 3) GO(2, 32);
  else if (H % 4 == 0) GO(4, 64);
  else if (H % 2 == 0) GO(2, 64);
  else GO(1, 64);
#undef GO
  return dgdm_launch_status();
}
