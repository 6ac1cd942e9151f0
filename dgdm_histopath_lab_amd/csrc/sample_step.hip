// K8-fused: ONE step of DiffusionLayer.sample (reference: core/diffusion.py:214-275 -- the "T-step denoise loop" of the hot path) as one
// launch:   eps = denoise_net([x | t_emb])   then   x <- sqrt(alpha) (x - sqrt(1 - ac) eps) / sqrt(ac) + sqrt(var) z.
// Round 5 ran a step as seven dependent launches (three tile GEMMs, two GroupNorm + SiLU rows, a normal draw, the update) that move
// 15.9 KB per row through HBM: 90 us per step at 10 000 rows, 0.22 of the HBM peak, because each launch is start-up and drain.  But every
// operation of the step is ROW-LOCAL (Linear, GroupNorm over the channels of a row, SiLU, the update): a workgroup takes 32 rows
// through the whole denoiser with the activations in LDS and leaves 3 x 4 C bytes per row of HBM traffic (x in, z in, x out).
//   denoise_net (core/diffusion.py:92-104, C = node_dim, hidden = 2 C):
//     Linear(C + 2C -> 4C) [the time half enters as a per-step bias: ops / DESIGN section 3] -> GroupNorm(8) -> SiLU -> (dropout: eval)
//     Linear(4C -> 2C) -> GroupNorm(8) -> SiLU -> Linear(2C -> C)
// Arithmetic = the image GEMMs': fp16 hi + lo operands, a.b = a_lo.b_hi + a_hi.b_lo + a_hi.b_hi on v_mfma_f32_32x32x16_f16, fp32
// accumulate; the weights come from the SAME pre-split images the training step uses (csrc/gemm_img.hip: fragment order, power-of-two
// scale in the header), read straight from L2 (0.9 MB for all three layers at C = 128); the activations are split when they are
// written to LDS.  x is scaled per workgroup by the power of two of its own maximum (early steps divide by sqrt(ac) ~ 0.02: the fp16
// range is not to be trusted with x), the normalised activations are O(10) and go unscaled.
// Workgroup = 4 waves x 32 rows.  Layer l's 32-column output tiles are dealt to the waves (4 / 2 / 1 per wave at C = 128); a wave's A
// fragments of a 32-k chunk are four 16-byte LDS reads shared by its tiles.  Between layers the accumulators go to LDS as fp32
// (pre-norm), thread (row, group) normalises ITS group of the row (8 threads per row = the 8 groups), and writes the result back as
// fp16 hi / lo planes over the same bytes -- the next layer's A operand.  LDS: 32 rows x (16 C + 16) bytes + the GroupNorm parameters = 72 KB at C = 128
// (two workgroups per CU), 144 KB at C = 256.
#include "gemm_epi.hpp"

namespace {

struct StepArgs {
  const float* x; int64_t ldx;
  const float* z; int64_t ldz;          // null on the last step
  float* out; int64_t ldo;
  int N;
  const char* img0; int tiles0;         // Linear 0, columns 0 .. C-1 of its weight (the x half): image of [4C, C]
  const char* img1; int tiles1;         // Linear 4: [2C, 4C]
  const char* img2; int tiles2;         // Linear 8: [C, 2C]
  const float* bias0;                   // [4C]: time half of Linear 0 through the step's t_emb + its bias
  const float* g1; const float* be1; float eps1;
  const float* bias1;
  const float* g2; const float* be2; float eps2;
  const float* bias2;
  float s1mac, sac, salpha, svar;
  int last;
};

// v sigmoid(v) with the hardware reciprocal (1 ulp) in place of the ~10-instruction IEEE division: 96 of these per thread and step
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// B fragment (tile t, 32-k chunk c, MFMA j, part p) of an image with T tiles per chunk: csrc/gemm_img.hip "Image layout"
// diagnostic builds only (tools/build_variant_lib.sh -DDGDM_STEP_DIAG=n): 1 = no weight loads (one fragment reused), 2 = no GroupNorm
// arithmetic, 4 = no MFMAs -- what each part of the kernel costs (tools/microbench_sample_step.py)
#ifndef DGDM_STEP_DIAG
#define DGDM_STEP_DIAG 0
#endif
__device__ __forceinline__ f16x8 b_frag(const char* __restrict__ img, int T, int c, int t, int j, int p, int lane) {
  if (DGDM_STEP_DIAG & 1) return *reinterpret_cast<const f16x8*>(img + IMG_HDR + lane * 16);
  return *reinterpret_cast<const f16x8*>(img + IMG_HDR + ((size_t)c * T + t) * BLK + (2 * j + p) * 1024 + lane * 16);
}

// acc[t] += A(rows of the workgroup, k) . B(tile, k) over K; A from the LDS planes (hi at byte 0, lo at byte 2 K of a row).
// The B fragments come straight from the weight image in L2 (a few hundred ns away): they are loaded PF chunks AHEAD into a ring of
// PF + 1 register sets -- 4 NT registers per fragment set, 128 registers of ring at every layer (PF = 1 / 3 / 7 for 4 / 2 / 1 tiles per
// wave) -- so that a wave always has several loads in flight behind the MFMAs it is issuing.  The loop is fully unrolled: every ring
// index is a constant.
template <int K, int NT, int PITCH, int PF>
__device__ __forceinline__ void layer_mfma(f32x16 (&acc)[NT], const char* __restrict__ smem, const char* __restrict__ img, int T, int tile0, int lane) {
  constexpr int NC = K / 32, RING = PF + 1;
  const char* arow = smem + (lane & 31) * PITCH + 32 * (lane >> 5);     // halfs 16 kg .. of the lane's row, hi plane
  f16x8 bq[RING][NT][2][2];                                              // [ring slot][tile][MFMA j][hi / lo]
#define DGDM_LOAD_B(c_)                                                                                             \
  _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                                     \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                  \
      bq[(c_) % RING][t][j][0] = b_frag(img, T, (c_), tile0 + t, j, 0, lane);                                        \
      bq[(c_) % RING][t][j][1] = b_frag(img, T, (c_), tile0 + t, j, 1, lane);                                        \
    }
#pragma unroll
  for (int c = 0; c < (PF < NC ? PF : NC); ++c) DGDM_LOAD_B(c)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c + PF < NC) DGDM_LOAD_B(c + PF)
    __builtin_amdgcn_sched_barrier(0);      // the loads stay HERE, PF chunks ahead of their use (left alone, hipcc sinks each next to its MFMA)
    f16x8 ah[2], al[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      ah[j] = *reinterpret_cast<const f16x8*>(arow + (32 * c + 8 * j) * 2);
      al[j] = *reinterpret_cast<const f16x8*>(arow + 2 * K + (32 * c + 8 * j) * 2);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (DGDM_STEP_DIAG & 4) { acc[t][0] += (float)al[j][0] * (float)bq[c % RING][t][j][0][0] + (float)ah[j][1] * (float)bq[c % RING][t][j][1][1]; continue; }
        acc[t] = mfma_hf(al[j], bq[c % RING][t][j][0], acc[t]);      // smaller terms first
        acc[t] = mfma_hf(ah[j], bq[c % RING][t][j][1], acc[t]);
        acc[t] = mfma_hf(ah[j], bq[c % RING][t][j][0], acc[t]);
      }
  }
#undef DGDM_LOAD_B
}

// accumulators (x inv + bias) -> fp32 [row][col] in LDS (the pre-norm activations of the layer)
template <int NT, int PITCH>
__device__ __forceinline__ void store_prenorm(const f32x16 (&acc)[NT], char* __restrict__ smem, int tile0, float inv, const float (&bias)[NT], int lane) {
  const int jc = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = 32 * (tile0 + t) + jc;
    const float bv = bias[t];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      *reinterpret_cast<float*>(smem + ((r & 3) + 8 * (r >> 2) + 4 * hi) * PITCH + col * 4) = fmaf(acc[t][r], inv, bv);
  }
}

// thread (row = tid >> 3, group = tid & 7): GroupNorm(8) over its W = width / 8 channels of the row, SiLU, then -- after everybody has
// read -- the hi / lo planes of the whole row over the same bytes (hi at 0, lo at 2 * width).
// The 8 threads of a row start a multiple of 128 bytes apart: read or written in the same order they would all sit on one 16-byte slot of
// the LDS (a 16-byte store is served 8 consecutive lanes at a time: 8-way; a 16-byte read 4-way).  Thread g therefore walks its 8-channel
// pieces ROTATED by rot(g): the stores of a row's 8 threads fall on 8 different slots, the reads are at worst 2-way.
template <int WIDTH, int PITCH>
__device__ __forceinline__ void group_norm_silu(char* __restrict__ smem, const float* gamma, const float* beta, float eps, int tid) {      // gamma / beta: LDS copies
  constexpr int W = WIDTH / 8, NP = W / 8;                // NP pieces of 8 channels
  const int row = tid >> 3, g = tid & 7;
  const int rot = W == 32 ? (g >> 1) : g;                 // W = 32: threads g and g + 1 are 64 bytes apart already
  float v[W];
  const float* src = reinterpret_cast<const float*>(smem + row * PITCH) + g * W;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int pc = (j + rot) & (NP - 1);
    const float4 q0 = *reinterpret_cast<const float4*>(src + 8 * pc), q1 = *reinterpret_cast<const float4*>(src + 8 * pc + 4);
    v[8 * j] = q0.x; v[8 * j + 1] = q0.y; v[8 * j + 2] = q0.z; v[8 * j + 3] = q0.w;
    v[8 * j + 4] = q1.x; v[8 * j + 5] = q1.y; v[8 * j + 6] = q1.z; v[8 * j + 7] = q1.w;
    s += ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w));
  }
  const float mean = s * (1.0f / W);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < W; ++i) { const float d = v[i] - mean; ss = fmaf(d, d, ss); }
  const float rstd = rsqrtf(ss * (1.0f / W) + eps);      // biased variance, as nn.GroupNorm
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int pc = (j + rot) & (NP - 1);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + g * W + 8 * pc), gb = *reinterpret_cast<const float4*>(gamma + g * W + 8 * pc + 4);
    const float4 ba = *reinterpret_cast<const float4*>(beta + g * W + 8 * pc), bb = *reinterpret_cast<const float4*>(beta + g * W + 8 * pc + 4);
    const float gm[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w}, bt[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) v[8 * j + k] = (DGDM_STEP_DIAG & 2) ? v[8 * j + k] * rstd : silu_f(fmaf((v[8 * j + k] - mean) * rstd, gm[k], bt[k]));
  }
  __syncthreads();                                       // every thread has its fp32 values: the row's bytes may be overwritten
  char* dst = smem + row * PITCH + g * W * 2;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int pc = (j + rot) & (NP - 1);
    uint4 h, l;
    split_pair(v[8 * j], v[8 * j + 1], &h.x, &l.x);
    split_pair(v[8 * j + 2], v[8 * j + 3], &h.y, &l.y);
    split_pair(v[8 * j + 4], v[8 * j + 5], &h.z, &l.z);
    split_pair(v[8 * j + 6], v[8 * j + 7], &h.w, &l.w);
    *reinterpret_cast<uint4*>(dst + pc * 16) = h;
    *reinterpret_cast<uint4*>(dst + 2 * WIDTH + pc * 16) = l;
  }
}

template <int C>
__global__ __launch_bounds__(256, C <= 128 ? 2 : 1) void k_denoise_ddpm_step(const StepArgs a) {
  constexpr int N1 = 4 * C, N2 = 2 * C, N3 = C;
  constexpr int PITCH = N1 * 4 + 16;                     // bytes per LDS row (+ 16: the 32 rows of an A read start in different banks)
  constexpr int T1 = N1 / 128, T2 = N2 / 128, T3 = N3 / 128;      // 32-column tiles per wave
  constexpr int RINGT = 8;                                            // fragment sets (of one tile) the B ring holds: 128 registers
  extern __shared__ __attribute__((aligned(16))) char smem[];      // 32 * PITCH, then gamma1 | beta1 | gamma2 | beta2
  __shared__ unsigned amax_sm[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * 32;
  // Everything small the step needs from global memory is asked for HERE, before the first dependent wait: a workgroup is one chain of
  // dependent phases (a lone workgroup took 25 us when every phase fetched its own parameters: tools/microbench_sample_step.py), and
  // only two of them share a CU.  Image scales and the lane's bias columns into registers, the GroupNorm parameters into LDS.
  float* gpar = reinterpret_cast<float*>(smem + 32 * PITCH);
  for (int i = tid; i < N1; i += 256) { gpar[i] = a.g1[i]; gpar[N1 + i] = a.be1[i]; }
  for (int i = tid; i < N2; i += 256) { gpar[2 * N1 + i] = a.g2[i]; gpar[2 * N1 + N2 + i] = a.be2[i]; }
  const float scb0 = *reinterpret_cast<const float*>(a.img0), scb1 = *reinterpret_cast<const float*>(a.img1), scb2 = *reinterpret_cast<const float*>(a.img2);
  float bia0[T1], bia1[T2], bia2[T3];
#pragma unroll
  for (int t = 0; t < T1; ++t) bia0[t] = a.bias0[32 * (wave * T1 + t) + (lane & 31)];
#pragma unroll
  for (int t = 0; t < T2; ++t) bia1[t] = a.bias1[32 * (wave * T2 + t) + (lane & 31)];
#pragma unroll
  for (int t = 0; t < T3; ++t) bia2[t] = a.bias2[32 * (wave * T3 + t) + (lane & 31)];

  // ---- stage x: thread (row, segment of C / 8 floats); its maximum decides the workgroup's power-of-two scale
  {
    constexpr int SEG = C / 8;
    const int r = tid >> 3, sgm = tid & 7;
    const float* src = a.x + (int64_t)min(row0 + r, a.N - 1) * a.ldx + sgm * SEG;
    float v[SEG];
    unsigned am = 0;
#pragma unroll
    for (int i = 0; i < SEG; i += 4) {
      const float4 q = *reinterpret_cast<const float4*>(src + i);
      v[i] = q.x; v[i + 1] = q.y; v[i + 2] = q.z; v[i + 3] = q.w;
      am = dgdm_amax4(am, q);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = max(am, (unsigned)__shfl_xor((int)am, o, 64));
    if (lane == 0) amax_sm[wave] = am;
    __syncthreads();
    const float sca = scale_of(max(max(amax_sm[0], amax_sm[1]), max(amax_sm[2], amax_sm[3])));
    char* dst = smem + r * PITCH + sgm * SEG * 2;
#pragma unroll
    for (int i = 0; i < SEG; i += 8) {
      uint4 h, l;
      split_pair(v[i] * sca, v[i + 1] * sca, &h.x, &l.x);
      split_pair(v[i + 2] * sca, v[i + 3] * sca, &h.y, &l.y);
      split_pair(v[i + 4] * sca, v[i + 5] * sca, &h.z, &l.z);
      split_pair(v[i + 6] * sca, v[i + 7] * sca, &h.w, &l.w);
      *reinterpret_cast<uint4*>(dst + i * 2) = h;
      *reinterpret_cast<uint4*>(dst + 2 * C + i * 2) = l;
    }
  }
  __syncthreads();
  const float sca = scale_of(max(max(amax_sm[0], amax_sm[1]), max(amax_sm[2], amax_sm[3])));

  // ---- layer 1: [32, C] . W0x^T -> [32, 4C]
  {
    f32x16 acc[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    layer_mfma<C, T1, PITCH, RINGT / T1 - 1>(acc, smem, a.img0, a.tiles0, wave * T1, lane);
    __syncthreads();                                     // everybody is done reading the x planes
    store_prenorm<T1, PITCH>(acc, smem, wave * T1, (1.0f / sca) * (1.0f / scb0), bia0, lane);
  }
  __syncthreads();
  group_norm_silu<N1, PITCH>(smem, gpar, gpar + N1, a.eps1, tid);
  __syncthreads();

  // ---- layer 2: [32, 4C] . W1^T -> [32, 2C]
  {
    f32x16 acc[T2];
#pragma unroll
    for (int t = 0; t < T2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    layer_mfma<N1, T2, PITCH, RINGT / T2 - 1>(acc, smem, a.img1, a.tiles1, wave * T2, lane);
    __syncthreads();
    store_prenorm<T2, PITCH>(acc, smem, wave * T2, 1.0f / scb1, bia1, lane);
  }
  __syncthreads();
  group_norm_silu<N2, PITCH>(smem, gpar + 2 * N1, gpar + 2 * N1 + N2, a.eps2, tid);
  __syncthreads();

  // ---- layer 3: [32, 2C] . W2^T -> eps [32, C], then the DDPM update on the accumulators' own (row, column) layout; the update's x and z
  // values are asked for BEFORE the layer's MFMAs (they arrive under them)
  {
    const int jc = lane & 31, hi = lane >> 5;
    float xv[T3][16], zv[T3][16];
#pragma unroll
    for (int t = 0; t < T3; ++t) {
      const int col = 32 * (wave * T3 + t) + jc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * hi, a.N - 1);
        xv[t][r] = a.x[(int64_t)row * a.ldx + col];
        zv[t][r] = a.last ? 0.f : a.z[(int64_t)row * a.ldz + col];
      }
    }
    f32x16 acc[T3];
#pragma unroll
    for (int t = 0; t < T3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    layer_mfma<N2, T3, PITCH, RINGT / T3 - 1>(acc, smem, a.img2, a.tiles2, wave * T3, lane);
    const float inv = 1.0f / scb2;
#pragma unroll
    for (int t = 0; t < T3; ++t) {
      const int col = 32 * (wave * T3 + t) + jc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < a.N) {
          const float e = fmaf(acc[t][r], inv, bia2[t]);
          float o = (xv[t][r] - a.s1mac * e) / a.sac;      // the reference's order of operations (diffusion.py:255-273)
          if (!a.last) o = a.salpha * o + a.svar * zv[t][r];
          a.out[(int64_t)row * a.ldo + col] = o;
        }
      }
    }
  }
}

// one status per kernel (the instances have the same function type: the instance is the template argument)
template <void (*KERN)(const StepArgs)>
int allow_lds(int bytes) {
  static int status = 1;
  if (status == 1)
    status = hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? DGDM_OK
                                                                                                                                     : DGDM_ERR_LAUNCH;
  return status;
}

}  // namespace

extern "C" int32_t dgdm_denoise_ddpm_step_supported(int32_t C) { return C == 128 || C == 256; }

extern "C" int dgdm_denoise_ddpm_step(const float* x, int64_t ldx, const float* z, int64_t ldz, int32_t N, int32_t C, const void* img0, int32_t tiles0,
                                      const void* img1, int32_t tiles1, const void* img2, int32_t tiles2, const float* bias0, const float* gn1_w,
                                      const float* gn1_b, float eps1, const float* bias1, const float* gn2_w, const float* gn2_b, float eps2,
                                      const float* bias2, float sqrt_one_minus_ac, float sqrt_ac, float sqrt_alpha, float sqrt_var, int32_t last,
                                      float* out, int64_t ldo, void* stream) {
  if (N < 0 || C <= 0) return DGDM_ERR_INVALID_ARG;
  if (N == 0) return DGDM_OK;
  if (!x || !out || !img0 || !img1 || !img2 || !bias0 || !gn1_w || !gn1_b || !bias1 || !gn2_w || !gn2_b || !bias2 || (!last && !z))
    return DGDM_ERR_INVALID_ARG;
  if (!dgdm_denoise_ddpm_step_supported(C)) return DGDM_ERR_UNSUPPORTED;
  if (tiles0 != (4 * C) / 32 || tiles1 != (2 * C) / 32 || tiles2 != C / 32) return DGDM_ERR_INVALID_ARG;      // images of [4C, C], [2C, 4C], [C, 2C]
  if ((ldx & 3) || ldx < C || ldo < C || (z && ldz < C) || !dgdm_aligned16(x) || !dgdm_aligned16(img0) || !dgdm_aligned16(img1) || !dgdm_aligned16(img2))
    return DGDM_ERR_UNSUPPORTED;
  StepArgs a{x, ldx, z, ldz, out, ldo, N, static_cast<const char*>(img0), tiles0, static_cast<const char*>(img1), tiles1,
             static_cast<const char*>(img2), tiles2, bias0, gn1_w, gn1_b, eps1, bias1, gn2_w, gn2_b, eps2, bias2,
             sqrt_one_minus_ac, sqrt_ac, sqrt_alpha, sqrt_var, last};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((N + 31) / 32);
  if (C == 128) {
    constexpr int LDS = 32 * (4 * 128 * 4 + 16) + 2 * (4 * 128 + 2 * 128) * 4;
    if (allow_lds<k_denoise_ddpm_step<128>>(LDS) != DGDM_OK) return DGDM_ERR_LAUNCH;
    hipLaunchKernelGGL(k_denoise_ddpm_step<128>, grid, dim3(256), LDS, s, a);
  } else {
    constexpr int LDS = 32 * (4 * 256 * 4 + 16) + 2 * (4 * 256 + 2 * 256) * 4;
    if (allow_lds<k_denoise_ddpm_step<256>>(LDS) != DGDM_OK) return DGDM_ERR_LAUNCH;
    hipLaunchKernelGGL(k_denoise_ddpm_step<256>, grid, dim3(256), LDS, s, a);
  }
  return dgdm_launch_status();
}
