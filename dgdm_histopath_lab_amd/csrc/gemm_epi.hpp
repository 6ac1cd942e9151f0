// Device code of the weight-image GEMMs (gemm_img.hip): image constants, operand scales and splits, the
// staging canary, and the fused epilogues.
#pragma once
#include "common.hpp"
#include "rowmath.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum { EPI_NONE = 0, EPI_ACT = 1, EPI_ACTBWD = 2, EPI_NORM = 3 };

struct EpiArgs {
  const float* pre_in;              // ACTBWD: pre-activation [M, ncols], leading dimension ldp
  float* pre_out;                   // ACT: nullable; receives A.B + bias (ldp)
  int64_t ldp;
  const float* res; int64_t ldr;    // NORM: nullable residual [M, ncols] -- or, with res_ptr, one row per SEGMENT of rows
  const int* res_ptr; int res_segments;   // NORM: nullable offsets [res_segments + 1]; row r takes res row g, res_ptr[g] <= r < res_ptr[g+1]
  float pre_drop_p; DgdmSeed pre_seed;    // NORM: dropout of A.B + bias BEFORE the residual is added (core/attention.py:176-181)
  const float* gamma; const float* beta;
  float* sum_out; int64_t lds;      // NORM: nullable; receives A.B + bias + res (what the norm's backward reads as its input)
  float* mean; float* rstd;         // NORM: [M * G]
  float eps; int L;                 // NORM: channels per group, a multiple of 32 that divides the wave's column count
  int act; float drop_p; DgdmSeed seed;
  unsigned* amax_out;               // nullable: slot group that receives max|C|
};

// (round 5's weight-stationary persistent kernel for the narrow layers -- measured slower at every shape, never compiled into the
// shipped library -- left the product tree in round 6: tools/attic/gemm_ws.hip, with the measurements in DESIGN.md section 4)

namespace {

constexpr int IMG_HDR = 256;     // bytes in front of an image's blocks (float scale_b)
constexpr int BLK = 4096;        // one image block: 32 columns x 32 k, fp16 hi + lo in MFMA fragment order

__device__ __forceinline__ unsigned amax_group(const unsigned* __restrict__ g) {
  const int lane = threadIdx.x & 63;
  unsigned m = lane < DGDM_AMAX_WAYS ? g[lane * DGDM_AMAX_STRIDE] : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  return m;
}

// 2^e with amax * 2^e in [2^14, 2^15) from the float bits of amax; 1 for a zero / denormal maximum (gemm_h.hip's rule)
__device__ __forceinline__ float scale_of(unsigned amax_bits) {
  const int E = (int)((amax_bits >> 23) & 0xffu) - 127;
  int e = (amax_bits & 0x7f800000u) ? 14 - E : 0;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return __uint_as_float((unsigned)(127 + e) << 23);
}

__device__ __forceinline__ void split_pair(float a, float b, unsigned* h, unsigned* l) {
  const f16x2 hh = __builtin_convertvector(f32x2{a, b}, f16x2);
  const unsigned hb = __builtin_bit_cast(unsigned, hh);
  // lo = fp16(x - hi) by one mixed-precision FMA per value (reads hi as fp16, x as fp32; x - hi is exact in fp32, so the result is
  // bit for bit the two-step form's): 3 instructions per pair where hipcc's lowering of the plain expression takes 5
  unsigned lb;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(lb)
      : "v"(hb), "v"(a), "v"(b));
  *h = hb;
  *l = lb;
}

// Staging canary (tests/test_hip_gemm_img.py, the lib/canary build: -DDGDM_STAGE_CANARY).  The activation loads below are inline asm
// retired by ONE hand-placed s_waitcnt per stage; nothing but the kernel's own text order keeps a consumer behind that wait.  In the
// canary build every destination register holds a NaN when its load is issued (the registers become in/out operands of the asm, so
// the compiler has to materialise the poison in front of it): a consumer that runs before the wait -- a scheduler that moved it, a
// copy the allocator slipped in -- multiplies NaNs into the result instead of silently using the previous stage's numbers.
#ifdef DGDM_STAGE_CANARY
#define DGDM_CANARY_OUT(r_) "+v"(r_)
#define DGDM_CANARY_POISON(a_, b_, c_, d_)                                                                          \
  {                                                                                                                 \
    const float nan__ = __builtin_nanf("");                                                                         \
    typedef float canary_f32x4 __attribute__((ext_vector_type(4)));                                                 \
    a_ = b_ = c_ = d_ = canary_f32x4{nan__, nan__, nan__, nan__};                                                    \
  }
#else
#define DGDM_CANARY_OUT(r_) "=&v"(r_)
#define DGDM_CANARY_POISON(a_, b_, c_, d_)
#endif

__device__ __forceinline__ f32x16 mfma_hf(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <bool TR>
__device__ __forceinline__ f32x16 mfma_o(f16x8 a, f16x8 b, f32x16 c) {
  return TR ? mfma_hf(b, a, c) : mfma_hf(a, b, c);
}

// ---- fused epilogues (round 5).  A GEMM whose result goes through an element-wise or row-wise layer before anything else reads
// it finishes that layer in its own registers:
//   EPI_ACT     C = dropout(act(A.B + bias)), the pre-activation optionally stored beside it (the backward needs it)
//               core/graph_layers.py:233-239  h = dropout(GELU(conv(x)))
//   EPI_ACTBWD  C = (A.B) * act'(pre) * mask  -- the backward of that layer as the epilogue of the GEMM that produces its
//               incoming gradient (dh = do . W_o, or dh = (A_hat^T dpre) . W for the second convolution)
//   EPI_NORM    C = dropout(act(norm_G(A.B + bias [+ res]) * gamma + beta)), the pre-norm sum stored beside it
//               core/graph_layers.py:241-245  norm1(output_proj(h) + x);  models/encoders.py:267-269;  core/diffusion.py:94-102
// These variants run the MFMAs with the operands SWAPPED (weight fragment as A, activation fragment as B: the two fragments have
// the same lane layout), so the accumulators hold the TRANSPOSED tile: lane (l & 31) owns output row r0 + (l & 31), and its 16
// registers of tile t are the channels 32 t + 8 (r >> 2) + 4 (l >> 5) + (r & 3) -- runs of four consecutive channels.  That is
// the shape the row kernels work in: the dropout word of rowmath.hpp covers four consecutive elements (the mask is the SAME
// function of (seed, element index) as in k_act_dropout / k_rownorm, so the backward kernels and the parity tests regenerate
// it), bias / gamma / beta / pre / res are 16-byte accesses, a row's statistics are an in-lane sum plus ONE exchange with lane
// l ^ 32, and the stores are float4.
__device__ __forceinline__ float4 act4(int act, const float4 v) {
  switch (act) {
    case DGDM_ACT_GELU: return make_float4(act_f<DGDM_ACT_GELU>(v.x), act_f<DGDM_ACT_GELU>(v.y), act_f<DGDM_ACT_GELU>(v.z), act_f<DGDM_ACT_GELU>(v.w));
    case DGDM_ACT_RELU: return make_float4(act_f<DGDM_ACT_RELU>(v.x), act_f<DGDM_ACT_RELU>(v.y), act_f<DGDM_ACT_RELU>(v.z), act_f<DGDM_ACT_RELU>(v.w));
    case DGDM_ACT_SILU: return make_float4(act_f<DGDM_ACT_SILU>(v.x), act_f<DGDM_ACT_SILU>(v.y), act_f<DGDM_ACT_SILU>(v.z), act_f<DGDM_ACT_SILU>(v.w));
    case DGDM_ACT_ELU: return make_float4(act_f<DGDM_ACT_ELU>(v.x), act_f<DGDM_ACT_ELU>(v.y), act_f<DGDM_ACT_ELU>(v.z), act_f<DGDM_ACT_ELU>(v.w));
    default: return v;
  }
}
__device__ __forceinline__ float4 act_d4(int act, const float4 v) {
  switch (act) {
    case DGDM_ACT_GELU: return make_float4(act_df<DGDM_ACT_GELU>(v.x), act_df<DGDM_ACT_GELU>(v.y), act_df<DGDM_ACT_GELU>(v.z), act_df<DGDM_ACT_GELU>(v.w));
    case DGDM_ACT_RELU: return make_float4(act_df<DGDM_ACT_RELU>(v.x), act_df<DGDM_ACT_RELU>(v.y), act_df<DGDM_ACT_RELU>(v.z), act_df<DGDM_ACT_RELU>(v.w));
    case DGDM_ACT_SILU: return make_float4(act_df<DGDM_ACT_SILU>(v.x), act_df<DGDM_ACT_SILU>(v.y), act_df<DGDM_ACT_SILU>(v.z), act_df<DGDM_ACT_SILU>(v.w));
    case DGDM_ACT_ELU: return make_float4(act_df<DGDM_ACT_ELU>(v.x), act_df<DGDM_ACT_ELU>(v.y), act_df<DGDM_ACT_ELU>(v.z), act_df<DGDM_ACT_ELU>(v.w));
    default: return make_float4(1.f, 1.f, 1.f, 1.f);
  }
}

// predicated 16-byte load (written as a branch: `ok ? *p : zero` makes hipcc select between the ADDRESS and a private copy of the zero)
__device__ __forceinline__ float4 ld4_if(const float* p, bool ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) v = *reinterpret_cast<const float4*>(p);
  return v;
}

// Where an epilogue takes its per-column vectors (bias, gamma, beta) from.  EpiVecGlobal: 16-byte loads (L1 / L2 hits, but a round
// trip the wave waits for with nothing beside it).  EpiVecLanes: the weight-stationary kernel loads each vector ONCE per launch, four
// columns per lane (lane l holds columns 4 l .. 4 l + 3, up to 128 columns in lanes 0 .. 31), and an epilogue fetches the float4 of
// column c from lane c / 4 through the LDS crossbar (ds_bpermute): no memory access after the kernel's first microsecond.
struct EpiVecGlobal {
  const float* p; int n;
  __device__ __forceinline__ float4 get(int col) const { return ld4_if(p + col, p && col < n); }
};
struct EpiVecLanes {
  float4 v;
  __device__ __forceinline__ float4 get(int col) const {
    const int src = col >> 2;           // columns past the matrix were loaded as zeros (or belong to lanes >= 32: also zeros)
    return make_float4(__shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64), __shfl(v.w, src, 64));
  }
};

#define DGDM_Q4(a_, q_) make_float4((a_)[4 * (q_)], (a_)[4 * (q_) + 1], (a_)[4 * (q_) + 2], (a_)[4 * (q_) + 3])
#define DGDM_SETQ4(a_, q_, v_) { (a_)[4 * (q_)] = (v_).x; (a_)[4 * (q_) + 1] = (v_).y; (a_)[4 * (q_) + 2] = (v_).z; (a_)[4 * (q_) + 3] = (v_).w; }

// Epilogue of a wave that holds TRANSPOSED accumulators of NT column tiles starting at column `col0`; `row` = this lane's output
// row (may be >= M: nothing is stored for it), hi = lane >> 5; seed / pseed: the dropout sites' seed values (DgdmSeed::value(), read by
// the caller before its main loop: one memory round trip less behind it); `am` collects max|C| (the caller commits it: dgdm_amax_commit has a
// workgroup barrier inside).
//
// The per-element work (erf, the dropout word) is ROLLED: a real loop over pairs of tiles that always works on acc[0] and acc[1]
// and then moves the remaining tiles two places down (16 v_mov per tile and round -- a tenth of the arithmetic).  Fully
// unrolled, the wide kernel's 128 outputs per lane are 15 000 instructions whose interleaved erf chains push the allocator past
// 256 registers: these kernels must not spill at all (their activation loads are retired by hand, tests/test_abi.py).  Stores
// leave straight from the round's temporaries; the next round's arithmetic covers their flight.
template <int NT, int EPI, class VEC>
__device__ __forceinline__ void epilogue_tr(f32x16 (&acc)[NT], const float inv, const int row, const int M, const int col0, const int Ncols,
                                            const VEC& bias, const VEC& gamma, const VEC& beta, float* __restrict__ C, const int64_t ldc,
                                            const EpiArgs& e, const int hi, unsigned& am, const uint32_t seed, const uint32_t pseed) {
  static_assert(NT % 2 == 0, "tiles are taken in pairs");
  const bool rok = row < M;
  const int cl = col0 + 4 * hi;                      // this lane's channels: cl + 32 t + 8 q + (0..3)
  const uint32_t thresh = (uint32_t)(e.drop_p * 65536.0f);
  const float keep_scale = e.drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  const uint64_t e0 = (uint64_t)row * (uint64_t)Ncols;   // element index of (row, 0) in the contiguous [M, Ncols] output
#define DGDM_ROTATE2(arr_)                                                                                          \
  _Pragma("unroll") for (int i__ = 0; i__ + 2 < NT; ++i__) arr_[i__] = arr_[i__ + 2];

  if (EPI == EPI_NONE) {       // plain C = A.B + bias from transposed accumulators (the weight-stationary kernel): float4 stores
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = cl + 32 * t + 8 * q;
        const float4 b = bias.get(col);
        acc[t][4 * q] = fmaf(acc[t][4 * q], inv, b.x); acc[t][4 * q + 1] = fmaf(acc[t][4 * q + 1], inv, b.y);
        acc[t][4 * q + 2] = fmaf(acc[t][4 * q + 2], inv, b.z); acc[t][4 * q + 3] = fmaf(acc[t][4 * q + 3], inv, b.w);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = cl + 32 * t + 8 * q;
        if (rok && col < Ncols) *reinterpret_cast<float4*>(C + (int64_t)row * ldc + col) = DGDM_Q4(acc[t], q);
      }
  } else if (EPI == EPI_ACT) {
#pragma clang loop unroll(disable)
    for (int t0 = 0; t0 < NT; t0 += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = cl + 32 * (t0 + u) + 8 * q;
          const bool ok = rok && col < Ncols;
          const float4 b = bias.get(col);
          const float4 v = make_float4(fmaf(acc[u][4 * q], inv, b.x), fmaf(acc[u][4 * q + 1], inv, b.y), fmaf(acc[u][4 * q + 2], inv, b.z),
                                       fmaf(acc[u][4 * q + 3], inv, b.w));
          float4 o = act4(e.act, v);
          if (e.drop_p > 0.f) {
            const float4 m = dropout_scale4(seed, e0 + (uint64_t)col, thresh, keep_scale);
            o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
          }
          if (ok) {
            am = dgdm_amax4(am, o);
            if (e.pre_out) *reinterpret_cast<float4*>(e.pre_out + (int64_t)row * e.ldp + col) = v;
            *reinterpret_cast<float4*>(C + (int64_t)row * ldc + col) = o;
          }
        }
      DGDM_ROTATE2(acc)
    }
  } else if (EPI == EPI_ACTBWD) {
    // narrow kernel: the pre-activations of the next pair of tiles are on their way while this pair is finished; the wide kernel
    // (128 accumulators) has no registers for a second set and loads each pair at the top of its round
    constexpr bool AHEAD = NT <= 4;
    float4 p[2][4], pn[2][4];
#define DGDM_LOAD_PRE(dst_, t_)                                                                                     \
  _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                     \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                 \
      const int col = cl + 32 * ((t_) + u) + 8 * q;                                                                 \
      dst_[u][q] = ld4_if(e.pre_in + (int64_t)row * e.ldp + col, rok && col < Ncols);                               \
    }
    if (AHEAD) DGDM_LOAD_PRE(p, 0)
#pragma clang loop unroll(disable)
    for (int t0 = 0; t0 < NT; t0 += 2) {
      if (AHEAD) {
        DGDM_LOAD_PRE(pn, t0 + 2)                    // past the last tile: col >= Ncols or a re-read inside the row -- never used
      } else {
        DGDM_LOAD_PRE(p, t0)
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = cl + 32 * (t0 + u) + 8 * q;
          const float4 d = act_d4(e.act, p[u][q]);
          float4 o = make_float4(acc[u][4 * q] * inv * d.x, acc[u][4 * q + 1] * inv * d.y, acc[u][4 * q + 2] * inv * d.z,
                                 acc[u][4 * q + 3] * inv * d.w);
          if (e.drop_p > 0.f) {
            const float4 m = dropout_scale4(seed, e0 + (uint64_t)col, thresh, keep_scale);
            o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
          }
          if (rok && col < Ncols) {
            am = dgdm_amax4(am, o);
            *reinterpret_cast<float4*>(C + (int64_t)row * ldc + col) = o;
          }
        }
      DGDM_ROTATE2(acc)
      if (AHEAD) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) p[u][q] = pn[u][q];
      }
    }
#undef DGDM_LOAD_PRE
  } else if (EPI == EPI_NORM) {
    // pass 1 (unrolled: adds only): A.B + bias [+ res] in place, stored as the norm's input for the backward, per-tile sums
    const int tpg = e.L >> 5;
    const float invL = 1.0f / (float)e.L;
    float mu[NT], rs[NT];
    int64_t rrow = row;                               // the residual's row: this row, or the row of this row's segment
    if (e.res_ptr && rok) {
      int lo = 0, hi_ = e.res_segments - 1;           // last g with res_ptr[g] <= row
      while (lo < hi_) {
        const int mid = (lo + hi_ + 1) >> 1;
        if (e.res_ptr[mid] <= row) lo = mid; else hi_ = mid - 1;
      }
      rrow = lo;
    }
    const uint32_t pthresh = (uint32_t)(e.pre_drop_p * 65536.0f);
    const float pkeep = e.pre_drop_p > 0.f ? 1.0f / (1.0f - (float)pthresh / 65536.0f) : 1.0f;
    // (rolled like pass 2: two tiles per round, then every array moves two places down -- after NT / 2 rounds all are back in order)
#pragma unroll
    for (int t = 0; t < NT; ++t) mu[t] = 0.f;
#pragma clang loop unroll(disable)
    for (int t0 = 0; t0 < NT; t0 += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = cl + 32 * (t0 + u) + 8 * q;
          const bool ok = rok && col < Ncols;
          const float4 b = bias.get(col);
          const float4 r4 = ld4_if(e.res + rrow * e.ldr + col, e.res && ok);
          float4 v = make_float4(fmaf(acc[u][4 * q], inv, b.x), fmaf(acc[u][4 * q + 1], inv, b.y), fmaf(acc[u][4 * q + 2], inv, b.z),
                                 fmaf(acc[u][4 * q + 3], inv, b.w));
          if (e.pre_drop_p > 0.f) {
            const float4 m = dropout_scale4(pseed, e0 + (uint64_t)col, pthresh, pkeep);
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
          }
          v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
          if (col >= Ncols) v = make_float4(0.f, 0.f, 0.f, 0.f);      // columns past the matrix: nothing for the statistics
          DGDM_SETQ4(acc[u], q, v)
          if (e.sum_out && ok) *reinterpret_cast<float4*>(e.sum_out + (int64_t)row * e.lds + col) = v;
          s += (v.x + v.y) + (v.z + v.w);
        }
        mu[u] = s;
      }
      {   // a true rotation: the finished pair goes to the end (pass 2 needs every tile again)
        const f32x16 a0 = acc[0], a1 = acc[1];
        const float m0 = mu[0], m1 = mu[1];
        DGDM_ROTATE2(acc)
        DGDM_ROTATE2(mu)
        acc[NT - 2] = a0; acc[NT - 1] = a1; mu[NT - 2] = m0; mu[NT - 1] = m1;
      }
    }
    // statistics of the groups of tpg = L / 32 consecutive tiles: per-tile sums (in-lane + the other half-wave), then a tree over
    // the tiles of a group -- static register indices, wave-uniform conditions
#define DGDM_GROUP_TREE(v_)                                                                                         \
  {                                                                                                                 \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) v_[t] += __shfl_xor(v_[t], 32, 64);                              \
    _Pragma("unroll") for (int w = 1; w < NT; w <<= 1)                                                              \
      if (tpg > w) {                                                                                                \
        _Pragma("unroll") for (int t = 0; t < NT; t += 2 * w) {                                                     \
          const float s__ = v_[t] + v_[t + w];                                                                      \
          _Pragma("unroll") for (int u = 0; u < 2 * w; ++u) v_[t + u] = s__;                                        \
        }                                                                                                           \
      }                                                                                                             \
  }
    DGDM_GROUP_TREE(mu)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      mu[t] *= invL;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float a = acc[t][4 * q] - mu[t], b = acc[t][4 * q + 1] - mu[t], c = acc[t][4 * q + 2] - mu[t], d = acc[t][4 * q + 3] - mu[t];
        s += (a * a + b * b) + (c * c + d * d);
      }
      rs[t] = s;
    }
    DGDM_GROUP_TREE(rs)
#undef DGDM_GROUP_TREE
    const int G = Ncols / e.L;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      rs[t] = 1.0f / sqrtf(rs[t] * invL + e.eps);
      const int c0 = col0 + 32 * t;
      if (rok && hi == 0 && c0 < Ncols && (c0 % e.L) == 0) {
        e.mean[(int64_t)row * G + c0 / e.L] = mu[t];
        e.rstd[(int64_t)row * G + c0 / e.L] = rs[t];
      }
    }
    // pass 2 (rolled): normalise, affine, activation, dropout, store
#pragma clang loop unroll(disable)
    for (int t0 = 0; t0 < NT; t0 += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = cl + 32 * (t0 + u) + 8 * q;
          const bool cok = col < Ncols;
          const float4 g4 = gamma.get(col);
          const float4 b4 = beta.get(col);
          float4 o = make_float4((acc[u][4 * q] - mu[u]) * rs[u] * g4.x + b4.x, (acc[u][4 * q + 1] - mu[u]) * rs[u] * g4.y + b4.y,
                                 (acc[u][4 * q + 2] - mu[u]) * rs[u] * g4.z + b4.z, (acc[u][4 * q + 3] - mu[u]) * rs[u] * g4.w + b4.w);
          o = act4(e.act, o);
          if (e.drop_p > 0.f) {
            const float4 m = dropout_scale4(seed, e0 + (uint64_t)col, thresh, keep_scale);
            o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
          }
          if (rok && cok) {
            am = dgdm_amax4(am, o);
            *reinterpret_cast<float4*>(C + (int64_t)row * ldc + col) = o;
          }
        }
      DGDM_ROTATE2(acc)
      DGDM_ROTATE2(mu)
      DGDM_ROTATE2(rs)
    }
  }
#undef DGDM_ROTATE2
}


}  // namespace
