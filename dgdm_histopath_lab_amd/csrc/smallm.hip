// Dense layers with FEW rows (one row per graph of a batch, or per diffusion timestep): the time-embedding MLP of the
// denoiser (core/diffusion.py:87-91,147-163), the time half of its first Linear (the concat of :165-170 folded into a
// per-graph bias), the global-token query and the output projection of GlobalAttentionPool (models/dgdm_model.py:596-615).
// M is the batch size (4 at the headline config), K and N are 128..512: a few hundred KFLOP -- a tile GEMM would spend its
// launch filling a 128-row tile with zeros.  These kernels do the exact fp32 arithmetic on the VALU, one wave per output
// column, every reduction in a fixed order (bitwise reproducible); they replace the library GEMM calls (hipBLASLt through
// torch) the path used for fewer than 256 rows.
//
//   dgdm_linear_small_fwd : y[m, n]  = act(sum_k x[m, k] w[n, k] + b[n])        x [M, K], w [N, K] (row stride ldw), y [M, N]
//   dgdm_linear_small_bwd : dx[m, k] = sum_n gy[m, n] w[n, k]                    (gy already multiplied by act')
//                           dw[n, k] = sum_m gy[m, n] x[m, k], db[n] = sum_m gy[m, n]   (dw row stride lddw: a column
//                           block of a larger gradient can be written in place)
//   dgdm_ddpm_step        : one update of DiffusionLayer.sample (core/diffusion.py:252-273), see below
#include "common.hpp"
#include "rowmath.hpp"

namespace {

constexpr int SM_MAX_K = 2048;   // 32 floats of w per lane

// one wave per output column n: w[n, :] in registers (K/64 per lane), rows of x streamed (L2-resident: M*K floats)
template <int ACT>
__global__ __launch_bounds__(256) void k_linear_small_fwd(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                          int64_t ldw, const float* __restrict__ b, int M, int N, int K,
                                                          float* __restrict__ y, int64_t ldy, float* __restrict__ pre, int64_t ldp) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float wr[SM_MAX_K / 64];
  const int kc = (K + 63) / 64;
#pragma unroll
  for (int j = 0; j < SM_MAX_K / 64; ++j) {
    const int k = j * 64 + lane;
    wr[j] = (j < kc && k < K) ? w[(int64_t)n * ldw + k] : 0.f;
  }
  const float bias = b ? b[n] : 0.f;
  for (int m = 0; m < M; ++m) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < SM_MAX_K / 64; ++j) {
      const int k = j * 64 + lane;
      if (j < kc && k < K) acc = fmaf(x[(int64_t)m * ldx + k], wr[j], acc);
    }
    acc = wave_sum(acc) + bias;
    if (lane == 0) {
      if (pre) pre[(int64_t)m * ldp + n] = acc;
      y[(int64_t)m * ldy + n] = act_f<ACT>(acc);
    }
  }
}

// dx[m, k] = sum_n g[m, n] w[n, k]; g = gy * act'(pre) when pre is given.  Block = (64 columns k) x (4 slices of n): lane = column
// (w rows are read 256 B at a time), wave = n slice with four independent accumulators; the slices meet in LDS in a fixed order.
template <int ACT>
__global__ __launch_bounds__(256) void k_linear_small_dx(const float* __restrict__ gy, int64_t ldg, const float* __restrict__ pre,
                                                         int64_t ldp, const float* __restrict__ w, int64_t ldw, int N, int K,
                                                         float* __restrict__ dx, int64_t lddx) {
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane, m = blockIdx.y;
  const int per = (N + 3) / 4, n0 = part * per, n1 = min(N, n0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  auto gval = [&](int n) {
    float g = gy[(int64_t)m * ldg + n];
    if (ACT != DGDM_ACT_NONE) g *= act_df<ACT>(pre[(int64_t)m * ldp + n]);
    return g;
  };
  if (k < K) {
    int n = n0;
    for (; n + 3 < n1; n += 4) {
      a0 = fmaf(gval(n), w[(int64_t)n * ldw + k], a0);
      a1 = fmaf(gval(n + 1), w[(int64_t)(n + 1) * ldw + k], a1);
      a2 = fmaf(gval(n + 2), w[(int64_t)(n + 2) * ldw + k], a2);
      a3 = fmaf(gval(n + 3), w[(int64_t)(n + 3) * ldw + k], a3);
    }
    for (; n < n1; ++n) a0 = fmaf(gval(n), w[(int64_t)n * ldw + k], a0);
  }
  __shared__ float sm[4][64];
  sm[part][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (part == 0 && k < K) dx[(int64_t)m * lddx + k] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

// dw[n, k] = sum_m g[m, n] x[m, k] (thread = column k, block row = n; rows in index order); db[n] by the blocks with blockIdx.x == 0
template <int ACT>
__global__ __launch_bounds__(256) void k_linear_small_dw(const float* __restrict__ gy, int64_t ldg, const float* __restrict__ pre,
                                                         int64_t ldp, const float* __restrict__ x, int64_t ldx, int M, int K,
                                                         float* __restrict__ dw, int64_t lddw, float* __restrict__ db) {
  const int k = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  float acc = 0.f, bsum = 0.f;
  for (int m = 0; m < M; ++m) {
    float g = gy[(int64_t)m * ldg + n];
    if (ACT != DGDM_ACT_NONE) g *= act_df<ACT>(pre[(int64_t)m * ldp + n]);
    bsum += g;
    if (k < K) acc = fmaf(g, x[(int64_t)m * ldx + k], acc);
  }
  if (k < K && dw) dw[(int64_t)n * lddw + k] = acc;
  if (db && blockIdx.x == 0 && threadIdx.x == 0) db[n] = bsum;
}

// One step of DDPM ancestral sampling, elementwise, in the reference's order of operations (core/diffusion.py:255-273):
//   x0 = (x - sqrt(1 - ac) * eps) / sqrt(ac);   out = last ? x0 : sqrt(alpha) * x0 + sqrt(var) * z
__global__ __launch_bounds__(256) void k_ddpm_step(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ z,
                                                   int64_t n4, float s1mac, float sac, float salpha, float svar, int last,
                                                   float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 xv = reinterpret_cast<const float4*>(x)[i], ev = reinterpret_cast<const float4*>(eps)[i];
    float4 o = make_float4((xv.x - s1mac * ev.x) / sac, (xv.y - s1mac * ev.y) / sac, (xv.z - s1mac * ev.z) / sac, (xv.w - s1mac * ev.w) / sac);
    if (!last) {
      const float4 zv = reinterpret_cast<const float4*>(z)[i];
      o = make_float4(salpha * o.x + svar * zv.x, salpha * o.y + svar * zv.y, salpha * o.z + svar * zv.z, salpha * o.w + svar * zv.w);
    }
    reinterpret_cast<float4*>(out)[i] = o;
  }
}

inline bool small_ok(int M, int N, int K) { return M >= 0 && N > 0 && K > 0 && K <= SM_MAX_K && M <= 65535 && N <= 65535; }

}  // namespace

extern "C" int dgdm_linear_small_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* b, int32_t M, int32_t N,
                                     int32_t K, int32_t act, float* y, int64_t ldy, float* pre, int64_t ldp, void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || act < 0 || act > DGDM_ACT_ELU) return DGDM_ERR_INVALID_ARG;
  if (M == 0) return DGDM_OK;
  if (!x || !w || !y) return DGDM_ERR_INVALID_ARG;
  if (!small_ok(M, N, K) || ldx < K || ldw < K || ldy < N || (pre && ldp < N)) return DGDM_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((N + 3) / 4), block(256);
#define GO(A) hipLaunchKernelGGL((k_linear_small_fwd<A>), grid, block, 0, st, x, ldx, w, ldw, b, M, N, K, y, ldy, pre, ldp)
  switch (act) {
    case DGDM_ACT_GELU: GO(DGDM_ACT_GELU); break;
    case DGDM_ACT_RELU: GO(DGDM_ACT_RELU); break;
    case DGDM_ACT_SILU: GO(DGDM_ACT_SILU); break;
    case DGDM_ACT_ELU: GO(DGDM_ACT_ELU); break;
    default: GO(DGDM_ACT_NONE); break;
  }
#undef GO
  return dgdm_launch_status();
}

extern "C" int dgdm_linear_small_bwd(const float* gy, int64_t ldg, const float* pre, int64_t ldp, int32_t act, const float* x, int64_t ldx,
                                     const float* w, int64_t ldw, int32_t M, int32_t N, int32_t K, float* dx, int64_t lddx, float* dw,
                                     int64_t lddw, float* db, void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || act < 0 || act > DGDM_ACT_ELU) return DGDM_ERR_INVALID_ARG;
  if (!gy && M > 0) return DGDM_ERR_INVALID_ARG;
  if (act != DGDM_ACT_NONE && !pre && M > 0) return DGDM_ERR_INVALID_ARG;
  if ((dx && !w) || ((dw || db) && !x && M > 0)) return DGDM_ERR_INVALID_ARG;
  if (!small_ok(M, N, K) || ldg < N || (pre && ldp < N) || (dx && (lddx < K || ldw < K)) || (dw && (lddw < K || ldx < K)))
    return DGDM_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 block(256);
#define GO(A)                                                                                                                    \
  do {                                                                                                                           \
    if (dx && M > 0)                                                                                                             \
      hipLaunchKernelGGL((k_linear_small_dx<A>), dim3((K + 63) / 64, M), block, 0, st, gy, ldg, pre, ldp, w, ldw, N, K, dx, lddx); \
    if (dw || db)                                                                                                                \
      hipLaunchKernelGGL((k_linear_small_dw<A>), dim3((K + 255) / 256, N), block, 0, st, gy, ldg, pre, ldp, x, ldx, M, K, dw, lddw, db); \
  } while (0)
  switch (act) {
    case DGDM_ACT_GELU: GO(DGDM_ACT_GELU); break;
    case DGDM_ACT_RELU: GO(DGDM_ACT_RELU); break;
    case DGDM_ACT_SILU: GO(DGDM_ACT_SILU); break;
    case DGDM_ACT_ELU: GO(DGDM_ACT_ELU); break;
    default: GO(DGDM_ACT_NONE); break;
  }
#undef GO
  return dgdm_launch_status();
}

extern "C" int dgdm_ddpm_step(const float* x, const float* eps, const float* z, int64_t n, float sqrt_one_minus_ac, float sqrt_ac,
                              float sqrt_alpha, float sqrt_var, int32_t last, float* out, void* stream) {
  if (n < 0) return DGDM_ERR_INVALID_ARG;
  if (n == 0) return DGDM_OK;
  if (!x || !eps || !out || (!last && !z)) return DGDM_ERR_INVALID_ARG;
  if ((n & 3) || !dgdm_aligned16(x) || !dgdm_aligned16(eps) || !dgdm_aligned16(out) || (!last && !dgdm_aligned16(z))) return DGDM_ERR_UNSUPPORTED;
  if (!(sqrt_ac > 0.f)) return DGDM_ERR_INVALID_ARG;
  const int64_t n4 = n >> 2;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_ddpm_step, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, eps, z, n4, sqrt_one_minus_ac,
                     sqrt_ac, sqrt_alpha, sqrt_var, last, out);
  return dgdm_launch_status();
}
