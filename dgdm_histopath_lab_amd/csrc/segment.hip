// Per-graph (segment) primitives over the node dimension of a batch; graphs are contiguous row
// ranges [ptr[g], ptr[g+1]).  They replace the reference's Python loops over graphs with boolean
// masks (models/dgdm_model.py:419-431, 607-613) and the gather/scatter torch would run for a
// batched restatement (`per_graph[batch]` and its index_put backward: 7 ms/step measured).
//
//   dgdm_segment_bcast_add : out[n,:] = x[n,:] + src[seg(n),:]        (time-embedding bias, K7)
//   dgdm_segment_sum       : out[g,:] = sum_{n in g} x[n,:]           (its backward; fixed order)
//   dgdm_attn_pool_fwd/bwd : GlobalAttentionPool (K10): one learned query per graph attends over
//                            the graph's nodes, softmax over the segment, per head.
// All reductions are tree/two-stage reductions in a fixed order: no float atomics.
#include "common.hpp"
#include "rowmath.hpp"

namespace {

// ------------------------------------------------------------------ broadcast add
__global__ __launch_bounds__(256) void k_segment_bcast_add(const float* __restrict__ x, const float* __restrict__ src,
                                                           const int32_t* __restrict__ ptr, int B, int64_t n4total, int c4,
                                                           float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4total; i += stride) {
    const int n = (int)(i / c4), k = (int)(i % c4);
    int g = 0;
    while (g + 1 < B && ptr[g + 1] <= n) ++g;
    float4 v = reinterpret_cast<const float4*>(src)[(int64_t)g * c4 + k];
    if (x) { const float4 t = reinterpret_cast<const float4*>(x)[i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// ------------------------------------------------------------------ segment sum
// stage 1: block (chunk, g) sums rows [r0, r1) of graph g into part[g][chunk][C];
// thread = (float4 column k, row lane): 256 threads cover `rl = 256 / c4p` rows at a time.
__global__ __launch_bounds__(256) void k_segment_sum_stage1(const float* __restrict__ x, const int32_t* __restrict__ ptr, int c4,
                                                            int c4p, int nchunks, float* __restrict__ part) {
  const int g = blockIdx.y, chunk = blockIdx.x;
  const int a = ptr[g], b = ptr[g + 1];
  const int per = (b - a + nchunks - 1) / nchunks;
  const int r0 = a + chunk * per, r1 = min(b, r0 + per);
  const int k = threadIdx.x % c4p, rl = threadIdx.x / c4p, nrl = 256 / c4p;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < c4)
    for (int r = r0 + rl; r < r1; r += nrl) {
      const float4 t = reinterpret_cast<const float4*>(x)[(int64_t)r * c4 + k];
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
  __shared__ float4 sm[256];
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0 && k < c4) {
    for (int j = 1; j < nrl; ++j) {  // fixed order
      const float4 t = sm[j * c4p + k];
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    reinterpret_cast<float4*>(part)[((int64_t)g * nchunks + chunk) * c4 + k] = acc;
  }
}

__global__ __launch_bounds__(256) void k_segment_sum_stage2(const float* __restrict__ part, int c4, int nchunks, int64_t total4,
                                                            float* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // over B * c4
  if (i >= total4) return;
  const int64_t g = i / c4;
  const int k = (int)(i % c4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* p = reinterpret_cast<const float4*>(part) + g * nchunks * c4 + k;
  int c = 0;
  for (; c + 16 <= nchunks; c += 16) {   // 16 loads in flight, added in chunk order (the sum keeps its fixed order): the plain loop
    float4 t[16];                        // waited for every load before issuing the next -- 64 round trips, 17 us for 8 KB of output
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = p[(int64_t)(c + u) * c4];
#pragma unroll
    for (int u = 0; u < 16; ++u) { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
  }
  for (; c < nchunks; ++c) {
    const float4 t = p[(int64_t)c * c4];
    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
  }
  reinterpret_cast<float4*>(out)[i] = acc;
}

constexpr int SEG_CHUNKS = 64;

// ------------------------------------------------------------------ attention pooling
template <int D>
__device__ __forceinline__ float dot_row(const float* __restrict__ row, const float* __restrict__ q) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < D; i += 4) {
    const float4 t = *reinterpret_cast<const float4*>(row + i);
    s = fmaf(t.x, q[i], s); s = fmaf(t.y, q[i + 1], s); s = fmaf(t.z, q[i + 2], s); s = fmaf(t.w, q[i + 3], s);
  }
  return s;
}

__device__ __forceinline__ float block_max(float v, float* sm) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  __syncthreads();
  return v;
}
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  v = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  __syncthreads();
  return v;
}

__device__ __forceinline__ float drop_factor(uint32_t seed, uint64_t e, uint32_t thresh, float keep_scale) {
  const uint32_t a = hash32((uint32_t)e * 0x9E3779B1U ^ seed ^ ((uint32_t)(e >> 32) * 0x85EBCA6BU));
  return (a & 0xFFFFu) >= thresh ? keep_scale : 0.f;
}

// Forward in two launches over (head, graph, chunk of POOL_ROWS nodes) instead of one block per (head, graph) walking
// 10k nodes three times: (1) per chunk: scores, chunk maximum m_c, e = exp(s - m_c) (left in P), l_c = sum e,
// acc_c = sum e * dropout * V  -> part[g][h][c] = {m_c, l_c, acc_c[D]};  (2) every block combines the chunk records of
// its (graph, head) in index order (m, l), rescales its chunk of P to exp(s - m) / l, and chunk 0 writes the pooled row.
// qs = q_proj(global_token) * 1/sqrt(d)  [H*D].
constexpr int POOL_ROWS = 512;   // 2 nodes per thread

template <int D>
__global__ __launch_bounds__(256) void k_attn_pool_part(const float* __restrict__ K, const float* __restrict__ V, int64_t ld,
                                                        const float* __restrict__ qs, const int32_t* __restrict__ ptr, int H,
                                                        float drop_p, DgdmSeed seed_in, float* __restrict__ P,
                                                        float* __restrict__ part) {
  const uint32_t seed = seed_in.value();
  const int h = blockIdx.x, g = blockIdx.y, c = blockIdx.z;
  const int a = ptr[g] + c * POOL_ROWS, b = min(ptr[g + 1], a + POOL_ROWS);
  __shared__ float red[4];
  __shared__ float accs[4][D];
  float* rec = part + (((int64_t)g * H + h) * gridDim.z + c) * (D + 2);
  if (a >= b) {   // chunk beyond this graph: neutral record
    if (threadIdx.x == 0) { rec[0] = -INFINITY; rec[1] = 0.f; }
    if (threadIdx.x < D) rec[2 + threadIdx.x] = 0.f;
    return;
  }
  float q[D];
#pragma unroll
  for (int i = 0; i < D; ++i) q[i] = qs[h * D + i];
  float sc[2];
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int n = a + threadIdx.x + 256 * r;
    sc[r] = n < b ? dot_row<D>(K + (int64_t)n * ld + h * D, q) : -INFINITY;
    m = fmaxf(m, sc[r]);
  }
  m = block_max(m, red);
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  float l = 0.f, acc[D];
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] = 0.f;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int n = a + threadIdx.x + 256 * r;
    if (n < b) {
      const float e = __expf(sc[r] - m);
      l += e;
      P[(int64_t)n * H + h] = e;
      const float em = drop_p > 0.f ? e * drop_factor(seed, (uint64_t)n * H + h, thresh, keep_scale) : e;
      const float* vr = V + (int64_t)n * ld + h * D;
#pragma unroll
      for (int i = 0; i < D; i += 4) {
        const float4 t = *reinterpret_cast<const float4*>(vr + i);
        acc[i] = fmaf(em, t.x, acc[i]); acc[i + 1] = fmaf(em, t.y, acc[i + 1]);
        acc[i + 2] = fmaf(em, t.z, acc[i + 2]); acc[i + 3] = fmaf(em, t.w, acc[i + 3]);
      }
    }
  }
  l = block_sum(l, red);
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const float s = wave_sum(acc[i]);
    if ((threadIdx.x & 63) == 0) accs[threadIdx.x >> 6][i] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) { rec[0] = m; rec[1] = l; }
  if (threadIdx.x < D) rec[2 + threadIdx.x] = (accs[0][threadIdx.x] + accs[1][threadIdx.x]) + (accs[2][threadIdx.x] + accs[3][threadIdx.x]);
}

template <int D>
__global__ __launch_bounds__(256) void k_attn_pool_combine(const float* __restrict__ part, const int32_t* __restrict__ ptr, int H,
                                                           float* __restrict__ P, float* __restrict__ out) {
  const int h = blockIdx.x, g = blockIdx.y, c = blockIdx.z, nc = gridDim.z;
  const float* recs = part + ((int64_t)g * H + h) * nc * (D + 2);
  float m = -INFINITY;
  for (int k = 0; k < nc; ++k) m = fmaxf(m, recs[k * (D + 2)]);
  float l = 0.f;
  for (int k = 0; k < nc; ++k) {   // every thread, same order: same value everywhere
    const float mk = recs[k * (D + 2)];
    l += mk > -INFINITY ? recs[k * (D + 2) + 1] * __expf(mk - m) : 0.f;
  }
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  const int a = ptr[g] + c * POOL_ROWS, b = min(ptr[g + 1], a + POOL_ROWS);
  if (a < b) {
    const float scale = __expf(recs[c * (D + 2)] - m) * inv;
    for (int n = a + threadIdx.x; n < b; n += 256) P[(int64_t)n * H + h] *= scale;
  }
  if (c == 0 && threadIdx.x < D) {
    float acc = 0.f;
    for (int k = 0; k < nc; ++k) {
      const float mk = recs[k * (D + 2)];
      acc += mk > -INFINITY ? recs[k * (D + 2) + 2 + threadIdx.x] * __expf(mk - m) : 0.f;
    }
    out[((int64_t)g * H + h) * D + threadIdx.x] = acc * inv;
  }
}

// grid (H, B).  dK = dS * qs, dV = Pm * dOut, dqs_part[g][h][:] = sum_n dS_n K_n.
template <int D>
__global__ __launch_bounds__(256) void k_attn_pool_bwd(const float* __restrict__ K, const float* __restrict__ V, int64_t ld,
                                                       const float* __restrict__ qs, const int32_t* __restrict__ ptr, int H,
                                                       float drop_p, DgdmSeed seed_in, const float* __restrict__ P,
                                                       const float* __restrict__ out, const float* __restrict__ dout,
                                                       float* __restrict__ dK, float* __restrict__ dV, int64_t ldg,
                                                       float* __restrict__ dqs_part) {
  const uint32_t seed = seed_in.value();
  const int h = blockIdx.x, g = blockIdx.y;
  const int a = ptr[g], b = ptr[g + 1];
  __shared__ float accs[4][D];
  float q[D], go[D];
  float delta = 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    q[i] = qs[h * D + i];
    go[i] = dout[((int64_t)g * H + h) * D + i];
    delta = fmaf(go[i], out[((int64_t)g * H + h) * D + i], delta);
  }
  const uint32_t thresh = (uint32_t)(drop_p * 65536.0f);
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - (float)thresh / 65536.0f) : 1.0f;
  float dq[D];
#pragma unroll
  for (int i = 0; i < D; ++i) dq[i] = 0.f;
  for (int n = a + threadIdx.x; n < b; n += 256) {
    const float p = P[(int64_t)n * H + h];
    const float f = drop_p > 0.f ? drop_factor(seed, (uint64_t)n * H + h, thresh, keep_scale) : 1.0f;
    const float ds = p * (f * dot_row<D>(V + (int64_t)n * ld + h * D, go) - delta);
    const float pm = p * f;
    const float* kr = K + (int64_t)n * ld + h * D;
    float* dkr = dK + (int64_t)n * ldg + h * D;
    float* dvr = dV + (int64_t)n * ldg + h * D;
#pragma unroll
    for (int i = 0; i < D; i += 4) {
      const float4 t = *reinterpret_cast<const float4*>(kr + i);
      dq[i] = fmaf(ds, t.x, dq[i]); dq[i + 1] = fmaf(ds, t.y, dq[i + 1]);
      dq[i + 2] = fmaf(ds, t.z, dq[i + 2]); dq[i + 3] = fmaf(ds, t.w, dq[i + 3]);
      *reinterpret_cast<float4*>(dkr + i) = make_float4(ds * q[i], ds * q[i + 1], ds * q[i + 2], ds * q[i + 3]);
      *reinterpret_cast<float4*>(dvr + i) = make_float4(pm * go[i], pm * go[i + 1], pm * go[i + 2], pm * go[i + 3]);
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const float s = wave_sum(dq[i]);
    if ((threadIdx.x & 63) == 0) accs[threadIdx.x >> 6][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < D)
    dqs_part[((int64_t)g * H + h) * D + threadIdx.x] =
        (accs[0][threadIdx.x] + accs[1][threadIdx.x]) + (accs[2][threadIdx.x] + accs[3][threadIdx.x]);
}

}  // namespace

extern "C" int dgdm_segment_bcast_add(const float* x, const float* src, const int32_t* ptr, int32_t B, int32_t N, int32_t C,
                                      float* out, void* stream) {
  DGDM_REQUIRE(B >= 0 && N >= 0 && C > 0);
  if (N == 0 || B == 0) return DGDM_OK;
  DGDM_REQUIRE(src && ptr && out);
  if ((C & 3) || !dgdm_aligned16(src) || !dgdm_aligned16(out) || (x && !dgdm_aligned16(x))) return DGDM_ERR_UNSUPPORTED;
  const int64_t n4 = (int64_t)N * (C >> 2);
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_segment_bcast_add, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, src, ptr, B, n4,
                     C >> 2, out);
  return dgdm_launch_status();
}

extern "C" size_t dgdm_segment_sum_workspace_bytes(int32_t B, int32_t C) {
  return (B <= 0 || C <= 0) ? 0 : (size_t)B * SEG_CHUNKS * C * sizeof(float);
}

extern "C" int dgdm_segment_sum(const float* x, const int32_t* ptr, int32_t B, int32_t C, float* out, void* workspace,
                                size_t workspace_bytes, void* stream_) {
  DGDM_REQUIRE(B >= 0 && C > 0);
  if (B == 0) return DGDM_OK;
  DGDM_REQUIRE(x && ptr && out && workspace);
  if ((C & 3) || C > 1024 || !dgdm_aligned16(x) || !dgdm_aligned16(out) || !dgdm_aligned16(workspace)) return DGDM_ERR_UNSUPPORTED;
  if (workspace_bytes < dgdm_segment_sum_workspace_bytes(B, C)) return DGDM_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const int c4 = C >> 2;
  int c4p = 1;
  while (c4p < c4) c4p <<= 1;  // threads per row: power of two >= c4 (<= 256)
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(k_segment_sum_stage1, dim3(SEG_CHUNKS, B), dim3(256), 0, s, x, ptr, c4, c4p, SEG_CHUNKS, part);
  const int64_t total4 = (int64_t)B * c4;
  hipLaunchKernelGGL(k_segment_sum_stage2, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, part, c4, SEG_CHUNKS, total4, out);
  return dgdm_launch_status();
}

// ------------------------------------------------------------------ segment max (GlobalMaxPool, models/dgdm_model.py:570-585)
// out[g][c] = max over the rows of graph g of x[r][c], arg[g][c] = the row that attains it (the FIRST one on ties -- what
// torch.max(dim=0) returns on the CPU, and where its backward sends the gradient); a graph without rows gives 0 / -1 (the reference
// leaves its zero-initialised row untouched).  Two stages like the segment sum: block (chunk, g) reduces a slice of the graph's
// rows, thread = one channel x one of 256 / cp row lanes (cp = power of two >= C, <= 256), then the chunk winners are merged in
// chunk order.  Workspace: [B][SEG_CHUNKS][C] values + the same of int32 row ids.
__global__ __launch_bounds__(256) void k_segment_max_stage1(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr, int C,
                                                            int cp, int c_base, int nchunks, float* __restrict__ pv, int32_t* __restrict__ pi) {
  const int g = blockIdx.y, chunk = blockIdx.x;
  const int a = ptr[g], b = ptr[g + 1];
  const int per = (b - a + nchunks - 1) / nchunks;
  const int r0 = a + chunk * per, r1 = min(b, r0 + per);
  const int k = c_base + (int)(threadIdx.x % cp), rl = threadIdx.x / cp, nrl = 256 / cp;
  float best = -INFINITY;
  int bi = -1;
  if (k < C)
    for (int r = r0 + rl; r < r1; r += nrl) {
      const float v = x[(int64_t)r * ldx + k];
      if (v > best || bi < 0) { best = v; bi = r; }      // strict: the first row wins a tie (rows ascend per lane)
    }
  __shared__ float sv[256];
  __shared__ int si[256];
  sv[threadIdx.x] = best; si[threadIdx.x] = bi;
  __syncthreads();
  if (rl == 0 && k < C) {
    for (int j = 1; j < nrl; ++j) {
      const float v = sv[j * cp + (threadIdx.x % cp)];
      const int i = si[j * cp + (threadIdx.x % cp)];
      if (i >= 0 && (bi < 0 || v > best || (v == best && i < bi))) { best = v; bi = i; }
    }
    pv[((int64_t)g * nchunks + chunk) * C + k] = best;
    pi[((int64_t)g * nchunks + chunk) * C + k] = bi;
  }
}

__global__ __launch_bounds__(256) void k_segment_max_stage2(const float* __restrict__ pv, const int32_t* __restrict__ pi, int C, int nchunks,
                                                            int64_t total, float* __restrict__ out, int32_t* __restrict__ arg) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // over B * C
  if (i >= total) return;
  const int64_t g = i / C;
  const int k = (int)(i % C);
  float best = 0.f;
  int bi = -1;
  for (int c = 0; c < nchunks; ++c) {          // chunks hold ascending row ranges: the first maximum wins
    const float v = pv[(g * nchunks + c) * C + k];
    const int r = pi[(g * nchunks + c) * C + k];
    if (r >= 0 && (bi < 0 || v > best)) { best = v; bi = r; }
  }
  out[i] = bi < 0 ? 0.f : best;
  arg[i] = bi;
}

// dx = 0 everywhere except dx[arg[g][c]][c] = gout[g][c]: the zero fill and the scatter in one launch over the rows
__global__ __launch_bounds__(256) void k_segment_max_bwd(const float* __restrict__ gout, const int32_t* __restrict__ arg,
                                                         const int32_t* __restrict__ ptr, int B, int C, int64_t total, float* __restrict__ dx) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride) {
    const int r = (int)(i / C), k = (int)(i % C);
    int g = 0;
    while (g + 1 < B && ptr[g + 1] <= r) ++g;
    dx[i] = arg[(int64_t)g * C + k] == r ? gout[(int64_t)g * C + k] : 0.f;
  }
}

extern "C" size_t dgdm_segment_max_workspace_bytes(int32_t B, int32_t C) {
  return (B <= 0 || C <= 0) ? 0 : (size_t)B * SEG_CHUNKS * C * (sizeof(float) + sizeof(int32_t));
}

extern "C" int dgdm_segment_max_fwd(const float* x, int64_t ldx, const int32_t* ptr, int32_t B, int32_t C, float* out, int32_t* arg,
                                    void* workspace, size_t workspace_bytes, void* stream_) {
  DGDM_REQUIRE(B >= 0 && C > 0);
  if (B == 0) return DGDM_OK;
  DGDM_REQUIRE(x && ptr && out && arg && workspace);
  if (ldx < C) return DGDM_ERR_INVALID_ARG;
  if (workspace_bytes < dgdm_segment_max_workspace_bytes(B, C)) return DGDM_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  float* pv = static_cast<float*>(workspace);
  int32_t* pi = reinterpret_cast<int32_t*>(pv + (size_t)B * SEG_CHUNKS * C);
  for (int c_base = 0; c_base < C; c_base += 256) {         // 256 channels per launch (one launch for every width of the path)
    int cp = 1;
    while (cp < C - c_base && cp < 256) cp <<= 1;
    hipLaunchKernelGGL(k_segment_max_stage1, dim3(SEG_CHUNKS, B), dim3(256), 0, s, x, ldx, ptr, C, cp, c_base, SEG_CHUNKS, pv, pi);
  }
  const int64_t total = (int64_t)B * C;
  hipLaunchKernelGGL(k_segment_max_stage2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pv, pi, C, SEG_CHUNKS, total, out, arg);
  return dgdm_launch_status();
}

extern "C" int dgdm_segment_max_bwd(const float* gout, const int32_t* arg, const int32_t* ptr, int32_t B, int32_t N, int32_t C, float* dx,
                                    void* stream_) {
  DGDM_REQUIRE(B >= 0 && N >= 0 && C > 0);
  if (N == 0) return DGDM_OK;
  DGDM_REQUIRE(gout && arg && ptr && dx && B > 0);
  const int64_t total = (int64_t)N * C;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_segment_max_bwd, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), gout, arg, ptr, B, C, total, dx);
  return dgdm_launch_status();
}

#define POOL_DISPATCH(D_, KERNEL, ...)                                                                           \
  switch (D_) {                                                                                                  \
    case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(H, B), dim3(256), 0, s, __VA_ARGS__); break;                    \
    case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(H, B), dim3(256), 0, s, __VA_ARGS__); break;                    \
    case 16: hipLaunchKernelGGL((KERNEL<16>), dim3(H, B), dim3(256), 0, s, __VA_ARGS__); break;                  \
    case 32: hipLaunchKernelGGL((KERNEL<32>), dim3(H, B), dim3(256), 0, s, __VA_ARGS__); break;                  \
    case 64: hipLaunchKernelGGL((KERNEL<64>), dim3(H, B), dim3(256), 0, s, __VA_ARGS__); break;                  \
    default: return DGDM_ERR_UNSUPPORTED;                                                                        \
  }

static int pool_chunks(int32_t max_rows) { return max_rows <= 0 ? 1 : (int)(((int64_t)max_rows + POOL_ROWS - 1) / POOL_ROWS); }

extern "C" size_t dgdm_attn_pool_fwd_workspace_bytes(int32_t B, int32_t H, int32_t D, int32_t max_rows) {
  if (B <= 0 || H <= 0 || D <= 0) return 0;
  return (size_t)B * H * pool_chunks(max_rows) * ((size_t)D + 2) * sizeof(float);
}

// max_rows: number of nodes of the largest graph of the batch (the caller knows the offsets on the host)
extern "C" int dgdm_attn_pool_fwd(const float* K, const float* V, int64_t ld, const float* q_scaled, const int32_t* ptr, int32_t B,
                                  int32_t H, int32_t D, int32_t max_rows, float drop_p, uint32_t seed, float* P, float* out,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
  DGDM_REQUIRE(B >= 0 && H > 0 && D > 0 && max_rows >= 0 && drop_p >= 0.f && drop_p < 1.f);
  if (B == 0) return DGDM_OK;
  DGDM_REQUIRE(K && V && q_scaled && ptr && P && out && workspace);
  if ((ld & 3) || ld < (int64_t)H * D || !dgdm_aligned16(K) || !dgdm_aligned16(V)) return DGDM_ERR_UNSUPPORTED;
  if (workspace_bytes < dgdm_attn_pool_fwd_workspace_bytes(B, H, D, max_rows)) return DGDM_ERR_WORKSPACE;
  const int nc = pool_chunks(max_rows);
  if (nc > 65535) return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  float* part = static_cast<float*>(workspace);
  const dim3 grid(H, B, nc);
#define POOL_FWD(D_)                                                                                                              \
  hipLaunchKernelGGL((k_attn_pool_part<D_>), grid, dim3(256), 0, s, K, V, ld, q_scaled, ptr, H, drop_p, dgdm_seed_arg(seed), P, part); \
  hipLaunchKernelGGL((k_attn_pool_combine<D_>), grid, dim3(256), 0, s, part, ptr, H, P, out)
  switch (D) {
    case 4: POOL_FWD(4); break;
    case 8: POOL_FWD(8); break;
    case 16: POOL_FWD(16); break;
    case 32: POOL_FWD(32); break;
    case 64: POOL_FWD(64); break;
    default: return DGDM_ERR_UNSUPPORTED;
  }
#undef POOL_FWD
  return dgdm_launch_status();
}

extern "C" int dgdm_attn_pool_bwd(const float* K, const float* V, int64_t ld, const float* q_scaled, const int32_t* ptr, int32_t B,
                                  int32_t H, int32_t D, float drop_p, uint32_t seed, const float* P, const float* out,
                                  const float* dout, float* dK, float* dV, int64_t ldg, float* dq_partial, void* stream_) {
  DGDM_REQUIRE(B >= 0 && H > 0 && D > 0 && drop_p >= 0.f && drop_p < 1.f);
  if (B == 0) return DGDM_OK;
  DGDM_REQUIRE(K && V && q_scaled && ptr && P && out && dout && dK && dV && dq_partial);
  if ((ld & 3) || (ldg & 3) || ld < (int64_t)H * D || ldg < (int64_t)H * D || !dgdm_aligned16(K) || !dgdm_aligned16(V) ||
      !dgdm_aligned16(dK) || !dgdm_aligned16(dV))
    return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  POOL_DISPATCH(D, k_attn_pool_bwd, K, V, ld, q_scaled, ptr, H, drop_p, dgdm_seed_arg(seed), P, out, dout, dK, dV, ldg, dq_partial);
  return dgdm_launch_status();
}
