// Fixed-order column sums of a [slots][width] fp32 buffer (two launches: chunked partial sums, then the
// final sum): the deterministic second stage of every split reduction in this library (dgamma/dbeta of
// the row norms, dW/db of the split-M GEMM).
#pragma once
#include "common.hpp"

namespace {

// column sums over `slots` rows of a [slots][width] buffer, fixed order => reproducible.
// grid = (ceil(width/64), nchunks): block (x, y) sums slot chunk y of 64 columns; 256 threads =
// 64 columns x 4 slot-lanes.  Stage 1 writes [nchunks][width], stage 2 (nchunks == 1) the result.
// split > 0 (final stage): columns [0, split) go to out, [split, width) to out1.
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ in, int64_t slots, int width, int64_t chunk,
                                                float* __restrict__ out, int split, float* __restrict__ out1) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  const int64_t s0 = (int64_t)blockIdx.y * chunk;
  const int64_t s1 = s0 + chunk < slots ? s0 + chunk : slots;
  float acc = 0.f;
  if (col < width)
    for (int64_t s = s0 + part; s < s1; s += 4) acc += in[s * width + col];
  __shared__ float sm[4][64];
  sm[part][threadIdx.x & 63] = acc;
  __syncthreads();
  if (part == 0 && col < width) {
    const float t = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
    if (split > 0) {
      if (col < split) out[col] = t; else out1[col - split] = t;
    } else {
      out[(int64_t)blockIdx.y * width + col] = t;
    }
  }
}

// Single-launch variant: each block owns 256/PARTS columns, its PARTS thread rows stride over the slots (two loads in
// flight), LDS combines them in fixed order.  PARTS = 16 for hundreds of slots, 4 for a few.
// split > 0: columns [0, split) go to out, [split, width) to out1; split <= 0: everything to out.
template <int PARTS>
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ in, int slots, int width, float* __restrict__ out,
                                                      int split, float* __restrict__ out1) {
  constexpr int COLS = 256 / PARTS;
  const int c = threadIdx.x % COLS, part = threadIdx.x / COLS;
  const int col = blockIdx.x * COLS + c;
  float a0 = 0.f, a1 = 0.f;
  if (col < width) {
    int s = part;
    for (; s + PARTS < slots; s += 2 * PARTS) {
      a0 += in[(int64_t)s * width + col];
      a1 += in[(int64_t)(s + PARTS) * width + col];
    }
    if (s < slots) a0 += in[(int64_t)s * width + col];
  }
  __shared__ float sm[PARTS][COLS];
  sm[part][c] = a0 + a1;
  __syncthreads();
  if (part == 0 && col < width) {
    float t = sm[0][c];
#pragma unroll
    for (int p = 1; p < PARTS; ++p) t += sm[p][c];
    if (split > 0 && col >= split) out1[col - split] = t; else out[col] = t;
  }
}

// Two stages in ONE launch for thousands of slots: block (x, y) sums slot chunk y of 64 columns into stage[y][:]
// (k_colsum's stage 1), then takes a ticket for its column block; the block that draws the last ticket adds the
// gridDim.y stage rows in index order and writes the result -- fixed summation order whichever block that is.
// tickets[gridDim.x] must be zero at launch (the producer kernel of `in` zeroes them: stream order).  gridDim.y <= MAXCH.
template <int MAXCH>
__global__ __launch_bounds__(256) void k_colsum_ticket(const float* __restrict__ in, int64_t slots, int width, int64_t chunk,
                                                       float* __restrict__ stage, unsigned* __restrict__ tickets,
                                                       float* __restrict__ out, int split, float* __restrict__ out1) {
  const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  const int64_t s0 = (int64_t)blockIdx.y * chunk;
  const int64_t s1 = s0 + chunk < slots ? s0 + chunk : slots;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < width) {
    int64_t s = s0 + part;
    for (; s + 12 < s1; s += 16) {   // four loads in flight
      a0 += in[s * width + col];
      a1 += in[(s + 4) * width + col];
      a2 += in[(s + 8) * width + col];
      a3 += in[(s + 12) * width + col];
    }
    for (; s < s1; s += 4) a0 += in[s * width + col];
  }
  __shared__ float sm[4][64];
  __shared__ bool last;
  sm[part][c] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (part == 0 && col < width) {
    stage[(int64_t)blockIdx.y * width + col] = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
    __threadfence();
  }
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&tickets[blockIdx.x], 1u) == gridDim.y - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (part == 0 && col < width) {
    float v[MAXCH];   // all loads in flight at once (a serial loop pays the memory latency gridDim.y times)
#pragma unroll
    for (int y = 0; y < MAXCH; ++y) v[y] = y < (int)gridDim.y ? stage[(int64_t)y * width + col] : 0.f;
    float t = 0.f;
#pragma unroll
    for (int y = 0; y < MAXCH; ++y) t += v[y];
    if (split > 0 && col >= split) out1[col - split] = t; else out[col] = t;
  }
}

inline void colsum_final_launch(const float* in, int slots, int width, float* out, int split, float* out1, hipStream_t s) {
  if (slots > 32)
    hipLaunchKernelGGL(k_colsum_final<16>, dim3((width + 15) / 16), dim3(256), 0, s, in, slots, width, out, split, out1);
  else
    hipLaunchKernelGGL(k_colsum_final<4>, dim3((width + 63) / 64), dim3(256), 0, s, in, slots, width, out, split, out1);
}

}  // namespace
