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

}  // namespace
