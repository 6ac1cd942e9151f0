// Input checks of DGDMModel.forward (models/dgdm_model.py:646-690 of the reference: NaN / inf in the node features,
// edge ids outside [0, N)) as ONE pass over x and ONE over edge_index, instead of the ~10 reductions the tensor
// expressions isnan().any(), isinf().any(), max(), min() launch.  The result is four flags the host reads back once.
#include "common.hpp"

namespace {

// flags[0]: some x is NaN   flags[1]: some x is +-inf   flags[2]: some edge id > N-1   flags[3]: some edge id < 0
__global__ __launch_bounds__(256) void k_validate(const float* __restrict__ x, int64_t nx, const int64_t* __restrict__ ei,
                                                  int64_t ne, int64_t N, uint32_t* __restrict__ flags) {
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool has_nan = false, has_inf = false, too_big = false, negative = false;
  const int64_t n4 = nx >> 2;
  for (int64_t i = tid; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      has_nan |= a[j] != a[j];
      has_inf |= fabsf(a[j]) == INFINITY;
    }
  }
  for (int64_t i = (n4 << 2) + tid; i < nx; i += stride) {
    has_nan |= x[i] != x[i];
    has_inf |= fabsf(x[i]) == INFINITY;
  }
  for (int64_t i = tid; i < ne; i += stride) {
    const int64_t v = ei[i];
    too_big |= v > N - 1;
    negative |= v < 0;
  }
  if (__any(has_nan) && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 1u);
  if (__any(has_inf) && (threadIdx.x & 63) == 0) atomicOr(&flags[1], 1u);
  if (__any(too_big) && (threadIdx.x & 63) == 0) atomicOr(&flags[2], 1u);
  if (__any(negative) && (threadIdx.x & 63) == 0) atomicOr(&flags[3], 1u);
}

}  // namespace

extern "C" int dgdm_validate_inputs(const float* x, int64_t x_numel, const int64_t* edge_index, int64_t edge_numel,
                                    int64_t num_nodes, uint32_t* flags4, void* stream_) {
  DGDM_REQUIRE(x_numel >= 0 && edge_numel >= 0 && num_nodes >= 0 && flags4);
  DGDM_REQUIRE((x_numel == 0 || x) && (edge_numel == 0 || edge_index));
  if (x_numel > 0 && !dgdm_aligned16(x)) return DGDM_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  dgdm_fill_async(flags4, 0, 4 * sizeof(uint32_t), s);
  const int64_t work = (x_numel >> 2) > edge_numel ? (x_numel >> 2) : edge_numel;
  if (work == 0 && x_numel == 0) return dgdm_launch_status();
  int64_t blocks = (work + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_validate, dim3((unsigned)blocks), dim3(256), 0, s, x, x_numel, edge_index, edge_numel, num_nodes, flags4);
  return dgdm_launch_status();
}
