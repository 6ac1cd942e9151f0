// How fast can a wave fetch its 32 x K fp32 rows?  The image GEMMs read A "row per lane" (lane l: row l & 31, 64 contiguous bytes per
// 32-k chunk -- 64 distinct cache lines per wave instruction); this compares that pattern with the same bytes fetched coalesced
// (a wave instruction = 1 KiB contiguous) and with LDS-DMA, one wave per 32 rows, 4 waves per workgroup, as k_gemm_img<4,1> launches.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/ubench/a_load_pattern.hip -o tools/ubench/a_load_pattern && tools/ubench/a_load_pattern 40000 128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e__), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k_read(const float* __restrict__ A, int M, int K, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = (blockIdx.x * 4 + wave) * 32;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  __shared__ __attribute__((aligned(16))) char smem[4][16384];
  if (MODE == 0) {          // row per lane
    const int row = min(r0 + (lane & 31), M - 1);
    const float* p = A + (int64_t)row * K + 16 * (lane >> 5);
    for (int c = 0; c < K / 32; ++c) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += *reinterpret_cast<const f32x4*>(p + 32 * c + 4 * i);
    }
  } else if (MODE == 1) {   // coalesced: the wave's 32 rows are one contiguous range of 32 K floats
    const float* p = A + (int64_t)min(r0, M - 32) * K + 4 * lane;
    for (int i = 0; i < K / 8; ++i) acc += *reinterpret_cast<const f32x4*>(p + 256 * i);
  } else {                  // LDS-DMA of the same range (K <= 128), then a row-per-lane read out of the LDS
    const char* g = reinterpret_cast<const char*>(A + (int64_t)min(r0, M - 32) * K) + lane * 16;
    for (int i = 0; i < K / 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 1024 * i),
                                       (__attribute__((address_space(3))) void*)(smem[wave] + 1024 * i), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float* p = reinterpret_cast<const float*>(smem[wave]) + (lane & 31) * K + 16 * (lane >> 5);
    for (int c = 0; c < K / 32; ++c) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += *reinterpret_cast<const f32x4*>(p + 32 * c + 4 * ((i + (lane & 3)) & 3));
    }
  }
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE>
int run(const char* name, const float* A, int M, int K, float* out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = (M + 127) / 128;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_read<MODE>, dim3(grid), dim3(256), 0, nullptr, A, M, K, out);
    CK(hipEventRecord(e1, nullptr)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-12s M=%d K=%d: %.2f us per launch = %.2f TB/s\n", name, M, K, ms * 1e3 / 50, (double)M * K * 4 / (ms * 1e-3 / 50) / 1e12);
  }
  return 0;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 40000, K = argc > 2 ? atoi(argv[2]) : 128;
  float *A, *out;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMemset(A, 0, (size_t)M * K * 4)); CK(hipMalloc(&out, (size_t)(M + 256) * 8));
  if (run<0>("row-per-lane", A, M, K, out)) return 1;
  if (run<1>("coalesced", A, M, K, out)) return 1;
  if (K <= 128 && run<2>("lds-dma", A, M, K, out)) return 1;
  return 0;
}
