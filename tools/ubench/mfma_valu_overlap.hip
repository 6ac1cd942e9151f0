// Microbenchmark: can fp32 MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2) overlap with fp32 VALU work in the same wave
// and across two waves of a SIMD?  Prints cycles per MFMA for V = 0..12 independent v_fma per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int V, int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc4[4]; f32x16 acc16[2];
  for (int i = 0; i < 4; ++i) acc4[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc16[i][r] = 0;
  float v[12];
  for (int i = 0; i < 12; ++i) v[i] = a0 + i + threadIdx.x;
  float a = a0 + threadIdx.x, b = b0;
  f16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a0 + i); hb[i] = (_Float16)(b0 * 0.01f * i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (SHAPE == 1632) acc4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4[u & 3], 0, 0, 0);
      else if (SHAPE == 16) acc4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[u & 3], 0, 0, 0);
      else acc16[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc16[u & 1], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc4[i][0] + acc4[i][3];
  for (int i = 0; i < 2; ++i) s += acc16[i][0] + acc16[i][15];
  for (int i = 0; i < 12; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V, int SHAPE>
void run(int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;  // 256-thread blocks = 1 wave per SIMD each
  float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<V, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f, 2.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 8 * waves_per_simd;
  const double flop = (double)blocks * 4 * iters * 8 * (SHAPE == 16 ? 2048.0 : (SHAPE == 1632 ? 16384.0 : 4096.0));
  printf("shape %2d  V=%2d  waves/SIMD=%d  %.3f ms  %.1f ns per MFMA per SIMD  (%.1f cycles @2.4GHz)  MFMA TFLOP/s %.1f\n", SHAPE, V,
         waves_per_simd, ms, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int w : {1, 2}) {
    run<0, 16>(w); run<8, 16>(w);
  }
  for (int w : {1, 2}) { run<0, 1632>(w); run<2, 1632>(w); run<4, 1632>(w); run<6, 1632>(w); run<8, 1632>(w); run<12, 1632>(w); }
  return 0;
}
