// Microbenchmark (round 6), follow-up of mfma_valu_overlap2: WHAT of an MFMA keeps vector instructions from issuing beside it?
// Two waves per SIMD, V v_fma_f32 per 16 K-FLOP of MFMA (per v_mfma_f32_16x16x32_f16; 2 V per 32x32x16).  Variants of the MFMA:
//   vgpr   : A, B, C / D in VGPRs (what hipcc emits with -amdgpu-mfma-vgpr-form=1)
//   agprC  : C / D in AGPRs
//   agprAB : A, B and C / D in AGPRs
//   zeroC  : C = inline constant 0, D in VGPRs (a "fresh" product: no accumulator read)
//   32x32  : v_mfma_f32_32x32x16_f16 (twice the FLOP per instruction), C / D in VGPRs or AGPRs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int V, int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float a0, float b0) {
  f32x4 acc[4]; f32x16 big[2];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) big[i][r] = 0;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = a0 + i + threadIdx.x;
  f16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a0 + i); hb[i] = (_Float16)(b0 * 0.01f * i); }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(ha), "v"(hb));
      if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[u & 3]) : "v"(ha), "v"(hb));
      if (MODE == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[u & 3]) : "a"(ha), "a"(hb));
      if (MODE == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(acc[u & 3]) : "v"(ha), "v"(hb));
      if (MODE == 4 && (u & 1) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(big[(u >> 1) & 1]) : "v"(ha), "v"(hb));
      if (MODE == 5 && (u & 1) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(big[(u >> 1) & 1]) : "v"(ha), "v"(hb));
#pragma unroll
      for (int j = 0; j < V; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(a0), "v"(b0));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 2; ++i) s += big[i][0] + big[i][15];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V, int MODE>
void run(int waves_per_simd) {
  static const char* names[] = {"vgpr", "agprC", "agprAB", "zeroC", "32x32 vgpr", "32x32 agprC", "no mfma"};
  const int blocks = 256 * waves_per_simd;
  float* out; long long* cyc; (void)hipMalloc(&out, sizeof(float) * blocks * 256); (void)hipMalloc(&cyc, 8 * blocks * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<V, MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, 100, 1.0f, 2.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f, 2.0f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  static long long h[4096]; (void)hipMemcpy(h, cyc, 8 * blocks * 4, hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < blocks * 4; ++i) mean += h[i]; mean /= blocks * 4;
  const double slots = (double)iters * 8;
  printf("%-12s V=%2d waves/SIMD=%d  %7.3f ms  %5.1f SIMD-cycles per 16 KFLOP slot  (counter %.0f MHz)\n", names[MODE], V, waves_per_simd, ms,
         mean / slots / waves_per_simd, mean / (ms * 1e3));
  (void)hipFree(out); (void)hipFree(cyc);
}

template <int MODE> void sweep() { for (int w : {1, 2}) { run<0, MODE>(w); run<4, MODE>(w); run<8, MODE>(w); run<12, MODE>(w); } }

int main() {
  for (int w : {1, 2}) { run<4, 6>(w); run<8, 6>(w); run<12, 6>(w); }
  sweep<0>(); sweep<1>(); sweep<2>(); sweep<3>(); sweep<4>(); sweep<5>();
  return 0;
}
