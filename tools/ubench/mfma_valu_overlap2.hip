// Microbenchmark (round 6): do v_mfma_f32_16x16x32_f16 and vector instructions overlap on gfx950 -- inside one wave's instruction stream,
// and between the two waves a SIMD holds?  Times are SHADER CYCLES (s_memtime around the loop, per wave; independent of the clock the chip
// holds under the load) next to wall time.  VALU work = v_fma_f32 on scalars the compiler cannot pack (different multiplicands per
// value are kept in inline asm).  Modes: M = MFMAs per iteration (0 or 8), V = vector instructions per MFMA slot.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int V, bool MFMA, int KIND>      // KIND 0: v_fma_f32, 1: v_exp_f32 (quarter rate), 2: v_pk_fma_f32
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float a0, float b0) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = a0 + i + threadIdx.x;
  f16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a0 + i); hb[i] = (_Float16)(b0 * 0.01f * i); }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MFMA) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[u & 3], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(a0), "v"(b0));
        else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double*>(&v[2 * (j & 7)])) : "v"(*reinterpret_cast<double*>(&v[14])), "v"(*reinterpret_cast<double*>(&v[12])));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V, bool MFMA, int KIND>
void run(int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;      // 256-thread blocks = 1 wave per SIMD each
  float* out; long long* cyc; hipMalloc(&out, sizeof(float) * blocks * 256); hipMalloc(&cyc, 8 * blocks * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<V, MFMA, KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, 100, 1.0f, 2.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, MFMA, KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f, 2.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  static long long h[4096]; hipMemcpy(h, cyc, 8 * blocks * 4, hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < blocks * 4; ++i) mean += h[i]; mean /= blocks * 4;
  const double slots = (double)iters * 8;      // MFMA slots per wave
  printf("%s V=%2d %-8s waves/SIMD=%d  %7.3f ms  %6.1f wave-cycles per slot  = %5.1f SIMD-cycles per slot  (counter runs at %.0f MHz)\n",
         MFMA ? "mfma+" : "     ", V, KIND == 0 ? "v_fma" : KIND == 1 ? "v_exp" : "v_pk_fma", waves_per_simd, ms, mean / slots, mean / slots / waves_per_simd,
         mean / (ms * 1e3));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {1, 2}) {
    run<0, true, 0>(w);
    run<2, false, 0>(w); run<4, false, 0>(w); run<8, false, 0>(w);
    run<1, true, 0>(w); run<2, true, 0>(w); run<3, true, 0>(w); run<4, true, 0>(w); run<6, true, 0>(w); run<8, true, 0>(w); run<12, true, 0>(w);
    run<1, false, 1>(w); run<1, true, 1>(w); run<2, true, 1>(w);
    run<4, false, 2>(w); run<4, true, 2>(w);
  }
  return 0;
}
