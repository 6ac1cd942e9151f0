// Microbenchmark (round 6): how does the matrix pipe round when it adds a block of products into an fp32 accumulator, and what does a
// long accumulation chain cost in accuracy?   hipcc --offload-arch=gfx950 -O2 -o mfma_rounding mfma_rounding.hip && ./mfma_rounding
//   part 1: C = +-1, ONE non-zero product d = t * 2^-24 per output: the printed result shows the rounding of C + d
//           (round-to-nearest-even gives 1, 1, 1(tie), 1+ulp, 1+ulp for t = .5 .75 1 1.25 1.5; truncation gives 1 five times);
//   part 2: chains of S accumulating MFMAs on random fp16 operands against float64: mean signed error (bias) and rms error of
//           (a) one accumulator for the whole chain, (b) a fresh accumulator every F steps, flushed into a second one by a VALU add
//           (round to nearest), (c) the same sums by a round-to-nearest fp32 add of exactly computed 32-term blocks (the ideal chain).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void k_round(float* out, float c0) {
  const int l = threadIdx.x;
  const float ts[6] = {0.5f, 0.75f, 1.0f, 1.25f, 1.5f, 1.75f};
  for (int sgn = 0; sgn < 2; ++sgn)
    for (int ti = 0; ti < 6; ++ti) {
      const float t = sgn ? -ts[ti] : ts[ti];
      // f16 16x16x32: lane (i = l & 15, g = l >> 4) holds k = 8 g + e
      f16x8 ha, hb;
      for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)0.f; hb[e] = (_Float16)0.f; }
      if ((l >> 4) == 0) { ha[0] = (_Float16)ldexpf(1.f, -12); hb[0] = (_Float16)ldexpf(t, -12); }
      f32x4 c = {c0, c0, c0, c0};
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c, 0, 0, 0);
      // f16 32x32x16: lane (i = l & 31, g = l >> 5) holds k = 8 g + e
      f16x8 ha2, hb2;
      for (int e = 0; e < 8; ++e) { ha2[e] = (_Float16)0.f; hb2[e] = (_Float16)0.f; }
      if ((l >> 5) == 0) { ha2[0] = (_Float16)ldexpf(1.f, -12); hb2[0] = (_Float16)ldexpf(t, -12); }
      f32x16 c2;
      for (int r = 0; r < 16; ++r) c2[r] = c0;
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha2, hb2, c2, 0, 0, 0);
      // fp32 32x32x2: lane (i = l & 31, k = l >> 5)
      const float a = (l >> 5) == 0 ? ldexpf(1.f, -12) : 0.f, b = (l >> 5) == 0 ? ldexpf(t, -12) : 0.f;
      f32x16 c3;
      for (int r = 0; r < 16; ++r) c3[r] = c0;
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
      // fp32 16x16x4: lane (i = l & 15, k = l >> 4)
      const float a4 = (l >> 4) == 0 ? ldexpf(1.f, -12) : 0.f, b4 = (l >> 4) == 0 ? ldexpf(t, -12) : 0.f;
      f32x4 c4 = {c0, c0, c0, c0};
      c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4, b4, c4, 0, 0, 0);
      if (l == 0) {
        float* o = out + (sgn * 6 + ti) * 4;
        o[0] = c[0]; o[1] = c2[0]; o[2] = c3[0]; o[3] = c4[0];
      }
    }
}

// every output of the tile is the same sum: A rows identical, B columns identical.  data: [trial][step][32] halves for a and b.
template <int FLUSH>
__global__ __launch_bounds__(64) void k_chain(const _Float16* __restrict__ av, const _Float16* __restrict__ bv, int steps, float* out) {
  const int l = threadIdx.x, g = l >> 4;
  const _Float16* a = av + (size_t)blockIdx.x * steps * 32;
  const _Float16* b = bv + (size_t)blockIdx.x * steps * 32;
  f32x4 acc = {0, 0, 0, 0}, tot = {0, 0, 0, 0};
  for (int s = 0; s < steps; ++s) {
    f16x8 ha, hb;
    for (int e = 0; e < 8; ++e) { ha[e] = a[s * 32 + 8 * g + e]; hb[e] = b[s * 32 + 8 * g + e]; }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0);
    if (FLUSH > 0 && (s + 1) % FLUSH == 0) { tot += acc; acc = f32x4{0, 0, 0, 0}; }
  }
  if (FLUSH > 0) acc += tot;
  if (l == 0) out[blockIdx.x] = acc[0];
}

int main() {
  float* d; hipMalloc(&d, 4 * 48 * 2);
  for (float c0 : {1.0f, -1.0f}) {
    hipLaunchKernelGGL(k_round, dim3(1), dim3(64), 0, 0, d, c0);
    float h[48]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("part 1: C = %+.0f, one product d = t * 2^-24 (ulp of C = 2^-23 above |C|, 2^-24 below); result - C in units of 2^-24\n", c0);
    printf("   t      f16 16x16x32  f16 32x32x16  f32 32x32x2   f32 16x16x4   round-to-nearest-even would give\n");
    const float ts[6] = {0.5f, 0.75f, 1.0f, 1.25f, 1.5f, 1.75f};
    for (int sgn = 0; sgn < 2; ++sgn)
      for (int ti = 0; ti < 6; ++ti) {
        const float t = sgn ? -ts[ti] : ts[ti];
        const float rn = (float)((double)c0 + (double)t * ldexp(1.0, -24));
        printf("  %+5.2f ", t);
        for (int j = 0; j < 4; ++j) printf("  %+10.2f  ", (h[(sgn * 6 + ti) * 4 + j] - c0) * ldexpf(1.f, 24));
        printf("  %+10.2f\n", (rn - c0) * ldexpf(1.f, 24));
      }
  }
  for (int steps : {24, 64, 192, 256, 1875, 2500}) {
    const int trials = steps > 256 ? 512 : 4096;
    std::vector<_Float16> a((size_t)trials * steps * 32), b(a.size());
    srand(1234 + steps);
    for (size_t i = 0; i < a.size(); ++i) {
      a[i] = (_Float16)(rand() / (float)RAND_MAX * 2 - 1);
      b[i] = (_Float16)(rand() / (float)RAND_MAX * 2 - 1 + 0.25f);      // a small common component, as activations have
    }
    _Float16 *da, *db; float* dout;
    hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dout, trials * 4);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    std::vector<double> ref(trials), mag(trials);
    std::vector<float> ideal(trials);
    for (int t = 0; t < trials; ++t) {
      double s = 0, m = 0; float f = 0;
      for (int st = 0; st < steps; ++st) {
        double blk = 0;
        for (int k = 0; k < 32; ++k) { const double p = (double)(float)a[((size_t)t * steps + st) * 32 + k] * (double)(float)b[((size_t)t * steps + st) * 32 + k]; blk += p; m += p * p; }
        s += blk; f = (float)((double)f + blk);       // (c) exact block, one round-to-nearest add
      }
      ref[t] = s; mag[t] = sqrt(m); ideal[t] = f;
    }
    auto stats = [&](const char* name, const float* got) {
      double se = 0, se2 = 0, sr = 0;
      for (int t = 0; t < trials; ++t) { const double e = (double)got[t] - ref[t]; se += e / mag[t]; se2 += e * e / (mag[t] * mag[t]); sr += fabs(ref[t]) / mag[t]; }
      printf("  %-44s mean signed err %+.3e   rms err %.3e   (units: sqrt(sum p^2); |result| is %.2f of it; 2^-24 = 5.96e-08)\n", name, se / trials,
             sqrt(se2 / trials), sr / trials);
    };
    printf("part 2: %d accumulation steps of v_mfma_f32_16x16x32_f16 (K = %d), %d trials\n", steps, steps * 32, trials);
    std::vector<float> got(trials);
    hipLaunchKernelGGL((k_chain<0>), dim3(trials), dim3(64), 0, 0, da, db, steps, dout);
    hipMemcpy(got.data(), dout, trials * 4, hipMemcpyDeviceToHost); stats("(a) one MFMA accumulator", got.data());
    hipLaunchKernelGGL((k_chain<96>), dim3(trials), dim3(64), 0, 0, da, db, steps, dout);
    hipMemcpy(got.data(), dout, trials * 4, hipMemcpyDeviceToHost); stats("(b) fresh accumulator every 96 steps + VALU add", got.data());
    hipLaunchKernelGGL((k_chain<48>), dim3(trials), dim3(64), 0, 0, da, db, steps, dout);
    hipMemcpy(got.data(), dout, trials * 4, hipMemcpyDeviceToHost); stats("(b) fresh accumulator every 48 steps + VALU add", got.data());
    hipLaunchKernelGGL((k_chain<24>), dim3(trials), dim3(64), 0, 0, da, db, steps, dout);
    hipMemcpy(got.data(), dout, trials * 4, hipMemcpyDeviceToHost); stats("(b) fresh accumulator every 24 steps + VALU add", got.data());
    hipLaunchKernelGGL((k_chain<8>), dim3(trials), dim3(64), 0, 0, da, db, steps, dout);
    hipMemcpy(got.data(), dout, trials * 4, hipMemcpyDeviceToHost); stats("(b) fresh accumulator every 8 steps + VALU add", got.data());
    hipLaunchKernelGGL((k_chain<2>), dim3(trials), dim3(64), 0, 0, da, db, steps, dout);
    hipMemcpy(got.data(), dout, trials * 4, hipMemcpyDeviceToHost); stats("(b) fresh accumulator every 2 steps + VALU add", got.data());
    hipLaunchKernelGGL((k_chain<1>), dim3(trials), dim3(64), 0, 0, da, db, steps, dout);
    hipMemcpy(got.data(), dout, trials * 4, hipMemcpyDeviceToHost); stats("(b) fresh accumulator every step + VALU add", got.data());
    stats("(c) exact blocks, round-to-nearest adds", ideal.data());
    hipFree(da); hipFree(db); hipFree(dout);
  }
  return 0;
}
