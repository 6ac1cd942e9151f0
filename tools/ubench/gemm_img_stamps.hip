// Diagnostic build of csrc/gemm_img.hip with in-kernel stamps: where does a small weight-image GEMM spend its time?
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I include -I dgdm_histopath_lab_amd/csrc tools/ubench/gemm_img_stamps.hip -o tools/ubench/gemm_img_stamps
// Runs M x K x 128 on random data, prints per-phase deltas of workgroup 0 / wave 0 in shader cycles and in ns (s_memrealtime, 100 MHz).
#define DGDM_GEMM_IMG_STAMPS 1
#include "../../dgdm_histopath_lab_amd/csrc/gemm_img.hip"
const uint32_t* dgdm_seed_epoch_ptr() { return nullptr; }      // api.hip's (the epilogue variants of the file take a dropout seed; none runs here)
#include <cstdio>
#include <vector>
#include <cstdlib>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e__), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 128, K = argc > 2 ? atoi(argv[2]) : 160, N = argc > 3 ? atoi(argv[3]) : 128;
  float *A, *W, *C; unsigned *amax; char* img; unsigned long long* stamps;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
  CK(hipMalloc(&amax, 2 * DGDM_AMAX_WAYS * DGDM_AMAX_STRIDE * 4)); CK(hipMemset(amax, 0, 2 * DGDM_AMAX_WAYS * DGDM_AMAX_STRIDE * 4));
  CK(hipMalloc(&img, dgdm_gemm_image_bytes(N, K))); CK(hipMalloc(&stamps, 64 * 8)); CK(hipMemset(stamps, 0, 64 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(dgdm_stamp_buf), &stamps, sizeof(stamps)));
  std::vector<float> h((size_t)M * K); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  h.resize((size_t)N * K); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  unsigned one = 0x3f800000u;   // amax = 1.0 for both operands
  CK(hipMemcpy(amax, &one, 4, hipMemcpyHostToDevice)); CK(hipMemcpy(amax + DGDM_AMAX_WAYS * DGDM_AMAX_STRIDE, &one, 4, hipMemcpyHostToDevice));
  if (dgdm_gemm_image_build(W, K, nullptr, 0, amax + DGDM_AMAX_WAYS * DGDM_AMAX_STRIDE, nullptr, img, N, K, 0, 0, nullptr) != 0) { printf("image build failed\n"); return 1; }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 20; ++i)
      if (dgdm_gemm_rows_img(A, K, M, K, img, (N + 31) / 32, 0, N, nullptr, C, N, 0, amax, nullptr) != 0) { printf("gemm failed\n"); return 1; }
    CK(hipEventRecord(e1, nullptr)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long s[20]; CK(hipMemcpy(s, stamps, sizeof(s), hipMemcpyDeviceToHost));
    printf("rep %d: %.2f us per launch (20 back to back); wg0/wave0 stamps [cycles | ns since kernel start]:", rep, ms * 1e3 / 20);
    for (int i = 1; i < 10; ++i) if (s[2 * i]) printf("  #%d %lld | %lld", i, (long long)(s[2 * i] - s[0]), (long long)(s[2 * i + 1] - s[1]) * 10);
    printf("\n");
  }
  return 0;
}
