// Microbenchmark / hardware check (gfx950): can a wave publish its 64-lane predicate as ONE 64-bit word without vector work, and can
// a later kernel consume such words as the condition of v_cndmask_b32 without a compare?
//   writer A: v_cmp -> SGPR pair -> s_store_dwordx2 (scalar store, s_dcache_wb at the end)         [0 vector instructions per word]
//   writer B: v_cmp -> SGPR pair -> v_mov into the lane that owns the word, one coalesced vector store per 64 words
//   reader  : s_load_dwordx16 -> v_cndmask_b32 with the SGPR pair as the condition; checks every bit against the predicate
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/scalar_mask_words.hip -o tools/ubench/scalar_mask_words ; run on a GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
#ifdef CHEAP_PRED   // 2 instructions: shows the ceiling of the store path instead of the hash
__device__ __forceinline__ bool pred(uint32_t word, uint32_t lane) { return ((word * 0x9E3779B1u) >> (lane & 31)) & 1u; }
#else
__device__ __forceinline__ bool pred(uint32_t word, uint32_t lane) { return (int)mix(word * 64u + lane + 0x9E3779B9u) >= -1288490189; }   // ~0.8 kept
#endif

// one wave per 16 consecutive words, WPB waves per block
constexpr int WORDS = 16;

__global__ __launch_bounds__(256) void k_write_scalar(uint64_t* out, int64_t nwords) {
  const uint32_t lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * WORDS;
  if (w0 >= nwords) return;
  uint64_t m[WORDS];
#pragma unroll
  for (int i = 0; i < WORDS; ++i) m[i] = __builtin_amdgcn_ballot_w64(pred((uint32_t)(w0 + i), lane));
  uint64_t* p = out + w0;
#pragma unroll
  for (int i = 0; i < WORDS; i += 2) {
    typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
    u64x2 v = {m[i], m[i + 1]};
    asm volatile("s_store_dwordx4 %0, %1, %2" :: "s"(v), "s"(p), "n"(i * 8) : "memory");
  }
  asm volatile("s_dcache_wb" ::: "memory");
}

__global__ __launch_bounds__(256) void k_write_vector(uint64_t* out, int64_t nwords) {
  const uint32_t lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * WORDS;
  if (w0 >= nwords) return;
  uint64_t mine = 0;
#pragma unroll
  for (int i = 0; i < WORDS; ++i) {
    const uint64_t m = __builtin_amdgcn_ballot_w64(pred((uint32_t)(w0 + i), lane));
    mine = lane == (uint32_t)i ? m : mine;     // two v_cndmask per word
  }
  if (lane < WORDS) out[w0 + lane] = mine;
}

__device__ __forceinline__ float sel(float a0, float a1, uint64_t m) {   // lane's bit of m ? a1 : a0, no compare
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a0), "v"(a1), "s"(m));
  return r;
}

__global__ __launch_bounds__(256) void k_read_check(const uint64_t* __restrict__ in, int64_t nwords, unsigned* bad, float* sink) {
  const uint32_t lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * WORDS;
  if (w0 >= nwords) return;
  const uint64_t* p = (const uint64_t*)__builtin_assume_aligned(in + w0, 128);
  unsigned wrong = 0;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < WORDS; ++i) {
    const uint64_t ms = p[i];            // wave-uniform address, read-only data: s_load
    const float v = sel(0.f, 1.f, ms);
    acc += v;
    wrong += (v != (pred((uint32_t)(w0 + i), lane) ? 1.f : 0.f));
  }
  if (wrong) atomicAdd(bad, wrong);
  if (acc == -1.f) sink[0] = acc;
}

int main() {
  const int64_t nwords = (int64_t)400e6 / 8;      // 400 MB: the mask of 4 x 10 000 nodes x 8 heads
  const int blocks = (int)((nwords / WORDS + 3) / 4);
  uint64_t *a, *b;
  unsigned* bad;
  float* sink;
  CHECK(hipMalloc(&a, nwords * 8)); CHECK(hipMalloc(&b, nwords * 8)); CHECK(hipMalloc(&bad, 8)); CHECK(hipMalloc(&sink, 8));
  CHECK(hipMemset(a, 0xEE, nwords * 8)); CHECK(hipMemset(b, 0xDD, nwords * 8)); CHECK(hipMemset(bad, 0, 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_write_scalar, dim3(blocks), dim3(256), 0, 0, a, nwords); CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("scalar-store writer: %.3f ms  (%.0f GB/s)\n", ms, nwords * 8 / ms / 1e6);
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_write_vector, dim3(blocks), dim3(256), 0, 0, b, nwords); CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("vector-store writer: %.3f ms  (%.0f GB/s)\n", ms, nwords * 8 / ms / 1e6);
  }
  std::vector<uint64_t> ha(1 << 16), hb(1 << 16);
  CHECK(hipMemcpy(ha.data(), a + nwords / 2, ha.size() * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hb.data(), b + nwords / 2, hb.size() * 8, hipMemcpyDeviceToHost));
  size_t diff = 0;
  for (size_t i = 0; i < ha.size(); ++i) diff += ha[i] != hb[i];
  printf("words differing between the two writers (sample of %zu): %zu ; first word %016llx\n", ha.size(), diff, (unsigned long long)ha[0]);
  for (int which = 0; which < 2; ++which) {
    CHECK(hipMemset(bad, 0, 8));
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read_check, dim3(blocks), dim3(256), 0, 0, which ? b : a, nwords, bad, sink); CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned hbad = 0;
    CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    printf("reader over the %s writer's words: %.3f ms, wrong bits %u\n", which ? "vector" : "scalar", ms, hbad);
  }
  return 0;
}
