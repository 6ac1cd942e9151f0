// ABI bookkeeping for libdgdm_hip.so.
#include "common.hpp"

extern "C" int dgdm_abi_version(void) { return DGDM_ABI_VERSION; }

extern "C" const char* dgdm_error_string(int code) {
  switch (code) {
    case DGDM_OK: return "ok";
    case DGDM_ERR_INVALID_ARG: return "invalid argument (null pointer, negative size or bad enum)";
    case DGDM_ERR_UNSUPPORTED: return "unsupported shape/alignment for this kernel";
    case DGDM_ERR_WORKSPACE: return "workspace too small";
    case DGDM_ERR_LAUNCH: return "HIP launch failed";
    default: return "unknown dgdm error code";
  }
}

// ---------------------------------------------------------------- dropout seed epoch (see common.hpp)
__device__ uint32_t g_dgdm_seed_epoch = 0;

const uint32_t* dgdm_seed_epoch_ptr() {
  static const uint32_t* cached[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (!cached[dev]) {
    void* p = nullptr;
    (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_dgdm_seed_epoch));
    cached[dev] = static_cast<const uint32_t*>(p);
  }
  return cached[dev];
}

namespace {
__global__ void k_seed_epoch(uint32_t* p, uint32_t value, int set) { *p = set ? value : *p + 1u; }
}  // namespace

extern "C" int dgdm_seed_epoch_advance(void* stream) {
  hipLaunchKernelGGL(k_seed_epoch, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), const_cast<uint32_t*>(dgdm_seed_epoch_ptr()), 0u, 0);
  return dgdm_launch_status();
}

extern "C" int dgdm_seed_epoch_set(uint32_t value, void* stream) {
  hipLaunchKernelGGL(k_seed_epoch, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), const_cast<uint32_t*>(dgdm_seed_epoch_ptr()), value, 1);
  return dgdm_launch_status();
}

// p[0 .. n) <- value (uint32 words); a kernel node, so it re-executes in a graph replay (see common.hpp, dgdm_fill_async)
extern "C" int dgdm_fill_u32(uint32_t* p, int64_t n, uint32_t value, void* stream) {
  if (n < 0 || (n > 0 && !p)) return DGDM_ERR_INVALID_ARG;
  if (n == 0) return DGDM_OK;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_dgdm_fill32, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, static_cast<hipStream_t>(stream), p,
                     (size_t)n, value);
  return dgdm_launch_status();
}
