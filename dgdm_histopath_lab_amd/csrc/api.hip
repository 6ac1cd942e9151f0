// ABI bookkeeping for libdgdm_hip.so.
#include "common.hpp"

extern "C" int dgdm_abi_version(void) { return 1; }

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
