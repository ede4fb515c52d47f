// Library identification and error text of the gaiaseg_hip C-ABI.
#include "common.h"

extern "C" int gs_abi_version(void) { return 2; }

extern "C" const char* gs_target_arch(void) { return "gfx950"; }

extern "C" const char* gs_error_string(int code) {
  switch (code) {
    case GS_OK: return "success";
    case GS_E_BADARG: return "gaiaseg_hip: inconsistent or unsupported descriptor";
    case GS_E_ALIGN: return "gaiaseg_hip: pointer, channel count or stride not aligned as documented";
    case GS_E_WORKSPACE: return "gaiaseg_hip: workspace missing or too small";
    case GS_E_NULL: return "gaiaseg_hip: required pointer is NULL";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "gaiaseg_hip: unknown error code";
}
