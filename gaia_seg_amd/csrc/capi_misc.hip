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

// Tuning hook (not part of the operator ABI): force the tile / split-K plan of the next conv calls.
// bm in {64,128}, bn in {32,48,64,80,96,128}, splits >= 1; bm = 0 restores the planner.
namespace gs { int g_force_plan[3] = {0, 0, 0}; }
extern "C" int gs_debug_force_plan(int bm, int bn, int splits) {
  if (bm != 0) {
    const bool bn_ok = bn == 32 || bn == 48 || bn == 64 || bn == 80 || bn == 96 || bn == 128;
    if ((bm != 64 && bm != 128) || !bn_ok || splits < 1) return GS_E_BADARG;
  }
  gs::g_force_plan[0] = bm; gs::g_force_plan[1] = bn; gs::g_force_plan[2] = splits;
  return GS_OK;
}
