// Library identity + error strings.
#include "dm_common.h"

extern "C" const char *dm_version(void) { return "detmatch_hip 0.1 (gfx950)"; }

extern "C" const char *dm_error_string(int code) {
  switch (code) {
    case DM_OK: return "ok";
    case DM_ERR_INVALID_ARG: return "invalid argument";
    case DM_ERR_WORKSPACE: return "workspace too small";
    case DM_ERR_INT32_RANGE: return "batch * volume exceeds the int32 cell-id range";
    case DM_ERR_UNSUPPORTED: return "unsupported channel count / kernel volume";
    case DM_ERR_LAUNCH: return "HIP launch / runtime error";
    default: return "unknown error";
  }
}
