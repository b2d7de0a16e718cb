// Version and error-string entry points of libdepthwarp.
#include "dw_common.h"

extern "C" int dw_version(void) { return DW_VERSION; }

extern "C" const char* dw_strerror(int code) {
  switch (code) {
    case DW_OK: return "ok";
    case DW_ERR_NULL: return "a required pointer is NULL";
    case DW_ERR_SHAPE: return "unsupported or inconsistent dimensions";
    case DW_ERR_WORKSPACE: return "workspace too small or not 256-byte aligned";
    case DW_ERR_LAUNCH: return "kernel launch failed";
    case DW_ERR_ARG: return "bad flag or enum value";
    case DW_ERR_TOO_LARGE:
      return "problem too large for the fixed scratch array of the allocation-free entry point "
             "(use the variant that takes a workspace)";
    default: return "unknown error code";
  }
}
