// core.hip — ABI identification for libclipk.so.
#include "common.h"

extern "C" int clipk_version(void) { return CLIPK_ABI_VERSION; }
extern "C" const char* clipk_arch(void) { return "gfx950"; }
extern "C" const char* clipk_status_string(int status) {
  switch (status) {
    case CLIPK_OK: return "ok";
    case CLIPK_ERR_BAD_ARG: return "bad argument (null / misaligned pointer or non-positive dimension)";
    case CLIPK_ERR_UNSUPPORTED: return "unsupported shape for the gfx950 kernels";
    case CLIPK_ERR_LAUNCH: return "HIP launch error";
    default: return "unknown clipk status";
  }
}
