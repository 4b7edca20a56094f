// ABI bookkeeping entry points of libmelgpt_hip.so.
#include "common.h"

extern "C" int melgpt_abi_version(void) { return MELGPT_ABI_VERSION; }

extern "C" const char* melgpt_strerror(int code) {
  switch (code) {
    case MELGPT_OK: return "ok";
    case MELGPT_ERR_BAD_ARG: return "bad argument (null pointer or non-positive size)";
    case MELGPT_ERR_UNSUPPORTED: return "unsupported shape or dtype";
    case MELGPT_ERR_LAUNCH: return "HIP launch failed";
    case MELGPT_ERR_ALIGN: return "pointer or stride misaligned";
    default: return "unknown melgpt error";
  }
}
