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

// Compute units left free by the persistent kernels (gemm256_kernel, conv3x3_gn_wide_kernel: one workgroup per CU, each
// walking a tile list for the whole launch).  Data-parallel training overlaps RCCL all-reduce kernels with the backward
// GEMMs; a persistent workgroup that finds its CU taken would start only when another one has finished its whole
// list.  With `n` CUs reserved the persistent grids use (CUs - n) workgroups, so RCCL's channels always find room.
static int g_reserved_cus = 0;
extern "C" int melgpt_set_reserved_cus(int n) {
  if (n < 0 || n > 128) return MELGPT_ERR_BAD_ARG;
  g_reserved_cus = n;
  return MELGPT_OK;
}
extern "C" int melgpt_get_reserved_cus(void) { return g_reserved_cus; }
