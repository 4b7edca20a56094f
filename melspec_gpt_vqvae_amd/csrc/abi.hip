// ABI bookkeeping entry points of libmelgpt_hip.so.
#include <atomic>
#include <cstdlib>

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

// K loop of the persistent GEMM: 1 = the ping-pong loop of gemm8p.hip where it is built, 0 = the ring of gemm256.hip only.
static int g_pingpong = -1;
static std::atomic<long long> g_loop_launches[2];
extern "C" int melgpt_set_gemm_pingpong(int on) {
  g_pingpong = on != 0;
  return MELGPT_OK;
}
extern "C" int melgpt_get_gemm_pingpong(void) {
  if (g_pingpong < 0) {
    const char* e = getenv("MELGPT_GEMM_8P");
    g_pingpong = e ? atoi(e) != 0 : 1;
  }
  return g_pingpong;
}
extern "C" int melgpt_gemm_loop_launches(long long* ring, long long* pingpong) {
  if (!ring || !pingpong) return MELGPT_ERR_BAD_ARG;
  *ring = g_loop_launches[0].load();
  *pingpong = g_loop_launches[1].load();
  return MELGPT_OK;
}
void melgpt_count_gemm_loop(int pingpong) { g_loop_launches[pingpong != 0].fetch_add(1); }  // (launchers of the two kernels)
