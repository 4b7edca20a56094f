// ABI bookkeeping entry points of libmelgpt_hip.so.
#include <atomic>
#include <cstdlib>
#include <mutex>

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

// Claimed tiles for the persistent GEMM (gemm256.hip): 1 = every tile is drawn from a counter at run time instead of
// a static per-workgroup list (data-parallel runs: a workgroup displaced by an RCCL kernel no longer strands a list).
static int g_dynamic_tiles = -1;
extern "C" int melgpt_set_dynamic_tiles(int on) {
  g_dynamic_tiles = on != 0;
  return MELGPT_OK;
}
extern "C" int melgpt_get_dynamic_tiles(void) {
  if (g_dynamic_tiles < 0) {
    const char* e = getenv("MELGPT_DYNAMIC_TILES");
    g_dynamic_tiles = e ? atoi(e) != 0 : 0;
  }
  return g_dynamic_tiles;
}

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

// Scheduler cells of the claimed-tile launches: MELGPT_TILE_CELL_INTS ints each (counter + per-workgroup mailboxes), handed
// out round-robin from one pool so that launches in flight on different streams never share a counter; a cell puts its
// counter back to 0 with the launch's last draw.  The pool is allocated on first use and zeroed BEFORE the first cell is
// handed out - the fill is waited for with a device synchronisation: a launch that started on a recycled allocation's
// old contents would draw wild tickets.  nullptr (static tile lists are used instead) if that first use falls into a
// stream capture, where neither the allocation nor the wait is allowed.
extern "C" int* melgpt_tile_cell(void) {
  constexpr int NCELL = 256;
  static int* pool = nullptr;
  static std::atomic<unsigned> seq{0};
  static std::mutex mu;
  if (!pool) {
    std::lock_guard<std::mutex> lk(mu);
    if (!pool) {
      int* q = nullptr;
      const size_t bytes = (size_t)NCELL * MELGPT_TILE_CELL_INTS * sizeof(int);
      if (hipMalloc(&q, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
      }
      if (hipMemset(q, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(q);
        return nullptr;
      }
      pool = q;
    }
  }
  return pool + (size_t)(seq.fetch_add(1) % NCELL) * MELGPT_TILE_CELL_INTS;
}
