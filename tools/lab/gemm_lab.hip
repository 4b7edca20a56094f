// Development harness: times gemm256_kernel variants (G256_LAB: 0 full, 1 no LDS-DMA refills in the loop,
// 2 no fragment reads / MFMAs, 3 no epilogue) to separate the load path, the LDS+MFMA path and the epilogue.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DG256_LAB=n -I include -I melspec_gpt_vqvae_amd/csrc tools/lab/gemm_lab.hip
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../melspec_gpt_vqvae_amd/csrc/gemm256.hip"


static int g_pad = 0;  // extra elements in the leading dimension of row-major operands (address-stride experiments)
static float run(int alay, int blay, int M, int N, int K, int cfg, int iters, int full = 0) {
  void *A, *B, *C;
  const size_t lda = alay == LAY_KMAJ ? M : K + g_pad, ldb = blay == LAY_KMAJ ? N : K + g_pad;
  const size_t an = (alay == LAY_KMAJ ? (size_t)K : (size_t)M) * lda, bn = (blay == LAY_KMAJ ? (size_t)K : (size_t)N) * ldb;
  hipMalloc(&A, an * 2); hipMalloc(&B, bn * 2); hipMalloc(&C, (size_t)M * N * 2);
  hipMemset(A, 0x3c, an * 2); hipMemset(B, 0x3c, bn * 2);
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = N; p.ldr = N;
  p.a_bytes = (unsigned)(an * 2); p.b_bytes = (unsigned)(bn * 2);
  p.alpha = 1.f; p.vec_io = 1;
  void *C2 = nullptr, *R = nullptr; float* bias = nullptr;
  if (!full && getenv("BIAS")) {  // the plain modes with a bias vector (what the step's Linear layers pass)
    hipMalloc(&bias, (size_t)N * 4);
    hipMemset(bias, 0, (size_t)N * 4);
    p.bias = bias;
  }
  if (full) {
    hipMalloc(&C2, (size_t)M * N * 2); hipMalloc(&R, (size_t)M * N * 2); hipMalloc(&bias, (size_t)N * 4);
    hipMemset(R, 0, (size_t)M * N * 2); hipMemset(bias, 0, (size_t)N * 4);
    p.bias = bias;
    if (full == 1) { p.act = MELGPT_ACT_GELU; p.C2 = C2; }
    if (full == 2) { p.R = R; p.drop_scale = 2.f; p.drop_thresh = 32768; p.seed = 7; }
    if (full == 3) { p.R = R; p.act = MELGPT_ACT_GELU_GRAD; p.drop_scale = 2.f; p.drop_thresh = 32768; p.seed = 7; }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_gemm256(p, alay, blay, 1, cfg, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch_gemm256(p, alay, blay, 1, cfg, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(A); hipFree(B); hipFree(C); hipFree(C2); hipFree(R); hipFree(bias);
  return ms / iters;
}

static void trace(int alay, int blay, int M, int N, int K, int full = 0) {
  void *A, *B, *C; unsigned long long* dbg;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
  hipMalloc(&dbg, 4096); hipMemset(dbg, 0, 4096);
  hipMemset(A, 0x3c, (size_t)M * K * 2); hipMemset(B, 0x3c, (size_t)N * K * 2);
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
  p.lda = alay == LAY_KMAJ ? M : K; p.ldb = blay == LAY_KMAJ ? N : K; p.ldc = N; p.ldr = N;
  p.a_bytes = (unsigned)((size_t)M * K * 2); p.b_bytes = (unsigned)((size_t)N * K * 2);
  p.alpha = 1.f; p.vec_io = 1;
#if G256_LAB & 8
  hipMemcpyToSymbol(HIP_SYMBOL(g256_dbg), &dbg, sizeof(dbg));
#endif
  void* R = nullptr; float* bias = nullptr;
  if (full) {
    hipMalloc(&R, (size_t)M * N * 2); hipMalloc(&bias, (size_t)N * 4);
    hipMemset(R, 0, (size_t)M * N * 2); hipMemset(bias, 0, (size_t)N * 4);
    p.bias = bias;
    if (full == 1) p.act = MELGPT_ACT_GELU;
    if (full == 2) { p.R = R; p.drop_scale = 2.f; p.drop_thresh = 32768; p.seed = 7; }
    if (full == 3) { p.R = R; p.act = MELGPT_ACT_GELU_GRAD; p.drop_scale = 2.f; p.drop_thresh = 32768; p.seed = 7; }
    if (full == 4 || full == 14) p.R = R;
    if (full == 5) { p.drop_scale = 2.f; p.drop_thresh = 32768; p.seed = 7; }
  }
  for (int rep = 0; rep < 2; ++rep) {
    if (full == 2 && getenv("DROPR")) launch_mode<LAY_ROW, LAY_ROW, EPI_DROPR16>(p, 1, 0);  // the lean mode of the same epilogue
    else if (full && full < 10) launch_mode<LAY_ROW, LAY_ROW, EPI_FULL16>(p, 1, 0);
    else launch_mode<LAY_ROW, LAY_ROW, EPI_PLAIN16>(p, 1, 0);
  }
  hipDeviceSynchronize();
  unsigned long long h[64];
  hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 10 && h[4 * i]; ++i)
    printf("tile %d: loop %6llu  drain+barrier %6llu  epilogue %6llu   (gap to next start %6lld) [memtime ticks]\n", i,
           h[4 * i + 1] - h[4 * i], h[4 * i + 2] - h[4 * i + 1], h[4 * i + 3] - h[4 * i + 2],
           h[4 * i + 4] ? (long long)(h[4 * i + 4] - h[4 * i + 3]) : -1LL);
  hipFree(A); hipFree(B); hipFree(C); hipFree(dbg);
}

int main(int argc, char** argv) {
  if (G256_LAB & 8) {
    printf("fc1\n"); trace(LAY_ROW, LAY_ROW, 33920, 4096, 1024);
    printf("fc1 drop+R\n"); trace(LAY_ROW, LAY_ROW, 33920, 4096, 1024, 2);
    printf("fc1 drop*gelu'\n"); trace(LAY_ROW, LAY_ROW, 33920, 4096, 1024, 3);
    printf("fc1 FULL16 bias only\n"); trace(LAY_ROW, LAY_ROW, 33920, 4096, 1024, 6);
    printf("fc1 FULL16 +R\n"); trace(LAY_ROW, LAY_ROW, 33920, 4096, 1024, 4);
    printf("fc1 FULL16 drop\n"); trace(LAY_ROW, LAY_ROW, 33920, 4096, 1024, 5);
    printf("fc1 PLAIN16 +R\n"); trace(LAY_ROW, LAY_ROW, 33920, 4096, 1024, 14);
    printf("proj (N = K = 1024) drop+R, as the step runs it\n"); trace(LAY_ROW, LAY_ROW, 33920, 1024, 1024, 2);
    printf("fc2 (N = 1024, K = 4096) drop+R, as the step runs it\n"); trace(LAY_ROW, LAY_ROW, 33920, 1024, 4096, 2);
    printf("proj plain (N = K = 1024)\n"); trace(LAY_ROW, LAY_ROW, 33920, 1024, 1024);
    printf("sq8k\n"); trace(LAY_ROW, LAY_ROW, 8192, 8192, 8192);
    return 0;
  }
  struct S { const char* name; int al, bl, M, N, K; } shapes[] = {
      {"nt fc1 ", LAY_ROW, LAY_ROW, 33920, 4096, 1024}, {"nt fc2 ", LAY_ROW, LAY_ROW, 33920, 1024, 4096},
      {"nt qkv ", LAY_ROW, LAY_ROW, 33920, 3072, 1024}, {"nt proj", LAY_ROW, LAY_ROW, 33920, 1024, 1024},
      {"nn fc2d", LAY_ROW, LAY_KMAJ, 33920, 4096, 1024}, {"nn fc1d", LAY_ROW, LAY_KMAJ, 33920, 1024, 4096},
      {"tn wfc1", LAY_KMAJ, LAY_KMAJ, 4096, 4096, 8192},
      {"nt sq8k", LAY_ROW, LAY_ROW, 8192, 8192, 8192}};
  for (int pad : {0, 64, 8})
    for (auto& s : shapes) {
      g_pad = pad;
      float ms = run(s.al, s.bl, s.M, s.N, s.K, 3, 10);
      printf("lab=%d pad=%2d %s  %8.3f ms  %7.1f TFLOP/s\n", G256_LAB, pad, s.name, ms, 2.0 * s.M * s.N * s.K / ms / 1e9);
    }
  g_pad = 0;
  if (G256_LAB == 0) {
    printf("fc1 +bias+gelu+C2     %8.3f ms\n", run(LAY_ROW, LAY_ROW, 33920, 4096, 1024, 3, 10, 1));
    printf("proj +bias+drop+R     %8.3f ms\n", run(LAY_ROW, LAY_ROW, 33920, 1024, 1024, 3, 10, 2));
    printf("fc2 +bias+drop+R      %8.3f ms\n", run(LAY_ROW, LAY_ROW, 33920, 1024, 4096, 3, 10, 2));
    printf("dfc2 nn +drop*gelu'   %8.3f ms\n", run(LAY_ROW, LAY_KMAJ, 33920, 4096, 1024, 3, 10, 3));
  }
  return 0;
}
// stand-ins for the library entry points gemm256.hip references (abi.hip is not linked into this harness)
extern "C" int melgpt_get_reserved_cus(void) { return 0; }
extern "C" int melgpt_get_dynamic_tiles(void) { return getenv("DYN") != nullptr; }
extern "C" int* melgpt_tile_cell(void) {
  static int* pool = nullptr;
  static unsigned seq = 0;
  if (!pool) {
    (void)hipMalloc(&pool, (size_t)64 * MELGPT_TILE_CELL_INTS * sizeof(int));
    (void)hipMemset(pool, 0, (size_t)64 * MELGPT_TILE_CELL_INTS * sizeof(int));
    (void)hipDeviceSynchronize();
  }
  return pool + (size_t)(seq++ % 64) * MELGPT_TILE_CELL_INTS;
}
