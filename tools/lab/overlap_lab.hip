// Development harness: does a concurrent collective-shaped kernel stall the persistent GEMM's STATIC
// tile lists, and what do claimed tiles / reserved CUs buy?  One process, two streams:
//   stream A: the backward GEMM chain of one minGPT Block at the training shape (through the library's C ABI, so the
//             kernels and their scheduling are exactly the product's);
//   stream B: a STAND-IN for an RCCL all-reduce kernel - G persistent 256-thread workgroups ("channels") that each stream
//             their share of a 50 MB buffer through registers (read, add, write) for as long as the chain runs.  RCCL
//             itself cannot be used on one GPU: a group of one rank launches no kernel, and two ranks on one device are
//             refused.  What matters for the question is the footprint - a few long-lived workgroups that hold CUs the
//             persistent grid (one 160 KiB-LDS, 512-register workgroup per CU) would otherwise own - and that is modelled.
// Settings x {no collective, collective}: static lists with 0 / 8 / 16 / 32 reserved CUs (always / around the weight gradients only),
// both K loops.  (Claimed tiles - melgpt_set_dynamic_tiles, the ring kernel's atomic tile counter - were removed from the library in
// round 6: the last commit that has them and this rig's arms for them is 2e241d9.)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/lab/overlap_lab.hip -L melspec_gpt_vqvae_amd/lib
//        -lmelgpt_hip -Wl,-rpath,'$ORIGIN/../../../melspec_gpt_vqvae_amd/lib' -o tools/lab/bin/overlap_lab
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "melgpt.h"

__global__ __launch_bounds__(256) void channel_kernel(const float4* src, float4* dst, long long n4, int rounds,
                                                      volatile int* stop) {
  // each "channel" owns a contiguous share, walks it `rounds` times (or until told to stop)
  const long long per = (n4 + gridDim.x - 1) / gridDim.x, lo = (long long)blockIdx.x * per;
  const long long hi = lo + per < n4 ? lo + per : n4;
  for (int r = 0; r < rounds; ++r) {
    if (*stop) break;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
      float4 a = src[i], b = dst[i];
      dst[i] = float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Bufs {
  void *x, *y4, *y1, *w1, *w2, *g4, *g1;
  float *dw1, *dw2;
};

static int gemm(const void* A, int ak, long long lda, const void* B, int bk, long long ldb, void* C, long long ldc, int M,
                int N, int K, int batch, long long sA, long long sB, long long sC, int out_f32, hipStream_t s) {
  return melgpt_gemm(A, ak, lda, sA, B, bk, ldb, sB, C, ldc, sC, M, N, K, batch, MELGPT_BF16, out_f32, 0, 1.0f, nullptr,
                     MELGPT_ACT_NONE, nullptr, 0, 0, nullptr, 0.f, 0, 0, s);
}

// the MLP half of one Block's backward: dgrad fc2 (NN, K = 1024), dgrad fc1 (NN, K = 4096), the two weight gradients
// (TN, split into 4 batches of 8480 rows as ops._wgrad_split does) - 4 launches, ~1.1 ms
// (g_ns = split-K batches of the weight gradients: 4 x 64 tiles fill 256 CUs exactly once - and take two rounds on 240 -,
// 5 x 64 = 320 tiles are 1.25 / 1.33 rounds; ops._wgrad_split picks by the workgroups a launch gets.  0 = dgrads only.)
static int g_ns = 4;
static int g_wg_res = 0;  // CUs reserved around the WEIGHT-GRADIENT launches only (the single-round launches are the ones that stall)
static void chain(const Bufs& b, hipStream_t s, int reps) {
  const int M = 33920, C = 1024, F = 4096, ns = g_ns;
  for (int r = 0; r < reps; ++r) {
    int st = 0;
    st |= gemm(b.g1, 0, C, b.w2, 1, F, b.g4, F, M, F, C, 1, 0, 0, 0, 0, s);                       // dY(M,C) @ W2(C,F) -> (M,F)
    st |= gemm(b.g4, 0, F, b.w1, 1, C, b.g1, C, M, C, F, 1, 0, 0, 0, 0, s);                       // (M,F) @ W1(F,C) -> (M,C)
    if (ns > 0) {
      if (g_wg_res) melgpt_set_reserved_cus(g_wg_res);
      st |= gemm(b.g4, 1, F, b.x, 1, C, b.dw1, C, F, C, M / ns, ns, (long long)(M / ns) * F, (long long)(M / ns) * C,
                 (long long)F * C, 1, s);                                                        // dW1 parts (ns,F,C)
      st |= gemm(b.g1, 1, C, b.y4, 1, F, b.dw2, F, C, F, M / ns, ns, (long long)(M / ns) * C, (long long)(M / ns) * F,
                 (long long)C * F, 1, s);                                                        // dW2 parts (ns,C,F)
      if (g_wg_res) melgpt_set_reserved_cus(0);
    }
    if (st) { printf("melgpt_gemm failed: %d\n", st); exit(1); }
  }
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 32;  // channels of the stand-in collective
  g_ns = argc > 2 ? atoi(argv[2]) : 4;
  const int M = 33920, C = 1024, F = 4096, REPS = 20;
  Bufs b{};
  CK(hipMalloc(&b.x, (size_t)M * C * 2)); CK(hipMalloc(&b.y4, (size_t)M * F * 2)); CK(hipMalloc(&b.g4, (size_t)M * F * 2));
  CK(hipMalloc(&b.g1, (size_t)M * C * 2)); CK(hipMalloc(&b.y1, (size_t)M * C * 2));
  CK(hipMalloc(&b.w1, (size_t)F * C * 2)); CK(hipMalloc(&b.w2, (size_t)C * F * 2));
  CK(hipMalloc(&b.dw1, (size_t)5 * F * C * 4)); CK(hipMalloc(&b.dw2, (size_t)5 * F * C * 4));
  // small random-ish bf16 fill (0x3c.. = ~0.01 .. 0.03): the data does not matter for the scheduling question
  std::vector<unsigned short> h((size_t)M * F);
  unsigned sd = 12345u;
  for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u | ((sd >> 20) & 0xffu) | ((sd >> 3) & 0x8000u)); }
  CK(hipMemcpy(b.x, h.data(), (size_t)M * C * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(b.y4, h.data(), (size_t)M * F * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(b.g1, h.data(), (size_t)M * C * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(b.w1, h.data(), (size_t)F * C * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(b.w2, h.data(), (size_t)F * C * 2, hipMemcpyHostToDevice));
  const long long n4 = 50ll * 1000 * 1000 / 16;
  float4 *cs, *cd; int* stop;
  CK(hipMalloc(&cs, n4 * 16)); CK(hipMalloc(&cd, n4 * 16)); CK(hipMemset(cs, 0, n4 * 16)); CK(hipMemset(cd, 0, n4 * 16));
  CK(hipHostMalloc(&stop, sizeof(int), hipHostMallocMapped)); *stop = 0;
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1, c0, c1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));

  // (ns_for[] = the weight gradients' split-K
  // batches ops._wgrad_split picks for the workgroups such a launch gets: 4 on 256, 5 on 248 / 240 CUs)
  struct Mode { const char* name; int pp, reserve, wg_res; } modes[] = {
      {"pingpong, static lists", 1, 0, 0},        {"pingpong, static, 8 reserved CUs", 1, 8, 0},
      {"pingpong, static, 16 reserved CUs", 1, 16, 0}, {"pingpong, static, 32 reserved CUs", 1, 32, 0},
      {"pingpong, static, 16 reserved for wgrads only", 1, 0, 16}, {"pingpong, static, 32 reserved for wgrads only", 1, 0, 32},
      {"ring, static lists", 0, 0, 0}};
  const int ns_arg = g_ns;
  printf("{\"bench\": \"backward MLP GEMM chain (2 dgrads%s) x %d beside a stand-in collective of %d channels x 256 threads over 50 MB\", \"wgrad_batches\": %d, \"rows\": [\n", g_ns ? " + 2 wgrads" : "", REPS, G, g_ns);
  for (int round = 0; round < 2; ++round)
    for (auto& m : modes) {
      melgpt_set_gemm_pingpong(m.pp);
      melgpt_set_reserved_cus(m.reserve);
      g_wg_res = m.wg_res;
      if (ns_arg < 0) g_ns = (m.reserve || m.wg_res) ? 5 : 4;   // (-1: the split the host picks for the grid the launch gets)
      chain(b, sa, 2);
      CK(hipStreamSynchronize(sa));
      float alone, beside, coll;
      CK(hipEventRecord(e0, sa)); chain(b, sa, REPS); CK(hipEventRecord(e1, sa)); CK(hipStreamSynchronize(sa));
      CK(hipEventElapsedTime(&alone, e0, e1));
      // the collective first (it is resident when the chain's launches arrive, as an all-reduce launched from a Block's hook is)
      *stop = 0;
      CK(hipEventRecord(c0, sb));
      hipLaunchKernelGGL(channel_kernel, dim3(G), dim3(256), 0, sb, cs, cd, n4, 100000, stop);
      CK(hipEventRecord(c1, sb));
      CK(hipEventRecord(e0, sa)); chain(b, sa, REPS); CK(hipEventRecord(e1, sa)); CK(hipStreamSynchronize(sa));
      *stop = 1;
      CK(hipStreamSynchronize(sb));
      CK(hipEventElapsedTime(&beside, e0, e1)); CK(hipEventElapsedTime(&coll, c0, c1));
      printf("  {\"mode\": \"%s\", \"wgrad_batches\": %d, \"round\": %d, \"chain_alone_ms\": %.3f, \"chain_beside_collective_ms\": %.3f, \"ratio\": %.3f, \"collective_resident_ms\": %.2f},\n",
             m.name, g_ns, round, alone / REPS, beside / REPS, beside / alone, coll);
    }
  melgpt_set_reserved_cus(0);
  melgpt_set_gemm_pingpong(1);
  printf("  {}]}\n");
  return 0;
}
