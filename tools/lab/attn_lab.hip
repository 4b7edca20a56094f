// Development harness: the fused attention kernels (attn_q_kernel forward / dQ, attn_dkv_kernel) at the training shape,
// with compile-time ablations -DATTN_LAB=n (see attn.hip) to split staging, per-tile operand fetches and the math.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast [-DATTN_LAB=n] -I include -I melspec_gpt_vqvae_amd/csrc
//        tools/lab/attn_lab.hip -o tools/lab/bin/attn_lab
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../melspec_gpt_vqvae_amd/csrc/attn.hip"

static unsigned short f2bf(float f) {
  unsigned u; std::memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
}

__global__ void spin_kernel(unsigned long long ticks, unsigned long long* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long n = 0;
  float x = threadIdx.x;
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) {
#pragma unroll
    for (int i = 0; i < 64; ++i) x = x * 1.0001f + 0.5f;  // 64 dependent VALU ops
    n += 64;
  }
  if (threadIdx.x == 0) { out[0] = n; out[1] = (unsigned long long)x; }
}

int main(int argc, char** argv) {
  {  // what s_memtime counts: spin for 2e8 ticks on one wave and compare with the wall clock
    unsigned long long* d; hipMalloc(&d, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0); spin_kernel<<<1, 64>>>(200000000ull, d); hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("s_memtime: 2e8 ticks in %.2f ms -> %.1f MHz; %llu dependent VALU ops -> %.2f ns each\n", ms, 2e5 / ms, h[0], ms * 1e6 / h[0]);
  }
  const int B = argc > 1 ? atoi(argv[1]) : 128, H = 16, T = argc > 2 ? atoi(argv[2]) : 265, C = H * 64;
  const float pdrop = argc > 3 ? atof(argv[3]) : 0.5f;
  const size_t rows = (size_t)B * T;
  std::vector<unsigned short> hq(rows * 3 * C), hd(rows * C);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
  for (auto& v : hq) v = f2bf(rnd());
  for (auto& v : hd) v = f2bf(rnd() * 0.1f);
  void *qkv, *o, *dout, *dqkv; float *lse, *delta;
  hipMalloc(&qkv, rows * 3 * C * 2); hipMalloc(&dqkv, rows * 3 * C * 2); hipMalloc(&o, rows * C * 2); hipMalloc(&dout, rows * C * 2);
  hipMalloc(&lse, (size_t)B * H * T * 4); hipMalloc(&delta, (size_t)B * H * T * 4);
  hipMemcpy(qkv, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dout, hd.data(), hd.size() * 2, hipMemcpyHostToDevice);
  hipMemset(dqkv, 0, rows * 3 * C * 2);
  const unsigned short* base = (const unsigned short*)qkv;
  unsigned short* gbase = (unsigned short*)dqkv;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double fl_full = 4.0 * B * H * (double)T * T * 64;
  for (int rep = 0; rep < 3; ++rep) {
    float msf, msb;
    hipEventRecord(e0, 0);
    int st = 0;
    for (int i = 0; i < 10; ++i)
      st |= melgpt_attn_fwd(base, base + C, base + 2 * C, 3 * C, o, C, lse, nullptr, B, H, T, 64, 0, pdrop, 1, 0, MELGPT_BF16, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&msf, e0, e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i)
      st |= melgpt_attn_bwd(base, base + C, base + 2 * C, 3 * C, o, dout, C, lse, delta, gbase, gbase + C, gbase + 2 * C, 3 * C,
                            B, H, T, 64, 0, pdrop, 1, 0, MELGPT_BF16, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&msb, e0, e1);
    printf("status %d  fwd %.1f us (%.0f TFLOP/s causal)   bwd %.1f us (%.0f TFLOP/s causal)\n", st, msf * 100,
           fl_full / 2 / (msf / 10) / 1e9, msb * 100, fl_full * 2.5 / 2 / (msb / 10) / 1e9);
  }
  for (int tt : {T, 224, 192, 128}) {
    int nb = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, attn_q_kernel<bf16_t, false, DM_HALF, false>, NTHREADS, lds_bytes<bf16_t>(tt, false));
    printf("occupancy API: T=%d lds %zu B -> %d workgroups per CU\n", tt, lds_bytes<bf16_t>(tt, false), nb);
  }
#if ATTN_LAB == 8
  unsigned long long hdbg[64];
  hipMemcpyFromSymbol(hdbg, HIP_SYMBOL(melgpt_attn_dbg), sizeof(hdbg));
  {
    std::vector<unsigned long long> w(4 * 4096);
    hipMemcpyFromSymbol(w.data(), HIP_SYMBOL(melgpt_attn_dbg), w.size() * 8, 256 * 8);
    FILE* f = fopen("gpurun_out/attn_wg.csv", "w");
    if (f) {
      fprintf(f, "wg,entry,staged,exit,rt_exit,rt_entry\n");
      for (int i = 0; i < B * H && i < 4096; ++i)
        fprintf(f, "%d,%llu,%llu,%llu,%llu,%llu\n", i, w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3] & 0xFFFFFFFFull, w[4 * i + 3] >> 32);
      fclose(f);
    }
  }
  for (int i = 0; i < 8 && hdbg[8 * i + 4]; ++i)
    printf("job %llu (%llu key tiles): start +%llu  pass1 %llu  pass2 %llu  tail %llu\n", hdbg[8 * i + 5], hdbg[8 * i + 4],
           hdbg[8 * i], hdbg[8 * i + 1], hdbg[8 * i + 2], hdbg[8 * i + 3]);
#endif
#if ATTN_LAB == 9
  {
    std::vector<unsigned long long> w(8 * 8 * 8);
    hipMemcpyFromSymbol(w.data(), HIP_SYMBOL(melgpt_attn_dbg), w.size() * 8);
    for (int it = 0; it < 8; ++it) {
      const unsigned long long t0 = w[(it * 8) * 8];
      printf("item %d (start %+lld after the previous item's end):\n", it, it ? (long long)(t0 - w[((it - 1) * 8) * 8 + 6]) : 0ll);
      for (int wv = 0; wv < 8; ++wv) {
        const unsigned long long* q = &w[(it * 8 + wv) * 8];
        printf("  wave %d: staged %6llu | phase 1 done %6llu (%llu steps) | barrier %6llu | K restaged %6llu | phase 2 done %6llu | end %6llu\n",
               wv, q[1] - t0, q[2] - t0, q[7], q[3] - t0, q[4] - t0, q[5] - t0, q[6] - t0);
      }
    }
  }
#endif
  std::vector<unsigned short> ho(16);
  hipMemcpy(ho.data(), o, 32, hipMemcpyDeviceToHost);
  printf("o[0..3] bits %04x %04x %04x %04x\n", ho[0], ho[1], ho[2], ho[3]);
  return 0;
}
