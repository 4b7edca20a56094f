// Development harness: phase stamps of the mel frontend kernel (mel_block_kernel) on synthetic clips.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DMEL_LAB=1 -I include -I melspec_gpt_vqvae_amd/csrc
//        tools/lab/mel_lab.hip -o tools/lab/bin/mel_lab
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
#include "../../melspec_gpt_vqvae_amd/csrc/mel.hip"

int main(int argc, char** argv) {
  const int clips = argc > 1 ? atoi(argv[1]) : 64, nm = 80;
  const long long L = 220500;
  std::vector<float> w((size_t)clips * L), basis((size_t)nm * 513, 0.f);
  std::vector<int> lo(nm), hi(nm);
  unsigned s = 1;
  for (auto& v : w) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
  {  // Slaney mel filters as librosa.filters.mel(22050, 1024, 80, 125, 7600) builds them (htk=False, norm='slaney')
    auto hz2mel = [](double f) { return f < 1000.0 ? f / (200.0 / 3) : 15.0 + log(f / 1000.0) / (log(6.4) / 27.0); };
    auto mel2hz = [](double m) { return m < 15.0 ? m * (200.0 / 3) : 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0)); };
    std::vector<double> pts(nm + 2);
    for (int i = 0; i < nm + 2; ++i) pts[i] = mel2hz(hz2mel(125.0) + (hz2mel(7600.0) - hz2mel(125.0)) * i / (nm + 1));
    for (int m = 0; m < nm; ++m) {
      lo[m] = 513; hi[m] = -1;
      for (int k = 0; k < 513; ++k) {
        const double f = k * 22050.0 / 1024.0;
        const double up = (f - pts[m]) / (pts[m + 1] - pts[m]), dn = (pts[m + 2] - f) / (pts[m + 2] - pts[m + 1]);
        const double wv = fmax(0.0, fmin(up, dn)) * 2.0 / (pts[m + 2] - pts[m]);
        basis[(size_t)m * 513 + k] = (float)wv;
        if (wv > 0) { if (k < lo[m]) lo[m] = k; hi[m] = k; }
      }
      if (hi[m] < 0) lo[m] = 0;
    }
  }
  float *dw, *db, *dmel; int *dlo, *dhi; void* dtile;
  hipMalloc(&dw, w.size() * 4); hipMalloc(&db, basis.size() * 4); hipMalloc(&dlo, nm * 4); hipMalloc(&dhi, nm * 4);
  hipMalloc(&dmel, (size_t)clips * nm * 860 * 4); hipMalloc(&dtile, (size_t)clips * nm * 848 * 2);
  hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, basis.data(), basis.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dlo, lo.data(), nm * 4, hipMemcpyHostToDevice); hipMemcpy(dhi, hi.data(), nm * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    int st = 0;
    for (int i = 0; i < 5; ++i)
      st |= melgpt_mel_frontend_fwd(dw, clips, L, 1024, 256, db, dlo, dhi, nm, 1e-5f, 20.f, 20.f, 100.f, 100.f, 0.f, 1.f, dmel, 860,
                                    dtile, MELGPT_BF16, 6, 848, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("status %d  %.1f us per call, %.0f clips/s\n", st, ms * 200, clips / (ms / 5) * 1e3);
  }
#if MEL_LAB
  unsigned long long h[64];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(melgpt_mel_dbg), sizeof(h));
  for (int i = 0; i < 7; ++i)
    printf("frame slot %llu: loop top at +%llu (gap %llu)  window+issue next %llu  fft %llu  unpack %llu   last block filters+write %llu\n",
           h[8 * i + 6], h[8 * i], h[8 * i + 5], h[8 * i + 1], h[8 * i + 2], h[8 * i + 3], h[8 * i + 4]);
  printf("setup: band table + tile plan %llu, dense filter blocks %llu, window + twiddles %llu\n", h[52], h[53], h[54]);
  printf("block phase per wave: after MFMAs of its last tile %llu %llu %llu %llu;  done %llu %llu %llu %llu\n", h[56], h[57], h[58], h[59],
         h[60], h[61], h[62], h[63]);
#endif
  return 0;
}
