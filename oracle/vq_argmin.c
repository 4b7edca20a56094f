/* oracle/vq_argmin.c - CPU restatement of the nearest-neighbour codebook lookup.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): never linked into or called by the product.
 *
 * Follows /root/reference/vqvae/big_model_attn_gan.py:
 *   :28-30  d[n][k] = (sum_c x^2 + sum_c e^2) - 2 * sum_c x*e        (fp32, this evaluation order)
 *   :33     argmin over k, FIRST minimal index on exact ties (torch.argmin)
 *   :40     quantized = E[idx];  :43-45 mse;  :49 forward value x + (q - x)
 *
 * The fp32 summation ORDER is fixed here (and mirrored by the F32 lane of the HIP kernel, whose
 * cross term runs on v_mfma_f32_16x16x4_f32 = a k-ordered chain of single-rounded f32 FMAs):
 *   x.e   : one FMA chain over c = 0..D-1 in natural order, starting from 0
 *   |x|^2 , |e|^2 : four FMA chains over the quarters [64j, 64j+64), combined (p0+p1)+(p2+p3)
 * so the HIP F32 lane and this file agree BIT FOR BIT on every distance and every index.
 * Against the reference (whose BLAS picks its own order) indices can differ only where the two
 * smallest distances are within a few ulp; the fixtures list those vectors (tests/golden).
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -o _build/liboracle_vq.so vq_argmin.c -lm
 */
#include <math.h>
#include <stdint.h>

static float sumsq_quarters(const float* v, int D) {
  float p[4];
  int q = D / 4;
  for (int j = 0; j < 4; ++j) {
    float acc = 0.0f;
    for (int c = 0; c < q; ++c) {
      float x = v[q * j + c];
      acc = fmaf(x, x, acc);
    }
    p[j] = acc;
  }
  return (p[0] + p[1]) + (p[2] + p[3]);
}

/* z: (N, D) row-major f32; codebook: (K, D) f32, K <= 1024.  distances (N,K), quantized (N,D) optional. */
void oracle_vq_argmin_f32(const float* z, long N, int D, const float* codebook, int K, int64_t* indices,
                          float* distances, float* quantized, double* sq_err_out) {
  float bsq[1024];
  for (int k = 0; k < K; ++k) bsq[k] = sumsq_quarters(codebook + (long)k * D, D);
  double sq = 0.0;
  for (long n = 0; n < N; ++n) {
    const float* x = z + n * D;
    float A = sumsq_quarters(x, D);
    float best = INFINITY;
    int bk = 0;
    for (int k = 0; k < K; ++k) {
      const float* e = codebook + (long)k * D;
      float m = 0.0f;
      for (int c = 0; c < D; ++c) m = fmaf(e[c], x[c], m);
      float ab = A + bsq[k];
      float d = ab - 2.0f * m; /* 2*m is exact, so fusing this into one FMA cannot change d */
      if (distances) distances[n * K + k] = d;
      if (d < best) {
        best = d;
        bk = k;
      }
    }
    indices[n] = bk;
    const float* e = codebook + (long)bk * D;
    for (int c = 0; c < D; ++c) {
      float dq = e[c] - x[c];
      sq += (double)dq * (double)dq;
      if (quantized) quantized[n * D + c] = x[c] + dq;
    }
  }
  if (sq_err_out) *sq_err_out = sq;
}
