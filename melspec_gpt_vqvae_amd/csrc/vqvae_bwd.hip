// Backward of the VQ-VAE's two building blocks (conv3x3_{bwd_data,bwd_weight}, groupnorm_swish_bwd): the gradient
// side of torch.nn.Conv2d(k = 3, s = 1, p = 1) and of Normalize -> nonlinearity (vqvae/big_model_attn_gan.py:85-99,117-127,
// 139-140,164-166), NHWC, both numerics lanes.  No scored configuration trains the VQ-VAE (README.md:16 of the reference: it is
// pre-trained elsewhere), so these - the backward of vqvae/autograd.py's differentiable path through LitVQVAE.forward (:622-634) - are
// CORRECTNESS-first compositions of the forward path's own kernels - the implicit-GEMM
// convolution, the K-major (split-K) GEMM, the fixed-order partial sums - plus three small kernels of their own:
//   dX  = conv3x3(dY, rot180(W)^T)             melgpt_conv2d_nhwc on a repacked weight (Cin, 3, 3, Cout)
//   dW[co][ky][kx][ci] = sum_p dY[p][co] X[p + (ky - 1, kx - 1)][ci]: on ZERO-BORDERED copies (B, H + 2, W + 2, C) of X and dY a
//        filter tap is a constant ROW OFFSET (ky - 1)(W + 2) + (kx - 1) of the flattened pixel index - the zero border of dY
//        kills every product that wraps around an image edge, the zero border of X is the convolution's padding - so a tap's
//        (Cout x Cin) block is ONE K-major GEMM dYp^T Xp[offset:], split over the rows into batches whose f32 partial sums are
//        added in fixed order (deterministic); dbias = column sums of dY.
//   GroupNorm + swish: with h = xhat gamma + beta, y = h sigmoid(h):  dh = dy sigmoid(h) (1 + h (1 - sigmoid(h))),
//        dgamma = sum dh xhat, dbeta = sum dh, and per (image, group) of n elements
//        dx = rstd (dh gamma - (sum_g dh gamma) / n - xhat (sum_g dh gamma xhat) / n): two passes over the tensor, the group
//        sums from per-(image, channel) partial sums (fixed order).
#include <cstdint>

#include "common.h"

namespace {

constexpr int GNB_GROUPS = 32;
constexpr int GNB_CHUNK = 256;  // pixels per partial-sum workgroup

// W (Cout, 3, 3, Cin) -> Wrot (Cin, 3, 3, Cout): Wrot[ci][ky][kx][co] = W[co][2 - ky][2 - kx][ci]
template <typename T>
__global__ void rot180_swap_kernel(const T* __restrict__ w, T* __restrict__ out, int Cout, int Cin) {
  const long long total = (long long)Cout * 9 * Cin;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout), tap = (int)((i / Cout) % 9);  // i indexes the OUTPUT (ci, tap, co)
    const int ci = (int)(i / ((long long)Cout * 9));
    out[i] = w[((long long)co * 9 + (8 - tap)) * Cin + ci];
  }
}

// x (B, H, W, C) -> rows [0, total) of a zero-bordered (B, H + 2, W + 2, C) image that starts `guard` rows into the buffer
// (guard rows, border pixels and the rows behind the last image are zeros); 16-byte pieces
__global__ void pad_border_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ out, int B, int H, int W, int cpr /* pieces per pixel */,
                                  long long guard, long long total_rows) {
  const long long total = total_rows * cpr, PW = W + 2, PH = H + 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / cpr - guard;
    const int c = (int)(i % cpr);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row >= 0 && row < (long long)B * PH * PW) {
      const long long b = row / (PH * PW), r = row - b * PH * PW;
      const int yy = (int)(r / PW), xx = (int)(r - (long long)yy * PW);
      if (yy >= 1 && yy <= H && xx >= 1 && xx <= W) v = x[(((b * H + (yy - 1)) * W) + (xx - 1)) * cpr + c];
    }
    out[i] = v;
  }
}

// dy (B, OH, OW, C) -> zero-dilated (B, 2 OH - 1, 2 OW - 1, C): out[b, 2 oy, 2 ox] = dy[b, oy, ox], zeros elsewhere; 16-byte pieces
__global__ void dilate2_kernel(const u32x4* __restrict__ dy, u32x4* __restrict__ out, int B, int OH, int OW, int cpr) {
  const long long DH = 2 * OH - 1, DW = 2 * OW - 1, total = (long long)B * DH * DW * cpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpr);
    const long long p = i / cpr, b = p / (DH * DW), r = p - b * DH * DW;
    const int yy = (int)(r / DW), xx = (int)(r - (long long)yy * DW);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (!(yy & 1) && !(xx & 1)) v = dy[((b * OH + (yy >> 1)) * OW + (xx >> 1)) * cpr + c];
    out[i] = v;
  }
}

// x (B, H, W, C) -> phase image (py, px) of its bottom / right zero-padded copy: out[guard + (b (OH + 1) + a) (OW + 1) + c] =
// xp[b, 2 a + py, 2 c + px] (zero beyond row H - 1 / column W - 1); guard rows and the tail are zeros
__global__ void phase_pad_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ out, int B, int H, int W, int OH, int OW, int py, int px,
                                 int cpr, long long guard, long long total_rows) {
  const long long total = total_rows * cpr, PW = OW + 1, PH = OH + 1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / cpr - guard;
    const int c = (int)(i % cpr);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row >= 0 && row < (long long)B * PH * PW) {
      const long long b = row / (PH * PW), r = row - b * PH * PW;
      const int a = (int)(r / PW), cc = (int)(r - (long long)a * PW);
      const int yy = 2 * a + py, xx = 2 * cc + px;
      if (yy < H && xx < W) v = x[((b * H + yy) * W + xx) * cpr + c];
    }
    out[i] = v;
  }
}

// dy (B, OH, OW, C) -> rows of a (B, OH + 1, OW + 1, C) image whose last row / column are zeros (the un-shifted operand of the
// stride-2 weight gradient); the rows behind the last image are zeros
__global__ void pad_last_kernel(const u32x4* __restrict__ dy, u32x4* __restrict__ out, int B, int OH, int OW, int cpr, long long total_rows) {
  const long long total = total_rows * cpr, PW = OW + 1, PH = OH + 1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / cpr;
    const int c = (int)(i % cpr);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < (long long)B * PH * PW) {
      const long long b = row / (PH * PW), r = row - b * PH * PW;
      const int a = (int)(r / PW), cc = (int)(r - (long long)a * PW);
      if (a < OH && cc < OW) v = dy[((b * OH + a) * OW + cc) * cpr + c];
    }
    out[i] = v;
  }
}

// nearest x2: y[b, 2 h + i, 2 w + j] = x[b, h, w] (F.interpolate(scale_factor=2, mode="nearest"), big_model_attn_gan.py:183), pieces
__global__ void upsample2_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int B, int H, int W, int cpr) {
  const long long total = (long long)B * 4 * H * W * cpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpr);
    const long long p = i / cpr, b = p / (4ll * H * W), r = p - b * 4ll * H * W;
    const int yy = (int)(r / (2 * W)), xx = (int)(r - (long long)yy * 2 * W);
    y[i] = x[((b * H + (yy >> 1)) * W + (xx >> 1)) * cpr + c];
  }
}

// its adjoint: y[b, h, w] = sum of the 2 x 2 block of x (B, 2 H, 2 W, C); f32 sums, stored in T
template <typename T>
__global__ void sumpool2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
  const long long total = (long long)B * H * W * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long p = i / C, b = p / ((long long)H * W), r = p - b * (long long)H * W;
    const int h = (int)(r / W), w = (int)(r - (long long)h * W);
    const long long o = ((b * 2 * H + 2 * h) * 2 * W + 2 * w) * C + c, rs = 2ll * W * C;
    Elem<T>::st(y + i, (Elem<T>::ld(x + o) + Elem<T>::ld(x + o + C)) + (Elem<T>::ld(x + o + rs) + Elem<T>::ld(x + o + rs + C)));
  }
}

// softmax backward, one wave per row: dS[c] = scale P[c] (dP[c] - sum_c' dP[c'] P[c']) for c < n, 0 for n <= c < ld_ds
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const T* __restrict__ P, long long ldp, const float* __restrict__ dP,
                                                               long long lddp, int n, long long rows, float scale, T* __restrict__ dS,
                                                               long long ldds) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* p = P + row * ldp;
  const float* d = dP + row * lddp;
  float s = 0.f;
  for (int c = lane; c < n; c += 64) s = fmaf(d[c], Elem<T>::ld(p + c), s);
  s = wave_sum(s);
  for (int c = lane; c < ldds; c += 64) Elem<T>::st(dS + row * ldds + c, c < n ? scale * Elem<T>::ld(p + c) * (d[c] - s) : 0.f);
}

// 1-channel image (B, H, W) -> the im2col matrix of a 3 x 3 / pad 1 convolution, (B H W) x 32: column t < 9 = img[y + t / 3 - 1,
// x + t % 3 - 1] (zero outside the image), columns 9 .. 31 zeros (one MFMA contraction step)
template <typename T>
__global__ void im2col_c1_kernel(const T* __restrict__ img, T* __restrict__ out, int B, int H, int W) {
  const long long total = (long long)B * H * W * 32;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i & 31);
    const long long p = i >> 5, b = p / ((long long)H * W), r = p - b * (long long)H * W;
    const int y = (int)(r / W) + t / 3 - 1, x = (int)(r % W) + t % 3 - 1;
    float v = 0.f;
    if (t < 9 && y >= 0 && y < H && x >= 0 && x < W) v = Elem<T>::ld(img + (b * H + y) * W + x);
    Elem<T>::st(out + i, v);
  }
}

__device__ __forceinline__ float sigmoidf_(float h) { return 1.f / (1.f + __expf(-h)); }

// pass 1: per (image, pixel chunk, channel): sum dh, sum dh xhat  ->  part[((b * nchunks + chunk) * C + c) * 2 + {0, 1}]
template <typename T>
__global__ __launch_bounds__(256) void gnb_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int HW, int C, int swish, float* __restrict__ part) {
  const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x, cpg = C / GNB_GROUPS;
  const int p0 = chunk * GNB_CHUNK, p1 = min(HW, p0 + GNB_CHUNK);
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const float mu = mean[b * GNB_GROUPS + g], rs = rstd[b * GNB_GROUPS + g], ga = gamma[c], be = beta[c];
    float s0 = 0.f, s1 = 0.f;
    for (int p = p0; p < p1; ++p) {
      const long long i = ((long long)b * HW + p) * C + c;
      const float xh = (Elem<T>::ld(x + i) - mu) * rs, h = fmaf(xh, ga, be);
      float d = Elem<T>::ld(dy + i);
      if (swish) {
        const float sg = sigmoidf_(h);
        d *= sg * (1.f + h * (1.f - sg));
      }
      s0 += d;
      s1 = fmaf(d, xh, s1);
    }
    float* o = part + (((long long)b * nchunks + chunk) * C + c) * 2;
    o[0] = s0;
    o[1] = s1;
  }
}

// fixed-order sums over the chunks: tot[(b * C + c) * 2 + k]
__global__ __launch_bounds__(256) void gnb_totals_kernel(const float* __restrict__ part, int nchunks, int B, int C, float* __restrict__ tot) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B * C; i += gridDim.x * 256) {
    const int b = i / C, c = i - b * C;
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < nchunks; ++k) {
      const float* o = part + (((long long)b * nchunks + k) * C + c) * 2;
      s0 += o[0];
      s1 += o[1];
    }
    tot[2 * i] = s0;
    tot[2 * i + 1] = s1;
  }
}
// per (b, group): gs[(b * 32 + g) * 2 + k] = sum_c gamma_c tot[b, c, k]; dgamma / dbeta = sums over the images, in order
__global__ __launch_bounds__(256) void gnb_groups_kernel(const float* __restrict__ tot, int B, int C, const float* __restrict__ gamma,
                                                         float* __restrict__ gs, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int cpg = C / GNB_GROUPS, t = blockIdx.x * 256 + threadIdx.x, nt = gridDim.x * 256;
  for (int i = t; i < B * GNB_GROUPS; i += nt) {
    const int b = i / GNB_GROUPS, g = i - b * GNB_GROUPS;
    float s0 = 0.f, s1 = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const float* q = tot + 2 * ((long long)b * C + c);
      s0 = fmaf(gamma[c], q[0], s0);
      s1 = fmaf(gamma[c], q[1], s1);
    }
    gs[2 * i] = s0;
    gs[2 * i + 1] = s1;
  }
  for (int c = t; c < C; c += nt) {
    float s0 = 0.f, s1 = 0.f;
    for (int b = 0; b < B; ++b) {
      const float* q = tot + 2 * ((long long)b * C + c);
      s0 += q[0];
      s1 += q[1];
    }
    if (dbeta) dbeta[c] = s0;
    if (dgamma) dgamma[c] = s1;
  }
}

// pass 2: dx = rstd (dh gamma - s0 / n - xhat s1 / n)
template <typename T>
__global__ __launch_bounds__(256) void gnb_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ gs, int HW, int C,
                                                        int swish, long long total, T* __restrict__ dx) {
  const int cpg = C / GNB_GROUPS;
  const float inv_n = 1.f / ((float)cpg * (float)HW);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long long b = i / ((long long)HW * C);
    const int bg = (int)b * GNB_GROUPS + c / cpg;
    const float mu = mean[bg], rs = rstd[bg], ga = gamma[c], be = beta[c];
    const float xh = (Elem<T>::ld(x + i) - mu) * rs, h = fmaf(xh, ga, be);
    float d = Elem<T>::ld(dy + i);
    if (swish) {
      const float sg = sigmoidf_(h);
      d *= sg * (1.f + h * (1.f - sg));
    }
    Elem<T>::st(dx + i, rs * (d * ga - gs[2 * bg] * inv_n - xh * gs[2 * bg + 1] * inv_n));
  }
}

inline int grid_cap(long long n, int per, int cap = 8192) {
  long long g = (n + per - 1) / per;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

struct BwdWeightPlan {
  long long PW, rows, guard, kb, rows_pad;  // padded width, padded pixels of all images, guard rows, rows per batch, rows incl. tail
  int nb;
  long long xp_bytes, dyp_bytes, part_bytes;
};
inline long long rup16(long long v) { return (v + 15) / 16 * 16; }
inline BwdWeightPlan plan_bwd_weight(int B, int H, int W, int Cin, int Cout, int es) {
  BwdWeightPlan p{};
  p.PW = W + 2;
  p.rows = (long long)B * (H + 2) * p.PW;
  p.guard = p.PW + 1;
  long long nb = p.rows / 2048;
  nb = nb < 1 ? 1 : nb > 256 ? 256 : nb;
  p.kb = (p.rows + nb - 1) / nb;
  p.kb = (p.kb + 63) / 64 * 64;   // whole K units per batch
  p.nb = (int)((p.rows + p.kb - 1) / p.kb);
  p.rows_pad = p.kb * p.nb;
  p.xp_bytes = rup16((p.guard + p.rows_pad + p.guard) * Cin * es);
  p.dyp_bytes = rup16(p.rows_pad * (long long)Cout * es);
  p.part_bytes = rup16((long long)p.nb * Cout * 9 * Cin * 4);
  if (p.part_bytes < 256ll * Cout * 4) p.part_bytes = 256ll * Cout * 4;   // (also melgpt_colsum's workspace for the bias gradient)
  return p;
}

// stride-2 weight gradient: four phase images of the padded input, each (guard + rows_pad + guard) rows of Cin, + the padded dY + partials
struct BwdWeightS2Plan {
  long long PW, rows, guard, kb, rows_pad;
  int nb, OH, OW;
  long long phase_bytes, dyp_bytes, part_bytes;
};
inline BwdWeightS2Plan plan_bwd_weight_s2(int B, int H, int W, int Cin, int Cout, int es) {
  BwdWeightS2Plan p{};
  p.OH = (H + 1 - 3) / 2 + 1;
  p.OW = (W + 1 - 3) / 2 + 1;
  p.PW = p.OW + 1;
  p.rows = (long long)B * (p.OH + 1) * p.PW;
  p.guard = p.PW + 1;
  long long nb = p.rows / 2048;
  nb = nb < 1 ? 1 : nb > 256 ? 256 : nb;
  p.kb = (p.rows + nb - 1) / nb;
  p.kb = (p.kb + 63) / 64 * 64;
  p.nb = (int)((p.rows + p.kb - 1) / p.kb);
  p.rows_pad = p.kb * p.nb;
  p.phase_bytes = rup16((p.rows_pad + 2 * p.guard) * Cin * es);
  p.dyp_bytes = rup16(p.rows_pad * (long long)Cout * es);
  p.part_bytes = rup16((long long)p.nb * Cout * 9 * Cin * 4);
  if (p.part_bytes < 256ll * Cout * 4) p.part_bytes = 256ll * Cout * 4;
  return p;
}

}  // namespace

// ---- Downsample (F.pad (0,1,0,1) + conv3x3 stride 2, big_model_attn_gan.py:151-159): x (B,H,W,Cin) -> y (B,OH,OW,Cout), OH = (H - 2) / 2 + 1
extern "C" long long melgpt_conv3x3_s2_bwd_workspace(int B, int H, int W, int Cin, int Cout, int dtype) {
  if (B <= 0 || H < 2 || W < 2 || Cin <= 0 || Cout <= 0 || (dtype != MELGPT_F32 && dtype != MELGPT_BF16)) return -1;
  const int es = dtype == MELGPT_F32 ? 4 : 2;
  const BwdWeightS2Plan p = plan_bwd_weight_s2(B, H, W, Cin, Cout, es);
  const long long data = rup16((long long)Cin * 9 * Cout * es) + rup16((long long)B * (2 * p.OH - 1) * (2 * p.OW - 1) * Cout * es);
  const long long weight = 4 * p.phase_bytes + p.dyp_bytes + p.part_bytes;
  return data > weight ? data : weight;
}

extern "C" int melgpt_conv3x3_s2_bwd_data(const void* dy, const void* wpack, void* dx, int B, int H, int W, int Cin, int Cout,
                                          void* workspace, int dtype, void* stream) {
  MELGPT_CHECK(dy && wpack && dx && workspace && B > 0 && H >= 2 && W >= 2 && Cin > 0 && Cout > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  const int es = dtype == MELGPT_F32 ? 4 : 2;
  MELGPT_CHECK(Cout % (dtype == MELGPT_F32 ? 32 : 64) == 0 && Cin % 8 == 0 && (Cout * es) % 16 == 0, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)dy & 15) == 0, MELGPT_ERR_ALIGN);
  const int OH = (H + 1 - 3) / 2 + 1, OW = (W + 1 - 3) / 2 + 1;
  hipStream_t s = (hipStream_t)stream;
  char* wrot = (char*)workspace;
  char* dyd = wrot + rup16((long long)Cin * 9 * Cout * es);
  const long long total = (long long)Cout * 9 * Cin;
  if (dtype == MELGPT_F32)
    hipLaunchKernelGGL(rot180_swap_kernel<float>, dim3(grid_cap(total, 256)), dim3(256), 0, s, (const float*)wpack, (float*)wrot, Cout, Cin);
  else
    hipLaunchKernelGGL(rot180_swap_kernel<bf16_t>, dim3(grid_cap(total, 256)), dim3(256), 0, s, (const bf16_t*)wpack, (bf16_t*)wrot, Cout, Cin);
  const int cy = Cout * es / 16;
  hipLaunchKernelGGL(dilate2_kernel, dim3(grid_cap((long long)B * (2 * OH - 1) * (2 * OW - 1) * cy, 256, 65536)), dim3(256), 0, s, (const u32x4*)dy,
                     (u32x4*)dyd, B, OH, OW, cy);
  int st = melgpt_launch_status();
  if (st != MELGPT_OK) return st;
  // dx[iy, ix] = sum_{ky', kx'} dyd[iy - 2 + ky', ix - 2 + kx'] Wrot[ky', kx']: a stride-1 convolution of the dilated gradient, pad 2
  return melgpt_conv2d_nhwc(dyd, B, 2 * OH - 1, 2 * OW - 1, Cout, wrot, Cin, 3, 3, 1, 2, 2, H, W, 0, nullptr, nullptr, dx, dtype, stream);
}

extern "C" int melgpt_conv3x3_s2_bwd_weight(const void* x, const void* dy, float* dw, float* dbias, int B, int H, int W, int Cin,
                                            int Cout, void* workspace, int dtype, void* stream) {
  MELGPT_CHECK(x && dy && dw && workspace && B > 0 && H >= 2 && W >= 2 && Cin > 0 && Cout > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  const int es = dtype == MELGPT_F32 ? 4 : 2;
  MELGPT_CHECK((Cin * es) % 16 == 0 && (Cout * es) % 16 == 0, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace) & 15) == 0, MELGPT_ERR_ALIGN);
  const BwdWeightS2Plan p = plan_bwd_weight_s2(B, H, W, Cin, Cout, es);
  MELGPT_CHECK(p.rows_pad + 2 * p.guard < 0x7FFFFF00LL, MELGPT_ERR_UNSUPPORTED);
  hipStream_t s = (hipStream_t)stream;
  char* ph = (char*)workspace;
  char* dyp = ph + 4 * p.phase_bytes;
  float* part = (float*)(dyp + p.dyp_bytes);
  const int cx = Cin * es / 16, cy = Cout * es / 16;
  for (int q = 0; q < 4; ++q)
    hipLaunchKernelGGL(phase_pad_kernel, dim3(grid_cap((p.rows_pad + 2 * p.guard) * cx, 256, 65536)), dim3(256), 0, s, (const u32x4*)x,
                       (u32x4*)(ph + q * p.phase_bytes), B, H, W, p.OH, p.OW, q >> 1, q & 1, cx, p.guard, p.rows_pad + 2 * p.guard);
  hipLaunchKernelGGL(pad_last_kernel, dim3(grid_cap(p.rows_pad * cy, 256, 65536)), dim3(256), 0, s, (const u32x4*)dy, (u32x4*)dyp, B, p.OH, p.OW,
                     cy, p.rows_pad);
  int st = melgpt_launch_status();
  if (st != MELGPT_OK) return st;
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap % 3;
    const long long off = (long long)(ky >> 1) * p.PW + (kx >> 1);
    const char* xb = ph + ((ky & 1) * 2 + (kx & 1)) * p.phase_bytes + (p.guard + off) * (long long)Cin * es;
    st = melgpt_gemm(dyp, 1, Cout, p.kb * Cout, xb, 1, Cin, p.kb * Cin, part + (long long)tap * Cin, 9ll * Cin, (long long)Cout * 9 * Cin, Cout,
                     Cin, (int)p.kb, p.nb, dtype, 1, 0, 1.0f, nullptr, MELGPT_ACT_NONE, nullptr, 0, 0, nullptr, 0.f, 0ull, 0u, stream);
    if (st != MELGPT_OK) return st;
  }
  st = melgpt_reduce_rows(part, p.nb, (long long)Cout * 9 * Cin, (long long)Cout * 9 * Cin, dw, 0, 1.0f, stream);
  if (st != MELGPT_OK) return st;
  if (dbias) {
    st = melgpt_colsum(dyp, p.rows_pad, Cout, Cout, dbias, 0, part, dtype, stream);
    if (st != MELGPT_OK) return st;
  }
  return MELGPT_OK;
}

// ---- Upsample (nearest x2 + conv3x3, :171-186): the materialised x2 tensor for the weight gradient, and the adjoint of the x2
extern "C" int melgpt_upsample2_nhwc(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
  MELGPT_CHECK(x && y && B > 0 && H > 0 && W > 0 && C > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  const int es = dtype == MELGPT_F32 ? 4 : 2;
  MELGPT_CHECK((C * es) % 16 == 0, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK((((uintptr_t)x | (uintptr_t)y) & 15) == 0, MELGPT_ERR_ALIGN);
  const int cpr = C * es / 16;
  hipLaunchKernelGGL(upsample2_kernel, dim3(grid_cap((long long)B * 4 * H * W * cpr, 256, 65536)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)x, (u32x4*)y, B, H, W, cpr);
  return melgpt_launch_status();
}

extern "C" int melgpt_sumpool2_nhwc(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
  MELGPT_CHECK(x && y && B > 0 && H > 0 && W > 0 && C > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  const long long total = (long long)B * H * W * C;
  if (dtype == MELGPT_F32)
    hipLaunchKernelGGL(sumpool2_kernel<float>, dim3(grid_cap(total, 256, 65536)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, B, H, W, C);
  else
    hipLaunchKernelGGL(sumpool2_kernel<bf16_t>, dim3(grid_cap(total, 256, 65536)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, B, H, W, C);
  return melgpt_launch_status();
}

extern "C" int melgpt_softmax_bwd_rows(const void* probs, long long ld_probs, const float* dprobs, long long ld_dprobs, int n, long long rows,
                                       float scale, void* dscores, long long ld_dscores, int dtype, void* stream) {
  MELGPT_CHECK(probs && dprobs && dscores && n > 0 && rows > 0 && ld_probs >= n && ld_dprobs >= n && ld_dscores >= n, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  hipStream_t s = (hipStream_t)stream;
  const int grid = (int)((rows + 3) / 4);
  if (dtype == MELGPT_F32)
    hipLaunchKernelGGL(softmax_bwd_rows_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)probs, ld_probs, dprobs, ld_dprobs, n, rows,
                       scale, (float*)dscores, ld_dscores);
  else
    hipLaunchKernelGGL(softmax_bwd_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)probs, ld_probs, dprobs, ld_dprobs, n, rows,
                       scale, (bf16_t*)dscores, ld_dscores);
  return melgpt_launch_status();
}

extern "C" int melgpt_im2col_c1(const void* img, void* out, int B, int H, int W, int dtype, void* stream) {
  MELGPT_CHECK(img && out && B > 0 && H > 0 && W > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)B * H * W * 32;
  if (dtype == MELGPT_F32)
    hipLaunchKernelGGL(im2col_c1_kernel<float>, dim3(grid_cap(total, 256, 65536)), dim3(256), 0, s, (const float*)img, (float*)out, B, H, W);
  else
    hipLaunchKernelGGL(im2col_c1_kernel<bf16_t>, dim3(grid_cap(total, 256, 65536)), dim3(256), 0, s, (const bf16_t*)img, (bf16_t*)out, B, H, W);
  return melgpt_launch_status();
}

extern "C" long long melgpt_conv3x3_bwd_workspace(int B, int H, int W, int Cin, int Cout, int dtype) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (dtype != MELGPT_F32 && dtype != MELGPT_BF16)) return -1;
  const int es = dtype == MELGPT_F32 ? 4 : 2;
  const BwdWeightPlan p = plan_bwd_weight(B, H, W, Cin, Cout, es);
  const long long data = rup16((long long)Cin * 9 * Cout * es);  // the rotated weight of bwd_data
  const long long weight = p.xp_bytes + p.dyp_bytes + p.part_bytes;
  return data > weight ? data : weight;
}

extern "C" int melgpt_conv3x3_bwd_data(const void* dy, const void* wpack, void* dx, int B, int H, int W, int Cin, int Cout,
                                       void* workspace, int dtype, void* stream) {
  MELGPT_CHECK(dy && wpack && dx && workspace && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  // (the forward kernel's contraction-width rule applies to Cout here: it is the reduction of the gradient convolution)
  MELGPT_CHECK(Cout % (dtype == MELGPT_F32 ? 32 : 64) == 0 && Cin % 8 == 0, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(((uintptr_t)workspace & 15) == 0, MELGPT_ERR_ALIGN);
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)Cout * 9 * Cin;
  if (dtype == MELGPT_F32)
    hipLaunchKernelGGL(rot180_swap_kernel<float>, dim3(grid_cap(total, 256)), dim3(256), 0, s, (const float*)wpack, (float*)workspace, Cout, Cin);
  else
    hipLaunchKernelGGL(rot180_swap_kernel<bf16_t>, dim3(grid_cap(total, 256)), dim3(256), 0, s, (const bf16_t*)wpack, (bf16_t*)workspace, Cout, Cin);
  int st = melgpt_launch_status();
  if (st != MELGPT_OK) return st;
  return melgpt_conv2d_nhwc(dy, B, H, W, Cout, workspace, Cin, 3, 3, 1, 1, 1, H, W, 0, nullptr, nullptr, dx, dtype, stream);
}

extern "C" int melgpt_conv3x3_bwd_weight(const void* x, const void* dy, float* dw, float* dbias, int B, int H, int W, int Cin,
                                         int Cout, void* workspace, int dtype, void* stream) {
  MELGPT_CHECK(x && dy && dw && workspace && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  const int es = dtype == MELGPT_F32 ? 4 : 2;
  MELGPT_CHECK((Cin * es) % 16 == 0 && (Cout * es) % 16 == 0, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace) & 15) == 0, MELGPT_ERR_ALIGN);
  const BwdWeightPlan p = plan_bwd_weight(B, H, W, Cin, Cout, es);
  MELGPT_CHECK(p.rows_pad + 2 * p.guard < 0x7FFFFF00LL, MELGPT_ERR_UNSUPPORTED);
  hipStream_t s = (hipStream_t)stream;
  char* xp = (char*)workspace;
  char* dyp = xp + p.xp_bytes;
  float* part = (float*)(dyp + p.dyp_bytes);
  const int cx = Cin * es / 16, cy = Cout * es / 16;
  hipLaunchKernelGGL(pad_border_kernel, dim3(grid_cap((p.rows_pad + 2 * p.guard) * cx, 256, 65536)), dim3(256), 0, s, (const u32x4*)x,
                     (u32x4*)xp, B, H, W, cx, p.guard, p.rows_pad + 2 * p.guard);
  hipLaunchKernelGGL(pad_border_kernel, dim3(grid_cap(p.rows_pad * cy, 256, 65536)), dim3(256), 0, s, (const u32x4*)dy, (u32x4*)dyp, B, H, W,
                     cy, 0ll, p.rows_pad);
  int st = melgpt_launch_status();
  if (st != MELGPT_OK) return st;
  for (int tap = 0; tap < 9; ++tap) {
    const long long off = (long long)(tap / 3 - 1) * p.PW + (tap % 3 - 1);
    const char* xb = xp + (p.guard + off) * (long long)Cin * es;
    // part[z] (Cout x 9 Cin, f32) [:, tap * Cin : (tap + 1) * Cin] = dYp_z^T Xp_z[off:]   (both operands K-major, K = kb rows)
    st = melgpt_gemm(dyp, 1, Cout, p.kb * Cout, xb, 1, Cin, p.kb * Cin, part + (long long)tap * Cin, 9ll * Cin, (long long)Cout * 9 * Cin, Cout,
                     Cin, (int)p.kb, p.nb, dtype, 1, 0, 1.0f, nullptr, MELGPT_ACT_NONE, nullptr, 0, 0, nullptr, 0.f, 0ull, 0u, stream);
    if (st != MELGPT_OK) return st;
  }
  st = melgpt_reduce_rows(part, p.nb, (long long)Cout * 9 * Cin, (long long)Cout * 9 * Cin, dw, 0, 1.0f, stream);
  if (st != MELGPT_OK) return st;
  if (dbias) {
    // column sums of dY over its B H W rows: the zero-bordered copy has the same sums; melgpt_colsum wants melgpt_colsum_rows() x Cout floats of workspace
    // -> taken from the partial-sum buffer, which the reduction above has already consumed (stream order)
    st = melgpt_colsum(dyp, p.rows_pad, Cout, Cout, dbias, 0, part, dtype, stream);
    if (st != MELGPT_OK) return st;
  }
  return MELGPT_OK;
}

extern "C" long long melgpt_groupnorm_swish_bwd_workspace(int B, int HW, int C) {
  if (B <= 0 || HW <= 0 || C <= 0) return -1;
  const long long nchunks = (HW + GNB_CHUNK - 1) / GNB_CHUNK;
  return ((long long)B * nchunks * C * 2 + (long long)B * C * 2 + (long long)B * GNB_GROUPS * 2 + 4) * 4;  // bytes
}

extern "C" int melgpt_groupnorm_swish_bwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                          const void* dy, void* dx, float* dgamma, float* dbeta, int B, int HW, int C, int swish,
                                          float* workspace, int dtype, void* stream) {
  MELGPT_CHECK(x && mean && rstd && gamma && beta && dy && dx && workspace && B > 0 && HW > 0 && C > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(C % GNB_GROUPS == 0 && B <= 65535, MELGPT_ERR_UNSUPPORTED);
  hipStream_t s = (hipStream_t)stream;
  const int nchunks = (HW + GNB_CHUNK - 1) / GNB_CHUNK;
  float* part = workspace;
  float* tot = part + (long long)B * nchunks * C * 2;
  float* gs = tot + (long long)B * C * 2;
  const long long total = (long long)B * HW * C;
  if (dtype == MELGPT_F32) {
    hipLaunchKernelGGL(gnb_partial_kernel<float>, dim3(nchunks, B), dim3(256), 0, s, (const float*)x, (const float*)dy, mean, rstd, gamma, beta,
                       HW, C, swish, part);
    hipLaunchKernelGGL(gnb_totals_kernel, dim3(grid_cap((long long)B * C, 256, 1024)), dim3(256), 0, s, part, nchunks, B, C, tot);
    hipLaunchKernelGGL(gnb_groups_kernel, dim3(grid_cap((long long)B * GNB_GROUPS + C, 256, 64)), dim3(256), 0, s, tot, B, C, gamma, gs, dgamma, dbeta);
    hipLaunchKernelGGL(gnb_apply_kernel<float>, dim3(grid_cap(total, 256, 65536)), dim3(256), 0, s, (const float*)x, (const float*)dy, mean, rstd,
                       gamma, beta, gs, HW, C, swish, total, (float*)dx);
  } else {
    hipLaunchKernelGGL(gnb_partial_kernel<bf16_t>, dim3(nchunks, B), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, mean, rstd, gamma,
                       beta, HW, C, swish, part);
    hipLaunchKernelGGL(gnb_totals_kernel, dim3(grid_cap((long long)B * C, 256, 1024)), dim3(256), 0, s, part, nchunks, B, C, tot);
    hipLaunchKernelGGL(gnb_groups_kernel, dim3(grid_cap((long long)B * GNB_GROUPS + C, 256, 64)), dim3(256), 0, s, tot, B, C, gamma, gs, dgamma, dbeta);
    hipLaunchKernelGGL(gnb_apply_kernel<bf16_t>, dim3(grid_cap(total, 256, 65536)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, mean,
                       rstd, gamma, beta, gs, HW, C, swish, total, (bf16_t*)dx);
  }
  return melgpt_launch_status();
}
