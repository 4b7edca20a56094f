// Backward of the VQ-VAE's two building blocks (conv3x3_{bwd_data,bwd_weight}, groupnorm_swish_bwd): the gradient
// side of torch.nn.Conv2d(k = 3, s = 1, p = 1) and of Normalize -> nonlinearity (vqvae/big_model_attn_gan.py:85-99,117-127,
// 139-140,164-166), NHWC, both numerics lanes.  No scored configuration trains the VQ-VAE (README.md:16 of the reference: it is
// pre-trained elsewhere), so these are CORRECTNESS-first compositions of the forward path's own kernels - the implicit-GEMM
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

}  // namespace

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
