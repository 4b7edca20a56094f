// VQ-VAE encoder/decoder support kernels for gfx950 (NHWC activations, reference vqvae/big_model_attn_gan.py):
//   GroupNorm(32, C, eps=1e-6) statistics + normalise(+swish)      (:139-140, :164-166, :117-126)
//   conv_in  3x3, Cin = 1  (mel tile -> 128 channels)               (:203-207)  - HBM-bound on its output
//   conv_out 3x3, Cout = 1 (128 channels -> mel tile)               (:355-359)
//   row softmax of the 265x265 single-head spatial attention        (:438-440)
//   OIHW f32 -> O,KH,KW,I repack of conv weights into the implicit-GEMM operand layout
// The 3x3 / 1x1 convolutions themselves are csrc/gemm.hip (implicit GEMM on MFMA).
#include "common.h"

namespace {

template <typename T>
struct V16;
template <>
struct V16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void ld(const float* p, float* o) {
    f32x4 v = *(const f32x4*)p;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  }
  static __device__ __forceinline__ void st(float* p, const float* o) { *(f32x4*)p = f32x4{o[0], o[1], o[2], o[3]}; }
};
template <>
struct V16<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void ld(const bf16_t* p, float* o) {
    u32x4 v = *(const u32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = half_lo(v[i]);
      o[2 * i + 1] = half_hi(v[i]);
    }
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float* o) {
    *(u32x4*)p = u32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]),
                       pack_bf16x2(o[6], o[7])};
  }
};

constexpr int GN_GROUPS = 32;
constexpr int GN_CHUNK = 512;  // pixels per stage-1 workgroup

// stage 1: per (batch, pixel chunk) partial sum / sum of squares for each of the 32 groups
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, int HW, int C,
                                                         float* __restrict__ partial) {
  constexpr int N = V16<T>::N;
  extern __shared__ float sh[];  // [nrows][C][2]
  const int t = threadIdx.x, ncols = C / N, nrows = 256 / ncols;
  const int col = t % ncols, prow = t / ncols;
  const int chunk = blockIdx.x, b = blockIdx.y, nchunks = gridDim.x;
  const int p0 = chunk * GN_CHUNK, p1 = min(p0 + GN_CHUNK, HW);
  float s[N], ss[N];
#pragma unroll
  for (int e = 0; e < N; ++e) s[e] = ss[e] = 0.f;
  if (prow < nrows) {
    const T* xb = x + ((long long)b * HW) * C + col * N;
    // eight pixels requested per trip, added in the same order as one at a time (same bits): a thread otherwise has a
    // single 16-byte load in flight and pays a memory latency per pixel
    int p = p0 + prow;
    for (; p + 7 * nrows < p1; p += 8 * nrows) {
      float v[8][N];
#pragma unroll
      for (int j = 0; j < 8; ++j) V16<T>::ld(xb + (long long)(p + j * nrows) * C, v[j]);
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < N; ++e) {
          s[e] += v[j][e];
          ss[e] = fmaf(v[j][e], v[j][e], ss[e]);
        }
    }
    for (; p < p1; p += nrows) {
      float v[N];
      V16<T>::ld(xb + (long long)p * C, v);
#pragma unroll
      for (int e = 0; e < N; ++e) {
        s[e] += v[e];
        ss[e] = fmaf(v[e], v[e], ss[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < N; ++e) {
      sh[(prow * C + col * N + e) * 2] = s[e];
      sh[(prow * C + col * N + e) * 2 + 1] = ss[e];
    }
  }
  __syncthreads();
  if (t < GN_GROUPS) {
    const int cg = C / GN_GROUPS;
    float a = 0.f, q = 0.f;
    for (int r = 0; r < nrows; ++r)
      for (int c = t * cg; c < (t + 1) * cg; ++c) {
        a += sh[(r * C + c) * 2];
        q += sh[(r * C + c) * 2 + 1];
      }
    float* o = partial + (((long long)b * nchunks + chunk) * GN_GROUPS + t) * 2;
    o[0] = a;
    o[1] = q;
  }
}

// stage 2: chunks are combined in double, var = E[x^2] - mean^2 (biased), rstd = 1/sqrt(var + eps).  16 lanes per
// (image, group): lane l adds chunks l, l + 16, ... (four requested per trip), then the 16 partial sums meet in a fixed
// shuffle tree - behind the fused conv an image has 265 chunks, and one thread walking them took 22 us per layer.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int nchunks, int B, double count,
                                                          float eps, float* __restrict__ mean, float* __restrict__ rstd) {
  const int gi = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = gi >> 4, l = gi & 15;  // (b, g), lane of its 16
  const bool live = i < B * GN_GROUPS;
  const int ic = live ? i : B * GN_GROUPS - 1;  // (dead lanes shadow the last pair: every lane reaches the shuffles)
  const int b = ic / GN_GROUPS, g = ic % GN_GROUPS;
  double a = 0.0, q = 0.0;
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t* o = (const f32x2_t*)partial + ((long long)b * nchunks) * GN_GROUPS + g;
  int c = l;
  for (; c + 48 < nchunks; c += 64) {
    f32x2_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = o[(long long)(c + 16 * j) * GN_GROUPS];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a += (double)v[j][0];
      q += (double)v[j][1];
    }
  }
  for (; c < nchunks; c += 16) {
    const f32x2_t v = o[(long long)c * GN_GROUPS];
    a += (double)v[0];
    q += (double)v[1];
  }
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    a += __shfl_xor(a, m);
    q += __shfl_xor(q, m);
  }
  if (!live || l != 0) return;
  const double m = a / count;
  double var = q / count - m * m;
  if (var < 0.0) var = 0.0;
  mean[i] = (float)m;
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

template <typename T>
__global__ void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
                                long long total_vec, int HW, int C, int swish) {
  constexpr int N = V16<T>::N;
  const int ncols = C / N, cg = C / GN_GROUPS;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < total_vec; i += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(i % ncols);
    const long long pix = i / ncols;
    const int b = (int)(pix / HW);
    float v[N];
    V16<T>::ld(x + i * N, v);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const int c = col * N + e, g = b * GN_GROUPS + c / cg;
      float o = (v[e] - mean[g]) * rstd[g] * gamma[c] + beta[c];
      if (swish) o = o / (1.0f + __expf(-o));
      v[e] = o;
    }
    V16<T>::st(y + i * N, v);
  }
}

// The same normalisation with the per-channel affine hoisted: a thread owns ONE 16-byte channel chunk of ONE image for
// all the pixels it visits, so y = x * a + b with a = rstd * gamma, b = beta - mean * a computed once (the generic
// kernel above re-derives image / group / channel per element: 4 dependent scalar loads and 3 integer divisions per
// 16-byte vector, ~1 TB/s).  grid = (pixel blocks, B); needs 256 % (C / N) == 0.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_cols_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y, int HW,
                                                            int C, int swish) {
  constexpr int N = V16<T>::N;
  const int ncols = C / N, cg = C / GN_GROUPS, ppb = 256 / ncols;
  const int col = threadIdx.x % ncols, pl = threadIdx.x / ncols, b = blockIdx.y;
  float a[N], sh[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const int c = col * N + e, g = b * GN_GROUPS + c / cg;
    a[e] = rstd[g] * gamma[c];
    sh[e] = beta[c] - mean[g] * a[e];
  }
  const T* xb = x + (long long)b * HW * C + col * N;
  T* yb = y + (long long)b * HW * C + col * N;
  for (int pix = blockIdx.x * ppb + pl; pix < HW; pix += gridDim.x * ppb) {
    float v[N];
    V16<T>::ld(xb + (long long)pix * C, v);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      float o = fmaf(v[e], a[e], sh[e]);
      if (swish) o = o * __builtin_amdgcn_rcpf(1.0f + __expf(-o));
      v[e] = o;
    }
    V16<T>::st(yb + (long long)pix * C, v);
  }
}

// GroupNorm(32) [+ swish] of a SMALL image in ONE pass (HW <= 1152 pixels: the 10 x 106 and 5 x 53 levels of the VQ-VAE,
// 256 / 512 channels - thirteen layers of the encoder at 12 + 5 + 10 us for three launches over 17 MB): a workgroup owns
// one 128-byte channel slab of one image and keeps it in registers - thread (row lane r0 = t >> 3, piece j = t & 7) holds
// the 16-byte pieces of rows r0, r0 + RL, ... (at most GNF_NP of them, all requested up front) - so the tensor is read once:
// sums / sums of squares per thread in f32 (its pieces all belong to one group: a group is >= one piece), over the 8 row
// lanes of a wave by shuffles, over the waves in double through LDS (wave order), var = E[x^2] - mean^2 as the
// three-kernel path, then y = x a + b [* sigmoid] from the registers.  Needs (C / 32) sizeof(T) >= 16 (one piece never
// straddles two groups) and C sizeof(T) % 128 == 0.
constexpr int GNF_NP = 9;
template <typename T, int NTH>
__global__ __launch_bounds__(NTH) void gn_fused_small_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, T* __restrict__ y, int HW,
                                                               int C, float eps, int swish, float* __restrict__ mean_out,
                                                               float* __restrict__ rstd_out) {
  constexpr int N = V16<T>::N, RL = NTH / 8, NW = NTH / 64;
  __shared__ double red[NW][8][2];
  __shared__ float stat[8][2];
  const int t = threadIdx.x, j = t & 7, r0 = t >> 3, lane = t & 63, wv = t >> 6;
  const int slab = blockIdx.x, b = blockIdx.y;
  const int cg = C / GN_GROUPS;             // channels per group
  const int pg = cg / N;                    // pieces per group: 1, 2 or 4
  const int c0 = slab * (8 * N) + j * N;    // this thread's first channel
  const T* xb = x + (long long)b * HW * C + c0;
  T* yb = y + (long long)b * HW * C + c0;
  float v[GNF_NP][N];
#pragma unroll
  for (int k = 0; k < GNF_NP; ++k) {
    const int r = r0 + k * RL;
    V16<T>::ld(xb + (long long)min(r, HW - 1) * C, v[k]);  // (unconditional: a branch around a load is waited for at its end)
  }
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int k = 0; k < GNF_NP; ++k) {
    const bool live = r0 + k * RL < HW;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float u = live ? v[k][e] : 0.f;
      s += u;
      ss = fmaf(u, u, ss);
    }
  }
  // the 8 row lanes of this wave that hold piece j: lanes j + 8 m
  s += __shfl_xor(s, 8);
  ss += __shfl_xor(ss, 8);
  s += __shfl_xor(s, 16);
  ss += __shfl_xor(ss, 16);
  s += __shfl_xor(s, 32);
  ss += __shfl_xor(ss, 32);
  // the pieces of a group (adjacent, aligned)
  if (pg >= 2) {
    s += __shfl_xor(s, 1);
    ss += __shfl_xor(ss, 1);
  }
  if (pg >= 4) {
    s += __shfl_xor(s, 2);
    ss += __shfl_xor(ss, 2);
  }
  if (lane < 8) {
    red[wv][lane][0] = (double)s;
    red[wv][lane][1] = (double)ss;
  }
  __syncthreads();
  if (t < 8) {  // piece t's group (the pieces of a group hold the same sums)
    double a = 0.0, q = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      a += red[w][t][0];
      q += red[w][t][1];
    }
    const double count = (double)HW * cg;
    const double m = a / count;
    double var = q / count - m * m;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
    stat[t][0] = mf;
    stat[t][1] = rf;
    if (mean_out && (t % pg) == 0) {
      const int g = (slab * 8 + t) / pg;
      mean_out[b * GN_GROUPS + g] = mf;
      rstd_out[b * GN_GROUPS + g] = rf;
    }
  }
  __syncthreads();
  const float mf = stat[j][0], rf = stat[j][1];
  float a[N], sh[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    a[e] = rf * gamma[c0 + e];
    sh[e] = beta[c0 + e] - mf * a[e];
  }
#pragma unroll
  for (int k = 0; k < GNF_NP; ++k) {
    const int r = r0 + k * RL;
    if (r < HW) {
      float o[N];
#pragma unroll
      for (int e = 0; e < N; ++e) {
        float u = fmaf(v[k][e], a[e], sh[e]);
        if (swish) u = u * __builtin_amdgcn_rcpf(1.0f + __expf(-u));
        o[e] = u;
      }
      V16<T>::st(yb + (long long)r * C, o);
    }
  }
}

// 3x3 convolution of a single-channel image into COUT channels (NHWC out).  16 threads per pixel x 8 channels.
template <typename TI, typename T>
__global__ __launch_bounds__(256) void conv_in_c1_kernel(const TI* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, T* __restrict__ y, int B, int H,
                                                         int W, int COUT) {
  extern __shared__ float wsh[];  // [COUT][9] + [COUT]
  for (int i = threadIdx.x; i < COUT * 9; i += 256) wsh[i] = w[i];
  for (int i = threadIdx.x; i < COUT; i += 256) wsh[COUT * 9 + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int groups = COUT / 8, ppb = 256 / groups;
  const int cgp = threadIdx.x % groups, pl = threadIdx.x / groups;
  const long long total = (long long)B * H * W;
  // The kernel writes 2.1 GB at the VAS shape and ends up bound by that (write-only traffic peaks near 2.4 TB/s on
  // this chip: 0.88 ms; two pixels per trip with 18 taps in flight, a prefetched next pixel and four pixels of a row
  // per thread all measured the same or worse).  What did pay (1.01 -> 0.88 ms):
  //  * the thread's 8 output channels never change: their 72 taps and 8 biases live in registers, as channel PAIRS, and
  //    the FMAs are packed (v_pk_fma_f32: two channels per issue slot, same arithmetic per channel);
  //  * (b, y, x) of the thread's pixel advance by a fixed stride: decomposed once, then carried - no 64-bit divisions;
  //  * the nine taps are unconditional loads at clamped coordinates, zeroed by a select (a load inside a branch is
  //    waited for where the branch ends: nine latencies per pixel).
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t wr[4][9], br[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = cgp * 8 + 2 * e;
    br[e] = f32x2_t{wsh[COUT * 9 + c], wsh[COUT * 9 + c + 1]};
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[e][k] = f32x2_t{wsh[c * 9 + k], wsh[(c + 1) * 9 + k]};
  }
  const long long stride = (long long)gridDim.x * ppb, first = (long long)blockIdx.x * ppb + pl;
  const int dxs = (int)(stride % W), dys = (int)((stride / W) % H);
  const long long dbs = stride / ((long long)W * H);
  int xw = (int)(first % W), yh = (int)((first / W) % H);
  long long b = first / ((long long)W * H);
  for (long long pix = first; pix < total; pix += stride, xw += dxs, yh += dys, b += dbs) {
    if (pl >= ppb) break;
    if (xw >= W) {
      xw -= W;
      ++yh;
    }
    if (yh >= H) {
      yh -= H;
      ++b;
    }
    float in[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = yh + ky - 1, ix = xw + kx - 1;
        const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
        const float v = Elem<TI>::ld(x + (b * H + cy) * W + cx);
        in[ky * 3 + kx] = (iy == cy && ix == cx) ? v : 0.f;
      }
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f32x2_t a = br[e];
#pragma unroll
      for (int k = 0; k < 9; ++k) a = __builtin_elementwise_fma(f32x2_t{in[k], in[k]}, wr[e][k], a);
      o[2 * e] = a[0];
      o[2 * e + 1] = a[1];
    }
    T* dst = y + pix * COUT + cgp * 8;
    if constexpr (sizeof(T) == 2) {
      V16<bf16_t>::st((bf16_t*)dst, o);
    } else {
      V16<float>::st((float*)dst, o);
      V16<float>::st((float*)dst + 4, o + 4);
    }
  }
}

// The stem conv AND the GroupNorm(32) statistics of its output in one pass (COUT = 128: a thread's 8 channels are two
// groups of 4): the first ResnetBlock's norm1 otherwise re-reads the 2.2 GB the kernel above has just written
// (gn_partial: 0.37 ms of a 128-tile VQ-encode).  One workgroup = STEM_CHUNK pixels of ONE image, 16 pixels per trip;
// per workgroup a (32, 2) partial of sums / sums of squares of the values AS STORED (rounded to the output format),
// added over the 16 pixel lanes in lane order; melgpt_groupnorm_finalize adds the chunks of an image in chunk order.
constexpr int STEM_CHUNK = 2048;
template <typename TI, typename T>
__global__ __launch_bounds__(256) void conv_in_c1_stats_kernel(const TI* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, T* __restrict__ y, int H,
                                                               int W, float* __restrict__ partial) {
  constexpr int COUT = 128;
  __shared__ float wsh[COUT * 10];
  __shared__ float red[16][16][4];  // [pixel lane][channel group of 8][s lo, ss lo, s hi, ss hi]
  // the chunk's input window: pixels p0 - W - 1 .. p1 + W of the image in linear order (rows above / below the image: zeros),
  // so tap (ky, kx) of pixel p is win[p - p0 + ky W + kx].  The nine taps used to be clamped global loads inside the pixel
  // loop - every trip waited out an L1 / L2 round trip with nothing else in flight: 383 us per 64 tiles for 1.1 GB of
  // stores; from LDS the loop is bound by its stores.
  extern __shared__ float win[];
  for (int i = threadIdx.x; i < COUT * 9; i += 256) wsh[i] = w[i];
  for (int i = threadIdx.x; i < COUT; i += 256) wsh[COUT * 9 + i] = bias ? bias[i] : 0.f;
  {
    const int HWs = H * W, q0 = (int)blockIdx.x * STEM_CHUNK - W - 1, nwin = min(STEM_CHUNK, HWs - (int)blockIdx.x * STEM_CHUNK) + 2 * W + 2;
    const TI* xs = x + (long long)blockIdx.y * HWs;
    for (int i = threadIdx.x; i < nwin; i += 256) {
      const int lin = q0 + i;
      win[i] = (lin >= 0 && lin < HWs) ? Elem<TI>::ld(xs + min(max(lin, 0), HWs - 1)) : 0.f;
    }
  }
  __syncthreads();
  const int cgp = threadIdx.x & 15, pl = threadIdx.x >> 4;
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t wr[4][9], br[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = cgp * 8 + 2 * e;
    br[e] = f32x2_t{wsh[COUT * 9 + c], wsh[COUT * 9 + c + 1]};
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[e][k] = f32x2_t{wsh[c * 9 + k], wsh[(c + 1) * 9 + k]};
  }
  const int b = blockIdx.y, HW = H * W;
  const int p0 = blockIdx.x * STEM_CHUNK, p1 = min(p0 + STEM_CHUNK, HW);
  float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};
  int p = p0 + pl;
  int yh = p / W, xw = p - yh * W;
  for (; p < p1; p += 16, xw += 16) {
    while (xw >= W) {  // (a loop: images narrower than the 16-pixel stride wrap more than once per trip)
      xw -= W;
      ++yh;
    }
    float in[9];
    const float* wp = win + (p - p0);
    const bool left = xw == 0, right = xw == W - 1;  // (the linear window wraps into the neighbouring rows there)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float v = wp[ky * W + kx];
        in[ky * 3 + kx] = ((kx == 0 && left) || (kx == 2 && right)) ? 0.f : v;
      }
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f32x2_t a = br[e];
#pragma unroll
      for (int k = 0; k < 9; ++k) a = __builtin_elementwise_fma(f32x2_t{in[k], in[k]}, wr[e][k], a);
      o[2 * e] = a[0];
      o[2 * e + 1] = a[1];
    }
    T* dst = y + ((long long)b * HW + p) * COUT + cgp * 8;
    if constexpr (sizeof(T) == 2) {
      V16<bf16_t>::st((bf16_t*)dst, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = bf16_to_f32(f32_to_bf16(o[e]));  // statistics of the stored values
    } else {
      V16<float>::st((float*)dst, o);
      V16<float>::st((float*)dst + 4, o + 4);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[e >> 2] += o[e];
      ss[e >> 2] = fmaf(o[e], o[e], ss[e >> 2]);
    }
  }
  red[pl][cgp][0] = s[0];
  red[pl][cgp][1] = ss[0];
  red[pl][cgp][2] = s[1];
  red[pl][cgp][3] = ss[1];
  __syncthreads();
  if (threadIdx.x < 64) {  // (group g = 2 cgp + half, statistic st): 32 x 2 values per workgroup
    const int g = threadIdx.x >> 1, st = threadIdx.x & 1;
    float a = 0.f;
    for (int r = 0; r < 16; ++r) a += red[r][g >> 1][2 * (g & 1) + st];
    partial[(((long long)b * gridDim.x + blockIdx.x) * GN_GROUPS + g) * 2 + st] = a;
  }
}

// 3x3 convolution of a C-channel NHWC tensor into ONE channel: 16 lanes per pixel, shuffle reduction.
template <typename T, typename TO>
__global__ __launch_bounds__(256) void conv_out_c1_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, TO* __restrict__ y, int B,
                                                          int H, int W, int C) {
  // w: (9, C) f32 (tap-major)
  const int lane16 = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const long long total = (long long)B * H * W;
  for (long long pix0 = (long long)blockIdx.x * 16; pix0 < total; pix0 += (long long)gridDim.x * 16) {
    const long long pix = pix0 + pl;
    float acc = 0.f;
    if (pix < total) {
      const int xw = (int)(pix % W), yh = (int)((pix / W) % H);
      const long long b = pix / ((long long)W * H);
      for (int k = 0; k < 9; ++k) {
        const int iy = yh + k / 3 - 1, ix = xw + k % 3 - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const T* src = x + ((b * H + iy) * W + ix) * C;
        for (int c = lane16; c < C; c += 16) acc = fmaf(Elem<T>::ld(src + c), w[k * C + c], acc);
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane16 == 0 && pix < total) Elem<TO>::st(y + pix, acc + (bias ? bias[0] : 0.f));
  }
}

// The same for the 16-bit lane at C = 128 (the VQ-VAE decoder's conv_out, big_model_attn_gan.py:341): the 16 lanes of a
// pixel each own 8 channels - ONE 16-byte load per tap (a wave's four pixels are 1 KiB of contiguous memory) against the
// generic kernel's eight 2-byte loads, their 72 weights in registers for the whole launch.  (The generic kernel took
// 2.7 ms for 64 tiles whose activation is 1.1 GB: 13 x the HBM time, 10 % of the decoder at batch 64.)
template <typename TO>
__global__ __launch_bounds__(256) void conv_out_c1_c128_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, TO* __restrict__ y, int B,
                                                               int H, int W) {
  constexpr int C = 128;
  const int lane16 = threadIdx.x & 15, pl = threadIdx.x >> 4;
  float wr[9][8];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const f32x4 a = *(const f32x4*)(w + k * C + 8 * lane16), b4 = *(const f32x4*)(w + k * C + 8 * lane16 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      wr[k][e] = a[e];
      wr[k][4 + e] = b4[e];
    }
  }
  const float bs = bias ? bias[0] : 0.f;
  const long long total = (long long)B * H * W;
  for (long long pix0 = (long long)blockIdx.x * 16; pix0 < total; pix0 += (long long)gridDim.x * 16) {
    const long long pix = pix0 + pl;
    float acc = 0.f;
    if (pix < total) {
      const int xw = (int)(pix % W), yh = (int)((pix / W) % H);
      const long long b = pix / ((long long)W * H);
      u32x4 v[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {  // all nine requests first (a tap outside the image: zeros)
        const int iy = yh + k / 3 - 1, ix = xw + k % 3 - 1;
        v[k] = u32x4{0u, 0u, 0u, 0u};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[k] = *(const u32x4*)(x + ((b * H + iy) * W + ix) * C + 8 * lane16);
      }
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc = fmaf(half_lo(v[k][e]), wr[k][2 * e], acc);
          acc = fmaf(half_hi(v[k][e]), wr[k][2 * e + 1], acc);
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane16 == 0 && pix < total) Elem<TO>::st(y + pix, acc + bs);
  }
}

// P[r, c] = softmax_c(scale * S[r, c]) for c < n; zero for n <= c < ldp (the padding feeds a K-padded GEMM)
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, long long lds_, int n,
                                                           long long rows, float scale, T* __restrict__ P,
                                                           long long ldp) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* s = S + row * lds_;
  float mx = -__builtin_inff();
  for (int c = lane; c < n; c += 64) mx = fmaxf(mx, s[c] * scale);
  mx = wave_max(mx);
  float l = 0.f;
  for (int c = lane; c < n; c += 64) l += __expf(s[c] * scale - mx);
  l = wave_sum(l);
  const float inv = 1.f / l;
  for (int c = lane; c < ldp; c += 64) Elem<T>::st(P + row * ldp + c, c < n ? __expf(s[c] * scale - mx) * inv : 0.f);
}

template <typename T>
__global__ void repack_oihw_kernel(const float* __restrict__ w, T* __restrict__ out, int O, int I, int KH, int KW) {
  const long long total = (long long)O * I * KH * KW;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < total; i += (long long)gridDim.x * blockDim.x) {  // i indexes the OUTPUT (o, kh, kw, ci)
    const int ci = (int)(i % I);
    const int kw = (int)((i / I) % KW);
    const int kh = (int)((i / ((long long)I * KW)) % KH);
    const int o = (int)(i / ((long long)I * KW * KH));
    Elem<T>::st(out + i, w[(((long long)o * I + ci) * KH + kh) * KW + kw]);
  }
}

// logical (B,C,HW) <-> (B,HW,C) permute with dtype conversion (API boundary only; the pipeline itself stays NHWC)
template <typename TI, typename TO>
__global__ void permute_kernel(const TI* __restrict__ x, TO* __restrict__ y, int B, int C, int HW, int to_nhwc) {
  const long long total = (long long)B * C * HW;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < total; i += (long long)gridDim.x * blockDim.x) {  // i indexes the OUTPUT
    long long src;
    if (to_nhwc) {
      const int c = (int)(i % C);
      const long long p = (i / C) % HW, b = i / ((long long)C * HW);
      src = (b * C + c) * HW + p;
    } else {
      const long long p = i % HW;
      const int c = (int)((i / HW) % C);
      const long long b = i / ((long long)C * HW);
      src = (b * HW + p) * C + c;
    }
    Elem<TO>::st(y + i, Elem<TI>::ld(x + src));
  }
}

inline int cap_grid(long long n, int per, int cap = 16384) {
  long long g = (n + per - 1) / per;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int melgpt_groupnorm_nchunks(int HW) { return (HW + GN_CHUNK - 1) / GN_CHUNK; }

extern "C" int melgpt_groupnorm_stats(const void* x, int B, int HW, int C, float eps, float* mean, float* rstd,
                                      float* workspace, int dtype, void* stream) {
  MELGPT_CHECK(x && mean && rstd && workspace && B > 0 && HW > 0 && C > 0, MELGPT_ERR_BAD_ARG);
  const int vec = dtype == MELGPT_F32 ? 4 : 8;
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(C % GN_GROUPS == 0 && C % vec == 0 && C / vec <= 256 && 256 % (C / vec) == 0, MELGPT_ERR_UNSUPPORTED);
  const int nchunks = melgpt_groupnorm_nchunks(HW);
  const int nrows = 256 / (C / vec);
  const size_t lds = (size_t)nrows * C * 2 * sizeof(float);
  MELGPT_CHECK(lds <= 64 * 1024, MELGPT_ERR_UNSUPPORTED);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MELGPT_F32)
    hipLaunchKernelGGL(gn_partial_kernel<float>, dim3(nchunks, B), dim3(256), lds, s, (const float*)x, HW, C, workspace);
  else
    hipLaunchKernelGGL(gn_partial_kernel<bf16_t>, dim3(nchunks, B), dim3(256), lds, s, (const bf16_t*)x, HW, C,
                       workspace);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * GN_GROUPS * 16 + 255) / 256), dim3(256), 0, s, workspace, nchunks, B,
                     (double)HW * (C / GN_GROUPS), eps, mean, rstd);
  return melgpt_launch_status();
}

extern "C" int melgpt_groupnorm_finalize(const float* partial, int nchunks, int B, double count, float eps, float* mean,
                                         float* rstd, void* stream) {
  MELGPT_CHECK(partial && mean && rstd && nchunks > 0 && B > 0 && count > 0, MELGPT_ERR_BAD_ARG);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * GN_GROUPS * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial,
                     nchunks, B, count, eps, mean, rstd);
  return melgpt_launch_status();
}

// GroupNorm(32) [+ swish] in one launch for small images (gn_fused_small_kernel); MELGPT_ERR_UNSUPPORTED - nothing launched -
// when the shape needs the three-kernel path (melgpt_groupnorm_stats + melgpt_groupnorm_apply).  mean / rstd: optional outputs.
extern "C" int melgpt_groupnorm_fused(const void* x, const float* gamma, const float* beta, void* y, int B, int HW, int C,
                                      float eps, int swish, float* mean, float* rstd, int dtype, void* stream) {
  MELGPT_CHECK(x && gamma && beta && y && B > 0 && HW > 0 && C > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK((mean == nullptr) == (rstd == nullptr), MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  const int es = dtype == MELGPT_F32 ? 4 : 2;
  // a group must be 1, 2 or 4 whole 16-byte pieces (gn_fused_small_kernel: every piece lies inside ONE group): 16, 32 or
  // 64 bytes - bf16 C = 384 (24-byte groups), 640 / 768 (40 / 48) and f32 C = 160 / 192 (20 / 24) are declined here
  const int gbytes = C % GN_GROUPS == 0 ? (C / GN_GROUPS) * es : 0;
  if (C % GN_GROUPS != 0 || (C * es) % 128 != 0 || (gbytes != 16 && gbytes != 32 && gbytes != 64) || B > 65535 ||
      HW > 128 * GNF_NP || ((((uintptr_t)x | (uintptr_t)y) & 15) != 0))
    return MELGPT_ERR_UNSUPPORTED;
  const dim3 grid(C * es / 128, B);
  hipStream_t s = (hipStream_t)stream;
#define GNF_LAUNCH(T, NTH) \
  hipLaunchKernelGGL((gn_fused_small_kernel<T, NTH>), grid, dim3(NTH), 0, s, (const T*)x, gamma, beta, (T*)y, HW, C, eps, swish, mean, rstd)
#define GNF_PICK(T)                               \
  if (HW <= 32 * GNF_NP) GNF_LAUNCH(T, 256);      \
  else if (HW <= 64 * GNF_NP) GNF_LAUNCH(T, 512); \
  else GNF_LAUNCH(T, 1024)
  if (dtype == MELGPT_F32) { GNF_PICK(float); } else { GNF_PICK(bf16_t); }
#undef GNF_PICK
#undef GNF_LAUNCH
  return melgpt_launch_status();
}

extern "C" int melgpt_groupnorm_apply(const void* x, const float* mean, const float* rstd, const float* gamma,
                                      const float* beta, void* y, int B, int HW, int C, int swish, int dtype,
                                      void* stream) {
  MELGPT_CHECK(x && mean && rstd && gamma && beta && y && B > 0 && HW > 0 && C > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  const int vec = dtype == MELGPT_F32 ? 4 : 8;
  MELGPT_CHECK(C % GN_GROUPS == 0 && C % vec == 0, MELGPT_ERR_UNSUPPORTED);
  const long long total = (long long)B * HW * (C / vec);
  hipStream_t s = (hipStream_t)stream;
  const int ncols = C / vec;
  if (256 % ncols == 0 && HW < 0x7FFFFFFF / 2) {
    const int ppb = 256 / ncols;
    int gx = (HW + ppb - 1) / ppb;
    const int want = (8 * 256 + B - 1) / B;        // ~8 blocks per CU over all images; at least 4 pixels per thread
    if (gx > want) gx = want;
    if (gx < 1) gx = 1;
    if (dtype == MELGPT_F32)
      hipLaunchKernelGGL(gn_apply_cols_kernel<float>, dim3(gx, B), dim3(256), 0, s, (const float*)x, mean, rstd, gamma, beta,
                         (float*)y, HW, C, swish);
    else
      hipLaunchKernelGGL(gn_apply_cols_kernel<bf16_t>, dim3(gx, B), dim3(256), 0, s, (const bf16_t*)x, mean, rstd, gamma,
                         beta, (bf16_t*)y, HW, C, swish);
    return melgpt_launch_status();
  }
  if (dtype == MELGPT_F32)
    hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(cap_grid(total, 256)), dim3(256), 0, s, (const float*)x, mean, rstd,
                       gamma, beta, (float*)y, total, HW, C, swish);
  else
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, dim3(cap_grid(total, 256)), dim3(256), 0, s, (const bf16_t*)x, mean,
                       rstd, gamma, beta, (bf16_t*)y, total, HW, C, swish);
  return melgpt_launch_status();
}

extern "C" int melgpt_conv_in_c1(const void* x, int x_dtype, const float* w, const float* bias, void* y, int dtype,
                                 int B, int H, int W, int Cout, void* stream) {
  MELGPT_CHECK(x && w && y && B > 0 && H > 0 && W > 0 && Cout > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(Cout % 8 == 0 && Cout <= 2048 && 256 % (Cout / 8) == 0, MELGPT_ERR_UNSUPPORTED);
  const size_t lds = (size_t)Cout * 10 * sizeof(float);
  const int ppb = 256 / (Cout / 8);
  const int grid = cap_grid((long long)B * H * W, ppb, 65536);
  hipStream_t s = (hipStream_t)stream;
#define CI_LAUNCH(TI, T) \
  hipLaunchKernelGGL((conv_in_c1_kernel<TI, T>), dim3(grid), dim3(256), lds, s, (const TI*)x, w, bias, (T*)y, B, H, W, Cout)
  if (x_dtype == MELGPT_F32 && dtype == MELGPT_F32) CI_LAUNCH(float, float);
  else if (x_dtype == MELGPT_F32 && dtype == MELGPT_BF16) CI_LAUNCH(float, bf16_t);
  else if (x_dtype == MELGPT_BF16 && dtype == MELGPT_BF16) CI_LAUNCH(bf16_t, bf16_t);
  else if (x_dtype == MELGPT_BF16 && dtype == MELGPT_F32) CI_LAUNCH(bf16_t, float);
  else return MELGPT_ERR_UNSUPPORTED;
#undef CI_LAUNCH
  return melgpt_launch_status();
}

extern "C" int melgpt_conv_in_c1_stats_workspace(int B, int H, int W) {
  return B * ((H * W + STEM_CHUNK - 1) / STEM_CHUNK) * GN_GROUPS * 2;
}

extern "C" int melgpt_groupnorm_finalize(const float* partial, int nchunks, int B, double count, float eps, float* mean,
                                         float* rstd, void* stream);

extern "C" int melgpt_conv_in_c1_stats(const void* x, int x_dtype, const float* w, const float* bias, void* y, int dtype,
                                       int B, int H, int W, int Cout, float eps, float* mean, float* rstd,
                                       float* workspace, void* stream) {
  MELGPT_CHECK(x && w && y && mean && rstd && workspace && B > 0 && H > 0 && W > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(Cout == 128 && (long long)H * W < 0x7FFFFFFF / 2, MELGPT_ERR_UNSUPPORTED);
  const int nchunks = (H * W + STEM_CHUNK - 1) / STEM_CHUNK;
  hipStream_t s = (hipStream_t)stream;
  const size_t win_bytes = (size_t)(STEM_CHUNK + 2 * W + 2) * sizeof(float);  // the chunk's input window (+ 9.2 KB static)
  MELGPT_CHECK(win_bytes <= 48 * 1024, MELGPT_ERR_UNSUPPORTED);
#define CIS_LAUNCH(TI, T)                                                                                             \
  hipLaunchKernelGGL((conv_in_c1_stats_kernel<TI, T>), dim3(nchunks, B), dim3(256), win_bytes, s, (const TI*)x, w, bias, \
                     (T*)y, H, W, workspace)
  if (x_dtype == MELGPT_F32 && dtype == MELGPT_F32) CIS_LAUNCH(float, float);
  else if (x_dtype == MELGPT_F32 && dtype == MELGPT_BF16) CIS_LAUNCH(float, bf16_t);
  else if (x_dtype == MELGPT_BF16 && dtype == MELGPT_BF16) CIS_LAUNCH(bf16_t, bf16_t);
  else if (x_dtype == MELGPT_BF16 && dtype == MELGPT_F32) CIS_LAUNCH(bf16_t, float);
  else return MELGPT_ERR_UNSUPPORTED;
#undef CIS_LAUNCH
  return melgpt_groupnorm_finalize(workspace, nchunks, B, (double)H * W * (Cout / GN_GROUPS), eps, mean, rstd, stream);
}

extern "C" int melgpt_conv_out_c1(const void* x, int dtype, const float* w_tap_major, const float* bias, void* y,
                                  int y_dtype, int B, int H, int W, int C, void* stream) {
  MELGPT_CHECK(x && w_tap_major && y && B > 0 && H > 0 && W > 0 && C > 0, MELGPT_ERR_BAD_ARG);
  const int grid = cap_grid((long long)B * H * W, 16, 65536);
  hipStream_t s = (hipStream_t)stream;
#define CO_LAUNCH(T, TO) \
  hipLaunchKernelGGL((conv_out_c1_kernel<T, TO>), dim3(grid), dim3(256), 0, s, (const T*)x, w_tap_major, bias, (TO*)y, B, H, W, C)
  if (dtype == MELGPT_BF16 && C == 128 && (((uintptr_t)x | (uintptr_t)w_tap_major) & 15) == 0) {
    if (y_dtype == MELGPT_F32)
      hipLaunchKernelGGL(conv_out_c1_c128_kernel<float>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, w_tap_major, bias, (float*)y, B, H, W);
    else if (y_dtype == MELGPT_BF16)
      hipLaunchKernelGGL(conv_out_c1_c128_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, w_tap_major, bias, (bf16_t*)y, B, H, W);
    else
      return MELGPT_ERR_UNSUPPORTED;
    return melgpt_launch_status();
  }
  if (dtype == MELGPT_F32 && y_dtype == MELGPT_F32) CO_LAUNCH(float, float);
  else if (dtype == MELGPT_BF16 && y_dtype == MELGPT_F32) CO_LAUNCH(bf16_t, float);
  else if (dtype == MELGPT_BF16 && y_dtype == MELGPT_BF16) CO_LAUNCH(bf16_t, bf16_t);
  else return MELGPT_ERR_UNSUPPORTED;
#undef CO_LAUNCH
  return melgpt_launch_status();
}

extern "C" int melgpt_softmax_rows(const float* scores, long long ld_scores, int n, long long rows, float scale,
                                   void* probs, long long ld_probs, int dtype, void* stream) {
  MELGPT_CHECK(scores && probs && n > 0 && rows > 0 && ld_scores >= n && ld_probs >= n, MELGPT_ERR_BAD_ARG);
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (dtype == MELGPT_F32)
    hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3(grid), dim3(256), 0, s, scores, ld_scores, n, rows, scale,
                       (float*)probs, ld_probs);
  else if (dtype == MELGPT_BF16)
    hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, scores, ld_scores, n, rows, scale,
                       (bf16_t*)probs, ld_probs);
  else
    return MELGPT_ERR_UNSUPPORTED;
  return melgpt_launch_status();
}

extern "C" int melgpt_repack_conv_weight(const float* w_oihw, void* out_ohwi, int out_dtype, int O, int I, int KH,
                                         int KW, void* stream) {
  MELGPT_CHECK(w_oihw && out_ohwi && O > 0 && I > 0 && KH > 0 && KW > 0, MELGPT_ERR_BAD_ARG);
  const long long total = (long long)O * I * KH * KW;
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == MELGPT_F32)
    hipLaunchKernelGGL(repack_oihw_kernel<float>, dim3(cap_grid(total, 256, 4096)), dim3(256), 0, s, w_oihw,
                       (float*)out_ohwi, O, I, KH, KW);
  else if (out_dtype == MELGPT_BF16)
    hipLaunchKernelGGL(repack_oihw_kernel<bf16_t>, dim3(cap_grid(total, 256, 4096)), dim3(256), 0, s, w_oihw,
                       (bf16_t*)out_ohwi, O, I, KH, KW);
  else
    return MELGPT_ERR_UNSUPPORTED;
  return melgpt_launch_status();
}

extern "C" int melgpt_permute_nchw_nhwc(const void* x, int x_dtype, void* y, int y_dtype, int B, int C, int HW,
                                        int to_nhwc, void* stream) {
  MELGPT_CHECK(x && y && B > 0 && C > 0 && HW > 0, MELGPT_ERR_BAD_ARG);
  const long long total = (long long)B * C * HW;
  hipStream_t s = (hipStream_t)stream;
  const int grid = cap_grid(total, 256, 16384);
#define PM(TI, TO) hipLaunchKernelGGL((permute_kernel<TI, TO>), dim3(grid), dim3(256), 0, s, (const TI*)x, (TO*)y, B, C, HW, to_nhwc)
  if (x_dtype == MELGPT_F32 && y_dtype == MELGPT_F32) PM(float, float);
  else if (x_dtype == MELGPT_F32 && y_dtype == MELGPT_BF16) PM(float, bf16_t);
  else if (x_dtype == MELGPT_BF16 && y_dtype == MELGPT_F32) PM(bf16_t, float);
  else if (x_dtype == MELGPT_BF16 && y_dtype == MELGPT_BF16) PM(bf16_t, bf16_t);
  else return MELGPT_ERR_UNSUPPORTED;
#undef PM
  return melgpt_launch_status();
}
