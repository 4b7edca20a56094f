// VQ codebook lookup for gfx950: nearest-neighbour argmin + gather + MSE partials + histogram in ONE
// pass over the latents (reference: vqvae/big_model_attn_gan.py:19-54 does it with 7 ATen ops and
// materialises the (N,K) distance and one-hot matrices).
//
//   F32 lane  : codebook resident in LDS (128 x 258 f32, 2-float row skew => conflict-free b32 fragment
//               reads), cross term on v_mfma_f32_16x16x4_f32 = exact k-ordered f32 FMA chain, so
//               oracle/vq_argmin.c reproduces every distance bit for bit.
//   BF16 lane : codebook fragments live in VGPRs for the whole (persistent) workgroup, latents staged
//               through an XOR-swizzled LDS tile with 16-byte coalesced loads, v_mfma_f32_16x16x32_bf16.
// Both: rows of the MFMA tile = codes, columns = latent vectors, so every lane owns ONE vector column
// and the argmin reduction is in-register + two xor-shuffles (first-index tie-break kept lexicographic).
#include "common.h"

namespace {

constexpr int VQ_D = 256;
constexpr int VQ_K = 128;
constexpr int VQ_MAX_GRID = 2048;

struct VqAddr {
  long long inner, s_outer, s_inner, s_c;
};
__device__ __forceinline__ long long vq_off(const VqAddr& a, long long n, int c) {
  return (n / a.inner) * a.s_outer + (n % a.inner) * a.s_inner + (long long)c * a.s_c;
}

// lexicographic (distance, code) minimum == torch.argmin's first-minimal-index rule
__device__ __forceinline__ void lexmin(float& d, int& k, float d2, int k2) {
  if (d2 < d || (d2 == d && k2 < k)) {
    d = d2;
    k = k2;
  }
}
__device__ __forceinline__ void lexmin_xor(float& d, int& k, int mask) {
  float d2 = __shfl_xor(d, mask, 64);
  int k2 = __shfl_xor(k, mask, 64);
  lexmin(d, k, d2, k2);
}

// =============================================================================================== F32
constexpr int F32_ROW = 258;  // floats per LDS row; 258 % 32 == 2
constexpr int F32_TILE = 16;  // vectors per tile
constexpr size_t F32_LDS_BYTES =
    (size_t)(VQ_K * F32_ROW + F32_TILE * F32_ROW + VQ_K + VQ_K * 4 + 4 * F32_TILE) * 4 +
    (size_t)(4 * F32_TILE + F32_TILE + VQ_K) * 4 + 16 * 4;

__global__ __launch_bounds__(256) void vq_f32_kernel(const float* __restrict__ z, VqAddr za, long long N,
                                                     const float* __restrict__ codebook,
                                                     long long* __restrict__ indices, float* __restrict__ qout,
                                                     float* __restrict__ sq_err, int* __restrict__ hist,
                                                     float* __restrict__ dist_out, int flat) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* cb = (float*)smem;                    // [128][258]
  float* xt = cb + VQ_K * F32_ROW;             // [16][258]
  float* bsq = xt + F32_TILE * F32_ROW;        // [128]
  float* part = bsq + VQ_K;                    // [128][4]
  float* red_d = part + VQ_K * 4;              // [4][16]
  int* red_k = (int*)(red_d + 4 * F32_TILE);   // [4][16]
  int* idx_s = red_k + 4 * F32_TILE;           // [16]
  int* hist_s = idx_s + F32_TILE;              // [128]
  float* err_s = (float*)(hist_s + VQ_K);      // [4]

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int r16 = lane & 15, g = lane >> 4;

  // ---- codebook -> LDS (coalesced float4, skewed rows), |e|^2 as four 64-long FMA chains per code
  for (int i = t; i < VQ_K * VQ_D / 4; i += 256) {
    f32x4 v = *(const f32x4*)(codebook + (size_t)i * 4);
    int code = i >> 6, c = (i & 63) * 4;
    float* dst = cb + code * F32_ROW + c;
    *(f32x2*)dst = f32x2{v[0], v[1]};
    *(f32x2*)(dst + 2) = f32x2{v[2], v[3]};
  }
  if (t < VQ_K) hist_s[t] = 0;
  __syncthreads();
  {
    int code = t & 127, jj = t >> 7;
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      int j = jj * 2 + j2;
      const float* e = cb + code * F32_ROW + 64 * j;
      float p = 0.f;
      for (int c = 0; c < 64; ++c) p = fmaf(e[c], e[c], p);
      part[code * 4 + j] = p;
    }
  }
  __syncthreads();
  if (t < VQ_K) bsq[t] = (part[t * 4 + 0] + part[t * 4 + 1]) + (part[t * 4 + 2] + part[t * 4 + 3]);

  const long long ntiles = (N + F32_TILE - 1) / F32_TILE;
  float xr[16];
  float err = 0.f;

  auto load_tile = [&](long long tile) {
    long long n0 = tile * F32_TILE;
    if (flat) {  // v = t>>4, c = (t&15)*4 + 64*i : 16-byte loads along the channel axis
      long long n = n0 + (t >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < N) v = *(const f32x4*)(z + vq_off(za, n, (t & 15) * 4 + 64 * i));
        xr[4 * i + 0] = v[0]; xr[4 * i + 1] = v[1]; xr[4 * i + 2] = v[2]; xr[4 * i + 3] = v[3];
      }
    } else {  // v = t&15 fastest: coalesced along the spatial axis of an NCHW latent
      long long n = n0 + (t & 15);
#pragma unroll
      for (int i = 0; i < 16; ++i) xr[i] = (n < N) ? z[vq_off(za, n, (t >> 4) + 16 * i)] : 0.f;
    }
  };
  auto elem_vc = [&](int i, int& v, int& c) {
    if (flat) { v = t >> 4; c = (t & 15) * 4 + 64 * (i >> 2) + (i & 3); }
    else { v = t & 15; c = (t >> 4) + 16 * i; }
  };

  long long tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    const long long n0 = tile * F32_TILE;
    __syncthreads();  // everyone is done with the previous tile's xt / idx_s / red_*
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int v, c;
      elem_vc(i, v, c);
      xt[v * F32_ROW + c] = xr[i];
    }
    __syncthreads();
    if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x);  // in flight under the MFMA loop

    // |x|^2 : lane (vec r16, quarter g) -> 64-long FMA chain; A = (p0+p1)+(p2+p3)
    float A;
    {
      const float* xv = xt + r16 * F32_ROW + 64 * g;
      float p = 0.f;
      for (int c = 0; c < 64; ++c) p = fmaf(xv[c], xv[c], p);
      float p0 = __shfl(p, r16, 64), p1 = __shfl(p, r16 + 16, 64);
      float p2 = __shfl(p, r16 + 32, 64), p3 = __shfl(p, r16 + 48, 64);
      A = (p0 + p1) + (p2 + p3);
    }
    // x.e : codes 32w .. 32w+31 (two 16-row MFMA tiles) against the 16 vectors; K index = 4s+g ascending
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    {
      const float* xb = xt + r16 * F32_ROW + g;
      const float* e0 = cb + (32 * w + r16) * F32_ROW + g;
      const float* e1 = e0 + 16 * F32_ROW;
#pragma unroll 8
      for (int s = 0; s < 64; ++s) {
        float b = xb[4 * s];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(e0[4 * s], b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[4 * s], b, acc1, 0, 0, 0);
      }
    }
    // d = (|x|^2 + |e|^2) - 2 x.e   (big_model_attn_gan.py:28-30, same evaluation order)
    float best = __builtin_inff();
    int bk = 0x7fffffff;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        int code = 32 * w + 16 * h + 4 * g + rg;
        float m = h ? acc1[rg] : acc0[rg];
        float d = (A + bsq[code]) - 2.0f * m;
        if (dist_out && n0 + r16 < N) dist_out[(n0 + r16) * VQ_K + code] = d;
        lexmin(best, bk, d, code);
      }
    }
    lexmin_xor(best, bk, 16);
    lexmin_xor(best, bk, 32);
    if (g == 0) {
      red_d[w * F32_TILE + r16] = best;
      red_k[w * F32_TILE + r16] = bk;
    }
    __syncthreads();
    if (t < F32_TILE) {
      float d = red_d[t];
      int k = red_k[t];
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) lexmin(d, k, red_d[ww * F32_TILE + t], red_k[ww * F32_TILE + t]);
      k &= (VQ_K - 1);  // NaN-only columns would leave the sentinel; keep memory accesses in range
      idx_s[t] = k;
      if (n0 + t < N) {
        indices[n0 + t] = (long long)k;
        atomicAdd(&hist_s[k], 1);
      }
    }
    __syncthreads();
    // gather + straight-through value + squared error
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int v, c;
      elem_vc(i, v, c);
      if (n0 + v < N) {
        float x = xt[v * F32_ROW + c];
        float dq = cb[idx_s[v] * F32_ROW + c] - x;
        err = fmaf(dq, dq, err);
        if (qout) qout[vq_off(za, n0 + v, c)] = x + dq;
      }
    }
  }
  // deterministic per-workgroup partial
  err = wave_sum(err);
  if (lane == 0) err_s[w] = err;
  __syncthreads();
  if (t == 0 && sq_err) sq_err[blockIdx.x] = (err_s[0] + err_s[1]) + (err_s[2] + err_s[3]);
  if (hist && t < VQ_K && hist_s[t]) atomicAdd(&hist[t], hist_s[t]);
}

// ============================================================================================== BF16
constexpr int B16_TILE = 64;  // vectors per tile (4 MFMA column tiles)
constexpr size_t B16_LDS_BYTES = (size_t)B16_TILE * 512 + (VQ_K + 4 * 16 + 4 * B16_TILE) * 4 +
                                 (4 * B16_TILE + B16_TILE + VQ_K) * 4 + 16 * 4;

__device__ __forceinline__ int b16_swz(int row, int chunk) { return row * 512 + ((chunk ^ (row & 15)) << 4); }

__global__ __launch_bounds__(256) void vq_bf16_kernel(const bf16_t* __restrict__ z, VqAddr za, long long N,
                                                      const float* __restrict__ codebook,
                                                      long long* __restrict__ indices, bf16_t* __restrict__ qout,
                                                      float* __restrict__ sq_err, int* __restrict__ hist,
                                                      float* __restrict__ dist_out, int flat) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xt = smem;                                        // [64][512 B], 16-B chunks XOR-swizzled by row&15
  float* bsq = (float*)(smem + B16_TILE * 512);           // [128]
  float* asq = bsq + VQ_K;                                // [4][16]
  float* red_d = asq + 4 * 16;                            // [4][64]
  int* red_k = (int*)(red_d + 4 * B16_TILE);              // [4][64]
  int* idx_s = red_k + 4 * B16_TILE;                      // [64]
  int* hist_s = idx_s + B16_TILE;                         // [128]
  float* err_s = (float*)(hist_s + VQ_K);                 // [4]

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int r16 = lane & 15, g = lane >> 4;

  // ---- this wave's 32 codes as MFMA A fragments, rounded to bf16 once; |e|^2 of the ROUNDED codes
  s16x8 cbf[2][8];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const float* e = codebook + (size_t)(32 * w + 16 * ct + r16) * VQ_D + 8 * g;
    float p = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      f32x4 lo = *(const f32x4*)(e + 32 * ks), hi = *(const f32x4*)(e + 32 * ks + 4);
      s16x8 f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[j] = (short)f32_to_bf16(lo[j]);
        f[4 + j] = (short)f32_to_bf16(hi[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = bf16_to_f32((bf16_t)f[j]);
        p = fmaf(v, v, p);
      }
      cbf[ct][ks] = f;
    }
    p += __shfl_xor(p, 16, 64);
    p += __shfl_xor(p, 32, 64);
    if (g == 0) bsq[32 * w + 16 * ct + r16] = p;
  }
  if (t < VQ_K) hist_s[t] = 0;

  const long long ntiles = (N + B16_TILE - 1) / B16_TILE;
  u32x4 xr[8];
  float err = 0.f;

  auto load_tile = [&](long long tile) {
    long long n0 = tile * B16_TILE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int q = t + 256 * i, row = q >> 5, ch = q & 31;
      long long n = n0 + row;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (n < N) {
        if (flat) {
          v = *(const u32x4*)(z + vq_off(za, n, ch * 8));
        } else {
          unsigned e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = z[vq_off(za, n, ch * 8 + j)];
          v = u32x4{e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
        }
      }
      xr[i] = v;
    }
  };

  long long tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    const long long n0 = tile * B16_TILE;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int q = t + 256 * i, row = q >> 5, ch = q & 31;
      *(u32x4*)(xt + b16_swz(row, ch)) = xr[i];
    }
    __syncthreads();
    if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x);

    f32x4 acc[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int vt = 0; vt < 4; ++vt) acc[ct][vt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float pa = 0.f;  // |x|^2 partial for column tile vt == w (each wave owns one)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int vt = 0; vt < 4; ++vt) {
        s16x8 xb = *(const s16x8*)(xt + b16_swz(16 * vt + r16, 4 * ks + g));
        acc[0][vt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cbf[0][ks], xb, acc[0][vt], 0, 0, 0);
        acc[1][vt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cbf[1][ks], xb, acc[1][vt], 0, 0, 0);
        if (vt == w) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float v = bf16_to_f32((bf16_t)xb[j]);
            pa = fmaf(v, v, pa);
          }
        }
      }
    }
    pa += __shfl_xor(pa, 16, 64);
    pa += __shfl_xor(pa, 32, 64);
    if (g == 0) asq[w * 16 + r16] = pa;
    __syncthreads();

#pragma unroll
    for (int vt = 0; vt < 4; ++vt) {
      float A = asq[vt * 16 + r16];
      float best = __builtin_inff();
      int bk = 0x7fffffff;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          int code = 32 * w + 16 * ct + 4 * g + rg;
          float d = (A + bsq[code]) - 2.0f * acc[ct][vt][rg];
          long long n = n0 + 16 * vt + r16;
          if (dist_out && n < N) dist_out[n * VQ_K + code] = d;
          lexmin(best, bk, d, code);
        }
      lexmin_xor(best, bk, 16);
      lexmin_xor(best, bk, 32);
      if (g == 0) {
        red_d[w * B16_TILE + 16 * vt + r16] = best;
        red_k[w * B16_TILE + 16 * vt + r16] = bk;
      }
    }
    __syncthreads();
    if (t < B16_TILE) {
      float d = red_d[t];
      int k = red_k[t];
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) lexmin(d, k, red_d[ww * B16_TILE + t], red_k[ww * B16_TILE + t]);
      k &= (VQ_K - 1);
      idx_s[t] = k;
      if (n0 + t < N) {
        indices[n0 + t] = (long long)k;
        atomicAdd(&hist_s[k], 1);
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int q = t + 256 * i, row = q >> 5, ch = q & 31;
      long long n = n0 + row;
      if (n < N) {
        s16x8 xb = *(const s16x8*)(xt + b16_swz(row, ch));
        const float* e = codebook + (size_t)idx_s[row] * VQ_D + ch * 8;
        f32x4 lo = *(const f32x4*)e, hi = *(const f32x4*)(e + 4);
        unsigned o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float x = bf16_to_f32((bf16_t)xb[j]);
          float ev = bf16_to_f32(f32_to_bf16(j < 4 ? lo[j] : hi[j - 4]));
          float dq = ev - x;
          err = fmaf(dq, dq, err);
          o[j] = f32_to_bf16(x + dq);
        }
        if (qout) {
          if (flat) {
            *(u32x4*)(qout + vq_off(za, n, ch * 8)) =
                u32x4{o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16)};
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) qout[vq_off(za, n, ch * 8 + j)] = (bf16_t)o[j];
          }
        }
      }
    }
  }
  err = wave_sum(err);
  if (lane == 0) err_s[w] = err;
  __syncthreads();
  if (t == 0 && sq_err) sq_err[blockIdx.x] = (err_s[0] + err_s[1]) + (err_s[2] + err_s[3]);
  if (hist && t < VQ_K && hist_s[t]) atomicAdd(&hist[t], hist_s[t]);
}

// ====================================================================================== small kernels
__global__ void vq_finalize_kernel(const float* sq_err, int n_partials, const int* hist, int K, long long N,
                                   int D, float commitment, float* out) {
  __shared__ float sh[256];
  const int t = threadIdx.x;
  float s = 0.f;
  for (int i = t; i < n_partials; i += 256) s += sq_err[i];
  sh[t] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) sh[t] += sh[t + o];
    __syncthreads();
  }
  float mse = sh[0] / ((float)N * (float)D);
  __syncthreads();
  float e = 0.f;
  for (int k = t; k < K; k += 256) {
    float p = (float)hist[k] / (float)N;
    e += p * logf(p + 1e-10f);
  }
  sh[t] = e;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) sh[t] += sh[t + o];
    __syncthreads();
  }
  if (t == 0) {
    out[0] = mse + commitment * mse;
    out[1] = expf(-sh[0]);
    out[2] = mse;
  }
}

template <typename T>
__global__ void vq_gather_kernel(const long long* idx, long long N, const float* cb, int K, int D, T* out,
                                 VqAddr oa) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = N * D;
  for (; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long n = i / D;
    int c = (int)(i % D);
    long long k = idx[n];
    k = k < 0 ? 0 : (k >= K ? K - 1 : k);
    Elem<T>::st(out + vq_off(oa, n, c), cb[k * D + c]);
  }
}

__global__ void vq_onehot_kernel(const long long* idx, long long N, int K, float* enc) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = N * K;
  for (; i < total; i += (long long)gridDim.x * blockDim.x) enc[i] = (idx[i / K] == (i % K)) ? 1.f : 0.f;
}

template <typename T>
__global__ void vq_bwd_kernel(const T* z, const T* gq, VqAddr za, long long N, int D, const float* cb, int K,
                              const long long* idx, const float* g_loss, float commitment, T* dz, float* dcb,
                              int round_cb) {
  const float gl = g_loss ? *g_loss : 0.f;
  const float inv = 1.0f / ((float)N * (float)D);
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = N * D;
  for (; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long n = i / D;
    int c = (int)(i % D);
    long long off = vq_off(za, n, c);
    long long k = idx[n];
    float x = Elem<T>::ld(z + off);
    float e = cb[k * D + c];
    if (round_cb) e = bf16_to_f32(f32_to_bf16(e));
    float g = gq ? Elem<T>::ld(gq + off) : 0.f;
    if (dz) Elem<T>::st(dz + off, g + gl * 2.0f * commitment * inv * (x - e));
    if (dcb && gl != 0.f) atomicAdd(dcb + k * D + c, gl * 2.0f * inv * (e - x));
  }
}

int validate_addr(int64_t n, int dim, int64_t inner, int num_codes) {
  if (n <= 0 || inner <= 0) return MELGPT_ERR_BAD_ARG;
  if (dim != VQ_D || num_codes != VQ_K) return MELGPT_ERR_UNSUPPORTED;
  return MELGPT_OK;
}

}  // namespace

extern "C" int melgpt_vq_max_grid(void) { return VQ_MAX_GRID; }

// `distances` is an optional debug/inspection output ((N,K) f32, the matrix of :28-30); exported through
// melgpt_vq_argmin_fwd_ex so that tests can check the F32 lane bit for bit against oracle/vq_argmin.c.
extern "C" int melgpt_vq_argmin_fwd_ex(const void* z, int z_dtype, int64_t n_vectors, int dim, int64_t inner,
                                       int64_t stride_outer, int64_t stride_inner, int64_t stride_c,
                                       const float* codebook, int num_codes, int64_t* indices,
                                       void* quantized, float* sq_err, int32_t* histogram, float* distances,
                                       int* grid_out, void* stream) {
  MELGPT_CHECK(z && codebook && indices, MELGPT_ERR_BAD_ARG);
  int st = validate_addr(n_vectors, dim, inner, num_codes);
  if (st != MELGPT_OK) return st;
  MELGPT_CHECK(((uintptr_t)codebook & 15) == 0, MELGPT_ERR_ALIGN);
  VqAddr za{inner, stride_outer, stride_inner, stride_c};
  hipStream_t s = (hipStream_t)stream;
  if (z_dtype == MELGPT_F32) {
    int flat = (stride_c == 1 && (stride_inner % 4) == 0 && (stride_outer % 4) == 0 &&
                ((uintptr_t)z & 15) == 0 && (!quantized || ((uintptr_t)quantized & 15) == 0));
    long long ntiles = (n_vectors + F32_TILE - 1) / F32_TILE;
    int grid = (int)(ntiles < 256 ? ntiles : 256);  // 152 KB of LDS => one workgroup per CU
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)vq_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)F32_LDS_BYTES) != hipSuccess)
        return MELGPT_ERR_LAUNCH;
      attr_set = true;
    }
    hipLaunchKernelGGL(vq_f32_kernel, dim3(grid), dim3(256), F32_LDS_BYTES, s, (const float*)z, za,
                       (long long)n_vectors, codebook, (long long*)indices, (float*)quantized, sq_err,
                       (int*)histogram, distances, flat);
    if (grid_out) *grid_out = grid;
  } else if (z_dtype == MELGPT_BF16) {
    int flat = (stride_c == 1 && (stride_inner % 8) == 0 && (stride_outer % 8) == 0 &&
                ((uintptr_t)z & 15) == 0 && (!quantized || ((uintptr_t)quantized & 15) == 0));
    long long ntiles = (n_vectors + B16_TILE - 1) / B16_TILE;
    int grid = (int)(ntiles < 768 ? ntiles : 768);  // ~3 workgroups per CU
    hipLaunchKernelGGL(vq_bf16_kernel, dim3(grid), dim3(256), B16_LDS_BYTES, s, (const bf16_t*)z, za,
                       (long long)n_vectors, codebook, (long long*)indices, (bf16_t*)quantized, sq_err,
                       (int*)histogram, distances, flat);
    if (grid_out) *grid_out = grid;
  } else {
    return MELGPT_ERR_UNSUPPORTED;
  }
  return melgpt_launch_status();
}

extern "C" int melgpt_vq_argmin_fwd(const void* z, int z_dtype, int64_t n_vectors, int dim, int64_t inner,
                                    int64_t stride_outer, int64_t stride_inner, int64_t stride_c,
                                    const float* codebook, int num_codes, int64_t* indices, void* quantized,
                                    float* sq_err, int32_t* histogram, int* grid_out, void* stream) {
  return melgpt_vq_argmin_fwd_ex(z, z_dtype, n_vectors, dim, inner, stride_outer, stride_inner, stride_c,
                                 codebook, num_codes, indices, quantized, sq_err, histogram, nullptr,
                                 grid_out, stream);
}

extern "C" int melgpt_vq_finalize(const float* sq_err, int n_partials, const int32_t* histogram,
                                  int num_codes, int64_t n_vectors, int dim, float commitment_cost,
                                  float* out, void* stream) {
  MELGPT_CHECK(sq_err && histogram && out && n_partials > 0 && n_vectors > 0 && num_codes > 0,
               MELGPT_ERR_BAD_ARG);
  hipLaunchKernelGGL(vq_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sq_err, n_partials,
                     (const int*)histogram, num_codes, (long long)n_vectors, dim, commitment_cost, out);
  return melgpt_launch_status();
}

extern "C" int melgpt_vq_gather(const int64_t* indices, int64_t n_vectors, const float* codebook,
                                int num_codes, int dim, void* out, int out_dtype, int64_t inner,
                                int64_t stride_outer, int64_t stride_inner, int64_t stride_c, void* stream) {
  MELGPT_CHECK(indices && codebook && out && n_vectors > 0 && dim > 0 && num_codes > 0 && inner > 0,
               MELGPT_ERR_BAD_ARG);
  VqAddr oa{inner, stride_outer, stride_inner, stride_c};
  long long total = (long long)n_vectors * dim;
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == MELGPT_F32)
    hipLaunchKernelGGL(vq_gather_kernel<float>, dim3(grid), dim3(256), 0, s, (const long long*)indices,
                       (long long)n_vectors, codebook, num_codes, dim, (float*)out, oa);
  else if (out_dtype == MELGPT_BF16)
    hipLaunchKernelGGL(vq_gather_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const long long*)indices,
                       (long long)n_vectors, codebook, num_codes, dim, (bf16_t*)out, oa);
  else
    return MELGPT_ERR_UNSUPPORTED;
  return melgpt_launch_status();
}

extern "C" int melgpt_vq_onehot(const int64_t* indices, int64_t n_vectors, int num_codes, float* encodings,
                                void* stream) {
  MELGPT_CHECK(indices && encodings && n_vectors > 0 && num_codes > 0, MELGPT_ERR_BAD_ARG);
  long long total = (long long)n_vectors * num_codes;
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(vq_onehot_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)indices, (long long)n_vectors, num_codes, encodings);
  return melgpt_launch_status();
}

extern "C" int melgpt_vq_bwd(const void* z, const void* g_quantized, int dtype, int64_t n_vectors, int dim,
                             int64_t inner, int64_t stride_outer, int64_t stride_inner, int64_t stride_c,
                             const float* codebook, int num_codes, const int64_t* indices,
                             const float* g_loss, float commitment_cost, void* dz, float* dcodebook,
                             void* stream) {
  MELGPT_CHECK(z && codebook && indices && n_vectors > 0 && dim > 0 && inner > 0, MELGPT_ERR_BAD_ARG);
  VqAddr za{inner, stride_outer, stride_inner, stride_c};
  long long total = (long long)n_vectors * dim;
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MELGPT_F32)
    hipLaunchKernelGGL(vq_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)z,
                       (const float*)g_quantized, za, (long long)n_vectors, dim, codebook, num_codes,
                       (const long long*)indices, g_loss, commitment_cost, (float*)dz, dcodebook, 0);
  else if (dtype == MELGPT_BF16)
    hipLaunchKernelGGL(vq_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)z,
                       (const bf16_t*)g_quantized, za, (long long)n_vectors, dim, codebook, num_codes,
                       (const long long*)indices, g_loss, commitment_cost, (bf16_t*)dz, dcodebook, 1);
  else
    return MELGPT_ERR_UNSUPPORTED;
  return melgpt_launch_status();
}
