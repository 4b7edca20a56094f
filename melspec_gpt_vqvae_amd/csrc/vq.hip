// VQ codebook lookup for gfx950: nearest-neighbour argmin + gather + MSE partials + histogram in ONE
// pass over the latents (reference: vqvae/big_model_attn_gan.py:19-54 does it with 7 ATen ops and
// materialises the (N,K) distance and one-hot matrices).
//
//   F32 lane  : codebook resident in LDS (128 x 258 f32, 2-float row skew => conflict-free b32 fragment
//               reads), cross term on v_mfma_f32_16x16x4_f32 = exact k-ordered f32 FMA chain, so
//               oracle/vq_argmin.c reproduces every distance bit for bit.
//   BF16 lane : codebook rounded to bf16 in LDS (XOR-swizzled), every wave owns 16 latent vectors end to end
//               (fragments straight from global, v_mfma_f32_16x16x32_bf16, no barrier in the tile loop).
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
// inner == 0 marks rows at ONE pitch (s_outer == inner * s_inner, folded on the host by vq_fold): no 64-bit
// division per row - it was ~130 instructions with branches in front of every tile's loads.
__device__ __forceinline__ long long vq_off(const VqAddr& a, long long n, int c) {
  if (a.inner == 0) return n * a.s_inner + (long long)c * a.s_c;
  return (n / a.inner) * a.s_outer + (n % a.inner) * a.s_inner + (long long)c * a.s_c;
}
inline VqAddr vq_fold(VqAddr a) {
  if (a.s_outer == a.inner * a.s_inner) a.inner = 0;
  return a;
}

__device__ __forceinline__ float bf16lo(unsigned v) { return half_lo(v); }
__device__ __forceinline__ float bf16hi(unsigned v) { return half_hi(v); }

// lexicographic (distance, code) minimum == torch.argmin's first-minimal-index rule
// Branch-free on purpose: written with `if`, each call became an exec-mask region (s_and_saveexec ... s_or exec) and
// the 32 regions per tile fenced the scheduler: no LDS read or MFMA of the next code tile could move above them.
__device__ __forceinline__ void lexmin(float& d, int& k, float d2, int k2) {
  const bool c = (d2 < d) | ((d2 == d) & (k2 < k));
  d = c ? d2 : d;
  k = c ? k2 : k;
}
// Same rule when k2 is known to exceed every index taken so far (codes visited in ascending order in a lane):
// the tie clause can never fire, a strict compare is the whole test.
__device__ __forceinline__ void lexmin_asc(float& d, int& k, float d2, int k2) {
  const bool c = d2 < d;
  d = c ? d2 : d;
  k = c ? k2 : k;
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
  // (eight loads in flight per trip: a load -> store loop pays one memory round trip per element, and this staging is
  // most of the kernel's time at small batches)
  for (int i0 = t; i0 < VQ_K * VQ_D / 4; i0 += 256 * 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(codebook + (size_t)(i0 + 256 * u) * 4);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 256 * u, code = i >> 6, c = (i & 63) * 4;
      float* dst = cb + code * F32_ROW + c;
      *(f32x2*)dst = f32x2{v[u][0], v[u][1]};
      *(f32x2*)(dst + 2) = f32x2{v[u][2], v[u][3]};
    }
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
// Codebook rounded to bf16 once per workgroup into LDS (128 x 512 B, 16-byte chunks XOR-swizzled by code&15 so
// that ds_read_b128 fragment reads are conflict-free).  Every WAVE then owns 16 latent vectors end to end:
//   x fragments (B operand: lane (v, g) holds x[v][32ks+8g .. +7]) are loaded straight from global memory,
//   8 code tiles x 8 k-steps of v_mfma_f32_16x16x32_bf16 with the code fragments streamed from LDS,
//   |x|^2 from the same registers, argmin in-lane + 2 xor-shuffles, optional gather / error / histogram.
// No staging buffer, no barrier inside the tile loop: waves run independently and 16 of them (2 workgroups of 8)
// share a CU, so HBM latency is hidden by occupancy.  Strided (NCHW) latents take a scalar gather path.
// What bounds it (tools/lab/vq_lab.py, profiles/r01_m_vq_lab.jsonl; 1.09 M vectors = 565 MB): the loads alone
// take 107 us (5.3 TB/s), the MFMA + compare work alone 93 us - 72 MFMAs per tile at the clock the chip holds under
// matrix load - and the two overlap only in part: 128 us.  A variant with coalesced loads transposed through a
// wave-private LDS block measured the same (131-139 us), so the fragment loads are not what limits it.
constexpr int B16_WAVES = 8;
constexpr size_t B16_LDS_BYTES = (size_t)VQ_K * 512 + VQ_K * 4 + VQ_K * 4;

__device__ __forceinline__ int cb_off(int code, int chunk) { return code * 512 + ((chunk ^ (code & 15)) << 4); }

// |x|^2 of a lane's vector on the matrix pipe: G = X.X^T over the fragments the distance MFMAs use anyway (8 MFMAs;
// as 128 shift / and / fma instructions it was 40 % of the tile's vector issue, and the kernel is issue-bound).
// Lane (v, g) ends with G[4g .. 4g+3][v]; the diagonal G[v][v] sits in lane (v, v >> 2), slot v & 3, and all four
// lanes of a vector fetch it from there, so they agree on A bit for bit.
__device__ __forceinline__ float xsq_mfma(const u32x4 (&xf)[8], int r16) {
  f32x4 gg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    gg = MELGPT_MFMA_16x16x32(xf[ks], xf[ks], gg);
  const int slot = r16 & 3;
  const float lo = (slot & 1) ? gg[1] : gg[0], hi = (slot & 1) ? gg[3] : gg[2];
  return __shfl((slot & 2) ? hi : lo, r16 + 16 * (r16 >> 2), 64);
}

// FLAT: channel-contiguous latents (16-byte fragment loads).  EXTRAS: gather / squared error / distances wanted.
// The lean <true,false> instantiation is the hot path of extract_codes (indices only).
template <bool FLAT, bool EXTRAS>
__global__ __launch_bounds__(64 * B16_WAVES, EXTRAS ? 2 : 4) void vq_bf16_kernel(const bf16_t* __restrict__ z, VqAddr za, long long N,
                                                                   const float* __restrict__ codebook,
                                                                   long long* __restrict__ indices,
                                                                   bf16_t* __restrict__ qout, float* __restrict__ sq_err,
                                                                   int* __restrict__ hist, float* __restrict__ dist_out) {
  constexpr bool flat = FLAT;
  // The kernels with extra outputs still need xf after the MFMAs (gather, error), so their next tile is fetched into a
  // second register set under the MFMAs.  The lean kernel reloads xf in place once the last MFMA has read it and
  // leaves the latency to the other three waves of its SIMD: measured faster at every size (r01_m_vq_lab.jsonl),
  // the 32 registers let the compiler run the LDS fragment reads ahead of the MFMAs instead of one at a time.
  constexpr bool PF = FLAT && EXTRAS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* cbs = smem;                                   // [128][512 B] bf16, swizzled
  float* bsq = (float*)(smem + VQ_K * 512);           // [128] |e|^2 of the ROUNDED codes
  int* hist_s = (int*)(bsq + VQ_K);                   // [128]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int r16 = lane & 15, g = lane >> 4;

  const int ntiles = (int)((N + 15) / 16);  // the launcher refuses N beyond 2^31 tiles
  const int stride = (int)gridDim.x * B16_WAVES;
  float err = 0.f;
  // this lane's fragments of the 16 vectors of `tile` (clamped row for the ragged tail)
  auto load_x = [&](int tile, u32x4 (&dst)[8]) {
    long long nn = (long long)tile * 16 + r16;
    if (nn >= N) nn = N - 1;
    if constexpr (FLAT) {
      const bf16_t* xp = z + vq_off(za, nn, 8 * g);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) dst[ks] = *(const u32x4*)(xp + 32 * ks);
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        unsigned e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = z[vq_off(za, nn, 32 * ks + 8 * g + j)];
        dst[ks] = u32x4{e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
      }
    }
  };
  u32x4 xf[8], xn[8];
  int tile = (int)blockIdx.x * B16_WAVES + w;
  {  // 16-byte chunk q = (code, chunk) of the rounded codebook; all 16 loads of a thread in flight before the first store
    static_assert(VQ_K * 32 == 8 * 64 * B16_WAVES, "eight chunks per thread");
    f32x4 lo[8], hi[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = t + 64 * B16_WAVES * u;
      const float* e = codebook + (size_t)(q >> 5) * VQ_D + (q & 31) * 8;
      lo[u] = *(const f32x4*)e;
      hi[u] = *(const f32x4*)(e + 4);
    }
    // The first tile's vectors are requested AFTER the codebook: loads return in issue order, so the codebook (L2
    // hits) is rounded and stored while the vectors are still on their way from HBM, instead of queueing behind them.
    if (tile < ntiles) load_x(tile, xf);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = t + 64 * B16_WAVES * u;
      *(u32x4*)(cbs + cb_off(q >> 5, q & 31)) = u32x4{pack_bf16x2(lo[u][0], lo[u][1]), pack_bf16x2(lo[u][2], lo[u][3]),
                                                      pack_bf16x2(hi[u][0], hi[u][1]), pack_bf16x2(hi[u][2], hi[u][3])};
    }
  }
  if (t < VQ_K) hist_s[t] = 0;
  __syncthreads();
  if (t < VQ_K) {  // |e|^2: one FMA chain over the 256 rounded components (any fixed order is fine in this lane)
    float p = 0.f;
    for (int ch = 0; ch < 32; ++ch) {
      u32x4 v = *(const u32x4*)(cbs + cb_off(t, ch));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p = fmaf(bf16lo(v[e]), bf16lo(v[e]), p);
        p = fmaf(bf16hi(v[e]), bf16hi(v[e]), p);
      }
    }
    bsq[t] = p;
  }
  __syncthreads();

  for (; tile < ntiles; tile += stride) {
    const long long n = (long long)tile * 16 + r16;
    const bool valid = n < N;
    const bool more = tile + stride < ntiles;
    if (PF && more) load_x(tile + stride, xn);  // next tile's loads fly under this tile's MFMAs
    const float A = xsq_mfma(xf, r16);

    // starts at the lane's first code, not at a sentinel: a row whose distances are all +inf then resolves to code 0
    // through the cross-lane rule below, as torch.argmin does
    float best = __builtin_inff();
    int bk = 4 * g;
    // one basic block for the eight code tiles: the code fragments and the |e|^2 quad of the next code tile are
    // requested while the current tile's distances are compared
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const f32x4 b4 = *(const f32x4*)(bsq + 16 * ct + 4 * g);  // this lane's four codes of the tile: one ds_read_b128
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const u32x4 cf = *(const u32x4*)(cbs + cb_off(16 * ct + r16, 4 * ks + g));
        acc = MELGPT_MFMA_16x16x32(cf, xf[ks], acc);
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int code = 16 * ct + 4 * g + rg;
        const float d = (A + b4[rg]) - 2.0f * acc[rg];
        if constexpr (EXTRAS) {
          if (dist_out && valid) dist_out[n * VQ_K + code] = d;
        }
        lexmin_asc(best, bk, d, code);  // a lane meets its codes in ascending order
      }
    }
    lexmin_xor(best, bk, 16);
    lexmin_xor(best, bk, 32);
    bk &= (VQ_K - 1);
    if (g == 0 && valid) {
      indices[n] = (long long)bk;
      if (hist) atomicAdd(&hist_s[bk], 1);
    }
    if (EXTRAS && (qout || sq_err) && valid) {  // gather + straight-through value + squared error (64 components / lane)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const u32x4 ev = *(const u32x4*)(cbs + cb_off(bk, 4 * ks + g));
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = bf16lo(xf[ks][e]), x1 = bf16hi(xf[ks][e]);
          const float d0 = bf16lo(ev[e]) - x0, d1 = bf16hi(ev[e]) - x1;
          err = fmaf(d0, d0, err);
          err = fmaf(d1, d1, err);
          o[e] = pack_bf16x2(x0 + d0, x1 + d1);
        }
        if (qout) {
          if (flat) {
            *(u32x4*)(qout + vq_off(za, n, 32 * ks + 8 * g)) = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              qout[vq_off(za, n, 32 * ks + 8 * g + 2 * e)] = (bf16_t)(o[e] & 0xFFFFu);
              qout[vq_off(za, n, 32 * ks + 8 * g + 2 * e + 1)] = (bf16_t)(o[e] >> 16);
            }
          }
        }
      }
    }
    if (more) {
      if constexpr (PF) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xf[ks] = xn[ks];
      } else {
        load_x(tile + stride, xf);
      }
    }
  }
  if (sq_err) {  // deterministic per-workgroup partial: wave sums combined in wave order
    __shared__ float err_s[B16_WAVES];
    err = wave_sum(err);
    if (lane == 0) err_s[w] = err;
    __syncthreads();
    if (t == 0) {
      float s = 0.f;
      for (int i = 0; i < B16_WAVES; ++i) s += err_s[i];
      sq_err[blockIdx.x] = s;
    }
  } else {
    __syncthreads();
  }
  if (hist && t < VQ_K && hist_s[t]) atomicAdd(&hist[t], hist_s[t]);
}

// ============================================================================== prepared codebook image
// Everything the bf16 lookup derives from the codebook on EVERY launch - the bf16 rounding, the swizzled fragment
// layout, the |e|^2 chain of 256 dependent FMAs per code - depends only on the codebook's values, so it is prepared
// ONCE per codebook version into a device-resident image and a launch just copies the image to LDS:
//     [VQ_K][512 B] "hi" fragments (bf16, chunks XOR-swizzled as cb_off) | optional [VQ_K][512 B] "lo" | c[VQ_K] f32
// Two kinds of image:
//   plain  : hi = bf16(E), c_k = |bf16(e_k)|^2 (same chain order as vq_bf16_kernel) - the lookup then produces the same
//            bits as vq_bf16_kernel<true, false>;
//   fused  : quant_conv (1x1, z = W x + b, big_model_attn_gan.py:578,607) folded in algebraically:
//            |z - e_k|^2 = |z|^2 + |e_k|^2 - 2 (x . (W^T e_k) + b . e_k), and |z|^2 does not depend on k, so
//            argmin_k = argmin_k ( c_k - 2 x . E'_k ),  E' = E W  (K x D_in),  c_k = |e_k|^2 - 2 b . e_k  (f32, unrounded).
//            The lookup then runs on the ENCODER's output x and z is never formed, stored or re-read on the
//            indices-only path.  E' is kept as hi + lo (two bf16 planes: 2^-17 relative) unless the caller asks for hi
//            only; the latents x are bf16 in this lane either way.
constexpr size_t IMG_PLANE = (size_t)VQ_K * 512;
__host__ __device__ constexpr size_t img_bytes(bool lo) { return (lo ? 2 : 1) * IMG_PLANE + VQ_K * 4; }

__global__ __launch_bounds__(256) void vq_prepare_kernel(const float* __restrict__ codebook, const float* __restrict__ W,
                                                         const float* __restrict__ bias, char* __restrict__ image,
                                                         int with_lo) {
  __shared__ float row[VQ_D];
  __shared__ float ek[VQ_D];
  const int k = blockIdx.x, c = threadIdx.x;
  ek[c] = codebook[(size_t)k * VQ_D + c];
  __syncthreads();
  float v = ek[c];
  if (W) {  // E'[k][c] = sum_j e_k[j] W[j][c]   (W is the Conv2d weight (out j, in c, 1, 1)); ascending j, one FMA chain
    v = 0.f;
    for (int j = 0; j < VQ_D; ++j) v = fmaf(ek[j], W[(size_t)j * VQ_D + c], v);
  }
  row[c] = v;
  __syncthreads();
  const int planes = (W && with_lo) ? 2 : 1;
  if (c < 32 * planes) {
    const int pl = c >> 5, q = c & 31;
    unsigned short h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = row[8 * q + e];
      const bf16_t hi = f32_to_bf16(x);
      h[e] = pl == 0 ? hi : f32_to_bf16(x - bf16_to_f32(hi));
    }
    *(u32x4*)(image + pl * IMG_PLANE + cb_off(k, q)) =
        u32x4{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16),
              (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16)};
  }
  if (c == 0) {
    float p = 0.f;
    if (W) {
      float be = 0.f;
      for (int j = 0; j < VQ_D; ++j) {
        p = fmaf(ek[j], ek[j], p);
        if (bias) be = fmaf(ek[j], bias[j], be);
      }
      p = p - 2.0f * be;
    } else {  // |bf16(e)|^2 in vq_bf16_kernel's order (ascending component, one chain)
      for (int j = 0; j < VQ_D; ++j) {
        const float r = bf16_to_f32(f32_to_bf16(ek[j]));
        p = fmaf(r, r, p);
      }
    }
    ((float*)(image + planes * IMG_PLANE))[k] = p;
  }
}

// Lean indices-only lookup on a prepared image (the hot path of extract_codes / encode_to_codes): per workgroup the
// image is copied to LDS with every load in flight at once (no rounding, no |e|^2 pass, ONE barrier), then every wave
// owns 16 vectors end to end exactly as in vq_bf16_kernel.  XSQ: add |x|^2 (plain images: keeps the reference's
// evaluation order (|x|^2 + |e|^2) - 2 x.e and the bits of vq_bf16_kernel); fused images skip it.
template <bool HAS_LO, bool XSQ>
__global__ __launch_bounds__(64 * B16_WAVES, HAS_LO ? 2 : 4) void vq_image_kernel(const bf16_t* __restrict__ z, VqAddr za,
                                                                                 long long N, const char* __restrict__ image,
                                                                                 long long* __restrict__ indices,
                                                                                 int* __restrict__ hist) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PLANES = HAS_LO ? 2 : 1;
  char* cbs = smem;
  float* bsq = (float*)(smem + PLANES * IMG_PLANE);
  int* hist_s = (int*)(bsq + VQ_K);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int ntiles = (int)((N + 15) / 16);
  const int stride = (int)gridDim.x * B16_WAVES;
  auto load_x = [&](int tile, u32x4 (&dst)[8]) {
    long long nn = (long long)tile * 16 + r16;
    if (nn >= N) nn = N - 1;
    const bf16_t* xp = z + vq_off(za, nn, 8 * g);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) dst[ks] = *(const u32x4*)(xp + 32 * ks);
  };
  u32x4 xf[8];
  int tile = (int)blockIdx.x * B16_WAVES + w;
  {
    constexpr int NCH = PLANES * (int)IMG_PLANE / 16 / (64 * B16_WAVES);  // 16-byte chunks per thread (8 or 16)
    u32x4 v[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) v[u] = *(const u32x4*)(image + (size_t)(t + 64 * B16_WAVES * u) * 16);
    float c4 = 0.f;
    if (t < VQ_K) c4 = ((const float*)(image + PLANES * IMG_PLANE))[t];
    if (tile < ntiles) load_x(tile, xf);  // after the image (L2 hits): in-order returns, the image is stored meanwhile
#pragma unroll
    for (int u = 0; u < NCH; ++u) *(u32x4*)(cbs + (size_t)(t + 64 * B16_WAVES * u) * 16) = v[u];
    if (t < VQ_K) {
      bsq[t] = c4;
      hist_s[t] = 0;
    }
  }
  __syncthreads();

  for (; tile < ntiles; tile += stride) {
    const long long n = (long long)tile * 16 + r16;
    const bool valid = n < N;
    float A = 0.f;
    if constexpr (XSQ) A = xsq_mfma(xf, r16);
    float best = __builtin_inff();
    int bk = 4 * g;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const f32x4 b4 = *(const f32x4*)(bsq + 16 * ct + 4 * g);
      if constexpr (HAS_LO) {  // the small terms first
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const u32x4 cf = *(const u32x4*)(cbs + IMG_PLANE + cb_off(16 * ct + r16, 4 * ks + g));
          acc = MELGPT_MFMA_16x16x32(cf, xf[ks], acc);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const u32x4 cf = *(const u32x4*)(cbs + cb_off(16 * ct + r16, 4 * ks + g));
        acc = MELGPT_MFMA_16x16x32(cf, xf[ks], acc);
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float d = XSQ ? (A + b4[rg]) - 2.0f * acc[rg] : b4[rg] - 2.0f * acc[rg];
        lexmin_asc(best, bk, d, 16 * ct + 4 * g + rg);
      }
    }
    lexmin_xor(best, bk, 16);
    lexmin_xor(best, bk, 32);
    bk &= (VQ_K - 1);
    if (g == 0 && valid) {
      indices[n] = (long long)bk;
      if (hist) atomicAdd(&hist_s[bk], 1);
    }
    if (tile + stride < ntiles) load_x(tile + stride, xf);
  }
  if (hist) {
    __syncthreads();
    if (t < VQ_K && hist_s[t]) atomicAdd(&hist[t], hist_s[t]);
  }
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
  VqAddr za = vq_fold(VqAddr{inner, stride_outer, stride_inner, stride_c});
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
    MELGPT_CHECK(n_vectors <= (int64_t)16 * 0x7ffffff0, MELGPT_ERR_UNSUPPORTED);  // tile numbers are 32-bit in the kernel
    long long wg_tiles = ((n_vectors + 15) / 16 + B16_WAVES - 1) / B16_WAVES;  // 16 vectors per wave, 8 waves
    int grid = (int)(wg_tiles < 512 ? wg_tiles : 512);                         // 2 persistent workgroups per CU
    static bool attr16 = false;
    if (!attr16) {
      bool ok = hipFuncSetAttribute((const void*)vq_bf16_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B16_LDS_BYTES) == hipSuccess;
      ok = ok && hipFuncSetAttribute((const void*)vq_bf16_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B16_LDS_BYTES) == hipSuccess;
      ok = ok && hipFuncSetAttribute((const void*)vq_bf16_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B16_LDS_BYTES) == hipSuccess;
      if (!ok) return MELGPT_ERR_LAUNCH;
      attr16 = true;
    }
    const bool extras = quantized || sq_err || distances;
#define VQ16_LAUNCH(F, E)                                                                                            \
  hipLaunchKernelGGL((vq_bf16_kernel<F, E>), dim3(grid), dim3(64 * B16_WAVES), B16_LDS_BYTES, s, (const bf16_t*)z, za, \
                     (long long)n_vectors, codebook, (long long*)indices, (bf16_t*)quantized, sq_err, (int*)histogram, \
                     distances)
    if (flat && !extras) VQ16_LAUNCH(true, false);
    else if (flat) VQ16_LAUNCH(true, true);
    else VQ16_LAUNCH(false, true);
#undef VQ16_LAUNCH
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

extern "C" int64_t melgpt_vq_image_bytes(int with_lo) { return (int64_t)img_bytes(with_lo != 0); }

extern "C" int melgpt_vq_prepare_image(const float* codebook, int num_codes, int dim, const float* conv_weight,
                                       const float* conv_bias, int with_lo, void* image, void* stream) {
  MELGPT_CHECK(codebook && image, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dim == VQ_D && num_codes == VQ_K, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(((uintptr_t)image & 15) == 0, MELGPT_ERR_ALIGN);
  MELGPT_CHECK(conv_weight || (!conv_bias && !with_lo), MELGPT_ERR_BAD_ARG);  // bias / lo plane only with a folded conv
  hipLaunchKernelGGL(vq_prepare_kernel, dim3(VQ_K), dim3(256), 0, (hipStream_t)stream, codebook, conv_weight, conv_bias,
                     (char*)image, with_lo);
  return melgpt_launch_status();
}

extern "C" int melgpt_vq_lookup_image(const void* z, int64_t n_vectors, int dim, int64_t inner, int64_t stride_outer,
                                      int64_t stride_inner, int64_t stride_c, const void* image, int with_lo, int fused,
                                      int64_t* indices, int32_t* histogram, void* stream) {
  MELGPT_CHECK(z && image && indices, MELGPT_ERR_BAD_ARG);
  int st = validate_addr(n_vectors, dim, inner, VQ_K);
  if (st != MELGPT_OK) return st;
  MELGPT_CHECK(!with_lo || fused, MELGPT_ERR_BAD_ARG);
  // channel-contiguous bf16 latents only (the encoder's NHWC output): everything else goes through melgpt_vq_argmin_fwd
  MELGPT_CHECK(stride_c == 1 && (stride_inner % 8) == 0 && (stride_outer % 8) == 0, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK((((uintptr_t)z | (uintptr_t)image) & 15) == 0, MELGPT_ERR_ALIGN);
  MELGPT_CHECK(n_vectors <= (int64_t)16 * 0x7ffffff0, MELGPT_ERR_UNSUPPORTED);
  VqAddr za = vq_fold(VqAddr{inner, stride_outer, stride_inner, stride_c});
  const long long wg_tiles = ((n_vectors + 15) / 16 + B16_WAVES - 1) / B16_WAVES;
  const int per_cu = with_lo ? 1 : 2;  // LDS: 128.5 KB with the lo plane, 64.5 KB without
  const int grid = (int)(wg_tiles < 256 * per_cu ? wg_tiles : 256 * per_cu);
  const size_t lds = img_bytes(with_lo != 0) + VQ_K * 4;
  static bool attr = false;
  if (!attr) {
    bool ok = hipFuncSetAttribute((const void*)vq_image_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(img_bytes(false) + VQ_K * 4)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)vq_image_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(img_bytes(false) + VQ_K * 4)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)vq_image_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(img_bytes(true) + VQ_K * 4)) == hipSuccess;
    if (!ok) return MELGPT_ERR_LAUNCH;
    attr = true;
  }
  hipStream_t s = (hipStream_t)stream;
#define VQIMG_LAUNCH(L, X)                                                                                              \
  hipLaunchKernelGGL((vq_image_kernel<L, X>), dim3(grid), dim3(64 * B16_WAVES), lds, s, (const bf16_t*)z, za,           \
                     (long long)n_vectors, (const char*)image, (long long*)indices, (int*)histogram)
  if (with_lo) VQIMG_LAUNCH(true, false);
  else if (fused) VQIMG_LAUNCH(false, false);
  else VQIMG_LAUNCH(false, true);
#undef VQIMG_LAUNCH
  return melgpt_launch_status();
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
  VqAddr oa = vq_fold(VqAddr{inner, stride_outer, stride_inner, stride_c});
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
  VqAddr za = vq_fold(VqAddr{inner, stride_outer, stride_inner, stride_c});
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
