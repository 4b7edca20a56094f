// Single-token (KV-cached) attention step for autoregressive sampling on gfx950 - SURVEY §8(f) rank 1.
// Reference: the sampling loops re-run the whole GPT per generated token (transformer/minGPT.py:293-360,
// transformer/decoders.py:89-123; CausalSelfAttention.forward(x, layer_past=None) at minGPT.py:72 never uses its
// cache argument).  The last row of a causal attention only needs the new token's query against the keys / values of
// the tokens before it, so a step is: append this token's k, v to a per-layer cache, then
//   y[b, h, :] = softmax_t( q[b, h, :] . K[b, t, h, :] / sqrt(hs) ) @ V[b, t, h, :],   t = 0 .. pos
// (every earlier token is visible to the newest one whatever n_unmasked is; eval mode, so no dropout).
// One workgroup per (batch, head); latency-bound on the cache read (2 * (pos+1) * 64 * sizeof(T) bytes).
#include "mma.h"

namespace {

template <typename T>
__device__ __forceinline__ float ldf(const T* p) {
  if constexpr (sizeof(T) == 2) return bf16_to_f32(*p);
  else return *p;
}
template <typename T>
__device__ __forceinline__ void stf(T* p, float v) {
  if constexpr (sizeof(T) == 2) *p = f32_to_bf16(v);
  else *p = v;
}

// qkv: (B, 3C) rows [key | query | value] of the new token (row stride ld); caches: B * Tmax * C elements, head-major
// One 256-thread workgroup per (batch, head).  A step is latency-bound (<= 2 x 320 cache rows of 128 / 256 bytes), so
// every load of a phase is independent and in flight at once:
//   scores: thread t owns key position t (and t + 256): its whole K row = ROWCH 16-byte loads, dotted with q from LDS;
//   output: thread = (position group g, 16-byte chunk c of the head dimension): V[g + G i][chunk c] for all i, weighted
//           by the probabilities in LDS, then summed over the groups (DPP inside a wave, LDS across the four waves).
template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(const T* __restrict__ qkv, long long ld, T* __restrict__ kc,
                                                          T* __restrict__ vc, int Tmax, int C, int pos,
                                                          const int* __restrict__ pos_dev, T* __restrict__ out,
                                                          float* __restrict__ att_row, float scale) {
  constexpr int HS = 64, MAXT = 320, NT = 256;
  constexpr int VEC = 16 / sizeof(T);      // elements per 16-byte chunk
  constexpr int ROWCH = HS / VEC;          // chunks per cache row: 8 (bf16) / 16 (f32)
  constexpr int G = NT / ROWCH;            // position groups of the output phase: 32 / 16
  constexpr int NIT = (MAXT + G - 1) / G;  // 10 / 20
  if (pos_dev) pos = *pos_dev;  // graph-replayed decoding: the position lives on the device
  if (pos >= Tmax) return;
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float qs[HS];
  __shared__ float vnew[HS];
  __shared__ float red[2][4];
  __shared__ float s_new_sh;
  __shared__ float osum[4][HS];
  const T* row = qkv + (long long)b * ld;
  // caches are HEAD-MAJOR, (B, H, Tmax, 64): the rows a workgroup reads are one contiguous run of 128-byte lines (as
  // (B, Tmax, C) every row was 2 KB from the next and the step read its cache at 2.9 TB/s)
  T* kb = kc + (((long long)b * gridDim.x + h) * Tmax) * HS;
  T* vb = vc + (((long long)b * gridDim.x + h) * Tmax) * HS;
  const int len = pos + 1;
  // the cache rows this thread will need do not depend on the new token: request them first.  BOTH caches are read the
  // same way: thread = (position group g, 16-byte chunk c of the head dimension), rows g + G i - the ROWCH lanes of a
  // row cover its 128 / 256 contiguous bytes, one full line per 8 / 16 lanes.  (One whole K row per thread - the first
  // version - made every load instruction touch 64 different lines, 16 bytes of each: 8 x the requests for the same
  // bytes; at batch 64 the step took 14.4 us against 6 us of cache bytes at the HBM rate, profiles/r03_decode_lab.md.)
  const int g = tid / ROWCH, cch = tid % ROWCH;
  u32x4 kr[NIT], vr[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int t = min(g + G * i, max(pos - 1, 0));  // clamped to rows that exist (their scores are masked below)
    kr[i] = *(const u32x4*)(kb + (long long)t * HS + cch * VEC);
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int t = min(g + G * i, max(pos - 1, 0));  // weight 0 beyond the end
    vr[i] = *(const u32x4*)(vb + (long long)t * HS + cch * VEC);
  }
  // append this token's key / value (wave 0: lane = head dimension); q and v stay in LDS, the newest key's score comes
  // from registers - nobody reads the freshly stored rows back
  if (wave == 0) {
    const float kn = ldf(row + h * HS + lane), qn = ldf(row + C + h * HS + lane), vn = ldf(row + 2 * C + h * HS + lane);
    stf(kb + (long long)pos * HS + lane, kn);
    stf(vb + (long long)pos * HS + lane, vn);
    qs[lane] = qn;
    // what the cache row holds is the ROUNDED value (bf16 lane): use the same for this step
    if constexpr (sizeof(T) == 2) {
      vnew[lane] = bf16_to_f32(f32_to_bf16(vn));
    } else {
      vnew[lane] = vn;
    }
    const float sn = wave_sum(kn * qn);
    if (lane == 0) s_new_sh = sn;
  }
  __syncthreads();
  // scores of rows g + G i: the lane's chunk of the dot product, summed over the ROWCH lanes of the row (DPP butterflies:
  // every lane of the row ends up with the whole dot product, bit-identical)
  float qv[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) qv[e] = qs[cch * VEC + e];
  const float s_new = s_new_sh * scale;
  float s[NIT];
  float mx = s_new;  // (position `pos` itself is always visible)
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    float acc = 0.f;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc = fmaf(half_lo(kr[i][e]), qv[2 * e], acc);
        acc = fmaf(half_hi(kr[i][e]), qv[2 * e + 1], acc);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fmaf(__uint_as_float(kr[i][e]), qv[e], acc);
    }
    acc += dpp_move<0xB1>(acc);   // lane ^ 1
    acc += dpp_move<0x4E>(acc);   // lane ^ 2
    acc += dpp_move<0x141>(acc);  // the other quad of the 8 (mirror inside 8 lanes)
    if constexpr (ROWCH == 16) acc += dpp_move<0x140>(acc);  // f32: 16 lanes per row (mirror inside 16)
    s[i] = (g + G * i < pos) ? acc * scale : -INFINITY;
    mx = fmaxf(mx, s[i]);
  }
  mx = wave_max(mx);
  if (lane == 0) red[0][wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  // every row's exponential is held by its ROWCH lanes: lane chunk 0 of each row contributes it to the sum
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    s[i] = (g + G * i < pos) ? __expf(s[i] - mx) : 0.f;
    if (cch == 0) sum += s[i];
  }
  const float e_new = __expf(s_new - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[1][wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) + e_new);
  if (att_row) {
#pragma unroll
    for (int i = 0; i < NIT; ++i)
      if (cch == 0 && g + G * i < pos) att_row[((long long)b * gridDim.x + h) * Tmax + g + G * i] = s[i] * inv;
    if (tid == 0) att_row[((long long)b * gridDim.x + h) * Tmax + pos] = e_new * inv;
  }
  // output: the lane's rows weighted by their probabilities (already in its registers), chunk c of the head dimension
  float o[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) o[e] = 0.f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const float p = s[i] * inv;  // 0 for rows at and beyond `pos` (a clamped load may hold anything: select, not multiply)
    if (g + G * i >= pos) continue;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[2 * e] = fmaf(p, half_lo(vr[i][e]), o[2 * e]);
        o[2 * e + 1] = fmaf(p, half_hi(vr[i][e]), o[2 * e + 1]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaf(p, __uint_as_float(vr[i][e]), o[e]);
    }
  }
  // sum over the position groups: lanes with equal chunk index differ in lane bits >= log2(ROWCH)
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
#pragma unroll
    for (int m = ROWCH; m < 64; m <<= 1) o[e] += __shfl_xor(o[e], m, 64);
  }
  if (lane < ROWCH) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) osum[wave][lane * VEC + e] = o[e];
  }
  __syncthreads();
  if (tid < HS) {
    const float v = (osum[0][tid] + osum[1][tid]) + (osum[2][tid] + osum[3][tid]) + (e_new * inv) * vnew[tid];
    stf(out + (long long)b * C + h * HS + tid, v);
  }
}

// x[b, :] = tok_emb[idx[b]] + pos_emb[*pos_dev]   (the stem of GPT.forward for one position, minGPT.py:170-180)
template <typename T>
__global__ __launch_bounds__(256) void embed_decode_kernel(const long long* __restrict__ idx, const float* __restrict__ tok,
                                                           const float* __restrict__ pos_emb,
                                                           const int* __restrict__ pos_dev, int C, int V,
                                                           T* __restrict__ out) {
  const int b = blockIdx.x, p = *pos_dev;
  long long tkn = idx[b];
  tkn = tkn < 0 ? 0 : tkn >= V ? V - 1 : tkn;
  for (int c = threadIdx.x; c < C; c += 256) stf(out + (long long)b * C + c, tok[tkn * C + c] + pos_emb[(long long)p * C + c]);
}

__global__ void incr_i32_kernel(int* p) { *p += 1; }

}  // namespace

extern "C" int melgpt_embed_decode(const long long* idx, const float* tok_emb, const float* pos_emb, const int* pos_dev,
                                   int B, int C, int V, void* out, int dtype, void* stream) {
  MELGPT_CHECK(idx && tok_emb && pos_emb && pos_dev && out && B > 0 && C > 0 && V > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MELGPT_F32)
    hipLaunchKernelGGL(embed_decode_kernel<float>, dim3(B), dim3(256), 0, s, idx, tok_emb, pos_emb, pos_dev, C, V, (float*)out);
  else
    hipLaunchKernelGGL(embed_decode_kernel<bf16_t>, dim3(B), dim3(256), 0, s, idx, tok_emb, pos_emb, pos_dev, C, V,
                       (bf16_t*)out);
  return melgpt_launch_status();
}

extern "C" int melgpt_incr_i32(int* counter, void* stream) {
  MELGPT_CHECK(counter, MELGPT_ERR_BAD_ARG);
  hipLaunchKernelGGL(incr_i32_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter);
  return melgpt_launch_status();
}

extern "C" int melgpt_attn_decode(const void* qkv, long long ld, void* kcache, void* vcache, int B, int H, int head_size,
                                  int Tmax, int pos, const int* pos_dev, void* out, float* att_row, int dtype,
                                  void* stream) {
  MELGPT_CHECK(qkv && kcache && vcache && out && B > 0 && H > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(head_size == 64 && Tmax > 0 && Tmax <= 320 && pos >= 0 && pos < Tmax, MELGPT_ERR_UNSUPPORTED);
  const int C = H * head_size;
  MELGPT_CHECK(ld >= 3LL * C && ld % 8 == 0 && ((((uintptr_t)qkv | (uintptr_t)kcache | (uintptr_t)vcache) & 15) == 0),
               MELGPT_ERR_ALIGN);
  const float scale = 1.0f / sqrtf((float)head_size);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MELGPT_F32)
    hipLaunchKernelGGL(attn_decode_kernel<float>, dim3(H, B), dim3(256), 0, s, (const float*)qkv, ld, (float*)kcache,
                       (float*)vcache, Tmax, C, pos, pos_dev, (float*)out, att_row, scale);
  else
    hipLaunchKernelGGL(attn_decode_kernel<bf16_t>, dim3(H, B), dim3(256), 0, s, (const bf16_t*)qkv, ld, (bf16_t*)kcache,
                       (bf16_t*)vcache, Tmax, C, pos, pos_dev, (bf16_t*)out, att_row, scale);
  return melgpt_launch_status();
}

// ====================================================================================== skinny-M linear layer
// y[m, n] = epi( sum_k x[m, k] W[n, k] + bias[n] ) (+ residual[m, n]),  m < M (a handful of rows: the decode batch),
// W (N, K) row-major = nn.Linear.weight.  A decode step streams every weight once and does almost no arithmetic, so
// the tiled MFMA GEMM (one 128-row tile => N/128 workgroups, each reading its weight slice alone) leaves most of the
// chip idle; here ONE WAVE owns 4 output columns: lanes split K in 16-byte chunks (coalesced 1 KiB per row per
// trip, all trips of a row in flight together), accumulate 4 x 16 dot products in registers
// (bf16 pairs unpacked to f32 by shift / mask), and a wave reduction finishes.  x rows are re-read from L1/L2.
// grid = (N / 4, ceil(M / 16)), 1-4 waves per workgroup by K: N = 1024 already gives one workgroup per CU.
namespace {

template <typename T, int MB, bool LN>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const T* __restrict__ x, long long ldx, const T* __restrict__ W,
                                                        long long ldw, const float* __restrict__ bias,
                                                        const T* __restrict__ res, long long ldr, void* __restrict__ y,
                                                        long long ldy, int M, int N, int K, int act, int out_f32,
                                                        const float* ln_g, const float* ln_b,
                                                        float ln_eps) {
  constexpr int VEC = 16 / sizeof(T);  // elements per 16-byte chunk
  // blockDim.x / 64 waves share the 4 columns and split K between them (the launcher aims at ~2 chunks per lane,
  // so that every load of a row is in flight at once whatever K is)
  const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int n0 = blockIdx.x * 4, m0 = blockIdx.y * MB;
  const int mrows = min(MB, M - m0);
  float acc[4][MB];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;
  // the epilogue's operands are requested now, so that they arrive together with the weights instead of costing a
  // second memory round trip after the reduction (a decode step is a chain of ~170 such latency-bound kernels)
  // (branch-free: a load inside a branch is waited for where the branch ends; absent operands read a valid dummy)
  T e_res_raw;
  float e_bias;
  {
    const int r = (lane / MB) & 3, m = min(lane % MB, mrows - 1);
    e_bias = *(bias ? bias + n0 + r : (const float*)W);
    e_res_raw = *(res ? res + (long long)(m0 + m) * ldr + n0 + r : W);
  }
  const int nchunk = K / VEC;
  bool done = false;
  if constexpr (MB <= 4) {
    // The common decode shapes give a lane at most two chunks of K (the launcher sizes nwave for that): then EVERY
    // operand - the weight rows, the x rows, the LayerNorm affine - is requested up front and the kernel costs one
    // memory round trip instead of three (statistics pass, second pass, weights).  With one wave per workgroup the
    // lane's chunks of x are the whole row's share, so the LayerNorm statistics come from the same registers.
    if (nchunk <= 128 * nwave && (!LN || nwave == 1)) {
      done = true;
      u32x4 wv[2][4], xv[2][MB];
      f32x4 gv[2][2], bv[2][2];  // LayerNorm gamma / beta of the lane's chunks (VEC / 4 quads each)
      bool ok[2];
      long long off[2];
      // x and the LayerNorm affine first (L2-resident), the weight rows (HBM) last: loads return in issue order, so the
      // statistics below run while the weights are still on their way
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = wv_id * 64 + lane + 64 * nwave * j;
        ok[j] = c < nchunk;
        off[j] = (long long)(ok[j] ? c : 0) * VEC;
#pragma unroll
        for (int m = 0; m < MB; ++m) xv[j][m] = *(const u32x4*)(x + (long long)(m0 + (m < mrows ? m : 0)) * ldx + off[j]);
        if constexpr (LN) {
#pragma unroll
          for (int q = 0; q < VEC / 4; ++q) {
            gv[j][q] = *(const f32x4*)(ln_g + off[j] + 4 * q);
            bv[j][q] = *(const f32x4*)(ln_b + off[j] + 4 * q);
          }
        }
      }
      // keep the issue order (the scheduler would put the weight loads first)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) wv[j][r] = *(const u32x4*)(W + (long long)(n0 + r) * ldw + off[j]);
      if constexpr (LN) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          if (m >= mrows) break;  // uniform; the rows beyond M are copies of row 0 and never stored
          float s1 = 0.f;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (ok[j]) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if constexpr (sizeof(T) == 2) s1 += half_lo(xv[j][m][e]) + half_hi(xv[j][m][e]);
                else s1 += __uint_as_float(xv[j][m][e]);
              }
            }
          const float mean = wave_sum(s1) / (float)K;
          float s2 = 0.f;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (ok[j]) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if constexpr (sizeof(T) == 2) {
                  const float d0 = half_lo(xv[j][m][e]) - mean, d1 = half_hi(xv[j][m][e]) - mean;
                  s2 = fmaf(d0, d0, s2);
                  s2 = fmaf(d1, d1, s2);
                } else {
                  const float d0 = __uint_as_float(xv[j][m][e]) - mean;
                  s2 = fmaf(d0, d0, s2);
                }
              }
            }
          const float rstd = rsqrtf(wave_sum(s2) / (float)K + ln_eps);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if constexpr (sizeof(T) == 2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float h0 = (half_lo(xv[j][m][e]) - mean) * rstd * gv[j][e >> 1][(2 * e) & 3] + bv[j][e >> 1][(2 * e) & 3];
                const float h1 = (half_hi(xv[j][m][e]) - mean) * rstd * gv[j][e >> 1][(2 * e + 1) & 3] +
                                 bv[j][e >> 1][(2 * e + 1) & 3];
                xv[j][m][e] = pack_bf16x2(h0, h1);
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                xv[j][m][e] = __float_as_uint((__uint_as_float(xv[j][m][e]) - mean) * rstd * gv[j][0][e] + bv[j][0][e]);
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (ok[j]) {
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  acc[r][m] = fmaf(half_lo(wv[j][r][e]), half_lo(xv[j][m][e]), acc[r][m]);
                  acc[r][m] = fmaf(half_hi(wv[j][r][e]), half_hi(xv[j][m][e]), acc[r][m]);
                }
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r][m] = fmaf(__uint_as_float(wv[j][r][e]), __uint_as_float(xv[j][m][e]), acc[r][m]);
              }
            }
        }
    }
  }
  // optional fused LayerNorm of the input rows (ln_g != null): y = W LN(x).  Every wave derives the row statistics
  // itself (two passes over the row, which sits in L1), then normalises the chunks it multiplies - in the bf16 lane
  // the normalised value is rounded to bf16 first, exactly what the separate LayerNorm kernel would have stored.
  float mu[MB], rs[MB];
  if (LN && !done) {
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      mu[m] = 0.f;
      rs[m] = 1.f;
      if (m < mrows) {
        const T* xr = x + (long long)(m0 + m) * ldx;
        float s1 = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
          const u32x4 xv = *(const u32x4*)(xr + (long long)c * VEC);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (sizeof(T) == 2) s1 += half_lo(xv[e]) + half_hi(xv[e]);
            else s1 += __uint_as_float(xv[e]);
          }
        }
        const float mean = wave_sum(s1) / (float)K;
        float s2 = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
          const u32x4 xv = *(const u32x4*)(xr + (long long)c * VEC);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (sizeof(T) == 2) {
              const float d0 = half_lo(xv[e]) - mean, d1 = half_hi(xv[e]) - mean;
              s2 = fmaf(d0, d0, s2);
              s2 = fmaf(d1, d1, s2);
            } else {
              const float d0 = __uint_as_float(xv[e]) - mean;
              s2 = fmaf(d0, d0, s2);
            }
          }
        }
        mu[m] = mean;
        rs[m] = rsqrtf(wave_sum(s2) / (float)K + ln_eps);
      }
    }
  }
#pragma unroll 2
  for (int c = done ? nchunk : wv_id * 64 + lane; c < nchunk; c += 64 * nwave) {
    u32x4 wv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wv[r] = *(const u32x4*)(W + (long long)(n0 + r) * ldw + (long long)c * VEC);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      if (m < mrows) {
        u32x4 xv = *(const u32x4*)(x + (long long)(m0 + m) * ldx + (long long)c * VEC);
        if constexpr (LN) {
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int k = c * VEC + 2 * e;
              const float h0 = (half_lo(xv[e]) - mu[m]) * rs[m] * ln_g[k] + ln_b[k];
              const float h1 = (half_hi(xv[e]) - mu[m]) * rs[m] * ln_g[k + 1] + ln_b[k + 1];
              xv[e] = pack_bf16x2(h0, h1);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int k = c * VEC + e;
              xv[e] = __float_as_uint((__uint_as_float(xv[e]) - mu[m]) * rs[m] * ln_g[k] + ln_b[k]);
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[r][m] = fmaf(half_lo(wv[r][e]), half_lo(xv[e]), acc[r][m]);
              acc[r][m] = fmaf(half_hi(wv[r][e]), half_hi(xv[e]), acc[r][m]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][m] = fmaf(__uint_as_float(wv[r][e]), __uint_as_float(xv[e]), acc[r][m]);
          }
        }
      }
    }
  }
  // wave reduction of the 4 x MB partial sums (lane r * MB + m keeps output (m, n0 + r)), then across the waves
  float mine = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      if (m >= mrows) continue;  // uniform
      const float s = wave_sum(acc[r][m]);
      if (lane == r * MB + m) mine = s;
    }
  __shared__ float part[4][64];
  if (nwave > 1) {
    part[wv_id][lane] = mine;
    __syncthreads();
    if (wv_id == 0) {
      mine = part[0][lane];
      for (int q = 1; q < nwave; ++q) mine += part[q][lane];
    }
  }
  if (wv_id == 0 && lane < 4 * MB) {
    const int r = lane / MB, m = lane % MB;
    if (m < mrows) {
      const int n = n0 + r;
      float v = mine + (bias ? e_bias : 0.f);
      if (act == MELGPT_ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
      if (res) v += ldf(&e_res_raw);
      if (out_f32) ((float*)y)[(long long)(m0 + m) * ldy + n] = v;
      else stf((T*)y + (long long)(m0 + m) * ldy + n, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------- skinny MFMA linear
// y (M, N) = epi(x (M, K) W (N, K)^T + bias) (+ residual) for 5 .. 128 rows (decode steps at batch 5 .. 128), bf16.
// The tiled GEMMs give such a problem one row of 128-wide tiles - 8 to 32 workgroups on a 256-CU chip - and the VALU
// weight-streaming kernel above re-reads the weights once per 16 rows.  Here a workgroup owns 16 output columns: its
// four waves split K in 128-element chunks, each streams its part of the 16 weight rows ONCE (A operand straight from
// memory, 16-byte fragments) against all rows of x (B operand, from L2), MFMA 16x16x32 with the output columns on the
// accumulator rows - so a lane ends up with 4 consecutive columns of one row; partial tiles are summed across the waves
// in LDS in wave order (deterministic) and the epilogue runs on the summing wave.  N / 16 workgroups: 64 .. 256 for the
// VAS layer shapes whatever M is.
// LN: the rows of x are LayerNorm-ed on the way in (y = W LN(x), the block's pre-LN folded into qkv / fc1 as in
// gemv_rows_kernel).  K is 512 or 1024 then, so a wave's two steps hold ALL of its share of x in registers: the row
// statistics (two passes, like the LayerNorm kernel) come from those registers - partial sums over the lane's
// fragments, over the four k-groups of the wave by shuffle, over the four waves through LDS - and the fragments are
// normalised and rounded to bf16 in place before the MFMAs.  Costs two workgroup barriers instead of a launch.
template <int MT, bool LN>  // 16-row tiles of x
__global__ __launch_bounds__(256) void linear_skinny_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                            const bf16_t* __restrict__ W, long long ldw,
                                                            const float* __restrict__ bias, const bf16_t* __restrict__ res,
                                                            long long ldr, void* __restrict__ y, long long ldy, int M, int N,
                                                            int K, int act, int out_f32, const float* ln_g,
                                                            const float* ln_b, float ln_eps) {
  constexpr int KS = MT <= 4 ? 4 : 2;  // k-steps (32 elements) whose loads are in flight together
  constexpr int NW = 4;
  __shared__ f32x4 part[NW - 1][MT][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16, m_base = blockIdx.y * 16 * MT;  // blockIdx.y: row blocks (same weights, same XCD)
  const bf16_t* wp = W + (long long)(n0 + r16) * ldw + 8 * g;
  const bf16_t* xp[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) xp[mt] = x + (long long)min(m_base + 16 * mt + r16, M - 1) * ldx + 8 * g;
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // wave w takes the 128-element chunks w, w + 4, ... in steps of KS k-steps; the loads of the next step are issued
  // before the MFMAs of the current one (two register sets, A / B), so a wave pays one memory round trip, not one per step
  constexpr int H = 4 / KS;
  constexpr int KW = 128 * NW;  // K covered by one round of the workgroup
  const int nst = 128 * w < K ? ((K - 128 * w + KW - 1) / KW) * H : 0;
  auto load = [&](int i, u32x4(&a)[KS], u32x4(&b)[KS][MT]) {
    i = min(i, nst - 1);  // past the end: a redundant reload instead of a branch around loads
    const int kb = 128 * w + KW * (i / H) + 32 * KS * (i % H);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      a[s] = *(const u32x4*)(wp + kb + 32 * s);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) b[s][mt] = *(const u32x4*)(xp[mt] + kb + 32 * s);
    }
  };
  auto mma = [&](const u32x4(&a)[KS], const u32x4(&b)[KS][MT]) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = MELGPT_MFMA_16x16x32(a[s], b[s][mt], acc[mt]);
  };
  if constexpr (LN) {
    // host: K % 512 == 0 and K <= 1024, so 1 <= nst <= 2 for every wave (KS = 4: MT <= 4)
    __shared__ float lnred[2][NW][MT][16];
    __shared__ f32x4 lng[256], lnb[256];  // gamma / beta of the whole row, staged once per workgroup
    u32x4 aA[KS], bA[KS][MT], aB[KS], bB[KS][MT];
    const bool two = nst > 1;
    const int t4 = min((int)threadIdx.x, K / 4 - 1);  // K = 512: the upper half rewrites the last quad
    const f32x4 g4 = *(const f32x4*)(ln_g + 4 * t4), b4 = *(const f32x4*)(ln_b + 4 * t4);
    load(0, aA, bA);
    load(1, aB, bB);  // nst == 1: a copy of step 0, left out below
    lng[t4] = g4;     // (issued first, so they are the first to arrive; visible after the barrier of pass 0)
    lnb[t4] = b4;
    float mean[MT], rstd[MT];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float p = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a0 = bf16lo(bA[s][mt][e]), a1 = bf16hi(bA[s][mt][e]);
            const float b0 = two ? bf16lo(bB[s][mt][e]) : 0.f, b1 = two ? bf16hi(bB[s][mt][e]) : 0.f;
            if (pass == 0) {
              p += (a0 + a1) + (b0 + b1);
            } else {
              const float mu = mean[mt];
              p = fmaf(a0 - mu, a0 - mu, p);
              p = fmaf(a1 - mu, a1 - mu, p);
              if (two) {
                p = fmaf(b0 - mu, b0 - mu, p);
                p = fmaf(b1 - mu, b1 - mu, p);
              }
            }
          }
        p += __shfl_xor(p, 16, 64);
        p += __shfl_xor(p, 32, 64);
        if (g == 0) lnred[pass][w][mt][r16] = p;
      }
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float t = (lnred[pass][0][mt][r16] + lnred[pass][1][mt][r16]) + (lnred[pass][2][mt][r16] + lnred[pass][3][mt][r16]);
        if (pass == 0) mean[mt] = t / (float)K;
        else rstd[mt] = rsqrtf(t / (float)K + ln_eps);
      }
    }
    auto norm = [&](u32x4(&b)[KS][MT], int st) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int q0 = (128 * w + KW * st + 32 * s + 8 * g) / 4;
        const f32x4 gm[2] = {lng[q0], lng[q0 + 1]}, bt[2] = {lnb[q0], lnb[q0 + 1]};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float h0 = (bf16lo(b[s][mt][e]) - mean[mt]) * rstd[mt] * gm[e >> 1][(2 * e) & 3] + bt[e >> 1][(2 * e) & 3];
            const float h1 = (bf16hi(b[s][mt][e]) - mean[mt]) * rstd[mt] * gm[e >> 1][(2 * e + 1) & 3] + bt[e >> 1][(2 * e + 1) & 3];
            b[s][mt][e] = pack_bf16x2(h0, h1);
          }
      }
    };
    norm(bA, 0);
    mma(aA, bA);
    if (two) {
      norm(bB, 1);
      mma(aB, bB);
    }
  } else if (nst > 0) {
    u32x4 aA[KS], bA[KS][MT], aB[KS], bB[KS][MT];
    load(0, aA, bA);
    int i = 0;
    for (; i + 1 < nst; i += 2) {  // both halves unconditional: a load used only under a branch is sunk into it
      load(i + 1, aB, bB);
      __builtin_amdgcn_sched_barrier(0);  // the scheduler would move the loads below the MFMAs (register pressure)
      mma(aA, bA);
      __builtin_amdgcn_sched_barrier(0);
      load(i + 2, aA, bA);
      __builtin_amdgcn_sched_barrier(0);
      mma(aB, bB);
    }
    if (i < nst) mma(aA, bA);  // odd tail
  }
  if (w > 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) part[w - 1][mt][lane] = acc[mt];
  }
  __syncthreads();
  if (w == 0) {
    // acc[mt][v] = y[16 mt + r16][n0 + 4 g + v]
    const int n = n0 + 4 * g;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *(const f32x4*)(bias + n);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m_base + 16 * mt + r16;
      f32x4 v = acc[mt];
#pragma unroll
      for (int q = 0; q < NW - 1; ++q) v += part[q][mt][lane];  // wave order: deterministic
      v += bv;
      if (act == MELGPT_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
      }
      if (m < M) {
        if (res) {
          const u32x2 rv = *(const u32x2*)(res + (long long)m * ldr + n);
          v += f32x4{bf16lo(rv[0]), bf16hi(rv[0]), bf16lo(rv[1]), bf16hi(rv[1])};
        }
        if (out_f32) *(f32x4*)((float*)y + (long long)m * ldy + n) = v;
        else *(u32x2*)((bf16_t*)y + (long long)m * ldy + n) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
    }
  }
}


// ------------------------------------------------------------------------------------ LDS-resident skinny linear
// The same product for 9 .. 128 rows with EVERY byte a workgroup needs requested up front.  linear_skinny_kernel keeps two
// register sets of loads in flight (~10 KB per workgroup): at 64 rows its 192-256 workgroups each walk K = 1024 in 16
// dependent round trips and a launch takes 13-15 us for 6-8 MB of weights (profiles/r03_decode_lab.md) - latency, not
// bandwidth.  Here a workgroup owns 16 output columns x ONE 1024-wide slice of K x up to 64 rows:
//   * its slice of x (rows x 2 KB, up to 128 KB) goes to LDS by LDS-DMA - 1 KB pieces, all issued at once, no registers -
//     with the 16-byte chunk index XOR-ed with the row on the SOURCE side and again on the fragment read (conflict-free
//     ds_read_b128 across the 16 rows of an MFMA operand);
//   * its 32 KB of weights go straight to registers as MFMA A fragments (8 x 16 bytes per lane: wave w takes k in
//     [256 w, 256 w + 256) of the slice), requested behind the DMA pieces;
//   * ONE wait, one barrier, then 8 k-steps x (rows / 16) MFMAs per wave on operands that are all on chip;
//   * the four waves' partial tiles are summed through LDS in wave order (deterministic), wave w finishing row tile w.
// K > 1024 (fc2: 4096): gridDim.y slices, each writing an f32 partial tile to `part` (slice-major); the host sums them
// with splitk_epilogue_kernel (fixed order) which also applies bias / residual.  K = 1024: the epilogue runs here.
// LN: the block's pre-LayerNorm folded in ALGEBRAICALLY (K = 1024 only; melgpt_ln_fold_prepare, once per weight version):
//   W LN(x) + b = rstd (W' x - mu c1) + c2,   W' = W diag(gamma),  c1 = W' 1,  c2 = W beta + b
// so the kernel multiplies the RAW rows by the prepared W' and the epilogue applies the two per-row scalars; mu and
// sum x^2 come off the matrix pipe too (ones . x and the diagonal of x x^T on the fragments already in registers).
// Normalising the rows in place in LDS first (four lanes per row, 1 600 VALU instructions per lane on one wave per SIMD)
// cost 8 of such a launch's 14 us (profiles/r03_decode_lab.md).
typedef u32x4 dec_rsrc_t;
__device__ __forceinline__ void dec_dma16(dec_rsrc_t rs, char* lds_wave_base, unsigned voff) {
  // (inline asm for the reasons given at dma16 in gemm256.hip: hipcc would complete the builtin only at vmcnt(0))
  const unsigned m0v = (unsigned)(size_t)LDS_PTR(char, lds_wave_base);
  asm volatile("s_nop 2\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :
               : "s"(m0v), "v"(voff), "s"(rs)
               : "memory", "m0");
}
__device__ __forceinline__ int xl_off(int row, int chunk) {  // 2 KB rows of 128 chunks; low 4 chunk bits XOR row
  return row * 2048 + ((chunk ^ (row & 15)) << 4);
}

template <int MT, bool LN>  // MT 16-row tiles of x per workgroup (1, 2 or 4)
__global__ __launch_bounds__(256) void linear_lds_kernel(const bf16_t* __restrict__ x, long long ldx, unsigned x_bytes,
                                                         const bf16_t* __restrict__ W, long long ldw,
                                                         const float* __restrict__ bias, const bf16_t* __restrict__ res,
                                                         long long ldr, void* __restrict__ y, long long ldy,
                                                         float* __restrict__ part, int M, int N, int act, int out_f32,
                                                         const float* ln_c1, const float* ln_c2, float ln_eps) {
  constexpr int KSL = 1024, NW = 4, ROWS = 16 * MT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xl = smem;                                        // [ROWS][2 KB] swizzled
  f32x4* psum = (f32x4*)(smem + ROWS * 2048);             // [NW][MT][64]
  float* pst = (float*)(psum + NW * MT * 64);             // LN: [NW][MT][64][2] partial sum x, sum x^2 of the lane's row
  const int t = threadIdx.x, lane = t & 63, r16 = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n0 = blockIdx.x * 16, k0 = blockIdx.y * KSL, m_base = blockIdx.z * ROWS;
  const int mrows = min(ROWS, M - m_base);
  // ---- x slice -> LDS: piece p = (row p >> 1, half p & 1), 2 * ROWS pieces over 4 waves; rows past M read zeros (OOB)
  {
    const unsigned long long a64 = (unsigned long long)x;
    dec_rsrc_t rs = {(unsigned)a64, (unsigned)(a64 >> 32) & 0xFFFFu, x_bytes, 0x00020000u};
#pragma unroll
    for (int e = 0; e < 4; ++e) rs[e] = __builtin_amdgcn_readfirstlane(rs[e]);
#pragma unroll
    for (int j = 0; j < 2 * ROWS / NW; ++j) {
      const int p = w + NW * j, row = p >> 1, cp = 64 * (p & 1) + lane;   // physical chunk this lane fills
      const int cl = cp ^ (row & 15);                                       // ... holds logical chunk cl
      const unsigned off = row < mrows ? (unsigned)(((long long)(m_base + row) * ldx + k0 + 8 * cl) * 2) : 0xFFFFFFF0u;
      dec_dma16(rs, xl + p * 1024, off);
    }
  }
  // ---- weights -> registers (A fragments: output column n0 + r16, k = k0 + 256 w + 32 ks + 8 g ..+7)
  u32x4 a[8];
  {
    const bf16_t* wp = W + (long long)(n0 + r16) * ldw + k0 + 256 * w + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) a[ks] = *(const u32x4*)(wp + 32 * ks);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces (and, in order behind them, its weights) have landed
  __syncthreads();
  // ---- 8 k-steps x MT row tiles: B fragment = x rows 16 mt + r16, logical chunk (256 w + 32 ks) / 8 + g
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 s1t[LN ? MT : 1], s2t[LN ? MT : 1];  // LN: ones . x (every row = sum_k x[m][k]) and x x^T (diagonal = sum_k x[m][k]^2)
  if constexpr (LN) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) s1t[mt] = s2t[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const unsigned one2 = pack_bf16x2(1.0f, 1.0f);
  const u32x4 ones = {one2, one2, one2, one2};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    u32x4 b[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) b[mt] = *(const u32x4*)(xl + xl_off(16 * mt + r16, 32 * w + 4 * ks + g));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[mt] = MELGPT_MFMA_16x16x32(a[ks], b[mt], acc[mt]);
      if constexpr (LN) {
        s1t[mt] = MELGPT_MFMA_16x16x32(ones, b[mt], s1t[mt]);
        s2t[mt] = MELGPT_MFMA_16x16x32(b[mt], b[mt], s2t[mt]);
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    psum[(w * MT + mt) * 64 + lane] = acc[mt];
    if constexpr (LN) {
      // D[i][j] of x x^T sits in lane (j = r16, g = i >> 2) element i & 3: the diagonal of column r16 is in the lane
      // with g == r16 >> 2; the other three lanes of the column contribute 0 and read the total back by shuffle below
      const float dg = (g == (r16 >> 2)) ? (((r16 & 3) == 0) ? s2t[mt][0] : ((r16 & 3) == 1) ? s2t[mt][1]
                                                              : ((r16 & 3) == 2) ? s2t[mt][2] : s2t[mt][3])
                                         : 0.f;
      pst[((w * MT + mt) * 64 + lane) * 2] = s1t[mt][0];
      pst[((w * MT + mt) * 64 + lane) * 2 + 1] = dg;
    }
  }
  __syncthreads();
  // ---- wave w finishes row tiles w, w + 4, ..: sum of the four waves' partials in wave order, then the epilogue.
  // v[e] = y[m_base + 16 mt + r16][n0 + 4 g + e]
  for (int mt = w; mt < MT; mt += NW) {
    f32x4 v = psum[(0 * MT + mt) * 64 + lane];
#pragma unroll
    for (int q = 1; q < NW; ++q) v += psum[(q * MT + mt) * 64 + lane];
    const int m = m_base + 16 * mt + r16, n = n0 + 4 * g;
    if constexpr (LN) {  // (before the row predicate: the shuffle needs every lane)
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        s1 += pst[((q * MT + mt) * 64 + lane) * 2];
        s2 += pst[((q * MT + mt) * 64 + lane) * 2 + 1];
      }
      s2 = __shfl(s2, r16 + 16 * (r16 >> 2), 64);
      const float mu = s1 * (1.0f / KSL);
      const float rstd = rsqrtf(fmaxf(s2 * (1.0f / KSL) - mu * mu, 0.f) + ln_eps);
      const f32x4 c1 = *(const f32x4*)(ln_c1 + n), c2 = *(const f32x4*)(ln_c2 + n);
      v = (v - c1 * mu) * rstd + c2;   // = W LN(x) + b
    }
    if (m >= M) continue;
    if (part) {  // one K slice of several: raw partial sums, slice-major
      *(f32x4*)(part + ((long long)blockIdx.y * M + m) * N + n) = v;
      continue;
    }
    if (!LN && bias) v += *(const f32x4*)(bias + n);   // (LN: the bias is inside c2)
    if (act == MELGPT_ACT_GELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
    }
    if (res) {
      const u32x2 rv = *(const u32x2*)(res + (long long)m * ldr + n);
      v += f32x4{bf16lo(rv[0]), bf16hi(rv[0]), bf16lo(rv[1]), bf16hi(rv[1])};
    }
    if (out_f32) *(f32x4*)((float*)y + (long long)m * ldy + n) = v;
    else *(u32x2*)((bf16_t*)y + (long long)m * ldy + n) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  }
}

// melgpt_ln_fold_prepare: W' = bf16(W gamma) (N, K), c1[n] = sum_k W'[n][k] (of the ROUNDED values), c2[n] = sum_k W[n][k]
// beta[k] + bias[n] - one wave per output row, fixed summation order
__global__ __launch_bounds__(256) void ln_fold_prepare_kernel(const bf16_t* __restrict__ W, long long ldw,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int N, int K,
                                                              bf16_t* __restrict__ Wf, float* __restrict__ c1,
                                                              float* __restrict__ c2) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float a1 = 0.f, a2 = 0.f;
  for (int k = 8 * lane; k < K; k += 512) {
    const u32x4 wv = *(const u32x4*)(W + (long long)n * ldw + k);
    const f32x4 g0 = *(const f32x4*)(gamma + k), g1 = *(const f32x4*)(gamma + k + 4);
    const f32x4 b0 = *(const f32x4*)(beta + k), b1 = *(const f32x4*)(beta + k + 4);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float wl = bf16lo(wv[e]), wh = bf16hi(wv[e]);
      const float gl = e < 2 ? g0[2 * e] : g1[2 * e - 4], gh = e < 2 ? g0[2 * e + 1] : g1[2 * e - 3];
      const float bl = e < 2 ? b0[2 * e] : b1[2 * e - 4], bh = e < 2 ? b0[2 * e + 1] : b1[2 * e - 3];
      o[e] = pack_bf16x2(wl * gl, wh * gh);
      a1 += bf16lo(o[e]) + bf16hi(o[e]);
      a2 = fmaf(wl, bl, a2);
      a2 = fmaf(wh, bh, a2);
    }
    *(u32x4*)(Wf + (long long)n * K + k) = o;
  }
  a1 = wave_sum(a1);
  a2 = wave_sum(a2);
  if (lane == 0) {
    c1[n] = a1;
    c2[n] = a2 + (bias ? bias[n] : 0.f);
  }
}

// y = epi(sum over the K slices of part (slices, M, N) f32, in slice order) - bias, GELU, residual as linear_lds_kernel
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ part, int slices, int M, int N,
                                                              const float* __restrict__ bias,
                                                              const bf16_t* __restrict__ res, long long ldr,
                                                              void* __restrict__ y, long long ldy, int act, int out_f32) {
  const long long quads = (long long)M * (N / 4);
  const long long stride = (long long)M * N;
  for (long long qd = (long long)blockIdx.x * 256 + threadIdx.x; qd < quads; qd += (long long)gridDim.x * 256) {
    const int m = (int)(qd / (N / 4)), n = 4 * (int)(qd % (N / 4));
    const float* p = part + (long long)m * N + n;
    f32x4 v = *(const f32x4*)p;
    for (int sl = 1; sl < slices; ++sl) v += *(const f32x4*)(p + sl * stride);
    if (bias) v += *(const f32x4*)(bias + n);
    if (act == MELGPT_ACT_GELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
    }
    if (res) {
      const u32x2 rv = *(const u32x2*)(res + (long long)m * ldr + n);
      v += f32x4{bf16lo(rv[0]), bf16hi(rv[0]), bf16lo(rv[1]), bf16hi(rv[1])};
    }
    if (out_f32) *(f32x4*)((float*)y + (long long)m * ldy + n) = v;
    else *(u32x2*)((bf16_t*)y + (long long)m * ldy + n) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  }
}

}  // namespace

extern "C" int melgpt_linear_skinny(const void* x, long long ldx, const void* W, long long ldw, const float* bias,
                                    const void* residual, long long ldr, void* y, long long ldy, int M, int N, int K,
                                    int act, int dtype, int out_f32, const float* ln_gamma, const float* ln_beta,
                                    float ln_eps, void* stream) {
  MELGPT_CHECK(x && W && y && M > 0 && N > 0 && K > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_BF16 && M <= 128, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(act == MELGPT_ACT_NONE || act == MELGPT_ACT_GELU, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(N % 16 == 0 && K % 128 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldy % 4 == 0 && (!residual || ldr % 4 == 0),
               MELGPT_ERR_ALIGN);
  MELGPT_CHECK((((uintptr_t)x | (uintptr_t)W | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)bias) & 15) == 0,
               MELGPT_ERR_ALIGN);
  MELGPT_CHECK((ln_gamma == nullptr) == (ln_beta == nullptr), MELGPT_ERR_BAD_ARG);
  // fused LayerNorm: a wave must hold its whole share of the rows in two register sets
  MELGPT_CHECK(!ln_gamma || ((K == 512 || K == 1024) && M <= 64 && ((((uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) == 0)),
               MELGPT_ERR_UNSUPPORTED);
  hipStream_t s = (hipStream_t)stream;
#define MELGPT_SKINNY_LAUNCH_LN(MT, LN)                                                                                \
  hipLaunchKernelGGL((linear_skinny_kernel<MT, LN>), dim3(N / 16, (mt_all + MT - 1) / MT), dim3(256), 0, s,             \
                     (const bf16_t*)x, ldx, (const bf16_t*)W, ldw, bias, (const bf16_t*)residual, ldr, y, ldy, M, N, K, \
                     act, out_f32, ln_gamma, ln_beta, ln_eps)
#define MELGPT_SKINNY_LAUNCH(MT)                          \
  do {                                                    \
    if (MT <= 4 && ln_gamma) MELGPT_SKINNY_LAUNCH_LN((MT <= 4 ? MT : 4), true); \
    else MELGPT_SKINNY_LAUNCH_LN(MT, false);              \
  } while (0)
  // N / 16 workgroups stream the weights; when that leaves CUs idle (N = 1024: 64) the rows are split over up to
  // 256 / (N / 16) workgroups per column block, which read the same weight slab (from L2 after the first).
  // What bounds these launches is the bytes ONE CU takes in (~16-25 GB/s per CU, L2 hits included: its share of W plus
  // all the x rows it multiplies): fc2 (N 1024, K 4096) costs ~12 us at any batch because every workgroup reads 128 KB
  // of weights plus its rows of x over K = 4096; 4-column workgroups (x read 4 x as often) and 16 waves per workgroup
  // measured the same or worse - only a split of K over workgroups (a cross-workgroup reduction) would cut it.
  const int mt_all = (M + 15) / 16, split = 256 / (N / 16) > 1 ? 256 / (N / 16) : 1;
  const int mt = (mt_all + split - 1) / split;
  if (mt <= 1) MELGPT_SKINNY_LAUNCH(1);
  else if (mt <= 2) MELGPT_SKINNY_LAUNCH(2);
  else if (mt <= 4) MELGPT_SKINNY_LAUNCH(4);
  else MELGPT_SKINNY_LAUNCH(8);
#undef MELGPT_SKINNY_LAUNCH
#undef MELGPT_SKINNY_LAUNCH_LN
  return melgpt_launch_status();
}

extern "C" int melgpt_gemv_rows(const void* x, long long ldx, const void* W, long long ldw, const float* bias,
                                const void* residual, long long ldr, void* y, long long ldy, int M, int N, int K, int act,
                                int dtype, int out_f32, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                void* stream) {
  MELGPT_CHECK(x && W && y && M > 0 && N > 0 && K > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(act == MELGPT_ACT_NONE || act == MELGPT_ACT_GELU, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK((ln_gamma == nullptr) == (ln_beta == nullptr), MELGPT_ERR_BAD_ARG);
  const int vec = dtype == MELGPT_F32 ? 4 : 8;
  MELGPT_CHECK(N % 4 == 0 && K % vec == 0 && ldx % vec == 0 && ldw % vec == 0 &&
                   ((((uintptr_t)x | (uintptr_t)W) & 15) == 0),
               MELGPT_ERR_ALIGN);
  hipStream_t s = (hipStream_t)stream;
  int nwave = (K / vec + 127) / 128;  // ~2 chunks of 16 bytes per lane
  nwave = nwave < 1 ? 1 : nwave > 4 ? 4 : nwave;
  // 16 rows of x per wave; fewer accumulators are instantiated for the common single-digit decode batches (1, 2-4)
#define MELGPT_GEMV_LAUNCH_LN(T, MB, LN)                                                                              \
  hipLaunchKernelGGL((gemv_rows_kernel<T, MB, LN>), dim3(N / 4, (M + MB - 1) / MB), dim3(64 * nwave), 0, s, (const T*)x, \
                     ldx, (const T*)W, ldw, bias, (const T*)residual, ldr, y, ldy, M, N, K, act,                        \
                     (dtype == MELGPT_F32) ? 1 : out_f32, ln_gamma, ln_beta, ln_eps)
#define MELGPT_GEMV_LAUNCH(T, MB)                    \
  do {                                               \
    if (ln_gamma) MELGPT_GEMV_LAUNCH_LN(T, MB, true); \
    else MELGPT_GEMV_LAUNCH_LN(T, MB, false);         \
  } while (0)
  if (dtype == MELGPT_F32) {
    if (M == 1) MELGPT_GEMV_LAUNCH(float, 1);
    else if (M <= 4) MELGPT_GEMV_LAUNCH(float, 4);
    else MELGPT_GEMV_LAUNCH(float, 16);
  } else {
    if (M == 1) MELGPT_GEMV_LAUNCH(bf16_t, 1);
    else if (M <= 4) MELGPT_GEMV_LAUNCH(bf16_t, 4);
    else MELGPT_GEMV_LAUNCH(bf16_t, 16);
  }
#undef MELGPT_GEMV_LAUNCH
#undef MELGPT_GEMV_LAUNCH_LN
  return melgpt_launch_status();
}

extern "C" long long melgpt_linear_lds_workspace(int M, int N, int K) {
  return K > 1024 ? (long long)(K / 1024) * M * N * 4 : 0;  // bytes of f32 partial tiles (K slices of 1024)
}

extern "C" int melgpt_linear_lds(const void* x, long long ldx, const void* W, long long ldw, const float* bias,
                                 const void* residual, long long ldr, void* y, long long ldy, int M, int N, int K, int act,
                                 int dtype, int out_f32, const float* ln_c1, const float* ln_c2, float ln_eps,
                                 void* workspace, void* stream) {
  MELGPT_CHECK(x && W && y && M > 0 && N > 0 && K > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_BF16 && M <= 128 && K % 1024 == 0 && K <= 8192, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(act == MELGPT_ACT_NONE || act == MELGPT_ACT_GELU, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(N % 16 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldy % 4 == 0 && (!residual || ldr % 4 == 0), MELGPT_ERR_ALIGN);
  MELGPT_CHECK((((uintptr_t)x | (uintptr_t)W | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)bias | (uintptr_t)workspace) & 15) == 0,
               MELGPT_ERR_ALIGN);
  MELGPT_CHECK((ln_c1 == nullptr) == (ln_c2 == nullptr), MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(!ln_c1 || (K == 1024 && ((((uintptr_t)ln_c1 | (uintptr_t)ln_c2) & 15) == 0)), MELGPT_ERR_UNSUPPORTED);
  const int slices = K / 1024;
  MELGPT_CHECK(slices == 1 || workspace, MELGPT_ERR_BAD_ARG);
  const long long xb = ((long long)(M - 1) * ldx + K) * 2;
  MELGPT_CHECK(xb < 0xFFFFFF00LL, MELGPT_ERR_UNSUPPORTED);
  hipStream_t s = (hipStream_t)stream;
  float* part = slices > 1 ? (float*)workspace : nullptr;
  const int mt_all = (M + 15) / 16;
  const int MTsel = mt_all <= 1 ? 1 : (mt_all <= 2 ? 2 : 4);
  const int zb = (mt_all + MTsel - 1) / MTsel;
#define MELGPT_LDS_LAUNCH(MT, LN)                                                                                      \
  do {                                                                                                                 \
    const size_t lds = (size_t)16 * MT * 2048 + (size_t)4 * MT * 64 * 16 + (LN ? (size_t)4 * MT * 64 * 8 : 0);                                \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      if (hipFuncSetAttribute((const void*)linear_lds_kernel<MT, LN>, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                              (int)lds) != hipSuccess)                                                                 \
        return MELGPT_ERR_LAUNCH;                                                                                      \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((linear_lds_kernel<MT, LN>), dim3(N / 16, slices, zb), dim3(256), lds, s, (const bf16_t*)x, ldx, \
                       (unsigned)xb, (const bf16_t*)W, ldw, bias, (const bf16_t*)residual, ldr, y, ldy, part, M, N, act, \
                       out_f32, ln_c1, ln_c2, ln_eps);                                                                 \
  } while (0)
  if (ln_c1) {
    if (MTsel == 1) MELGPT_LDS_LAUNCH(1, true);
    else if (MTsel == 2) MELGPT_LDS_LAUNCH(2, true);
    else MELGPT_LDS_LAUNCH(4, true);
  } else {
    if (MTsel == 1) MELGPT_LDS_LAUNCH(1, false);
    else if (MTsel == 2) MELGPT_LDS_LAUNCH(2, false);
    else MELGPT_LDS_LAUNCH(4, false);
  }
#undef MELGPT_LDS_LAUNCH
  if (slices > 1) {
    const long long quads = (long long)M * (N / 4);
    const int grid = (int)((quads + 255) / 256 < 1024 ? (quads + 255) / 256 : 1024);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(grid), dim3(256), 0, s, part, slices, M, N, bias,
                       (const bf16_t*)residual, ldr, y, ldy, act, out_f32);
  }
  return melgpt_launch_status();
}

extern "C" int melgpt_ln_fold_prepare(const void* W, long long ldw, const float* bias, const float* gamma,
                                      const float* beta, int N, int K, int dtype, void* W_folded, float* c1, float* c2,
                                      void* stream) {
  MELGPT_CHECK(W && gamma && beta && W_folded && c1 && c2 && N > 0 && K > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(K % 8 == 0 && ldw % 8 == 0, MELGPT_ERR_ALIGN);
  MELGPT_CHECK((((uintptr_t)W | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)W_folded) & 15) == 0, MELGPT_ERR_ALIGN);
  hipLaunchKernelGGL(ln_fold_prepare_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, ldw,
                     bias, gamma, beta, N, K, (bf16_t*)W_folded, c1, c2);
  return melgpt_launch_status();
}
