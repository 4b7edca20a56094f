// HBM-bound row kernels of the minGPT training step for gfx950: LayerNorm, embedding stem, cross entropy,
// dropout replay, column sums, fused AdamW, casts.  One 64-lane wavefront owns one row; every global access
// is a 16-byte vector; reductions are xor-shuffles inside the wave (no LDS, no atomics -> deterministic).
// Reference call sites: transformer/minGPT.py:97-98,141 (nn.LayerNorm), :170-180 (tok_emb/pos_emb/drop),
// :197,:416 (F.cross_entropy), :660-664 (AdamW); transformer/decoders.py:20-21,64-68 (per-token CE).
#include <type_traits>

#include "common.h"

namespace {

template <typename T>
struct V16;  // a 16-byte vector of T unpacked to floats
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

constexpr int LN_MAXCH = 8;  // chunks per lane kept in registers: C <= 64*8*(16/ES) = 4096 bf16 / 2048 f32

__device__ __forceinline__ void keep_mask(unsigned long long seed, unsigned sid, unsigned long long e0, int n,
                                          unsigned thresh, float scale, float* v) {
  // e0 is a multiple of 4; n in {4, 8}
  for (int q = 0; q < n / 4; ++q) {
    unsigned k = dropout_keep4(seed, sid, (e0 >> 2) + q, thresh);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * q + e] = (k >> e & 1) ? v[4 * q + e] * scale : 0.f;
  }
}

// ================================================================================== LayerNorm
// NCH = 16-byte chunks per lane held in registers (the row has C / N chunks, NCH = ceil(that / 64) rounded up to a
// power of two): the kernels are instantiated per NCH so that a 1024-wide row costs 2 chunks of registers, not 8.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            long long M, int C, float eps) {
  constexpr int N = V16<T>::N;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = C / N;
  const T* xr = x + row * C;
  float v[NCH][N];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      V16<T>::ld(xr + ch * N, v[i]);
#pragma unroll
      for (int e = 0; e < N; ++e) s += v[i][e];
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (lane + 64 * i < nch) {
#pragma unroll
      for (int e = 0; e < N; ++e) {
        float d = v[i][e] - mu;
        ss = fmaf(d, d, ss);
      }
    }
  }
  const float rs = rsqrtf(wave_sum(ss) / (float)C + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  T* yr = y + row * C;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      float o[N];
#pragma unroll
      for (int e4 = 0; e4 < N; e4 += 4) {
        f32x4 gm = *(const f32x4*)(gamma + ch * N + e4), bt = *(const f32x4*)(beta + ch * N + e4);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e4 + e] = (v[i][e4 + e] - mu) * rs * gm[e] + bt[e];
      }
      V16<T>::st(yr + ch * N, o);
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma   [+ add_in]   [then optional dropout replay
// into a second output `dx_drop` = keep(dx)*scale, which is the gradient of the dropout-ed branch input]
// WPB waves per workgroup: their column sums (the gamma / beta gradients) are combined through LDS into ONE partial row
// per workgroup - 8 where 8 x 2C floats fit 64 KiB (C <= 1024: 512 partial rows instead of 1 024 for the second stage to
// read: 17 -> 9 us per call at C = 1024), 4 otherwise.
template <typename T, int NCH, bool ADD, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const T* __restrict__ add_in, T* __restrict__ dx,
                                                            float* __restrict__ partials, long long M, int C,
                                                            T* __restrict__ dxm, float mscale, unsigned mthresh,
                                                            unsigned long long mseed, unsigned msid) {
  constexpr int N = V16<T>::N;
  typedef typename std::conditional<sizeof(T) == 2, u32x4, f32x4>::type Raw;  // one 16-byte chunk as loaded
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * WPB;
  const int nch = C / N;
  float dg[NCH][N], db[NCH][N];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < N; ++e) dg[i][e] = db[i][e] = 0.f;
  // EVERY load of this kernel is unconditional, with its index clamped into range: a value loaded inside a branch is
  // waited for where the branch ends (`s_waitcnt vmcnt(0)`), which drained the row requested ahead along with it and
  // made each row cost a full memory latency.  Lanes / rows beyond the end compute on copies and store nothing.
  int chc[NCH];   // this lane's chunks, clamped
  bool live[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    live[i] = lane + 64 * i < nch;
    chc[i] = min(lane + 64 * i, nch - 1);
  }
  // gamma chunks of this lane (row-invariant)
  float gm[NCH][N];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
#pragma unroll
    for (int e4 = 0; e4 < N; e4 += 4) {
      const f32x4 gv = *(const f32x4*)(gamma + chc[i] * N + e4);
#pragma unroll
      for (int e = 0; e < 4; ++e) gm[i][e4 + e] = gv[e];
    }
  }

  // A row's three inputs (dy, x, add_in) are requested TOGETHER and one row ahead of their use: a wave handles its
  // rows one after another, so without this every row pays two dependent memory latencies (inputs, then add_in).
  struct RowIn {
    Raw d[NCH], xv[NCH], a[NCH];
    float mu, rs;
  };
  auto load_row = [&](long long row, RowIn& r) {
    row = row < M ? row : M - 1;
    r.mu = mean[row];
    r.rs = rstd[row];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      r.d[i] = *(const Raw*)(dy + row * C + chc[i] * N);
      r.xv[i] = *(const Raw*)(x + row * C + chc[i] * N);
      if constexpr (ADD) r.a[i] = *(const Raw*)(add_in + row * C + chc[i] * N);
    }
  };
  auto unpack = [](const Raw& v, float* o) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[2 * i] = half_lo(v[i]);
        o[2 * i + 1] = half_hi(v[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = v[i];
    }
  };
  auto do_row = [&](long long row, const RowIn& r) {
    float g[NCH][N], xh[NCH][N];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      float d[N], xv[N];
      unpack(r.d[i], d);
      unpack(r.xv[i], xv);
      const float on = live[i] ? 1.f : 0.f;  // a lane past the end of the row adds nothing to the row sums
#pragma unroll
      for (int e = 0; e < N; ++e) {
        const float xhat = (xv[e] - r.mu) * r.rs;
        const float gg = d[e] * gm[i][e];
        xh[i][e] = xhat;
        g[i][e] = gg;
        s1 = fmaf(on, gg, s1);
        s2 = fmaf(on * gg, xhat, s2);
        dg[i][e] = fmaf(d[e], xhat, dg[i][e]);
        db[i][e] += d[e];
      }
    }
    const float c1 = wave_sum(s1) / (float)C, c2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      float o[N];
      if constexpr (ADD) unpack(r.a[i], o);
      else {
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] += r.rs * (g[i][e] - c1 - xh[i][e] * c2);
      if (live[i]) V16<T>::st(dx + row * C + chc[i] * N, o);
      if (dxm) {
        // second output: dx under the dropout mask of the branch that consumes it next (melgpt_dropout_apply's
        // result on the STORED dx, bit for bit: rounded to T first, then keep * scale) - saves that pass's read
        if constexpr (sizeof(T) == 2) {
#pragma unroll
          for (int e = 0; e < N; e += 2) {
            const unsigned pk = pack_bf16x2(o[e], o[e + 1]);
            o[e] = half_lo(pk);
            o[e + 1] = half_hi(pk);
          }
        }
        keep_mask(mseed, msid, (unsigned long long)(row * C + chc[i] * N), N, mthresh, mscale, o);
        if (live[i]) V16<T>::st(dxm + row * C + chc[i] * N, o);
      }
    }
  };
  RowIn ra, rb;
  long long row = wave;
  load_row(row, ra);
  for (; row < M; row += 2LL * nwaves) {  // two rows per trip: the register sets swap roles without copies
    const long long r1 = row + nwaves, r2 = r1 + nwaves;
    load_row(r1, rb);
    do_row(row, ra);
    load_row(r2, ra);
    if (r1 < M) do_row(r1, rb);
  }
  if (partials) {
    // the block's waves combine their column sums through LDS (fixed order) -> ONE partial row per block
    extern __shared__ float lnsh[];  // [WPB][2C]
    float* mine = lnsh + (threadIdx.x >> 6) * 2 * C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          mine[ch * N + e] = dg[i][e];
          mine[C + ch * N + e] = db[i][e];
        }
      }
    }
    __syncthreads();
    float* out = partials + (long long)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < 2 * C; c += 64 * WPB) {
      float v = (lnsh[c] + lnsh[2 * C + c]) + (lnsh[4 * C + c] + lnsh[6 * C + c]);
      if constexpr (WPB == 8) v += (lnsh[8 * C + c] + lnsh[10 * C + c]) + (lnsh[12 * C + c] + lnsh[14 * C + c]);
      out[c] = v;
    }
  }
}

// out[c] (+)= scale * sum_{r<R} part[r*ld + c]  - fixed order => deterministic.
// block = 64 columns x 4 row phases (each phase sums rows r = phase, phase+4, ... with 4 independent chains)
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, int R, long long ld,
                                                          long long ncols, float* __restrict__ out, int accumulate,
                                                          float scale) {
  __shared__ float sh[4][64];
  const int lc = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const long long c = (long long)blockIdx.x * 64 + lc;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < ncols) {
    int r = ph;
    for (; r + 12 < R; r += 16) {
      s0 += part[(long long)r * ld + c];
      s1 += part[(long long)(r + 4) * ld + c];
      s2 += part[(long long)(r + 8) * ld + c];
      s3 += part[(long long)(r + 12) * ld + c];
    }
    for (; r < R; r += 4) s0 += part[(long long)r * ld + c];
  }
  sh[ph][lc] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ph == 0 && c < ncols) {
    const float s = ((sh[0][lc] + sh[1][lc]) + (sh[2][lc] + sh[3][lc])) * scale;
    out[c] = accumulate ? out[c] + s : s;
  }
}

// The same reduction with 16-byte loads and 16 row phases per block (64 columns x 16 phases x 4 chains = 64 rows in
// flight per column group): the partial tables of the LayerNorm / bias / split-K reductions have up to a thousand
// rows, which a block of 4 phases walks latency-bound.  Optional second destination: columns >= split go to
// out2[c - split] (LayerNorm's [dgamma | dbeta] partial rows are reduced by one launch).
// Fixed summation order (phase-major, then the four chains, then the phases pairwise) => deterministic.
__device__ __forceinline__ void reduce_rows4_body(int blk, const float* __restrict__ part, int R, long long ld,
                                                  long long ncols, float* __restrict__ out,
                                                  float* __restrict__ out2, long long split, int accumulate,
                                                  float scale) {
  __shared__ f32x4 sh[16][16];
  const int lc = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const long long c = ((long long)blk * 16 + lc) * 4;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (c < ncols) {
    int r = ph;
    for (; r + 48 < R; r += 64) {
      s0 += *(const f32x4*)(part + (long long)r * ld + c);
      s1 += *(const f32x4*)(part + (long long)(r + 16) * ld + c);
      s2 += *(const f32x4*)(part + (long long)(r + 32) * ld + c);
      s3 += *(const f32x4*)(part + (long long)(r + 48) * ld + c);
    }
    for (; r < R; r += 16) s0 += *(const f32x4*)(part + (long long)r * ld + c);
  }
  sh[ph][lc] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ph == 0 && c < ncols) {
    f32x4 t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = sh[2 * i][lc] + sh[2 * i + 1][lc];
    f32x4 v = (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) * scale;
    float* dst = (out2 && c >= split) ? out2 + (c - split) : out + c;
    if (accumulate) v += *(const f32x4*)dst;
    *(f32x4*)dst = v;
  }
}

__global__ __launch_bounds__(256) void reduce_rows4_kernel(const float* __restrict__ part, int R, long long ld,
                                                           long long ncols, float* __restrict__ out,
                                                           float* __restrict__ out2, long long split, int accumulate,
                                                           float scale) {
  reduce_rows4_body((int)blockIdx.x, part, R, ld, ncols, out, out2, split, accumulate, scale);
}

// The same sum for a FEW partial rows (R <= RB <= 8: split-K slabs of a weight gradient, N K columns each): one thread
// per 16-byte column chunk requests all its rows at once (unconditional loads, row index clamped - a branch around a load
// would be waited for at its end) and adds them in the order reduce_rows4_kernel does (phase sums, then the pair tree),
// so both kernels give the same bits.  reduce_rows4_kernel spends a 256-thread workgroup on 64 columns and keeps
// R of its 16 row phases busy: 3 TB/s on 4 x 16 MB slabs.
template <int RB>  // row phases in use: 2, 4 or 8 (R <= RB)
__device__ __forceinline__ void reduce_few_rows_body(int blk, const float* __restrict__ part, int R, long long ld,
                                                     long long ncols, float* __restrict__ out,
                                                     float* __restrict__ out2, long long split, int accumulate,
                                                     float scale) {
  const long long c = ((long long)blk * 256 + threadIdx.x) * 4;
  if (c >= ncols) return;
  float* dst = (out2 && c >= split) ? out2 + (c - split) : out + c;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 prev = *(const f32x4*)dst;  // (only used when accumulating; dst is always readable)
  f32x4 v[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) v[r] = *(const f32x4*)(part + (long long)min(r, R - 1) * ld + c);
#pragma unroll
  for (int r = 0; r < RB; ++r) v[r] = r < R ? v[r] : zero;
  f32x4 t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = (2 * i + 1 < RB) ? v[(2 * i) % RB] + v[(2 * i + 1) % RB] : ((2 * i < RB) ? v[(2 * i) % RB] + zero : zero + zero);
  f32x4 sum = (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) * scale;
  if (accumulate) sum += prev;
  *(f32x4*)dst = sum;
}

template <int RB>
__global__ __launch_bounds__(256) void reduce_few_rows_kernel(const float* __restrict__ part, int R, long long ld,
                                                              long long ncols, float* __restrict__ out,
                                                              float* __restrict__ out2, long long split, int accumulate,
                                                              float scale) {
  reduce_few_rows_body<RB>((int)blockIdx.x, part, R, ld, ncols, out, out2, split, accumulate, scale);
}

// TWO sums in one launch (a weight gradient's split-K slabs and the bias gradient's row-sum partials behind the same
// GEMM: 96 launches of ~4 us per training step otherwise): workgroups [0, g1) run the first job - the few-rows form when
// RB > 0, else the 16-phase form -, the rest the second job in the 16-phase form.  The bodies are the stand-alone
// kernels': same bits.
struct ReduceJob {
  const float* part;
  int R;
  long long ld, ncols;
  float* out;
  int accumulate;
};
template <int RB>
__global__ __launch_bounds__(256) void reduce_pair_kernel(ReduceJob a, int g1, ReduceJob b) {
  if ((int)blockIdx.x < g1) {
    if constexpr (RB > 0) reduce_few_rows_body<RB>((int)blockIdx.x, a.part, a.R, a.ld, a.ncols, a.out, nullptr, 0, a.accumulate, 1.0f);
    else reduce_rows4_body((int)blockIdx.x, a.part, a.R, a.ld, a.ncols, a.out, nullptr, 0, a.accumulate, 1.0f);
  } else {
    reduce_rows4_body((int)blockIdx.x - g1, b.part, b.R, b.ld, b.ncols, b.out, nullptr, 0, b.accumulate, 1.0f);
  }
}

// picks the vectorised form when columns, leading dimension, split and pointers allow 16-byte accesses
static void launch_reduce_rows(const float* part, int R, long long ld, long long ncols, float* out, float* out2,
                               long long split, int accumulate, float scale, hipStream_t s) {
  const bool vec = ncols % 4 == 0 && ld % 4 == 0 && split % 4 == 0 &&
                   ((((uintptr_t)part | (uintptr_t)out | (uintptr_t)out2) & 15) == 0);
  if (vec && R <= 8 && ncols >= (1 << 16)) {  // (16 rows keep every phase of reduce_rows4_kernel busy: 11 vs 14.5 us at 1 M columns)
    const dim3 grid((unsigned)((ncols / 4 + 255) / 256));
#define MELGPT_FEW(RB) \
  hipLaunchKernelGGL(reduce_few_rows_kernel<RB>, grid, dim3(256), 0, s, part, R, ld, ncols, out, out2, split, accumulate, scale)
    if (R <= 2) MELGPT_FEW(2);
    else if (R <= 4) MELGPT_FEW(4);
    else MELGPT_FEW(8);
#undef MELGPT_FEW
    return;
  }
  if (vec) {
    hipLaunchKernelGGL(reduce_rows4_kernel, dim3((unsigned)((ncols / 4 + 15) / 16)), dim3(256), 0, s, part, R, ld, ncols,
                       out, out2, split, accumulate, scale);
    return;
  }
  if (out2) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((split + 63) / 64)), dim3(256), 0, s, part, R, ld, split, out,
                       accumulate, scale);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((ncols - split + 63) / 64)), dim3(256), 0, s, part + split, R,
                       ld, ncols - split, out2, accumulate, scale);
    return;
  }
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((ncols + 63) / 64)), dim3(256), 0, s, part, R, ld, ncols, out,
                     accumulate, scale);
}

// column sums of a (M,N) matrix in dtype T into partials[(gridDim.y)][N]; second stage = reduce_rows_kernel
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ a, long long M, int N, long long lda,
                                                             float* __restrict__ partials) {
  constexpr int NV = V16<T>::N;
  const int ch = blockIdx.x * 64 + (threadIdx.x & 63);  // chunk column
  const int rsub = threadIdx.x >> 6;                    // 4 row phases per block
  const int nch = N / NV;
  float acc[NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) acc[e] = 0.f;
  if (ch < nch) {
    // four rows per trip with independent accumulators: four 16-byte loads in flight per lane (the sum order is
    // fixed by the grid, so the result is reproducible)
    const long long step = (long long)gridDim.y * 4;
    long long r = (long long)blockIdx.y * 4 + rsub;
    float a1[NV], a2[NV], a3[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) a1[e] = a2[e] = a3[e] = 0.f;
    // eight rows per trip while they last (same accumulator order as two trips of four: the sums are unchanged, but
    // eight loads are in flight per lane - at N = 1024 only 512 workgroups cover the matrix)
    for (; r + 7 * step < M; r += 8 * step) {
      float v[8][NV];
#pragma unroll
      for (int j = 0; j < 8; ++j) V16<T>::ld(a + (r + j * step) * lda + ch * NV, v[j]);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < NV; ++e) {
          acc[e] += v[4 * h][e];
          a1[e] += v[4 * h + 1][e];
          a2[e] += v[4 * h + 2][e];
          a3[e] += v[4 * h + 3][e];
        }
    }
    for (; r + 3 * step < M; r += 4 * step) {
      float v0[NV], v1[NV], v2[NV], v3[NV];
      V16<T>::ld(a + r * lda + ch * NV, v0);
      V16<T>::ld(a + (r + step) * lda + ch * NV, v1);
      V16<T>::ld(a + (r + 2 * step) * lda + ch * NV, v2);
      V16<T>::ld(a + (r + 3 * step) * lda + ch * NV, v3);
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        acc[e] += v0[e];
        a1[e] += v1[e];
        a2[e] += v2[e];
        a3[e] += v3[e];
      }
    }
    for (; r < M; r += step) {
      float v[NV];
      V16<T>::ld(a + r * lda + ch * NV, v);
#pragma unroll
      for (int e = 0; e < NV; ++e) acc[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) acc[e] = (acc[e] + a1[e]) + (a2[e] + a3[e]);
  }
  __shared__ float sh[4][64][NV + 1];
#pragma unroll
  for (int e = 0; e < NV; ++e) sh[rsub][threadIdx.x & 63][e] = acc[e];
  __syncthreads();
  if (rsub == 0 && ch < nch) {
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int e = 0; e < NV; ++e)
      partials[(long long)blockIdx.y * N + ch * NV + e] = (sh[0][l][e] + sh[1][l][e]) + (sh[2][l][e] + sh[3][l][e]);
  }
}

// y = keep(x) * scale (the dropout mask of a forward site replayed on its upstream gradient) AND the column sums of y in
// one pass: the backward of `drop(linear(...))` needs both d = drop'(dy) and the bias gradient sum_m d[m, :], and the
// separate column-sum pass re-read the tensor this kernel has just written.  Same grid, row order and accumulators as
// colsum_partial_kernel on the ROUNDED values, so the partials - and the bias gradients - are the same bits.
template <typename T>
__global__ __launch_bounds__(256) void dropout_colsum_kernel(const T* __restrict__ x, T* __restrict__ y, long long M, int N,
                                                             float scale, unsigned thresh, unsigned long long seed,
                                                             unsigned sid, float* __restrict__ partials) {
  constexpr int NV = V16<T>::N;
  const int ch = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rsub = threadIdx.x >> 6;
  const int nch = N / NV;
  float acc[4][NV];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < NV; ++e) acc[j][e] = 0.f;
  if (ch < nch) {
    const long long step = (long long)gridDim.y * 4;
    long long r = (long long)blockIdx.y * 4 + rsub;
    auto finish = [&](long long row, float* v, float* a) {
      keep_mask(seed, sid, (unsigned long long)row * N + ch * NV, NV, thresh, scale, v);
      V16<T>::st(y + row * N + ch * NV, v);
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        if constexpr (sizeof(T) == 2) a[e] += bf16_to_f32(f32_to_bf16(v[e]));
        else a[e] += v[e];
      }
    };
    for (; r + 3 * step < M; r += 4 * step) {
      float v[4][NV];
#pragma unroll
      for (int j = 0; j < 4; ++j) V16<T>::ld(x + (r + j * step) * N + ch * NV, v[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) finish(r + j * step, v[j], acc[j]);
    }
    for (; r < M; r += step) {
      float v[NV];
      V16<T>::ld(x + r * N + ch * NV, v);
      finish(r, v, acc[0]);
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) acc[0][e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
  }
  __shared__ float sh[4][64][NV + 1];
#pragma unroll
  for (int e = 0; e < NV; ++e) sh[rsub][threadIdx.x & 63][e] = acc[0][e];
  __syncthreads();
  if (rsub == 0 && ch < nch) {
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int e = 0; e < NV; ++e)
      partials[(long long)blockIdx.y * N + ch * NV + e] = (sh[0][l][e] + sh[1][l][e]) + (sh[2][l][e] + sh[3][l][e]);
  }
}

// ================================================================================== embedding stem
// out[b,t,:] = drop( (t < n_pre ? PRE(b,t) : tok_emb[idx[b,t-n_pre]]) + pos_emb[t] )
//   PRE = pre_table[pre_idx[b*n_pre+t]]  (GPTClass.embedder, minGPT.py:207-212)  or  pre_vals[b,t,:] (f32)
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ idx, const float* __restrict__ tok,
                                                        const float* __restrict__ pos,
                                                        const long long* __restrict__ pre_idx,
                                                        const float* __restrict__ pre_table,
                                                        const float* __restrict__ pre_vals, int n_pre, int B, int Tt,
                                                        long long idx_ld, int C, int V, T* __restrict__ out,
                                                        float drop_scale,
                                                        unsigned thresh, unsigned long long seed, unsigned sid) {
  constexpr int N = V16<T>::N;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int Ttot = Tt + n_pre;
  if (row >= (long long)B * Ttot) return;
  const int b = (int)(row / Ttot), tt = (int)(row % Ttot);
  const float* src;
  if (tt < n_pre) {
    src = pre_idx ? pre_table + pre_idx[(long long)b * n_pre + tt] * C : pre_vals + ((long long)b * n_pre + tt) * C;
  } else {
    long long k = idx[(long long)b * idx_ld + (tt - n_pre)];
    k = k < 0 ? 0 : (k >= V ? V - 1 : k);
    src = tok + k * C;
  }
  const float* pr = pos + (long long)tt * C;
  for (int ch = lane; ch < C / N; ch += 64) {
    float o[N];
#pragma unroll
    for (int e4 = 0; e4 < N; e4 += 4) {
      f32x4 a = *(const f32x4*)(src + ch * N + e4), p4 = *(const f32x4*)(pr + ch * N + e4);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e4 + e] = a[e] + p4[e];
    }
    if (drop_scale != 0.f) keep_mask(seed, sid, (unsigned long long)row * C + ch * N, N, thresh, drop_scale, o);
    V16<T>::st(out + row * C + ch * N, o);
  }
}

// table_grad[v,:] (+)= sum over items j with item_idx[j]==v of keep(dX[row(j),:]),  row(j) = (j/ipb)*Ttot + off + j%ipb
// one workgroup per table row; positions are visited in ascending order -> deterministic, no atomics.
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_table_kernel(const T* __restrict__ dx, const long long* __restrict__ item_idx,
                                                              long long n_items, int ipb, long long idx_ld, int Ttot,
                                                              int off, int C,
                                                              float* __restrict__ grad, int accumulate, float drop_scale,
                                                              unsigned thresh, unsigned long long seed, unsigned sid) {
  __shared__ int list[256];
  __shared__ int count;
  const int v = blockIdx.x, t = threadIdx.x;
  // thread owns 4 consecutive columns c = 4*(t + 256*i)
  constexpr int MAXC4 = 8;  // C <= 8192
  f32x4 acc[MAXC4];
#pragma unroll
  for (int i = 0; i < MAXC4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (long long base = 0; base < n_items; base += 256) {
    if (t == 0) count = 0;
    __syncthreads();
    const long long j = base + t;
    const bool hit = j < n_items && item_idx[(j / ipb) * idx_ld + (j % ipb)] == v;
    // ordered compaction: ballot per wave, waves in order
    unsigned long long bal = __ballot(hit);
    __shared__ int wcount[4];
    if ((t & 63) == 0) wcount[t >> 6] = __popcll(bal);
    __syncthreads();
    int wbase = 0;
    for (int ww = 0; ww < (t >> 6); ++ww) wbase += wcount[ww];
    if (hit) list[wbase + __popcll(bal & ((1ull << (t & 63)) - 1ull))] = (int)(j - base);
    if (t == 0) count = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    __syncthreads();
    const int cnt = count;
    for (int s = 0; s < cnt; ++s) {
      const long long jj = base + list[s];
      const long long row = (jj / ipb) * Ttot + off + (jj % ipb);
#pragma unroll
      for (int i = 0; i < MAXC4; ++i) {
        const int c = 4 * (t + 256 * i);
        if (c < C) {
          float d[4];
          if constexpr (sizeof(T) == 4) {
            f32x4 q = *(const f32x4*)((const float*)dx + row * C + c);
            d[0] = q[0]; d[1] = q[1]; d[2] = q[2]; d[3] = q[3];
          } else {
            u32x2 q = *(const u32x2*)((const bf16_t*)dx + row * C + c);
            d[0] = half_lo(q[0]); d[1] = half_hi(q[0]);
            d[2] = half_lo(q[1]); d[3] = half_hi(q[1]);
          }
          if (drop_scale != 0.f) keep_mask(seed, sid, (unsigned long long)row * C + c, 4, thresh, drop_scale, d);
          acc[i] += f32x4{d[0], d[1], d[2], d[3]};
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < MAXC4; ++i) {
    const int c = 4 * (t + 256 * i);
    if (c < C) {
      f32x4* g = (f32x4*)(grad + (long long)v * C + c);
      *g = accumulate ? *g + acc[i] : acc[i];
    }
  }
}

// pos_grad[t,:] (+)= sum_b keep(dX[b,t,:])   and   (optional) pre_vals_grad[b,t,:] = keep(dX[b,t,:]) for t < n_pre
// One wave per (position, group of 64 chunks); the batch is walked in order (fixed summation order) with EIGHT rows'
// loads requested before the first is used (one row at a time was a chain of B dependent round trips on 67 workgroups:
// 135 us for the 69 MB of the training step's dX).
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_pos_kernel(const T* __restrict__ dx, int B, int Ttot, int C,
                                                            float* __restrict__ pos_grad, int accumulate,
                                                            float* __restrict__ pre_grad, int n_pre, float drop_scale,
                                                            unsigned thresh, unsigned long long seed, unsigned sid) {
  constexpr int N = V16<T>::N, UB = 8;
  const int lane = threadIdx.x & 63;
  const int ngrp = (C / N + 63) / 64;
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tt = wv / ngrp, ch = (wv - tt * ngrp) * 64 + lane;
  if (tt >= Ttot || ch >= C / N) return;
  float acc[N];
#pragma unroll
  for (int e = 0; e < N; ++e) acc[e] = 0.f;
  auto take = [&](int b, float (&d)[N]) {
    const long long row = (long long)b * Ttot + tt;
    if (drop_scale != 0.f) keep_mask(seed, sid, (unsigned long long)row * C + ch * N, N, thresh, drop_scale, d);
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] += d[e];
    if (pre_grad && tt < n_pre) {
#pragma unroll
      for (int e = 0; e < N; ++e) pre_grad[((long long)b * n_pre + tt) * C + ch * N + e] = d[e];
    }
  };
  int b = 0;
  for (; b + UB <= B; b += UB) {
    float d[UB][N];
#pragma unroll
    for (int u = 0; u < UB; ++u) V16<T>::ld(dx + ((long long)(b + u) * Ttot + tt) * C + ch * N, d[u]);
#pragma unroll
    for (int u = 0; u < UB; ++u) take(b + u, d[u]);
  }
  for (; b < B; ++b) {
    float d[N];
    V16<T>::ld(dx + ((long long)b * Ttot + tt) * C + ch * N, d);
    take(b, d);
  }
  float* g = pos_grad + (long long)tt * C + ch * N;
#pragma unroll
  for (int e = 0; e < N; ++e) g[e] = accumulate ? g[e] + acc[e] : acc[e];
}

// ================================================================================== cross entropy
// logits (M,V) f32 with row stride ld.  loss_rows[m] = lse - logit[target]; lse saved for the backward.
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, long long ld,
                                                     const long long* __restrict__ target, long long M, int V,
                                                     float* __restrict__ loss_rows, float* __restrict__ lse) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* lr = logits + row * ld;
  float mx = -__builtin_inff();
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, lr[c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < V; c += 64) s += __expf(lr[c] - mx);
  s = wave_sum(s);
  const float l = mx + __logf(s);
  if (lane == 0) {
    long long tg = target[row];
    tg = tg < 0 ? 0 : (tg >= V ? V - 1 : tg);
    loss_rows[row] = l - lr[tg];
    lse[row] = l;
  }
}

// dlogits[m,v] = (exp(logit - lse) - [v == target]) * g_rows[m] * g_scale
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long long ld,
                                                     const long long* __restrict__ target,
                                                     const float* __restrict__ lse, const float* __restrict__ g_rows,
                                                     int g_group, const float* __restrict__ g_scalar, float g_scale,
                                                     long long M, int V, T* __restrict__ dlogits, long long ldd) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* lr = logits + row * ld;
  const float l = lse[row];
  const float g = (g_rows ? g_rows[row / g_group] : 1.f) * (g_scalar ? *g_scalar : 1.f) * g_scale;
  const long long tg = target[row];
  for (int c = lane; c < V; c += 64) {
    float p = __expf(lr[c] - l) - (c == tg ? 1.f : 0.f);
    Elem<T>::st(dlogits + row * ldd + c, p * g);
  }
}

// out[0] (+)= scale * sum(in[0..n))   single workgroup, fixed order
__global__ void sum_kernel(const float* __restrict__ in, long long n, float scale, float* __restrict__ out,
                           int accumulate) {
  __shared__ float sh[256];
  float s = 0.f;
  long long i = threadIdx.x;
  for (; i + 7 * 256 < n; i += 8 * 256) {  // eight loads in flight, added in the order of the plain loop (same bits)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = in[i + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < n; i += 256) s += in[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + sh[0] * scale : sh[0] * scale;
}

// ================================================================================== elementwise
// y = keep(x) * scale  (dropout replay for a backward pass), element index = linear index in a contiguous tensor
template <typename T>
__global__ void dropout_apply_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, float scale,
                                     unsigned thresh, unsigned long long seed, unsigned sid) {
  constexpr int N = V16<T>::N;
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * N;
  const long long step = (long long)gridDim.x * blockDim.x * N;
  for (; i < n; i += step) {
    float v[N];
    V16<T>::ld(x + i, v);
    keep_mask(seed, sid, (unsigned long long)i, N, thresh, scale, v);
    V16<T>::st(y + i, v);
  }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long long n) {
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const long long step = (long long)gridDim.x * blockDim.x * 8;
  for (; i + 7 < n; i += step) {
    f32x4 a = *(const f32x4*)(x + i), b = *(const f32x4*)(x + i + 4);
    *(u32x4*)(y + i) =
        u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
  }
  if (i < n && i + 7 >= n)
    for (long long j = i; j < n; ++j) y[j] = f32_to_bf16(x[j]);
}

template <typename TI, typename TO>
__global__ void cast_strided_kernel(const TI* __restrict__ x, TO* __restrict__ y, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long long)gridDim.x * blockDim.x) Elem<TO>::st(y + i, Elem<TI>::ld(x + i));
}

// torch.optim.AdamW semantics (decoupled decay first, bias-corrected step), fused over a flat f32 buffer
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, bf16_t* __restrict__ p_bf16, long long n, float lr, float beta1,
                             float beta2, float eps, float wd, float bc1, float bc2_sqrt, float grad_scale) {
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long long step = (long long)gridDim.x * blockDim.x * 4;
  const float step_size = lr / bc1;
  auto update4 = [&](f32x4& pp, f32x4 gg, f32x4& mm, f32x4& vv) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pp[e] *= (1.f - lr * wd);
      mm[e] = beta1 * mm[e] + (1.f - beta1) * gg[e];
      vv[e] = beta2 * vv[e] + (1.f - beta2) * gg[e] * gg[e];
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pp[e] -= step_size * (mm[e] / denom);
    }
  };
  // two sweeps' chunks per trip: eight 16-byte loads in flight per lane instead of four (30 bytes move per element:
  // the kernel is a pure stream)
  for (; i + step + 3 < n; i += 2 * step) {
    const long long j = i + step;
    f32x4 p0 = *(f32x4*)(p + i), g0 = *(const f32x4*)(g + i) * grad_scale, m0 = *(f32x4*)(m + i), v0 = *(f32x4*)(v + i);
    f32x4 p1 = *(f32x4*)(p + j), g1 = *(const f32x4*)(g + j) * grad_scale, m1 = *(f32x4*)(m + j), v1 = *(f32x4*)(v + j);
    update4(p0, g0, m0, v0);
    update4(p1, g1, m1, v1);
    *(f32x4*)(p + i) = p0; *(f32x4*)(m + i) = m0; *(f32x4*)(v + i) = v0;
    *(f32x4*)(p + j) = p1; *(f32x4*)(m + j) = m1; *(f32x4*)(v + j) = v1;
    if (p_bf16) {
      *(u32x2*)(p_bf16 + i) = u32x2{pack_bf16x2(p0[0], p0[1]), pack_bf16x2(p0[2], p0[3])};
      *(u32x2*)(p_bf16 + j) = u32x2{pack_bf16x2(p1[0], p1[1]), pack_bf16x2(p1[2], p1[3])};
    }
  }
  for (; i < n; i += step) {
    if (i + 3 < n) {
      f32x4 pp = *(f32x4*)(p + i), gg = *(const f32x4*)(g + i) * grad_scale, mm = *(f32x4*)(m + i), vv = *(f32x4*)(v + i);
      update4(pp, gg, mm, vv);
      *(f32x4*)(p + i) = pp; *(f32x4*)(m + i) = mm; *(f32x4*)(v + i) = vv;
      if (p_bf16) *(u32x2*)(p_bf16 + i) = u32x2{pack_bf16x2(pp[0], pp[1]), pack_bf16x2(pp[2], pp[3])};
    } else {
      for (long long j = i; j < n; ++j) {
        float pp = p[j] * (1.f - lr * wd), gg = g[j] * grad_scale;
        float mm = beta1 * m[j] + (1.f - beta1) * gg, vv = beta2 * v[j] + (1.f - beta2) * gg * gg;
        pp -= step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
        p[j] = pp; m[j] = mm; v[j] = vv;
        if (p_bf16) p_bf16[j] = f32_to_bf16(pp);
      }
    }
  }
}

// (B,H,W) row-major codes -> (B, W*H) time-major sequence p = w*H + h  (Lit_minGPT.get_x, minGPT.py:387-394;
// make_idx/code_reader :431-456) and its inverse
__global__ void codes_permute_kernel(const long long* __restrict__ in, long long* __restrict__ out, int B, int H, int W,
                                     int reverse) {
  const long long total = (long long)B * H * W;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / (H * W);
    const int r = (int)(i % (H * W));
    if (!reverse) {  // out[b, w*H + h] = in[b, h*W + w]
      const int w = r / H, h = r % H;
      out[i] = in[b * H * W + h * W + w];
    } else {         // out[b, h*W + w] = in[b, w*H + h]
      const int h = r / W, w = r % W;
      out[i] = in[b * H * W + w * H + h];
    }
  }
}

// one-hot rows of the embedding-gradient GEMM: row (b,t) = e_{idx[b,t-n_pre]} (zero row for the n_pre prepended
// positions), so that  d tok_emb = OneHot^T (V x M) @ dX (M x C)  runs on the MFMA GEMM, split over K as a batch
template <typename T>
__global__ void onehot_rows_kernel(const long long* __restrict__ idx, long long idx_ld, int B, int Tt, int n_pre, int V,
                                   T* __restrict__ out) {
  const long long total = (long long)B * (Tt + n_pre) * V;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % V);
    const long long row = i / V;
    const int tt = (int)(row % (Tt + n_pre));
    const long long b = row / (Tt + n_pre);
    const bool one = tt >= n_pre && idx[b * idx_ld + (tt - n_pre)] == v;
    Elem<T>::st(out + i, one ? 1.f : 0.f);
  }
}

// out[r] = scale * sum_{j<n} in[r*n + j]   (one wave per group; per-sequence sums of the per-token losses)
__global__ __launch_bounds__(256) void group_sum_kernel(const float* __restrict__ in, long long R, int n, float scale,
                                                        float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float s = 0.f;
  for (int j = lane; j < n; j += 64) s += in[r * n + j];
  s = wave_sum(s);
  if (lane == 0) out[r] = s * scale;
}

// one decoding step of Lit_minGPT.sample / GPTDecoder.sample (minGPT.py:345-358, decoders.py:108-121):
// logits/temperature -> optional top-k filter (values below the k-th largest -> -inf, ties kept) -> softmax ->
// argmax (sample = 0) or one multinomial draw by inverse CDF with a Philox uniform.  One workgroup per row, V <= 1024.
__global__ __launch_bounds__(256) void sample_logits_kernel(const float* __restrict__ logits, long long ld, int V,
                                                            float temperature, int top_k, int do_sample,
                                                            unsigned long long seed, unsigned step,
                                                            long long* __restrict__ out, float* __restrict__ probs_out,
                                                            const int* __restrict__ pos_dev, int step_offset,
                                                            long long* __restrict__ seq, long long seq_ld) {
  // pos_dev (graph-replayed decoding): the step number and the slot of `seq` to fill are read on the device, so the
  // same captured launch serves every position
  if (pos_dev) step = (unsigned)(*pos_dev + step_offset);
  __shared__ float v[1024], srt[1024], red[256];
  __shared__ int redi[256];
  const int t = threadIdx.x, row = blockIdx.x;
  const float NEG = -__builtin_inff();
  int P = 64;  // sort width: the power of two that covers the vocabulary (128 for the VAS codebook, 1024 for VGGSound)
  while (P < V) P <<= 1;
  for (int i = t; i < 1024; i += 256) {
    float x = i < V ? logits[(long long)row * ld + i] / temperature : NEG;
    v[i] = x;
    srt[i] = x;
  }
  __syncthreads();
  if (top_k > 0 && top_k < V) {
    for (int k = 2; k <= P; k <<= 1)             // bitonic sort, descending
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = t; i < P; i += 256) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const bool up = (i & k) == 0;
            const float a = srt[i], b = srt[ixj];
            if (up ? (a < b) : (a > b)) { srt[i] = b; srt[ixj] = a; }
          }
        }
        __syncthreads();
      }
    const float thr = srt[top_k - 1];
    for (int i = t; i < V; i += 256)
      if (v[i] < thr) v[i] = NEG;
    __syncthreads();
  }
  // max / argmax (lowest index on ties)
  float m = NEG;
  int mi = 0x7fffffff;
  for (int i = t; i < V; i += 256)
    if (v[i] > m) { m = v[i]; mi = i; }
  red[t] = m; redi[t] = mi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
      if (red[t + o] > red[t] || (red[t + o] == red[t] && redi[t + o] < redi[t])) { red[t] = red[t + o]; redi[t] = redi[t + o]; }
    }
    __syncthreads();
  }
  const float mx = red[0];
  const int amax = redi[0];
  __syncthreads();
  float l = 0.f;
  for (int i = t; i < V; i += 256) {
    const float e = __expf(v[i] - mx);
    v[i] = e;
    l += e;
  }
  red[t] = l;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  const float total = red[0];
  if (probs_out)
    for (int i = t; i < V; i += 256) probs_out[(long long)row * V + i] = v[i] / total;
  if (t == 0) {
    int pick = amax;
    if (do_sample) {
      Philox4 r = philox4x32_10(seed, ((unsigned long long)step << 32) | (unsigned)row, 0x5A3Du);
      const float u = ((r.x >> 8) + 0.5f) * (1.0f / 16777216.0f) * total;
      float c = 0.f;
      pick = V - 1;
      for (int i = 0; i < V; ++i) {
        c += v[i];
        if (c >= u && v[i] > 0.f) { pick = i; break; }
      }
    }
    out[row] = pick;
    if (seq && pos_dev) seq[(long long)row * seq_ld + *pos_dev] = pick;
  }
}

// GPTEncoder.reparameterize + KL (encoders.py:62-104): stats (B, 2*nz) f32 = [mu | logvar] (the last position's
// logits); z[b,s,:] = mu + eps[b,s,:]*exp(logvar/2); KL[b] = 0.5*sum(mu^2 + exp(logvar) - logvar - 1).
// eps is given (tests) or drawn in-kernel from Philox + Box-Muller (and written out for the backward).
__global__ __launch_bounds__(256) void vae_reparam_fwd_kernel(const float* __restrict__ stats, float* __restrict__ eps,
                                                              int gen_eps, unsigned long long seed, int B, int ns,
                                                              int nz, float* __restrict__ z, float* __restrict__ kl) {
  __shared__ float sh[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* mu = stats + (long long)b * 2 * nz;
  const float* lv = mu + nz;
  float acc = 0.f;
  for (int c = t; c < nz; c += 256) {
    const float m = mu[c], l = lv[c], sd = __expf(0.5f * l);
    acc += m * m + __expf(l) - l - 1.0f;
    for (int s = 0; s < ns; ++s) {
      const long long o = ((long long)b * ns + s) * nz + c;
      float e;
      if (gen_eps) {
        Philox4 r = philox4x32_10(seed, (unsigned long long)o, 0x7AE5u);
        const float u1 = ((r.x >> 8) + 1.0f) * (1.0f / 16777217.0f), u2 = (r.y >> 8) * (1.0f / 16777216.0f);
        e = sqrtf(-2.0f * __logf(u1)) * cospif(2.0f * u2);
        eps[o] = e;
      } else {
        e = eps[o];
      }
      z[o] = m + e * sd;
    }
  }
  sh[t] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) sh[t] += sh[t + o];
    __syncthreads();
  }
  if (t == 0) kl[b] = 0.5f * sh[0];
}

// d stats from dz (B,ns,nz) and dKL (B):  dmu = sum_s dz + dKL*mu ;  dlogvar = sum_s dz*eps*sd/2 + dKL*(exp(lv)-1)/2
__global__ void vae_reparam_bwd_kernel(const float* __restrict__ stats, const float* __restrict__ eps,
                                       const float* __restrict__ dz, const float* __restrict__ dkl, int B, int ns, int nz,
                                       float* __restrict__ dstats) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * nz) return;
  const int b = (int)(i / nz), c = (int)(i % nz);
  const float m = stats[(long long)b * 2 * nz + c], l = stats[(long long)b * 2 * nz + nz + c];
  const float sd = __expf(0.5f * l), g = dkl ? dkl[b] : 0.f;
  float dm = g * m, dl = g * 0.5f * (__expf(l) - 1.0f);
  if (dz)
    for (int s = 0; s < ns; ++s) {
      const long long o = ((long long)b * ns + s) * nz + c;
      dm += dz[o];
      dl += dz[o] * eps[o] * sd * 0.5f;
    }
  dstats[(long long)b * 2 * nz + c] = dm;
  dstats[(long long)b * 2 * nz + nz + c] = dl;
}

// GPTEncoder.eval_inference_dist and the density table of calc_mi (encoders.py:106-134, 154-163):
// log N(z; mu, exp(logvar)) = -0.5 sum_c (z - mu)^2 / exp(logvar) - 0.5 (nz log 2pi + sum_c logvar).  One WAVE per
// (z row, statistics row) pair.  pairwise == 0: z (X, S, nz), pair p = (x, s) uses row x's statistics -> out (X, S);
// pairwise != 0: z (S, nz), pair p = (i, x): z row i under row x's statistics -> out (S, X).
__device__ __forceinline__ float gauss_logq_wave(const float* __restrict__ zr, const float* __restrict__ mu,
                                                 const float* __restrict__ lv, int nz, int lane) {
  float q = 0.f, sl = 0.f;
  for (int c = lane; c < nz; c += 64) {
    const float l = lv[c], d = zr[c] - mu[c];
    q += d * d / expf(l);
    sl += l;
  }
  q = wave_sum(q);
  sl = wave_sum(sl);
  return -0.5f * q - 0.5f * ((float)nz * 1.8378770664093453f + sl);  // log(2 pi)
}

__global__ __launch_bounds__(256) void gauss_log_density_kernel(const float* __restrict__ z, const float* __restrict__ mu,
                                                                const float* __restrict__ logvar, long long ld_stats, int X,
                                                                int S, int nz, int pairwise, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long pair = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), npairs = (long long)X * S;
  if (pair >= npairs) return;
  const long long zrow = pairwise ? pair / X : pair;
  const int x = pairwise ? (int)(pair - zrow * X) : (int)(pair / S);
  const float v = gauss_logq_wave(z + zrow * nz, mu + x * ld_stats, logvar + x * ld_stats, nz, lane);
  if (lane == 0) out[pair] = v;
}

// GPTEncoder.calc_mi (encoders.py:136-170), one workgroup per row i of the batch: z_i = mu_i + eps_i exp(logvar_i / 2)
// (ONE draw per row, given or drawn as in vae_reparam_fwd_kernel), log q(z_i | x_j) for every j (a wave per j), the
// aggregate posterior log q(z_i) = logsumexp_j - log B (utils.log_sum_exp: max first), and the row's entropy term;
// term[i] = (-0.5 nz log 2pi - 0.5 sum_c (1 + logvar_i)) - log q(z_i).  MI = mean_i term[i] (melgpt_sum_f32).
__global__ __launch_bounds__(256) void vae_calc_mi_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                          long long ld_stats, float* __restrict__ eps, int gen_eps,
                                                          unsigned long long seed, int B, int nz,
                                                          float* __restrict__ zbuf, float* __restrict__ term) {
  __shared__ float sm[4], ss[4];
  const int i = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const float* mi = mu + i * ld_stats;
  const float* li = logvar + i * ld_stats;
  float* zi = zbuf + (long long)i * nz;
  for (int c = t; c < nz; c += 256) {
    const long long o = (long long)i * nz + c;
    float e;
    if (gen_eps) {
      Philox4 r = philox4x32_10(seed, (unsigned long long)o, 0x7AE5u);
      const float u1 = ((r.x >> 8) + 1.0f) * (1.0f / 16777217.0f), u2 = (r.y >> 8) * (1.0f / 16777216.0f);
      e = sqrtf(-2.0f * __logf(u1)) * cospif(2.0f * u2);
      eps[o] = e;
    } else {
      e = eps[o];
    }
    zi[c] = mi[c] + e * expf(0.5f * li[c]);
  }
  __syncthreads();  // (zi is read back by this workgroup only)
  float m = -INFINITY, sum = 0.f;  // running max / sum of exp over this wave's j (wave-uniform)
  for (int j = w; j < B; j += 4) {
    const float v = gauss_logq_wave(zi, mu + j * ld_stats, logvar + j * ld_stats, nz, lane);
    const float mn = fmaxf(m, v);
    sum = sum * expf(m - mn) + expf(v - mn);
    m = mn;
  }
  if (lane == 0) {
    sm[w] = m;
    ss[w] = sum;
  }
  float sl = 0.f;
  for (int c = t; c < nz; c += 256) sl += 1.0f + li[c];
  __shared__ float sh[256];
  sh[t] = sl;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) sh[t] += sh[t + o];
    __syncthreads();
  }
  if (t == 0) {
    float mm = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])), tot = 0.f;
    for (int k = 0; k < 4; ++k) tot += sm[k] == -INFINITY ? 0.f : ss[k] * expf(sm[k] - mm);
    const float log_qz = mm + logf(tot) - logf((float)B);
    term[i] = (-0.5f * (float)nz * 1.8378770664093453f - 0.5f * sh[0]) - log_qz;
  }
}

inline int grid_for(long long work_items, int per_block, int cap = 8192) {
  long long g = (work_items + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
inline unsigned thresh_of(float p) {
  double th = (double)p * 4294967296.0;
  return th >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)th;
}

}  // namespace

// LayerNorm kernels: element type x chunks-per-lane (C / (16 / sizeof(T)) chunks spread over 64 lanes)
#define DISPATCH_LN_NCH(C_, VEC_, ...)                          \
  {                                                             \
    const int per_lane_ = ((C_) / (VEC_) + 63) / 64;            \
    if (per_lane_ <= 1) {                                       \
      constexpr int NCH = 1;                                    \
      __VA_ARGS__;                                              \
    } else if (per_lane_ <= 2) {                                \
      constexpr int NCH = 2;                                    \
      __VA_ARGS__;                                              \
    } else if (per_lane_ <= 4) {                                \
      constexpr int NCH = 4;                                    \
      __VA_ARGS__;                                              \
    } else {                                                    \
      constexpr int NCH = LN_MAXCH;                             \
      __VA_ARGS__;                                              \
    }                                                           \
  }
#define DISPATCH_LN(dtype, C_, ...)                             \
  if ((dtype) == MELGPT_F32) {                                  \
    using T = float;                                            \
    DISPATCH_LN_NCH(C_, 4, __VA_ARGS__)                         \
  } else if ((dtype) == MELGPT_BF16) {                          \
    using T = bf16_t;                                           \
    DISPATCH_LN_NCH(C_, 8, __VA_ARGS__)                         \
  } else                                                        \
    return MELGPT_ERR_UNSUPPORTED;

#define DISPATCH_T(dtype, ...)                                  \
  if ((dtype) == MELGPT_F32) {                                  \
    using T = float;                                            \
    __VA_ARGS__;                                                \
  } else if ((dtype) == MELGPT_BF16) {                          \
    using T = bf16_t;                                           \
    __VA_ARGS__;                                                \
  } else                                                        \
    return MELGPT_ERR_UNSUPPORTED;

extern "C" int melgpt_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                    float* rstd, long long M, int C, float eps, int dtype, void* stream) {
  MELGPT_CHECK(x && gamma && beta && y && M > 0 && C > 0, MELGPT_ERR_BAD_ARG);
  const int vec = dtype == MELGPT_F32 ? 4 : 8;
  MELGPT_CHECK(C % vec == 0 && C / vec <= 64 * LN_MAXCH, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, MELGPT_ERR_ALIGN);
  DISPATCH_LN(dtype, C, hipLaunchKernelGGL((layernorm_fwd_kernel<T, NCH>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0,
                                       (hipStream_t)stream, (const T*)x, gamma, beta, (T*)y, mean, rstd, M, C, eps));
  return melgpt_launch_status();
}

extern "C" int melgpt_layernorm_bwd_nwaves(long long M) {
  long long w = (M + 7) / 8;  // >= 8 rows per wave where possible
  if (w < 4) w = 4;
  // ONE 8-wave workgroup per CU: at 33 920 x 1024 bf16 rows (+ residual gradient, + dgamma / dbeta) 1 024 waves 57.7 us,
  // 1 536 48.0, 2 048 47.9 (5.8 TB/s), 3 072 53.8, 4 096 51.4, 8 192 and up 60.0 (tools/lab/stream_ab.py): more resident
  // waves only lengthen the partial-row reduction and thrash the rows' lines
  if (w > 2048) w = 2048;
  return (int)((w + 3) / 4 * 4);
}

extern "C" int melgpt_layernorm_bwd_masked(const void* dy, const void* x, const float* gamma, const float* mean,
                                           const float* rstd, const void* add_in, void* dx, float* dgamma,
                                           float* dbeta, int accumulate, float* workspace, long long M, int C,
                                           void* dx_masked, float drop_p, unsigned long long seed, unsigned stream_id,
                                           int dtype, void* stream) {
  MELGPT_CHECK(dy && x && gamma && mean && rstd && dx && M > 0 && C > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(!dx_masked || (drop_p > 0.f && drop_p < 1.f && dx_masked != dx), MELGPT_ERR_BAD_ARG);
  const float mscale = dx_masked ? 1.f / (1.f - drop_p) : 0.f;
  const unsigned mthresh = dx_masked ? thresh_of(drop_p) : 0u;
  MELGPT_CHECK((dgamma == nullptr) == (dbeta == nullptr), MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(!dgamma || workspace, MELGPT_ERR_BAD_ARG);
  const int vec = dtype == MELGPT_F32 ? 4 : 8;
  MELGPT_CHECK(C % vec == 0 && C / vec <= 64 * LN_MAXCH, MELGPT_ERR_UNSUPPORTED);
  const int nwaves = melgpt_layernorm_bwd_nwaves(M);
  hipStream_t s = (hipStream_t)stream;
  // 8 waves per workgroup where their partial rows fit 64 KiB (C <= 1024); at the GPT-VAE XL width (1472: 94 KB) the
  // 8-wave form was measured at 189 us against the 4-wave form's 87 (tools/lab/stream_ab.py)
  const bool wide = nwaves % 8 == 0 && (size_t)8 * 2 * C * sizeof(float) <= 64 * 1024;
  const int wpb = wide ? 8 : 4;
  const size_t lds = dgamma ? (size_t)wpb * 2 * C * sizeof(float) : 0;
  MELGPT_CHECK(lds <= 64 * 1024, MELGPT_ERR_UNSUPPORTED);
#define LN_BWD_LAUNCH(ADD_, WPB_)                                                                                         \
  DISPATCH_LN(dtype, C, hipLaunchKernelGGL((layernorm_bwd_kernel<T, NCH, ADD_, WPB_>), dim3(nwaves / WPB_), dim3(64 * WPB_), \
                                           lds, s, (const T*)dy, (const T*)x, gamma, mean, rstd, (const T*)add_in, (T*)dx, \
                                           dgamma ? workspace : nullptr, M, C, (T*)dx_masked, mscale, mthresh, seed,      \
                                           stream_id))
  if (add_in) {
    if (wide) { LN_BWD_LAUNCH(true, 8); } else { LN_BWD_LAUNCH(true, 4); }
  } else {
    if (wide) { LN_BWD_LAUNCH(false, 8); } else { LN_BWD_LAUNCH(false, 4); }
  }
#undef LN_BWD_LAUNCH
  if (dgamma) {
    const int nblocks = nwaves / wpb;
    launch_reduce_rows(workspace, nblocks, 2LL * C, 2LL * C, dgamma, dbeta, (long long)C, accumulate, 1.0f, s);
  }
  return melgpt_launch_status();
}

extern "C" int melgpt_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                    const float* rstd, const void* add_in, void* dx, float* dgamma, float* dbeta,
                                    int accumulate, float* workspace, long long M, int C, int dtype, void* stream) {
  return melgpt_layernorm_bwd_masked(dy, x, gamma, mean, rstd, add_in, dx, dgamma, dbeta, accumulate, workspace, M, C,
                                     nullptr, 0.f, 0ull, 0u, dtype, stream);
}

extern "C" int melgpt_colsum_rows(void) { return 256; }

extern "C" int melgpt_colsum(const void* a, long long M, int N, long long lda, float* out, int accumulate,
                             float* workspace, int dtype, void* stream) {
  MELGPT_CHECK(a && out && workspace && M > 0 && N > 0, MELGPT_ERR_BAD_ARG);
  const int vec = dtype == MELGPT_F32 ? 4 : 8;
  MELGPT_CHECK(N % vec == 0 && lda % vec == 0 && ((uintptr_t)a & 15) == 0, MELGPT_ERR_ALIGN);
  long long rows = (M + 3) / 4;
  const int nch = N / vec;
  // enough row groups for ~2048 workgroups (<= melgpt_colsum_rows() partial rows)
  int gy = 2048 / ((nch + 63) / 64);
  gy = gy < 16 ? 16 : gy > 256 ? 256 : gy;
  if (rows < gy) gy = (int)rows;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_T(dtype, hipLaunchKernelGGL(colsum_partial_kernel<T>, dim3((nch + 63) / 64, gy), dim3(256), 0, s,
                                       (const T*)a, M, N, lda, workspace));
  launch_reduce_rows(workspace, gy, (long long)N, (long long)N, out, nullptr, 0, accumulate, 1.0f, s);
  return melgpt_launch_status();
}

extern "C" int melgpt_embed_fwd(const long long* idx, const float* tok_emb, const float* pos_emb,
                                const long long* pre_idx, const float* pre_table, const float* pre_vals, int n_pre,
                                int B, int Tt, long long idx_ld, int C, int V, void* out, int dtype, float drop_p,
                                unsigned long long seed, unsigned stream_id, void* stream) {
  MELGPT_CHECK(tok_emb && pos_emb && out && B > 0 && Tt >= 0 && C > 0 && V > 0 && n_pre >= 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(Tt == 0 || (idx && idx_ld >= Tt), MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(n_pre == 0 || (pre_idx && pre_table) || pre_vals, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(C % 8 == 0 && drop_p >= 0.f && drop_p < 1.f, MELGPT_ERR_UNSUPPORTED);
  const long long rows = (long long)B * (Tt + n_pre);
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 0.f;
  DISPATCH_T(dtype, hipLaunchKernelGGL(embed_fwd_kernel<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                                       (hipStream_t)stream, idx, tok_emb, pos_emb, pre_idx, pre_table, pre_vals, n_pre,
                                       B, Tt, idx_ld, C, V, (T*)out, sc, thresh_of(drop_p), seed, stream_id));
  return melgpt_launch_status();
}

extern "C" int melgpt_embed_bwd(const void* dx, const long long* idx, long long idx_ld, const long long* pre_idx,
                                int n_pre, int B, int Tt, int C, int V, int n_pre_rows, float* tok_grad, float* pos_grad,
                                float* pre_table_grad, float* pre_vals_grad, int accumulate, int dtype, float drop_p,
                                unsigned long long seed, unsigned stream_id, void* stream) {
  MELGPT_CHECK(dx && B > 0 && C > 0 && V > 0 && n_pre >= 0 && Tt >= 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(C % 8 == 0 && C <= 8192, MELGPT_ERR_UNSUPPORTED);
  const int Ttot = Tt + n_pre;
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 0.f;
  const unsigned th = thresh_of(drop_p);
  hipStream_t s = (hipStream_t)stream;
  if (tok_grad && Tt > 0) {
    MELGPT_CHECK(idx && idx_ld >= Tt, MELGPT_ERR_BAD_ARG);
    DISPATCH_T(dtype, hipLaunchKernelGGL(embed_bwd_table_kernel<T>, dim3(V), dim3(256), 0, s, (const T*)dx, idx,
                                         (long long)B * Tt, Tt, idx_ld, Ttot, n_pre, C, tok_grad, accumulate, sc, th, seed,
                                         stream_id));
  }
  if (pre_table_grad && n_pre > 0) {
    MELGPT_CHECK(pre_idx && n_pre_rows > 0, MELGPT_ERR_BAD_ARG);
    DISPATCH_T(dtype, hipLaunchKernelGGL(embed_bwd_table_kernel<T>, dim3(n_pre_rows), dim3(256), 0, s, (const T*)dx,
                                         pre_idx, (long long)B * n_pre, n_pre, (long long)n_pre, Ttot, 0, C,
                                         pre_table_grad, accumulate,
                                         sc, th, seed, stream_id));
  }
  if (pos_grad) {
    const int vec = dtype == MELGPT_F32 ? 4 : 8, ngrp = (C / vec + 63) / 64;   // one wave per (position, 64 chunks)
    DISPATCH_T(dtype, hipLaunchKernelGGL(embed_bwd_pos_kernel<T>, dim3((Ttot * ngrp + 3) / 4), dim3(256), 0, s, (const T*)dx, B,
                                         Ttot, C, pos_grad, accumulate, pre_vals_grad, n_pre, sc, th, seed, stream_id));
  }
  return melgpt_launch_status();
}

extern "C" int melgpt_cross_entropy_fwd(const float* logits, long long ld, const long long* target, long long M, int V,
                                        float* loss_rows, float* lse, void* stream) {
  MELGPT_CHECK(logits && target && loss_rows && lse && M > 0 && V > 0 && ld >= V, MELGPT_ERR_BAD_ARG);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, ld,
                     target, M, V, loss_rows, lse);
  return melgpt_launch_status();
}

extern "C" int melgpt_cross_entropy_bwd(const float* logits, long long ld, const long long* target, const float* lse,
                                        const float* g_rows, int g_group, const float* g_scalar, float g_scale,
                                        long long M, int V, void* dlogits, long long ldd, int dtype, void* stream) {
  MELGPT_CHECK(logits && target && lse && dlogits && M > 0 && V > 0 && g_group >= 1, MELGPT_ERR_BAD_ARG);
  DISPATCH_T(dtype, hipLaunchKernelGGL(ce_bwd_kernel<T>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0,
                                       (hipStream_t)stream, logits, ld, target, lse, g_rows, g_group, g_scalar, g_scale,
                                       M, V, (T*)dlogits, ldd));
  return melgpt_launch_status();
}

extern "C" int melgpt_sum_f32(const float* in, long long n, float scale, float* out, int accumulate, void* stream) {
  MELGPT_CHECK(in && out && n > 0, MELGPT_ERR_BAD_ARG);
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, in, n, scale, out, accumulate);
  return melgpt_launch_status();
}

extern "C" int melgpt_dropout_apply(const void* x, void* y, long long n, float drop_p, unsigned long long seed,
                                    unsigned stream_id, int dtype, void* stream) {
  MELGPT_CHECK(x && y && n > 0 && drop_p > 0.f && drop_p < 1.f, MELGPT_ERR_BAD_ARG);
  const int vec = dtype == MELGPT_F32 ? 4 : 8;
  MELGPT_CHECK(n % vec == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, MELGPT_ERR_ALIGN);
  DISPATCH_T(dtype, hipLaunchKernelGGL(dropout_apply_kernel<T>, dim3(grid_for(n / vec, 256)), dim3(256), 0,
                                       (hipStream_t)stream, (const T*)x, (T*)y, n, 1.f / (1.f - drop_p),
                                       thresh_of(drop_p), seed, stream_id));
  return melgpt_launch_status();
}

extern "C" int melgpt_dropout_apply_colsum(const void* x, void* y, long long M, int N, float drop_p,
                                           unsigned long long seed, unsigned stream_id, float* out, int accumulate,
                                           float* workspace, int dtype, void* stream) {
  MELGPT_CHECK(x && y && out && workspace && M > 0 && N > 0 && drop_p > 0.f && drop_p < 1.f, MELGPT_ERR_BAD_ARG);
  const int vec = dtype == MELGPT_F32 ? 4 : 8;
  MELGPT_CHECK(N % vec == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, MELGPT_ERR_ALIGN);
  // the geometry of melgpt_colsum
  long long rows = (M + 3) / 4;
  const int nch = N / vec;
  int gy = 2048 / ((nch + 63) / 64);
  gy = gy < 16 ? 16 : gy > 256 ? 256 : gy;
  if (rows < gy) gy = (int)rows;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_T(dtype, hipLaunchKernelGGL(dropout_colsum_kernel<T>, dim3((nch + 63) / 64, gy), dim3(256), 0, s, (const T*)x,
                                       (T*)y, M, N, 1.f / (1.f - drop_p), thresh_of(drop_p), seed, stream_id, workspace));
  launch_reduce_rows(workspace, gy, (long long)N, (long long)N, out, nullptr, 0, accumulate, 1.0f, s);
  return melgpt_launch_status();
}

// ---------------------------------------------------------------- zero fill
// (the library's own memset: the guard rows / padded slices the host code clears inside a step go through the C ABI like
// every other tensor operation of the path - no torch fill kernel inside a training step, tests/test_step_kernels_gpu.py)
__global__ __launch_bounds__(256) void zero_bytes_kernel(char* p, long long bytes) {
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, nth = (long long)gridDim.x * 256;
  long long head = (16 - (long long)((uintptr_t)p & 15)) & 15;  // bytes in front of the first 16-byte boundary
  head = head < bytes ? head : bytes;
  const long long nvec = (bytes - head) >> 4;
  for (long long i = tid; i < nvec; i += nth) *(u32x4*)(p + head + 16 * i) = u32x4{0u, 0u, 0u, 0u};
  for (long long j = tid; j < head; j += nth) p[j] = 0;
  for (long long j = head + 16 * nvec + tid; j < bytes; j += nth) p[j] = 0;
}

extern "C" int melgpt_zero_bytes(void* p, long long bytes, void* stream) {
  MELGPT_CHECK(p && bytes > 0, MELGPT_ERR_BAD_ARG);
  hipLaunchKernelGGL(zero_bytes_kernel, dim3(grid_for((bytes + 15) / 16, 256)), dim3(256), 0, (hipStream_t)stream, (char*)p, bytes);
  return melgpt_launch_status();
}

extern "C" int melgpt_cast(const void* x, int src_dtype, void* y, int dst_dtype, long long n, void* stream) {
  MELGPT_CHECK(x && y && n > 0, MELGPT_ERR_BAD_ARG);
  hipStream_t s = (hipStream_t)stream;
  if (src_dtype == MELGPT_F32 && dst_dtype == MELGPT_BF16 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for((n + 7) / 8, 256)), dim3(256), 0, s, (const float*)x,
                       (bf16_t*)y, n);
  } else if (src_dtype == MELGPT_F32 && dst_dtype == MELGPT_BF16) {
    hipLaunchKernelGGL((cast_strided_kernel<float, bf16_t>), dim3(grid_for(n, 256)), dim3(256), 0, s, (const float*)x,
                       (bf16_t*)y, n);
  } else if (src_dtype == MELGPT_BF16 && dst_dtype == MELGPT_F32) {
    hipLaunchKernelGGL((cast_strided_kernel<bf16_t, float>), dim3(grid_for(n, 256)), dim3(256), 0, s,
                       (const bf16_t*)x, (float*)y, n);
  } else if (src_dtype == MELGPT_F32 && dst_dtype == MELGPT_F32) {
    hipLaunchKernelGGL((cast_strided_kernel<float, float>), dim3(grid_for(n, 256)), dim3(256), 0, s, (const float*)x,
                       (float*)y, n);
  } else {
    return MELGPT_ERR_UNSUPPORTED;
  }
  return melgpt_launch_status();
}

extern "C" int melgpt_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16,
                            long long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                            float grad_scale, void* stream) {
  MELGPT_CHECK(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0 &&
                   ((uintptr_t)param_bf16 & 7) == 0,
               MELGPT_ERR_ALIGN);
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  // one 16-byte chunk of each of the seven streams per thread, no grid cap: a 4 096-workgroup grid walking the buffer in
  // two-chunk trips ran the 302.85 M-parameter step at 4.67 TB/s, 65 536 workgroups at 5.13, one chunk per thread
  // (296 k workgroups) at 5.45 (1.945 -> 1.667 ms; tools/lab/stream_ab.py)
  constexpr int cap = 1 << 22;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((n + 3) / 4, 256, cap)), dim3(256), 0, (hipStream_t)stream, param,
                     grad, exp_avg, exp_avg_sq, (bf16_t*)param_bf16, n, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                     (float)sqrt(bc2), grad_scale);
  return melgpt_launch_status();
}

extern "C" int melgpt_codes_permute(const long long* in, long long* out, int B, int H, int W, int reverse,
                                    void* stream) {
  MELGPT_CHECK(in && out && in != out && B > 0 && H > 0 && W > 0, MELGPT_ERR_BAD_ARG);
  hipLaunchKernelGGL(codes_permute_kernel, dim3(grid_for((long long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                     in, out, B, H, W, reverse);
  return melgpt_launch_status();
}

extern "C" int melgpt_onehot_rows(const long long* idx, long long idx_ld, int B, int Tt, int n_pre, int V, void* out,
                                  int dtype, void* stream) {
  MELGPT_CHECK(idx && out && B > 0 && Tt > 0 && n_pre >= 0 && V > 0 && idx_ld >= Tt, MELGPT_ERR_BAD_ARG);
  const long long total = (long long)B * (Tt + n_pre) * V;
  DISPATCH_T(dtype, hipLaunchKernelGGL(onehot_rows_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0,
                                       (hipStream_t)stream, idx, idx_ld, B, Tt, n_pre, V, (T*)out));
  return melgpt_launch_status();
}

extern "C" int melgpt_reduce_rows(const float* partials, int R, long long ld, long long ncols, float* out,
                                  int accumulate, float scale, void* stream) {
  MELGPT_CHECK(partials && out && R > 0 && ncols > 0 && ld >= ncols, MELGPT_ERR_BAD_ARG);
  launch_reduce_rows(partials, R, ld, ncols, out, nullptr, 0, accumulate, scale, (hipStream_t)stream);
  return melgpt_launch_status();
}

// melgpt_reduce_rows twice in ONE launch (jobs a, b; scale 1): out_x[c] (+)= sum_r part_x[r * ld_x + c].  The same bits as two
// calls.  Operands that do not allow 16-byte accesses: two launches.
extern "C" int melgpt_reduce_rows_pair(const float* part_a, int Ra, long long lda, long long ncols_a, float* out_a,
                                       int accumulate_a, const float* part_b, int Rb, long long ldb, long long ncols_b,
                                       float* out_b, int accumulate_b, void* stream) {
  MELGPT_CHECK(part_a && out_a && Ra > 0 && ncols_a > 0 && lda >= ncols_a, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(part_b && out_b && Rb > 0 && ncols_b > 0 && ldb >= ncols_b, MELGPT_ERR_BAD_ARG);
  hipStream_t s = (hipStream_t)stream;
  const bool vec = ncols_a % 4 == 0 && lda % 4 == 0 && ncols_b % 4 == 0 && ldb % 4 == 0 &&
                   ((((uintptr_t)part_a | (uintptr_t)out_a | (uintptr_t)part_b | (uintptr_t)out_b) & 15) == 0);
  const long long ga4 = (ncols_a / 4 + 15) / 16, gafew = (ncols_a / 4 + 255) / 256, gb = (ncols_b / 4 + 15) / 16;
  if (!vec || ga4 + gb > 0x7FFFFFFF) {
    launch_reduce_rows(part_a, Ra, lda, ncols_a, out_a, nullptr, 0, accumulate_a, 1.0f, s);
    launch_reduce_rows(part_b, Rb, ldb, ncols_b, out_b, nullptr, 0, accumulate_b, 1.0f, s);
    return melgpt_launch_status();
  }
  const ReduceJob a{part_a, Ra, lda, ncols_a, out_a, accumulate_a}, b{part_b, Rb, ldb, ncols_b, out_b, accumulate_b};
  const bool few = Ra <= 8 && ncols_a >= (1 << 16);  // (launch_reduce_rows' rule)
  const int g1 = (int)(few ? gafew : ga4);
  const dim3 grid((unsigned)(g1 + gb));
  if (!few) hipLaunchKernelGGL(reduce_pair_kernel<0>, grid, dim3(256), 0, s, a, g1, b);
  else if (Ra <= 2) hipLaunchKernelGGL(reduce_pair_kernel<2>, grid, dim3(256), 0, s, a, g1, b);
  else if (Ra <= 4) hipLaunchKernelGGL(reduce_pair_kernel<4>, grid, dim3(256), 0, s, a, g1, b);
  else hipLaunchKernelGGL(reduce_pair_kernel<8>, grid, dim3(256), 0, s, a, g1, b);
  return melgpt_launch_status();
}

extern "C" int melgpt_group_sum_f32(const float* in, long long groups, int n, float scale, float* out, void* stream) {
  MELGPT_CHECK(in && out && groups > 0 && n > 0, MELGPT_ERR_BAD_ARG);
  hipLaunchKernelGGL(group_sum_kernel, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, groups,
                     n, scale, out);
  return melgpt_launch_status();
}

extern "C" int melgpt_sample_logits(const float* logits, long long ld, int rows, int V, float temperature, int top_k,
                                    int do_sample, unsigned long long seed, unsigned step, long long* out,
                                    float* probs_out, void* stream) {
  MELGPT_CHECK(logits && out && rows > 0 && V > 0 && ld >= V && temperature > 0.f, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(V <= 1024, MELGPT_ERR_UNSUPPORTED);
  hipLaunchKernelGGL(sample_logits_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ld, V, temperature,
                     top_k, do_sample, seed, step, out, probs_out, (const int*)nullptr, 0, (long long*)nullptr, 0LL);
  return melgpt_launch_status();
}

extern "C" int melgpt_sample_logits_dev(const float* logits, long long ld, int rows, int V, float temperature, int top_k,
                                        int do_sample, unsigned long long seed, const int* pos_dev, int step_offset,
                                        long long* out, long long* seq, long long seq_ld, void* stream) {
  MELGPT_CHECK(logits && out && pos_dev && rows > 0 && V > 0 && ld >= V && temperature > 0.f, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(V <= 1024, MELGPT_ERR_UNSUPPORTED);
  hipLaunchKernelGGL(sample_logits_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ld, V, temperature,
                     top_k, do_sample, seed, 0u, out, (float*)nullptr, pos_dev, step_offset, seq, seq_ld);
  return melgpt_launch_status();
}

extern "C" int melgpt_vae_reparam_fwd(const float* stats, float* eps, int gen_eps, unsigned long long seed, int B, int ns,
                                      int nz, float* z, float* kl, void* stream) {
  MELGPT_CHECK(stats && eps && z && kl && B > 0 && ns > 0 && nz > 0, MELGPT_ERR_BAD_ARG);
  hipLaunchKernelGGL(vae_reparam_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, stats, eps, gen_eps, seed, B, ns,
                     nz, z, kl);
  return melgpt_launch_status();
}

extern "C" int melgpt_gauss_log_density(const float* z, const float* mu, const float* logvar, long long ld_stats, int X,
                                        int S, int nz, int pairwise, float* out, void* stream) {
  MELGPT_CHECK(z && mu && logvar && out && X > 0 && S > 0 && nz > 0 && ld_stats >= nz, MELGPT_ERR_BAD_ARG);
  const long long pairs = (long long)X * S;
  MELGPT_CHECK(pairs <= 0x7FFFFFFFLL * 4, MELGPT_ERR_UNSUPPORTED);
  hipLaunchKernelGGL(gauss_log_density_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z, mu,
                     logvar, ld_stats, X, S, nz, pairwise, out);
  return melgpt_launch_status();
}

extern "C" int melgpt_vae_calc_mi(const float* mu, const float* logvar, long long ld_stats, float* eps, int gen_eps,
                                  unsigned long long seed, int B, int nz, float* workspace, float* mi, void* stream) {
  MELGPT_CHECK(mu && logvar && eps && workspace && mi && B > 0 && nz > 0 && ld_stats >= nz, MELGPT_ERR_BAD_ARG);
  float* zbuf = workspace;                         // (B, nz)
  float* term = workspace + (long long)B * nz;     // (B,)
  hipLaunchKernelGGL(vae_calc_mi_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mu, logvar, ld_stats, eps, gen_eps,
                     seed, B, nz, zbuf, term);
  int st = melgpt_launch_status();
  if (st != MELGPT_OK) return st;
  return melgpt_sum_f32(term, B, 1.0f / (float)B, mi, 0, stream);
}

extern "C" int melgpt_vae_reparam_bwd(const float* stats, const float* eps, const float* dz, const float* dkl, int B,
                                      int ns, int nz, float* dstats, void* stream) {
  MELGPT_CHECK(stats && eps && dstats && B > 0 && ns > 0 && nz > 0, MELGPT_ERR_BAD_ARG);
  const long long n = (long long)B * nz;
  hipLaunchKernelGGL(vae_reparam_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stats,
                     eps, dz, dkl, B, ns, nz, dstats);
  return melgpt_launch_status();
}
