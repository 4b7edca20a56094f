// Fused multi-head self-attention for minGPT on gfx950 (reference transformer/minGPT.py:72-90):
//   att = softmax(mask(q k^T / sqrt(hs)));  y = dropout(att) v      - forward, plus the two backward kernels.
// The (B,H,T,T) score / probability / dropped tensors of the reference are never written to HBM; the
// post-softmax, pre-dropout `att` that the reference RETURNS (:90) is emitted only on request (last block).
//
// Shape regime: T <= 288 (block_size 265/266), head size 64.  The whole K and V (or Q and dO) of one
// (batch, head) live in LDS, so softmax is a plain two-pass row softmax in registers - no online rescaling.
//   forward / dQ kernel : one 512-thread workgroup per (batch, head): K and V staged into LDS ONCE, the 8 waves pull
//                         16-query-row tiles from an LDS work counter, heaviest (latest causal rows) first; keys
//                         beyond a tile's causal frontier are skipped
//   dK/dV kernel        : same shape with Q and dO in LDS; waves pull 16-key tiles (key tile 0 sees every query)
// MFMA orientation is chosen so that no probability tile ever crosses lanes or LDS:
//   S^T = K Q^T puts the query on the lane -> row max/sum are in-register + 2 xor-shuffles, and the S^T
//   accumulators are directly the B operand of O^T = V^T P^T (V^T via ds_read_b64_tr_b16 transposed reads);
//   in the dK/dV kernel S = Q K^T puts the key on the lane and the accumulators feed dV^T and dK^T.
// The causal / n_unmasked mask (minGPT.py:65-69) is computed from (row, col, n_unmasked) - the persistent
// (1,1,bs,bs) `mask` buffer of the reference is never read.  Dropout masks are Philox4x32-10 keyed by
// (seed, stream, (b,h,q), key/4) and are regenerated bit-identically in the backward kernels.
// T = bf16: v_mfma_f32_16x16x32_bf16;  T = f32: v_mfma_f32_16x16x4_f32 (exact f32) - same code path.
#include "mma.h"

namespace {

constexpr int HS = 64;
constexpr int NTHREADS = 512;  // 8 waves per (batch, head)
constexpr int MAXT = 288;
constexpr int MAXKT = MAXT / 16;

struct AttnParams {
  const void *Q, *K, *V;  // rows b*T+t, row stride ld (elements); head h at columns [h*64, h*64+64)
  long long ld;
  void* O;  // fwd out / bwd in: (B*T, H*64) row stride ldo
  long long ldo;
  const void* dO;
  void *dQ, *dK, *dV;  // row stride ldg
  long long ldg;
  float* lse;    // (B,H,T)
  float* delta;  // (B,H,T)
  float* att;    // optional (B,H,T,T) f32
  int B, H, T, n_unmasked;
  float scale;
  float drop_scale;  // 1/(1-p) or 0
  unsigned drop_thresh;
  unsigned long long seed;
  unsigned stream_id;
};

template <typename T>
struct AT {
  static constexpr int ES = Tr<T>::ES;
  static constexpr int ROWB = HS * ES;    // 128 / 256 bytes per tile row
  static constexpr int CP = ROWB / 16;    // 16-byte chunks per row
  static constexpr int NKS = CP / 4;      // k-substeps over the head dimension (2 / 4)
  static constexpr int TPS = ES == 2 ? 2 : 1;  // accumulator tiles consumed per acc-as-operand MFMA step
  static constexpr int MAXST = MAXKT / TPS;
  static constexpr int VEC = 16 / ES;
};

// row-read friendly swizzle (also serves transposed reads: conflict-free for f32, 2-way for bf16)
template <typename T>
__device__ __forceinline__ int offK(int row, int c) {
  if constexpr (Tr<T>::ES == 2) return row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
  else return row * 256 + ((c ^ (row & 15)) << 4);
}
// transposed-read friendly swizzle (tile that is only read through ds_read_b64_tr_b16)
template <typename T>
__device__ __forceinline__ int offV(int row, int c) {
  if constexpr (Tr<T>::ES == 2) return row * 128 + ((c ^ (((row >> 1) & 3) << 1)) << 4);
  else return offK<T>(row, c);
}

template <typename T, bool VSWZ>
__device__ __forceinline__ int off(int row, int c) {
  if constexpr (VSWZ) return offV<T>(row, c);
  else return offK<T>(row, c);
}

template <typename T, bool VSWZ>
__device__ __forceinline__ void load_tile(char* tile, const T* base, long long ld, int nvalid, int nfill, int t) {
  constexpr int CP = AT<T>::CP, VEC = AT<T>::VEC;
  for (int q = t; q < nfill * CP; q += NTHREADS) {
    const int row = q / CP, c = q % CP;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < nvalid) v = *(const u32x4*)(base + (long long)row * ld + c * VEC);
    *(u32x4*)(tile + off<T, VSWZ>(row, c)) = v;
  }
}

// fragment whose 16 "row" indices are tile rows 16*tile16 + (lane&15); contraction over the head dim (substep ks)
template <typename T, bool VSWZ>
__device__ __forceinline__ u32x4 frag_row(const char* tile, int tile16, int ks, int lane) {
  return *(const u32x4*)(tile + off<T, VSWZ>(16 * tile16 + (lane & 15), 4 * ks + (lane >> 4)));
}

// fragment whose 16 "row" indices are head-dim columns 16*dt + (lane&15); contraction over TILE ROWS, step st
// (32 rows for bf16: k-slot j<4 -> row 32st+4g+j, j>=4 -> row 32st+16+4g+j-4;  16 rows for f32: slot e -> 16st+4g+e)
template <typename T, bool VSWZ>
__device__ __forceinline__ u32x4 frag_tr(const char* tile, int st, int dt, int lane) {
  const int i = lane & 15, g = lane >> 4;
  if constexpr (Tr<T>::ES == 2) {
    const int qq = i >> 2, pp = i & 3;
    const int r0 = 32 * st + 4 * g + qq, c = 2 * dt + (pp >> 1);
    const char* a0 = tile + off<T, VSWZ>(r0, c) + 8 * (pp & 1);
    const char* a1 = tile + off<T, VSWZ>(r0 + 16, c) + 8 * (pp & 1);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1));
    s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(u32x4, f);
  } else {
    const int col = 16 * dt + i;
    u32x4 f;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      f[e] = *(const unsigned*)(tile + off<T, VSWZ>(16 * st + 4 * g + e, col >> 2) + (col & 3) * 4);
    return f;
  }
}

// accumulator tile(s) -> MFMA operand contracting over the accumulator's ROW index
template <typename T>
__device__ __forceinline__ u32x4 pack_operand(f32x4 lo, f32x4 hi) {
  if constexpr (Tr<T>::ES == 2)
    return u32x4{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]),
                 pack_bf16x2(hi[2], hi[3])};
  else
    return __builtin_bit_cast(u32x4, lo);
}

template <typename T>
__device__ __forceinline__ void store4(T* p, f32x4 v) {
  if constexpr (Tr<T>::ES == 2) *(u32x2*)p = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  else *(f32x4*)p = v;
}

__device__ __forceinline__ bool allowed(int q, int key, int T, int nu) {
  return key < T && (key <= q || (q < nu && key < nu));
}

__device__ __forceinline__ int rup(int x, int m) { return (x + m - 1) / m * m; }

// ================================================================================================ forward
template <typename T, bool BWD>
__global__ __launch_bounds__(NTHREADS, sizeof(T) == 2 ? 4 : 2) void attn_q_kernel(AttnParams p) {
  // BWD == false: forward (O, lse, optional att).   BWD == true: dQ (+ delta) from dO, recomputing P from lse.
  using A = AT<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, i16 = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z, Tn = p.T, nu = p.n_unmasked;
  const int TP = rup(Tn, 32);
  char* Kt = smem;
  char* Vt = smem + (size_t)TP * A::ROWB;
  int* ctr = (int*)(smem + 2 * (size_t)TP * A::ROWB);
  const T* Kg = (const T*)p.K + (long long)b * Tn * p.ld + h * HS;
  const T* Vg = (const T*)p.V + (long long)b * Tn * p.ld + h * HS;
  load_tile<T, false>(Kt, Kg, p.ld, Tn, TP, t);
  load_tile<T, !BWD>(Vt, Vg, p.ld, Tn, TP, t);  // fwd: V only via transposed reads
  if (t == 0) *ctr = 0;
  __syncthreads();
  const long long bh = (long long)b * p.H + h;
  const int ntiles = (Tn + 15) / 16;

 for (;;) {  // ---- this wave's next 16-row query tile (heaviest first)
  int job = 0;
  if (lane == 0) job = atomicAdd(ctr, 1);
  job = __shfl(job, 0, 64);
  if (job >= ntiles) break;
  const int q0 = 16 * (ntiles - 1 - job);
  const int q = q0 + i16, qc = min(q, Tn - 1);
  int kw = min(q0 + 16, Tn);
  if (nu > q0) kw = max(kw, min(nu, Tn));
  const int nkt = (kw + 15) / 16;

  u32x4 qf[A::NKS];
  {
    const T* qp = (const T*)p.Q + ((long long)b * Tn + qc) * p.ld + h * HS;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) qf[ks] = *(const u32x4*)(qp + (4 * ks + g) * A::VEC);
  }
  u32x4 dof[A::NKS];
  float delta = 0.f, lse_q = 0.f;
  if constexpr (BWD) {
    const T* dp = (const T*)p.dO + ((long long)b * Tn + qc) * p.ldo + h * HS;
    const T* op = (const T*)p.O + ((long long)b * Tn + qc) * p.ldo + h * HS;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      dof[ks] = *(const u32x4*)(dp + (4 * ks + g) * A::VEC);
      u32x4 ov = *(const u32x4*)(op + (4 * ks + g) * A::VEC);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (A::ES == 2) {
          delta = fmaf(bf16lo(dof[ks][e]), bf16lo(ov[e]), delta);
          delta = fmaf(bf16hi(dof[ks][e]), bf16hi(ov[e]), delta);
        } else {
          delta = fmaf(__uint_as_float(dof[ks][e]), __uint_as_float(ov[e]), delta);
        }
      }
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    if (g == 0 && q < Tn) p.delta[bh * Tn + q] = delta;
    lse_q = p.lse[bh * Tn + qc];
  }

  f32x4 s[MAXKT];
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt) s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned long long row_ctr = (unsigned long long)(bh * Tn + qc) * 128ull;

  if constexpr (!BWD) {
    float m = -__builtin_inff();
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt) {
      if (kt < nkt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < A::NKS; ++ks) mma<T>(acc, frag_row<T, false>(Kt, kt, ks, lane), qf[ks]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * kt + 4 * g + r;
          const float v = allowed(qc, key, Tn, nu) ? acc[r] * p.scale : -__builtin_inff();
          acc[r] = v;
          m = fmaxf(m, v);
        }
        s[kt] = acc;
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt) {
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __expf(s[kt][r] - m);
          s[kt][r] = e;
          l += e;
        }
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (g == 0 && q < Tn) p.lse[bh * Tn + q] = m + __logf(l);
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt) {
      if (kt < nkt) {
        f32x4 pv = s[kt] * inv;
        if (p.att && q < Tn) {
          float* ap = p.att + (bh * Tn + q) * Tn;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * g + r;
            if (key < Tn) ap[key] = pv[r];
          }
        }
        if (p.drop_scale != 0.f) {
          const unsigned keep = dropout_keep4(p.seed, p.stream_id, row_ctr + (4 * kt + g), p.drop_thresh);
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[r] = (keep >> r & 1) ? pv[r] * p.drop_scale : 0.f;
        }
        s[kt] = pv;
      }
    }
    // keys the other waves' causal frontier reaches but this wave's does not must read as probability 0
    if (p.att && q < Tn) {
      float* ap = p.att + (bh * Tn + q) * Tn;
      for (int key = 16 * nkt + 4 * g; key < Tn; key += 16)
        for (int r = 0; r < 4; ++r)
          if (key + r < Tn) ap[key + r] = 0.f;
    }
  } else {
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt) {
      if (kt < nkt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < A::NKS; ++ks) mma<T>(acc, frag_row<T, false>(Kt, kt, ks, lane), qf[ks]);
#pragma unroll
        for (int ks = 0; ks < A::NKS; ++ks) mma<T>(dp, frag_row<T, false>(Vt, kt, ks, lane), dof[ks]);
        unsigned keep = 0xF;
        if (p.drop_scale != 0.f) keep = dropout_keep4(p.seed, p.stream_id, row_ctr + (4 * kt + g), p.drop_thresh);
        const float dsc = p.drop_scale != 0.f ? p.drop_scale : 1.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * kt + 4 * g + r;
          const float pr = allowed(qc, key, Tn, nu) ? __expf(acc[r] * p.scale - lse_q) : 0.f;
          const float dpm = (keep >> r & 1) ? dp[r] * dsc : 0.f;
          acc[r] = pr * (dpm - delta) * p.scale;
        }
        s[kt] = acc;
      }
    }
  }

  // O^T (or dQ^T) [d][q] = sum_key  X^T[d][key] * S^T[key][q],  X = V (fwd) or K (bwd)
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nst = (nkt + A::TPS - 1) / A::TPS;
  const char* Xt = BWD ? Kt : Vt;
#pragma unroll
  for (int st = 0; st < A::MAXST; ++st) {
    if (st < nst) {
      const u32x4 bop = pack_operand<T>(s[A::TPS * st], s[A::TPS * st + (A::TPS - 1)]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) mma<T>(o[dt], frag_tr<T, !BWD>(Xt, st, dt, lane), bop);
    }
  }
  if (q < Tn) {
    T* op = BWD ? (T*)p.dQ + ((long long)b * Tn + q) * p.ldg + h * HS
                : (T*)p.O + ((long long)b * Tn + q) * p.ldo + h * HS;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4<T>(op + 16 * dt + 4 * g, o[dt]);
  }
 }  // tile loop
}

// ============================================================================================ dK / dV
// mma<T>(acc, a, b) computes D[row of a][col of b]:  here rows = queries (Q / dO tile fragments on the A port),
// columns = this wave's 16 keys (K / V rows held in registers on the B port), so every lane owns one key column
// and the S / dP accumulators are directly the B operands of  dV^T = dO^T P_drop  and  dK^T = Q^T dS.
template <typename T>
__global__ __launch_bounds__(NTHREADS, sizeof(T) == 2 ? 4 : 2) void attn_dkv_kernel(AttnParams p) {
  using A = AT<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, i16 = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z, Tn = p.T, nu = p.n_unmasked;
  const int TP = rup(Tn, 32);
  char* Qt = smem;
  char* Dt = smem + (size_t)TP * A::ROWB;
  float* lse_s = (float*)(smem + 2 * (size_t)TP * A::ROWB);
  float* del_s = lse_s + TP;
  int* ctr = (int*)(del_s + TP);
  const long long bh = (long long)b * p.H + h;
  load_tile<T, false>(Qt, (const T*)p.Q + (long long)b * Tn * p.ld + h * HS, p.ld, Tn, TP, t);
  load_tile<T, false>(Dt, (const T*)p.dO + (long long)b * Tn * p.ldo + h * HS, p.ldo, Tn, TP, t);
  for (int j = t; j < TP; j += NTHREADS) {
    lse_s[j] = j < Tn ? p.lse[bh * Tn + j] : 0.f;
    del_s[j] = j < Tn ? p.delta[bh * Tn + j] : 0.f;
  }
  if (t == 0) *ctr = 0;
  __syncthreads();
  const int ntiles = (Tn + 15) / 16;

 for (;;) {  // ---- this wave's next 16-key tile (key tile 0 is seen by every query: heaviest first)
  int job = 0;
  if (lane == 0) job = atomicAdd(ctr, 1);
  job = __shfl(job, 0, 64);
  if (job >= ntiles) break;
  const int kt = job, key0 = 16 * kt;
  const int key = key0 + i16, kc = min(key, Tn - 1);
  u32x4 kf[A::NKS], vf[A::NKS];
  {
    const T* kp = (const T*)p.K + ((long long)b * Tn + kc) * p.ld + h * HS;
    const T* vp = (const T*)p.V + ((long long)b * Tn + kc) * p.ld + h * HS;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      kf[ks] = *(const u32x4*)(kp + (4 * ks + g) * A::VEC);
      vf[ks] = *(const u32x4*)(vp + (4 * ks + g) * A::VEC);
    }
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int q_start = (nu > key0) ? 0 : key0;  // first query row that can see any of these keys
  const int st0 = (q_start / 16) / A::TPS;
  const int st1 = ((Tn + 15) / 16 + A::TPS - 1) / A::TPS;
  const float dsc = p.drop_scale != 0.f ? p.drop_scale : 1.f;
  for (int st = st0; st < st1; ++st) {
    f32x4 pd[2], ds[2];
    pd[1] = ds[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tt = 0; tt < A::TPS; ++tt) {
      const int qt = A::TPS * st + tt;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < A::NKS; ++ks) mma<T>(acc, frag_row<T, false>(Qt, qt, ks, lane), kf[ks]);
#pragma unroll
      for (int ks = 0; ks < A::NKS; ++ks) mma<T>(dp, frag_row<T, false>(Dt, qt, ks, lane), vf[ks]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 16 * qt + 4 * g + r;  // accumulator row
        const bool ok = q < Tn && allowed(q, key, Tn, nu);
        const float pr = ok ? __expf(acc[r] * p.scale - lse_s[q]) : 0.f;
        bool keep = true;
        if (p.drop_scale != 0.f) {
          const unsigned long long ctr = (unsigned long long)(bh * Tn + min(q, Tn - 1)) * 128ull + (unsigned)(key >> 2);
          keep = (dropout_keep4(p.seed, p.stream_id, ctr, p.drop_thresh) >> (key & 3)) & 1;
        }
        const float pdr = keep ? pr * dsc : 0.f;
        const float dpm = keep ? dp[r] * dsc : 0.f;
        acc[r] = pdr;
        dp[r] = pr * (dpm - del_s[q]) * p.scale;
      }
      pd[tt] = acc;
      ds[tt] = dp;
    }
    const u32x4 bp = pack_operand<T>(pd[0], pd[1]);
    const u32x4 bs = pack_operand<T>(ds[0], ds[1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      mma<T>(dv[dt], frag_tr<T, false>(Dt, st, dt, lane), bp);
      mma<T>(dk[dt], frag_tr<T, false>(Qt, st, dt, lane), bs);
    }
  }
  if (key < Tn) {
    T* kp = (T*)p.dK + ((long long)b * Tn + key) * p.ldg + h * HS;
    T* vp = (T*)p.dV + ((long long)b * Tn + key) * p.ldg + h * HS;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      store4<T>(kp + 16 * dt + 4 * g, dk[dt]);
      store4<T>(vp + 16 * dt + 4 * g, dv[dt]);
    }
  }
 }  // tile loop
}

template <typename T>
size_t lds_bytes(int Tn, bool with_stats) {
  const int TP = (Tn + 31) / 32 * 32;
  return 2 * (size_t)TP * AT<T>::ROWB + (with_stats ? 2 * (size_t)TP * 4 : 0) + 16;
}

int validate(const AttnParams& p, int hs, int dtype) {
  if (p.B <= 0 || p.H <= 0 || p.T <= 0) return MELGPT_ERR_BAD_ARG;
  if (hs != HS || p.T > MAXT) return MELGPT_ERR_UNSUPPORTED;
  if (dtype != MELGPT_F32 && dtype != MELGPT_BF16) return MELGPT_ERR_UNSUPPORTED;
  const int vec = dtype == MELGPT_F32 ? 4 : 8;
  if (p.ld % vec || p.ldo % vec || p.ldg % vec) return MELGPT_ERR_ALIGN;
  return MELGPT_OK;
}

void set_dropout(AttnParams& p, float drop_p, unsigned long long seed, unsigned sid) {
  if (drop_p > 0.f) {
    p.drop_scale = 1.0f / (1.0f - drop_p);
    double th = (double)drop_p * 4294967296.0;
    p.drop_thresh = th >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)th;
  }
  p.seed = seed;
  p.stream_id = sid;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess
             ? MELGPT_OK
             : MELGPT_ERR_LAUNCH;
}

}  // namespace

extern "C" int melgpt_attn_fwd(const void* q, const void* k, const void* v, long long ld, void* out, long long ldo,
                               float* lse, float* att, int B, int H, int T, int head_size, int n_unmasked,
                               float drop_p, unsigned long long seed, unsigned stream_id, int dtype, void* stream) {
  MELGPT_CHECK(q && k && v && out && lse, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(drop_p >= 0.f && drop_p < 1.f, MELGPT_ERR_BAD_ARG);
  AttnParams p{};
  p.Q = q; p.K = k; p.V = v; p.ld = ld; p.O = out; p.ldo = ldo; p.ldg = ldo; p.lse = lse; p.att = att;
  p.B = B; p.H = H; p.T = T; p.n_unmasked = n_unmasked;
  p.scale = 1.0f / sqrtf((float)head_size);
  int st = validate(p, head_size, dtype);
  if (st != MELGPT_OK) return st;
  MELGPT_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, MELGPT_ERR_ALIGN);
  set_dropout(p, drop_p, seed, stream_id);
  dim3 grid(1, H, B);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MELGPT_F32) {
    size_t lds = lds_bytes<float>(T, false);
    if (set_lds(attn_q_kernel<float, false>, lds_bytes<float>(MAXT, false)) != MELGPT_OK) return MELGPT_ERR_LAUNCH;
    hipLaunchKernelGGL((attn_q_kernel<float, false>), grid, dim3(NTHREADS), lds, s, p);
  } else {
    size_t lds = lds_bytes<bf16_t>(T, false);
    if (set_lds(attn_q_kernel<bf16_t, false>, lds_bytes<bf16_t>(MAXT, false)) != MELGPT_OK) return MELGPT_ERR_LAUNCH;
    hipLaunchKernelGGL((attn_q_kernel<bf16_t, false>), grid, dim3(NTHREADS), lds, s, p);
  }
  return melgpt_launch_status();
}

extern "C" int melgpt_attn_bwd(const void* q, const void* k, const void* v, long long ld, const void* out,
                               const void* dout, long long ldo, const float* lse, float* delta, void* dq, void* dk,
                               void* dv, long long ldg, int B, int H, int T, int head_size, int n_unmasked,
                               float drop_p, unsigned long long seed, unsigned stream_id, int dtype, void* stream) {
  MELGPT_CHECK(q && k && v && out && dout && lse && delta && dq && dk && dv, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(drop_p >= 0.f && drop_p < 1.f, MELGPT_ERR_BAD_ARG);
  AttnParams p{};
  p.Q = q; p.K = k; p.V = v; p.ld = ld; p.O = const_cast<void*>(out); p.dO = dout; p.ldo = ldo;
  p.dQ = dq; p.dK = dk; p.dV = dv; p.ldg = ldg; p.lse = const_cast<float*>(lse); p.delta = delta;
  p.B = B; p.H = H; p.T = T; p.n_unmasked = n_unmasked;
  p.scale = 1.0f / sqrtf((float)head_size);
  int st = validate(p, head_size, dtype);
  if (st != MELGPT_OK) return st;
  MELGPT_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dq |
                 (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
               MELGPT_ERR_ALIGN);
  set_dropout(p, drop_p, seed, stream_id);
  hipStream_t s = (hipStream_t)stream;
  dim3 gq(1, H, B), gk(1, H, B);
  if (dtype == MELGPT_F32) {
    if (set_lds(attn_q_kernel<float, true>, lds_bytes<float>(MAXT, false)) != MELGPT_OK ||
        set_lds(attn_dkv_kernel<float>, lds_bytes<float>(MAXT, true)) != MELGPT_OK)
      return MELGPT_ERR_LAUNCH;
    hipLaunchKernelGGL((attn_q_kernel<float, true>), gq, dim3(NTHREADS), lds_bytes<float>(T, false), s, p);
    hipLaunchKernelGGL((attn_dkv_kernel<float>), gk, dim3(NTHREADS), lds_bytes<float>(T, true), s, p);
  } else {
    if (set_lds(attn_q_kernel<bf16_t, true>, lds_bytes<bf16_t>(MAXT, false)) != MELGPT_OK ||
        set_lds(attn_dkv_kernel<bf16_t>, lds_bytes<bf16_t>(MAXT, true)) != MELGPT_OK)
      return MELGPT_ERR_LAUNCH;
    hipLaunchKernelGGL((attn_q_kernel<bf16_t, true>), gq, dim3(NTHREADS), lds_bytes<bf16_t>(T, false), s, p);
    hipLaunchKernelGGL((attn_dkv_kernel<bf16_t>), gk, dim3(NTHREADS), lds_bytes<bf16_t>(T, true), s, p);
  }
  return melgpt_launch_status();
}
