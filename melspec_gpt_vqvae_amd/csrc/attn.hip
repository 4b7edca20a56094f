// Fused multi-head self-attention for minGPT on gfx950 (reference transformer/minGPT.py:72-90):
//   att = softmax(mask(q k^T / sqrt(hs)));  y = dropout(att) v      - forward, plus the two backward kernels.
// The (B,H,T,T) score / probability / dropped tensors of the reference are never written to HBM; the
// post-softmax, pre-dropout `att` that the reference RETURNS (:90) is emitted only on request (last block).
//
// Shape regime: T <= 288 (block_size 265/266), head size 64.  The whole K and V (or Q and dO) of one
// (batch, head) live in LDS, so softmax is a plain two-pass row softmax - no online rescaling; pass 2 RECOMPUTES the
// logits instead of keeping a 17-tile row of them in registers (the MFMA pipe is ~10 % busy here).
//   forward / dQ kernel : one 512-thread workgroup per (batch, head): K and V staged into LDS ONCE, the 8 waves pull
//                         16-query-row tiles from an LDS work counter, heaviest (latest causal rows) first, and fetch
//                         the next tile's Q / dO / O rows under the current tile's math; keys beyond a tile's causal
//                         frontier are skipped, and only the key tiles the frontier cuts test a mask
//   dK/dV kernel        : same shape with Q and dO in LDS; waves pull 16-key tiles (key tile 0 sees every query)
// MFMA orientation is chosen so that no probability tile ever crosses lanes or LDS:
//   S^T = K Q^T puts the query on the lane -> row max/sum are in-register + 2 xor-shuffles, and the S^T
//   accumulators are directly the B operand of O^T = V^T P^T (V^T via ds_read_b64_tr_b16 transposed reads);
//   in the dK/dV kernel S = Q K^T puts the key on the lane and the accumulators feed dV^T and dK^T.
// The causal / n_unmasked mask (minGPT.py:65-69) is computed from (row, col, n_unmasked) - the persistent
// (1,1,bs,bs) `mask` buffer of the reference is never read.  Dropout masks are a counter hash (common.h) keyed by
// (seed, stream, (b,h)), regenerated bit-identically in the backward kernels: general p - one hash per (query, 4 keys),
// counter q * 128 + key / 4; p = 1/2 (the reference's attn_pdrop) needs ONE BIT per probability, so one 32-bit hash
// serves a query's 32 keys {128 kb + 16 t + 4 gk + r : t < 8, r < 4} - the keys ONE LANE of the forward / dQ kernels
// meets in eight consecutive key tiles - counter q * 16 + 4 gk + kb, bit 4 t + r: one hash per lane per EIGHT tiles
// instead of one per tile (the hash's two integer multiplies were a third of the per-probability VALU work).
// T = bf16: v_mfma_f32_16x16x32_bf16;  T = f32: v_mfma_f32_16x16x4_f32 (exact f32) - same code path.
#include <cstdlib>

#include "mma.h"



MELGPT_CLK_DECL(clk_attn_fwd)
MELGPT_CLK_DECL(clk_attn_bwd)
MELGPT_CLK_DECL(clk_attn32_ph)   // phase stamps of the four waves of workgroup (head 5, batch 3) of attn_fwd32_kernel: 64 slots per wave
#ifdef MELGPT_CLOCK_STAMPS
#define PH32_STAMP() do { if (ph_on && ph_n < 62) ph[ph_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PH32_STAMP() do { } while (0)
#endif

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
// the keys query row `row` sees are exactly [0, vis_keys): its causal frontier, widened to the n_unmasked prefix block
__device__ __forceinline__ int vis_keys(int row, int T, int nu) {
  int v = row + 1;
  if (row < nu) v = max(v, nu);
  return min(v, T);
}

__device__ __forceinline__ int rup(int x, int m) { return (x + m - 1) / m * m; }

constexpr float LOG2E = 1.4426950408889634f;

// Both tiles of a workgroup staged with every global load in flight at once (a load -> store loop pays the memory
// latency once per iteration: 20 of the forward's 118 us at the training shape).
template <typename T, bool SWZ0, bool SWZ1, int NTH = 512>
__device__ __forceinline__ void load_tiles(char* t0, const T* b0, long long ld0, char* t1, const T* b1, long long ld1,
                                           int nvalid, int nfill, int t) {
  constexpr int NTHREADS = NTH;
  constexpr int CP = AT<T>::CP, VEC = AT<T>::VEC, NIT = (MAXT * CP + NTHREADS - 1) / NTHREADS;
  u32x4 v0[NIT], v1[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    // unconditional loads of a clamped row (a branch around a load makes the compiler drain the loads before it)
    const int q = t + i * NTHREADS, row = min(q / CP, nvalid - 1), c = q % CP;
    v0[i] = *(const u32x4*)(b0 + (long long)row * ld0 + c * VEC);
    v1[i] = *(const u32x4*)(b1 + (long long)row * ld1 + c * VEC);
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int q = t + i * NTHREADS, row = q / CP, c = q % CP;
    if (q < nfill * CP) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      *(u32x4*)(t0 + off<T, SWZ0>(row, c)) = row < nvalid ? v0[i] : z;
      *(u32x4*)(t1 + off<T, SWZ1>(row, c)) = row < nvalid ? v1[i] : z;
    }
  }
}

__device__ __forceinline__ f32x4 exp2_4(f32x4 x) {
  return f32x4{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1]), __builtin_amdgcn_exp2f(x[2]),
               __builtin_amdgcn_exp2f(x[3])};
}
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }
// Dropout lanes of the kernels (compile-time, so the tile loops carry no mode branches):
//   DM_NONE  no dropout;  DM_HALF  p = 1/2 exactly (the reference's attn_pdrop): keep = top bit of the element's hash
//   byte, applied as a sign-extended 1-bit field AND - two VALU ops per probability, no VCC;  DM_ANY  any other p.
enum { DM_NONE = 0, DM_HALF = 1, DM_ANY = 2 };

// v with the dropped ones of its 4 elements zeroed; the elements are 4 consecutive keys at low counter word c0
template <int DM>
__device__ __forceinline__ f32x4 drop4(const DropKeys& d, unsigned c0, f32x4 v) {
  if constexpr (DM == DM_HALF) {
    const int hsh = (int)hash32(c0 + d.k0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      v[r] = __uint_as_float(__float_as_uint(v[r]) & (unsigned)__builtin_amdgcn_sbfe(hsh, 8 * r + 7, 1));
  } else if constexpr (DM == DM_ANY) {
    bool keep[4];
    drop_keep4(d, c0, keep);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = keep[r] ? v[r] : 0.f;
  }
  return v;
}

// value held by lane R of this lane's quad (DPP quad_perm broadcast)
template <int R>
__device__ __forceinline__ int quad_lane(int v) {
  return __builtin_amdgcn_mov_dpp(v, 0x55 * R, 0xF, 0xF, true);
}

// ================================================================================================ forward
// These kernels issue ~10x more VALU than MFMA cycles (an MFMA 16x16x32 is 16 cycles, a wave64 VALU op 4), so the
// per-probability work is pared to: max3 (pass 1); fma + exp2 + add (pass 2, packed f32 pairs); one AND with a
// sign-extended hash bit (dropout 1/2; ONE hash per four keys), cvt_pk.  Scaling by 1/sqrt(hs), the softmax normaliser
// and 1/(1-p) are applied to the 16 x 64 OUTPUT tile, not to the probabilities.
// Measured at B 128, H 16, T 265, p 1/2 (tools/lab/attn_lab.hip): forward 118 -> 82 us, backward 364 -> 275 us.
template <typename T, bool BWD, int DM, bool ATT>
__global__ __launch_bounds__(NTHREADS, sizeof(T) == 2 ? 4 : 2) void attn_q_kernel(AttnParams p) {
  // BWD == false: forward (O, lse, optional att).   BWD == true: dQ (+ delta) from dO, recomputing P from lse.
  using A = AT<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  MELGPT_CLK_BEGIN();
  const int t = threadIdx.x, lane = t & 63, i16 = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z, Tn = p.T, nu = p.n_unmasked;
  const int TP = rup(Tn, 32);
  char* Kt = smem;
  char* Vt = smem + (size_t)TP * A::ROWB;
  int* ctr = (int*)(smem + 2 * (size_t)TP * A::ROWB);
  const T* Kg = (const T*)p.K + (long long)b * Tn * p.ld + h * HS;
  const T* Vg = (const T*)p.V + (long long)b * Tn * p.ld + h * HS;
  load_tiles<T, false, !BWD>(Kt, Kg, p.ld, Vt, Vg, p.ld, Tn, TP, t);  // fwd: V only via transposed reads
  if (t == 0) *ctr = 0;
  __syncthreads();
  const long long bh = (long long)b * p.H + h;
  const int ntiles = (Tn + 15) / 16;
  const float c2 = p.scale * LOG2E;
  const float dsc = DM != DM_NONE ? p.drop_scale : 1.f;
  const DropKeys dkeys = drop_keys(p.seed, p.stream_id, (unsigned)bh, p.drop_thresh);

  auto grab = [&]() {  // next 16-row query tile of this wave (heaviest = latest causal rows first); wave-uniform
    int j = 0;
    if (lane == 0) j = atomicAdd(ctr, 1);
    return __builtin_amdgcn_readfirstlane(j);
  };
  auto fetch = [&](int job, u32x4 (&qf)[A::NKS], u32x4 (&dof)[A::NKS], u32x4 (&ov)[A::NKS], float& lse_q) {
    const int qc = min(16 * (ntiles - 1 - job) + i16, Tn - 1);
    const T* qp = (const T*)p.Q + ((long long)b * Tn + qc) * p.ld + h * HS;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) qf[ks] = *(const u32x4*)(qp + (4 * ks + g) * A::VEC);
    if constexpr (BWD) {
      const T* dp = (const T*)p.dO + ((long long)b * Tn + qc) * p.ldo + h * HS;
      const T* op = (const T*)p.O + ((long long)b * Tn + qc) * p.ldo + h * HS;
#pragma unroll
      for (int ks = 0; ks < A::NKS; ++ks) {
        dof[ks] = *(const u32x4*)(dp + (4 * ks + g) * A::VEC);
        ov[ks] = *(const u32x4*)(op + (4 * ks + g) * A::VEC);
      }
      lse_q = p.lse[bh * Tn + qc];
    }
  };

  int job = grab();
  u32x4 qf[A::NKS], dof[A::NKS], ov[A::NKS];
  float lse_q = 0.f;
  fetch(min(job, ntiles - 1), qf, dof, ov, lse_q);

 while (job < ntiles) {
  // the NEXT tile's rows are requested now and land under this tile's math
  const int njob = grab();
  u32x4 nqf[A::NKS], ndof[A::NKS], nov[A::NKS];
  float nlse = 0.f;
  fetch(min(njob, ntiles - 1), nqf, ndof, nov, nlse);

  const int q0 = 16 * (ntiles - 1 - job);
  const int q = q0 + i16, qc = min(q, Tn - 1);
  const int lim_g = vis_keys(qc, Tn, nu) - 4 * g;                      // key 16 kt + 4 g + r visible <=> 16 kt + r < lim_g
  const int nkt = (vis_keys(min(q0 + 15, Tn - 1), Tn, nu) + 15) / 16;  // key tiles any row of the tile sees
  const int nfull = vis_keys(q0, Tn, nu) / 16;                         // key tiles EVERY row sees whole: no mask test
  const int nst = (nkt + A::TPS - 1) / A::TPS, nst_full = nfull / A::TPS;
  const unsigned cb = (unsigned)q * 128u + (unsigned)g;                // dropout counter of keys 16 kt + 4 g ..+3: cb + 4 kt
  // p = 1/2: the lane's hash words for key blocks kb = 0, 1, 2 (128 keys each; T <= 288); bit 4 (kt & 7) + r = key 16 kt + 4 g + r
  int hblk[3] = {0, 0, 0};
  if constexpr (DM == DM_HALF) {
    const unsigned ch = (unsigned)q * 16u + 4u * (unsigned)g + dkeys.k0;
    hblk[0] = (int)hash32(ch);
    if (nkt > 8) hblk[1] = (int)hash32(ch + 1u);
    if (nkt > 16) hblk[2] = (int)hash32(ch + 2u);
  }
  const f32x4 c2v = splat4(c2);

  auto scores = [&](int kt, bool masked) {  // S^T tile: keys 16 kt + 4 g + r of query q, raw (unscaled) logits
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) mma<T>(acc, frag_row<T, false>(Kt, kt, ks, lane), qf[ks]);
    if (masked) {
      const int lim = lim_g - 16 * kt;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = (r < lim) ? acc[r] : -__builtin_inff();
    }
    return acc;
  };
  auto dropped = [&](f32x4 v, int kt) {
    if constexpr (DM == DM_HALF) {
      const int kb = kt >> 3, sh = 4 * (kt & 7);                       // wave-uniform
      const int hsh = kb == 0 ? hblk[0] : (kb == 1 ? hblk[1] : hblk[2]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        v[r] = __uint_as_float(__float_as_uint(v[r]) & (unsigned)__builtin_amdgcn_sbfe(hsh, sh + r, 1));
      return v;
    } else {
      return drop4<DM>(dkeys, cb + 4u * (unsigned)kt, v);
    }
  };

  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float oscale;
  const char* Xt = BWD ? Kt : Vt;

  if constexpr (!BWD) {
    // pass 1: the row maximum alone.  The logits are NOT kept: pass 2 recomputes them (the MFMA pipe idles under the
    // softmax VALU work anyway), which frees the 72 registers a 17-tile row of scores would pin.
    float m = -__builtin_inff();
    int kt4 = 0;
    for (; kt4 + 4 <= nfull; kt4 += 4) {  // four independent tiles in flight (the compiler does not unroll MFMA loops)
      const f32x4 a0 = scores(kt4, false), a1 = scores(kt4 + 1, false), a2 = scores(kt4 + 2, false),
                  a3 = scores(kt4 + 3, false);
      const float m0 = fmaxf(fmaxf(a0[0], a0[1]), fmaxf(a0[2], a0[3])), m1 = fmaxf(fmaxf(a1[0], a1[1]), fmaxf(a1[2], a1[3]));
      const float m2 = fmaxf(fmaxf(a2[0], a2[1]), fmaxf(a2[2], a2[3])), m3 = fmaxf(fmaxf(a3[0], a3[1]), fmaxf(a3[2], a3[3]));
      m = fmaxf(m, fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)));
    }
    for (int kt = kt4; kt < nfull; ++kt) {
      const f32x4 a = scores(kt, false);
      m = fmaxf(fmaxf(m, a[0]), fmaxf(fmaxf(a[1], a[2]), a[3]));
    }
    for (int kt = max(kt4, nfull); kt < nkt; ++kt) {
      const f32x4 a = scores(kt, true);
      m = fmaxf(fmaxf(m, a[0]), fmaxf(fmaxf(a[1], a[2]), a[3]));
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const f32x4 mcv = splat4(m * c2);
    f32x4 lv = {0.f, 0.f, 0.f, 0.f};
    float* ap = ATT ? p.att + (bh * Tn + qc) * Tn : nullptr;
    // pass 2: e = 2^((s - m) c2), row sum, dropout, O^T += V^T e   (an absent odd partner tile masks to e = 0)
    auto step = [&](int st, bool masked) {
      f32x4 e[2];
#pragma unroll
      for (int tt = 0; tt < A::TPS; ++tt) {
        const int kt = A::TPS * st + tt;
        e[tt] = exp2_4(scores(kt, masked) * c2v - mcv);
        lv += e[tt];
        if constexpr (ATT) {  // unnormalised here, rescaled below once the row sum is known
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * kt + 4 * g + r < Tn) ap[16 * kt + 4 * g + r] = e[tt][r];
        }
        e[tt] = dropped(e[tt], kt);
      }
      const u32x4 bop = pack_operand<T>(e[0], e[A::TPS - 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) mma<T>(o[dt], frag_tr<T, true>(Xt, st, dt, lane), bop);
    };
    int st2 = 0;
    for (; st2 + 2 <= nst_full; st2 += 2) {  // two independent operand steps in flight
      step(st2, false);
      step(st2 + 1, false);
    }
    if (st2 < nst_full) step(st2, false);
    for (int st = nst_full; st < nst; ++st) step(st, true);
    float l = (lv[0] + lv[1]) + (lv[2] + lv[3]);
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    oscale = inv * dsc;
    if (g == 0 && q < Tn) p.lse[bh * Tn + q] = m * p.scale + __logf(l);
    if (ATT && q < Tn) {  // the reference's returned post-softmax, pre-dropout map (only on request)
      const int kend = 16 * A::TPS * nst;
      for (int key = 4 * g; key < Tn; key += 16)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (key + r < Tn) ap[key + r] = key < kend ? ap[key + r] * inv : 0.f;  // beyond this tile's frontier: 0
    }
  } else {
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (A::ES == 2) {
          delta = fmaf(bf16lo(dof[ks][e]), bf16lo(ov[ks][e]), delta);
          delta = fmaf(bf16hi(dof[ks][e]), bf16hi(ov[ks][e]), delta);
        } else {
          delta = fmaf(__uint_as_float(dof[ks][e]), __uint_as_float(ov[ks][e]), delta);
        }
      }
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    if (g == 0 && q < Tn) p.delta[bh * Tn + q] = delta;
    // dS = P (keep dP / (1-p) - delta) scale  =  [P (keep dP - delta (1-p))] * scale / (1-p): the bracket per element,
    // the constant on the dQ tile
    const f32x4 l2v = splat4(lse_q * LOG2E), dlv = splat4(delta / dsc);
    oscale = p.scale * dsc;
    auto step = [&](int st, bool masked) {
      f32x4 ds[2];
#pragma unroll
      for (int tt = 0; tt < A::TPS; ++tt) {
        const int kt = A::TPS * st + tt;
        const f32x4 pr = exp2_4(scores(kt, masked) * c2v - l2v);
        f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < A::NKS; ++ks) mma<T>(dp, frag_row<T, false>(Vt, kt, ks, lane), dof[ks]);
        dp = dropped(dp, kt);
        ds[tt] = pr * (dp - dlv);
      }
      const u32x4 bop = pack_operand<T>(ds[0], ds[A::TPS - 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) mma<T>(o[dt], frag_tr<T, false>(Xt, st, dt, lane), bop);
    };
    for (int st = 0; st < nst_full; ++st) step(st, false);  // (two steps in flight spill here)
    for (int st = nst_full; st < nst; ++st) step(st, true);
  }

  if (q < Tn) {
    T* op = BWD ? (T*)p.dQ + ((long long)b * Tn + q) * p.ldg + h * HS
                : (T*)p.O + ((long long)b * Tn + q) * p.ldo + h * HS;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4<T>(op + 16 * dt + 4 * g, o[dt] * oscale);
  }

  job = njob;
#pragma unroll
  for (int ks = 0; ks < A::NKS; ++ks) {
    qf[ks] = nqf[ks];
    dof[ks] = ndof[ks];
    ov[ks] = nov[ks];
  }
  lse_q = nlse;
 }  // tile loop
  if constexpr (!BWD) MELGPT_CLK_END(clk_attn_fwd);
}

// ============================================================================================ dK / dV
// mma<T>(acc, a, b) computes D[row of a][col of b]:  here rows = queries (Q / dO tile fragments on the A port),
// columns = this wave's 16 keys (K / V rows held in registers on the B port), so every lane owns one key column
// and the S / dP accumulators are directly the B operands of  dV^T = dO^T P_drop  and  dK^T = Q^T dS.
// An accumulator register holds ONE query row here, so the four lanes of a quad (keys 4m..4m+3 of one dropout counter)
// would each hash the same four counters: instead quad lane j hashes row 4g+j once and the quad shares them by DPP.
template <typename T, int DM>
__global__ __launch_bounds__(NTHREADS, sizeof(T) == 2 ? 4 : 2) void attn_dkv_kernel(AttnParams p) {
  using A = AT<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, i16 = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z, Tn = p.T, nu = p.n_unmasked;
  const int TP = rup(Tn, 32);
  char* Qt = smem;
  char* Dt = smem + (size_t)TP * A::ROWB;
  float* lse_s = (float*)(smem + 2 * (size_t)TP * A::ROWB);  // lse * log2(e)
  float* del_s = lse_s + TP;                                  // delta * (1 - p)
  int* ctr = (int*)(del_s + TP);
  const long long bh = (long long)b * p.H + h;
  const float dsc = DM != DM_NONE ? p.drop_scale : 1.f;
  load_tiles<T, false, false>(Qt, (const T*)p.Q + (long long)b * Tn * p.ld + h * HS, p.ld, Dt,
                              (const T*)p.dO + (long long)b * Tn * p.ldo + h * HS, p.ldo, Tn, TP, t);
  for (int j = t; j < TP; j += NTHREADS) {
    lse_s[j] = j < Tn ? p.lse[bh * Tn + j] * LOG2E : 0.f;
    del_s[j] = j < Tn ? p.delta[bh * Tn + j] / dsc : 0.f;
  }
  if (t == 0) *ctr = 0;
  __syncthreads();
  const int ntiles = (Tn + 15) / 16;
  const f32x4 c2v = splat4(p.scale * LOG2E);
  const DropKeys dkeys = drop_keys(p.seed, p.stream_id, (unsigned)bh, p.drop_thresh);
  const int qsh = 8 * (lane & 3);  // (general p) this lane's key is element (key & 3) of its counter: byte qsh of the 8-bit hash

 for (;;) {  // ---- this wave's next 16-key tile (key tile 0 is seen by every query: heaviest first); wave-uniform
  int job = 0;
  if (lane == 0) job = atomicAdd(ctr, 1);
  job = __builtin_amdgcn_readfirstlane(job);
  if (job >= ntiles) break;
  const int key0 = 16 * job, key = key0 + i16, kc = min(key, Tn - 1);
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

  // dropout counter of (query q, keys key&~3 ..+3) is q * 128 + key / 4; quad lane j hashes query 16 qt + 4 g + j
  const unsigned cq = (unsigned)(4 * g + (lane & 3)) * 128u + (unsigned)(key >> 2) + dkeys.k0;
  const unsigned ce = (unsigned)(4 * g) * 128u + (unsigned)(key >> 2);
  // p = 1/2: (query q, this key) is bit 4 ((key >> 4) & 7) + (key & 3) of the hash of q * 16 + 4 ((key & 15) >> 2) + (key >> 7)
  const unsigned chq = (unsigned)(4 * g + (lane & 3)) * 16u + (unsigned)(((key & 15) >> 2) * 4 + (key >> 7)) + dkeys.k0;
  const int hbit = 4 * ((key >> 4) & 7) + (key & 3);

  auto step = [&](int st, bool masked) {
    f32x4 pd[2], ds[2];
#pragma unroll
    for (int tt = 0; tt < A::TPS; ++tt) {
      const int qt = A::TPS * st + tt, qb = 16 * qt;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < A::NKS; ++ks) mma<T>(acc, frag_row<T, false>(Qt, qt, ks, lane), kf[ks]);
#pragma unroll
      for (int ks = 0; ks < A::NKS; ++ks) mma<T>(dp, frag_row<T, false>(Dt, qt, ks, lane), vf[ks]);
      const f32x4 l4 = *(const f32x4*)(lse_s + qb + 4 * g), d4 = *(const f32x4*)(del_s + qb + 4 * g);
      f32x4 pr = exp2_4(acc * c2v - l4);
      if (masked) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = qb + 4 * g + r;
          pr[r] = (q < Tn && allowed(q, key, Tn, nu)) ? pr[r] : 0.f;
        }
      }
      f32x4 pk = pr;
      if constexpr (DM == DM_HALF) {
        const int hq = (int)hash32(chq + (unsigned)qb * 16u);
        const int hr[4] = {quad_lane<0>(hq), quad_lane<1>(hq), quad_lane<2>(hq), quad_lane<3>(hq)};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned km = (unsigned)__builtin_amdgcn_sbfe(hr[r], hbit, 1);  // all ones = kept
          pk[r] = __uint_as_float(__float_as_uint(pr[r]) & km);
          dp[r] = __uint_as_float(__float_as_uint(dp[r]) & km);
        }
      } else if constexpr (DM == DM_ANY) {
        if (dkeys.b8) {
          const int hq = (int)hash32(cq + (unsigned)qb * 128u);
          const int hr[4] = {quad_lane<0>(hq), quad_lane<1>(hq), quad_lane<2>(hq), quad_lane<3>(hq)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool keep = __builtin_amdgcn_ubfe((unsigned)hr[r], (unsigned)qsh, 8u) >= dkeys.t;
            pk[r] = keep ? pr[r] : 0.f;
            dp[r] = keep ? dp[r] : 0.f;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            bool k4[4];
            drop_keep4(dkeys, ce + (unsigned)(qb + r) * 128u, k4);
            const bool keep = (key & 2) ? ((key & 1) ? k4[3] : k4[2]) : ((key & 1) ? k4[1] : k4[0]);
            pk[r] = keep ? pr[r] : 0.f;
            dp[r] = keep ? dp[r] : 0.f;
          }
        }
      }
      pd[tt] = pk;
      ds[tt] = pr * (dp - d4);
    }
    const u32x4 bp = pack_operand<T>(pd[0], pd[A::TPS - 1]);
    const u32x4 bs = pack_operand<T>(ds[0], ds[A::TPS - 1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      mma<T>(dv[dt], frag_tr<T, false>(Dt, st, dt, lane), bp);
      mma<T>(dk[dt], frag_tr<T, false>(Qt, st, dt, lane), bs);
    }
  };
  // query tiles that can see these keys: from the diagonal (or from 0 inside the unmasked block).  A tile is seen
  // whole - no mask test - when it lies below the diagonal tile and holds no padded row or key.
  const int q_start = (nu > key0) ? 0 : key0;
  const int st0 = (q_start / 16) / A::TPS;
  const int st1 = (ntiles + A::TPS - 1) / A::TPS;
  const int stf0 = key0 + 16 <= Tn ? min(st1, (job + 1 + A::TPS - 1) / A::TPS) : st1;  // first whole step ..
  const int stf1 = max(stf0, (Tn / 16) / A::TPS);                                        // .. and one past the last
  for (int st = st0; st < stf0; ++st) step(st, true);
  for (int st = stf0; st < stf1; ++st) step(st, false);
  for (int st = stf1; st < st1; ++st) step(st, true);

  if (key < Tn) {
    T* kp = (T*)p.dK + ((long long)b * Tn + key) * p.ldg + h * HS;
    T* vp = (T*)p.dV + ((long long)b * Tn + key) * p.ldg + h * HS;
    const float ksc = p.scale * dsc;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      store4<T>(kp + 16 * dt + 4 * g, dk[dt] * ksc);
      store4<T>(vp + 16 * dt + 4 * g, dv[dt] * dsc);
    }
  }
 }  // tile loop
}

// ====================================================================== forward on 32-row query tiles (16-bit lane)
// The structure DESIGN section 4 specified in round 2 (built in round 5): v_mfma_f32_32x32x16, a wave owns a 32-query tile
// and keeps the WHOLE row of logits in registers (9 key tiles x 16 accumulator registers at T <= 288), so there is no
// pass 1 and nothing is recomputed: S^T = K Q^T (4 MFMAs per 32 x 32 block, the query on the lane), row maximum in
// registers (v_max3 + ONE cross-half exchange), e = 2^(s c - m c), row sum, dropout, and the converted accumulators ARE the
// B operand of O^T = V^T P^T (guide: "an accumulator tile as the next MFMA's operand" - element j of lane half hh of k-step s
// is key 16 s + 8 (j >> 2) + 4 hh + (j & 3) of the tile, so V^T's fragment takes its two 4-key blocks 8 keys apart, by
// ds_read_b64_tr_b16 from the same swizzled V image the 16-row kernel uses).  Against the 16-row kernel per 1 024
// probabilities: 8 MFMAs of 32 issue-blocking cycles instead of 24 of 16 (no recompute), half the K / V^T fragment bytes
// from LDS, and half the per-tile fixed cost (9 tiles instead of 17: Q fetch, exchanges, 1 / l, lse, stores).
// 256 threads = 4 waves = one (batch, head); two workgroups per CU (2 x 72 KB of LDS, 256 registers per lane).
// The dropout masks are the SAME bits as everywhere else (header): a lane's keys of a 32-key tile are 8 j + 4 hh + r,
// i.e. gk = 2 (j & 1) + hh of 16-key tile 2 kt + (j >> 1): p = 1/2 takes two hash words per 128 keys, all 32 bits of each used.
template <int DM>
__global__ __launch_bounds__(256, 2) void attn_fwd32_kernel(AttnParams p) {
  using T = bf16_t;
  constexpr int MAXKT32 = MAXT / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, r32 = lane & 31, hh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z, Tn = p.T, nu = p.n_unmasked;
  const int TP = rup(Tn, 32);
  char* Kt = smem;
  char* Vt = smem + (size_t)TP * 128;
  int* ctr = (int*)(smem + 2 * (size_t)TP * 128);
#ifdef MELGPT_CLOCK_STAMPS
  const bool ph_on = blockIdx.y == 5 && blockIdx.z == 3 && lane == 0;
  unsigned long long* ph = clk_attn32_ph + (t >> 6) * 64;
  int ph_n = 0;
#endif
  MELGPT_CLK_BEGIN();
  PH32_STAMP();   // [0] entry
  const int ntiles = (Tn + 31) / 32;
  // The four waves' FIRST query tiles are fixed (jobs 0-3, the heaviest) and their Q rows requested before K / V: behind
  // the staging barrier the first tile waited a whole memory round trip for them (3 k of a workgroup's ~23 k cycles,
  // tools/lab/clock_lab.py stamps); the work counter hands out jobs from 4 on.
  int job = __builtin_amdgcn_readfirstlane(t >> 6);   // (wave-uniform: in a VGPR every branch on a tile count became an EXEC-mask region)
  u32x4 qf[4];
  {
    const int qc = min(32 * (ntiles - 1 - min(job, ntiles - 1)) + r32, Tn - 1);
    const T* qp = (const T*)p.Q + ((long long)b * Tn + qc) * p.ld + h * HS;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const u32x4*)(qp + 16 * ks + 8 * hh);
  }
  const T* Kg = (const T*)p.K + (long long)b * Tn * p.ld + h * HS;
  const T* Vg = (const T*)p.V + (long long)b * Tn * p.ld + h * HS;
  load_tiles<T, false, true, 256>(Kt, Kg, p.ld, Vt, Vg, p.ld, Tn, TP, t);
  if (t == 0) *ctr = 4;
  __syncthreads();
  PH32_STAMP();   // [1] K / V staged
  const long long bh = (long long)b * p.H + h;
  const float c2 = p.scale * LOG2E;
  const float dsc = DM != DM_NONE ? p.drop_scale : 1.f;
  const DropKeys dkeys = drop_keys(p.seed, p.stream_id, (unsigned)bh, p.drop_thresh);
  // this lane's addresses inside a 32-key tile: K rows by ds_read_b128 (row r32, chunk 2 ks + hh), V^T by transposed reads
  // (16-lane group: row q4 of the 4-key block, 4 columns from 16 g1 + 4 p4 of the 32-column block db)
  const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;

  auto grab = [&]() {
    int j = 0;
    if (lane == 0) j = atomicAdd(ctr, 1);
    return __builtin_amdgcn_readfirstlane(j);
  };
  auto fetch = [&](int job, u32x4 (&qf)[4]) {
    const int qc = min(32 * (ntiles - 1 - job) + r32, Tn - 1);
    const T* qp = (const T*)p.Q + ((long long)b * Tn + qc) * p.ld + h * HS;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const u32x4*)(qp + 16 * ks + 8 * hh);   // B[k = 8 hh + j][query]: d = 16 ks + 8 hh + j
  };

  while (job < ntiles) {
    PH32_STAMP();   // per tile: start
    const int q0 = 32 * (ntiles - 1 - job);
    const int q = q0 + r32, qc = min(q, Tn - 1);
    const int lim = vis_keys(qc, Tn, nu) - 4 * hh;                        // key 32 kt + 8 j + 4 hh + r visible <=> 32 kt + 8 j + r < lim
    const int nkt = (vis_keys(min(q0 + 31, Tn - 1), Tn, nu) + 31) / 32;   // key tiles any row of the tile sees (wave-uniform)
    const int nfull = vis_keys(q0, Tn, nu) / 32;                          // key tiles every row sees whole

    // ---- S^T tiles: keys on the accumulator rows, the query on the lane.  Key tiles go in PAIRS through one basic block
    // (a wave-uniform branch per pair, the masks afterwards): the two 4-MFMA chains and their eight fragment reads are
    // independent, so the compiler interleaves them - tile by tile every chain waited for its own LDS round trip
    PH32_STAMP();   // tile bounds known
    f32x16 S[MAXKT32];
    auto s_chain = [&](int kt) -> f32x16 {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4 kf = *(const u32x4*)(Kt + offK<T>(32 * kt + r32, 2 * ks + hh));
        acc = MELGPT_MFMA_32x32x16(kf, qf[ks], acc);
      }
      return acc;
    };
#pragma unroll
    for (int kp = 0; kp < (MAXKT32 + 1) / 2; ++kp) {
      const int k0 = 2 * kp, k1 = 2 * kp + 1;
      if (k1 < MAXKT32 && k1 < nkt) {
        S[k0] = s_chain(k0);
        S[k1 < MAXKT32 ? k1 : k0] = s_chain(k1);
      } else if (k0 < nkt) {
        S[k0] = s_chain(k0);
      }
#ifdef MELGPT_CLOCK_STAMPS
      if (kp == 0) {
        asm volatile("" ::"v"(S[0]));
        PH32_STAMP();   // first pair of key tiles issued
      }
#endif
    }
#pragma unroll
    for (int kt = 0; kt < MAXKT32; ++kt) {
      if (kt >= nfull && kt < nkt) {   // (the tiles the causal frontier or the end of the sequence cuts: one or two per query tile)
        const int lk = lim - 32 * kt;
#pragma unroll
        for (int i = 0; i < 16; ++i) S[kt][i] = (8 * (i >> 2) + (i & 3) < lk) ? S[kt][i] : -__builtin_inff();
      }
    }
#ifdef MELGPT_CLOCK_STAMPS
    asm volatile("" ::"v"(S[0]));
#endif
    PH32_STAMP();   // S tiles issued
    // the logits are in registers and the Q fragments dead: the NEXT tile's rows are requested into them now and land
    // under the softmax / P V phase (a second register set for them spilled)
    const int njob = grab();
    fetch(min(njob, ntiles - 1), qf);
    // ---- row maximum (every query sees key 0: never -inf)
    float m = -__builtin_inff();
#pragma unroll
    for (int kt = 0; kt < MAXKT32; ++kt) {
      if (kt < nkt) {
#pragma unroll
        for (int i = 0; i < 16; i += 2)   // (asm: fmaxf on MFMA results makes hipcc canonicalise every operand with a v_max of its own)
          asm("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(S[kt][i]), "v"(S[kt][i + 1]));
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mc = m * c2;
#ifdef MELGPT_CLOCK_STAMPS
    asm volatile("" ::"v"(mc));
#endif
    PH32_STAMP();   // row maximum known
    // p = 1/2: hash word a of key block kb (128 keys, drawn when the loop below enters it) serves keys
    // 128 kb + 16 tt + 4 (2 a + hh) + r, bit 4 tt + r
    int hw[2] = {0, 0};
    const unsigned ch = (unsigned)q * 16u + 4u * (unsigned)hh + dkeys.k0;
    // ---- e = 2^(s c - m c), row sum, dropout, O^T += V^T e
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float ls[4] = {0.f, 0.f, 0.f, 0.f};   // four partial row sums (independent chains)
    // probabilities of key tile kt as the two B operands (k-steps) of the P V product
    auto probs = [&](int kt, u32x4 (&pb)[2]) {
      if constexpr (DM == DM_HALF) {
        if ((kt & 3) == 0) {
          hw[0] = (int)hash32(ch + (unsigned)(kt >> 2));
          hw[1] = (int)hash32(ch + 8u + (unsigned)(kt >> 2));
        }
      }
      float e[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        e[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kt][i], c2, -mc));
        asm("v_add_f32 %0, %0, %1" : "+v"(ls[i & 3]) : "v"(e[i]));   // (asm: hipcc SLP-packs plain adds into v_pk_add_f32, slower beside MFMAs)
      }
      if constexpr (DM == DM_HALF) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int j = i >> 2, r = i & 3;
          // keep-mask = the hash bit sign-extended (asm: written with the builtin, hipcc turns sbfe + and into and + cmp + cndmask)
          unsigned km;
          asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(km) : "v"(hw[j & 1]), "n"(8 * (kt & 3) + 4 * (j >> 1) + r));
          e[i] = __uint_as_float(__float_as_uint(e[i]) & km);
        }
      } else if constexpr (DM == DM_ANY) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bool keep[4];
          drop_keep4(dkeys, (unsigned)q * 128u + (unsigned)(8 * kt + 2 * j + hh), keep);
#pragma unroll
          for (int r = 0; r < 4; ++r) e[4 * j + r] = keep[r] ? e[4 * j + r] : 0.f;
        }
      }
#pragma unroll
      for (int sx = 0; sx < 2; ++sx)
        pb[sx] = u32x4{pack_bf16x2(e[8 * sx + 0], e[8 * sx + 1]), pack_bf16x2(e[8 * sx + 2], e[8 * sx + 3]),
                       pack_bf16x2(e[8 * sx + 4], e[8 * sx + 5]), pack_bf16x2(e[8 * sx + 6], e[8 * sx + 7])};
    };
    auto pv = [&](int kt, const u32x4 (&pb)[2]) {
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) {
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          // A[row d = 32 db + r32][k = 8 hh + j] = V[key 32 kt + 16 sx + 8 (j >> 2) + 4 hh + (j & 3)][d]
          const int row = 32 * kt + 16 * sx + 4 * hh + q4, c = 4 * db + 2 * g1 + (p4 >> 1);
          const char* a0 = Vt + offV<T>(row, c) + 8 * (p4 & 1);
          const char* a1 = Vt + offV<T>(row + 8, c) + 8 * (p4 & 1);
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1));
          const s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[db] = MELGPT_MFMA_32x32x16(__builtin_bit_cast(u32x4, f), pb[sx], o[db]);
        }
      }
    };
#pragma unroll
    for (int kt = 0; kt < MAXKT32; ++kt) {   // (tile by tile here: a pair's 32 probabilities + two operand sets do not fit beside the logits)
      if (kt < nkt) {
        u32x4 pb[2];
        probs(kt, pb);
        pv(kt, pb);
      }
    }
#ifdef MELGPT_CLOCK_STAMPS
    asm volatile("" ::"v"(o[0]), "v"(o[1]));
#endif
    PH32_STAMP();   // softmax + P V done
    const float lsum = (ls[0] + ls[1]) + (ls[2] + ls[3]);
    const float l = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / l;
    const float oscale = inv * dsc;
    // The next tile's Q rows (requested a whole softmax / P V phase ago) are taken HERE, in front of this tile's stores:
    // vmcnt counts stores too and in order, so a wait for them placed at the top of the next tile (where the compiler puts
    // it) also waited for the nine stores below - ~1.9 k cycles per tile, whatever its size (clock_lab.py stamps).
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));
    if (q < Tn) {
      if (hh == 0) p.lse[bh * Tn + q] = m * p.scale + __logf(l);
      T* op = (T*)p.O + ((long long)b * Tn + q) * p.ldo + h * HS;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)   // registers 4 jj .. 4 jj + 3: d = 32 db + 8 jj + 4 hh + 0..3
          store4<T>(op + 32 * db + 8 * jj + 4 * hh,
                    f32x4{o[db][4 * jj], o[db][4 * jj + 1], o[db][4 * jj + 2], o[db][4 * jj + 3]} * oscale);
    }
    job = njob;
#ifdef MELGPT_CLOCK_STAMPS
    if (ph_on && ph_n < 62) ph[ph_n++] = (unsigned long long)nkt;   // (key tiles of the tile just done)
#endif
  }
  PH32_STAMP();   // this wave is done
#ifdef MELGPT_CLOCK_STAMPS
  if (ph_on) ph[63] = (unsigned long long)ph_n;
#endif
  MELGPT_CLK_END(clk_attn_fwd);
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

int drop_mode(const AttnParams& p) {
  return p.drop_scale == 0.f ? DM_NONE : p.drop_thresh == 0x80000000u ? DM_HALF : DM_ANY;
}

template <typename T, bool BWD, int DM, bool ATT>
int launch_q_att(const AttnParams& p, hipStream_t s) {
  if (set_lds(attn_q_kernel<T, BWD, DM, ATT>, lds_bytes<T>(MAXT, false)) != MELGPT_OK) return MELGPT_ERR_LAUNCH;
  hipLaunchKernelGGL((attn_q_kernel<T, BWD, DM, ATT>), dim3(1, p.H, p.B), dim3(NTHREADS), lds_bytes<T>(p.T, false), s, p);
  return MELGPT_OK;
}
template <typename T, bool BWD, int DM>
int launch_q(const AttnParams& p, hipStream_t s) {
  if constexpr (!BWD)
    if (p.att) return launch_q_att<T, BWD, DM, true>(p, s);
  return launch_q_att<T, BWD, DM, false>(p, s);
}
template <typename T, int DM>
int launch_dkv(const AttnParams& p, hipStream_t s) {
  if (set_lds(attn_dkv_kernel<T, DM>, lds_bytes<T>(MAXT, true)) != MELGPT_OK) return MELGPT_ERR_LAUNCH;
  hipLaunchKernelGGL((attn_dkv_kernel<T, DM>), dim3(1, p.H, p.B), dim3(NTHREADS), lds_bytes<T>(p.T, true), s, p);
  return MELGPT_OK;
}
template <int DM>
int launch_fwd32(const AttnParams& p, hipStream_t s) {
  const size_t lds = 2 * (size_t)((p.T + 31) / 32 * 32) * 128 + 16;
  static bool attr = false;
  if (!attr) {
    if (set_lds(attn_fwd32_kernel<DM>, 2 * (size_t)MAXT * 128 + 16) != MELGPT_OK) return MELGPT_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((attn_fwd32_kernel<DM>), dim3(1, p.H, p.B), dim3(256), lds, s, p);
  return MELGPT_OK;
}

template <typename T, bool BWD>
int launch_q_mode(const AttnParams& p, hipStream_t s) {
  switch (drop_mode(p)) {
    case DM_NONE: return launch_q<T, BWD, DM_NONE>(p, s);
    case DM_HALF: return launch_q<T, BWD, DM_HALF>(p, s);
    default: return launch_q<T, BWD, DM_ANY>(p, s);
  }
}
template <typename T>
int launch_dkv_mode(const AttnParams& p, hipStream_t s) {
  switch (drop_mode(p)) {
    case DM_NONE: return launch_dkv<T, DM_NONE>(p, s);
    case DM_HALF: return launch_dkv<T, DM_HALF>(p, s);
    default: return launch_dkv<T, DM_ANY>(p, s);
  }
}

// ======================================================================================= single-pass backward
// dQ, dK and dV of one (batch, head) from ONE evaluation of S, P, dP and dS (16-bit lane, causal mask: n_unmasked == 0).
// The two kernels above each recompute the probabilities (exp2, dropout hash, mask: the VALU work these kernels are
// bound by) and each stage two 34 KB tiles; here
//   phase 1 = attn_dkv_kernel's loop (Q and dO in LDS, a wave owns 16 keys, dK^T / dV^T in its accumulators) which ALSO
//             leaves every dS tile in LDS - bf16, key-major 16 x 16 tiles of 512 bytes, one per causal (key tile, query
//             tile) pair: 153 pairs = 78 KB at T = 265 - written straight from the packed dK operand (one ds_write_b64);
//   phase 2 = dQ^T[d][q] = sum_keys K^T[d][key] dS^T[key][q]: K re-staged over Q's tile (L2-hot), the B operand read
//             back with ds_read_b64_tr_b16 (a 4-key x 16-query block per 16 lanes = exactly the fragment), 4 MFMAs per
//             32 keys, no VALU; query tiles pulled from a second work counter.
// delta = rowsum(dO * O) is taken while Q / dO / O are staged (the same 8-lanes-per-row loads).  152 KB of LDS: one
// workgroup per CU (the two-kernel path runs two), 256 registers per lane.  Falls back to the two kernels when the dS tiles
// do not fit (T > 272) or the mask is not plain causal.  Measured: profiles/r03_attn_lab.md.
__device__ __forceinline__ int ds_pair(int kt, int qt, int nt) { return kt * nt - (kt * (kt - 1)) / 2 + (qt - kt); }

template <int DM>
__global__ __launch_bounds__(NTHREADS, 2) void attn_bwd1_kernel(AttnParams p) {
  typedef bf16_t T;
  using A = AT<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, i16 = lane & 15, g = lane >> 4;
  const int Tn = p.T;
  const int TP = rup(Tn, 32), ntiles = (Tn + 15) / 16;
  char* Qt = smem;                                   // Q (phase 1), K (phase 2)
  char* Dt = smem + (size_t)TP * A::ROWB;            // dO
  float* lse_s = (float*)(smem + 2 * (size_t)TP * A::ROWB);  // lse * log2(e)
  float* del_s = lse_s + TP;                                  // delta * (1 - p)
  int* ctr = (int*)(del_s + TP);                              // [0] phase 1, [1] phase 2
  char* dsb = (char*)(ctr + 4);                               // dS tiles (16-byte aligned: TP is a multiple of 32)
  MELGPT_CLK_BEGIN();
  const float dsc = DM != DM_NONE ? p.drop_scale : 1.f;
  const f32x4 c2v = splat4(p.scale * LOG2E);
  const int qsh = 8 * (lane & 3);
  // PERSISTENT: gridDim.x workgroups (one per CU: 152 KB of LDS) walk the (batch, head) items.  The workgroups run in
  // lockstep (same work each), so un-overlapped staging is the whole chip pulling 26 MB at once and then leaving HBM
  // idle: 10-15 k of an item's 53 k cycles (stamps: profiles/r03_attn_lab.md).  The NEXT item's Q / dO / O rows are
  // therefore requested into 60 spare registers under the current item's phase 1.
  constexpr int NIT = (MAXT * 8 + NTHREADS - 1) / NTHREADS;
  const int nitems = p.B * p.H;
  u32x4 vq[NIT], vd[NIT], vo[NIT];
  float vl = 0.f;
  // (32-bit element offsets from wave-uniform bases: 64-bit per-lane addresses for the 15 requests cost 30 registers and
  // pushed the staging code into scratch - whose reloads sit in the same in-order queue as the requests.)
  auto request = [&](int it) {  // 8 lanes per 128-byte row: every request a full line; rows past T re-read the last one
    const int bb = it / p.H, hh = it - bb * p.H;
    const T* Qg = (const T*)p.Q + (long long)bb * Tn * p.ld + hh * HS;
    const T* Dg = (const T*)p.dO + (long long)bb * Tn * p.ldo + hh * HS;
    const T* Og = (const T*)p.O + (long long)bb * Tn * p.ldo + hh * HS;
    const unsigned ld = (unsigned)p.ld, ldo = (unsigned)p.ldo;
    unsigned tl = t;
    asm volatile("" : "+v"(tl));  // (offsets recomputed per item: hoisted out of the item loop they hold 15 registers)
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const unsigned q = tl + i * NTHREADS, row = min(q >> 3, (unsigned)Tn - 1u), c8 = (q & 7u) * 8u;
      vq[i] = *(const u32x4*)(Qg + (row * ld + c8));
      vd[i] = *(const u32x4*)(Dg + (row * ldo + c8));
      vo[i] = *(const u32x4*)(Og + (row * ldo + c8));
    }
    vl = (p.lse + (long long)it * Tn)[min(tl, (unsigned)Tn - 1u)];
  };
  request(blockIdx.x);
 for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
#define LAB_STAMP(k) do { } while (0)
  LAB_STAMP(0);
  const int b = item / p.H, h = item - b * p.H;
  const long long bh = item;
  const T* Kg = (const T*)p.K + (long long)b * Tn * p.ld + h * HS;
  // ---- Q and dO into LDS; delta = rowsum(dO * O) on the way
  int ts = t;
  asm volatile("" : "+v"(ts));  // (as in request(): LDS offsets recomputed per item)
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int q = ts + i * NTHREADS, row = q >> 3, c = q & 7;
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d = fmaf(bf16lo(vd[i][e]), bf16lo(vo[i][e]), d);
      d = fmaf(bf16hi(vd[i][e]), bf16hi(vo[i][e]), d);
    }
    d += dpp_move<0xB1>(d);
    d += dpp_move<0x4E>(d);
    d += dpp_move<0x141>(d);
    if (q < TP * 8) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      *(u32x4*)(Qt + offK<T>(row, c)) = row < Tn ? vq[i] : z;
      *(u32x4*)(Dt + offK<T>(row, c)) = row < Tn ? vd[i] : z;
      if (c == 0) {
        del_s[row] = row < Tn ? d / dsc : 0.f;
        if (row < Tn) (p.delta + bh * Tn)[(unsigned)row] = d;
      }
    }
  }
  if (t < TP) lse_s[t] = t < Tn ? vl * LOG2E : 0.f;  // (TP <= 288 < the 512 threads)
  if (t < 2) ctr[t] = 0;
  __syncthreads();
  LAB_STAMP(1);
  const DropKeys dkeys = drop_keys(p.seed, p.stream_id, (unsigned)bh, p.drop_thresh);

  // ---- phase 1: this wave's next 16-key tile (key tile 0 is seen by every query: heaviest first); wave-uniform
  //      A wave claims its NEXT tile when it starts one and requests that tile's K / V fragments (global loads, ~2-4 k
  //      cycles under load) under the current tile's steps.
  auto claim = [&]() {
    int j = 0;
    if (lane == 0) j = atomicAdd(ctr, 1);
    return __builtin_amdgcn_readfirstlane(j);
  };
  auto load_kv = [&](int jb, u32x4* kf_, u32x4* vf_) {
    const unsigned kc = (unsigned)min(16 * min(jb, ntiles - 1) + i16, Tn - 1);
    const T* kp = Kg + kc * (unsigned)p.ld;
    const T* vp = (const T*)p.V + (long long)b * Tn * p.ld + h * HS + kc * (unsigned)p.ld;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      kf_[ks] = *(const u32x4*)(kp + (4 * ks + g) * A::VEC);
      vf_[ks] = *(const u32x4*)(vp + (4 * ks + g) * A::VEC);
    }
  };
  u32x4 kf[A::NKS], vf[A::NKS];
  int job = claim();
  load_kv(job, kf, vf);
  // Loads return in order, and the compiler places the wait for a load at its first use - for these fragments INSIDE
  // the step loop, where it assumes the worst of all paths: with the next item's rows requested ahead of or behind them
  // the first step waited for all 16 requests (the prefetch bought 4 us of a possible 40).  So: wait for the first
  // tile's fragments here, by hand, and only then put the rows in flight.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
  request(min(item + (int)gridDim.x, nitems - 1));  // (past the end: a redundant reload instead of a branch)
  while (job < ntiles) {
    const int njob = claim();
    u32x4 nkf[A::NKS], nvf[A::NKS];
    load_kv(njob, nkf, nvf);  // (past the last tile: the last tile's again, unused)
    const int key0 = 16 * job, key = key0 + i16;
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned cq = (unsigned)(4 * g + (lane & 3)) * 128u + (unsigned)(key >> 2) + dkeys.k0;
    const unsigned ce = (unsigned)(4 * g) * 128u + (unsigned)(key >> 2);
    const unsigned chq = (unsigned)(4 * g + (lane & 3)) * 16u + (unsigned)(((key & 15) >> 2) * 4 + (key >> 7)) + dkeys.k0;
    const int hbit = 4 * ((key >> 4) & 7) + (key & 3);
    char* ds_row = dsb + (size_t)ds_pair(job, job, ntiles) * 512 + i16 * 32 + g * 8;  // pair (job, qt): + (qt - job) * 512

    auto step = [&](int st, bool masked) {
      f32x4 pd[2], ds[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int qt = 2 * st + tt, qb = 16 * qt;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < A::NKS; ++ks) mma<T>(acc, frag_row<T, false>(Qt, qt, ks, lane), kf[ks]);
#pragma unroll
        for (int ks = 0; ks < A::NKS; ++ks) mma<T>(dp, frag_row<T, false>(Dt, qt, ks, lane), vf[ks]);
        const f32x4 l4 = *(const f32x4*)(lse_s + qb + 4 * g), d4 = *(const f32x4*)(del_s + qb + 4 * g);
        f32x4 pr = exp2_4(acc * c2v - l4);
        if (masked) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = qb + 4 * g + r;
            pr[r] = (q < Tn && key < Tn && key <= q) ? pr[r] : 0.f;
          }
        }
        f32x4 pk = pr;
        if constexpr (DM == DM_HALF) {
          const int hq = (int)hash32(chq + (unsigned)qb * 16u);
          const int hr[4] = {quad_lane<0>(hq), quad_lane<1>(hq), quad_lane<2>(hq), quad_lane<3>(hq)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned km = (unsigned)__builtin_amdgcn_sbfe(hr[r], hbit, 1);  // all ones = kept
            pk[r] = __uint_as_float(__float_as_uint(pr[r]) & km);
            dp[r] = __uint_as_float(__float_as_uint(dp[r]) & km);
          }
        } else if constexpr (DM == DM_ANY) {
          if (dkeys.b8) {
            const int hq = (int)hash32(cq + (unsigned)qb * 128u);
            const int hr[4] = {quad_lane<0>(hq), quad_lane<1>(hq), quad_lane<2>(hq), quad_lane<3>(hq)};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool keep = __builtin_amdgcn_ubfe((unsigned)hr[r], (unsigned)qsh, 8u) >= dkeys.t;
              pk[r] = keep ? pr[r] : 0.f;
              dp[r] = keep ? dp[r] : 0.f;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              bool k4[4];
              drop_keep4(dkeys, ce + (unsigned)(qb + r) * 128u, k4);
              const bool keep = (key & 2) ? ((key & 1) ? k4[3] : k4[2]) : ((key & 1) ? k4[1] : k4[0]);
              pk[r] = keep ? pr[r] : 0.f;
              dp[r] = keep ? dp[r] : 0.f;
            }
          }
        }
        pd[tt] = pk;
        ds[tt] = pr * (dp - d4);
      }
      const u32x4 bp = pack_operand<T>(pd[0], pd[1]);
      const u32x4 bs = pack_operand<T>(ds[0], ds[1]);
      // dS tiles of this step for phase 2: lane = key, its 4 consecutive queries are 8 contiguous bytes of the key's row
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int qt = 2 * st + tt;  // (wave-uniform)
        if (qt >= job && qt < ntiles) *(u32x2*)(ds_row + (qt - job) * 512) = u32x2{bs[2 * tt], bs[2 * tt + 1]};
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        mma<T>(dv[dt], frag_tr<T, false>(Dt, st, dt, lane), bp);
        mma<T>(dk[dt], frag_tr<T, false>(Qt, st, dt, lane), bs);
      }
    };
    // query tiles that can see these keys: from the diagonal.  A tile is seen whole - no mask test - when it lies
    // below the diagonal tile and holds no padded row or key.
    const int st0 = job / 2;
    const int st1 = (ntiles + 1) / 2;
    const int stf0 = key0 + 16 <= Tn ? min(st1, (job + 2) / 2) : st1;  // first whole step ..
    const int stf1 = max(stf0, (Tn / 16) / 2);                          // .. and one past the last
    for (int st = st0; st < st1; ++st) {
      step(st, st < stf0 || st >= stf1);
    }

    if (key < Tn) {
      T* kp = (T*)p.dK + ((long long)b * Tn + key) * p.ldg + h * HS;
      T* vp = (T*)p.dV + ((long long)b * Tn + key) * p.ldg + h * HS;
      const float ksc = p.scale * dsc;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        store4<T>(kp + 16 * dt + 4 * g, dk[dt] * ksc);
        store4<T>(vp + 16 * dt + 4 * g, dv[dt] * dsc);
      }
    }
    job = njob;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      kf[ks] = nkf[ks];
      vf[ks] = nvf[ks];
    }
  }
  LAB_STAMP(2);
  // ---- phase 2: K over Q's tile (requested before the barrier: a wave that is done early waits there anyway), then
  //      dQ^T = K^T dS^T per 16-query tile
  {
    u32x4 vk[NIT];
    unsigned tk = t;
    asm volatile("" : "+v"(tk));
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const unsigned q = tk + i * NTHREADS, row = min(q >> 3, (unsigned)Tn - 1u), c8 = (q & 7u) * 8u;
      vk[i] = *(const u32x4*)(Kg + (row * (unsigned)p.ld + c8));
    }
    __syncthreads();  // every dS tile is in LDS; nobody reads Q any more
    LAB_STAMP(3);
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int q = (int)tk + i * NTHREADS, row = q >> 3, c = q & 7;
      const u32x4 z = {0u, 0u, 0u, 0u};
      if (q < TP * 8) *(u32x4*)(Qt + offK<T>(row, c)) = row < Tn ? vk[i] : z;
    }
  }
  __syncthreads();
  LAB_STAMP(4);
  const float oscale = p.scale * dsc;
  for (;;) {
    int job = 0;
    if (lane == 0) job = atomicAdd(ctr + 1, 1);
    job = __builtin_amdgcn_readfirstlane(job);
    if (job >= ntiles) break;
    const int qt = ntiles - 1 - job;  // latest query tiles (most key tiles) first
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tr_off = (4 * g + (i16 >> 2)) * 32 + (i16 & 3) * 8;  // this lane's address inside a 4-key x 16-query block
    for (int st = 0; 2 * st <= qt; ++st) {
      const int kt0 = 2 * st, kt1 = 2 * st + 1;
      const char* a0 = dsb + (size_t)ds_pair(kt0, qt, ntiles) * 512 + tr_off;
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a0));
      s16x4 hi = {0, 0, 0, 0};
      if (kt1 <= qt) {  // (wave-uniform) the odd key tile of the pair lies beyond the diagonal otherwise: dS = 0
        const char* a1 = dsb + (size_t)ds_pair(kt1, qt, ntiles) * 512 + tr_off;
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1));
      }
      const s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      const u32x4 bop = __builtin_bit_cast(u32x4, f);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) mma<T>(o[dt], frag_tr<T, false>(Qt, st, dt, lane), bop);
    }
    const int q = 16 * qt + i16;
    if (q < Tn) {
      T* op = (T*)p.dQ + ((long long)b * Tn + q) * p.ldg + h * HS;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store4<T>(op + 16 * dt + 4 * g, o[dt] * oscale);
    }
  }
  LAB_STAMP(5);
  __syncthreads();  // the next item's rows replace K / dO / the statistics / the counters
  LAB_STAMP(6);
 }  // item loop
  MELGPT_CLK_END(clk_attn_bwd);
}

static size_t bwd1_lds_bytes(int Tn) {
  const int TP = (Tn + 31) / 32 * 32, nt = (Tn + 15) / 16;
  return 2 * (size_t)TP * 128 + 2 * (size_t)TP * 4 + 16 + (size_t)(nt * (nt + 1) / 2) * 512;
}

template <int DM>
int launch_bwd1(const AttnParams& p, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    if (set_lds(attn_bwd1_kernel<DM>, 160 * 1024) != MELGPT_OK) return MELGPT_ERR_LAUNCH;
    attr = true;
  }
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n <= 0)
      n = 256;
    ncu = n;
  }
  const int items = p.B * p.H;
  hipLaunchKernelGGL((attn_bwd1_kernel<DM>), dim3(items < ncu ? items : ncu), dim3(NTHREADS), bwd1_lds_bytes(p.T), s, p);
  return MELGPT_OK;
}

}  // namespace

#ifdef MELGPT_CLOCK_STAMPS
// (diagnostic build only) workgroups of attn_fwd32_kernel<DM_HALF> the runtime will keep resident per CU at sequence length T
extern "C" int melgpt_clk_attn_fwd32_occupancy(int T) {
  int n = -1;
  const size_t lds = 2 * (size_t)((T + 31) / 32 * 32) * 128 + 16;
  set_lds(attn_fwd32_kernel<DM_HALF>, 2 * (size_t)MAXT * 128 + 16);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)attn_fwd32_kernel<DM_HALF>, 256, lds) != hipSuccess) return -1;
  return n;
}
#endif

static int g_fwd32 = -2;   // -2: environment not read yet; -1: by shape (default); 0 / 1: forced off / on
static void fwd32_env() {
  if (g_fwd32 == -2) {
    const char* e = getenv("MELGPT_ATTN_FWD32");
    g_fwd32 = e ? (atoi(e) != 0 ? 1 : 0) : -1;
  }
}
extern "C" int melgpt_set_attn_fwd32(int mode) {
  fwd32_env();
  const int prev = g_fwd32;
  g_fwd32 = mode < 0 ? -1 : (mode != 0 ? 1 : 0);
  return prev;
}

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
  hipStream_t s = (hipStream_t)stream;
  // 16-bit lane without the attention map: attn_fwd32_kernel (32-row query tiles, the row of logits in registers) where
  // it is the faster one - measured against the 16-row kernel in one process (profiles/r05_attn_lab.md): the full-square
  // mask of the GPT-VAE encoder (n_unmasked >= T: 126.6 against 138.1 us per layer at 128 x 23 x 265 without dropout,
  // 144 against 144-150 with dropout 1/2); under the causal mask the two tie within 2 % either way (71.3 / 72.5 us
  // without, 75.7-76.9 / 73.7-75.2 with dropout 1/2 at 128 x 16 x 265) and the 16-row kernel keeps the launch.
  // MELGPT_ATTN_FWD32=0 / 1 or melgpt_set_attn_fwd32 force one kernel for every shape (tests run both on every shape).
  fwd32_env();
  const bool use32 = g_fwd32 == 1 || (g_fwd32 == -1 && n_unmasked >= T);
  if (use32 && dtype == MELGPT_BF16 && !att) {
    switch (drop_mode(p)) {
      case DM_NONE: st = launch_fwd32<DM_NONE>(p, s); break;
      case DM_HALF: st = launch_fwd32<DM_HALF>(p, s); break;
      default: st = launch_fwd32<DM_ANY>(p, s); break;
    }
    return st != MELGPT_OK ? st : melgpt_launch_status();
  }
  st = dtype == MELGPT_F32 ? launch_q_mode<float, false>(p, s) : launch_q_mode<bf16_t, false>(p, s);
  return st != MELGPT_OK ? st : melgpt_launch_status();
}

static int g_bwd_two_pass = -1;
extern "C" int melgpt_set_attn_bwd_two_pass(int on) {
  const int prev = g_bwd_two_pass > 0;
  g_bwd_two_pass = on != 0;
  return prev;
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
  // 16-bit lane, plain causal mask, dS tiles fit LDS: ONE launch that evaluates the probabilities once (attn_bwd1_kernel)
  if (g_bwd_two_pass < 0) {  // MELGPT_ATTN_BWD_TWO_PASS=1 / melgpt_set_attn_bwd_two_pass(1) keep the two-kernel path
    const char* e = getenv("MELGPT_ATTN_BWD_TWO_PASS");
    g_bwd_two_pass = e ? atoi(e) != 0 : 0;
  }
  if (!g_bwd_two_pass && dtype == MELGPT_BF16 && n_unmasked == 0 && bwd1_lds_bytes(T) <= 160 * 1024) {
    switch (drop_mode(p)) {
      case DM_NONE: st = launch_bwd1<DM_NONE>(p, s); break;
      case DM_HALF: st = launch_bwd1<DM_HALF>(p, s); break;
      default: st = launch_bwd1<DM_ANY>(p, s); break;
    }
    return st != MELGPT_OK ? st : melgpt_launch_status();
  }
  st = dtype == MELGPT_F32 ? launch_q_mode<float, true>(p, s) : launch_q_mode<bf16_t, true>(p, s);  // dQ (writes delta, which the dK/dV kernel reads)
  if (st != MELGPT_OK) return st;
  st = dtype == MELGPT_F32 ? launch_dkv_mode<float>(p, s) : launch_dkv_mode<bf16_t>(p, s);
  if (st != MELGPT_OK) return st;
  return melgpt_launch_status();
}
