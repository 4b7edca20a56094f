// Shared between the two MFMA GEMM kernels (gemm.hip: 128x128 register-staged, both numerics lanes;
// gemm256.hip: 256-wide LDS-DMA staged, bf16 lane): parameters, LDS swizzles, fused epilogue.
#pragma once
#include <type_traits>
#include <utility>

#include "mma.h"

namespace gemmk {

enum { LAY_ROW = 0, LAY_KMAJ = 1, LAY_CONV = 2,
       LAY_CONV1D = 3 };  // generic 128 x 128 kernel only: LAY_CONV + dilation / reflection / per-chunk taps / LeakyReLU on the operand
                          // (an instantiation of its own: compiled into the 2-D kernels these options cost 15-20 spilled VGPRs)

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a constant expression
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

struct GemmParams {
  const void* A;
  const void* B;
  void* C;
  void* C2;           // optional second output (pre-activation), same dtype/ld as C
  const float* bias;  // (N,) f32 or null
  const void* R;      // residual (ACT none/gelu) or pre-activation (MELGPT_ACT_GELU_GRAD); dtype T
  int M, N, K;
  long long lda, ldb, ldc, ldr;
  long long sA, sB, sC, sR;  // batch strides (elements)
  unsigned a_bytes, b_bytes;  // addressable bytes of ONE batch of A / B (loads beyond return 0)
  int out_f32, accumulate, act;
  int vec_io;  // C / C2 / R rows are 16-byte aligned -> staged row-contiguous epilogue
  float alpha;
  float drop_scale;  // 1/(1-p), or 0 when dropout is off
  unsigned drop_thresh;
  unsigned long long seed;
  unsigned stream_id;
  // implicit-GEMM convolution (A = NHWC input)
  int cH, cW, cC, OH, OW, cstride, pad_t, pad_l, ups, KW;
  // 1-D extensions of the same addressing (MelGAN, generic 128 x 128 kernel only): taps cdil_m1 + 1 apart along x,
  // nn.ReflectionPad1d as an address reflection, LeakyReLU(a_leaky) applied to the A operand on its way into LDS
  // (0 = off).  All zero for the 2-D convolutions.
  int cdil_m1, creflect;
  float a_leaky;
  float out_leaky;  // generic epilogue only: LeakyReLU(out_leaky) on alpha * acc + bias, before the residual (0 = off)
  // optional (persistent kernel, both operands K-major, f32 output: the weight-gradient GEMM): partial sums over K of the
  // rows of A - the bias gradient when A = dY - as (batch * ceil(N / 256)) rows of M floats, row stride ld_rowsum
  float* a_rowsum;
  long long ld_rowsum;
};

constexpr unsigned OOB = 0xFFFFFFF0u;

// exact-erf GELU (nn.GELU(), minGPT.py:102) and its derivative.  erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, i.e. below f32 rounding of the result) sharing ONE exponential between the cdf and the pdf:
//   erf(u) = 1 - (a1 t + ... + a5 t^5) exp(-u^2),  t = 1/(1 + p|u|),  u = x/sqrt(2)  =>  exp(-u^2) = exp(-x^2/2)
// (constants folded - p / sqrt(2) into the reciprocal's FMA, -log2(e) / 2 into the exponent so that v_exp_f32 is used
// directly - and Phi(x) = 1/2 + copysign(erf(|u|) / 2, x) instead of a compare and a select: 23 instead of 29 VALU
// instructions per element pair in the packed form; the fc1 forward evaluates 139 M of these per layer.)
__device__ __forceinline__ float copysign_bits(float mag, float sgn) {
  return __uint_as_float((__float_as_uint(mag) & 0x7FFFFFFFu) | (__float_as_uint(sgn) & 0x80000000u));
}
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf_times_x) {
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.3275911f * 0.70710678118654752440f, 1.0f));
  const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.44269504088896340736f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float half_erf = fmaf(poly * t * e, -0.5f, 0.5f);  // erf(|u|) / 2
  cdf = 0.5f + copysign_bits(half_erf, x);                  // Phi(x)
  pdf_times_x = x * e * 0.39894228040143267794f;            // x * phi(x)
}
__device__ __forceinline__ float gelu_exact(float x) {
  float cdf, xp;
  gelu_parts(x, cdf, xp);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad(float x) {
  float cdf, xp;
  gelu_parts(x, cdf, xp);
  return cdf + xp;
}
// Two elements at a time: the polynomial runs as packed f32 FMAs (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of
// work per issue slot); only the reciprocal and the exponential stay scalar.  Same arithmetic, same results.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_parts2(f32x2 x, f32x2& cdf, f32x2& pdf_times_x) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 d = ax * (0.3275911f * 0.70710678118654752440f) + 1.0f;
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  const f32x2 h = x * x * (-0.5f * 1.44269504088896340736f);
  const f32x2 e = {__builtin_amdgcn_exp2f(h[0]), __builtin_amdgcn_exp2f(h[1])};
  f32x2 poly = t * 1.061405429f + -1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t + -0.284496736f;
  poly = poly * t + 0.254829592f;
  const f32x2 half_erf = poly * t * e * -0.5f + 0.5f;
  cdf = f32x2{copysign_bits(half_erf[0], x[0]), copysign_bits(half_erf[1], x[1])} + 0.5f;
  pdf_times_x = x * e * 0.39894228040143267794f;
}
__device__ __forceinline__ f32x4 gelu_exact4(f32x4 v) {
  f32x2 c0, p0, c1, p1;
  gelu_parts2(f32x2{v[0], v[1]}, c0, p0);
  gelu_parts2(f32x2{v[2], v[3]}, c1, p1);
  return f32x4{v[0] * c0[0], v[1] * c0[1], v[2] * c1[0], v[3] * c1[1]};
}
// activation and derivative from ONE evaluation of the shared parts (forward of Linear -> GELU in MELGPT_ACT_GELU_DACT)
__device__ __forceinline__ void gelu_both4(f32x4 v, f32x4& act, f32x4& der) {
  f32x2 c0, p0, c1, p1;
  gelu_parts2(f32x2{v[0], v[1]}, c0, p0);
  gelu_parts2(f32x2{v[2], v[3]}, c1, p1);
  act = f32x4{v[0] * c0[0], v[1] * c0[1], v[2] * c1[0], v[3] * c1[1]};
  der = f32x4{c0[0] + p0[0], c0[1] + p0[1], c1[0] + p1[0], c1[1] + p1[1]};
}
__device__ __forceinline__ f32x4 gelu_grad4(f32x4 v) {
  f32x2 c0, p0, c1, p1;
  gelu_parts2(f32x2{v[0], v[1]}, c0, p0);
  gelu_parts2(f32x2{v[2], v[3]}, c1, p1);
  return f32x4{c0[0] + p0[0], c0[1] + p0[1], c1[0] + p1[0], c1[1] + p1[1]};
}

// 128-byte rows, 16-byte chunk index XOR-ed with (row>>1)&7: conflict-free ds_read_b128 fragment reads
__device__ __forceinline__ int row_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

// XCD-aware tile order (bijective for any grid size): consecutive tiles of one A row-panel share an L2
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int qd = nwg >> 3, rm = nwg & 7, xcd = b & 7;
  return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (b >> 3);
}

// Fused epilogue for a wave that owns TM x TN accumulator tiles of 16x16 (rows = n, cols = m, see gemm.hip):
// lane (i16, g) holds C[m_base + 16 mt + i16][n_base + 16 nt + 4 g + 0..3].
//
// Global traffic goes through a 4 KiB per-wave LDS staging block, one 16-row slab (one `mt`) at a time, so that
// every global access is a 16-byte piece of a full, contiguous output row (128..256 B per row per instruction)
// instead of 8-byte pieces scattered over 16 rows:
//   R / old C : coalesced 16-byte loads (all issued up front) -> LDS -> read back in accumulator layout
//   C2, C     : accumulator layout -> LDS -> 16-byte row pieces -> global
// LDS rows are XOR-swizzled by (row & 7) on the 16-byte chunk index.  Requires 16-byte aligned rows (host flag
// p.vec_io); otherwise the direct per-lane path below is used.
template <int ROWB>
__device__ __forceinline__ int stage_off(int row, int chunk) { return row * ROWB + ((chunk ^ (row & 7)) << 4); }

// Where a lane's 16-byte pieces of one staged 16-row slab sit - in the wave's staging block and in the output rows -
// worked out ONCE per tile.  (Recomputed per piece - signed division of the lane index, the swizzle, a 64-bit row x ld
// product and three predicates in front of every store, each store behind its own LDS read and lgkmcnt(0) - the plain
// bf16 epilogue of a 256 x 256 tile was 1 120 instructions and 11 k cycles for 128 KiB; profiles/r03_gemm_lab.md.)
template <int ROWB, int OS>  // staged row bytes, output element size
struct OutPlan {
  static constexpr int CPR = ROWB / 16, RPI = 64 / CPR, OSZ = OS;  // chunks per row, rows per 64-lane round
  static constexpr bool WHOLE = (16 * CPR) % 64 == 0;              // (checked where a plan is used)
  static constexpr int PER = WHOLE ? (16 * CPR) / 64 : 1;          // pieces per lane per slab
  // Three registers per lane (the epilogues run at the register limit): piece j sits RPI j rows below piece 0 - in the
  // staging block a constant apart (RPI = 8: the same swizzle) or a constant and one XOR apart (RPI = 4: row & 7
  // alternates by 4, which flips bit 2 of the chunk index), in the output RPI j ld elements further on (wave-uniform).
  int rd0;        // staging-block offset of piece 0
  unsigned go0;   // byte offset of piece 0 from the slab's first output element (row mr, column n_base)
  int lrow;       // row of piece 0 inside the slab; a lane whose columns are out of range: past any row count
  static_assert(!WHOLE || RPI == 8 || RPI == 4 || RPI == 16, "row rounds of 4, 8 or 16");
  __device__ __forceinline__ void init(int lane, long long ld, int n_base, int N) {
    const unsigned ul = (unsigned)lane, ch = ul % CPR, row = ul / CPR;
    rd0 = stage_off<ROWB>((int)row, (int)ch);
    go0 = (row * (unsigned)ld + ch * (16 / OS)) * OS;
    lrow = n_base + (int)(ch * (16 / OS)) < N ? (int)row : 0x3FFFFFFF;
  }
  __device__ __forceinline__ int rd(int j) const {  // (j is a compile-time constant at every call)
    if constexpr (RPI == 4) return (rd0 ^ ((j & 1) << 6)) + j * RPI * ROWB;
    else return rd0 + j * RPI * ROWB;
  }
};

// MODE picks what is compiled in (the persistent 256-wide kernel instantiates one kernel per mode so that its
// epilogue stays small enough for the instruction cache; the generic mode serves gemm.hip and conv_fused.hip):
enum { EPI_GENERIC = 0, EPI_PLAIN16 = 1, EPI_PLAIN32 = 2, EPI_FULL16 = 3,
       // the two plain modes without a residual operand and without accumulation: NO load besides the bias slabs is
       // compiled in (a conditional load the row predicates can skip leaves its registers "pending" in the compiler's
       // wait-count bookkeeping, and the persistent GEMM's next tile then opens with a vmcnt(0) that waits for this
       // tile's output rows to be acknowledged)
       EPI_PLAIN16N = 4, EPI_PLAIN32N = 5,
       // forward of Linear -> GELU alone (MELGPT_ACT_GELU_DACT, bias, bf16 outputs C = gelu(v), C2 = gelu'(v); no R, no
       // dropout, no accumulation): the full mode's dropout / residual machinery costs 16 spilled VGPRs at 256 rows, whose
       // reloads sit in the K loop and drain the LDS-DMA ring on every unit (fc1: 9.7 -> 17 ms per step)
       EPI_DACT16 = 6,
       // Linear -> dropout -> + residual alone (the attention projection and the MLP's second Linear in training: bias,
       // dropout, R, bf16 output; no activation, no second output, no accumulation): unrolled slabs over the per-tile
       // plan, the element counter of the dropout mask from a wave-uniform 64-bit row part + one 32-bit lane part.  In
       // the rolled full mode - every activation compiled in, accumulators copied out through a switch, 100+ scalar
       // registers spilled to lanes - this epilogue was 4.2 k instructions and 28 k cycles per 192 x 256 tile.
       EPI_DROPR16 = 7 };
//   EPI_GENERIC  everything, decided at run time; slabs unrolled
//   EPI_PLAIN16  alpha, bias, +R, accumulate; bf16 output; slabs unrolled (a few hundred instructions in all)
//   EPI_PLAIN32  the same with f32 output (split-K weight gradients)
//   EPI_FULL16   activation / dropout / second output as well, bf16 output; ONE copy of the slab body in a rolled
//                loop (unrolled it is ~100 KiB of code, which a persistent kernel would re-fetch on every tile)
// EPI_PLAIN* / EPI_FULL16 require p.vec_io.
template <typename T, int TM, int TN, int MODE = EPI_GENERIC, bool RALL = false>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, f32x4 (&acc)[TM][TN], const long long (&mrow)[TM],
                                              int row_limit, int n_base, int bz, int lane, char* stage);

template <typename T, int TM, int TN, int MODE = EPI_GENERIC, bool RALL = false>
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x4 (&acc)[TM][TN], int m_base, int n_base, int bz,
                                         int lane, char* stage) {
  long long mrow[TM];
#pragma unroll
  for (int mt = 0; mt < TM; ++mt) mrow[mt] = m_base + mt * 16;
  epilogue_rows<T, TM, TN, MODE, RALL>(p, acc, mrow, 16, n_base, bz, lane, stage);
}

// mrow[mt] = output row (in C) of the first of the 16 consecutive rows held by accumulator slab mt, or < 0 when the
// slab is entirely out of range; rows mrow[mt] + r with r >= row_limit (or >= p.M) are skipped.
template <typename T, int TM, int TN, int MODE, bool RALL>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, f32x4 (&acc)[TM][TN], const long long (&mrow)[TM],
                                              int row_limit, int n_base, int bz, int lane, char* stage) {
  constexpr int ES = Tr<T>::ES;
  constexpr bool ACT = MODE == EPI_GENERIC || MODE == EPI_FULL16;           // activation / dropout / C2 compiled in
  constexpr bool CAN32 = MODE == EPI_GENERIC || MODE == EPI_PLAIN32 || MODE == EPI_PLAIN32N;
  constexpr bool CAN16 = MODE != EPI_PLAIN32 && MODE != EPI_PLAIN32N && ES == 2;
  constexpr bool LOADS = MODE != EPI_PLAIN16N && MODE != EPI_PLAIN32N && MODE != EPI_DACT16;  // R and accumulate compiled in
  constexpr bool ROLLED = MODE == EPI_FULL16;  // (EPI_DACT16 unrolled: its rolled body spent a tenth of its instructions copying accumulators out)
  static_assert(CAN32 || CAN16, "no output type left");
  using RV = typename std::conditional<ES == 4, f32x4, u32x2>::type;  // one 4-element group of R / C in dtype T
  const int i16 = lane & 15, g = lane >> 4;
  const bool f32out = CAN32 && (!CAN16 || p.out_f32);
  char* Cb = (char*)p.C + (long long)bz * p.sC * (f32out ? 4 : ES);
  char* C2b = ((ACT || MODE == EPI_DACT16) && p.C2) ? (char*)p.C2 + (long long)bz * p.sC * (f32out ? 4 : ES) : nullptr;
  const char* Rb = (LOADS && p.R) ? (const char*)p.R + (long long)bz * p.sR * ES : nullptr;

  f32x4 bv[TN];
#pragma clang loop unroll(full)
  for (int nt = 0; nt < TN; ++nt) {
    const int n = n_base + nt * 16 + g * 4;
    bv[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias && n < p.N) bv[nt] = *(const f32x4*)(p.bias + n);
  }
  // an (empty) use right here: the compiler waits for the four loads now, on every path.  Left to the first arithmetic
  // use - which row predicates can skip - their registers stayed "pending" past the epilogue, and the persistent GEMM's
  // next tile opened with a vmcnt(0) guarding their re-use.
#pragma clang loop unroll(full)
  for (int nt = 0; nt < TN; ++nt) asm volatile("" : "+v"(bv[nt]));
  auto unpack = [](RV r) -> f32x4 {
    if constexpr (ES == 4) return r;
    else return f32x4{bf16lo(r[0]), bf16hi(r[0]), bf16lo(r[1]), bf16hi(r[1])};
  };
  auto math = [&](f32x4 v, f32x4 r4, long long m, int n, bool has_r) -> f32x4 {
    if constexpr (ACT) {
      if (p.act == MELGPT_ACT_GELU) {
        v = gelu_exact4(v);
      } else if (p.act == MELGPT_ACT_GELU_GRAD) {
        v *= gelu_grad4(r4);
      } else if (MODE == EPI_GENERIC && p.act == MELGPT_ACT_MUL) {  // the 256-wide kernel serves MUL in its plain modes
        v *= r4;
      }
      if (MODE == EPI_GENERIC && p.out_leaky != 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * p.out_leaky);
      }
      if (p.drop_scale != 0.f) {
        const unsigned long long e0 = ((unsigned long long)bz * p.M + m) * (unsigned long long)p.N + n;
        const unsigned keep = dropout_keep4(p.seed, p.stream_id, e0 >> 2, p.drop_thresh);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (keep >> e & 1) ? v[e] * p.drop_scale : 0.f;
      }
      if (has_r && p.act != MELGPT_ACT_GELU_GRAD && p.act != MELGPT_ACT_MUL) v += r4;
    } else {
      if (has_r) v = p.act == MELGPT_ACT_MUL ? v * r4 : v + r4;   // plain modes: residual add, or the saved-derivative product
    }
    return v;
  };

  if (MODE != EPI_GENERIC || p.vec_io) {
    // ------------------------------------------------------------------ staged, row-contiguous path
    constexpr int IN_ROWB = TN * 16 * ES, IN_CPR = IN_ROWB / 16, IN_PER = (16 * IN_CPR) / 64;  // chunks / lane / slab
    constexpr int NR = IN_PER > 0 ? IN_PER : 1;
    // (the rolled full mode at 256 rows keeps the per-piece forms: with the plan's registers live across its loop that
    // instantiation spills 10 VGPRs whose reloads land in the K loop)
    constexpr bool PLAN16 = CAN16 && (MODE != EPI_FULL16 || TM <= 6);
    OutPlan<TN * 32, 2> plan16;
    unsigned go_r0 = 0;  // the plan's go0 with R's leading dimension
    int wr0 = 0;         // staging offset of this lane's accumulator-layout 8 bytes of column tile 0; tile nt: ^ (nt << 5)
    if (PLAN16 && (!f32out || Rb)) {
      plan16.init(lane, p.ldc, n_base, p.N);
      if (LOADS && Rb) go_r0 = (unsigned)(((unsigned)lane / plan16.CPR) * (unsigned)p.ldr + ((unsigned)lane % plan16.CPR) * 8u) * 2u;
      wr0 = stage_off<TN * 32>(i16, g >> 1) + (g & 1) * 8;
    }
    const unsigned lq = MODE == EPI_DROPR16 ? (unsigned)i16 * (unsigned)(p.N >> 2) + (unsigned)(n_base >> 2) + (unsigned)g : 0u;
    auto wr16 = [&](int nt) { return PLAN16 ? (wr0 ^ (nt << 5)) : stage_off<TN * 32>(i16, 2 * nt + (g >> 1)) + (g & 1) * 8; };
    // R is fetched for HALF the slabs at a time (NB register sets), consumed, then fetched for the other half.
    // Finer-grained prefetching does not survive the compiler: with LDS-DMA pieces possibly in flight (the persistent
    // kernel's ring) hipcc puts a full vmcnt(0) in front of every use of an ordinary load, so any fetch issued
    // between two uses is waited for at once.  Two batches = two exposed latencies per tile instead of TM.
    // (slabs per batch: half the tile where the registers allow it; the rolled full epilogue is at the 256-VGPR
    // limit with two sets)
    // (RALL: the plain bf16 mode requests ALL its slabs up front: the loads are ordinary, counted loads - the persistent
    // kernel's LDS-DMA is invisible to the compiler since round 2 - so only the first slab's latency is exposed; the
    // implicit-GEMM convolution's 256-row instantiation has no registers for it)
    // (the lean dropout + residual mode at 192 rows sits at 191 VGPRs: all six slabs' R up front as well - one exposed
    // latency per tile instead of two)
    constexpr int NB = (MODE == EPI_GENERIC || ROLLED) ? (TM < 2 ? 1 : 2)
                       : ((RALL && MODE == EPI_PLAIN16) || (MODE == EPI_DROPR16 && TM <= 6) ? TM : (TM + 1) / 2);
    u32x4 rin[NB][NR];
    auto fetch_r = [&](long long mr, u32x4 (&dst)[NR]) {
      if constexpr (ES == 2 && PLAN16) {  // same geometry as the bf16 output slab: the plan's rows and pieces
        const int rows = mr < 0 ? 0 : (int)(p.M - mr < row_limit ? p.M - mr : row_limit);  // (wave-uniform)
        const char* sb = Rb + (mr * p.ldr + n_base) * 2;
        const long long rstep = p.ldr * (plan16.RPI * 2);
#pragma clang loop unroll(full)
        for (int j = 0; j < IN_PER; ++j) {
          dst[j] = u32x4{0u, 0u, 0u, 0u};
          if (plan16.lrow + plan16.RPI * j < rows) dst[j] = *(const u32x4*)(sb + j * rstep + go_r0);
        }
        return;
      }
#pragma clang loop unroll(full)
      for (int j = 0; j < IN_PER; ++j) {
        const int q = lane + 64 * j, row = q / IN_CPR, ch = q % IN_CPR;
        const long long m = mr + row;
        const int n = n_base + ch * (16 / ES);
        dst[j] = u32x4{0u, 0u, 0u, 0u};
        if (mr >= 0 && row < row_limit && m < p.M && n < p.N) dst[j] = *(const u32x4*)(Rb + (m * p.ldr + n) * ES);
      }
    };
    // all of a lane's pieces of the staged slab are read back first (ONE LDS round trip), then stored; `rmw` adds the old
    // contents of C (accumulate)
    auto flush = [&](auto& pl, char* Ob, const char* src, long long mr, auto rmw) {
      using PL = typename std::remove_reference<decltype(pl)>::type;
      constexpr int OS = PL::OSZ;
      static_assert(PL::WHOLE, "a staged slab is a whole number of 64-lane rounds");
      const int rows = mr < 0 ? 0 : (int)(p.M - mr < row_limit ? p.M - mr : row_limit);  // (wave-uniform)
      char* sb = Ob + (mr * p.ldc + n_base) * OS;
      const long long rstep = p.ldc * (PL::RPI * OS);  // (wave-uniform) bytes between a lane's pieces in the output
      u32x4 o[PL::PER];
#pragma clang loop unroll(full)
      for (int j = 0; j < PL::PER; ++j) o[j] = *(const u32x4*)(src + pl.rd(j));
#pragma clang loop unroll(full)
      for (int j = 0; j < PL::PER; ++j)
        if (pl.lrow + PL::RPI * j < rows) {
          u32x4* dst = (u32x4*)(sb + j * rstep + pl.go0);
          rmw(o[j], dst);
          *dst = o[j];
        }
    };
    auto no_rmw = [](u32x4&, const u32x4*) {};
    // one 16-row slab: accumulators av, first output row mr, R in register set SET
    auto slab = [&](const f32x4 (&av)[TN], long long mr, auto set_c) {
      constexpr int SET = decltype(set_c)::value;
      const long long m = mr + i16;
      f32x4 r4[TN];
      if (Rb) {
#pragma clang loop unroll(full)
        for (int j = 0; j < IN_PER; ++j) {
          const int q = lane + 64 * j;
          if constexpr (ES == 2 && PLAN16) *(u32x4*)(stage + plan16.rd(j)) = rin[SET][j];
          else *(u32x4*)(stage + stage_off<IN_ROWB>(q / IN_CPR, q % IN_CPR)) = rin[SET][j];
        }
#pragma clang loop unroll(full)
        for (int nt = 0; nt < TN; ++nt) {
          if constexpr (ES == 4) r4[nt] = *(const f32x4*)(stage + stage_off<IN_ROWB>(i16, 4 * nt + g));
          else r4[nt] = unpack(*(const u32x2*)(stage + wr16(nt)));
        }
      }
      // MELGPT_ACT_GELU_DACT (forward of Linear -> GELU): activation and derivative come from ONE evaluation of the
      // shared exponential / reciprocal.  bf16 outputs: both 2 KiB slabs are staged side by side in the wave's 4 KiB
      // block and written out together - no second copy of the tile in registers (the 256-wide kernel's epilogue is
      // at the register limit).  f32 outputs (generic mode only): the activation waits in registers for the C pass.
      // (the full mode does not carry this path: with it the 256-row kernel spills 16 VGPRs into its K loop)
      const bool dact = MODE == EPI_DACT16 || (MODE == EPI_GENERIC && p.act == MELGPT_ACT_GELU_DACT);
      if constexpr ((MODE == EPI_GENERIC || MODE == EPI_DACT16) && CAN16) {
        if (dact && !f32out) {
          constexpr int ROWB = TN * 32;
          static_assert(2 * 16 * ROWB <= 4096, "two bf16 slabs must fit the wave's staging block");
          char* st2 = stage + 16 * ROWB;
#pragma clang loop unroll(full)
          for (int nt = 0; nt < TN; ++nt) {
            f32x4 a, d;
            gelu_both4(av[nt] * p.alpha + bv[nt], a, d);
            if constexpr (MODE != EPI_DACT16) a = math(a, Rb ? r4[nt] : a, m, n_base + nt * 16 + g * 4, Rb != nullptr);
            const int o = wr16(nt);
            *(u32x2*)(stage + o) = u32x2{pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3])};
            *(u32x2*)(st2 + o) = u32x2{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3])};
          }
          flush(plan16, C2b, stage, mr, no_rmw);
          flush(plan16, Cb, st2, mr, no_rmw);
          return;
        }
      }
      f32x4 actv[MODE == EPI_GENERIC ? TN : 1];
      // (EPI_DROPR16) mask counter of the slab's row mr, lane part added per column tile; N % 4 == 0 (as math() assumes)
      const unsigned long long sq = MODE == EPI_DROPR16 ? ((unsigned long long)bz * p.M + mr) * (unsigned long long)(p.N >> 2) : 0ull;
      auto value = [&](int pass, int nt) -> f32x4 {
        f32x4 v = av[nt] * p.alpha + bv[nt];
        if constexpr (MODE == EPI_DROPR16) {  // (the launcher picks this mode only with dropout on and R present)
          const unsigned keep = dropout_keep4(p.seed, p.stream_id, sq + (lq + 4u * nt), p.drop_thresh);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (keep >> e & 1) ? v[e] * p.drop_scale : 0.f;
          return v + r4[nt];
        }
        if constexpr (MODE == EPI_GENERIC) {
          if (dact) {
            if (pass == 0) {
              f32x4 der;
              gelu_both4(v, actv[nt], der);
              return der;
            }
            v = actv[nt];
          }
        }
        if (pass == 1) v = math(v, Rb ? r4[nt] : v, m, n_base + nt * 16 + g * 4, Rb != nullptr);
        return v;
      };
      // one or two outputs, each staged and written as full rows
#pragma clang loop unroll(full)
      for (int pass = ACT ? 0 : 1; pass < 2; ++pass) {
        char* Ob = pass == 0 ? C2b : Cb;
        if (!Ob) continue;
        if (f32out) {
          if constexpr (CAN32) {
            constexpr int ROWB = TN * 64;
#pragma clang loop unroll(full)
            for (int nt = 0; nt < TN; ++nt) *(f32x4*)(stage + stage_off<ROWB>(i16, 4 * nt + g)) = value(pass, nt);
            // (f32 rows keep the per-piece form: four pieces per lane read back together measured 1.5 % slower on the
            // weight-gradient GEMMs - 1 453-1 467 -> 1 437 TFLOP/s in the lab - where the bf16 modes gained 3-5 %)
            constexpr int CPR = ROWB / 16, PER = (16 * CPR) / 64;
#pragma clang loop unroll(full)
            for (int j = 0; j < PER; ++j) {
              const int q = lane + 64 * j, row = q / CPR, ch = q % CPR;
              const long long mm = mr + row;
              const int nn = n_base + ch * 4;
              if (mr >= 0 && row < row_limit && mm < p.M && nn < p.N) {
                f32x4 o = *(const f32x4*)(stage + stage_off<ROWB>(row, ch));
                float* dst = (float*)(Ob + (mm * p.ldc + nn) * 4);
                if (LOADS && pass == 1 && p.accumulate) o += *(const f32x4*)dst;
                *(f32x4*)dst = o;
              }
            }
          }
        } else {
          if constexpr (CAN16) {
            constexpr int ROWB = TN * 32;
#pragma clang loop unroll(full)
            for (int nt = 0; nt < TN; ++nt) {
              const f32x4 v = value(pass, nt);
              *(u32x2*)(stage + wr16(nt)) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
            if constexpr (!PLAN16) {
              constexpr int CPR = ROWB / 16, PER = (16 * CPR) / 64;
#pragma clang loop unroll(full)
              for (int j = 0; j < PER; ++j) {
                const int q = lane + 64 * j, row = q / CPR, ch = q % CPR;
                const long long mm = mr + row;
                const int nn = n_base + ch * 8;
                if (mr >= 0 && row < row_limit && mm < p.M && nn < p.N) {
                  u32x4 o = *(const u32x4*)(stage + stage_off<ROWB>(row, ch));
                  u32x4* dst = (u32x4*)(Ob + (mm * p.ldc + nn) * 2);
                  if (LOADS && pass == 1 && p.accumulate) {
                    const u32x4 c = *dst;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                      o[e] = pack_bf16x2(bf16lo(o[e]) + bf16lo(c[e]), bf16hi(o[e]) + bf16hi(c[e]));
                  }
                  *dst = o;
                }
              }
            } else if (LOADS && MODE != EPI_DROPR16 && pass == 1 && p.accumulate) {
              flush(plan16, Ob, stage, mr, [](u32x4& o, const u32x4* dst) {
                const u32x4 c = *dst;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  o[e] = pack_bf16x2(bf16lo(o[e]) + bf16lo(c[e]), bf16hi(o[e]) + bf16hi(c[e]));
              });
            } else {
              flush(plan16, Ob, stage, mr, no_rmw);
            }
          }
        }
      }
    };

    // batch b covers slabs b*NB .. b*NB + NB - 1
    auto fetch_batch = [&](auto b_c) {
      constexpr int B0 = decltype(b_c)::value * NB;
#pragma clang loop unroll(full)
      for (int k = 0; k < NB; ++k)
        if (B0 + k < TM) fetch_r(mrow[B0 + k < TM ? B0 + k : 0], rin[k]);
    };
    auto run_batch = [&](auto b_c) {
      constexpr int B0 = decltype(b_c)::value * NB;
      if (Rb) fetch_batch(b_c);
      static_for<NB>([&](auto k_c) {
        constexpr int k = decltype(k_c)::value;
        if constexpr (B0 + k < TM) {
          // slabs strictly one after another: interleaving the unrolled slabs only raises register pressure (the
          // accumulators already fill half the file) and ends in scratch spills, whose reloads are memory round trips
          __builtin_amdgcn_sched_barrier(0);
          slab(acc[B0 + k], mrow[B0 + k], k_c);
        }
      });
    };
    constexpr int NBATCH = (TM + NB - 1) / NB;
    if constexpr (!ROLLED) {
      static_for<NBATCH>([&](auto b_c) { run_batch(b_c); });
    } else {
      // one copy of the batch body: a uniform switch copies the batch's accumulators and row origins out of the
      // register arrays
      static_assert(!ROLLED || (NB == 2 && TM % 2 == 0 && TM <= 8), "rolled epilogue: up to four batches of two slabs");
      // R is fetched ONE BATCH AHEAD into a second register set (the persistent kernel's LDS-DMA is invisible to the
      // compiler now, so a fetch issued between two uses keeps flying - before, hipcc drained vmcnt(0) at every use)
      auto rows_of = [&](int b, long long (&mr)[NB]) {
#pragma clang loop unroll(full)
        for (int k = 0; k < NB; ++k) {
          mr[k] = mrow[k];
#pragma clang loop unroll(full)
          for (int q = 1; q < NBATCH; ++q) mr[k] = b == q ? mrow[q * NB + k] : mr[k];
        }
      };
      // (only for the 192-row tile: at 256 rows the second set does not fit the register file - 30 spilled VGPRs)
      constexpr bool AHEAD = TM <= 6;
      u32x4 rnx[AHEAD ? NB : 1][NR];
      if (AHEAD && Rb) {
        long long mr0[NB];
        rows_of(0, mr0);
#pragma clang loop unroll(full)
        for (int k = 0; k < NB; ++k) fetch_r(mr0[k], rnx[AHEAD ? k : 0]);
      }
#pragma clang loop unroll(disable)
      for (int b = 0; b < NBATCH; ++b) {
        long long mr[NB];
        rows_of(b, mr);
        if constexpr (AHEAD) {
          if (Rb) {
#pragma clang loop unroll(full)
            for (int k = 0; k < NB; ++k)
#pragma clang loop unroll(full)
              for (int j = 0; j < NR; ++j) rin[k][j] = rnx[AHEAD ? k : 0][j];
            if (b + 1 < NBATCH) {
              long long mn[NB];
              rows_of(b + 1, mn);
#pragma clang loop unroll(full)
              for (int k = 0; k < NB; ++k) fetch_r(mn[k], rnx[AHEAD ? k : 0]);
            }
          }
        } else if (Rb) {
#pragma clang loop unroll(full)
          for (int k = 0; k < NB; ++k) fetch_r(mr[k], rin[k]);
        }
        static_for<NB>([&](auto k_c) {
          constexpr int k = decltype(k_c)::value;
          __builtin_amdgcn_sched_barrier(0);
          f32x4 av[TN];
          auto take = [&](auto b_c) {
            constexpr int MT = decltype(b_c)::value * NB + k;
            if constexpr (MT < TM) {
#pragma clang loop unroll(full)
              for (int nt = 0; nt < TN; ++nt) av[nt] = acc[MT][nt];
            }
          };
          switch (b) {
            case 0: take(std::integral_constant<int, 0>{}); break;
            case 1: take(std::integral_constant<int, 1>{}); break;
            case 2: take(std::integral_constant<int, 2>{}); break;
            default: take(std::integral_constant<int, 3>{}); break;
          }
          slab(av, mr[k], k_c);
        });
      }
    }
    return;
  }

  // ---------------------------------------------------------------------- direct per-lane path (unaligned rows)
  if constexpr (MODE == EPI_GENERIC) {
#pragma clang loop unroll(full)
    for (int nt = 0; nt < TN; ++nt) {
      const int n = n_base + nt * 16 + g * 4;
      if (n >= p.N) continue;
#pragma clang loop unroll(full)
      for (int mt = 0; mt < TM; ++mt) {
        const long long m = mrow[mt] + i16;
        if (mrow[mt] < 0 || i16 >= row_limit || m >= p.M) continue;
        f32x4 v = acc[mt][nt] * p.alpha + bv[nt];
        if (C2b) {
          f32x4 c2 = v;
          if (p.act == MELGPT_ACT_GELU_DACT) gelu_both4(v, v, c2);  // v becomes the activation, C2 its derivative
          if (f32out) *(f32x4*)(C2b + ((long long)m * p.ldc + n) * 4) = c2;
          else *(u32x2*)(C2b + ((long long)m * p.ldc + n) * 2) = u32x2{pack_bf16x2(c2[0], c2[1]), pack_bf16x2(c2[2], c2[3])};
        }
        f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
        if (Rb) r4 = unpack(*(const RV*)(Rb + ((long long)m * p.ldr + n) * ES));
        v = math(v, r4, m, n, Rb != nullptr);
        if (f32out) {
          float* dst = (float*)(Cb + ((long long)m * p.ldc + n) * 4);
          if (p.accumulate) v += *(const f32x4*)dst;
          *(f32x4*)dst = v;
        } else {
          u32x2* dst = (u32x2*)(Cb + ((long long)m * p.ldc + n) * 2);
          if (p.accumulate) {
            u32x2 o = *dst;
            v += f32x4{bf16lo(o[0]), bf16hi(o[0]), bf16lo(o[1]), bf16hi(o[1])};
          }
          *dst = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
      }
    }
  }
}

// implemented in gemm256.hip: returns MELGPT_ERR_UNSUPPORTED when the shape/layout is not covered
int launch_gemm256(const GemmParams& p, int alay, int blay, int batch, int tile_cfg, hipStream_t s);
// gemm8p.hip: the ping-pong K loop for one (layout, epilogue mode, tile height) of the persistent GEMM, same tile lists
int launch_gemm8p(const GemmParams& p, int alay, int blay, int mode, int tm, int tiles_m, int tiles_n, int batch, int RN,
                  int grid, hipStream_t s);

int launch_conv8p_n128(const GemmParams& p, hipStream_t s);  // gemm8p.hip: Cout <= 128 convolutions as 256 x 128 tiles
}  // namespace gemmk
