// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/melgpt.h"

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef unsigned short bf16_t;  // raw bf16 bits

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

#define MELGPT_CHECK(cond, code) \
  do {                           \
    if (!(cond)) return (code);  \
  } while (0)

// ---------------------------------------------------------------- hand-counted waits
// Kernels that order LDS-DMA / in-flight loads by COUNTED waits (s_waitcnt vmcnt(N) / lgkmcnt(N) with N > 0: "all but the
// N youngest have landed") spell the count through these, so that the race-screen build (-DMELGPT_VMCNT0,
// build.py flavour "vm0", tests/test_race_screens_gpu.py) can turn every one of them into a full drain: that build
// cannot read a piece too early, so its outputs ARE the intended ones, and the production build must reproduce them bit
// for bit - a miscounted wait shows up as a difference in the production arm only.
#ifdef MELGPT_VMCNT0
#define MELGPT_WAITN(n) 0
#else
#define MELGPT_WAITN(n) (n)
#endif
#define MELGPT_STR2(x) #x
#define MELGPT_STR(x) MELGPT_STR2(x)
#ifdef MELGPT_VMCNT0
#define MELGPT_VMCNT(n) "vmcnt(0)"
#else
#define MELGPT_VMCNT(n) "vmcnt(" MELGPT_STR(n) ")"
#endif

// ---------------------------------------------------------------- in-kernel clock (diagnostic build only)
// -DMELGPT_CLOCK_STAMPS (tools/lab/build_clock_lib.py -> tools/lab/clock_lab.py): thread 0 of every workgroup stamps
// s_memtime (shader cycles) and s_memrealtime (100 MHz, one clock for the whole chip) at the top of the kernel and when it
// leaves; in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz, median over workgroups, after >= 2 s of back-to-back
// launches on random data (MI355X_MICROARCH "DVFS give-back" item 6).  The stamps go to a buffer of their own that no kernel
// reads; in the product build the macros are empty and no stamp executes.
#ifdef MELGPT_CLOCK_STAMPS
#define MELGPT_CLK_SLOTS 2048
#define MELGPT_CLK_DECL(name)                                                                                     \
  __device__ unsigned long long name[MELGPT_CLK_SLOTS * 2];                                                       \
  extern "C" int melgpt_##name(unsigned long long* out) {                                                         \
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(name), sizeof(unsigned long long) * MELGPT_CLK_SLOTS * 2) == hipSuccess ? 0 : -1; \
  }
#define MELGPT_CLK_BEGIN() \
  const unsigned long long clk_c0_ = __builtin_amdgcn_s_memtime(), clk_r0_ = __builtin_amdgcn_s_memrealtime()
#define MELGPT_CLK_END(name)                                                                                      \
  do {                                                                                                            \
    const unsigned slot_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                        \
    if (threadIdx.x == 0 && slot_ < MELGPT_CLK_SLOTS) {                                                           \
      name[2 * slot_] = __builtin_amdgcn_s_memtime() - clk_c0_;                                                   \
      name[2 * slot_ + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0_;                                           \
    }                                                                                                             \
  } while (0)
#else
#define MELGPT_CLK_DECL(name)
#define MELGPT_CLK_BEGIN() do { } while (0)
#define MELGPT_CLK_END(name) do { } while (0)
#endif

void melgpt_count_gemm_loop(int pingpong);  // abi.hip: launch counters behind melgpt_gemm_loop_launches

static inline int melgpt_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MELGPT_OK : MELGPT_ERR_LAUNCH;
}

// ---------------------------------------------------------------- 16-bit storage format <-> f32
// The 16-bit lane of every kernel is written against `bf16_t` = 16 raw bits and the handful of helpers below; the
// library is built in two flavours from the same sources: bfloat16 (default: libmelgpt_hip.so) and IEEE half
// (-DMELGPT_HALF_FP16: libmelgpt_hip_fp16.so - BASELINE configs[4] names fp16).  Same MFMA rate either way
// (v_mfma_f32_16x16x32_{bf16,f16}); dtype code MELGPT_BF16 means "the library's 16-bit format".
// Conversions are round-to-nearest-even; a NaN stays a NaN (plain casts).
#ifdef MELGPT_HALF_FP16
typedef _Float16 melgpt_half_native;
typedef __attribute__((ext_vector_type(8))) _Float16 melgpt_half8;
#define MELGPT_MFMA_16x16x32(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(melgpt_half8, a), __builtin_bit_cast(melgpt_half8, b), c, 0, 0, 0)
#define MELGPT_MFMA_32x32x16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(melgpt_half8, a), __builtin_bit_cast(melgpt_half8, b), c, 0, 0, 0)
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
__device__ __forceinline__ float half_lo(unsigned v) { return bf16_to_f32((bf16_t)(v & 0xFFFFu)); }
__device__ __forceinline__ float half_hi(unsigned v) { return bf16_to_f32((bf16_t)(v >> 16)); }
#else
#define MELGPT_MFMA_16x16x32(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0)
#define MELGPT_MFMA_32x32x16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0)
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  // plain cast semantic (round-to-nearest-even); hipcc lowers __bf16 casts to v_cvt_pk_bf16_f32 on gfx950
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float half_lo(unsigned v) { return __uint_as_float(v << 16); }          // element 0 of a packed pair
__device__ __forceinline__ float half_hi(unsigned v) { return __uint_as_float(v & 0xFFFF0000u); }  // element 1
#endif
// two values -> one packed pair.  As a VECTOR conversion: written as two scalar casts + shift + or, hipcc emitted two
// v_cvt_pk_bf16_f32 (each with a dead half) and a v_or_b32_sdwa per pair - 192 instructions instead of 64 in a
// 256 x 256 GEMM tile's epilogue; the vector form is ONE v_cvt_pk_bf16_f32 (same round-to-nearest-even, same NaNs).
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef float melgpt_f32x2 __attribute__((ext_vector_type(2)));
#ifdef MELGPT_HALF_FP16
  typedef _Float16 melgpt_h16x2 __attribute__((ext_vector_type(2)));
#else
  typedef __bf16 melgpt_h16x2 __attribute__((ext_vector_type(2)));
#endif
  return __builtin_bit_cast(unsigned, __builtin_convertvector(melgpt_f32x2{lo, hi}, melgpt_h16x2));
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// ---------------------------------------------------------------- wave64 reductions
// DPP inside each row of 16 lanes (xor 1, xor 2, half mirror, mirror: ~4 VALU issue slots each, against ~100 cycles
// for a ds_bpermute shuffle), then the four row results through v_readlane.  Every lane gets the same value.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v, float absent = 0.f) {  // absent: what a disabled source lane yields
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(absent), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_value(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v, v));
  v = fmaxf(v, dpp_move<0x4E>(v, v));
  v = fmaxf(v, dpp_move<0x141>(v, v));
  v = fmaxf(v, dpp_move<0x140>(v, v));
  return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// ---------------------------------------------------------------- buffer resources (OOB loads return 0)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

// ---------------------------------------------------------------- counter-based dropout masks
// Stateless: the keep decision of an element is a pure function of (seed, stream id, counter), so the backward
// kernels regenerate the forward's mask instead of storing it.  (A Philox4x32-10 call costs ~100 integer ops; at one
// call per 4 attention probabilities / GEMM outputs it was the largest VALU cost of both kernels.)
// 32-bit integer hash with two multiplies ("lowbias32", bias 0.17): a bijection of the counter, good avalanche.
// Integer multiplies are quarter-rate on CDNA, so the mask generator is priced in multiplies: 4 per 4 elements.
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// The keys fold in the seed, the stream id and the HIGH counter word (almost always zero or wave-uniform), with shifts
// and adds only; a kernel whose high word is fixed per workgroup computes them once (attention: c1 = batch*heads + head).
struct DropKeys {
  unsigned k0, k1;  // hash keys of the first / second chain (8-bit mode uses k0 only)
  unsigned t;       // threshold on the uniform's width: thresh >> 24 (8-bit mode) or thresh >> 16
  bool b8;          // p is a multiple of 1/256: ONE hash chain yields the four 8-bit uniforms
};
__device__ __forceinline__ DropKeys drop_keys(unsigned long long seed, unsigned stream_id, unsigned c1, unsigned thresh) {
  DropKeys d;
  d.b8 = (thresh & 0x00FFFFFFu) == 0u;
  d.k0 = ((unsigned)seed ^ (stream_id * 0x9E3779B9u)) + ((c1 << 13) | (c1 >> 19));
  if (d.b8) d.k0 += (unsigned)(seed >> 32);
  d.k1 = ((unsigned)(seed >> 32) + stream_id * 0x85EBCA6Bu + 0x6A09E667u) ^ c1;
  d.t = d.b8 ? thresh >> 24 : thresh >> 16;
  return d;
}
// keep decisions of the 4 consecutive elements at low counter word c0 (k[i] = element i kept)
__device__ __forceinline__ void drop_keep4(const DropKeys& d, unsigned c0, bool (&k)[4]) {
  if (d.b8) {
    const unsigned r = hash32(c0 + d.k0);
    k[0] = (r & 0xFFu) >= d.t; k[1] = ((r >> 8) & 0xFFu) >= d.t; k[2] = ((r >> 16) & 0xFFu) >= d.t; k[3] = (r >> 24) >= d.t;
  } else {
    const unsigned a = hash32(c0 + d.k0), b = hash32((c0 ^ 0x5bd1e995u) + d.k1);
    k[0] = (a & 0xFFFFu) >= d.t; k[1] = (a >> 16) >= d.t; k[2] = (b & 0xFFFFu) >= d.t; k[3] = (b >> 16) >= d.t;
  }
}
// keep-mask for 4 consecutive elements whose first linear index is 4*q: bit i set = element kept.
// `thresh` = round(p_drop * 2^32) (kept for ABI stability); compared on its top 16 bits: p is honoured to 2^-16
// (on its top 8 bits when the lower 24 are zero, i.e. when p is a multiple of 1/256 - then exactly; the reference
// configs use 0.5 and 0.0).
__device__ __forceinline__ unsigned dropout_keep4(unsigned long long seed, unsigned stream_id,
                                                  unsigned long long q, unsigned thresh) {
  const DropKeys d = drop_keys(seed, stream_id, (unsigned)(q >> 32), thresh);
  bool k[4];
  drop_keep4(d, (unsigned)q, k);
  return (k[0] ? 1u : 0u) | (k[1] ? 2u : 0u) | (k[2] ? 4u : 0u) | (k[3] ? 8u : 0u);
}

// ---------------------------------------------------------------- Philox4x32-10 (sampling / reparameterisation draws)
struct Philox4 {
  unsigned x, y, z, w;
};
__device__ __forceinline__ Philox4 philox4x32_10(unsigned long long seed, unsigned long long ctr_lo,
                                                 unsigned ctr_hi) {
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  unsigned c0 = (unsigned)ctr_lo, c1 = (unsigned)(ctr_lo >> 32), c2 = ctr_hi, c3 = 0x9E3779B9u;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    unsigned n1 = (unsigned)p1;
    unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

extern "C" int melgpt_get_reserved_cus(void);  // abi.hip: CUs the persistent kernels leave free (data-parallel runs)
