// MFMA helpers shared by the GEMM and attention kernels (gfx950).
//   bf16 lane: v_mfma_f32_16x16x32_bf16 - lane l holds A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15], j = 0..7
//   f32 lane : v_mfma_f32_16x16x4_f32   - lane l holds A[row l&15][k = l>>4],     B[k = l>>4][col l&15]
//              a 16-byte fragment carries 4 such k-slices (one MFMA each); A and B use the same k permutation,
//              so the contraction is exact and both lanes share ONE fragment addressing scheme.
//   C/D (both): col = l&15, row = 4(l>>4) + reg.
#pragma once
#include "common.h"

template <typename T>
struct Tr;
template <>
struct Tr<bf16_t> {
  static constexpr int ES = 2, KSTEP = 64;
};
template <>
struct Tr<float> {
  static constexpr int ES = 4, KSTEP = 32;
};

template <typename T>
__device__ __forceinline__ void mma(f32x4& acc, u32x4 a, u32x4 b) {
  if constexpr (Tr<T>::ES == 2) {
    acc = MELGPT_MFMA_16x16x32(a, b, acc);
  } else {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[e], acc, 0, 0, 0);
  }
}

__device__ __forceinline__ float bf16lo(unsigned v) { return half_lo(v); }
__device__ __forceinline__ float bf16hi(unsigned v) { return half_hi(v); }
