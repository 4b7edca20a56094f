// Shared between the two MFMA GEMM kernels (gemm.hip: 128x128 register-staged, both numerics lanes;
// gemm256.hip: 256-wide LDS-DMA staged, bf16 lane): parameters, LDS swizzles, fused epilogue.
#pragma once
#include "mma.h"

namespace gemmk {

enum { LAY_ROW = 0, LAY_KMAJ = 1, LAY_CONV = 2 };

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
  float alpha;
  float drop_scale;  // 1/(1-p), or 0 when dropout is off
  unsigned drop_thresh;
  unsigned long long seed;
  unsigned stream_id;
  // implicit-GEMM convolution (A = NHWC input)
  int cH, cW, cC, OH, OW, cstride, pad_t, pad_l, ups, KW;
};

constexpr unsigned OOB = 0xFFFFFFF0u;

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float x) {
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
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
template <typename T, int TM, int TN>
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x4 (&acc)[TM][TN], int m_base, int n_base, int bz,
                                         int lane) {
  constexpr int ES = Tr<T>::ES;
  const int i16 = lane & 15, g = lane >> 4;
  char* Cb = (char*)p.C + (long long)bz * p.sC * (p.out_f32 ? 4 : ES);
  char* C2b = p.C2 ? (char*)p.C2 + (long long)bz * p.sC * (p.out_f32 ? 4 : ES) : nullptr;
  const char* Rb = p.R ? (const char*)p.R + (long long)bz * p.sR * ES : nullptr;
#pragma clang loop unroll(full)
  for (int nt = 0; nt < TN; ++nt) {
    const int n = n_base + nt * 16 + g * 4;
    if (n >= p.N) continue;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *(const f32x4*)(p.bias + n);
#pragma clang loop unroll(full)
    for (int mt = 0; mt < TM; ++mt) {
      const int m = m_base + mt * 16 + i16;
      if (m >= p.M) continue;
      f32x4 v = acc[mt][nt] * p.alpha + bv;
      if (C2b) {
        if (p.out_f32 || ES == 4) *(f32x4*)(C2b + ((long long)m * p.ldc + n) * 4) = v;
        else *(u32x2*)(C2b + ((long long)m * p.ldc + n) * 2) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
      f32x4 rv = {0.f, 0.f, 0.f, 0.f};
      if (Rb) {
        if constexpr (ES == 4) {
          rv = *(const f32x4*)(Rb + ((long long)m * p.ldr + n) * 4);
        } else {
          u32x2 r = *(const u32x2*)(Rb + ((long long)m * p.ldr + n) * 2);
          rv = f32x4{bf16lo(r[0]), bf16hi(r[0]), bf16lo(r[1]), bf16hi(r[1])};
        }
      }
      if (p.act == MELGPT_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_exact(v[e]);
      } else if (p.act == MELGPT_ACT_GELU_GRAD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gelu_grad(rv[e]);
      }
      if (p.drop_scale != 0.f) {
        const unsigned long long e0 = ((unsigned long long)bz * p.M + m) * (unsigned long long)p.N + n;
        const unsigned keep = dropout_keep4(p.seed, p.stream_id, e0 >> 2, p.drop_thresh);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (keep >> e & 1) ? v[e] * p.drop_scale : 0.f;
      }
      if (Rb && p.act != MELGPT_ACT_GELU_GRAD) v += rv;
      if (p.out_f32 || ES == 4) {
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

// implemented in gemm256.hip: returns MELGPT_ERR_UNSUPPORTED when the shape/layout is not covered
int launch_gemm256(const GemmParams& p, int alay, int blay, int batch, int tile_cfg, hipStream_t s);

}  // namespace gemmk
