// MelGAN generator glue kernels (reference vocoder/modules.py:24-79).  The generator's Conv1d /
// ConvTranspose1d layers run as per-tap batched GEMMs on shifted row windows of a channels-last (B, L, C) activation
// (melgpt_gemm with `accumulate`; the host side is melspec_gpt_vqvae_amd/vocoder/modules.py), so the only new device
// code is: (1) the padded copy that also applies the LeakyReLU(0.2) which precedes every convolution
// (nn.ReflectionPad1d / zero rows for the transposed convolutions), (2) the final 7-tap convolution to ONE channel
// followed by tanh.  Both are one pass over the activation (HBM bound).
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ float vld(const T* p) {
  if constexpr (sizeof(T) == 2) return bf16_to_f32(*p);
  else return *p;
}
template <typename T>
__device__ __forceinline__ void vst(T* p, float v) {
  if constexpr (sizeof(T) == 2) *p = f32_to_bf16(v);
  else *p = v;
}

// y[b, j, c] = act(x[b, src(j - pad), c]),  j in [0, L + 2 pad);  reflect: src(i) = -i for i < 0, 2(L-1) - i for i >= L
// (nn.ReflectionPad1d, needs pad < L); zero mode: rows outside [0, L) are zeros.  act = LeakyReLU(slope) or identity
// (slope = 1).  16-byte vectors along C.
template <typename T>
__global__ __launch_bounds__(256) void pad1d_act_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int L, int C,
                                                        int pad, int reflect, float slope) {
  constexpr int VEC = 16 / sizeof(T);
  const int cv = C / VEC;
  const long long total = (long long)B * (L + 2 * pad) * cv;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % cv);
    const long long r = idx / cv;
    const int j = (int)(r % (L + 2 * pad)), b = (int)(r / (L + 2 * pad));
    int i = j - pad;
    bool zero = false;
    if (i < 0 || i >= L) {
      if (reflect) i = i < 0 ? -i : 2 * (L - 1) - i;
      else zero = true;
    }
    u32x4 v = {0u, 0u, 0u, 0u};
    if (!zero) {
      v = *(const u32x4*)(x + ((long long)b * L + i) * C + (long long)c * VEC);
      if (slope != 1.0f) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float lo = half_lo(v[e]), hi = half_hi(v[e]);
            lo = lo >= 0.f ? lo : lo * slope;
            hi = hi >= 0.f ? hi : hi * slope;
            v[e] = pack_bf16x2(lo, hi);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float f = __uint_as_float(v[e]);
            v[e] = __float_as_uint(f >= 0.f ? f : f * slope);
          }
        }
      }
    }
    *(u32x4*)(y + ((long long)b * (L + 2 * pad) + j) * C + (long long)c * VEC) = v;
  }
}

// y[b, l] = tanh?( bias + sum_{t < K, c < C} xp[b, l + t, c] * w[t * C + c] ),  xp (B, L + K - 1, C) already padded.
// One wave per output sample group: lanes split the K*C products of 1 sample; 4 samples per 256-thread block.
template <typename T>
__global__ __launch_bounds__(256) void conv1d_out1_kernel(const T* __restrict__ xp, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int B,
                                                          int L, int C, int K, int do_tanh) {
  const int lane = threadIdx.x & 63;
  const long long s = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= (long long)B * L) return;
  const int b = (int)(s / L), l = (int)(s % L);
  const T* base = xp + ((long long)b * (L + K - 1) + l) * C;  // K consecutive rows = K*C contiguous elements
  float acc = 0.f;
  for (int i = lane; i < K * C; i += 64) acc = fmaf(vld(base + i), w[i], acc);
  acc = wave_sum(acc);
  if (lane == 0) {
    float v = acc + (bias ? bias[0] : 0.f);
    y[s] = do_tanh ? tanhf(v) : v;
  }
}

}  // namespace

extern "C" int melgpt_pad1d_act(const void* x, void* y, int B, int L, int C, int pad, int reflect, float slope, int dtype,
                                void* stream) {
  MELGPT_CHECK(x && y && B > 0 && L > 0 && C > 0 && pad >= 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(!reflect || pad < L, MELGPT_ERR_BAD_ARG);
  const int vec = dtype == MELGPT_F32 ? 4 : 8;
  MELGPT_CHECK(C % vec == 0 && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0), MELGPT_ERR_ALIGN);
  const long long total = (long long)B * (L + 2 * pad) * (C / vec);
  const unsigned grid = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MELGPT_F32)
    hipLaunchKernelGGL(pad1d_act_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, B, L, C, pad,
                       reflect, slope);
  else
    hipLaunchKernelGGL(pad1d_act_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, L, C, pad,
                       reflect, slope);
  return melgpt_launch_status();
}

extern "C" int melgpt_conv1d_out1(const void* xp, const float* w, const float* bias, float* y, int B, int L, int C, int K,
                                  int do_tanh, int dtype, void* stream) {
  MELGPT_CHECK(xp && w && y && B > 0 && L > 0 && C > 0 && K > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  const long long n = (long long)B * L;
  MELGPT_CHECK(n < 0x7FFFFFFFLL * 4, MELGPT_ERR_UNSUPPORTED);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MELGPT_F32)
    hipLaunchKernelGGL(conv1d_out1_kernel<float>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, (const float*)xp, w, bias,
                       y, B, L, C, K, do_tanh);
  else
    hipLaunchKernelGGL(conv1d_out1_kernel<bf16_t>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, (const bf16_t*)xp, w,
                       bias, y, B, L, C, K, do_tanh);
  return melgpt_launch_status();
}
