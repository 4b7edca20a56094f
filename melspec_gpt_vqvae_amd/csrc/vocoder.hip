// MelGAN generator glue kernels (reference vocoder/modules.py:24-79).  The generator's Conv1d /
// ConvTranspose1d layers run as per-tap batched GEMMs on shifted row windows of a channels-last (B, L, C) activation
// (melgpt_gemm with `accumulate`; the host side is melspec_gpt_vqvae_amd/vocoder/modules.py), so the only new device
// code is: (1) the padded copy that also applies the LeakyReLU(0.2) which precedes every convolution
// (nn.ReflectionPad1d / zero rows for the transposed convolutions), (2) the final 7-tap convolution to ONE channel
// followed by tanh.  Both are one pass over the activation (HBM bound).
#include "mma.h"

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

// The generator's output layer in ONE pass over the activation h (B, L, C), 16-bit lane:
//   y[b, l] = tanh( bias + sum_{t < K, c < C} w[t, c] * leaky(h[b, refl(l + t - K/2), c]) )
// (LeakyReLU(0.2) -> ReflectionPad1d(3) -> WNConv1d(ngf, 1, 7) -> Tanh, vocoder/modules.py:72-77).  A workgroup takes 256
// consecutive samples of one clip: their 256 + K - 1 rows go through LDS once (coalesced 16-byte loads, activation and
// reflection applied on the way, rows padded to C * 2 + 16 bytes so that the per-thread row reads spread over the banks),
// the K * C weights sit in LDS as f32 and are read as broadcasts.  (The first version - padded copy, then one WAVE per
// output sample - took 367 us for 8 clips whose activation is 111 MB: 17 x the HBM time.)
template <int C>
__global__ __launch_bounds__(256) void conv1d_out1_fused_kernel(const bf16_t* __restrict__ h, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ y, int L,
                                                                int K, float slope, int do_tanh) {
  constexpr int ROWB = C * 2 + 16, CPR = C / 8;  // bytes per LDS row, 16-byte chunks per row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ws = (float*)smem;                     // [K * C]
  char* rows = smem + ((K * C * 4 + 15) & ~15);  // [256 + K - 1][ROWB]
  const int t = threadIdx.x, b = blockIdx.y, l0 = blockIdx.x * 256, half = K / 2;
  for (int i = t; i < K * C; i += 256) ws[i] = w[i];
  const int nrows = 256 + K - 1;
  for (int q = t; q < nrows * CPR; q += 256) {
    const int r = q / CPR, c = q - r * CPR;
    int i = l0 + r - half;
    i = i < 0 ? -i : i;
    i = i >= L ? 2 * (L - 1) - i : i;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (i >= 0 && i < L) {  // (rows past the clip's last tile fall outside even after one reflection: unused)
      v = *(const u32x4*)(h + ((long long)b * L + i) * C + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = half_lo(v[e]), hi = half_hi(v[e]);
        v[e] = pack_bf16x2(fmaxf(lo, lo * slope), fmaxf(hi, hi * slope));
      }
    }
    *(u32x4*)(rows + r * ROWB + c * 16) = v;
  }
  __syncthreads();
  const int l = l0 + t;
  if (l >= L) return;
  float acc0 = 0.f, acc1 = 0.f;
  for (int tp = 0; tp < K; ++tp) {
    const char* row = rows + (t + tp) * ROWB;
    const float* wr = ws + tp * C;
#pragma unroll
    for (int c = 0; c < CPR; ++c) {
      const u32x4 v = *(const u32x4*)(row + c * 16);
      const f32x4 w0 = *(const f32x4*)(wr + c * 8), w1 = *(const f32x4*)(wr + c * 8 + 4);
      acc0 = fmaf(half_lo(v[0]), w0[0], acc0);
      acc1 = fmaf(half_hi(v[0]), w0[1], acc1);
      acc0 = fmaf(half_lo(v[1]), w0[2], acc0);
      acc1 = fmaf(half_hi(v[1]), w0[3], acc1);
      acc0 = fmaf(half_lo(v[2]), w1[0], acc0);
      acc1 = fmaf(half_hi(v[2]), w1[1], acc1);
      acc0 = fmaf(half_lo(v[3]), w1[2], acc0);
      acc1 = fmaf(half_hi(v[3]), w1[3], acc1);
    }
  }
  const float v = acc0 + acc1 + (bias ? bias[0] : 0.f);
  y[(long long)b * L + l] = do_tanh ? tanhf(v) : v;
}

// One MelGAN ResnetBlock (vocoder/modules.py:48-64) of a NARROW stage (dim = 32 or 64) in ONE pass over the activation:
//   y = shortcut(x) + conv1( leaky( conv3_dilated( reflect_pad( leaky(x) ) ) ) )        x, y (B, L, C) channels-last, 16-bit lane
// As separate implicit GEMMs on 128 x 128 tiles these layers are all per-workgroup overhead: N = 32 fills a quarter of a
// tile, K = 96 is two K steps, and 13 568 workgroups each pay launch + load -> LDS -> MFMA -> LDS -> store latency for
// 8 KB of data - 820 us per block at 8 clips whose tensor is 111 MB (45 us of HBM time for one read + one write).
// Here a WAVE owns 16 consecutive positions at a time and walks the tensor with a grid stride; no LDS, no barrier:
//   * every weight lives in REGISTERS as ready-made MFMA fragments (packed on the host in lane order: 10 fragments at
//     dim 32, 40 at dim 64, loaded once per workgroup);
//   * the input fragments come straight from global memory: lane (i16, g) of a 16-row x 32-channel fragment needs the 16
//     bytes of row i16 at channel 8 g - with 64- or 128-byte rows the 16 rows of a tile are ONE contiguous block, so the
//     load is fully coalesced; the three dilated taps are three such loads (reflected at the clip's ends per lane);
//   * LeakyReLU is applied to the fragments in registers;
//   * t1 = conv3(...) never leaves the registers: the MFMA leaves position m = lane & 15 with channels 4 g .. 4 g + 3 of
//     each 16-channel tile in a lane - exactly an operand fragment over positions with the channel pairs of two
//     tiles as its k-slots (k-slot j of lane g = channel 4 g + j, or 16 + 4 g + j - 4 from slot 4 on); the 1 x 1
//     convolution's weights are packed with that same slot order, so no lane exchange is needed (the attention kernels'
//     accumulator-as-operand trick);
//   * the shortcut's 1 x 1 convolution of the raw centre tap accumulates into the same output registers.
template <int C, int RT>  // RT 16-row tiles per wave iteration: their loads are in flight together (a wave that walks the
                          // tensor one tile at a time pays one HBM latency per 2 KB: 1.0 ms per block at 64 clips / dim 64,
                          // 2.8 x its HBM time)
__global__ __launch_bounds__(256) void resblock_narrow_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                              const u32x4* __restrict__ wfrag, const float* __restrict__ b3,
                                                              const float* __restrict__ b1s, int L, int dil,
                                                              long long ntiles, float slope) {
  typedef bf16_t T;
  constexpr int NT = C / 16, KS = C / 32;
  const int lane = threadIdx.x & 63, i16 = lane & 15, g = lane >> 4;
  // ---- weights: [3 KS][NT] conv3, [KS][NT] shortcut, [KS][NT] conv1 fragments, 64 lanes x 16 bytes each
  u32x4 w3[3 * KS][NT], wsc[KS][NT], w1[KS][NT];
#pragma unroll
  for (int k = 0; k < 3 * KS; ++k)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) w3[k][nt] = wfrag[(k * NT + nt) * 64 + lane];
#pragma unroll
  for (int k = 0; k < KS; ++k)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      wsc[k][nt] = wfrag[((3 * KS + k) * NT + nt) * 64 + lane];
      w1[k][nt] = wfrag[((4 * KS + k) * NT + nt) * 64 + lane];
    }
  f32x4 bias3[NT], bias1s[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    bias3[nt] = *(const f32x4*)(b3 + 16 * nt + 4 * g);
    bias1s[nt] = *(const f32x4*)(b1s + 16 * nt + 4 * g);
  }
  auto leaky_frag = [&](u32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = half_lo(v[e]), hi = half_hi(v[e]);
      v[e] = pack_bf16x2(fmaxf(lo, lo * slope), fmaxf(hi, hi * slope));
    }
    return v;
  };
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  const long long ngroups = (ntiles + RT - 1) / RT;
  // the NEXT group's fragments are requested before the current group is computed (a lone wave per SIMD - dim 64 - has
  // nobody else to cover its loads): two register sets, rotated by copies
  auto request = [&](long long grp, u32x4 (&xr)[RT][3][KS], long long (&r0)[RT]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const long long tile = grp * RT + rt < ntiles ? grp * RT + rt : ntiles - 1;  // (a group's tail repeats the last tile)
      r0[rt] = tile * 16;                      // L % 16 == 0: a tile never straddles two clips
      const long long clip0 = r0[rt] / L * L;  // first row of the tile's clip
      const int l = (int)(r0[rt] - clip0) + i16;
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        int r = l + (tap - 1) * dil;
        r = r < 0 ? -r : r;
        r = r >= L ? 2 * (L - 1) - r : r;
        const T* src = x + (clip0 + r) * C + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xr[rt][tap][ks] = *(const u32x4*)(src + 32 * ks);
      }
    }
  };
  u32x4 xa[RT][3][KS], xn[RT][3][KS];
  long long row0[RT], rown[RT];
  if (wave < ngroups) request(wave, xa, row0);
  for (long long grp = wave; grp < ngroups; grp += nwaves) {
    request(grp + nwaves < ngroups ? grp + nwaves : ngroups - 1, xn, rown);  // (past the end: a redundant reload)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 t1[NT], acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        t1[nt] = bias3[nt];
        acc[nt] = bias1s[nt];
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)  // shortcut on the raw centre tap
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mma<T>(acc[nt], wsc[ks][nt], xa[rt][1][ks]);
#pragma unroll
      for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const u32x4 xl = leaky_frag(xa[rt][tap][ks]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) mma<T>(t1[nt], w3[tap * KS + ks][nt], xl);
        }
#pragma unroll
      for (int p = 0; p < KS; ++p) {  // channels 32 p .. 32 p + 31 of leaky(t1) as one operand fragment
        f32x4 lo = t1[2 * p], hi = t1[2 * p + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo[e] = fmaxf(lo[e], lo[e] * slope);
          hi[e] = fmaxf(hi[e], hi[e] * slope);
        }
        const u32x4 pf = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mma<T>(acc[nt], w1[p][nt], pf);
      }
      T* dst = y + (row0[rt] + i16) * C + 4 * g;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *(u32x2*)(dst + 16 * nt) = u32x2{pack_bf16x2(acc[nt][0], acc[nt][1]), pack_bf16x2(acc[nt][2], acc[nt][3])};
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      row0[rt] = rown[rt];
#pragma unroll
      for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xa[rt][tap][ks] = xn[rt][tap][ks];
    }
  }
}

// The same block at dim = 128 (the generator's second stage): its 160 weight fragments are 160 KiB - EXACTLY the LDS, and
// the kernel needs no other byte of it (operands come straight from global memory, t1 stays in registers).  One
// 256-thread workgroup per CU holds them for the whole launch; a wave owns 64 positions at a time (four 16-row tiles), so
// that every weight fragment it reads from LDS feeds four MFMAs: 160 KiB of fragment reads per 64 rows per wave is
// 20 LDS cycles per row per CU against 40 of MFMA issue and 26 of HBM time - matrix-pipe-bound where the three separate
// implicit GEMMs were per-workgroup-latency-bound (380 us per block at 8 clips; 222 MB in and out).  t1 (128 registers) is
// packed into operand fragments (64) before the output accumulators (128) take its place.
__global__ __launch_bounds__(256) void resblock128_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                          const u32x4* __restrict__ wfrag, const float* __restrict__ b3,
                                                          const float* __restrict__ b1s, int L, int dil, long long ntiles,
                                                          float slope) {
  typedef bf16_t T;
  constexpr int C = 128, NT = 8, KS = 4, RT = 4, NF = 5 * KS * NT;  // 160 fragments of 1 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, i16 = lane & 15, g = lane >> 4;
  for (int f = threadIdx.x; f < NF * 64; f += 256) *(u32x4*)(smem + (size_t)f * 16) = wfrag[f];
  __syncthreads();
  auto wld = [&](int f) { return *(const u32x4*)(smem + (size_t)f * 1024 + lane * 16); };
  auto leaky_frag = [&](u32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = half_lo(v[e]), hi = half_hi(v[e]);
      v[e] = pack_bf16x2(fmaxf(lo, lo * slope), fmaxf(hi, hi * slope));
    }
    return v;
  };
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  for (long long tile = wave; tile < ntiles; tile += nwaves) {  // 64 rows; L % 64 == 0: never across two clips
    const long long row0 = tile * 64;
    const long long clip0 = row0 / L * L;
    const int l = (int)(row0 - clip0) + i16;
    // ---- phase A: t1 = conv3(reflect_pad(leaky(x))) for the 64 rows (128 accumulator registers).  ALL 48 operand
    // fragments of the tile (3 taps x 4 k-steps x 4 row tiles, 192 registers) are requested before the first is used:
    // requested step by step, the lone wave of a SIMD paid one HBM latency per k-step - 12 per tile, 1.0 ms per block
    // at 64 clips against 0.23 ms of MFMA issue.
    u32x4 xa[3][KS][RT];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        int r = l + 16 * rt + (tap - 1) * dil;
        r = r < 0 ? -r : r;
        r = r >= L ? 2 * (L - 1) - r : r;
        const T* src = x + (clip0 + r) * C + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xa[tap][ks][rt] = *(const u32x4*)(src + 32 * ks);
      }
    f32x4 t1[RT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {  // (biases re-read per tile - L1 hits - instead of registers held for the launch)
      const f32x4 c3 = *(const f32x4*)(b3 + 16 * nt + 4 * g);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) t1[rt][nt] = c3;
    }
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        u32x4 xl[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) xl[rt] = leaky_frag(xa[tap][ks][rt]);
        u32x4 wv[NT];  // all eight fragment reads of the k-step, then its 32 MFMAs
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wv[nt] = wld((tap * KS + ks) * NT + nt);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) mma<T>(t1[rt][nt], wv[nt], xl[rt]);
        __builtin_amdgcn_sched_barrier(0);  // (one k-step at a time: weight reads hoisted across steps end in scratch)
      }
    }
    // ---- phase B: leaky(t1) as operand fragments (channels 32 p .. 32 p + 31: see resblock_narrow_kernel); t1 is dead
    // after this - the output accumulators take its registers (two 128-register sets at once spilled 100)
    u32x4 pf[KS][RT];
#pragma unroll
    for (int p = 0; p < KS; ++p)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        f32x4 lo = t1[rt][2 * p], hi = t1[rt][2 * p + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo[e] = fmaxf(lo[e], lo[e] * slope);
          hi[e] = fmaxf(hi[e], hi[e] * slope);
        }
        pf[p][rt] = u32x4{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
      }
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase C: y = shortcut(x) + conv1(leaky(t1)) + biases
    f32x4 acc[RT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 c1 = *(const f32x4*)(b1s + 16 * nt + 4 * g);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][nt] = c1;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {  // (the raw centre-tap fragments are still in registers)
      u32x4 wv[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wv[nt] = wld((3 * KS + ks) * NT + nt);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) mma<T>(acc[rt][nt], wv[nt], xa[1][ks][rt]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < KS; ++p) {
      u32x4 wv[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wv[nt] = wld((4 * KS + p) * NT + nt);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) mma<T>(acc[rt][nt], wv[nt], pf[p][rt]);
      __builtin_amdgcn_sched_barrier(0);  // (one channel group at a time: hoisted weight reads end in scratch)
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      __builtin_amdgcn_sched_barrier(0);
      T* dst = y + (row0 + 16 * rt + i16) * C + 4 * g;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *(u32x2*)(dst + 16 * nt) = u32x2{pack_bf16x2(acc[rt][nt][0], acc[rt][nt][1]), pack_bf16x2(acc[rt][nt][2], acc[rt][nt][3])};
    }
  }
}

}  // namespace

// One ResnetBlock of a narrow MelGAN stage: y (B, L, C) = shortcut(x) + conv1(leaky(conv3_dil(reflect_pad(leaky(x))))).
// wfrag: the three weight matrices as MFMA fragments in lane order (vocoder/modules.py packs them: 5 * (C/32) * (C/16)
// fragments of 64 lanes x 16 bytes); b3 = conv3's bias, b1s = conv1's + the shortcut's.  16-bit lane, C in {32, 64},
// L % 16 == 0, dil < L; anything else: MELGPT_ERR_UNSUPPORTED (callers then run the three convolutions separately).
extern "C" int melgpt_resblock_narrow(const void* x, void* y, const void* wfrag, const float* b3, const float* b1s, int B,
                                      int L, int C, int dilation, float slope, int dtype, void* stream) {
  MELGPT_CHECK(x && y && wfrag && b3 && b1s && B > 0 && L > 0 && dilation > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_BF16 && (C == 32 || C == 64 || C == 128) && L % (C == 128 ? 64 : 16) == 0 && dilation < L &&
                   slope > 0.f && slope <= 1.f,
               MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK((((uintptr_t)x | (uintptr_t)y | (uintptr_t)wfrag | (uintptr_t)b3 | (uintptr_t)b1s) & 15) == 0, MELGPT_ERR_ALIGN);
  const long long ntiles = (long long)B * L / 16;
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (C == 128) {  // weights in LDS (all of it), one workgroup per CU, 64-row wave tiles
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)resblock128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return MELGPT_ERR_LAUNCH;
      attr = true;
    }
    const long long nt64 = ntiles / 4, want128 = (nt64 + 3) / 4;
    const unsigned grid128 = (unsigned)(want128 < ncu ? want128 : ncu);
    hipLaunchKernelGGL(resblock128_kernel, dim3(grid128), dim3(256), 160 * 1024, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)y, (const u32x4*)wfrag, b3, b1s, L, dilation, nt64, slope);
    return melgpt_launch_status();
  }
  const long long want = (ntiles + 3) / 4;  // workgroups of four waves
  const long long cap = (long long)ncu * (C == 32 ? 4 : 2);
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  hipStream_t s = (hipStream_t)stream;
  if (C == 32)
    hipLaunchKernelGGL((resblock_narrow_kernel<32, 4>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y,
                       (const u32x4*)wfrag, b3, b1s, L, dilation, ntiles, slope);
  else
    hipLaunchKernelGGL((resblock_narrow_kernel<64, 2>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y,
                       (const u32x4*)wfrag, b3, b1s, L, dilation, ntiles, slope);
  return melgpt_launch_status();
}

// y (B, L) f32 = [tanh]( bias + conv_K( reflect_pad( LeakyReLU_slope( h (B, L, C) ) ) ) ) to ONE channel; w (K * C) f32
// tap-major.  16-bit lane, C in {32, 64}, odd K < L; anything else: MELGPT_ERR_UNSUPPORTED (callers then use
// melgpt_pad1d_act + melgpt_conv1d_out1).
extern "C" int melgpt_conv1d_out1_fused(const void* h, const float* w, const float* bias, float* y, int B, int L, int C,
                                        int K, float slope, int do_tanh, int dtype, void* stream) {
  MELGPT_CHECK(h && w && y && B > 0 && L > 0 && C > 0 && K > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_BF16 && (C == 32 || C == 64) && (K & 1) && K / 2 < L && slope > 0.f && slope <= 1.f,
               MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK((((uintptr_t)h | (uintptr_t)w | (uintptr_t)y) & 15) == 0, MELGPT_ERR_ALIGN);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((L + 255) / 256), (unsigned)B);
  const size_t lds = ((size_t)(K * C * 4 + 15) & ~(size_t)15) + (size_t)(256 + K - 1) * (C * 2 + 16);
  if (C == 32)
    hipLaunchKernelGGL(conv1d_out1_fused_kernel<32>, grid, dim3(256), lds, s, (const bf16_t*)h, w, bias, y, L, K, slope, do_tanh);
  else
    hipLaunchKernelGGL(conv1d_out1_fused_kernel<64>, grid, dim3(256), lds, s, (const bf16_t*)h, w, bias, y, L, K, slope, do_tanh);
  return melgpt_launch_status();
}

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
