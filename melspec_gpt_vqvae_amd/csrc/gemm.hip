// Tiled MFMA GEMM for gfx950 with fused epilogues - the workhorse behind every nn.Linear of minGPT
// (reference transformer/minGPT.py:56-63,76-78,88,100-105,149,188), their backward passes, and the
// VQ-VAE convolutions (implicit GEMM over NHWC activations; reference vqvae/big_model_attn_gan.py:85-99,
// 151-159,176-186,203-207,247-251,403-422,578-579).
//
//   C[m,n] = epilogue( alpha * sum_k A(m,k) * B(n,k) )          (optionally batched over blockIdx.z)
//
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 = 4x4 MFMA tiles of 16x16),
//     K step = 128 bytes per row (64 bf16 / 32 f32), LDS double buffer (2 x 32 KiB -> 2 workgroups / CU).
//   * operands are staged global -> VGPR -> LDS with 16-byte buffer loads whose out-of-range lanes return 0
//     (edges, conv padding and ragged K cost nothing); the loads of tile t+1 are issued before the MFMAs
//     of tile t and written to LDS after them.
//   * layouts: ROW = reduction index contiguous, read with ds_read_b128 from an XOR-swizzled image
//              KMAJ = reduction index strided (transposed operand): bf16 uses ds_read_b64_tr_b16
//                     (hardware transpose), f32 plain b32 reads - no transposed copies are ever made
//              CONV = implicit im2col of an NHWC activation (3x3 / 1x1, stride 1|2, asymmetric pad,
//                     optional nearest x2 upsample folded into the addressing)
//   * MFMA roles are swapped (weights/B rows feed the A port) so that each lane ends up holding 4
//     CONSECUTIVE output columns of one output row -> vector stores, float4 bias loads and one Philox4
//     call per 4 dropout decisions.
//   * T = bf16 : v_mfma_f32_16x16x32_bf16, f32 accumulate.   T = f32 : v_mfma_f32_16x16x4_f32 (exact f32
//     FMA chain) - the parity lane.  Fragment addressing is identical for both.
//   * workgroup -> tile mapping is XCD-aware (consecutive tiles of one A row-panel share an L2).
#include <math.h>
#include <stdlib.h>

#include "gemm_common.h"

using namespace gemmk;

namespace {

constexpr int BM = 128, BN = 128, TILE_BYTES = 16384;

// ------------------------------------------------------------------------------- LDS addressing
template <typename T>
__device__ __forceinline__ int kmaj_off(int krow, int lc) {
  if constexpr (Tr<T>::ES == 2) {
    int s = (krow & 3) | (((krow >> 3) & 1) << 2);
    return krow * 256 + ((lc ^ (s << 1)) << 4);
  } else {
    return krow * 512 + ((lc ^ (((krow >> 2) & 7) << 2)) << 4);
  }
}

// fragment of a 16-wide (m or n) sub-tile `st` for k-substep ks (0/1): 16 bytes per lane
template <typename T, int LAY>
__device__ __forceinline__ u32x4 load_frag(const char* tile, int st, int ks, int lane) {
  const int i = lane & 15, g = lane >> 4;
  if constexpr (LAY != LAY_KMAJ) {
    return *(const u32x4*)(tile + row_off(st * 16 + i, 4 * ks + g));
  } else if constexpr (Tr<T>::ES == 2) {
    const int q = i >> 2, p = i & 3;
    const int k0 = 32 * ks + 8 * g + q;
    const int lc = 2 * st + (p >> 1);
    const int a0 = kmaj_off<T>(k0, lc) + (p & 1) * 8;
    const int a1 = kmaj_off<T>(k0 + 4, lc) + (p & 1) * 8;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + a1));
    s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(u32x4, f);
  } else {
    const int mn = 16 * st + i;
    u32x4 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int krow = 16 * ks + 4 * g + e;
      f[e] = *(const unsigned*)(tile + kmaj_off<T>(krow, mn >> 2) + (mn & 3) * 4);
    }
    return f;
  }
}


// ---------------------------------------------------------------------------------- the kernel
template <typename T, int ALAY, int BLAY>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  constexpr int ES = Tr<T>::ES, KSTEP = Tr<T>::KSTEP;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 16K | B 16K]

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;

  const int tilesN = (p.N + BN - 1) / BN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wg / tilesN) * BM, n0 = (wg % tilesN) * BN;
  const int bz = blockIdx.z;

  const char* Ab = (const char*)p.A + (long long)bz * p.sA * ES;
  const char* Bb = (const char*)p.B + (long long)bz * p.sB * ES;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(Ab, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(Bb, p.b_bytes);

  // ---- per-thread staging plan: 4 chunks of 16 B for A and 4 for B per K step
  unsigned a_base[4], b_base[4];
  int a_lds[4], b_lds[4];
  int cy[4], cx[4];  // conv only: top-left input coordinate of the output pixel
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = t + 256 * i;
    if constexpr (ALAY == LAY_ROW) {
      const int row = q >> 3, ch = q & 7;
      a_base[i] = (unsigned)(((long long)(m0 + row) * p.lda) * ES) + ch * 16;
      a_lds[i] = row_off(row, ch);
    } else if constexpr (ALAY == LAY_KMAJ) {
      constexpr int CPR = 128 * ES / 16;
      const int krow = q / CPR, lc = q % CPR;
      const bool ok = m0 + lc * (16 / ES) < p.M;
      a_base[i] = ok ? (unsigned)(((long long)krow * p.lda + m0) * ES) + lc * 16 : OOB;
      a_lds[i] = kmaj_off<T>(krow, lc);
    } else {
      const int row = q >> 3, ch = q & 7;
      const int m = m0 + row;
      const int ohw = p.OH * p.OW;
      const int bb = m / ohw, rem = m - bb * ohw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      cy[i] = (m < p.M) ? oy * p.cstride - p.pad_t : -100000;
      cx[i] = ox * p.cstride - p.pad_l;
      a_base[i] = (unsigned)((long long)bb * p.cH * p.cW * p.cC * ES) + ch * 16;
      a_lds[i] = row_off(row, ch);
    }
    if constexpr (BLAY == LAY_ROW) {
      const int row = q >> 3, ch = q & 7;
      b_base[i] = (unsigned)(((long long)(n0 + row) * p.ldb) * ES) + ch * 16;
      b_lds[i] = row_off(row, ch);
    } else {
      constexpr int CPR = 128 * ES / 16;
      const int krow = q / CPR, lc = q % CPR;
      const bool ok = n0 + lc * (16 / ES) < p.N;
      b_base[i] = ok ? (unsigned)(((long long)krow * p.ldb + n0) * ES) + lc * 16 : OOB;
      b_lds[i] = kmaj_off<T>(krow, lc);
    }
  }

  // two register sets: while tile t is multiplied, tile t+1 waits in one set (landed during the previous step, it
  // is written to LDS after the MFMAs) and the loads of tile t+2 are issued into the other - every global load
  // gets a full K step + the MFMAs of the next one to land before anything waits on it
  u32x4 ar[2][4], br[2][4];
  auto issue = [&](int kt, auto setc) {
    constexpr int S = decltype(setc)::value;
    const int k0 = kt * KSTEP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned off;
      if constexpr (ALAY == LAY_ROW) {
        const int ch = (t + 256 * i) & 7;
        off = (k0 * ES + ch * 16 < p.K * ES) ? a_base[i] + k0 * ES : OOB;
      } else if constexpr (ALAY == LAY_KMAJ) {
        off = (a_base[i] == OOB) ? OOB : a_base[i] + (unsigned)((long long)k0 * p.lda * ES);
      } else if constexpr (ALAY == LAY_CONV) {
        const int tap = k0 / p.cC, ci0 = k0 - tap * p.cC;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        int iy = cy[i] + ky, ix = cx[i] + kx;
        const bool ok = iy >= 0 && ix >= 0 && iy < (p.cH << p.ups) && ix < (p.cW << p.ups);
        iy >>= p.ups;
        ix >>= p.ups;
        off = ok ? a_base[i] + (unsigned)((iy * p.cW + ix) * p.cC + ci0) * ES : OOB;
      } else {  // LAY_CONV1D (MelGAN): one row (H = 1), taps cdil_m1 + 1 apart, reflection or zeros outside [0, W);
                // Cin need not be a multiple of the K step: a lane's 16-byte chunk carries its own tap (Cin % (16 / ES) == 0)
        const int chl = (t + 256 * i) & 7;
        const int k = k0 + chl * (16 / ES);
        const int tap = k / p.cC, ci = k - tap * p.cC;
        int ix = cx[i] + tap + tap * p.cdil_m1;
        if (p.creflect) {  // nn.ReflectionPad1d: -j -> j, W - 1 + j -> W - 1 - j
          ix = ix < 0 ? -ix : ix;
          ix = ix >= p.cW ? 2 * (p.cW - 1) - ix : ix;
        }
        const bool ok = k < p.K && cy[i] == 0 && ix >= 0 && ix < p.cW;
        off = ok ? a_base[i] - chl * 16 + (unsigned)(ix * p.cC + ci) * ES : OOB;
      }
      ar[S][i] = buf_load16(ra, off);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned off;
      if constexpr (BLAY == LAY_ROW) {
        const int ch = (t + 256 * i) & 7;
        off = (k0 * ES + ch * 16 < p.K * ES) ? b_base[i] + k0 * ES : OOB;
      } else {
        off = (b_base[i] == OOB) ? OOB : b_base[i] + (unsigned)((long long)k0 * p.ldb * ES);
      }
      br[S][i] = buf_load16(rb, off);
    }
  };
  auto commit = [&](int buf, auto setc) {
    constexpr int S = decltype(setc)::value;
    char* sa = smem + buf * 2 * TILE_BYTES;
    char* sb = sa + TILE_BYTES;
    if (ALAY == LAY_CONV1D && p.a_leaky != 0.f) {  // (wave-uniform) LeakyReLU on the A operand: max(x, slope x), 0 < slope < 1; padding zeros stay zeros
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x4 v = ar[S][i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (ES == 4) {
            const float x = __uint_as_float(v[e]);
            v[e] = __float_as_uint(fmaxf(x, x * p.a_leaky));
          } else {
            const float lo = bf16lo(v[e]), hi = bf16hi(v[e]);
            v[e] = pack_bf16x2(fmaxf(lo, lo * p.a_leaky), fmaxf(hi, hi * p.a_leaky));
          }
        }
        *(u32x4*)(sa + a_lds[i]) = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) *(u32x4*)(sa + a_lds[i]) = ar[S][i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(sb + b_lds[i]) = br[S][i];
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int cur) {
    const char* sa = smem + cur * 2 * TILE_BYTES;
    const char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fa[4], fb[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) fa[mt] = load_frag<T, ALAY>(sa, wm * 4 + mt, ks, lane);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) fb[nt] = load_frag<T, BLAY>(sb, wn * 4 + nt, ks, lane);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) mma<T>(acc[mt][nt], fb[nt], fa[mt]);  // rows = n, cols = m
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  const int nk = (p.K + KSTEP - 1) / KSTEP;
  // depth-2 prefetch only where the extra 32 staging VGPRs do not spill (measured on MI355X, profiles/: K-contiguous
  // bf16 operands +8..14 %; the transposed-read variants need the registers for their fragment addressing)
  constexpr bool DEEP = ES == 2 && ALAY != LAY_KMAJ && ALAY != LAY_CONV1D && BLAY != LAY_KMAJ;  // (the 1-D convolution's per-lane taps + activation spill 60 VGPRs with it)
  if constexpr (DEEP) {
    issue(0, S0{});
    commit(0, S0{});
    if (nk > 1) issue(1, S1{});
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      // even step: tile kt in LDS buffer 0, tile kt+1 in register set 1
      if (kt + 2 < nk) issue(kt + 2, S0{});
      compute(0);
      if (kt + 1 < nk) commit(1, S1{});
      __syncthreads();
      if (kt + 1 >= nk) break;
      // odd step: tile kt+1 in LDS buffer 1, tile kt+2 in register set 0
      if (kt + 3 < nk) issue(kt + 3, S1{});
      compute(1);
      if (kt + 2 < nk) commit(0, S0{});
      __syncthreads();
    }
  } else {
    issue(0, S0{});
    commit(0, S0{});
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) issue(kt + 1, S0{});
      compute(cur);
      if (kt + 1 < nk) commit(cur ^ 1, S0{});
      __syncthreads();
    }
  }

  // the last K step ended with a barrier: every wave is done with the operand tiles, LDS can be reused
  epilogue<T, 4, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, bz, lane, smem + w * 4096);
}

template <typename T, int ALAY, int BLAY>
int launch(const GemmParams& p, int batch, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_kernel<T, ALAY, BLAY>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            4 * TILE_BYTES) != hipSuccess)
      return MELGPT_ERR_LAUNCH;
    attr = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL((gemm_kernel<T, ALAY, BLAY>), dim3(tiles, 1, batch), dim3(256), 4 * TILE_BYTES, s, p);
  return melgpt_launch_status();
}

template <typename T>
int dispatch(const GemmParams& p, int alay, int blay, int batch, hipStream_t s) {
  if (alay == LAY_ROW && blay == LAY_ROW) return launch<T, LAY_ROW, LAY_ROW>(p, batch, s);
  if (alay == LAY_ROW && blay == LAY_KMAJ) return launch<T, LAY_ROW, LAY_KMAJ>(p, batch, s);
  if (alay == LAY_KMAJ && blay == LAY_KMAJ) return launch<T, LAY_KMAJ, LAY_KMAJ>(p, batch, s);
  if (alay == LAY_KMAJ && blay == LAY_ROW) return launch<T, LAY_KMAJ, LAY_ROW>(p, batch, s);
  if (alay == LAY_CONV && blay == LAY_ROW) return launch<T, LAY_CONV, LAY_ROW>(p, batch, s);
  if (alay == LAY_CONV1D && blay == LAY_ROW) return launch<T, LAY_CONV1D, LAY_ROW>(p, batch, s);
  return MELGPT_ERR_UNSUPPORTED;
}

bool fits32(long long rows, long long ld, int es) { return (rows + 260) * ld * es < 0xFFFFFF00LL; }

// tile configuration for the bf16 lane: 1 = 128x128 (this file), 2 = 256x128, 3 = 256x256 (gemm256.hip).
int pick_tile(const GemmParams& p, int batch, int kmin = 256, int kmin_partial = 512) {
  // The persistent 256 x 256 kernel (gemm256.hip) wins once there are enough tiles to occupy the chip and the
  // tile grid is not mostly padding; small or skinny problems stay on the 128 x 128 kernel (two workgroups per CU).
  const long long tiles256 = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256) * batch;
  const double fill = (double)p.M * p.N / ((double)((p.M + 191) / 192) * 192.0 * ((p.N + 255) / 256) * 256.0);
  if (tiles256 >= 192 && fill >= 0.8 && p.K >= kmin) return 3;
  // Fewer tiles than CUs but a LONG K loop (the VQ-VAE's 5 x 53 / 512-channel layers at 64 clips: M = 16 960, N = 512,
  // K = 4 608 -> 178 tiles of 192 x 256): one partial round of the persistent kernel (72 K units x 2.9 k cycles = 92 us)
  // still beats this kernel's 532 workgroups at 263 TFLOP/s (304 us per layer, 16 % of the decoder at batch 64).
  const long long tiles192 = (long long)((p.M + 191) / 192) * ((p.N + 255) / 256) * batch;
  // (80 tiles: Encoder.conv_out at 64 clips - 512 -> 256 channels, 89 tiles - 94 -> 80 us)
  // (the ping-pong loop's K tile is a third shorter: K >= 512 pays there; launches that will take the RING loop - the switch
  // off, or claimed tiles on - keep the ring's threshold)
  if (tiles192 >= 80 && fill >= 0.8 && p.K >= (melgpt_get_gemm_pingpong() ? kmin_partial : 1024)) return 3;
  return 1;
}

}  // namespace

static int gemm_impl(const void* A, int a_kmajor, long long lda, long long strideA, const void* B,
                     int b_kmajor, long long ldb, long long strideB, void* C, long long ldc,
                     long long strideC, int M, int N, int K, int batch, int dtype, int out_f32,
                     int accumulate, float alpha, const float* bias, int act, const void* R,
                     long long ldr, long long strideR, void* C2, float drop_p,
                     unsigned long long seed, unsigned stream_id, float* a_rowsum, long long ld_rowsum, void* stream) {
  MELGPT_CHECK(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  const int es = dtype == MELGPT_F32 ? 4 : 2, vec = 16 / es;
  MELGPT_CHECK(act >= MELGPT_ACT_NONE && act <= MELGPT_ACT_MUL, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK((act != MELGPT_ACT_GELU_GRAD && act != MELGPT_ACT_MUL) || R, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(act != MELGPT_ACT_GELU_DACT || C2, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(drop_p >= 0.f && drop_p < 1.f, MELGPT_ERR_BAD_ARG);
  // 16-byte vector accesses everywhere
  // K only has to be a whole number of 16-byte chunks for operands whose reduction index is contiguous
  MELGPT_CHECK((a_kmajor && b_kmajor) || K % vec == 0, MELGPT_ERR_ALIGN);
  MELGPT_CHECK(N % 4 == 0 && lda % vec == 0 && ldb % vec == 0 && ldc % 4 == 0 &&
                   strideA % vec == 0 && strideB % vec == 0 && strideC % 4 == 0,
               MELGPT_ERR_ALIGN);
  MELGPT_CHECK(!a_kmajor || M % vec == 0, MELGPT_ERR_ALIGN);
  MELGPT_CHECK(!b_kmajor || N % vec == 0, MELGPT_ERR_ALIGN);
  MELGPT_CHECK((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)R | (uintptr_t)C2 | (uintptr_t)bias) & 15) == 0,
               MELGPT_ERR_ALIGN);
  MELGPT_CHECK(!R || (ldr % 4 == 0 && strideR % 4 == 0), MELGPT_ERR_ALIGN);
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.C2 = C2; p.bias = bias; p.R = R;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
  p.sA = strideA; p.sB = strideB; p.sC = strideC; p.sR = strideR;
  const long long a_rows = a_kmajor ? K : M, a_cols = a_kmajor ? M : K;
  const long long b_rows = b_kmajor ? K : N, b_cols = b_kmajor ? N : K;
  MELGPT_CHECK(lda >= a_cols && ldb >= b_cols && ldc >= N, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(fits32(a_rows, lda, es) && fits32(b_rows, ldb, es), MELGPT_ERR_UNSUPPORTED);
  p.a_bytes = (unsigned)(((a_rows - 1) * lda + a_cols) * es);
  p.b_bytes = (unsigned)(((b_rows - 1) * ldb + b_cols) * es);
  p.out_f32 = out_f32; p.accumulate = accumulate; p.act = act; p.alpha = alpha;
  {
    const int oes = (out_f32 || dtype == MELGPT_F32) ? 4 : 2;
    const bool c_ok = (ldc * oes) % 16 == 0 && (strideC * oes) % 16 == 0 && N % (16 / oes) == 0;
    const bool r_ok = !R || ((ldr * es) % 16 == 0 && (strideR * es) % 16 == 0 && N % vec == 0);
    p.vec_io = c_ok && r_ok;
  }
  if (drop_p > 0.f) {
    p.drop_scale = 1.0f / (1.0f - drop_p);
    double th = (double)drop_p * 4294967296.0;
    p.drop_thresh = th >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)th;
  }
  p.seed = seed; p.stream_id = stream_id;
  hipStream_t s = (hipStream_t)stream;
  const int alay = a_kmajor ? LAY_KMAJ : LAY_ROW, blay = b_kmajor ? LAY_KMAJ : LAY_ROW;
  if (a_rowsum) {  // only the persistent kernel's weight-gradient instantiation carries it: nothing is launched otherwise
    MELGPT_CHECK(dtype == MELGPT_BF16 && ld_rowsum >= M, MELGPT_ERR_UNSUPPORTED);
    if (pick_tile(p, batch) != 3) return MELGPT_ERR_UNSUPPORTED;
    p.a_rowsum = a_rowsum;
    p.ld_rowsum = ld_rowsum;
    return launch_gemm256(p, alay, blay, batch, 3, s);
  }
  if (dtype == MELGPT_F32) return dispatch<float>(p, alay, blay, batch, s);
  const int cfg = pick_tile(p, batch);
  if (cfg != 1) {
    int st = launch_gemm256(p, alay, blay, batch, cfg, s);
    if (st != MELGPT_ERR_UNSUPPORTED) return st;
  }
  return dispatch<bf16_t>(p, alay, blay, batch, s);
}

extern "C" int melgpt_gemm(const void* A, int a_kmajor, long long lda, long long strideA, const void* B,
                           int b_kmajor, long long ldb, long long strideB, void* C, long long ldc,
                           long long strideC, int M, int N, int K, int batch, int dtype, int out_f32,
                           int accumulate, float alpha, const float* bias, int act, const void* R,
                           long long ldr, long long strideR, void* C2, float drop_p,
                           unsigned long long seed, unsigned stream_id, void* stream) {
  return gemm_impl(A, a_kmajor, lda, strideA, B, b_kmajor, ldb, strideB, C, ldc, strideC, M, N, K, batch, dtype, out_f32,
                   accumulate, alpha, bias, act, R, ldr, strideR, C2, drop_p, seed, stream_id, nullptr, 0, stream);
}

extern "C" int melgpt_wgrad_rowsum_rows(int N, int batch) { return batch * ((N + 255) / 256); }

extern "C" int melgpt_wgrad_rowsum(const void* dY, long long lda, long long strideA, const void* X, long long ldb,
                                   long long strideB, float* dW_part, long long ldc, long long strideC, int M, int N,
                                   int K, int batch, int dtype, float* rowsum_part, long long ld_rowsum, void* stream) {
  MELGPT_CHECK(rowsum_part, MELGPT_ERR_BAD_ARG);
  return gemm_impl(dY, 1, lda, strideA, X, 1, ldb, strideB, dW_part, ldc, strideC, M, N, K, batch, dtype, 1, 0, 1.0f,
                   nullptr, MELGPT_ACT_NONE, nullptr, 0, 0, nullptr, 0.f, 0ULL, 0u, rowsum_part, ld_rowsum, stream);
}

extern "C" int melgpt_conv2d_nhwc(const void* x, int B, int H, int W, int Cin, const void* wpack, int Cout, int KH,
                                  int KW, int stride, int pad_t, int pad_l, int OH, int OW, int upsample,
                                  const float* bias, const void* residual, void* y, int dtype, void* stream) {
  MELGPT_CHECK(x && wpack && y && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && OH > 0 && OW > 0,
               MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  const int es = dtype == MELGPT_F32 ? 4 : 2, kstep = dtype == MELGPT_F32 ? 32 : 64;
  MELGPT_CHECK((KH == 3 && KW == 3) || (KH == 1 && KW == 1), MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(Cin % kstep == 0 && Cout % 4 == 0, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(stride == 1 || stride == 2, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(upsample == 0 || upsample == 1, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK((((uintptr_t)x | (uintptr_t)wpack | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)bias) & 15) == 0,
               MELGPT_ERR_ALIGN);
  const long long in_bytes = (long long)B * H * W * Cin * es;
  const long long M = (long long)B * OH * OW;
  MELGPT_CHECK(in_bytes < 0xFFFFFF00LL && M < 0x7FFFFF00LL, MELGPT_ERR_UNSUPPORTED);
  GemmParams p{};
  p.A = x; p.B = wpack; p.C = y; p.bias = bias; p.R = residual;
  p.M = (int)M; p.N = Cout; p.K = KH * KW * Cin;
  p.lda = Cin; p.ldb = p.K; p.ldc = Cout; p.ldr = Cout;
  p.a_bytes = (unsigned)in_bytes;
  p.b_bytes = (unsigned)((long long)Cout * p.K * es);
  p.alpha = 1.0f;
  p.vec_io = (Cout * es) % 16 == 0;
  p.cH = H; p.cW = W; p.cC = Cin; p.OH = OH; p.OW = OW; p.cstride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.ups = upsample; p.KW = KW;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MELGPT_F32) return dispatch<float>(p, LAY_CONV, LAY_ROW, 1, s);
  if (Cout <= 128 && p.K >= 512 && p.vec_io) {
    // at most 128 output channels (the stride-2 Downsample layers): 256 x 128 tiles of the ping-pong loop (gemm8p.hip),
    // whose 32-bit operand offsets end at 2 GiB - whole images per launch (128 tiles of 80 x 848 x 128: two launches)
    const long long per_img = (long long)H * W * Cin * es;
    const int nlaunch = (int)((in_bytes + 0x7FFFFFFFLL - 1) / 0x7FFFFFFFLL);
    const int per = (B + nlaunch - 1) / nlaunch;
    if (per_img < 0x7FFFFFFFLL && per * per_img < 0x7FFFFFFFLL) {
      for (int b0 = 0; b0 < B; b0 += per) {
        const int bn = B - b0 < per ? B - b0 : per;
        GemmParams q = p;
        q.A = (const char*)x + b0 * per_img;
        q.C = (char*)y + (long long)b0 * OH * OW * Cout * es;
        if (residual) q.R = (const char*)residual + (long long)b0 * OH * OW * Cout * es;
        q.M = bn * OH * OW;
        q.a_bytes = (unsigned)(bn * per_img);
        int st = launch_conv8p_n128(q, s);
        if (st == MELGPT_ERR_UNSUPPORTED) st = dispatch<bf16_t>(q, LAY_CONV, LAY_ROW, 1, s);
        if (st != MELGPT_OK) return st;
      }
      return MELGPT_OK;
    }
  }
  // (1 x 1 convolutions with K = 128 - the 128 -> 256 nin_shortcut at 20 x 212 - are pure streams: 208 MB at 64 tiles; the
  // persistent kernel's request stream runs across tile boundaries, the 128 x 128 kernel's workgroups each wait out their own
  // load -> multiply -> store chain: 92 us there)
  const int cfg = pick_tile(p, 1, 128, 256);   // (... and the 256 -> 512 nin_shortcut at 5 x 53: one partial round of four K tiles)
  if (cfg != 1) {
    int st = launch_gemm256(p, LAY_CONV, LAY_ROW, 1, cfg, s);
    if (st != MELGPT_ERR_UNSUPPORTED) return st;
  }
  return dispatch<bf16_t>(p, LAY_CONV, LAY_ROW, 1, s);
}

// MelGAN's Conv1d / ConvTranspose1d layers (vocoder/modules.py:23-79) as ONE implicit GEMM each on a channels-last
// (B, L, Cin) activation: y[b, l, :] (row stride ldy) = bias + sum_{t < KW} W_t . f(x[b, l - pad_l + t dilation, :]) (+ residual)
// (+ y when accumulate), f = LeakyReLU(in_slope) (in_slope = 0: none); out_slope != 0: LeakyReLU(out_slope) on bias + sum
// before the residual (the block's conv3 hands conv1 an activated t1); positions outside [0, L): reflected
// (nn.ReflectionPad1d) or zero.  wpack (Cout, KW * Cin), tap-major along K.  Generic 128 x 128 kernel (the operand passes
// through registers on its way into LDS, where the activation is applied); Cin % (16 / es) == 0.
extern "C" int melgpt_conv1d_nlc(const void* x, int B, int L, int Cin, const void* wpack, int Cout, int KW, int dilation,
                                 int pad_l, int reflect, float in_slope, const float* bias, const void* residual,
                                 long long ldr, int accumulate, float out_slope, void* y, long long ldy, int dtype,
                                 void* stream) {
  MELGPT_CHECK(x && wpack && y && B > 0 && L > 0 && Cin > 0 && Cout > 0 && KW > 0 && dilation > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  const int es = dtype == MELGPT_F32 ? 4 : 2;
  MELGPT_CHECK(Cin % (16 / es) == 0 && Cout % (16 / es) == 0 && in_slope >= 0.f && in_slope < 1.f && out_slope >= 0.f &&
                   out_slope < 1.f,
               MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(!reflect || (pad_l < L && (KW - 1) * dilation - pad_l < L), MELGPT_ERR_UNSUPPORTED);  // one reflection only
  MELGPT_CHECK((((uintptr_t)x | (uintptr_t)wpack | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)bias) & 15) == 0,
               MELGPT_ERR_ALIGN);
  MELGPT_CHECK((ldy * es) % 16 == 0 && (!residual || (ldr * es) % 16 == 0), MELGPT_ERR_ALIGN);
  const long long in_bytes = (long long)B * L * Cin * es, M = (long long)B * L;
  MELGPT_CHECK(in_bytes < 0xFFFFFF00LL && M < 0x7FFFFF00LL, MELGPT_ERR_UNSUPPORTED);
  GemmParams p{};
  p.A = x; p.B = wpack; p.C = y; p.bias = bias; p.R = residual;
  p.M = (int)M; p.N = Cout; p.K = KW * Cin;
  p.lda = Cin; p.ldb = p.K; p.ldc = ldy; p.ldr = ldr;
  p.a_bytes = (unsigned)in_bytes;
  p.b_bytes = (unsigned)((long long)Cout * p.K * es);
  p.alpha = 1.0f;
  p.accumulate = accumulate;
  p.vec_io = 1;
  p.cH = 1; p.cW = L; p.cC = Cin; p.OH = 1; p.OW = L; p.cstride = 1; p.pad_t = 0; p.pad_l = pad_l; p.ups = 0; p.KW = KW;
  p.cdil_m1 = dilation - 1; p.creflect = reflect; p.a_leaky = in_slope; p.out_leaky = out_slope;
  hipStream_t s = (hipStream_t)stream;
  return dtype == MELGPT_F32 ? dispatch<float>(p, LAY_CONV1D, LAY_ROW, 1, s) : dispatch<bf16_t>(p, LAY_CONV1D, LAY_ROW, 1, s);
}
