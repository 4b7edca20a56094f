// Halo-tiled 3x3 convolution (stride 1, pad 1) with GroupNorm(+swish) fused into its input staging, for the
// ResnetBlocks of the VQ-VAE encoder/decoder on gfx950 (reference vqvae/big_model_attn_gan.py:114-135: norm -> swish
// -> conv; Normalize :139-140; nonlinearity :164-166).
//
// The plain implicit-GEMM kernel (gemm.hip, LAY_CONV) re-gathers its A operand from L2 for each of the 9 taps and
// needs the normalised activation to exist in HBM, i.e. a separate GroupNorm-apply pass that reads and writes the
// whole (B, 80*848, 128) tensor.  Here a workgroup owns an 8 x 16 pixel output tile x 128 output channels and
//   1. stages the (8+2) x (16+2) pixel input patch into LDS ONCE, applying y = swish((x - mean) * rstd * gamma + beta)
//      on the way (zero padding is applied AFTER the normalisation, exactly like the reference); the raw activation
//      is read 1.4x instead of 9x, and the normalised tensor never exists in HBM;
//   2. runs the 9 taps x Cin/64 K steps reading A fragments straight from shifted windows of that patch
//      (16 consecutive pixels of one row = one MFMA row block; pixel-index XOR swizzle keeps ds_read_b128
//      conflict-free), while the weight tile of each K step is double-buffered through LDS like in gemm.hip;
//   3. finishes with the shared fused epilogue (bias, residual x + h of the ResnetBlock, row-contiguous stores).
// MFMA: v_mfma_f32_16x16x32_bf16 (bf16 lane) / v_mfma_f32_16x16x4_f32 (f32 parity lane), same fragment maps as gemm.hip.
#include <cstdlib>

#include "gemm_common.h"

using namespace gemmk;

namespace {

constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, NPIX = PH * PW;  // 180 patch pixels

struct FusedConvParams {
  GemmParams g;          // C, bias, R, ldc, ldr, M (= B*H*W), N (= Cout), B operand = packed weights (Cout, 9*Cin)
  const void* x;         // (B, H, W, Cin) raw activation
  const float* mean;     // (B*32) or null (no normalisation: plain conv)
  const float* rstd;
  const float* gamma;    // (Cin)
  const float* beta;
  int H, W, Cin, swish, tiles_x, tiles_y;
  unsigned x_bytes;      // addressable bytes of x (persistent kernel: patch loads through a buffer resource)
  unsigned r_bytes;      // ... of the residual (persistent kernel: fetched in accumulator layout, see there)
  float* stat_part;      // persistent kernel, optional: (B, tiles_y*tiles_x, 32, 2) sum / sum of squares of the OUTPUT
};

template <typename T>
__device__ __forceinline__ int patch_off(int pix, int chunk, int pix_bytes) {
  return pix * pix_bytes + ((chunk ^ (pix & 15)) << 4);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void conv3x3_gn_kernel(FusedConvParams q) {  // two workgroups per CU: VGPR + AGPR <= 256
  constexpr int ES = Tr<T>::ES, KSTEP = Tr<T>::KSTEP, VEC = 16 / ES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GemmParams& p = q.g;
  const int Cin = q.Cin, pix_bytes = Cin * ES, cpp = pix_bytes / 16;  // chunks per pixel
  char* patch = smem;                                     // [180][Cin] swizzled
  char* wtile = smem + (size_t)NPIX * pix_bytes;          // [2][128 rows x 128 B]
  float* ab = (float*)(wtile + 2 * 16384);                // [Cin][2]: scale, shift of this image's GroupNorm

  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int i16 = lane & 15, g = lane >> 4;
  int tile = blockIdx.x;
  const int tx = tile % q.tiles_x;
  tile /= q.tiles_x;
  const int ty = tile % q.tiles_y, b = tile / q.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = blockIdx.y * 128;

  // ---- per-channel affine of the GroupNorm for image b
  const bool norm = q.mean != nullptr;
  if (norm) {
    const int cg = Cin / 32;
    for (int c = t; c < Cin; c += 256) {
      // all four loads before the first LDS store (which may alias them, as far as the compiler knows: written as
      // a = ...; ab[2c] = a; ab[2c+1] = beta - mean * a it compiled to two dependent round trips)
      const float rs = q.rstd[b * 32 + c / cg], ga = q.gamma[c], be = q.beta[c], me = q.mean[b * 32 + c / cg];
      const float a = rs * ga;
      ab[2 * c] = a;
      ab[2 * c + 1] = be - me * a;
    }
  }
  // ---- weight tile staging: the (128 x 128 B) tile of a K step goes L2 -> LDS by LDS-DMA in sixteen 1 KiB pieces
  // (8 rows x 128 B each, four per wave), no VGPR round trip; the XOR swizzle of row_off() is applied on the source
  // side (the lane fetches the chunk that belongs in its linear LDS position)
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);
  unsigned b_base[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = w + 4 * i, row = piece * 8 + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
    b_base[i] = (n0 + row < p.N) ? (unsigned)(((long long)(n0 + row) * p.ldb) * ES) + ch * 16 : OOB;
  }
  auto issue_b = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(wtile + buf * 16384 + (w + 4 * i) * 1024),
                                               16, b_base[i] == OOB ? OOB : b_base[i] + kt * KSTEP * ES, 0, 0, 0);
  };
  issue_b(0, 0);
  __syncthreads();  // ab[] visible (the fence also drains the DMA)

  // ---- stage the input patch once (normalise + swish on the fly; out-of-image pixels are zeros)
  const T* xb = (const T*)q.x + (long long)b * q.H * q.W * Cin;
  for (int idx = t; idx < NPIX * cpp; idx += 256) {
    const int pix = idx / cpp, ch = idx - pix * cpp;
    const int py = pix / PW, px = pix - py * PW;
    const int iy = y0 + py - 1, ix = x0 + px - 1;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (iy >= 0 && iy < q.H && ix >= 0 && ix < q.W) {
      v = *(const u32x4*)(xb + ((long long)iy * q.W + ix) * Cin + ch * VEC);
      if (norm) {
        float f[VEC];
        if constexpr (ES == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[2 * e] = bf16lo(v[e]);
            f[2 * e + 1] = bf16hi(v[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) f[e] = __uint_as_float(v[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const int c = ch * VEC + e;
          float o = fmaf(f[e], ab[2 * c], ab[2 * c + 1]);
          if (q.swish) o = o * __builtin_amdgcn_rcpf(1.0f + __expf(-o));
          f[e] = o;
        }
        if constexpr (ES == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(f[e]);
        }
      }
    }
    *(u32x4*)(patch + patch_off<T>(pix, ch, pix_bytes)) = v;
  }
  __syncthreads();

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kpt = Cin / KSTEP;  // K steps per tap
  const int nk = 9 * kpt;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue_b(kt + 1, cur ^ 1);  // lands under this step's MFMAs; the barrier below waits for it
    const int tap = kt / kpt, kc = kt - tap * kpt;
    const int ky = tap / 3, kx = tap - ky * 3;
    const char* sb = wtile + cur * 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fa[4], fb[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int pix = (wm * 4 + mt + ky) * PW + i16 + kx;  // output row (wm*4+mt), shifted by the tap
        fa[mt] = *(const u32x4*)(patch + patch_off<T>(pix, kc * 8 + 4 * ks + g, pix_bytes));
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) fb[nt] = *(const u32x4*)(sb + row_off((wn * 4 + nt) * 16 + i16, 4 * ks + g));
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) mma<T>(acc[mt][nt], fb[nt], fa[mt]);
    }
    __syncthreads();
  }

  long long mrow[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int y = y0 + wm * 4 + mt;
    mrow[mt] = (y < q.H) ? ((long long)b * q.H + y) * q.W + x0 : -1;
  }
  // bf16 lane: the compact "plain bf16" epilogue (bias, residual, bf16 rows) - the generic one carries every
  // activation / dropout / f32 path and is several times the size of the K loop in instruction-cache terms
  constexpr int EPI = sizeof(T) == 2 ? EPI_PLAIN16 : EPI_GENERIC;
  epilogue_rows<T, 4, 4, EPI>(p, acc, mrow, min(TW, q.W - x0), n0 + wn * 64, 0, lane, smem + w * 4096);
}

// ------------------------------------------------------------------------------------------------------------------
// bf16, narrow layers (Cin * 2 B * 324 pixels + 64 KiB fit the LDS: Cin <= 128 - the 80x848 / 40x424 levels that hold
// 3/4 of the encoder's FLOPs).  Same idea with the memory system in mind:
//   * one PERSISTENT 512-thread workgroup per CU walks 16 x 16-pixel tiles: the 288 KiB weight matrix is streamed once
//     per 256 pixels instead of once per 128, and - since every tile wants the same 18 weight K-steps in the same
//     order - through a 4-stage LDS-DMA ring that never restarts: three K-steps are always in flight, under counted
//     `s_waitcnt vmcnt` + one raw `s_barrier` per K-step.  (The 128-pixel kernel above has one K-step = 0.24 us of
//     MFMA work in flight against >= 1 us of L2 latency: it runs at the latency, not at the matrix pipe.)
//   * 8 waves as 4 (pixel rows) x 2 (output-channel halves), each 64 pixels x 64 channels like above.
#ifndef CONVW_LAB
#define CONVW_LAB 0  // 1: phase stamps of workgroup 7 into melgpt_convw_dbg (development builds only)
#endif
#if CONVW_LAB
__device__ unsigned long long melgpt_convw_dbg[64];
__device__ unsigned long long melgpt_convw_dbg2[16];
#endif

constexpr int WNST = 4;
// Output tile of the persistent kernel: 16 x 16 pixels, or 8 x 32 (W8) where the height pads badly to 16 rows (40 x 424:
// 3 x 27 tiles of 16 x 16 compute 20 % more pixels than the image has, 5 x 14 of 8 x 32 compute 6 %).  Either way 16
// row-blocks of 16 consecutive pixels; row-block rb sits at tile row rb (16 x 16) or row rb / 2, columns 16 (rb % 2) ...
template <bool W8>
struct WideTile {
  static constexpr int TH = W8 ? 8 : 16, TW = W8 ? 32 : 16, PH = TH + 2, PW = TW + 2, NPIX = PH * PW;
  static __device__ __forceinline__ int row(int rb) { return W8 ? rb >> 1 : rb; }
  static __device__ __forceinline__ int xh(int rb) { return W8 ? (rb & 1) * 16 : 0; }
};
// (host) the shape a layer runs with: 8 x 32 when it computes fewer pixels and the last column tile's second half is
// either whole or empty (the epilogue takes ONE valid-pixel count per tile)
static bool wide_w8(int H, int W) {
  static int off = -1;  // lab switch: MELGPT_CONV_W8=0 keeps every layer on 16 x 16 tiles
  if (off < 0) off = getenv("MELGPT_CONV_W8") && atoi(getenv("MELGPT_CONV_W8")) == 0;
  if (off) return false;
  const long long p16 = (long long)((H + 15) / 16) * ((W + 15) / 16), p8 = (long long)((H + 7) / 8) * ((W + 31) / 32);
  return p8 < p16 && (W % 32 == 0 || W % 32 <= 16);
}
static int wide_tiles_x(int H, int W) { return wide_w8(H, W) ? (W + 31) / 32 : (W + 15) / 16; }
static int wide_tiles_y(int H, int W) { return wide_w8(H, W) ? (H + 7) / 8 : (H + 15) / 16; }

// STATS: also emit the GroupNorm partial sums of the output tile (a separate instance: its extra live values would cost
// the plain one 8 spilled VGPRs)
template <bool STATS, bool W8>
__global__ __launch_bounds__(512) void conv3x3_gn_wide_kernel(FusedConvParams q, int total_tiles) {
  typedef bf16_t T;
  typedef WideTile<W8> WT;
  constexpr int WTH = WT::TH, WTW = WT::TW, WPW = WT::PW, WNPIX = WT::NPIX;
  constexpr int ES = 2, KSTEP = 64, VEC = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GemmParams& p = q.g;
  const int Cin = q.Cin, pix_bytes = Cin * ES, cpp = pix_bytes / 16;
  char* patch = smem;                                        // [324][Cin] swizzled (also the epilogue's staging)
  char* wring = smem + (size_t)WNPIX * pix_bytes;            // [4][128 rows x 128 B]
  float* ab = (float*)(wring + WNST * 16384);                // [Cin][2]
  float* stp = ab + 2 * Cin;                                 // [4 wm][32 groups][2]: output statistics of a tile
  char* rturn = (char*)(stp + 256) + (threadIdx.x >> 6) * 1024;  // per wave: 8 residual rows on their way into accumulator layout
  float* gb = (float*)((char*)(stp + 256) + 8 * 1024);           // [Cin][2]: gamma, beta
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w >> 1, wn = w & 1;
  const int i16 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * 128;
  const bool norm = q.mean != nullptr;
  const int nk = 9 * (Cin / KSTEP);  // 18: Cin = 128, the launcher checks (a run-time value on purpose: with the trip
                                     // count known the compiler re-pipelines the pair loop into 22 spilled VGPRs)

  // weight ring: stage s <- K-step (kg % nk); 16 pieces of 1 KiB per stage, two per wave
  // (the pieces are issued by an asm block, not the builtin: hipcc completes the builtin's LDS write only at vmcnt(0) and
  // put that wait in front of every K-step's fragment reads, draining the ring and the next tile's patch loads - see
  // dma16 in gemm256.hip)
  const unsigned long long wb_addr = (unsigned long long)p.B;
  const u32x4 rb = {(unsigned)wb_addr, (unsigned)(wb_addr >> 32) & 0xFFFFu, p.b_bytes, 0x00020000u};
  unsigned b_base[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = w + 8 * i, row = piece * 8 + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
    b_base[i] = (n0 + row < p.N) ? (unsigned)(((long long)(n0 + row) * p.ldb) * ES) + ch * 16 : OOB;
  }
  int kg_issue = 0;  // next K-step (global, over all tiles) to request
  int kt_issue = 0;  // ... within its tile (a running counter: `kg_issue % nk` is a 30-instruction division per request)
  auto issue_w = [&]() {
    const int kt = kt_issue, st = kg_issue & (WNST - 1);
    kt_issue = kt_issue + 1 == nk ? 0 : kt_issue + 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned m0v = (unsigned)(size_t)LDS_PTR(char, wring + st * 16384 + (w + 8 * i) * 1024);
      const unsigned voff = b_base[i] == OOB ? OOB : b_base[i] + kt * KSTEP * ES;
      asm volatile("s_nop 2\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                   :
                   : "s"(m0v), "v"(voff), "s"(rb)
                   : "memory", "m0");
    }
    ++kg_issue;
  };
  issue_w();
  issue_w();  // the first PAIR of K-steps (the K loop below works in pairs: one barrier per two steps)
  int kg = 0;  // K-step being multiplied

  // raw input chunks of a tile: requested one tile AHEAD (during the previous tile's K loop), so the staging pass below
  // starts from registers.  cpp (16-byte chunks per pixel) is a power of two that divides 512: a thread keeps ONE
  // channel chunk for all its pixels, pixel / chunk come from shifts.
  constexpr int MAXCH = (WNPIX * 16 + 511) / 512;  // chunks per thread at Cin = 128
  const int cshift = __builtin_ctz(cpp), ch = t & (cpp - 1), ppi = 512 >> cshift;  // pixels advanced per trip
  u32x4 raw[MAXCH];
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(q.x, q.x_bytes);
  const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.R, p.R ? q.r_bytes : 0u);
  auto fetch_patch = [&](int tile) {
    const int tx = tile % q.tiles_x, ty = (tile / q.tiles_x) % q.tiles_y, b = tile / (q.tiles_x * q.tiles_y);
    const int y0 = ty * WTH, x0 = tx * WTW;
    // exactly MAXCH buffer loads per wave, whatever the tile (out-of-image / out-of-range lanes get an out-of-bounds
    // offset and read zeros): the K loop's counted waits below rely on that number
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int pix = (t >> cshift) + ppi * i;
      const int py = pix / WPW, px = pix - py * WPW;
      const int iy = y0 + py - 1, ix = x0 + px - 1;
      const bool ok = tile < total_tiles && pix < WNPIX && iy >= 0 && iy < q.H && ix >= 0 && ix < q.W;
      const unsigned off = ok ? (unsigned)((((long long)b * q.H + iy) * q.W + ix) * Cin * ES) + ch * 16 : OOB;
      raw[i] = buf_load16(rx, off);
    }
  };
  // GroupNorm constants (a, b) of a tile's image: thread c < Cin owns channel c (Cin <= 512: the launcher checks).  The
  // image's rstd / mean are REQUESTED with the tile's patch (one tile ahead) and folded into (a, b) before the previous
  // tile's epilogue stores go out; gamma / beta wait in LDS.  Computed at the top of the tile from four global loads, the
  // constants cost 2-5 k of a tile's 47 k cycles: two dependent round trips (the LDS store between them may alias, as
  // far as the compiler knows), each behind a vmcnt(0) that also waits for the epilogue's stores to retire.
  const bool abt = norm && t < Cin;
  const int cgs = Cin / 32, tiles_img = q.tiles_x * q.tiles_y;
  float nr = 0.f, nm = 0.f, na = 0.f, nbb = 0.f;
  auto fetch_stats = [&](int tile) {
    if (abt && tile < total_tiles) {
      const int b = tile / tiles_img;
      nr = q.rstd[b * 32 + t / cgs];
      nm = q.mean[b * 32 + t / cgs];
    }
  };
  auto fold_stats = [&]() {
    if (abt) {
      na = nr * gb[2 * t];
      nbb = gb[2 * t + 1] - nm * na;
    }
    asm volatile("" : "+v"(na), "+v"(nbb));  // here, not at their use behind the epilogue's stores
  };
  if (abt) {
    gb[2 * t] = q.gamma[t];
    gb[2 * t + 1] = q.beta[t];
  }
  fetch_stats(blockIdx.x);
  fetch_patch(blockIdx.x);
  fold_stats();  // (a thread reads back its own two LDS words: no barrier)

  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int tx = tile % q.tiles_x, ty = (tile / q.tiles_x) % q.tiles_y, b = tile / (q.tiles_x * q.tiles_y);
    const int y0 = ty * WTH, x0 = tx * WTW;
#if CONVW_LAB
    unsigned long long st0 = __builtin_amdgcn_s_memtime(), st1 = 0, st2 = 0;
#endif
    if (abt) {
      ab[2 * t] = na;
      ab[2 * t + 1] = nbb;
    }
    // The accumulators START from bias + residual instead of zero, and the epilogue is left with rounding and stores.
    // (Fetched by the epilogue - two batches of row slabs through the LDS staging block - the residual cost a tile 7 500
    // of its 60 000 cycles, the bias round trip another ~2 000: tools/lab/convw_lab.hip.)  The residual is requested
    // here as full rows - 8 lanes per 128-byte row of the wave's 64 channels, 8 pixels per load: requested in
    // accumulator layout (8 bytes per lane, neighbouring lanes 256 bytes apart) every lane is a memory request of its
    // own and the 128 loads of a tile take as long to issue as they save - lands under the staging pass below and is
    // turned into accumulator layout through a 1 KiB block of LDS per wave when the accumulators are initialised.
    u32x4 rrow[4][2];
    f32x4 bv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + g * 4;
      bv[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias && n < p.N) bv[nt] = *(const f32x4*)(p.bias + n);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        const int y = y0 + WT::row(wm * 4 + mt), x = x0 + WT::xh(wm * 4 + mt) + hp * 8 + (lane >> 3), n = n0 + wn * 64 + (lane & 7) * 8;
        const bool ok = p.R && y < q.H && x < q.W && n < p.N;
        const unsigned off = ok ? (unsigned)(((((long long)b * q.H + y) * q.W + x) * p.ldr + n) * ES) : OOB;
        rrow[mt][hp] = buf_load16(rres, off);
      }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // ab visible; previous tile's epilogue is out of the patch
#if CONVW_LAB
    const unsigned long long stA = __builtin_amdgcn_s_memtime();
#endif
    // ---- stage the input patch (normalise + swish on the fly; out-of-image pixels are zeros AFTER the normalisation)
    f32x4 sc[4];  // (a, b) of channels 8 ch .. 8 ch + 7, interleaved
    if (norm) {
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[e] = *(const f32x4*)(ab + 2 * (ch * VEC) + 4 * e);
    }
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int pix = (t >> cshift) + ppi * i;
      if (pix < WNPIX) {
        const int py = pix / WPW, px = pix - py * WPW;
        const int iy = y0 + py - 1, ix = x0 + px - 1;
        u32x4 v = raw[i];
        if (norm && iy >= 0 && iy < q.H && ix >= 0 && ix < q.W) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float o0 = fmaf(bf16lo(v[e]), sc[e][0], sc[e][1]);
            float o1 = fmaf(bf16hi(v[e]), sc[e][2], sc[e][3]);
            if (q.swish) {
              o0 = o0 * __builtin_amdgcn_rcpf(1.0f + __expf(-o0));
              o1 = o1 * __builtin_amdgcn_rcpf(1.0f + __expf(-o1));
            }
            v[e] = pack_bf16x2(o0, o1);
          }
        }
        *(u32x4*)(patch + patch_off<T>(pix, ch, pix_bytes)) = v;
      }
    }
#if CONVW_LAB
    st1 = __builtin_amdgcn_s_memtime();
#endif
    fetch_stats(tile + gridDim.x);
    fetch_patch(tile + gridDim.x);  // lands under this tile's K loop
    // (the first K-step's barrier below publishes the patch)

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      u32x2 rr[4] = {u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}};
      if (p.R) {
        // 8 pixel rows of 128 bytes at a time through the wave's block (16-byte chunk index XOR-ed with the row); the LDS
        // executes one wave's operations in order: no wait between a pass's reads and the next pass's write
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
          *(u32x4*)(rturn + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4)) = rrow[a][hp];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const u32x2 v = *(const u32x2*)(rturn + (i16 & 7) * 128 + (((2 * c + (g >> 1)) ^ (i16 & 7)) << 4) + (g & 1) * 8);
            if ((i16 >> 3) == hp) rr[c] = v;
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[a][c] = f32x4{bf16lo(rr[c][0]), bf16hi(rr[c][0]), bf16lo(rr[c][1]), bf16hi(rr[c][1])} + bv[c];
    }

    // Two K-steps per barrier (nk is even: the launcher checks): the ring's four stages are two pairs - while pair p is
    // multiplied pair p + 1 lands in the stages pair p - 1 was read from.  At one barrier per 64-wide step the SIMD's two
    // waves spent as long waiting (barrier skew + the fragment reads right behind it) as multiplying.
    // (Cin = 128, the only width this kernel is launched for: a pair of K-steps is one filter tap, its two steps the tap's
    // two 64-channel halves - written with a run-time `kpt` the tap / channel split cost the scalar unit two divisions per
    // step, in front of the step's first fragment reads)
    for (int pr = 0; pr < nk / 2; ++pr) {
      const int ky = pr / 3, kx = pr - 3 * ky;
      // this wave's pieces of the pair have landed; after the barrier everybody's have, and everybody is past the
      // previous pair, whose two stages are refilled below.  (The next tile's MAXCH patch loads were issued just before
      // this loop: younger than the first pair's pieces - they stay in flight - and older than all later ones.)
      if (pr == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(MAXCH) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int hh = 0; hh < 2; ++hh, ++kg) {
        const int kc = hh;
        const char* sb = wring + (kg & (WNST - 1)) * 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 fa[4], fb[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int pix = (WT::row(wm * 4 + mt) + ky) * WPW + WT::xh(wm * 4 + mt) + i16 + kx;  // row-block wm*4+mt, shifted by the tap
            fa[mt] = *(const u32x4*)(patch + patch_off<T>(pix, kc * 8 + 4 * ks + g, pix_bytes));
          }
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) fb[nt] = *(const u32x4*)(sb + row_off((wn * 4 + nt) * 16 + i16, 4 * ks + g));
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) mma<T>(acc[mt][nt], fb[nt], fa[mt]);
          // the previous pair's two stages are refilled from INSIDE the MFMA stream (everybody is past the barrier), one
          // K-step's pieces behind each of the pair's first two 32-wide steps: issued right behind the barrier the four
          // requests (with their m0 writes and s_nops) stood in front of the pair's first fragment reads - K loop
          // 35.1 k -> 33.1 k cycles per tile (tools/lab/convw_lab.hip)
          if (hh == 0) issue_w();
        }
      }
    }
#if CONVW_LAB
    st2 = __builtin_amdgcn_s_memtime();
#endif
    fold_stats();  // the next tile's (a, b)
    asm volatile("s_barrier" ::: "memory");  // everybody is done reading the patch: it becomes the epilogue's staging
    if constexpr (STATS) {
      // GroupNorm(32) statistics of THIS conv's output (Cout = 128: a lane's four consecutive channels are one group),
      // taken on the values as they are stored (bias and residual added, rounded to bf16): the next ResnetBlock norm then needs no
      // pass over the 2.2 GB tensor.  Per tile a (32, 2) partial, summed across the four pixel-row waves in wave
      // order; melgpt_groupnorm_finalize adds the tiles of an image in tile order.
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          if (x0 + WT::xh(wm * 4 + mt) + i16 < q.W && y0 + WT::row(wm * 4 + mt) < q.H) {
            // rounded as the epilogue will round them, two values per v_cvt_pk (bias and residual are in the accumulators)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              const unsigned pk = pack_bf16x2(acc[mt][nt][e], acc[mt][nt][e + 1]);
              const float v0 = bf16lo(pk), v1 = bf16hi(pk);
              s1[nt] += v0;
              s2[nt] = fmaf(v0, v0, s2[nt]);
              s1[nt] += v1;
              s2[nt] = fmaf(v1, v1, s2[nt]);
            }
          }
        }
      }
      // the 16 pixels of a row of lanes: four DPP steps (a ds_bpermute shuffle per step was 32 LDS round trips per tile)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        s1[nt] += dpp_move<0xB1>(s1[nt]);
        s2[nt] += dpp_move<0xB1>(s2[nt]);
        s1[nt] += dpp_move<0x4E>(s1[nt]);
        s2[nt] += dpp_move<0x4E>(s2[nt]);
        s1[nt] += dpp_move<0x141>(s1[nt]);
        s2[nt] += dpp_move<0x141>(s2[nt]);
        s1[nt] += dpp_move<0x140>(s1[nt]);
        s2[nt] += dpp_move<0x140>(s2[nt]);
      }
      if (i16 == 0) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int grp = wn * 16 + nt * 4 + g;
          stp[(wm * 32 + grp) * 2] = s1[nt];
          stp[(wm * 32 + grp) * 2 + 1] = s2[nt];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (w == 0) {
        const float v = ((stp[lane] + stp[64 + lane]) + stp[128 + lane]) + stp[192 + lane];
        q.stat_part[(long long)tile * 64 + lane] = v;  // tile = (b * tiles_y + ty) * tiles_x + tx
      }
    }
    long long mrow[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int y = y0 + WT::row(wm * 4 + mt), xs = x0 + WT::xh(wm * 4 + mt);
      mrow[mt] = (y < q.H && xs < q.W) ? ((long long)b * q.H + y) * q.W + xs : -1;
    }
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));  // keep the epilogue's per-lane offsets out of the tile loop's live range
    GemmParams pe = p;  // (bias and residual are in the accumulators: the loadless plain mode)
    pe.bias = nullptr;
    pe.R = nullptr;
    // (valid pixels of a row-block: the launcher picks 8 x 32 only where the second half of the last column tile is whole or empty)
    epilogue_rows<T, 4, 4, EPI_PLAIN16N>(pe, acc, mrow, min(16, q.W - x0), n0 + wn * 64, 0, lane_e, smem + w * 4096);
#if CONVW_LAB
    if (blockIdx.x == 7 && t == 0) {
      const int k = (tile - 7) / gridDim.x;
      if (k < 15) {
        melgpt_convw_dbg2[k] = stA;
        melgpt_convw_dbg[4 * k] = st0; melgpt_convw_dbg[4 * k + 1] = st1; melgpt_convw_dbg[4 * k + 2] = st2;
        melgpt_convw_dbg[4 * k + 3] = __builtin_amdgcn_s_memtime();
      }
    }
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool W8>
int launch_fused_wide_t(const FusedConvParams& q0, int B, hipStream_t s) {
  FusedConvParams q = q0;
  q.tiles_x = (q.W + WideTile<W8>::TW - 1) / WideTile<W8>::TW;
  q.tiles_y = (q.H + WideTile<W8>::TH - 1) / WideTile<W8>::TH;
  const size_t lds = (size_t)WideTile<W8>::NPIX * q.Cin * 2 + WNST * 16384 + (size_t)q.Cin * 8 + 1024 + 8 * 1024 + (size_t)q.Cin * 8;
  if (q.Cin > 512 || lds > 160 * 1024) return MELGPT_ERR_UNSUPPORTED;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      return MELGPT_ERR_LAUNCH;
    if (hipFuncSetAttribute((const void*)conv3x3_gn_wide_kernel<false, W8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv3x3_gn_wide_kernel<true, W8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return MELGPT_ERR_LAUNCH;
    ncu = n;
  }
  const long long total = (long long)q.tiles_x * q.tiles_y * B;
  if (total > 0x7FFFFFFF) return MELGPT_ERR_UNSUPPORTED;
  const int gy = (q.g.N + 127) / 128;
  const int avail = ncu - melgpt_get_reserved_cus() >= 8 ? ncu - melgpt_get_reserved_cus() : ncu;
  int gx = avail / gy;
  if (gx < 1) gx = 1;
  if (gx > total) gx = (int)total;
  if (q.stat_part) {
    if (gy != 1) return MELGPT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv3x3_gn_wide_kernel<true, W8>), dim3(gx, gy), dim3(512), lds, s, q, (int)total);
  } else {
    hipLaunchKernelGGL((conv3x3_gn_wide_kernel<false, W8>), dim3(gx, gy), dim3(512), lds, s, q, (int)total);
  }
  return melgpt_launch_status();
}
int launch_fused_wide(const FusedConvParams& q, int B, hipStream_t s) {
  return wide_w8(q.H, q.W) ? launch_fused_wide_t<true>(q, B, s) : launch_fused_wide_t<false>(q, B, s);
}

template <typename T>
int launch_fused(const FusedConvParams& q, int B, hipStream_t s) {
  const int ES = Tr<T>::ES;
  const size_t lds = (size_t)NPIX * q.Cin * ES + 2 * 16384 + (size_t)q.Cin * 8;
  if (lds > 160 * 1024) return MELGPT_ERR_UNSUPPORTED;
  static size_t attr = 0;
  if (lds > attr) {
    if (hipFuncSetAttribute((const void*)conv3x3_gn_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return MELGPT_ERR_LAUNCH;
    attr = 160 * 1024;
  }
  dim3 grid(q.tiles_x * q.tiles_y * B, (q.g.N + 127) / 128);
  hipLaunchKernelGGL(conv3x3_gn_kernel<T>, grid, dim3(256), lds, s, q);
  return melgpt_launch_status();
}

}  // namespace

extern "C" int melgpt_groupnorm_finalize(const float* partial, int nchunks, int B, double count, float eps, float* mean,
                                         float* rstd, void* stream);

// stat_part != null: the output's GroupNorm partials are wanted; MELGPT_ERR_UNSUPPORTED (nothing launched) when this
// configuration does not run on the persistent kernel or the lane-to-group mapping does not hold
static int conv3x3_gn_impl(const void* x, int B, int H, int W, int Cin, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int swish, const void* wpack, int Cout,
                           const float* bias, const void* residual, void* y, int dtype, float* stat_part, void* stream) {
  MELGPT_CHECK(x && wpack && y && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(dtype == MELGPT_F32 || dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK((mean == nullptr) == (rstd == nullptr), MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(!mean || (gamma && beta), MELGPT_ERR_BAD_ARG);
  const int es = dtype == MELGPT_F32 ? 4 : 2, kstep = dtype == MELGPT_F32 ? 32 : 64;
  MELGPT_CHECK(Cin % kstep == 0 && Cin % 32 == 0 && (Cin * es) / 16 >= 16 && Cout % 8 == 0, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK((((uintptr_t)x | (uintptr_t)wpack | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)bias) & 15) == 0,
               MELGPT_ERR_ALIGN);
  const long long M = (long long)B * H * W;
  MELGPT_CHECK(M < 0x7FFFFF00LL && (long long)Cout * 9 * Cin * es < 0xFFFFFF00LL, MELGPT_ERR_UNSUPPORTED);
  FusedConvParams q{};
  q.g.B = wpack; q.g.C = y; q.g.bias = bias; q.g.R = residual;
  q.g.M = (int)M; q.g.N = Cout; q.g.K = 9 * Cin;
  q.g.ldb = 9LL * Cin; q.g.ldc = Cout; q.g.ldr = Cout;
  q.g.b_bytes = (unsigned)((long long)Cout * 9 * Cin * es);
  q.g.alpha = 1.0f;
  q.g.vec_io = (Cout * es) % 16 == 0;
  q.x = x; q.mean = mean; q.rstd = rstd; q.gamma = gamma; q.beta = beta;
  q.H = H; q.W = W; q.Cin = Cin; q.swish = swish;
  q.tiles_x = (W + TW - 1) / TW; q.tiles_y = (H + TH - 1) / TH;
  hipStream_t s = (hipStream_t)stream;
  q.stat_part = stat_part;
  if (stat_part && (dtype != MELGPT_BF16 || Cout != 128)) return MELGPT_ERR_UNSUPPORTED;
  if (dtype == MELGPT_F32) return launch_fused<float>(q, B, s);
  // narrow bf16 layers with plenty of 16 x 16 tiles: the persistent kernel with the weight ring
  const bool w8 = wide_w8(H, W);
  const size_t wide_lds = (size_t)(w8 ? WideTile<true>::NPIX : WideTile<false>::NPIX) * Cin * 2 + WNST * 16384 + (size_t)Cin * 16 + 1024 + 8 * 1024;
  const long long wide_tiles = (long long)wide_tiles_x(H, W) * wide_tiles_y(H, W) * B;
  static int wide_off = -1;
  if (wide_off < 0) wide_off = getenv("MELGPT_CONV_WIDE") && atoi(getenv("MELGPT_CONV_WIDE")) == 0;
  // 16-row tiles pay for the rows they pad: take them only while they compute at most 1/4 more pixels than 8-row tiles
  const long long wide_px = wide_tiles * 256, narrow_px = (long long)q.tiles_x * q.tiles_y * B * TH * TW;
  if (!wide_off && wide_lds <= 160 * 1024 && wide_tiles >= 512 && wide_px * 4 <= narrow_px * 5 && q.g.vec_io &&
      Cin == 128 &&
      M * Cin * 2 < 0xFFFFFF00LL && (!residual || M * Cout * 2 < 0xFFFFFF00LL)) {
    q.x_bytes = (unsigned)(M * Cin * 2);
    q.r_bytes = residual ? (unsigned)(M * Cout * 2) : 0u;
    return launch_fused_wide(q, B, s);
  }
  if (stat_part) return MELGPT_ERR_UNSUPPORTED;
  return launch_fused<bf16_t>(q, B, s);
}

extern "C" int melgpt_conv3x3_gn_nhwc(const void* x, int B, int H, int W, int Cin, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, int swish, const void* wpack, int Cout,
                                      const float* bias, const void* residual, void* y, int dtype, void* stream) {
  return conv3x3_gn_impl(x, B, H, W, Cin, mean, rstd, gamma, beta, swish, wpack, Cout, bias, residual, y, dtype, nullptr,
                         stream);
}

extern "C" int melgpt_conv3x3_gn_stats_workspace(int B, int H, int W) {
  return B * wide_tiles_x(H, W) * wide_tiles_y(H, W) * 64;
}

extern "C" int melgpt_conv3x3_gn_nhwc_stats(const void* x, int B, int H, int W, int Cin, const float* mean,
                                            const float* rstd, const float* gamma, const float* beta, int swish,
                                            const void* wpack, int Cout, const float* bias, const void* residual,
                                            void* y, int dtype, float out_eps, float* out_mean, float* out_rstd,
                                            float* workspace, void* stream) {
  MELGPT_CHECK(out_mean && out_rstd && workspace, MELGPT_ERR_BAD_ARG);
  int st = conv3x3_gn_impl(x, B, H, W, Cin, mean, rstd, gamma, beta, swish, wpack, Cout, bias, residual, y, dtype,
                           workspace, stream);
  if (st != MELGPT_OK) return st;
  const int nchunks = wide_tiles_x(H, W) * wide_tiles_y(H, W);
  return melgpt_groupnorm_finalize(workspace, nchunks, B, (double)H * W * (Cout / 32), out_eps, out_mean, out_rstd, stream);
}
