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

MELGPT_CLK_DECL(clk_conv_ws)

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
      if (pr == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(MELGPT_WAITN(MAXCH)) : "memory");
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
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------------------------
// WAVE-SPECIALISED form of the persistent kernel (round 4; bf16, Cin = Cout = 128, 16 x 16-pixel tiles).
// In the kernel above all 8 waves do everything in turn: stage the patch (VALU: affine + swish, 10.5 k cycles), multiply
// (33 k), store (3.8 k) - 49 k cycles per tile for 18.4 k cycles of MFMA issue, and the two waves of a SIMD run the same
// stream in lockstep, so VALU work placed "under" the MFMAs stalls the matrix pipe for both (profiles/r03_gemm_lab.md).
// Here the two waves of a SIMD have different JOBS:
//   waves 0-3 (one per SIMD) MULTIPLY with v_mfma_f32_32x32x16 (an MFMA holds the SIMD's vector issue for 8 of its 32
//     cycles - the 16x16x32 form for 8 of 16 - which is what leaves the partner wave its VALU slots): wave tile 64 pixels
//     x 128 channels = 2 x 4 tiles of 32 x 32, per 16-wide step 2 patch + 4 weight fragments for 8 MFMAs, the next step's
//     fragments requested under the current step's MFMAs, every fragment address = one per-lane base + a compile-time
//     offset (the patch is LINEAR with a 272-byte pixel pitch: 16 consecutive pixels cover all 64 banks without an XOR,
//     so tap shifts and channel steps are immediates - no address VALU in the loop);
//   waves 4-7 STAGE: raw patch loads one phase ahead, GroupNorm affine + swish, ds_write, and all LDS-DMA requests of the
//     weight ring - on the VALU / memory pipes while their SIMD partner owns the matrix pipe (tools/lab/ws_lab.hip).
// One patch buffer serves both: the K loop runs CHANNEL-HALF-major (half 0: 9 taps, half 1: 9 taps), so while half 1 of tile
// t is multiplied, half 0 of tile t + 1 is staged over the bytes half 0 of tile t no longer needs, and vice versa:
//   phase A(t): multiply half 0 (t)   | stage half 1 (t)      -> barrier
//   phase B(t): multiply half 1 (t)   | stage half 0 (t + 1)  -> epilogue (multiplying waves) -> barrier
// Two workgroup barriers per tile.  The weight ring (4 stages x 16 KiB, a stage = 64 channels of one tap) is handed over
// through two sets of LDS counters instead: full[s] (K-steps whose pieces staging wave s has seen land: its counted
// s_waitcnt vmcnt, then a ds_write) and free[w] (K-steps multiplying wave w has read: LDS executes a wave's operations in
// order, so the ds_write behind the last fragment read is the release).  K-step k is requested when k - 4 is free - three
// K-steps before it is needed.  Every vector-memory operation of a staging wave is inline asm with a hand-counted wait
// (the compiler cannot count LDS-DMA, and would drain the ring in front of any load it can see).
// MFMA rows are weight rows in a PERMUTED order - fragment row rho of tile nt is channel 32 nt + 16 ((rho >> 2) & 1) +
// 4 (rho >> 3) + (rho & 3) - so that the 16 accumulator registers of a lane are 16 CONSECUTIVE channels of its pixel:
// stores and residual loads are 16 bytes per lane (two per tile), 64 contiguous bytes per pixel.  The residual of the next
// tile is fetched during this tile's epilogue and the accumulators start from bias + residual.  Every spin is bounded.
// Output tile 16 x 16 pixels, or 8 x 32 (W8) where 16 rows pad badly (40 x 424: 70 tiles per image instead of 81), as in the
// kernel above; a 32-pixel fragment is two 16-pixel rows of the patch (16 x 16) or 32 consecutive pixels of one row (8 x 32).
constexpr int WS_PP = 272, WS_RING = 4 * 16384;
template <bool W8>
struct WsTile {
  static constexpr int TH = W8 ? 8 : 16, TW = W8 ? 32 : 16, PH = TH + 2, PW = TW + 2, NPIX = PH * PW, PATCH = NPIX * WS_PP;
  static constexpr int LDS = PATCH + WS_RING + 1024 /* gamma, beta */ + 512 /* bias */ + 1024 /* output statistics */ + 64 /* counters */;
};
constexpr int WS_SPIN = 1 << 22;
constexpr float LOG2E_F = 1.4426950408889634f;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool STATS, bool W8>
__global__ __launch_bounds__(512) void conv3x3_gn_ws_kernel(FusedConvParams q, int total_tiles) {
  typedef WsTile<W8> WT;
  constexpr int PP = WS_PP, PW = WT::PW, PH = WT::PH, TH = WT::TH, TW = WT::TW, WS_PATCH = WT::PATCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GemmParams& p = q.g;
  char* patch = smem;
  char* ring = smem + WS_PATCH;
  float* gb = (float*)(ring + WS_RING);           // [128][2]: gamma, beta
  float* bias_l = gb + 256;                       // [128]
  float* stp = bias_l + 128;                      // [4 waves][32 groups][2]
  // (LDS-address-space pointers on purpose: through a generic volatile pointer the counters became FLAT loads / stores,
  // which count on vmcnt AND lgkmcnt, out of order - every poll drained the staging waves' LDS-DMA ring)
  volatile __attribute__((address_space(3))) u32x4* cnt4 = LDS_PTR(volatile u32x4, stp + 256);   // [0] full[0..3], [1] free[0..3]
  volatile __attribute__((address_space(3))) unsigned* cnt = LDS_PTR(volatile unsigned, stp + 256);
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool norm = q.mean != nullptr;
  const int tiles_img = q.tiles_x * q.tiles_y, G = gridDim.x;
  // a workgroup walks a CONTIGUOUS run of tiles (row-major over image, tile row, tile column): the next tile is the right
  // neighbour except at the end of a tile row, and its two left halo columns are the two right columns already staged
  const int t_per = total_tiles / G, t_rem = total_tiles - t_per * G;
  const int t_first = (int)blockIdx.x * t_per + min((int)blockIdx.x, t_rem), t_last = t_first + t_per + ((int)blockIdx.x < t_rem ? 1 : 0);
  MELGPT_CLK_BEGIN();
  if (t < 128) {
    gb[2 * t] = norm ? q.gamma[t] : 1.f;
    gb[2 * t + 1] = norm ? q.beta[t] : 0.f;
    bias_l[t] = p.bias ? p.bias[t] : 0.f;
  }
  if (t < 8) cnt[t] = 0u;
  __syncthreads();
  auto min4 = [](u32x4 v) -> unsigned {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)min(min(v[0], v[1]), min(v[2], v[3])));
  };

  if (w < 4) {
    // ================================================================== multiplying waves (v_mfma_f32_16x16x32)
    // The same roles, protocol and phases as the 32x32x16 form below, with the K-step's 64 channels as TWO 32-wide sub-steps
    // of 32 MFMAs (4 pixel fragments of 16 x 8 channel fragments of 16).  Why a second form: on random operands the chip
    // holds a higher clock under the 16x16x32 instruction than under 32x32x16 at equal cycles per FLOP (guide, "DVFS
    // give-back" item 7: 1.12-1.15 x the FLOP/s; this kernel's own harness, profiles/r04_conv_lab.md section 8: 0.486
    // against 0.622 ms for the bare multiplying loops) - and this kernel ran at the lowest clock of the step.
    //   patch fragment mt (16 pixels of one patch row): lane = pixel i16, 16-byte k-chunk 4 ks + g of the K-step's 64 channels
    //   weight fragment nt: lane = ring row 16 nt + i16, chunk 4 ks + g (natural rows: conflict-free with row_off); the
    //     staging waves request channel chan(R) = 32 (R >> 5) + 8 ((R >> 2) & 3) + 4 ((R >> 4) & 1) + (R & 3) into ring row R,
    //     so that accumulator rows 4 g + r of fragments 2 u, 2 u + 1 are the 8 CONSECUTIVE channels 32 u + 8 g + 0..7 of
    //     the lane's pixel: one 16-byte store / residual load per (pixel fragment, u), 64 contiguous bytes per pixel.
    const int wm = w, i16 = lane & 15, g = lane >> 4;
    auto foff = [](int mt) { return W8 ? (mt >> 1) * PW + 16 * (mt & 1) : mt * PW; };   // patch pixels from fragment 0 to fragment mt
    const char* abase = patch + ((W8 ? wm * 2 : wm * 4) * PW + i16) * PP + g * 16;
    const char* bb[2] = {ring + row_off(i16, g), ring + row_off(i16, 4 + g)};           // sub-step ks; fragment nt: + nt * 2048
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.R, p.R ? q.r_bytes : 0u);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.C, q.r_bytes);
    auto out_off = [&](int tile, int mt) -> unsigned {  // byte offset of (this lane's pixel of fragment mt, channel 8 g) in y / R
      const int b = tile / tiles_img, r = tile - b * tiles_img, ty = r / q.tiles_x, tx = r - ty * q.tiles_x;
      const int y = W8 ? ty * 8 + wm * 2 + (mt >> 1) : ty * 16 + wm * 4 + mt, x = W8 ? tx * 32 + 16 * (mt & 1) + i16 : tx * 16 + i16;
      const bool ok = tile < total_tiles && y < q.H && x < q.W;
      return ok ? (unsigned)(((((long long)b * q.H + y) * q.W + x) * 128 + 8 * g) * 2) : OOB;
    };
    u32x4 rr[4][4];  // residual of the tile about to be multiplied: piece (mt, u) = channels 32 u + 8 g + 0..7 of the lane's pixel
    auto fetch_res = [&](int tile) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const unsigned o = out_off(tile, mt);
#pragma unroll
        for (int u = 0; u < 4; ++u) rr[mt][u] = (p.R && o != OOB) ? buf_load16(rres, o + u * 64) : u32x4{0u, 0u, 0u, 0u};
      }
    };
    auto wait_full = [&](unsigned k) {   // K-step k has landed for every staging wave
      for (int it = 0; min4(cnt4[0]) <= k && it < WS_SPIN; ++it) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
    };
    u32x4 fa0[4], fa1[4], fb[8];
    auto loadA = [&](const char* ab, int kx, int ks, u32x4 (&fa)[4]) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) fa[mt] = *(const u32x4*)(ab + (foff(mt) + kx) * PP + ks * 64);
    };
    f32x4 acc[4][8];
    // one 32-wide sub-step: channel fragment by channel fragment - a weight fragment is re-requested (for the next sub-step:
    // `sbn`) as soon as its four MFMAs are out; the fences pin the order as written (see the 32x32x16 form)
    // (NO scheduling fences here, unlike the 32x32x16 form: pinned fragment by fragment the reloads took fresh registers
    // while the old fragments were still allocated and the loop spilled 57-73 VGPRs - reloads between the MFMAs; the
    // 32x32x16 harness measured the unfenced order at the same wall time, profiles/r04_conv_lab.md section 8)
    auto step = [&](u32x4 (&fc)[4], const char* sbn, bool reload) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = MELGPT_MFMA_16x16x32(fb[nt], fc[mt], acc[mt][nt]);
        if (reload) fb[nt] = *(const u32x4*)(sbn + nt * 2048);
      }
      // scheduling GROUPS instead of fences: four MFMAs, then one LDS read, eight times - each weight reload right behind
      // its fragment's MFMAs (the fenced form's order) without pinning anything else
      if (reload) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        }
      }
    };
    unsigned kg = 0;  // K-steps multiplied so far (over all tiles)
    fetch_res(t_first);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // half 0 of the first tile is staged
    for (int tile = t_first; tile < t_last; ++tile) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 b0 = *(const f32x4*)(bias_l + 32 * u + 8 * g), b1 = *(const f32x4*)(bias_l + 32 * u + 8 * g + 4);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const u32x4 pk = rr[mt][u];
          acc[mt][2 * u] = f32x4{bf16lo(pk[0]) + b0[0], bf16hi(pk[0]) + b0[1], bf16lo(pk[1]) + b0[2], bf16hi(pk[1]) + b0[3]};
          acc[mt][2 * u + 1] = f32x4{bf16lo(pk[2]) + b1[0], bf16hi(pk[2]) + b1[1], bf16lo(pk[3]) + b1[2], bf16hi(pk[3]) + b1[3]};
        }
      }
#pragma unroll 1
      for (int hk = 0; hk < 6; ++hk) {  // (channel half, filter row): three K-steps (kx = 0, 1, 2) each
        const int half = hk >= 3 ? 1 : 0, ky = hk - 3 * half;
        const char* ab = abase + ky * PW * PP + half * 128;
        if (hk == 0 || hk == 3) {
          // a phase starts: (hk == 3) everybody is done with half 0 and half 1 is staged
          if (hk == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          wait_full(kg);
          loadA(ab, 0, 0, fa0);
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) fb[nt] = *(const u32x4*)(bb[0] + ((kg & 3u) << 14) + nt * 2048);
        }
        const bool last_hk = hk == 2 || hk == 5;
        const char* abn = ab + PW * PP;  // (not used behind the phase's last filter row)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx, ++kg) {
          const unsigned so = (kg & 3u) << 14, son = ((kg + 1u) & 3u) << 14;
          const bool last = last_hk && kx == 2;  // the phase's last K-step: nothing of the next one may be touched yet
          const u32x4 pf = cnt4[0];               // the NEXT K-step's counters, looked at a sub-step later (they only grow)
          loadA(ab, kx, 1, fa1);
          step(fa0, bb[1] + so, true);
          asm volatile("" ::: "memory");          // (no fragment load of this stage may sink below its release)
          if (lane == 0) cnt[4 + wm] = kg + 1u;   // behind this wave's last read of the stage (LDS runs a wave's ops in order)
          if (!last) {
            if (min4(pf) > kg + 1u) asm volatile("" ::: "memory");
            else wait_full(kg + 1u);
            if (kx < 2) loadA(ab, kx + 1, 0, fa0);
            else loadA(abn, 0, 0, fa0);
          }
          step(fa1, bb[0] + son, !last);
        }
      }
      // ---- epilogue: the next tile's residual first (into registers the K loop does not hold)
      fetch_res(tile + 1 < t_last ? tile + 1 : total_tiles);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const u32x4 pk = {pack_bf16x2(acc[mt][2 * u][0], acc[mt][2 * u][1]), pack_bf16x2(acc[mt][2 * u][2], acc[mt][2 * u][3]),
                            pack_bf16x2(acc[mt][2 * u + 1][0], acc[mt][2 * u + 1][1]), pack_bf16x2(acc[mt][2 * u + 1][2], acc[mt][2 * u + 1][3])};
          const unsigned o = out_off(tile, mt);
          if (o != OOB)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, pk), ry, o + u * 64, 0, 0);
          if constexpr (STATS) {
            // GroupNorm(32) statistics of THIS conv's output on the values as stored: the lane's words 2 j, 2 j + 1 are group
            // 8 u + 2 g + j of its pixel
            if (o != OOB) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float v0 = bf16lo(pk[e]), v1 = bf16hi(pk[e]);
                s1[e >> 1] += v0 + v1;
                s2[e >> 1] = fmaf(v1, v1, fmaf(v0, v0, s2[e >> 1]));
              }
            }
          }
        }
        if constexpr (STATS) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {   // over the 16 pixels of the lane's DPP row: four DPP steps
            float a = s1[j], c = s2[j];
            a += dpp_move<0xB1>(a);
            c += dpp_move<0xB1>(c);
            a += dpp_move<0x4E>(a);
            c += dpp_move<0x4E>(c);
            a += dpp_move<0x141>(a);
            c += dpp_move<0x141>(c);
            a += dpp_move<0x140>(a);
            c += dpp_move<0x140>(c);
            if (i16 == 0) {
              stp[(wm * 32 + 8 * u + 2 * g + j) * 2] = a;
              stp[(wm * 32 + 8 * u + 2 * g + j) * 2 + 1] = c;
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // half 1 is free; half 0 of the next tile is staged
      if constexpr (STATS) {
        if (wm == 0) {
          const float v = ((stp[lane] + stp[64 + lane]) + stp[128 + lane]) + stp[192 + lane];
          q.stat_part[(long long)tile * 64 + lane] = v;  // tile = (b * tiles_y + ty) * tiles_x + tx
        }
      }
    }
  } else {
    // ================================================================== staging waves
    const int s = w - 4, sthr = t - 256, ch = sthr & 7, prow = sthr >> 3;   // 8 chunks (64 channels) x 32 pixels per trip
    const unsigned long long wb_addr = (unsigned long long)p.B, x_addr = (unsigned long long)q.x;
    const u32x4 rb = {(unsigned)wb_addr, (unsigned)(wb_addr >> 32) & 0xFFFFu, p.b_bytes, 0x00020000u};
    const u32x4 rx = {(unsigned)x_addr, (unsigned)(x_addr >> 32) & 0xFFFFu, q.x_bytes, 0x00020000u};
    const unsigned long long rs_addr = (unsigned long long)q.rstd, mn_addr = (unsigned long long)q.mean;
    const unsigned st_bytes = norm ? (unsigned)((total_tiles / tiles_img) * 32 * 4) : 0u;
    const u32x4 rrs = {(unsigned)rs_addr, (unsigned)(rs_addr >> 32) & 0xFFFFu, st_bytes, 0x00020000u};
    const u32x4 rmn = {(unsigned)mn_addr, (unsigned)(mn_addr >> 32) & 0xFFFFu, st_bytes, 0x00020000u};
    // weight pieces of this wave: four 1 KiB pieces (8 weight rows x 128 bytes) per K-step; per-lane source offsets fixed,
    // the K-step's (tap, half) goes into the instruction's scalar offset
    unsigned b_base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = s + 4 * i, row = piece * 8 + (lane >> 3), chs = (lane & 7) ^ ((row >> 1) & 7);
      // (the 16x16x32 form reads ring rows in natural order: ring row R then holds channel chan(R), see the multiplying waves)
      const int src = 32 * (row >> 5) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3);
      b_base[i] = row < p.N ? (unsigned)(((long long)src * p.ldb) * 2) + chs * 16 : OOB;
    }
    auto issue_w = [&](int kt, unsigned stage) {  // K-step kt of a tile (channel-half-major)
      const int half = kt >= 9 ? 1 : 0, tap = kt - 9 * half;
      const unsigned koff = (unsigned)(tap * 128 + half * 64) * 2u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned m0v = (unsigned)(size_t)LDS_PTR(char, ring + stage * 16384 + (s + 4 * i) * 1024);
        // (an out-of-range lane offset stays out of range: 0xFFFFFFF0 + koff < 2^32)
        asm volatile("s_nop 2\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :
                     : "s"(m0v), "v"(b_base[i]), "s"(rb), "s"(koff)
                     : "memory", "m0");
      }
    };
    // raw chunks of one channel half of a tile: 11 trips of 32 pixels x 8 chunks (+ the image's rstd / mean pairs of this
    // thread's two groups): 13 vector-memory operations per call, whatever the tile (the waits below count on it).
    // Trips 0 .. NEWT-1 cover patch columns 2 .. PW-1 (TW columns x PH rows: exactly NEWT x 32 pixels); the remaining
    // EDGE trips cover columns 0 and 1, which a tile whose left neighbour was staged just before it does not need (they
    // are copied from that neighbour's columns TW, TW + 1 inside the LDS) - their loads are then out of range and cost
    // an issue slot, their conversion is skipped.
    // Tile-invariant per trip: the pixel's patch coordinates and its byte offset relative to the tile's origin pixel.
    constexpr int NEWT = PH * TW / 32;        // 9 (16 x 16 tiles) / 10 (8 x 32)
    int pyx[11], poff[11], doff[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      int py, px;
      bool in;
      if (i < NEWT) {
        const int n = prow + 32 * i;
        py = n / TW;
        px = 2 + n - py * TW;
        in = true;
      } else {
        const int m = prow + 32 * (i - NEWT);
        py = m >> 1;
        px = m & 1;
        in = m < 2 * PH;
      }
      pyx[i] = in ? (py << 8 | px) : 0x7F7F;
      poff[i] = ((py - 1) * q.W + (px - 1)) * 256 + ch * 16;
      doff[i] = (py * PW + px) * PP + ch * 16;
    }
    // (the patch column of a thread's pixels in the trips 0 .. NEWT-1 is the same in every trip: 2 + prow mod TW.)  The
    // threads of columns TW, TW + 1 copy the previous tile's values of those columns to columns 0, 1 - in one burst at the
    // head of a phase, before their own conversions overwrite the sources: no ordering between the staging waves is needed.
    const bool cp_thread = 2 + (prow % TW) >= TW;
    struct Raw {
      u32x4 v[11];
      u32x2 rs, mn;
      unsigned ok;  // bit i: pixel of trip i lies inside the image
    };
    // shared(tile): its columns 0, 1 come out of the LDS (the tile staged before it in this workgroup is its left neighbour)
    int tx_of_first = (t_first % tiles_img) % q.tiles_x;   // tile column of the run's first tile; tile k of the run: (tx_of_first + k) mod tiles_x
    auto shared_halo = [&](int tile) -> bool {
      if (tile <= t_first || tile >= t_last) return false;
      return ((unsigned)(tx_of_first + (tile - t_first))) % (unsigned)q.tiles_x != 0u;
    };
    auto load_raw = [&](int tile, int half, Raw& r) {
      const int b = tile / tiles_img, rt = tile - b * tiles_img, ty = rt / q.tiles_x, tx = rt - ty * q.tiles_x;
      const int y0 = ty * TH, x0 = tx * TW;
      // patch rows [ylo, yhi) and columns [xlo, xhi) lie inside the image (all wave-uniform)
      const int live = tile < total_tiles;
      const int ylo = (live && y0 == 0) ? 1 : 0, yhi = live ? min(PH, q.H - y0 + 1) : 0;
      const int xlo = shared_halo(tile) ? 2 : (x0 == 0 ? 1 : 0), xhi = min(PW, q.W - x0 + 1);
      const unsigned base = (unsigned)((((long long)b * q.H + y0) * q.W + x0) * 256) + half * 128;
      r.ok = 0u;
#pragma unroll
      for (int i = 0; i < 11; ++i) {
        const int py = pyx[i] >> 8, px = pyx[i] & 255;
        const bool ok = (unsigned)(py - ylo) < (unsigned)(yhi - ylo) && (unsigned)(px - xlo) < (unsigned)(xhi - xlo);
        r.ok |= ok ? (1u << i) : 0u;
        const unsigned off = ok ? base + (unsigned)poff[i] : OOB;
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r.v[i]) : "v"(off), "s"(rx) : "memory");
      }
      const unsigned so = (norm && live) ? (unsigned)((b * 32 + half * 16 + 2 * ch) * 4) : OOB;
      asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(r.rs) : "v"(so), "s"(rrs) : "memory");
      asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(r.mn) : "v"(so), "s"(rmn) : "memory");
    };
    // wait until at most N of this wave's vector-memory operations are outstanding; names every register of r so that no
    // use of them is scheduled above it
    auto wait_raw = [&](auto n_c, Raw& r) {
      constexpr int N = decltype(n_c)::value;
      asm volatile("s_waitcnt vmcnt(%[n])"
                   : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]),
                     "+v"(r.v[7]), "+v"(r.v[8]), "+v"(r.v[9]), "+v"(r.v[10]), "+v"(r.rs), "+v"(r.mn)
                   : [n] "n"(MELGPT_WAITN(N))
                   : "memory");
    };
    float ca[8], cb[8];  // affine of this thread's 8 channels for the half being staged
    auto affine = [&](int half, const Raw& r) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = half * 64 + ch * 8 + e;
        const float rs = __uint_as_float(r.rs[e >> 2]), mn = __uint_as_float(r.mn[e >> 2]);
        const float a = norm ? rs * gb[2 * c] : 1.f;
        ca[e] = a;
        cb[e] = norm ? gb[2 * c + 1] - mn * a : 0.f;
      }
    };
    auto convert = [&](int half, int i, const Raw& r, bool shared) {
      if (i >= NEWT && shared) return;   // (wave-uniform) columns 0, 1 are copied below instead
      if ((pyx[i] & 255) < 64) {   // (the last edge trip covers 4 or 20 pixels only)
        u32x4 v = r.v[i];
        if (norm) {   // branch-free: computed for every pixel, zeroed outside the image
          const unsigned keep = (r.ok >> i & 1u) ? 0xFFFFFFFFu : 0u;
          float o[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[2 * e] = fmaf(bf16lo(v[e]), ca[2 * e], cb[2 * e]);
            o[2 * e + 1] = fmaf(bf16hi(v[e]), ca[2 * e + 1], cb[2 * e + 1]);
          }
          if (q.swish) {   // ONE uniform branch around all eight chains (a test per pair kept the compiler from interleaving them)
            float d[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = __builtin_amdgcn_exp2f(o[e] * -LOG2E_F);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = __builtin_amdgcn_rcpf(1.0f + d[e]);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= d[e];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]) & keep;
        }
        *(u32x4*)(patch + doff[i] + half * 128) = v;   // (outside the image: the load returned zeros)
      }
    };
    // One phase: `use` is staged into channel half `half` while `next` (the half after it) is requested; the nine slots
    // follow the multiplying waves' K-steps sigma0 .. sigma0 + 8: slot j requests K-step sigma0 + j + 3 (tile-local index
    // kt0 + j, mod 18) once K-step sigma0 + j - 1 is free, then publishes K-step sigma0 + j + 1.
    unsigned free_seen = 0;   // K-steps every multiplying wave was seen to have read
    u32x4 pf_free = {0u, 0u, 0u, 0u};
#define WS_T(k)
    auto phase = [&](int half, Raw& use, bool use_shared, int next_tile, int next_half, Raw& next, unsigned sigma0, int kt0) {
      load_raw(next_tile, next_half, next);
      WS_T(0)
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const unsigned sigma = sigma0 + j;
        // FIRST the publication the multiplying waves may be waiting for: this wave's pieces of K-step sigma + 1 (requested
        // two slots ago) have landed - younger are K-step sigma + 2 (4 pieces) and, in slots 0 and 1, the 13 loads of `next`
        if (j == 0) wait_raw(std::integral_constant<int, 17>{}, use);   // ... which also covers `use` (requested a phase ago)
        else if (j == 1) asm volatile("s_waitcnt " MELGPT_VMCNT(17) ::: "memory");
        else asm volatile("s_waitcnt " MELGPT_VMCNT(4) ::: "memory");
        if (lane == 0) cnt[s] = sigma + 2u;
        WS_T(3)
        // then the request of K-step sigma + 3, once K-step sigma - 1 is free (the counters only grow: a poll that saw the
        // multiplying waves ahead serves the following slots too, and the value read before the previous slot's
        // conversion work is looked at first)
        free_seen = max(free_seen, min4(pf_free));
        if (free_seen < sigma) {
          for (int it = 0; (free_seen = min4(cnt4[1])) < sigma && it < WS_SPIN; ++it) {
            __builtin_amdgcn_s_sleep(1);
          }
        }
        asm volatile("" ::: "memory");
        WS_T(1)
        int kt = kt0 + j;
        kt = kt >= 18 ? kt - 18 : kt;
        issue_w(kt, (sigma + 3u) & 3u);
        WS_T(2)
        pf_free = cnt4[1];
        if (j == 0) {
          affine(half, use);
          if (use_shared && cp_thread) {   // columns TW, TW + 1 of the previous tile -> columns 0, 1 of this one
            u32x4 tmp[NEWT];
#pragma unroll
            for (int i = 0; i < NEWT; ++i) tmp[i] = *(const u32x4*)(patch + doff[i] + half * 128);
#pragma unroll
            for (int i = 0; i < NEWT; ++i) *(u32x4*)(patch + doff[i] - TW * PP + half * 128) = tmp[i];
          }
        }
        convert(half, j, use, use_shared);
        if (j == 2) convert(half, 9, use, use_shared);
        if (j == 5) convert(half, 10, use, use_shared);
        WS_T(4)
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    Raw r0, r1;
    // prologue: both halves of the first tile requested, K-steps 0, 1, 2 requested, half 0 staged, K-step 0 published
    load_raw(t_first, 0, r0);
    load_raw(t_first, 1, r1);
    issue_w(0, 0u);
    issue_w(1, 1u);
    issue_w(2, 2u);
    wait_raw(std::integral_constant<int, 25>{}, r0);   // younger than r0's 13: r1's 13 + 12 pieces
    affine(0, r0);
#pragma unroll
    for (int i = 0; i < 11; ++i) convert(0, i, r0, false);
    asm volatile("s_waitcnt " MELGPT_VMCNT(8) ::: "memory");   // K-step 0 (and r1) landed
    if (lane == 0) cnt[s] = 1u;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    unsigned sigma0 = 0;
    for (int tile = t_first; tile < t_last; ++tile) {
      const int nxt = tile + 1 < t_last ? tile + 1 : total_tiles;
      phase(1, r1, shared_halo(tile), nxt, 0, r0, sigma0, 3);        // A(t): stage half 1 of t, request half 0 of the next tile
      phase(0, r0, shared_halo(nxt), nxt, 1, r1, sigma0 + 9u, 12);  // B(t): stage half 0 of the next tile, request its half 1
      sigma0 += 18u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
  }
  MELGPT_CLK_END(clk_conv_ws);
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
template <bool W8>
int launch_fused_ws_t(const FusedConvParams& q0, int B, hipStream_t s) {
  typedef WsTile<W8> WT;
  FusedConvParams q = q0;
  q.tiles_x = (q.W + WT::TW - 1) / WT::TW;
  q.tiles_y = (q.H + WT::TH - 1) / WT::TH;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      return MELGPT_ERR_LAUNCH;
    if (hipFuncSetAttribute((const void*)conv3x3_gn_ws_kernel<false, W8>, hipFuncAttributeMaxDynamicSharedMemorySize, WT::LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv3x3_gn_ws_kernel<true, W8>, hipFuncAttributeMaxDynamicSharedMemorySize, WT::LDS) != hipSuccess)
      return MELGPT_ERR_LAUNCH;
    ncu = n;
  }
  const long long total = (long long)q.tiles_x * q.tiles_y * B;
  if (total > 0x3FFFFFFF) return MELGPT_ERR_UNSUPPORTED;
  const int avail = ncu - melgpt_get_reserved_cus() >= 8 ? ncu - melgpt_get_reserved_cus() : ncu;
  const int gx = (int)(total < avail ? total : avail);
  if (q.stat_part) hipLaunchKernelGGL((conv3x3_gn_ws_kernel<true, W8>), dim3(gx), dim3(512), WT::LDS, s, q, (int)total);
  else hipLaunchKernelGGL((conv3x3_gn_ws_kernel<false, W8>), dim3(gx), dim3(512), WT::LDS, s, q, (int)total);
  return melgpt_launch_status();
}
int launch_fused_ws(const FusedConvParams& q, int B, hipStream_t s) {
  return wide_w8(q.H, q.W) ? launch_fused_ws_t<true>(q, B, s) : launch_fused_ws_t<false>(q, B, s);
}
int launch_fused_wide(const FusedConvParams& q, int B, hipStream_t s) {
  // (Cin = 128 is checked by the caller; the wave-specialised kernel also wants exactly 128 output channels)
  if (q.g.N == 128) return launch_fused_ws(q, B, s);   // the wave-specialised kernel; other widths: the round-3 kernel
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
  // 16-row tiles pay for the rows they pad: take them only while they compute at most 1/4 more pixels than 8-row tiles
  const long long wide_px = wide_tiles * 256, narrow_px = (long long)q.tiles_x * q.tiles_y * B * TH * TW;
  if (wide_lds <= 160 * 1024 && wide_tiles >= 512 && wide_px * 4 <= narrow_px * 5 && q.g.vec_io &&
      Cin == 128 &&
      M * Cin * 2 < 0xFFFFFF00LL && M * Cout * 2 < 0xFFFFFF00LL) {   // (outputs past a descriptor's range: the narrow kernel)
    q.x_bytes = (unsigned)(M * Cin * 2);
    q.r_bytes = (unsigned)(M * Cout * 2);   // bytes of y (and of the residual, when there is one: same shape)
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
