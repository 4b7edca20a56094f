// Wide-tile bf16 MFMA GEMM for gfx950: 256x256 (or 256x128) output tile per 512-thread workgroup, operands
// streamed HBM/L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPR round trip, out-of-range lanes
// deposit zeros), two 64 KiB (48 KiB) LDS stages, v_mfma_f32_16x16x32_bf16.
//
// Why a second kernel: a 128x128 tile moves 64 FLOP per byte staged from L2, i.e. ~39 TB/s of L2->LDS traffic at
// the 2.5 PFLOP/s MFMA peak - more than the ~34 TB/s the eight L2s deliver; 256x256 halves that (128 FLOP/B), and
// LDS-DMA frees the 32 staging VGPRs so a wave can hold its 128x64 accumulator block (128 VGPRs).
// Same operand layouts, swizzles, fragment maps and fused epilogue as gemm.hip (gemm_common.h); the f32 parity
// lane stays on gemm.hip.
//   LDS-DMA writes are lane-linear (wave-uniform base + 16*lane), so the XOR swizzle is applied to the SOURCE
//   address (which chunk a lane fetches) and again on the fragment read - both sides or neither.
#include "gemm_common.h"

using namespace gemmk;

namespace {

constexpr int KSTEP = 64;  // bf16 elements per K step = 128 bytes per ROW-layout tile row

template <int MN>
__device__ __forceinline__ int kmaj_off(int krow, int lc) {
  // K-major tile: 64 k-rows of MN bf16 (MN*2 bytes); 32-byte column blocks XOR-ed so that the 8 rows a half-wave
  // touches in one ds_read_b64_tr_b16 land on 8 different 32-byte slots of the 256-byte bank row
  const int s = (krow & 3) | (((krow >> 3) & 1) << 2);
  return krow * (MN * 2) + ((lc ^ (s << 1)) << 4);
}

template <int LAY, int MN>
__device__ __forceinline__ u32x4 load_frag(const char* tile, int st, int ks, int lane) {
  const int i = lane & 15, g = lane >> 4;
  if constexpr (LAY != LAY_KMAJ) {
    return *(const u32x4*)(tile + row_off(st * 16 + i, 4 * ks + g));
  } else {
    const int q = i >> 2, pp = i & 3;
    const int k0 = 32 * ks + 8 * g + q;
    const int lc = 2 * st + (pp >> 1);
    const char* a0 = tile + kmaj_off<MN>(k0, lc) + (pp & 1) * 8;
    const char* a1 = tile + kmaj_off<MN>(k0 + 4, lc) + (pp & 1) * 8;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1));
    s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(u32x4, f);
  }
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds_wave_base, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 0);
}

// WM x WN waves, each TM x TN accumulator tiles of 16x16  ->  BM = 16*WM*TM, BN = 16*WN*TN
template <int ALAY, int BLAY, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * WM * WN) void gemm256_kernel(GemmParams p) {
  constexpr int NW = WM * WN, BM = 16 * WM * TM, BN = 16 * WN * TN;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = A_BYTES / 1024, B_INSTR = B_BYTES / 1024;  // 1 KiB LDS-DMA pieces per tile
  constexpr int A_PER = A_INSTR / NW, B_PER = B_INSTR / NW;
  static_assert(A_INSTR % NW == 0 && B_INSTR % NW == 0, "tile must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A | B]

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w / WN, wn = w % WN;
  const int tilesN = (p.N + BN - 1) / BN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wg / tilesN) * BM, n0 = (wg % tilesN) * BN;
  const int bz = blockIdx.z;

  const __amdgpu_buffer_rsrc_t ra = make_rsrc((const char*)p.A + (long long)bz * p.sA * 2, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc((const char*)p.B + (long long)bz * p.sB * 2, p.b_bytes);

  // ---- per-lane source plan for this wave's LDS-DMA pieces (piece j of the wave = tile piece w + NW*j)
  unsigned a_base[A_PER], b_base[B_PER];
  int a_kc[A_PER], b_kc[B_PER];      // ROW: byte offset of the lane's chunk inside the 128-byte K slice
  int cy[A_PER], cx[A_PER];          // CONV
#pragma unroll
  for (int j = 0; j < A_PER; ++j) {
    const int piece = w + NW * j;
    if constexpr (ALAY == LAY_KMAJ) {
      constexpr int RB = BM * 2, RPP = 1024 / RB;  // rows per piece
      const int krow = piece * RPP + (lane * 16) / RB, pch = ((lane * 16) % RB) >> 4;
      const int s = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int lc = pch ^ (s << 1);
      a_base[j] = (m0 + lc * 8 < p.M) ? (unsigned)(((long long)krow * p.lda + m0) * 2) + lc * 16 : OOB;
      a_kc[j] = 0;
    } else {
      const int row = piece * 8 + (lane >> 3), pch = lane & 7;
      const int c = pch ^ ((row >> 1) & 7);
      a_kc[j] = c * 16;
      if constexpr (ALAY == LAY_ROW) {
        a_base[j] = (unsigned)(((long long)(m0 + row) * p.lda) * 2) + c * 16;
      } else {
        const int m = m0 + row, ohw = p.OH * p.OW;
        const int bb = m / ohw, rem = m - bb * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        cy[j] = (m < p.M) ? oy * p.cstride - p.pad_t : -100000;
        cx[j] = ox * p.cstride - p.pad_l;
        a_base[j] = (unsigned)((long long)bb * p.cH * p.cW * p.cC * 2) + c * 16;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < B_PER; ++j) {
    const int piece = w + NW * j;
    if constexpr (BLAY == LAY_KMAJ) {
      constexpr int RB = BN * 2, RPP = 1024 / RB;
      const int krow = piece * RPP + (lane * 16) / RB, pch = ((lane * 16) % RB) >> 4;
      const int s = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int lc = pch ^ (s << 1);
      b_base[j] = (n0 + lc * 8 < p.N) ? (unsigned)(((long long)krow * p.ldb + n0) * 2) + lc * 16 : OOB;
      b_kc[j] = 0;
    } else {
      const int row = piece * 8 + (lane >> 3), pch = lane & 7;
      const int c = pch ^ ((row >> 1) & 7);
      b_kc[j] = c * 16;
      b_base[j] = (unsigned)(((long long)(n0 + row) * p.ldb) * 2) + c * 16;
    }
  }

  auto issue = [&](int kt, int buf) {
    char* sa = smem + buf * STAGE;
    char* sb = sa + A_BYTES;
    const int k0 = kt * KSTEP;
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
      unsigned off;
      if constexpr (ALAY == LAY_ROW) {
        off = (k0 * 2 + a_kc[j] < p.K * 2) ? a_base[j] + k0 * 2 : OOB;
      } else if constexpr (ALAY == LAY_KMAJ) {
        off = (a_base[j] == OOB) ? OOB : a_base[j] + (unsigned)((long long)k0 * p.lda * 2);
      } else {
        const int tap = k0 / p.cC, ci0 = k0 - tap * p.cC;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        int iy = cy[j] + ky, ix = cx[j] + kx;
        const bool ok = iy >= 0 && ix >= 0 && iy < (p.cH << p.ups) && ix < (p.cW << p.ups);
        iy >>= p.ups;
        ix >>= p.ups;
        off = ok ? a_base[j] + (unsigned)((iy * p.cW + ix) * p.cC + ci0) * 2 : OOB;
      }
      dma16(ra, sa + (w + NW * j) * 1024, off);
    }
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      unsigned off;
      if constexpr (BLAY == LAY_ROW) {
        off = (k0 * 2 + b_kc[j] < p.K * 2) ? b_base[j] + k0 * 2 : OOB;
      } else {
        off = (b_base[j] == OOB) ? OOB : b_base[j] + (unsigned)((long long)k0 * p.ldb * 2);
      }
      dma16(rb, sb + (w + NW * j) * 1024, off);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + KSTEP - 1) / KSTEP;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);  // lands while this tile is being multiplied
    const char* sa = smem + cur * STAGE;
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fa[TM], fb[TN];
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) fb[nt] = load_frag<BLAY, BN>(sb, wn * TN + nt, ks, lane);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) fa[mt] = load_frag<ALAY, BM>(sa, wm * TM + mt, ks, lane);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) mma<bf16_t>(acc[mt][nt], fb[nt], fa[mt]);  // rows = n, cols = m
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  epilogue<bf16_t, TM, TN>(p, acc, m0 + wm * TM * 16, n0 + wn * TN * 16, bz, lane, smem + w * 4096);
}

template <int ALAY, int BLAY, int WM, int WN, int TM, int TN>
int launch_cfg(const GemmParams& p, int batch, hipStream_t s) {
  constexpr int BM = 16 * WM * TM, BN = 16 * WN * TN, LDS = 2 * (BM + BN) * 128;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm256_kernel<ALAY, BLAY, WM, WN, TM, TN>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return MELGPT_ERR_LAUNCH;
    attr = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL((gemm256_kernel<ALAY, BLAY, WM, WN, TM, TN>), dim3(tiles, 1, batch), dim3(64 * WM * WN), LDS, s, p);
  return melgpt_launch_status();
}

template <int ALAY, int BLAY>
int launch_lay(const GemmParams& p, int batch, int cfg, hipStream_t s) {
  if (cfg == 3) return launch_cfg<ALAY, BLAY, 2, 4, 8, 4>(p, batch, s);  // 256 x 256
  return launch_cfg<ALAY, BLAY, 4, 2, 4, 4>(p, batch, s);                // 256 x 128
}

}  // namespace

int gemmk::launch_gemm256(const GemmParams& p, int alay, int blay, int batch, int tile_cfg, hipStream_t s) {
  if (tile_cfg != 2 && tile_cfg != 3) return MELGPT_ERR_UNSUPPORTED;
  if (alay == LAY_ROW && blay == LAY_ROW) return launch_lay<LAY_ROW, LAY_ROW>(p, batch, tile_cfg, s);
  if (alay == LAY_ROW && blay == LAY_KMAJ) return launch_lay<LAY_ROW, LAY_KMAJ>(p, batch, tile_cfg, s);
  if (alay == LAY_KMAJ && blay == LAY_KMAJ) return launch_lay<LAY_KMAJ, LAY_KMAJ>(p, batch, tile_cfg, s);
  if (alay == LAY_CONV && blay == LAY_ROW) return launch_lay<LAY_CONV, LAY_ROW>(p, batch, tile_cfg, s);
  return MELGPT_ERR_UNSUPPORTED;
}
