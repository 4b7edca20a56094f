// 256 x 256 bf16 MFMA GEMM for gfx950, second structure: persistent 512-thread workgroups as gemm256.hip (same tile
// lists, operand layouts, LDS swizzles, fragment maps and fused epilogue), but the K loop is a two-group PING-PONG over
// 16 KiB half-tiles instead of a lockstep walk over a ring of 32 KiB half-units (the guide's "256^2 8-phase" schedule,
// cdna_hip_programming.md; measured against the ring in one process: profiles/r04_gemm_lab.md section 4).
//
// A K tile (64 of K) is four half-tiles: A0 / A1 = the rows of the output tile that belong to accumulator row-quadrant
// 0 / 1 of every wave, B0 / B1 likewise for columns; two K tiles are resident (2 x 64 KiB of LDS) + a 32 KiB block that
// only the epilogue stages through (160 KiB in all).  8 waves = 2 groups (wr) x 4 (wc); a wave's 128 x 64 accumulator block
// is four 64 x 32 quadrants, and a PHASE multiplies one quadrant over the K tile (16 MFMAs) out of a register subtile:
//   phase 1 (A0,B0) | 2 (A0,B1) | 3 (A1,B1) | 4 (A1,B0)
// Every phase: { fragment reads; ONE half-tile of LDS-DMA (2 pieces per wave) } barrier { 16 MFMAs } barrier.
// Group 1 runs one barrier behind group 0, so on every SIMD one wave multiplies while the other reads and requests: the
// LDS-DMA issue slots and the read latency of one wave sit under the MFMAs of the other (in the ring kernel both waves
// of a SIMD do the same thing at the same time).  With 256-cycle MFMA phases the READ phase is the critical path: one
// v_add per LDS-DMA piece, fragment addresses in registers (flipped between the buffers by XOR once per K tile), the tile
// walk incremental - and the fragment reads BALANCED over the phases: phase 1 reads A0, 2 the K tile's second B subtile,
// 3 A1, 4 the NEXT K tile's first B subtile out of the other buffer, into the fragment registers phase 3 has finished
// with - the two B fragment sets swap roles from one K tile to the next (PAR), and the loop runs two K tiles per trip.
// A tile's first K tile reads its first B subtile itself (and takes zero C operands instead of zeroed accumulators).
//
// Request stream (continuous across tiles, as the ring's): half-tiles of K tile tau are requested in phases 2, 3, 4 of
// K tile tau - 2 and phase 1 of K tile tau - 1, into the buffer that is being multiplied, each slot after its last read:
//   B row-major (rows of a wave's quadrant nq live in B half nq: 32-row slabs, permuted by the SOURCE address):
//     order B0, A0, B1, A1; counted waits: phase 3 vmcnt(10) (B0 of the next K tile has landed: read in phase 4),
//     phase 4 vmcnt(6) (the rest of the next K tile; the three youngest half-tiles stay in flight)
//   B K-major (natural columns - a k-row of a half-tile is 256 contiguous bytes; wave wc reads half wc >> 1 only):
//     order A0, B0, B1, A1; A0's reads are retired in front of phase 1's barrier and B's in front of phase 2's (their
//     slots are requested again one phase later); counted waits: phase 2 vmcnt(10) (this K tile's A1: read in phase 3),
//     phase 3 vmcnt(6) (the next K tile's A0 and B halves: read from phase 4 on)
// Every wait sits in front of a barrier, and a landed half-tile is read one phase after that barrier at the earliest.
// (The unbalanced forms - 12 / 4 / 8 / 0 reads, one vmcnt(6) per K tile in phase 4 - were removed in round 6: git history.)
// Also in here: an EDGE variant for K-major operands that end inside a 128-column half-tile (the GPT-VAE XL widths),
// the convolutions' implicit-im2col A operand (running offsets rebuilt per filter tap), single-round launches (fewer
// tiles than CUs), the weight gradients' split-K batches with the bias row sums on the A fragments, and a half-height
// last round (128-row tiles behind the last whole round of 256-row tiles, same loop with the A1 half absent).
// NHALF (the implicit-GEMM convolutions with Cout <= 128: the stride-2 Downsample layers): a 256 x 128 output tile - the
// B1 half does not exist (its pieces are requested out of range so that the counted waits keep their meaning, quadrants
// (*, B1) are neither read nor multiplied), B0 holds weight rows 0 .. 127 (wave column wc: rows 32 wc .. 32 wc + 31) and a
// wave owns a contiguous 128 x 32 block of C.
// A half-tiles are always slab-permuted (a wave's rows of quadrant mq live in A half mq: 64-row slabs, 48 of them used
// by the 192-row tile), so a wave still owns a CONTIGUOUS 128 (96) x 64 block of C and the epilogue is gemm_common.h's.
#include <cstdlib>

#include "gemm_common.h"

using namespace gemmk;

MELGPT_CLK_DECL(clk_gemm8p)

namespace {

constexpr int KU = 64;

// One LDS-DMA piece (inline asm on purpose, see gemm256.hip: the compiler must not see an LDS store it would guard with
// vmcnt(0)).  `s_nop 4`: the resource SGPRs may have been written by a VALU instruction (v_readlane out of a spill lane)
// right in front of the block - 5 wait states before a VMEM instruction reads them.
typedef u32x4 rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc4(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  return rsrc_t{(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
}
__device__ __forceinline__ void dma16(rsrc_t rs, char* lds_wave_base, unsigned voff) {
  const unsigned m0v = (unsigned)(size_t)LDS_PTR(char, lds_wave_base);
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :
               : "s"(m0v), "v"(voff), "s"(rs)
               : "memory", "m0");
}

#define P8_BARRIER() asm volatile("s_barrier" ::: "memory")


template <int ALAY, int BLAY, int MODE, int TM, bool EDGE = false, bool NHALF = false>
__global__ __launch_bounds__(512) void gemm8p_kernel(GemmParams p, int tiles_m, int tiles_n, int batch, int RN, int tail_m0) {
  constexpr int MT = TM / 2, TN = 4, BM = 32 * TM, BN = 256;
  constexpr int SLAB = 16 * MT;                      // rows of one wave's quadrant (64 or 48)
  constexpr int HT = 16384, BUF = 4 * HT;            // half-tile, one K tile's buffer
  constexpr int S_A0 = 0, S_A1 = HT, S_B0 = 2 * HT, S_B1 = 3 * HT;
  constexpr bool BNAT = BLAY == LAY_KMAJ;            // B in natural column order (see the header)
  constexpr bool BATCHED = ALAY == LAY_KMAJ && BLAY == LAY_KMAJ;  // the weight gradients' split-K batches; others: batch == 1
  static_assert(TM == 8 || (TM == 6 && ALAY != LAY_KMAJ), "the K-major A image assumes 64-row slabs");
  static_assert(!NHALF || (BLAY == LAY_ROW && !EDGE), "the 128-column tile exists for row-major B only");
  constexpr bool CONV = ALAY == LAY_CONV;  // implicit im2col
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A0 A1 B0 B1][16 KiB] + 32 KiB epilogue staging

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int G = gridDim.x;
  const int nu = (p.K + KU - 1) / KU;
  const int rows_m = tiles_m * batch;
  MELGPT_CLK_BEGIN();
  // ---- this workgroup's tile list: the static XCD-block lists of gemm256.hip (the 32 workgroups of an XCD take an
  // RM x RN block of tiles per item).  This kernel is launched only with RN > 0 (a full grid, a multiple of 8
  // workgroups) and batch == 1; everything else stays on the ring kernel.
  // Two cursors walk the list: the multiplying one and, up to two K tiles ahead, the requesting one.  A step is a few
  // scalar additions (block row / column of the XCD's next block kept incrementally): with the ring kernel's
  // decode-by-division (and its linear-order and batch branches) the plan of a tile switch cost 1.2 - 1.6 k cycles in
  // the read phase it sits in - every wave of the workgroup waits for it at the next barrier.
  // (RN == 0: a launch of at most one tile per workgroup - fewer tiles than CUs - in the linear XCD-remapped order; it
  // is walked as ONE item whose "block" is the tile itself)
  const bool single = RN == 0;
  // HALF-HEIGHT LAST ROUND (tail_m0 > 0; row-major A, 256-row tiles, no batches): the XCD-block lists cover the rows
  // below tail_m0 in whole rounds, and the rows from tail_m0 on - less than half a round of 256-row tiles - are cut into
  // 128-row tiles, ONE per workgroup, walked as the workgroup's last item: such a tile has no A1 half (a wave's 64 rows
  // are its quadrants (A0,B0), (A0,B1)), its K tiles skip phases 3 and 4's reads and MFMAs but keep their requests
  // (A1's pieces marked), waits and barriers, so the request stream runs on into it like into any other tile.
  constexpr bool TAILS = ALAY == LAY_ROW && TM == 8 && !BATCHED;
  const bool tails = TAILS && tail_m0 > 0;
  const int rows_main = tails ? tail_m0 / BM : rows_m;
  const int RNe = single ? 1 : RN;
  const int RM = single ? 1 : (G >> 3) / RNe;
  const int blocks_n = single ? tiles_n : (tiles_n + RNe - 1) / RNe;
  const int nblocks = single ? 1 : ((rows_main + RM - 1) / RM) * blocks_n;
  const int xcd = single ? 0 : blockIdx.x & 7, jslot = blockIdx.x >> 3;
  const int jm = single ? 0 : jslot / RNe, jn = single ? 0 : jslot - (jslot / RNe) * RNe;
  const int tail_tl = tails ? xcd_remap(blockIdx.x, G) : 0;                      // this workgroup's 128-row tile, if it exists
  const int tail_cnt = tails ? ((p.M - tail_m0 + 127) >> 7) * tiles_n : 0;
  const int tail_tr = tails ? tail_tl / tiles_n : 0, tail_tc = tail_tl - tail_tr * tiles_n;
  struct Walk {
    int blk, bm, bn;  // block number of this XCD's current item, its block row / column
    int tail;         // 0: in the block lists; 1: at the workgroup's 128-row tile; 2: past it
  };
  // Which blocks an XCD takes: every eighth block of the row-major block order, as gemm256.hip (a contiguous run per XCD -
  // consecutive items of a workgroup sharing their A row panel - measured no faster, profiles/r04_gemm_lab.md).
  const int blk_first = xcd;
  const int blk_end = nblocks;
  constexpr int BSTEP = 8;
  auto walk_live = [&](const Walk& k) { return k.tail == 0 ? k.blk < blk_end : k.tail == 1; };
  auto walk_tile = [&](const Walk& k, int& bz, int& m0, int& n0, bool& half) -> bool {
    half = false;
    if (TAILS && k.tail == 1) {
      bz = 0;
      m0 = tail_m0 + tail_tr * 128;
      n0 = tail_tc * BN;
      half = true;
      return true;
    }
    const int tm = k.bm * RM + jm, tn = k.bn * RNe + jn;
    bz = BATCHED ? tm / tiles_m : 0;
    m0 = (tm - bz * tiles_m) * BM;
    n0 = tn * BN;
    return tm < rows_main && tn < tiles_n;
  };
  auto walk_next = [&](Walk& k) {  // the next item that is a tile, or the first dead one
    if (k.tail) {
      k.tail = 2;
      return;
    }
    int bz, m0, n0;
    bool half;
    do {
      k.blk += BSTEP;
      k.bn += BSTEP;
      while (k.bn >= blocks_n) {
        k.bn -= blocks_n;
        ++k.bm;
      }
    } while (k.blk < blk_end && !walk_tile(k, bz, m0, n0, half));
    if (k.blk >= blk_end) k.tail = tail_tl < tail_cnt ? 1 : 2;
  };
  auto walk_first = [&](Walk& k) {
    k.tail = 0;
    if (single) {
      const int tl = xcd_remap(blockIdx.x, G);
      k.blk = 0;
      k.bm = tl / tiles_n;
      k.bn = tl - k.bm * tiles_n;
      return;
    }
    const int b0 = blk_first - BSTEP;  // (walk_next takes the first step)
    k.blk = b0;
    k.bm = b0 >= 0 ? b0 / blocks_n : 0;
    k.bn = b0 - k.bm * blocks_n;
    walk_next(k);
  };

  // ---- request cursor: the tile and K tile whose half-tiles are being requested.  A wave issues pieces w and w + 8 of
  // every half-tile (1 KiB each: 8 rows of 128 B, or 4 k-rows of 256 B).  A piece's source offset is ONE add in the loop:
  // a per-lane running offset (a_run / b_run: this lane's chunk at the cursor's K tile; K-major operands keep one per
  // piece j because their K tail cuts k-rows) + a wave-uniform constant per (half, j).  Lanes that must not load carry
  // MARK (2 GiB: past any operand this kernel accepts, and MARK + constants does not wrap), rows / columns past the
  // operand's end fall to the buffer's own range check (zeros), a ragged K tail is cut when the cursor enters the tile's
  // last K tile.
  constexpr unsigned MARK = 0x80000000u;
  constexpr int AJ = ALAY == LAY_KMAJ ? 2 : 1, BJ = BLAY == LAY_KMAJ ? 2 : 1;
  rsrc_t ra, rb;
  unsigned a_run[AJ], b_run[BJ];
  int pn0 = 0;
  const int l3 = lane >> 3;
  // row-major half-tiles: LDS row 8 (w + 8 j) + l3 holds, at physical chunk lane & 7, logical chunk r_c (row_off's swizzle)
  const int r_c = (lane & 7) ^ ((4 * w + (lane >> 4)) & 7);
  const int a_row0 = 8 * w + l3;                        // row inside the slab (used while < SLAB); slab j = piece j
  const int b_row0 = (w >> 2) * (NHALF ? 32 : 64) + 8 * (w & 3) + l3;  // tile column of B piece (half 0, j = 0): slabs of 32, j adds 128 (NHALF: 64)
  // K-major half-tiles: k-rows 4 (w + 8 j) + (lane >> 4), 16 chunks of 8 columns, logical = physical ^ (s << 1)
  const int k_row0 = 4 * w + (lane >> 4);
  const int k_lc = (lane & 15) ^ (((lane >> 4) | (((w >> 1) & 1) << 2)) << 1);
  const int ka_mn0 = (k_lc >> 3) * 128 + (k_lc & 7) * 8;  // K-major A: tile row of the chunk in half 0 (slab-permuted)
  // wave-uniform byte offsets of piece (h, j) from piece (0, 0), and of one K tile
  unsigned ca[2][2], cb[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      ca[h][j] = ALAY == LAY_KMAJ ? (unsigned)(h * 128) : (unsigned)((long long)(j * (BM / 2) + h * SLAB) * p.lda * 2);
      cb[h][j] = BLAY == LAY_KMAJ ? (unsigned)(h * 256)
                                  : (unsigned)((long long)(NHALF ? j * 64 : j * 128 + h * 32) * p.ldb * 2);
    }
  const unsigned a_kstep = ALAY == LAY_KMAJ ? (unsigned)(64 * p.lda * 2) : 128u;
  const unsigned b_kstep = BLAY == LAY_KMAJ ? (unsigned)(64 * p.ldb * 2) : 128u;
  const bool ktail = (p.K & 63) != 0;
  unsigned bh_mark[2] = {0u, 0u};  // K-major B: MARK when half h of the planned tile lies past column N (N % 128 == 0)
  // EDGE (K-major operands whose M / N is not a multiple of 128 - the GPT-VAE XL widths): a half-tile can end inside a
  // lane's columns, so every piece compares the lane's first column with what is left of the operand in that half
  int a_lim[2] = {0, 0}, b_lim[2] = {0, 0};
  // CONV: row m of A is output pixel m, a K tile is 64 channels of ONE filter tap (Cin % 64 == 0).  Per piece q = 2 h + j
  // the lane's pixel is kept as the byte offset of its image (cv_lin, + the lane's chunk) and its top-left tap's position
  // (cv_yx: y + 2048 in the high half, x + 2048 in the low half; rows past M sit far outside); when the cursor enters a
  // new tap the four running offsets are rebuilt (tap inside the image ? cv_lin + its pixel's offset : MARK), inside a tap they
  // advance by 128 bytes per K tile like any row-major operand.
  int cv_lin[CONV ? 4 : 1], cv_yx[CONV ? 4 : 1];
  unsigned a_runq[CONV ? 4 : 1];
  int ktap = 0, kci = 0;
  const int nci = CONV ? p.cC >> 6 : 1;
  auto tap_setup = [&]() {
    if constexpr (CONV) {
      const int ky = ktap / p.KW, kx = ktap - ky * p.KW;
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // (nearest x2 upsampling of the input, p.ups = 1: the tap lands on pixel (y >> 1, x >> 1))
        const int y = (cv_yx[q] >> 16) - 2048 + ky, x = (cv_yx[q] & 0xFFFF) - 2048 + kx;
        const bool ok = (unsigned)y < (unsigned)(p.cH << p.ups) && (unsigned)x < (unsigned)(p.cW << p.ups);
        a_runq[q] = ok ? (unsigned)(cv_lin[q] + (((y >> p.ups) * p.cW + (x >> p.ups)) * p.cC) * 2) : MARK;
      }
    }
  };
  int iu = 0;                      // K tile of the request cursor inside its item
  auto cut_tail = [&]() {          // the cursor is in the last, ragged K tile: lanes past K do not load
    const int k0 = iu * KU;
    if constexpr (ALAY == LAY_KMAJ) {
#pragma unroll
      for (int j = 0; j < 2; ++j) a_run[j] = k0 + k_row0 + 32 * j < p.K ? a_run[j] : MARK;
    } else {
      a_run[0] = k0 * 2 + r_c * 16 < p.K * 2 ? a_run[0] : MARK;
    }
    if constexpr (BLAY == LAY_KMAJ) {
#pragma unroll
      for (int j = 0; j < 2; ++j) b_run[j] = k0 + k_row0 + 32 * j < p.K ? b_run[j] : MARK;
    } else {
      b_run[0] = k0 * 2 + r_c * 16 < p.K * 2 ? b_run[0] : MARK;
    }
  };
  // per-lane parts of the source offsets that do not depend on the tile
  const unsigned a_lane = CONV ? 0u : ALAY == LAY_KMAJ ? (unsigned)(((long long)k_row0 * p.lda + ka_mn0) * 2)
                                           : (unsigned)((long long)a_row0 * p.lda * 2) + r_c * 16;
  const unsigned b_lane = BLAY == LAY_KMAJ ? (unsigned)(((long long)k_row0 * p.ldb + k_lc * 8) * 2)
                                           : (unsigned)((long long)b_row0 * p.ldb * 2) + r_c * 16;
  ra = make_rsrc4(p.A, p.a_bytes);
  rb = make_rsrc4(p.B, p.b_bytes);
  Walk kc;  // the requesting cursor's item
  bool kc_half = false;  // ... is a 128-row tile: slabs 64 rows apart, no A1 half
  const unsigned ca_half1 = (unsigned)((long long)64 * p.lda * 2);
  auto plan = [&]() {
    int bz = 0, pm0 = 0;
    pn0 = 0;
    const bool pok = walk_live(kc) && walk_tile(kc, bz, pm0, pn0, kc_half);  // dead: every request is out of range (zero fills nobody reads)
    if constexpr (BATCHED) {
      ra = make_rsrc4((const char*)p.A + (long long)bz * p.sA * 2, p.a_bytes);
      rb = make_rsrc4((const char*)p.B + (long long)bz * p.sB * 2, p.b_bytes);
    }
    if constexpr (CONV) {
      const int ohw = p.OH * p.OW;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = pm0 + (q & 1) * (BM / 2) + (q >> 1) * SLAB + a_row0;
        const int bb = m / ohw, rem = m - bb * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        const int y0 = oy * p.cstride - p.pad_t, x0 = ox * p.cstride - p.pad_l;
        const bool ok = pok && a_row0 < SLAB && m < p.M;
        cv_lin[q] = (int)((long long)bb * p.cH * p.cW * p.cC * 2) + r_c * 16;
        cv_yx[q] = ok ? ((y0 + 2048) << 16) | (x0 + 2048) : (30000 << 16);
      }
      ktap = kci = 0;
      tap_setup();
    } else if constexpr (ALAY == LAY_KMAJ) {
      const unsigned b0 = (pok && (EDGE || pm0 + ka_mn0 < p.M)) ? a_lane + (unsigned)(pm0 * 2) : MARK;  // (!EDGE: M % 128 == 0)
      a_lim[0] = p.M - pm0;
      a_lim[1] = p.M - pm0 - 64;
      a_run[0] = b0;
      a_run[1] = b0 == MARK ? MARK : b0 + (unsigned)(32 * p.lda * 2);
    } else {
      a_run[0] = (pok && a_row0 < SLAB) ? a_lane + (unsigned)((long long)pm0 * p.lda * 2) : MARK;
    }
    if constexpr (BLAY == LAY_KMAJ) {
      const unsigned b0 = pok ? b_lane + (unsigned)(pn0 * 2) : MARK;
      b_run[0] = b0;
      b_run[1] = b0 == MARK ? MARK : b0 + (unsigned)(32 * p.ldb * 2);
      bh_mark[0] = pn0 < p.N ? 0u : MARK;
      bh_mark[1] = pn0 + 128 < p.N ? 0u : MARK;
      b_lim[0] = p.N - pn0;
      b_lim[1] = p.N - pn0 - 128;
    } else {
      b_run[0] = pok ? b_lane + (unsigned)((long long)pn0 * p.ldb * 2) : MARK;
    }
  };
  // half-tile h of A / B for the cursor's K tile -> slot (wave-uniform LDS address)
  auto issue_a = [&](int h, char* slot) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned off = a_run[AJ == 2 ? j : 0] + ca[h][j];
      if constexpr (TAILS) {
        if (kc_half) off = h ? MARK : a_run[0] + (j ? ca_half1 : 0u);  // (wave-uniform)
      }
      if constexpr (CONV) off = a_runq[2 * h + j];
      if constexpr (EDGE && ALAY == LAY_KMAJ) off = ka_mn0 < a_lim[h] ? off : MARK;
      dma16(ra, slot + (w + 8 * j) * 1024, off);
    }
  };
  auto issue_b = [&](int h, char* slot) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned off = b_run[BJ == 2 ? j : 0] + cb[h][j];
      if constexpr (NHALF) off = h ? MARK : off;  // (no B1 half: requested out of range, the request stream keeps its count)
      if constexpr (EDGE && BLAY == LAY_KMAJ) off = k_lc * 8 < b_lim[h] ? off : MARK;
      else if constexpr (BLAY == LAY_KMAJ) off |= bh_mark[h];
      dma16(rb, slot + (w + 8 * j) * 1024, off);
    }
  };

  walk_first(kc);
  plan();
  if (ktail && nu == 1) cut_tail();
  auto advance = [&]() {
    if (++iu == nu) {
      iu = 0;
      walk_next(kc);
      plan();
    } else {
      if constexpr (CONV) {
        if (++kci == nci) {
          kci = 0;
          ++ktap;
          tap_setup();
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) a_runq[q] += 128u;
        }
      } else {
#pragma unroll
        for (int j = 0; j < AJ; ++j) a_run[j] += a_kstep;
      }
#pragma unroll
      for (int j = 0; j < BJ; ++j) b_run[j] += b_kstep;
    }
    if (ktail && iu == nu - 1) cut_tail();
  };
  // prologue: the first K tile complete, the next one without its A1 half (that is phase 1's request)
  issue_b(0, smem + S_B0);
  issue_a(0, smem + S_A0);
  issue_b(1, smem + S_B1);
  issue_a(1, smem + S_A1);
  advance();
  // (IN THE STREAM'S ORDER: the loop's counted waits count the pieces issued behind the half-tile they wait for.  With A0
  // in front of B0 for row-major B, phase 3's vmcnt(10) of a workgroup's very first K tile left B0 of the second one
  // uncovered - phase 4 read it anyway: one launch in a few thousand came out different)
  if constexpr (BNAT) {
    issue_a(0, smem + BUF + S_A0);
    issue_b(0, smem + BUF + S_B0);
  } else {
    issue_b(0, smem + BUF + S_B0);
    issue_a(0, smem + BUF + S_A0);
  }
  issue_b(1, smem + BUF + S_B1);
  asm volatile("s_waitcnt " MELGPT_VMCNT(6) ::: "memory");
  P8_BARRIER();
  int par = 0;  // buffer of the K tile that is multiplied next
  int xa[4], xb[4];
  {
    const int i = lane & 15, g = lane >> 4, q = i >> 2, pp = i & 3;
    const int sw = q | ((g & 1) << 2);                                                  // kmaj_off's s for this lane's k-rows
    const int kbase = (8 * g + q) * 256 + (pp >> 1) * 16 + (pp & 1) * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (ALAY == LAY_KMAJ) xa[e] = kbase + 128 * (wr ^ (sw >> 2)) + 32 * (e ^ (sw & 3));          // tile wr * 4 + e
      else xa[e] = row_off(wr * 64 + i, 4 * (e & 1) + g);
      if constexpr (BLAY == LAY_KMAJ) xb[e] = kbase + 128 * ((wc & 1) ^ (sw >> 2)) + 32 * (e ^ (sw & 3));    // tile (wc & 1) * 4 + e
      else xb[e] = row_off(wc * 32 + i, 4 * (e & 1) + g);
    }
  }

  Walk km;  // the multiplying cursor's item
  for (walk_first(km); walk_live(km); walk_next(km)) {
    int bz, m0, n0;
    bool km_half;
    walk_tile(km, bz, m0, n0, km_half);
    // (not zeroed: the tile's first K tile is a copy of the loop body whose first MFMA per accumulator tile takes a zero
    // C operand - 128 v_mov per wave and tile, ~500 cycles in front of the first phase, otherwise)
    f32x4 acc[TM][TN];
    u32x4 fa[MT][2], fb0[2][2], fb1[2][2];
    // ROW SUMS OF A (p.a_rowsum, weight-gradient instantiation only; see gemm256.hip): tile column tn takes K tiles tn,
    // tn + tiles_n, ...; wave wc the 16-row fragments 2 wc, 2 wc + 1 of its group's 8 - they sit in A half wc >> 1, so
    // the two MFMAs per k-step against a fragment of ones ride on the phase that has just read that half (1 or 3).
    constexpr bool CS = BATCHED && MODE == EPI_PLAIN32N && TM == 8;
    f32x4 cs[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    int cs_next = 0x7FFFFFFF;
    if constexpr (CS) {
      if (p.a_rowsum) cs_next = n0 >> 8;
    }
    auto row_sums = [&](auto mq_c, int u) {
      if constexpr (CS) {
        constexpr int MQ = decltype(mq_c)::value;
        if (u == cs_next && (wc >> 1) == MQ) {  // (wave-uniform)
          const unsigned one2 = pack_bf16x2(1.0f, 1.0f);
          const u32x4 ones = {one2, one2, one2, one2};
          static_for<2>([&](auto h_c) {
            constexpr int H = decltype(h_c)::value;
            if ((wc & 1) == H) {
#pragma unroll
              for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) mma<bf16_t>(cs[a], ones, fa[H * 2 + a][ks]);
            }
          });
        }
      }
    };

    // fragment reads: the lane's address inside the buffer of `par` is kept in registers (xa / xb, flipped between the
    // two buffers once per K tile); slot, row tile / k-step are immediates.  Row-major half-tiles: one register per
    // k-step (row tiles 2 KiB apart).  K-major half-tiles: one per 16-column tile (the swizzle XORs the tile index into
    // address bits 5-6), k-steps 8 KiB apart, the fragment's second four k-rows 1 KiB on.
    auto rd = [&](auto lay_c, const int (&x)[4], int e, int ks, int slot, int other = 0) -> u32x4 {
      if constexpr (decltype(lay_c)::value == LAY_KMAJ) {
        const char* a0 = smem + ((x[e] ^ other) + slot + ks * 8192);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a0 + 1024));
        s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(u32x4, f);
      } else {
        return *(const u32x4*)(smem + ((x[ks] ^ other) + slot + e * 2048));
      }
    };
    auto ld_a = [&](int slot) {  // this wave's quadrant rows of an A half-tile (slab wr)
      if constexpr (TAILS) {
        if (slot == S_A1 && km_half) return;  // (a 128-row tile has no A1 half; wave-uniform)
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[mt][ks] = rd(std::integral_constant<int, ALAY>{}, xa, mt, ks, slot);
    };
    auto ld_b = [&](int slot, int e0, u32x4 (&fb)[2][2], int other = 0) {  // (e0: K-major B, natural order: first tile of the subtile; other = BUF: the buffer that is NOT being multiplied)
      if constexpr (NHALF) {
        if (slot == S_B1) return;  // (compile-time at every call site)
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb[nt][ks] = rd(std::integral_constant<int, BLAY>{}, xb, e0 + nt, ks, slot, other);
    };
    auto mul = [&](auto first_c, auto mq_c, auto nq_c, u32x4 (&fb)[2][2]) {
      constexpr int MQ = decltype(mq_c)::value, NQ = decltype(nq_c)::value;
      constexpr bool FIRST = decltype(first_c)::value;
      if constexpr (TAILS && MQ == 1) {
        if (km_half) return;  // (a 128-row tile: quadrants (A1, *) do not exist; wave-uniform)
      }
      if constexpr (NHALF && NQ == 1) return;  // (a 128-column tile: quadrants (*, B1) do not exist)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            f32x4& c = acc[MQ * MT + mt][NQ * 2 + nt];
            if constexpr (FIRST) {
              if (ks == 0) c = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            mma<bf16_t>(c, fb[nt][ks], fa[mt][ks]);  // rows = n, cols = m
          }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    constexpr int RD_A = (ALAY == LAY_KMAJ ? 4 : 2) * MT;  // LDS read instructions of an A subtile / a B subtile
    constexpr int RD_B = (BLAY == LAY_KMAJ ? 8 : 4);
    static_assert(RD_A <= 15 || BNAT, "lgkmcnt is a 4-bit counter");

    auto ktile = [&](auto first_c, auto par_c, int u) {
      char* cur = smem + par * BUF;
      char* oth = smem + (BUF - par * BUF);
      if constexpr (!BNAT) {
        // Row-major B with the reads balanced as for K-major B below (8 / 4 / 8 / 4 instead of 12 / 4 / 8 / 0): phase 4
        // reads the NEXT K tile's B0 subtile out of the other buffer into the registers phase 3 has finished with (the
        // two fragment sets swap roles per K tile: PAR); B0 of the next K tile must then have landed a phase earlier:
        // vmcnt(10) in phase 3 next to the K tile's vmcnt(6) in phase 4.  No slot is re-requested one phase after its last
        // read any more, except B0 in a tile's first K tile (which reads it itself: the counted lgkmcnt stays there).
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int PAR = decltype(par_c)::value;
        auto& fbA = PAR ? fb1 : fb0;  // B0 subtile (quadrants (A0,B0), (A1,B0)), B1 subtile
        auto& fbB = PAR ? fb0 : fb1;
        if constexpr (FIRST) {
          ld_b(S_B0, 0, fbA);
          __builtin_amdgcn_sched_barrier(0);
        }
        ld_a(S_A0);
        __builtin_amdgcn_sched_barrier(0);
        issue_a(1, oth + S_A1);
        if constexpr (FIRST) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MELGPT_WAITN(RD_A)) : "memory");
        P8_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mul(first_c, c0{}, c0{}, fbA);
        P8_BARRIER();
        // phase 2: (A0, B1)
        ld_b(S_B1, 0, fbB);
        __builtin_amdgcn_sched_barrier(0);
        advance();
        issue_b(0, cur + S_B0);
        P8_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mul(first_c, c0{}, c1{}, fbB);
        P8_BARRIER();
        // phase 3: (A1, B1); B0 of the next K tile has landed (read in phase 4)
        ld_a(S_A1);
        __builtin_amdgcn_sched_barrier(0);
        issue_a(0, cur + S_A0);
        asm volatile("s_waitcnt " MELGPT_VMCNT(10) ::: "memory");
        P8_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mul(first_c, c1{}, c1{}, fbB);
        P8_BARRIER();
        // phase 4: (A1, B0); reads the next K tile's B0 subtile; the K tile's counted wait (everything but the three
        // youngest half-tiles: the rest of the next K tile)
        ld_b(S_B0, 0, fbB, BUF);
        __builtin_amdgcn_sched_barrier(0);
        issue_b(1, cur + S_B1);
        asm volatile("s_waitcnt " MELGPT_VMCNT(6) ::: "memory");
        P8_BARRIER();
        mul(first_c, c1{}, c0{}, fbA);
        P8_BARRIER();
      } else {
        // K-major B, reads BALANCED over the phases (8 / 8 / 8 / 8 instead of 16 / 8 / 8 / 0 fragment-read instructions
        // with row-major A; `ds_read_b64_tr_b16` moves half the bytes per LDS cycle of ds_read_b128, and phase 1's burst
        // outlasted the partner's 256 MFMA cycles: -7 % cycles per K tile with the burst ablated): phase 4 reads the NEXT
        // K tile's first B subtile out of the other buffer into the fragment registers that phase 3 has just finished with,
        // so the two register sets swap roles from one K tile to the next (PAR).  That read needs the next K tile's B
        // halves one phase earlier: the counted wait moves to phase 3 (vmcnt(6): everything up to B1 of the next K tile),
        // and A1 of the next K tile - younger than those - gets a wait of its own in phase 2 (vmcnt(10)), one phase
        // before its reader.  A tile's first K tile reads its first subtile itself (whatever the previous tile's last
        // phase 4 fetched is overwritten).
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int PAR = decltype(par_c)::value;
        auto& fbA = PAR ? fb1 : fb0;  // first subtile (quadrants (A0,s0) and (A1,s0)), second subtile
        auto& fbB = PAR ? fb0 : fb1;
        const int bh = wc >> 1 ? S_B1 : S_B0;
        // phase 1: (A0, s0); A0's reads retired before the barrier: its slot is requested again in phase 2
        ld_a(S_A0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FIRST) {
          ld_b(bh, 0, fbA);
          __builtin_amdgcn_sched_barrier(0);
        }
        issue_a(1, oth + S_A1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MELGPT_WAITN(FIRST ? RD_B : 0)) : "memory");
        P8_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mul(first_c, c0{}, c0{}, fbA);
        row_sums(c0{}, u);
        P8_BARRIER();
        // phase 2: (A0, s1); B's reads retired before the barrier (its slots are requested again in phases 3 and 4);
        // A1 of this K tile has landed (read in phase 3)
        ld_b(bh, 2, fbB);
        __builtin_amdgcn_sched_barrier(0);
        advance();
        issue_a(0, cur + S_A0);
        asm volatile("s_waitcnt lgkmcnt(0) " MELGPT_VMCNT(10) ::: "memory");
        P8_BARRIER();
        mul(first_c, c0{}, c1{}, fbB);
        P8_BARRIER();
        // phase 3: (A1, s1); the next K tile's A0 and B halves have landed (read from phase 4 on)
        ld_a(S_A1);
        __builtin_amdgcn_sched_barrier(0);
        issue_b(0, cur + S_B0);
        asm volatile("s_waitcnt " MELGPT_VMCNT(6) ::: "memory");
        P8_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mul(first_c, c1{}, c1{}, fbB);
        row_sums(c1{}, u);
        P8_BARRIER();
        // phase 4: (A1, s0); reads the next K tile's s0 into the registers s1 has left
        ld_b(bh, 0, fbB, BUF);
        __builtin_amdgcn_sched_barrier(0);
        issue_b(1, cur + S_B1);
        P8_BARRIER();
        mul(first_c, c1{}, c0{}, fbA);
        if constexpr (CS) {
          if (u == cs_next) cs_next += tiles_n;
        }
        P8_BARRIER();
      }
    };

    if (wr == 1) P8_BARRIER();  // group 1 runs one barrier behind group 0 inside a tile
    auto flip = [&](int u) {  // the other buffer
      par ^= 1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ALAY == LAY_KMAJ || e < 2) xa[e] ^= BUF;
        if (BLAY == LAY_KMAJ || e < 2) xb[e] ^= BUF;
      }
    };
    ktile(std::true_type{}, c0{}, 0);
    flip(0);
    {
      // (two K tiles per trip: the fragment sets swap roles from one K tile to the next, and a loop whose body picked the
      // role at run time made the register allocator keep both sets alive over the back edge - 70-275 spilled VGPRs)
      int u = 1;
      for (; u + 1 < nu; u += 2) {
        ktile(std::false_type{}, c1{}, u);
        flip(u);
        ktile(std::false_type{}, c0{}, u + 1);
        flip(u + 1);
      }
      if (u < nu) {
        ktile(std::false_type{}, c1{}, u);
        flip(u);
      }
    }
    if (wr == 0) P8_BARRIER();  // both groups run the epilogue together
    // the next tile's first K tile has landed (phase 4's wait), three half-tiles of its second are in flight: drained
    // here with the builtin (see gemm256.hip) - the epilogue's own loads are counted by the compiler from zero
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    {
      // (128-row tile: the wave's 64 rows are its first MT slabs, 64 rows apart between the two wave groups)
      long long mrow[TM];
      const int m_base = m0 + wr * ((TAILS && km_half) ? 64 : BM / 2);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) mrow[mt] = (TAILS && km_half && mt >= MT) ? -1ll : (long long)(m_base + mt * 16);
      if constexpr (NHALF) {
        // the wave's 64-column block of the shared epilogue with its columns cut behind the 32 that exist (column tiles 2
        // and 3 of the accumulator block were never multiplied; the epilogue's own range checks drop them)
        GemmParams pw = p;
        pw.N = min(p.N, n0 + wc * 32 + 32);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) acc[mt][2] = acc[mt][3] = f32x4{0.f, 0.f, 0.f, 0.f};
        epilogue_rows<bf16_t, TM, TN, MODE, false>(pw, acc, mrow, 16, n0 + wc * 32, bz, lane_e, smem + 2 * BUF + w * 4096);
      } else {
        epilogue_rows<bf16_t, TM, TN, MODE, ALAY != LAY_CONV>(p, acc, mrow, 16, n0 + wc * 64, bz, lane_e, smem + 2 * BUF + w * 4096);
      }
    }
    if constexpr (CS) {
      if (p.a_rowsum && (lane_e >> 4) == 0) {  // every tile writes its slice, zeros when it took no K tile
        float* dst = p.a_rowsum + ((long long)bz * tiles_n + (n0 >> 8)) * p.ld_rowsum;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int m = m0 + wr * 128 + (wc * 2 + a) * 16 + (lane_e & 15);
          if (m < p.M) dst[m] = cs[a][0];
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  MELGPT_CLK_END(clk_gemm8p);
}

template <int ALAY, int BLAY, int MODE, int TM, bool EDGE, bool NHALF = false>
int launch8p_e(const GemmParams& p, int tiles_m, int tiles_n, int batch, int RN, int grid, int tail_m0, hipStream_t s) {
  constexpr int LDS = 160 * 1024;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm8p_kernel<ALAY, BLAY, MODE, TM, EDGE, NHALF>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return MELGPT_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((gemm8p_kernel<ALAY, BLAY, MODE, TM, EDGE, NHALF>), dim3(grid), dim3(512), LDS, s, p, tiles_m, tiles_n, batch, RN, tail_m0);
  melgpt_count_gemm_loop(1);
  return melgpt_launch_status();
}
// Half-height last round (row-major A, 256-row tiles, one batch, XCD-block lists): if the block lists end with a partial
// round, the block rows that fill whole rounds stay 256-row tiles and the rows behind them become 128-row tiles, one per
// workgroup - returned as the first row of that tail, 0 = none.  (M = 33 920, N = 4096: 8.3 rounds of tiles = 8 rounds +
// 144 half tiles; the ninth round of 80 full tiles on 256 CUs costs a whole tile time, the half tiles ~ 0.8 of one.)
static int tail_rows(const GemmParams& p, int tiles_m, int tiles_n, int batch, int RN, int grid) {
  if (RN <= 0 || batch != 1 || (grid & 7)) return 0;
  const int RM = (grid >> 3) / RN;
  const int blocks_n = (tiles_n + RN - 1) / RN, blocks_m = (tiles_m + RM - 1) / RM;
  const int rounds = (blocks_m * blocks_n + 7) / 8;
  for (int br = blocks_m - 1; br > 0; --br) {
    if ((br * blocks_n) % 8 != 0) continue;
    if (br * blocks_n / 8 + 1 != rounds) return 0;  // (more than one round would turn into half tiles)
    const long long m0 = (long long)br * RM * 256;
    if (m0 >= p.M) return 0;
    const long long cnt = ((p.M - m0 + 127) / 128) * tiles_n;
    return cnt <= grid ? (int)m0 : 0;
  }
  return 0;
}

template <int ALAY, int BLAY, int MODE, int TM>
int launch8p(const GemmParams& p, int tiles_m, int tiles_n, int batch, int RN, int grid, hipStream_t s) {
  // K-major operands that do not end on a 128 boundary take the variant with per-piece column checks
  if constexpr (BLAY == LAY_KMAJ) {
    if ((ALAY == LAY_KMAJ && p.M % 128 != 0) || p.N % 128 != 0) {
      const int tail = (ALAY == LAY_ROW && TM == 8) ? tail_rows(p, tiles_m, tiles_n, batch, RN, grid) : 0;
      return launch8p_e<ALAY, BLAY, MODE, TM, true>(p, tiles_m, tiles_n, batch, RN, grid, tail, s);
    }
  }
  const int tail = (ALAY == LAY_ROW && TM == 8) ? tail_rows(p, tiles_m, tiles_n, batch, RN, grid) : 0;
  return launch8p_e<ALAY, BLAY, MODE, TM, false>(p, tiles_m, tiles_n, batch, RN, grid, tail, s);
}

template <int ALAY, int BLAY, int MODE>
int launch8p_tm(const GemmParams& p, int tm, int tiles_m, int tiles_n, int batch, int RN, int grid, hipStream_t s) {
  if (tm == 8) return launch8p<ALAY, BLAY, MODE, 8>(p, tiles_m, tiles_n, batch, RN, grid, s);
  if constexpr (ALAY != LAY_KMAJ) {
    if (tm == 6) return launch8p<ALAY, BLAY, MODE, 6>(p, tiles_m, tiles_n, batch, RN, grid, s);
  }
  return MELGPT_ERR_UNSUPPORTED;
}

template <int ALAY, int BLAY>
int launch8p_mode(const GemmParams& p, int mode, int tm, int tiles_m, int tiles_n, int batch, int RN, int grid, hipStream_t s) {
  switch (mode) {
    case EPI_PLAIN16: return launch8p_tm<ALAY, BLAY, EPI_PLAIN16>(p, tm, tiles_m, tiles_n, batch, RN, grid, s);
    case EPI_PLAIN16N: return launch8p_tm<ALAY, BLAY, EPI_PLAIN16N>(p, tm, tiles_m, tiles_n, batch, RN, grid, s);
    case EPI_DACT16:
      if constexpr (ALAY == LAY_ROW && BLAY == LAY_ROW) return launch8p_tm<ALAY, BLAY, EPI_DACT16>(p, tm, tiles_m, tiles_n, batch, RN, grid, s);
      break;
    case EPI_DROPR16:
      if constexpr (ALAY == LAY_ROW && BLAY == LAY_ROW) return launch8p_tm<ALAY, BLAY, EPI_DROPR16>(p, tm, tiles_m, tiles_n, batch, RN, grid, s);
      break;
    case EPI_PLAIN32:
      if constexpr (ALAY == LAY_KMAJ && BLAY == LAY_KMAJ) return launch8p_tm<ALAY, BLAY, EPI_PLAIN32>(p, tm, tiles_m, tiles_n, batch, RN, grid, s);
      break;
    case EPI_PLAIN32N:
      if constexpr (ALAY == LAY_KMAJ && BLAY == LAY_KMAJ) return launch8p_tm<ALAY, BLAY, EPI_PLAIN32N>(p, tm, tiles_m, tiles_n, batch, RN, grid, s);
      break;
    default: break;
  }
  return MELGPT_ERR_UNSUPPORTED;
}

}  // namespace


// The ping-pong form of one (layout, epilogue mode, tile height) of the persistent GEMM, or MELGPT_ERR_UNSUPPORTED when
// that combination is served by the ring kernel only (gemm256.hip then launches its own).  MELGPT_GEMM_8P=0 keeps every
// launch on the ring.
int gemmk::launch_gemm8p(const GemmParams& p, int alay, int blay, int mode, int tm, int tiles_m, int tiles_n, int batch,
                         int RN, int grid, hipStream_t s) {
  if (!melgpt_get_gemm_pingpong()) return MELGPT_ERR_UNSUPPORTED;
  if (p.a_bytes >= 0x80000000u || p.b_bytes >= 0x80000000u) return MELGPT_ERR_UNSUPPORTED;  // (the kernel's MARK offset)
  if (RN <= 0 && (long long)grid != (long long)tiles_m * tiles_n * batch) return MELGPT_ERR_UNSUPPORTED;  // (its tile walk)
  if (alay == LAY_KMAJ && blay == LAY_KMAJ)
    return launch8p_mode<LAY_KMAJ, LAY_KMAJ>(p, mode, tm, tiles_m, tiles_n, batch, RN, grid, s);
  if (batch != 1 || p.a_rowsum) return MELGPT_ERR_UNSUPPORTED;
  if (alay == LAY_ROW && blay == LAY_ROW) return launch8p_mode<LAY_ROW, LAY_ROW>(p, mode, tm, tiles_m, tiles_n, batch, RN, grid, s);
  if (alay == LAY_ROW && blay == LAY_KMAJ)
    return launch8p_mode<LAY_ROW, LAY_KMAJ>(p, mode, tm, tiles_m, tiles_n, batch, RN, grid, s);
  if (alay == LAY_CONV && blay == LAY_ROW && p.cC % 64 == 0 && mode == EPI_PLAIN16) {
    if (tm == 8) return launch8p<LAY_CONV, LAY_ROW, EPI_PLAIN16, 8>(p, tiles_m, tiles_n, batch, RN, grid, s);
    if (tm == 6) return launch8p<LAY_CONV, LAY_ROW, EPI_PLAIN16, 6>(p, tiles_m, tiles_n, batch, RN, grid, s);
  }
  return MELGPT_ERR_UNSUPPORTED;
}

// The implicit-GEMM convolutions with at most 128 output channels (Downsample: 3x3, stride 2, 128 -> 128 at 80 x 848 and
// 40 x 424 - 631 TFLOP/s on the 128 x 128 kernel) as 256 x 128 tiles of the ping-pong loop (NHALF).  One tile column, the
// block lists degenerate to RM = workgroups per XCD consecutive tile rows; fewer tiles than CUs: one tile per workgroup.
// MELGPT_ERR_UNSUPPORTED: the caller's 128 x 128 kernel serves the launch.
int gemmk::launch_conv8p_n128(const GemmParams& p, hipStream_t s) {
  if (!melgpt_get_gemm_pingpong()) return MELGPT_ERR_UNSUPPORTED;
  if (p.N > 128 || p.cC % 64 != 0 || p.K < 512 || !p.vec_io || p.out_f32 || p.accumulate || p.C2 || p.drop_scale != 0.f ||
      p.act != MELGPT_ACT_NONE || p.a_bytes >= 0x80000000u || p.b_bytes >= 0x80000000u)
    return MELGPT_ERR_UNSUPPORTED;
  int ncu = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
    return MELGPT_ERR_LAUNCH;
  if (ncu - melgpt_get_reserved_cus() >= 8) ncu -= melgpt_get_reserved_cus();
  const int tiles_m = (p.M + 255) / 256;
  if (tiles_m < ncu / 2) return MELGPT_ERR_UNSUPPORTED;  // (a small image: the 128 x 128 kernel's two workgroups per CU)
  int grid = tiles_m < ncu ? tiles_m : ncu, RN = 0;
  if (grid == ncu && ncu % 8 == 0) RN = 1;
  else if (grid != tiles_m) return MELGPT_ERR_UNSUPPORTED;
  return launch8p_e<LAY_CONV, LAY_ROW, EPI_PLAIN16, 8, false, true>(p, tiles_m, 1, 1, RN, grid, 0, s);
}
