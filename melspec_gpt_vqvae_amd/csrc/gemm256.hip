// Wide-tile bf16 MFMA GEMM for gfx950: persistent 512-thread workgroups, 256x256 output tiles, operands streamed
// L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPR round trip, out-of-range lanes deposit zeros) into
// a ring of five 32 KiB half-unit slots, v_mfma_f32_16x16x32_bf16.
//
// Why a second kernel: a 128x128 tile moves 64 FLOP per byte staged from L2 - at the 2.5 PFLOP/s MFMA peak that is
// ~39 TB/s of L2->LDS traffic, and the measured ceiling of that path is ~30 TB/s with everything hitting in L2
// (tools/lab/dma_lab.hip).  256x256 halves the traffic (128 FLOP/B); LDS-DMA frees the staging VGPRs so a wave
// can hold its 128x64 accumulator block (128 VGPRs).
//
// Synchronisation: per K-unit ONE counted `s_waitcnt vmcnt(n)` (this wave's pieces of the unit have landed, the
// younger half-unit keeps flying) and ONE raw `s_barrier` (everybody's pieces landed + everybody is done reading
// the slots that are refilled next).  __syncthreads() is not used in the loop: its fence makes the compiler
// drain vmcnt(0), which would serialise the ring.
// Same operand layouts, fragment maps and fused epilogue as gemm.hip (gemm_common.h); the f32 parity lane stays
// on gemm.hip.
//   LDS-DMA writes are lane-linear (wave-uniform base + 16*lane), so the XOR swizzle is applied to the SOURCE
//   address (which chunk a lane fetches) and again on the fragment read - both sides or neither.
#include <cstdlib>

#include "gemm_common.h"

using namespace gemmk;


namespace {


constexpr int KU = 64;  // bf16 elements of K per unit = 128 bytes per ROW-layout row (full L2 lines per request)

template <int MN>
__device__ __forceinline__ int kmaj_off(int krow, int lc) {
  // K-major half-unit: 64 k-rows of MN bf16 (MN*2 bytes); 32-byte column blocks XOR-ed so that the 8 rows a
  // half-wave touches in one ds_read_b64_tr_b16 land on 8 different 32-byte slots of the 256-byte bank row
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

// One LDS-DMA piece: 64 lanes x 16 bytes from the buffer `rs` (per-lane byte offset voff, out of range -> zeros) to the
// 1 KiB of LDS at lds_wave_base.  Written as inline asm ON PURPOSE: hipcc models the builtin as an LDS store that is
// complete only at vmcnt(0) and put a full `s_waitcnt vmcnt(0)` into every K unit (in front of the first LDS read, and in
// front of the first write to the registers next to the piece's address register), draining the half-unit the ring is
// built to keep in flight.  Instructions the compiler cannot see get no such waits; ordering is ours: the counted wait +
// barrier of wait_vm_barrier, and the drain in front of the epilogue.  (Invisible operations can only make the
// compiler's own vmcnt waits for the epilogue's loads longer, never shorter: the counter is in order.)
// Hazards are ours as well: the block opens with `s_nop 2` because the resource SGPRs may just have been restored from a
// spill lane by v_readlane (VALU write of an SGPR -> VMEM read: 5 wait states, which the hazard recogniser only inserts
// for instructions it can see), and keeps one wait state between the m0 write and the load.
typedef u32x4 rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc4(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  return rsrc_t{(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};  // stride 0, raw buffer (as make_rsrc)
}
__device__ __forceinline__ void dma16(rsrc_t rs, char* lds_wave_base, unsigned voff) {
  const unsigned m0v = (unsigned)(size_t)LDS_PTR(char, lds_wave_base);
  // Cache policy of the requests (measured in round 3, then fixed): " sc1" (served by L2 without allocating in the CU's vector L1) measured
  // +0.5 .. 2.5 % per shape on constant operands and NEUTRAL in the training step (GEMM family 88.8-89.4 ms per step
  // either way, three alternating runs) while FETCH_SIZE read 15 % more bytes per step; " nt" -3 .. -15 % (the panels are
  // re-read out of L2 by the other tiles of the XCD's block).  Default policy kept.  profiles/r03_gemm_lab.md
  asm volatile("s_nop 2\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :
               : "s"(m0v), "v"(voff), "s"(rs)
               : "memory", "m0");
}

// wait until at most N of this wave's vector-memory operations are outstanding, then workgroup barrier
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(MELGPT_WAITN(N)) : "memory");
}

// 256 x 256 output tile, 8 waves as 2 (M) x 4 (N), each wave 128 x 64 = 8 x 4 accumulator tiles of 16 x 16.
//
// LDS = five 32 KiB slots holding HALF-units: H(2u) = the A operand of K-unit u (256 rows x 128 B), H(2u+1) = its
// B operand; H(j) lives in slot j mod 5.  While unit u is multiplied (two slots), H(2u+2..2u+4) - 96 KiB - are in
// flight or landed; when unit u is done its two slots are refilled with H(2u+5), H(2u+6).  (Measured with
// tools/lab/dma_lab.hip: the L2 -> LDS path needs ~96 KiB in flight per CU and full 128-byte lines per request to
// reach ~30 TB/s; 64 KiB in flight gives ~20, 64-byte row slices ~17.)
//
// PERSISTENT: gridDim.x workgroups (one per CU) walk their tile lists; the ring does not stop at tile boundaries,
// so the next tile's first three half-units land while this tile's accumulators go through the epilogue, and the
// epilogue's stores drain under the next tile's MFMAs.  The epilogue stages through the slot that is refilled next.
//
// Tile order: the 32 workgroups of one XCD (blockIdx & 7, round-robin dispatch) take an RM x RN block of tiles, so
// one L2 serves RM row panels of A and RN column panels of B instead of 1 + 32.
template <int ALAY, int BLAY, int MODE, int TM>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmParams p, int tiles_m, int tiles_n, int batch, int RN) {
  constexpr int WN = 4, TN = 4, NW = 8, BM = 32 * TM, BN = 256;  // TM = 8: 256 x 256; TM = 6: 192 x 256 (A row-major only)
  constexpr int HALF = 256 * 128, NSLOT = 5;
  constexpr int PER = HALF / 1024 / NW;   // LDS-DMA pieces per wave per B half-unit (4)
  constexpr int PER_A = BM * 128 / 1024 / NW;  // ... per A half-unit (4 or 3)
  static_assert(TM == 8 || (TM == 6 && ALAY != LAY_KMAJ), "the K-major A image assumes 512-byte k-rows");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [5][32 KiB]

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w / WN, wn = w % WN;
  const int G = gridDim.x;
  const int nu = (p.K + KU - 1) / KU;
  const int rows_m = tiles_m * batch;  // tile rows over all batches
  const int total = rows_m * tiles_n;

  // ---- this workgroup's tile list: item r -> (batch, m0, n0) or "not a tile"
  const int RM = RN > 0 ? (G >> 3) / RN : 0;
  const int blocks_n = RN > 0 ? (tiles_n + RN - 1) / RN : 0;
  const int nblocks = RN > 0 ? ((rows_m + RM - 1) / RM) * blocks_n : 0;
  const int xcd = blockIdx.x & 7, jslot = blockIdx.x >> 3;
  auto live = [&](int r) {
    return RN > 0 ? r * 8 + xcd < nblocks : r * G + (int)blockIdx.x < total;
  };
  auto decode = [&](int r, int& bz, int& m0, int& n0) -> bool {
    int tm, tn;
    if (RN > 0) {
      const int blk = r * 8 + xcd;
      tm = (blk / blocks_n) * RM + jslot / RN;
      tn = (blk % blocks_n) * RN + jslot % RN;
      if (tm >= rows_m || tn >= tiles_n) return false;
    } else {
      const int gsz = min(G, total - r * G);
      const int tl = r * G + xcd_remap(blockIdx.x, gsz);
      tm = tl / tiles_n;
      tn = tl - tm * tiles_n;
    }
    bz = tm / tiles_m;
    m0 = (tm - bz * tiles_m) * BM;
    n0 = tn * BN;
    return true;
  };
  auto next_item = [&](int r) {  // first item after r that is a tile, or the first dead one
    int bz, m0, n0;
    do ++r;
    while (live(r) && !decode(r, bz, m0, n0));
    return r;
  };
  // ---- issue cursor: source plan of the tile whose half-units are being requested.  Piece j of a wave is piece
  // w + 8 j of the half-unit; its rows are 64 j rows (ROW) / 16 j k-rows (K-major) below piece 0's, and the swizzled
  // chunk a lane fetches is the same for every j, so one base offset per operand describes all four pieces.
  rsrc_t ra, rb;
  unsigned a_base0, b_base0;  // byte offset of this lane's chunk in piece 0 at k = 0, or OOB when its column is out
  int pm0 = 0, pn0 = 0;       // tile origin of the plan
  int cv_y[ALAY == LAY_CONV ? 4 : 1], cv_x[ALAY == LAY_CONV ? 4 : 1];  // LAY_CONV: per piece, top-left tap position
  unsigned cv_img[ALAY == LAY_CONV ? 4 : 1];                           // ... and byte offset of the pixel's image
  bool pok = false;
  // ROW: rows 8 w + 64 j + (lane >> 3), logical chunk = physical chunk ^ ((row >> 1) & 7)  (row_off's swizzle)
  const int r_row0 = 8 * w + (lane >> 3), r_c = (lane & 7) ^ ((4 * w + (lane >> 4)) & 7);
  // K-major: k-rows 2 w + 16 j + (lane >> 5), 32 column chunks, logical = physical ^ (s << 1)  (kmaj_off's swizzle)
  const int k_row0 = 2 * w + (lane >> 5);
  const int k_lc = (lane & 31) ^ (((k_row0 & 3) | (((k_row0 >> 3) & 1) << 2)) << 1);
  auto plan = [&](int r) {
    int bz = 0;
    pm0 = pn0 = 0;
    pok = live(r) && decode(r, bz, pm0, pn0);  // dead: every request is out of bounds (zero fills nobody reads)
    ra = make_rsrc4((const char*)p.A + (long long)bz * p.sA * 2, p.a_bytes);
    rb = make_rsrc4((const char*)p.B + (long long)bz * p.sB * 2, p.b_bytes);
    if constexpr (ALAY == LAY_KMAJ) {
      a_base0 = (pok && pm0 + k_lc * 8 < p.M) ? (unsigned)(((long long)k_row0 * p.lda + pm0) * 2) + k_lc * 16 : OOB;
    } else if constexpr (ALAY == LAY_CONV) {
      // implicit im2col: row m of A is output pixel m; per piece keep the pixel's image offset and its top-left tap
      // position (a far-out y marks rows past M, so that every tap of such a row is out of the image)
      a_base0 = pok ? (unsigned)(r_c * 16) : OOB;
      const int ohw = p.OH * p.OW;
#pragma unroll
      for (int j = 0; j < PER_A; ++j) {
        const int m = pm0 + r_row0 + 64 * j;
        const int bb = m / ohw, rem = m - bb * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        cv_y[j] = (pok && m < p.M) ? oy * p.cstride - p.pad_t : -100000;
        cv_x[j] = ox * p.cstride - p.pad_l;
        cv_img[j] = (unsigned)((long long)bb * p.cH * p.cW * p.cC * 2);
      }
    } else {
      a_base0 = pok ? (unsigned)(((long long)(pm0 + r_row0) * p.lda) * 2) + r_c * 16 : OOB;
    }
    if constexpr (BLAY == LAY_KMAJ)
      b_base0 = (pok && pn0 + k_lc * 8 < p.N) ? (unsigned)(((long long)k_row0 * p.ldb + pn0) * 2) + k_lc * 16 : OOB;
    else
      b_base0 = pok ? (unsigned)(((long long)(pn0 + r_row0) * p.ldb) * 2) + r_c * 16 : OOB;
  };
  // piece j (of PER) of one operand's half-unit u: `lay`-layout source with leading dimension ld, tile origin o0 of
  // extent lim
  auto issue_piece = [&](auto lay, rsrc_t rs, unsigned base0, long long ld, int o0, int lim, int u,
                         char* dst, int j) {
    const int k0 = u * KU;
    unsigned off;
    if constexpr (decltype(lay)::value == LAY_KMAJ)
      off = (base0 != OOB && k0 + k_row0 + 16 * j < p.K) ? base0 + (unsigned)((long long)(k0 + 16 * j) * ld * 2) : OOB;
    else
      off = (base0 != OOB && o0 + r_row0 + 64 * j < lim && k0 * 2 + r_c * 16 < p.K * 2)
                ? base0 + (unsigned)((long long)(64 * j) * ld * 2) + k0 * 2
                : OOB;
    dma16(rs, dst + (w + NW * j) * 1024, off);
  };
  auto issue_a_piece = [&](int u, char* dst, int j) {
    if constexpr (ALAY == LAY_CONV) {
      // one K unit = 64 channels of one filter tap (Cin % 64 == 0): the row's source is the tap-shifted input pixel;
      // padding, the stride-2 asymmetric pad and nearest-x2 upsampling are address predicates (zero fill)
      const int k0 = u * KU;
      const int tap = k0 / p.cC, ci0 = k0 - tap * p.cC;
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      int iy = cv_y[j] + ky, ix = cv_x[j] + kx;
      const bool ok = a_base0 != OOB && k0 < p.K && iy >= 0 && ix >= 0 && iy < (p.cH << p.ups) && ix < (p.cW << p.ups);
      iy >>= p.ups;
      ix >>= p.ups;
      const unsigned off = ok ? cv_img[j] + (unsigned)(((iy * p.cW + ix) * p.cC + ci0) * 2) + a_base0 : OOB;
      dma16(ra, dst + (w + NW * j) * 1024, off);
    } else {
      issue_piece(std::integral_constant<int, ALAY>{}, ra, a_base0, p.lda, pm0, p.M, u, dst, j);
    }
  };
  auto issue_b_piece = [&](int u, char* dst, int j) {
    issue_piece(std::integral_constant<int, BLAY>{}, rb, b_base0, p.ldb, pn0, p.N, u, dst, j);
  };

  int first_item = -1;
  first_item = next_item(first_item);
  // Issue cursor.  The stream of half-units is A(0) B(0) A(1) | B(1) A(2) | B(2) A(3) | ... : three up front, then
  // every K-unit iteration requests the B half of unit `iu` and the A half of the unit after it.
  int ir = first_item, iu = 0;  // item and K-unit of the next B half to request
  int fill = 0;                 // slot of the next half-unit to request
  plan(ir);
  auto next_slot = [&]() { fill = fill + 1 == NSLOT ? 0 : fill + 1; };
  auto next_unit = [&]() {      // the B half of unit iu is out: move on (possibly to the next tile)
    if (++iu == nu) {
      iu = 0;
      ir = next_item(ir);
      plan(ir);
    }
  };
#pragma unroll
  for (int j = 0; j < PER_A; ++j) issue_a_piece(0, smem, j);           // A(0) -> slot 0
#pragma unroll
  for (int j = 0; j < PER; ++j) issue_b_piece(0, smem + HALF, j);      // B(0) -> slot 1
  next_unit();
#pragma unroll
  for (int j = 0; j < PER_A; ++j) issue_a_piece(iu, smem + 2 * HALF, j);  // A(1) -> slot 2
  fill = 3;
  int slot = 0;  // slot of the A half of the unit being multiplied
  bool first = true;
  for (int r = first_item; live(r); r = next_item(r)) {
    int bz, m0, n0;
    decode(r, bz, m0, n0);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ROW SUMS OF A (p.a_rowsum, weight-gradient instantiation only): the bias gradient is the sum over the reduction
    // rows of the same dY the K loop streams through LDS anyway - one more MFMA per A fragment against a fragment of
    // ones (every output row then holds sum_k A[m][k]) instead of a pass of its own over the tensor.  The tiles of one
    // tile row share the work: tile column tn takes K units tn, tn + tiles_n, ... (each of the four waves of a row group
    // two of its eight 16-row fragments: 8 registers - with all eight in one wave the kernel spilled 31), and writes its
    // partial sums as row (batch, tn) of p.a_rowsum; the host adds the rows in fixed order.
    constexpr bool CS = ALAY == LAY_KMAJ && BLAY == LAY_KMAJ && MODE == EPI_PLAIN32N && TM == 8;
    constexpr int CSN = TM / WN;
    f32x4 cs[CS ? CSN : 1];
    int cs_next = 0x7FFFFFFF;
    if constexpr (CS) {
#pragma unroll
      for (int a = 0; a < CSN; ++a) cs[a] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.a_rowsum) cs_next = n0 >> 8;
    }

    auto unit = [&](int u) {
      // H(2u), H(2u+1) must have landed for every wave and every wave must be past its reads of unit u-1, whose two
      // slots are refilled now.  One younger half-unit (an A half: PER_A operations) may still be in flight.  The half-units
      // requested before the previous tile's epilogue were drained there (vmcnt(0)): barrier only.
      if (first || u > 0) wait_vm_barrier<PER_A>();
      else asm volatile("s_barrier" ::: "memory");
      // The eight LDS-DMA pieces of this iteration (B of unit iu, then A of the unit after) are issued ONE AT A TIME
      // between groups of MFMAs: a piece occupies the issuing wave for 60-180 cycles, and all 64 of a workgroup's
      // pieces issued together right after the barrier stall every wave for as long as the whole K-unit's MFMAs
      // take (measured: K-unit time = MFMA time + DMA time).  Spread out, the SIMD's other wave keeps the matrix
      // pipe busy meanwhile.
      char* dst_b = smem + fill * HALF;
      next_slot();
      char* dst_a = smem + fill * HALF;
      next_slot();
      const char* sa = smem + slot * HALF;
      const char* sb = smem + (slot + 1 == NSLOT ? 0 : slot + 1) * HALF;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // all 12 fragment reads of this k-step go out back to back, then the MFMAs run at raised priority: the LDS
        // sees a short read burst and is otherwise free for the LDS-DMA writes that are landing.  (Left to itself
        // the compiler reads one A fragment at a time, each behind an lgkmcnt(0), to save registers.)
        u32x4 fa[TM], fb[TN];
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) fb[nt] = load_frag<BLAY, BN>(sb, wn * TN + nt, ks, lane);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) fa[mt] = load_frag<ALAY, BM>(sa, wm * TM + mt, ks, lane);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
          for (int nt = 0; nt < TN; ++nt) {
            mma<bf16_t>(acc[mt][nt], fb[nt], fa[mt]);  // rows = n, cols = m
          }
          // piece k of NP goes out after MFMA row floor((k + 1) TM / NP) - 1
          const int np = ks == 0 ? PER : PER_A;
          const int k_here = ((mt + 1) * np) / TM - (mt * np) / TM;  // 0 or 1 pieces after this row
          const int k_idx = (mt * np) / TM;
          if (k_here > 0) {
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) issue_b_piece(iu, dst_b, k_idx);
            else issue_a_piece(iu, dst_a, k_idx);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if constexpr (CS) {
          if (u == cs_next) {  // (wave-uniform)
            const unsigned one2 = pack_bf16x2(1.0f, 1.0f);
            const u32x4 ones = {one2, one2, one2, one2};
            static_for<WN>([&](auto wn_c) {  // (wave-uniform: one copy of the two MFMAs per wave column)
              constexpr int W0 = decltype(wn_c)::value;
              if (wn == W0) {
#pragma unroll
                for (int a = 0; a < CSN; ++a) mma<bf16_t>(cs[a], ones, fa[W0 * CSN + a]);
              }
            });
            if (ks == 1) cs_next += tiles_n;
          }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) next_unit();
      }
      slot = slot + 2 >= NSLOT ? slot + 2 - NSLOT : slot + 2;
    };
    // The first unit of a tile is peeled off the loop: the registers that held the previous tile's epilogue loads are
    // re-used by the loop body, the compiler guards their first re-use with a vector-memory wait, and inside the loop
    // that wait would run - and drain the ring - on every unit.
    unit(0);
    for (int u = 1; u < nu; ++u) unit(u);
    first = false;
    // this wave's pieces of the next three half-units have landed; every wave is past its reads of the last unit,
    // whose A slot ( = `fill`, refilled at the next barrier) is the epilogue's staging block
    // (the drain is written with the builtin, not inline asm, so that the compiler's own wait-count bookkeeping
    // knows that no vector-memory operation is outstanding here: with LDS-DMA pieces "possibly in flight" it would
    // put a full vmcnt(0) in front of every use of an ordinary load in the epilogue - the residual slabs)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
    asm volatile("s_barrier" ::: "memory");
    // the epilogue's per-lane offsets are tile-invariant; hoisted out of the tile loop they would sit in scratch
    // (the K loop owns the whole register file) and every reload is a memory round trip - recompute them per tile
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    epilogue<bf16_t, TM, TN, MODE, ALAY != LAY_CONV>(p, acc, m0 + wm * TM * 16, n0 + wn * TN * 16, bz, lane_e, smem + fill * HALF + w * 4096);
    if constexpr (CS) {
      if (p.a_rowsum && (lane_e >> 4) == 0) {  // every tile writes its slice, zeros when it took no K unit
        float* dst = p.a_rowsum + ((long long)bz * tiles_n + (n0 >> 8)) * p.ld_rowsum;
#pragma unroll
        for (int a = 0; a < CSN; ++a) {
          const int m = m0 + wm * TM * 16 + (wn * CSN + a) * 16 + (lane_e & 15);
          if (m < p.M) dst[m] = cs[a][0];
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

static int device_cus();

template <int ALAY, int BLAY, int MODE, int TM>
int launch_tm(const GemmParams& p, int batch, int ncu, hipStream_t s) {
  constexpr int LDS = 5 * 256 * 128;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm256_kernel<ALAY, BLAY, MODE, TM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS) != hipSuccess)
      return MELGPT_ERR_LAUNCH;
    attr = true;
  }
  constexpr int BM = 32 * TM;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + 255) / 256;
  const long long total = (long long)tiles_m * tiles_n * batch;
  if (total > 0x3FFFFFFF) return MELGPT_ERR_UNSUPPORTED;
  int grid = (int)(total < ncu ? total : ncu);  // LDS footprint: exactly one workgroup per CU
  // XCD blocks need every XCD to own the same number of workgroups; otherwise the linear order is used
  int RN = 0;
  if (grid == ncu && ncu % 8 == 0) {
    // block shape RM x RN with RM * RN = workgroups per XCD: the one that pads the tile grid least (a 3-wide block on a
    // 4-column grid leaves a third of the workgroups of every second block without a tile - 240 workgroups, 30 per XCD,
    // N = 1024: + 18 % on the step), the most square among equals
    const int per = ncu / 8;
    const long long rows_m = (long long)tiles_m * batch;
    long long best = 0;
    for (int c = 1; c * c <= per; ++c) {
      if (per % c != 0 || c > tiles_n) continue;
      const int rm = per / c;
      const long long slots = ((tiles_n + c - 1) / c) * (long long)c * (((rows_m + rm - 1) / rm) * rm);
      if (RN == 0 || slots <= best) {
        best = slots;
        RN = c;
      }
    }
  }
  // the ping-pong K loop where it is built (gemm8p.hip), the ring otherwise (the rolled full epilogue, melgpt_set_gemm_pingpong(0))
  const int st = launch_gemm8p(p, ALAY, BLAY, MODE, TM, tiles_m, tiles_n, batch, RN, grid, s);
  if (st != MELGPT_ERR_UNSUPPORTED) return st;
  melgpt_count_gemm_loop(0);
  hipLaunchKernelGGL((gemm256_kernel<ALAY, BLAY, MODE, TM>), dim3(grid), dim3(512), LDS, s, p, tiles_m, tiles_n, batch, RN);
  return melgpt_launch_status();
}

static int device_cus() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
      ncu = n;
  }
  return ncu;
}

// Tile height: 256 rows, or 192 when that leaves fewer padded tile-rounds (rounds x tile rows) on this many CUs -
// e.g. M = 33920, N = 1024: 532 tiles of 256 x 256 are 3 rounds on 256 CUs (cost 3 x 8), 708 of 192 x 256 are 3 shorter
// rounds (3 x 6).
template <int ALAY, int BLAY, int MODE>
int launch_mode(const GemmParams& p, int batch, hipStream_t s) {
  int ncu = device_cus();
  if (ncu <= 0) return MELGPT_ERR_LAUNCH;
  if (ncu - melgpt_get_reserved_cus() >= 8) ncu -= melgpt_get_reserved_cus();  // CUs left to concurrent RCCL kernels
  if constexpr (ALAY != LAY_KMAJ) {
    // cost of a launch = tile rounds x cycles of one tile.  A 192-row tile is NOT 3/4 of a 256-row one: its K unit takes
    // 2.9 k cycles against 3.45 k (the B half-unit is staged for fewer rows) and its epilogue 6.3 k against 8 k, + ~2 k of
    // drain and tile switch either way (s_memtime stamps, profiles/r03_gemm_lab.md).  These are the RING loop's cycles; the
    // ping-pong loop (gemm8p.hip: 2.25-2.45 k per K tile, the default since round 4) keeps the same RATIO between the two
    // heights - a re-fit to its own stamps chose the same heights for the class-GPT shapes and worse ones at the GPT-VAE XL
    // widths (profiles/r04_gemm_lab.md) - and its choices were re-checked launch by launch in round 5 (tools/lab/tail_ab.py,
    // profiles/r05_n_tail_ab.jsonl: N = 1024 -> 192 rows, N = 3072 / 4096 -> 256 rows + half-height tail, as picked).  Counting rounds x rows - the first
    // model - put the qkv projection (N = 3072: 9 rounds of 192 rows against 7 of 256) on the wrong side: 5.10 -> 4.72 ms
    // per step with 256 rows.
    const long long nu = (p.K + KU - 1) / KU;
    auto cost = [&](int bm, int tm) {
      const long long tiles = (long long)((p.M + bm - 1) / bm) * ((p.N + 255) / 256) * batch;
      return ((tiles + ncu - 1) / ncu) * (tm == 6 ? nu * 290 + 830 : nu * 345 + 1000);
    };
    // the full-epilogue kernel with a K-major B operand spills 8 VGPRs at 256 rows, and its reloads sit in the K loop
    // where their vmcnt(0) drains the ring on every unit (GELU' dgrad: 11.1 -> 15.3 ms per step): ties go to 192 rows
    const bool tie6 = MODE == EPI_FULL16 && BLAY == LAY_KMAJ;
    const long long c6 = cost(192, 6), c8 = cost(256, 8);
    if (c6 < c8 || (tie6 && c6 == c8)) return launch_tm<ALAY, BLAY, MODE, 6>(p, batch, ncu, s);
  }
  return launch_tm<ALAY, BLAY, MODE, 8>(p, batch, ncu, s);
}

template <int ALAY, int BLAY>
int launch_lay(const GemmParams& p, int batch, hipStream_t s) {
  if (!p.vec_io) return MELGPT_ERR_UNSUPPORTED;
  // MELGPT_ACT_MUL (v *= R: the saved GELU derivative) is a plain mode: same loads and stores as a residual add
  const bool plain = (p.act == MELGPT_ACT_NONE || p.act == MELGPT_ACT_MUL) && p.drop_scale == 0.f && !p.C2;
  if (p.a_rowsum) {  // row sums of A ride on the weight-gradient instantiation only (K-major x K-major, f32 out, no loads)
    if constexpr (!(ALAY == LAY_KMAJ && BLAY == LAY_KMAJ)) return MELGPT_ERR_UNSUPPORTED;
    if (!p.out_f32 || !plain || p.R || p.accumulate || p.act != MELGPT_ACT_NONE) return MELGPT_ERR_UNSUPPORTED;
  }
  if constexpr (ALAY == LAY_CONV) {  // convolutions: bias + residual, bf16 out - the only form the VQ-VAE uses
    return (plain && !p.out_f32) ? launch_mode<ALAY, BLAY, EPI_PLAIN16>(p, batch, s) : MELGPT_ERR_UNSUPPORTED;
  } else {
    const bool loads = p.R || p.accumulate;
    if (p.out_f32) {
      if (!plain) return MELGPT_ERR_UNSUPPORTED;
      return loads ? launch_mode<ALAY, BLAY, EPI_PLAIN32>(p, batch, s) : launch_mode<ALAY, BLAY, EPI_PLAIN32N>(p, batch, s);
    }
    if (p.act == MELGPT_ACT_GELU_DACT) {  // forward of Linear -> GELU: its own lean mode, or the generic 128-tile kernel
      if constexpr (ALAY == LAY_ROW && BLAY == LAY_ROW) {
        if (p.drop_scale == 0.f && !loads && p.C2) return launch_mode<ALAY, BLAY, EPI_DACT16>(p, batch, s);
      }
      return MELGPT_ERR_UNSUPPORTED;
    }
    if (p.act == MELGPT_ACT_MUL && !plain) return MELGPT_ERR_UNSUPPORTED;
    if constexpr (ALAY == LAY_ROW && BLAY == LAY_ROW) {  // Linear -> dropout -> + residual: its own lean mode
      if (p.act == MELGPT_ACT_NONE && p.drop_scale != 0.f && p.R && !p.C2 && !p.accumulate && (p.N & 3) == 0)
        return launch_mode<ALAY, BLAY, EPI_DROPR16>(p, batch, s);
    }
    if (!plain) return launch_mode<ALAY, BLAY, EPI_FULL16>(p, batch, s);
    return loads ? launch_mode<ALAY, BLAY, EPI_PLAIN16>(p, batch, s) : launch_mode<ALAY, BLAY, EPI_PLAIN16N>(p, batch, s);
  }
}

}  // namespace

int gemmk::launch_gemm256(const GemmParams& p, int alay, int blay, int batch, int tile_cfg, hipStream_t s) {
  if (tile_cfg != 3) return MELGPT_ERR_UNSUPPORTED;
  if (alay == LAY_ROW && blay == LAY_ROW) return launch_lay<LAY_ROW, LAY_ROW>(p, batch, s);
  if (alay == LAY_ROW && blay == LAY_KMAJ) return launch_lay<LAY_ROW, LAY_KMAJ>(p, batch, s);
  if (alay == LAY_KMAJ && blay == LAY_KMAJ) return launch_lay<LAY_KMAJ, LAY_KMAJ>(p, batch, s);
  if (alay == LAY_CONV && blay == LAY_ROW) return launch_lay<LAY_CONV, LAY_ROW>(p, batch, s);
  return MELGPT_ERR_UNSUPPORTED;
}
