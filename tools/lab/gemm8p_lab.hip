// LAB (not part of the library): the guide's "256^2 8-phase" GEMM structure rebuilt from its description
// (cdna_hip_programming.md, "The 256^2 8-phase template"), to be measured in ONE process against the shipped
// five-slot-ring kernel (tools/lab/gemm8p_ab.py).  C[M,N] = A[M,K] * B[N,K]^T, bf16 in, f32 accumulate, bf16 out;
// M, N multiples of 256, K a multiple of 128.  One 512-thread workgroup per 256 x 256 tile (not persistent).
//
// Structure: K tiles of 64; a K tile is four 16 KiB HALF-TILES (A rows 0-127 / 128-255, B rows 0-127 / 128-255 of the
// output tile), double-buffered: 128 KiB of LDS.  8 waves = 2 groups (wr) x 4 (wc); a wave's 128 x 64 output block is
// 64 rows in each A half x 32 columns in each B half, so its four 64 x 32 QUADRANTS are (A half, B half) pairs and a
// phase multiplies ONE quadrant over the K tile (16 MFMAs) from a register subtile (A: 8 ds_read_b128, B: 4).
// Per K tile, phases 1-4: (A0,B0) reads B0 + A0 | (A0,B1) reads B1 | (A1,B1) reads A1 | (A1,B0) reads nothing.
// Every phase: { fragment reads, ONE half-tile of LDS-DMA (2 pieces per wave) } barrier { 16 MFMAs } barrier; group 1
// runs one barrier behind group 0, so on every SIMD one wave multiplies while the other reads and issues requests.
// Request stream per K tile tau: B0, A0, B1, A1 issued in phases 2, 3, 4 of K tile tau - 2 and phase 1 of tau - 1; one
// counted wait per K tile (phase 4: vmcnt(6) = the three youngest half-tiles stay in flight).
#include "common.h"

namespace {

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
constexpr unsigned OOB = 0xFFFFFFF0u;
__device__ __forceinline__ int row_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int qd = nwg >> 3, rm = nwg & 7, xcd = b & 7;
  return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (b >> 3);
}

#define BARRIER() asm volatile("s_barrier" ::: "memory")
#ifndef P8_PRIO
#define P8_PRIO 1
#endif

constexpr int HT = 16384;                                   // one half-tile
constexpr int S_A0 = 0, S_A1 = HT, S_B0 = 2 * HT, S_B1 = 3 * HT;  // slots inside a 64 KiB buffer

__global__ __launch_bounds__(512) void gemm8p_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                     bf16_t* __restrict__ C, int M, int N, int K, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A0 A1 B0 B1][16 KiB]
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int nk = K >> 6;

  // tile of this workgroup: every XCD takes a contiguous run of the order; the order walks 4 x 8 blocks of tiles
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  if ((tiles_m & 3) == 0 && (tiles_n & 7) == 0) {
    const int blk = L >> 5, in = L & 31, nbn = tiles_n >> 3;
    const int bm = blk / nbn, bn = blk - bm * nbn;
    tm = bm * 4 + (in >> 3);
    tn = bn * 8 + (in & 7);
  } else {
    tm = L / tiles_n;
    tn = L - tm * tiles_n;
  }
  const int m0 = tm * 256, n0 = tn * 256;

  const rsrc_t ra = make_rsrc4(A, (unsigned)((long long)M * K * 2));
  const rsrc_t rb = make_rsrc4(B, (unsigned)((long long)N * K * 2));
  // piece p = w + 8 j of a half-tile: local rows 8 p + (lane >> 3); the chunk a lane fetches undoes row_off's swizzle
  const int r_c = (lane & 7) ^ ((4 * w + (lane >> 4)) & 7);
  const unsigned a_src = (unsigned)((m0 + 8 * w + (lane >> 3)) * K * 2 + r_c * 16);
  const unsigned b_src = (unsigned)((n0 + 8 * w + (lane >> 3)) * K * 2 + r_c * 16);
  const unsigned j_step = (unsigned)(64 * K * 2), h_step = (unsigned)(128 * K * 2);
  // half-tile `half` of operand X for K tile kt -> dst (wave-uniform LDS address of the half-tile)
  auto stage = [&](rsrc_t rs, unsigned src, int half, int kt, char* dst) {
    const bool ok = kt < nk;
    const unsigned o = src + half * h_step + (unsigned)kt * 128u;
    dma16(rs, dst + w * 1024, ok ? o : OOB);
    dma16(rs, dst + (w + 8) * 1024, ok ? o + j_step : OOB);
  };

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 fa[4][2], fb0[2][2], fb1[2][2];

  const int i = lane & 15, g = lane >> 4;
  auto ld_a = [&](const char* half) {  // this wave's 64 rows of an A half-tile
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[mt][ks] = *(const u32x4*)(half + row_off(wr * 64 + mt * 16 + i, 4 * ks + g));
  };
  auto ld_b = [&](const char* half, u32x4 (&fb)[2][2]) {  // this wave's 32 rows of a B half-tile
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb[nt][ks] = *(const u32x4*)(half + row_off(wc * 32 + nt * 16 + i, 4 * ks + g));
  };
  auto mul = [&](f32x4 (&q)[4][2], u32x4 (&fb)[2][2]) {
    __builtin_amdgcn_s_setprio(P8_PRIO);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) q[mt][nt] = MELGPT_MFMA_16x16x32(fb[nt][ks], fa[mt][ks], q[mt][nt]);  // rows = n, cols = m
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: K tile 0 complete, the first three half-tiles of K tile 1 in flight
  stage(rb, b_src, 0, 0, smem + S_B0);
  stage(ra, a_src, 0, 0, smem + S_A0);
  stage(rb, b_src, 1, 0, smem + S_B1);
  stage(ra, a_src, 1, 0, smem + S_A1);
  stage(rb, b_src, 0, 1, smem + 65536 + S_B0);
  stage(ra, a_src, 0, 1, smem + 65536 + S_A0);
  stage(rb, b_src, 1, 1, smem + 65536 + S_B1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  BARRIER();
  if (wr == 1) BARRIER();  // group 1 runs one barrier behind group 0 from here on

  auto ktile = [&](auto buf_c, int kt) {
    constexpr int BUF = decltype(buf_c)::value;
    char* cur = smem + BUF * 65536;
    char* oth = smem + (1 - BUF) * 65536;
    // phase 1: (A0, B0)
    ld_b(cur + S_B0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    ld_a(cur + S_A0);
    __builtin_amdgcn_sched_barrier(0);
    stage(ra, a_src, 1, kt + 1, oth + S_A1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the B0 reads are back: its slot is requested again in phase 2
    BARRIER();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mul(acc[0][0], fb0);
    __builtin_amdgcn_sched_barrier(0);
    BARRIER();
    // phase 2: (A0, B1)
    ld_b(cur + S_B1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    stage(rb, b_src, 0, kt + 2, cur + S_B0);
    BARRIER();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mul(acc[0][1], fb1);
    __builtin_amdgcn_sched_barrier(0);
    BARRIER();
    // phase 3: (A1, B1)
    ld_a(cur + S_A1);
    __builtin_amdgcn_sched_barrier(0);
    stage(ra, a_src, 0, kt + 2, cur + S_A0);
    BARRIER();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mul(acc[1][1], fb1);
    __builtin_amdgcn_sched_barrier(0);
    BARRIER();
    // phase 4: (A1, B0); the K tile's one counted wait: everything but the three youngest half-tiles has landed
    stage(rb, b_src, 1, kt + 2, cur + S_B1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    BARRIER();
    __builtin_amdgcn_sched_barrier(0);
    mul(acc[1][0], fb0);
    __builtin_amdgcn_sched_barrier(0);
    BARRIER();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    ktile(std::integral_constant<int, 0>{}, kt);
    ktile(std::integral_constant<int, 1>{}, kt + 1);
  }
  if (wr == 0) BARRIER();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: lane holds 4 consecutive n of row m = .. + (lane & 15)
#pragma unroll
  for (int mq = 0; mq < 2; ++mq)
#pragma unroll
    for (int nq = 0; nq < 2; ++nq)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int m = m0 + mq * 128 + wr * 64 + mt * 16 + i;
          const int n = n0 + nq * 128 + wc * 32 + nt * 16 + 4 * g;
          const f32x4 v = acc[mq][nq][mt][nt];
          *(u32x2*)(C + (long long)m * N + n) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
}

}  // namespace

extern "C" int lab_gemm8p(const void* A, const void* B, void* C, int M, int N, int K, void* stream) {
  if (M % 256 || N % 256 || K % 128 || (long long)M * K * 2 >= 0xFFFFFFF0ll || (long long)N * K * 2 >= 0xFFFFFFF0ll) return -1;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm8p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess) return -2;
    attr = true;
  }
  const int tiles_m = M / 256, tiles_n = N / 256;
  hipLaunchKernelGGL(gemm8p_kernel, dim3(tiles_m * tiles_n), dim3(512), 131072, (hipStream_t)stream, (const bf16_t*)A,
                     (const bf16_t*)B, (bf16_t*)C, M, N, K, tiles_m, tiles_n);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
