// Layout probe for v_mfma_f32_32x32x16_bf16 on gfx950: which (row, column) of D an accumulator register of a lane holds,
// and which k-slices a lane's operand registers carry.  Prints the maps derived from three products of small integers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned short bf(float f) { return (unsigned short)(__float_as_uint(f) >> 16); }
__global__ void k(float* out) {
  const int l = threadIdx.x, idx = l & 31, kg = l >> 5;
  for (int test = 0; test < 3; ++test) {
    s16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      const int kk = 8 * kg + j;  // hypothesis: lane (idx, kg) register j <-> k = 8 kg + j
      float av, bv;
      if (test == 0) { av = kk == 0 ? (float)idx : 0.f; bv = kk == 0 ? 1.f : 0.f; }        // D[m][n] = m
      else if (test == 1) { av = kk == (idx & 15) ? 1.f : 0.f; bv = (float)idx; }           // D[m][n] = n
      else { av = (float)kk; bv = kk == (idx & 15) ? 1.f : 0.f; }                           // D[m][n] = n % 16 iff k maps agree
      a[j] = (short)bf(av); b[j] = (short)bf(bv);
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 16; ++j) out[(test * 64 + l) * 16 + j] = c[j];
  }
}
int main() {
  float* d; hipMalloc(&d, 3 * 64 * 16 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[3 * 64 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  bool ok = true;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 16; ++j) {
      const int m = 8 * (j / 4) + 4 * (l >> 5) + (j % 4), n = l & 31;
      ok = ok && h[(0 * 64 + l) * 16 + j] == (float)m && h[(1 * 64 + l) * 16 + j] == (float)n && h[(2 * 64 + l) * 16 + j] == (float)(n % 16);
    }
  printf("hypothesis D[m = 8(j/4) + 4(lane>>5) + j%%4][n = lane&31] = acc[j], operand k = 8(lane>>5) + reg: %s\n", ok ? "CONFIRMED" : "WRONG");
  if (!ok) for (int l = 0; l < 64; l += 31) { for (int j = 0; j < 16; ++j) printf("%g/%g/%g ", h[l * 16 + j], h[(64 + l) * 16 + j], h[(128 + l) * 16 + j]); printf("\n"); }
  return ok ? 0 : 1;
}
