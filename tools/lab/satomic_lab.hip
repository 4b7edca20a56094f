// Does gfx950 execute scalar-memory atomics (s_atomic_add with return)?  Every workgroup claims tickets from one
// counter through the scalar path; the tickets must be a permutation of 0 .. n-1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(int* ctr, int* out, int per) {
  for (int i = 0; i < per; ++i) {
    int v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
    if (threadIdx.x == 0) out[blockIdx.x * per + i] = v;
  }
}
int main() {
  const int nb = 2048, per = 4, n = nb * per;
  int *ctr, *out;
  hipMalloc(&ctr, 4); hipMalloc(&out, n * 4); hipMemset(ctr, 0, 4);
  hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, ctr, out, per);
  hipDeviceSynchronize();
  std::vector<int> h(n); int c;
  hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost); hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  bool ok = c == n;
  for (int i = 0; i < n; ++i) ok = ok && h[i] == i;
  printf("scalar atomics: counter %d (expect %d), tickets %s, err %s\n", c, n, ok ? "a permutation of 0..n-1" : "WRONG",
         hipGetErrorString(hipGetLastError()));
  return ok ? 0 : 1;
}
