// Does gfx950 execute scalar atomics (s_atomic_add with return)?  One add per wave; checks the final count and that the returned tickets are a permutation.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/micro/satomic tools/micro/satomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned int* c, unsigned int* out) {
  unsigned int v = 1;
  asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(c) : "memory");
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
  const int n = 100000;
  unsigned int *c, *out; hipMalloc(&c, 4); hipMalloc(&out, n * 4); hipMemset(c, 0, 4);
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, c, out);
  hipError_t e = hipDeviceSynchronize();
  unsigned int hc; std::vector<unsigned int> h(n);
  hipMemcpy(&hc, c, 4, hipMemcpyDeviceToHost); hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  bool perm = true; for (int i = 0; i < n; ++i) perm &= h[i] == (unsigned)i;
  printf("sync %s; counter %u (expected %d); tickets are a permutation: %s\n", hipGetErrorString(e), hc, n, perm ? "yes" : "no");
  return 0;
}
