// k workgroups of `threads` threads that spin for `us` microseconds (wall clock, 100 MHz): stand-ins for a communication kernel that holds CUs
// while the compute stream runs (tools/occupy_probe.py).  Build: hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/libpdae_occupy.so tools/micro/occupy.hip
#include <hip/hip_runtime.h>
__global__ void occupy_kernel(long long ticks, unsigned int* sink, int lds_bytes) {
  extern __shared__ unsigned int sh[];
  const long long t0 = wall_clock64();
  unsigned int acc = threadIdx.x;
  while (wall_clock64() - t0 < ticks) { acc = acc * 1664525u + 1013904223u; __builtin_amdgcn_s_sleep(8); }
  if (lds_bytes > 0) sh[threadIdx.x] = acc;
  if (acc == 0xdeadbeefu) sink[0] = acc + (lds_bytes > 0 ? sh[0] : 0);
}
extern "C" int pdae_occupy(int blocks, int threads, int us, int lds_bytes, void* sink, void* stream) {
  hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream, (long long)us * 100, (unsigned int*)sink, lds_bytes);
  return (int)hipGetLastError();
}
