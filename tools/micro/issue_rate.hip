// What other instructions cost a wave that is the only one on its SIMD while it streams MFMAs (conv3x3y's situation): per v_mfma_f32_32x32x16_f16
// (32 cycles of matrix pipe) NV independent VALU, NS SALU, NL ds_read_b128 (+ one wait per unit of 12) are issued; prints cycles per MFMA.
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/micro/issue_rate.hip -o tools/micro/issue_rate && tools/micro/issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NS, int NL>
__global__ void __launch_bounds__(256, 1) k(float* out, long long* cyc, int steps) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4096];
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i * 0.01f); }
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
  __syncthreads();
  float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  unsigned sc = blockIdx.x;
  u32x4 lv = {0, 0, 0, 0};
  const unsigned laddr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
  const long long t0 = clock64();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int m = 0; m < 48; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
#pragma unroll
      for (int i = 0; i < NS; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc) :: "scc");
#pragma unroll
      for (int i = 0; i < NL; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(lv) : "v"(laddr));
      if (NL && (m % 12) == 11) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  const long long t1 = clock64();
  float sum = 0.f;
  for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) sum += acc[n][i];
  for (int i = 0; i < 8; ++i) sum += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = sum + sc + lv[0];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int NS, int NL> void run() {
  float* out; long long* cyc;
  const int grid = 256, steps = 100;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<NV, NS, NL>), dim3(grid), dim3(256), 0, 0, out, cyc, steps);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("per MFMA: %d VALU + %d SALU + %d ds_read_b128  ->  %.1f cycles per MFMA (32 = matrix pipe bound)\n", NV, NS, NL, (double)h / (steps * 48.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0, 0>(); run<1, 0, 0>(); run<2, 0, 0>(); run<3, 0, 0>(); run<4, 0, 0>(); run<6, 0, 0>(); run<8, 0, 0>();
  run<0, 1, 0>(); run<0, 2, 0>(); run<0, 4, 0>(); run<0, 8, 0>();
  run<0, 0, 1>(); run<0, 0, 2>();
  run<2, 2, 1>(); run<2, 1, 1>(); run<4, 2, 1>();
  return 0;
}
