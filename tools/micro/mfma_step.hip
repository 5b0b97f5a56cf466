// What one "step" of conv3x3y costs with nothing but its MFMAs: 144 v_mfma_f32_32x32x16_f16 on 16 accumulators (4 groups of 4, three dependent
// rounds per group of 12), then s_barrier; 4 waves per workgroup (one per SIMD, 512 registers), one workgroup per CU, `steps` iterations.
// Variants: with / without the barrier, with an LDS read + wait in front of each group of 12 (the patch-fragment dependency).
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_step.hip -o tools/micro/mfma_step && tools/micro/mfma_step
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256, 1) step_kernel(float* out, long long* cyc, int steps) {
  extern __shared__ float lds[];
  f16x8 a[4], b;
  for (int i = 0; i < 8; ++i) { b[i] = (_Float16)(0.5f + i * 0.01f); for (int j = 0; j < 4; ++j) a[j][i] = (_Float16)(threadIdx.x * 0.001f + i + j); }
  f32x16 acc[4][4];
  for (int c = 0; c < 4; ++c) for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[c][n][i] = 0.f;
  lds[threadIdx.x] = 1.0f;
  __syncthreads();
  const long long t0 = clock64();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      if (MODE & 2) {                                   // a dependent LDS read in front of the unit
        const float v = lds[(threadIdx.x + u * 64 + s) & 1023];
        a[0][0] = (_Float16)v;
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[u & 3][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[n], b, acc[u & 3][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE & 1) __syncthreads();
  }
  const long long t1 = clock64();
  float sum = 0.f;
  for (int c = 0; c < 4; ++c) for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) sum += acc[c][n][i];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, size_t smem) {
  float* out; long long* cyc;
  const int grid = 256, steps = 64;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
  hipFuncSetAttribute((const void*)step_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((step_kernel<MODE>), dim3(grid), dim3(256), smem, 0, out, cyc, steps);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-50s %.0f cycles per step of 144 MFMAs (4608 = back to back)\n", name, (double)h / steps);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("MFMAs only", 4096);
  run<1>("MFMAs + s_barrier per step", 4096);
  run<1>("MFMAs + s_barrier, 160 KB of LDS allocated", 160 * 1024);
  run<3>("LDS read in front of every unit + s_barrier", 4096);
  return 0;
}
