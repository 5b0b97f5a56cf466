// Issue rate of back-to-back v_mfma_f32_32x32x16_f16 on gfx950: one wave per SIMD (256 threads / CU, every CU), NACC independent accumulators,
// REP x NACC MFMAs in a row; prints shader cycles (s_memtime) per MFMA and the clock derived from the 100 MHz wall clock.
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int DEP>
__global__ void __launch_bounds__(256, 1) rate_kernel(float* out, long long* cyc, int rep) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i * 0.01f); }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  __syncthreads();
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int r = 0; r < rep; ++r) {
#pragma unroll
    for (int d = 0; d < DEP; ++d)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int NACC, int DEP> void run(const char* name, int grid) {
  float* out; long long* cyc;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 16);
  const int rep = 20000 / (NACC * DEP);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((rate_kernel<NACC, DEP>), dim3(grid), dim3(256), 0, 0, out, cyc, rep);
  hipDeviceSynchronize();
  long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  const double n = (double)rep * NACC * DEP;
  printf("%-40s grid %3d: %.2f s_memtime ticks / MFMA, %.2f ns / MFMA (wall 100 MHz), ticks per ns %.3f\n", name, grid, h[0] / n, h[1] * 10.0 / n, (double)h[0] / (h[1] * 10.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int grid : {1, 256}) {
    run<16, 1>("16 independent accumulators", grid);
    run<4, 3>("4 accumulators x 3 dependent rounds", grid);
    run<4, 1>("4 independent, back to back", grid);
    run<1, 1>("1 accumulator (fully dependent)", grid);
  }
  return 0;
}
