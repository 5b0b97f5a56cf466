// conv3x3y's unit as a microbenchmark: per unit of 12 MFMAs the wave (alone on its SIMD, four per CU) requests NA ds_read_b128 patch fragments for the unit
// DIST units ahead and NB buffer_load_dwordx4 weight fragments (L2-resident table) for the unit 5 ahead, then multiplies with what it requested earlier.
// Prints cycles per unit (384 = matrix pipe bound).  Build + run: hipcc --offload-arch=gfx950 -O3 tools/micro/unit_pipe.hip -o tools/micro/unit_pipe && tools/micro/unit_pipe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4* lds_u4;

template <int NA, int DIST, int NB>
__global__ void __launch_bounds__(256, 1) k(const u32x4* __restrict__ wtab, float* out, long long* cyc, int steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = 0x3c003c00u;
  __syncthreads();
  f32x16 acc[4][4];
  for (int c = 0; c < 4; ++c) for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[c][n][i] = 0.f;
  const unsigned lbase = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
  u32x4 fa[DIST + 1][8], qb[6][2];
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(wtab), 0, 0x7fffffff, 0x00020000);
  const int lane16 = (threadIdx.x & 63) * 16;
#pragma unroll
  for (int d = 0; d < DIST; ++d)
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[d][i] = *(lds_u4)(size_t)(lbase + (d * 8 + i) * 1024);
#pragma unroll
  for (int d = 0; d < 5; ++d)
#pragma unroll
    for (int i = 0; i < 2; ++i) qb[d][i] = __builtin_amdgcn_raw_buffer_load_b128(srd, lane16, (d * 2 + i) * 1024, 0);
  const long long t0 = clock64();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
#pragma unroll
      for (int i = 0; i < NA; ++i) fa[(u + DIST) % (DIST + 1)][i] = *(lds_u4)(size_t)(lbase + (((u + DIST) % 12) * 8 + i) * 1024 + (s & 1) * 64);
#pragma unroll
      for (int i = 0; i < NB; ++i) qb[(u + 5) % 6][i] = __builtin_amdgcn_raw_buffer_load_b128(srd, lane16, ((((s * 12 + u + 5) * 2 + i) & 1023) * 1024), 0);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[u & 3][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[u % (DIST + 1)][(n * 2 + (r & 1)) % (NA ? NA : 1)]),
                                                                 __builtin_bit_cast(f16x8, qb[u % 6][NB ? (r >> 1) % NB : 0]), acc[u & 3][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float sum = 0.f;
  for (int c = 0; c < 4; ++c) for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) sum += acc[c][n][i];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NA, int DIST, int NB> void run(const u32x4* wtab) {
  float* out; long long* cyc;
  const int grid = 256, steps = 64;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
  hipFuncSetAttribute((const void*)k<NA, DIST, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<NA, DIST, NB>), dim3(grid), dim3(256), 131072, 0, wtab, out, cyc, steps);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%d ds_read_b128 (%d unit%s ahead) + %d buffer_load_dwordx4 per unit of 12 MFMAs: %.0f cycles per unit (384 = matrix pipe bound)\n", NA, DIST, DIST > 1 ? "s" : "", NB,
         (double)h / (steps * 12.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  u32x4* wtab; hipMalloc(&wtab, 1024 * 1024 + 4096); hipMemset(wtab, 0x3c, 1024 * 1024 + 4096);
  run<0, 1, 0>(wtab); run<4, 1, 0>(wtab); run<8, 1, 0>(wtab); run<8, 2, 0>(wtab); run<0, 1, 2>(wtab); run<0, 1, 4>(wtab); run<8, 1, 2>(wtab); run<8, 2, 2>(wtab); run<4, 1, 4>(wtab);
  return 0;
}
