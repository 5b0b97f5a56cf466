#include <hip/hip_runtime.h>
__device__ __forceinline__ void split2s(float e0, float e1, float sc, unsigned& head, unsigned& resid) {
  unsigned h, l;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(e0), "v"(sc));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(e1), "v"(sc));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(e0), "v"(sc), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(e1), "v"(sc), "v"(h));
  head = h; resid = l;
}
__global__ void k(const float2* in, uint2* out, float sc) {
  float2 e = in[threadIdx.x];
  unsigned h, l; split2s(e.x, e.y, sc, h, l);
  out[threadIdx.x] = make_uint2(h, l);
}
__global__ void kref(const float2* in, uint2* out, float sc) {
  float2 e = in[threadIdx.x];
  float a = e.x * sc, b = e.y * sc;
  _Float16 h0 = (_Float16)a, h1 = (_Float16)b;
  _Float16 l0 = (_Float16)(a - (float)h0), l1 = (_Float16)(b - (float)h1);
  out[threadIdx.x] = make_uint2((unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16),
                                (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16));
}
#include <cstdio>
#include <cstdlib>
#include <cmath>
int main() {
  const int n = 1 << 16; float2* hin = (float2*)malloc(n * 8); 
  srand(1);
  for (int i = 0; i < n; ++i) { 
    float m = ldexpf((float)rand() / RAND_MAX * 2 - 1, rand() % 40 - 30);
    float m2 = ldexpf((float)rand() / RAND_MAX * 2 - 1, rand() % 40 - 30);
    hin[i] = make_float2(m, m2); }
  hin[0] = make_float2(0.f, -0.f); hin[1] = make_float2(65504.f/16, 1e-9f); hin[2] = make_float2(INFINITY, NAN);
  float2* din; uint2 *d1, *d2; hipMalloc(&din, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(din, hin, n * 8, hipMemcpyHostToDevice);
  int bad = 0;
  uint2* o1 = (uint2*)malloc(n * 8); uint2* o2 = (uint2*)malloc(n * 8);
  for (int blk = 0; blk < n / 256; ++blk) { k<<<1, 256>>>(din + blk * 256, d1 + blk * 256, 16.f); kref<<<1, 256>>>(din + blk * 256, d2 + blk * 256, 16.f); }
  hipMemcpy(o1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(o2, d2, n * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) if (o1[i].x != o2[i].x || o1[i].y != o2[i].y) { if (bad < 10) printf("diff %d: in %g %g  mix %08x %08x ref %08x %08x\n", i, hin[i].x, hin[i].y, o1[i].x, o1[i].y, o2[i].x, o2[i].y); ++bad; }
  printf("mismatches: %d of %d\n", bad, n);
  return 0;
}
