// probe of ds_read_b64_tr_b16 semantics on gfx950: LDS[e] = e; every lane passes the byte address of "its" 8-byte row piece
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(unsigned short* out, int S) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  for (int e = threadIdx.x; e < 8192; e += 64) lds[e] = (unsigned short)e;
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  // hypothesis: the 16 lanes of a group supply a [4 rows][16 cols] block: lane i -> row (i>>2), cols (i&3)*4..+3 ; group g -> cols 16g..
  unsigned a = ((i >> 2) * S + g * 16 + (i & 3) * 4) * 2;
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int S : {64, 40, 72}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 8192 * 2, 0, d, S);
    std::vector<unsigned short> h(256);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int exp = j * S + (l >> 4) * 16 + (l & 15); if (h[l * 4 + j] != exp) ok = 0; }
    printf("S=%d hypothesis(lane i of group g receives rows 0..3 of column 16g+i): %s\n", S, ok ? "CONFIRMED" : "WRONG");
    if (!ok) { for (int l = 0; l < 20; ++l) printf(" lane %d: %d %d %d %d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
  }
  return 0;
}
