// Issue rate of the VALU instructions the operand-split code is made of, one wave per SIMD, 256 independent back-to-back instances per loop trip.
// 1, 2 and 4 waves per SIMD (a single wave only issues a VALU instruction every ~8 cycles).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; prints cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
#define KERNEL(NAME, ASM)                                                                          \
  __global__ void NAME(float* out, long long* cyc) {                                               \
    float a = threadIdx.x * 1.001f, b = 2.5f, c = 0.f, d = 1.f;                                    \
    float2 p = make_float2(a, b), q = make_float2(b, a);                                           \
    unsigned u = 0;                                                                                \
    long long t0 = clock64();                                                                      \
    for (int i = 0; i < 64; ++i) { REP64(asm volatile(ASM : "+v"(c), "+v"(d), "+v"(u), "+v"(p), "+v"(q) : "v"(a), "v"(b));) } \
    long long t1 = clock64();                                                                      \
    out[threadIdx.x + blockIdx.x * blockDim.x] = c + d + u + p.x + q.y;                            \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                       \
  }
KERNEL(k_mul, "v_mul_f32 %0, %5, %6")
KERNEL(k_and, "v_and_b32 %2, %5, %6")
KERNEL(k_cvt_f16, "v_cvt_f16_f32 %0, %5")
KERNEL(k_cvt_f32, "v_cvt_f32_f16 %0, %5")
KERNEL(k_cvt_f32_sdwa, "v_cvt_f32_f16_sdwa %0, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
KERNEL(k_cvt_f16_sdwa, "v_cvt_f16_f32_sdwa %0, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD")
KERNEL(k_cvt_pk, "v_cvt_pk_f16_f32 %2, %5, %6")
KERNEL(k_cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %2, %5, %6")
KERNEL(k_pk_add, "v_pk_add_f32 %3, %4, %4")
KERNEL(k_pk_mul, "v_pk_mul_f32 %3, %4, %4")
KERNEL(k_max3, "v_max3_f32 %0, %5, %6, %6")
KERNEL(k_perm, "v_perm_b32 %2, %5, %6, %6")
KERNEL(k_lshl_or, "v_lshl_or_b32 %2, %5, 16, %6")
KERNEL(k_bfe, "v_bfe_u32 %2, %5, 13, 10")
KERNEL(k_sub, "v_sub_f32 %0, %5, %6")
KERNEL(k_cvt_bf16pk, "v_cvt_pk_bf16_f32 %2, %5, %6")
int main() {
  float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
#define RUN(NAME) { printf("%-16s", #NAME); for (int w = 1; w <= 8; w *= 2) { const int thr = w <= 4 ? 256 * w : 1024, grid = 256 * (w <= 4 ? 1 : w / 4); \
    NAME<<<grid, thr>>>(out, cyc); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); NAME<<<grid, thr>>>(out, cyc); hipEventRecord(e1); hipEventSynchronize(e1); \
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("  %d w/SIMD: %.2f ns", w, ms * 1e6 / (64.0 * 64 * w)); } printf("  per wave-instruction per SIMD (wall clock, whole GPU busy)\n"); }
  RUN(k_mul) RUN(k_and) RUN(k_sub) RUN(k_cvt_f16) RUN(k_cvt_f32) RUN(k_cvt_f32_sdwa) RUN(k_cvt_f16_sdwa) RUN(k_cvt_pk) RUN(k_cvt_pkrtz) RUN(k_pk_add) RUN(k_pk_mul)
  RUN(k_max3) RUN(k_perm) RUN(k_lshl_or) RUN(k_bfe) RUN(k_cvt_bf16pk)
  return 0;
}
