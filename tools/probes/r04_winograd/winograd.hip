// Winograd F(2x2, 3x3) forward convolution for gfx950 -- 3x3 / stride 1 / pad 1, fp32 NHWC tensors, WEIGHT-CONSTANT layers (the frozen trunk and
// eps branch of ShiftUNet in training, every 3x3 convolution of a sampling pass): the transformed weights U = G g G^T are prepared once.
//
// Why: the direct patch kernels (conv3x3r / conv3x3p) sit at the chip's power limit with 0.7 of their issue slots on the matrix pipe
// (only removing work helps).  F(2x2, 3x3) needs 16 instead of 36 products per 2 x 2 outputs: 2.25x fewer MFMAs and
// 2.25x fewer operand-fragment reads for the same convolution (module.py:242,265 / unet.py:62,174 of the reference).
//
//   V[xi] = B^T d B      input transform of every 4 x 4 input patch d (stride 2), xi = (r, c) in 4 x 4: sums / differences of 4 inputs
//   M[xi] = V[xi] U[xi]  16 independent GEMMs [tiles x Cin] x [Cin x Cout]  -- the MFMA work
//   Y     = A^T M A      2 x 2 outputs per tile: sums / differences of 9 products
//
// Mapping.  One persistent 4-wave workgroup per CU (one wave per SIMD, 512 registers per lane); workgroup tile = 16 x 16 output pixels
// (64 Winograd tiles "wt") x 64 output channels; the accumulators of the 16 transform positions are 16 x 64 x 64 floats = the CU's whole
// accumulator budget (256 AGPRs per lane): wave w owns row r = w of the transform domain, acc[c][wt half][channel half] = 16 tiles of
// 32 x 32.  Per 16-channel chunk (ONE MFMA k-step) and wave: 4 xi x 4 accumulators x 3 products (the f16x3 split of conv3x3p.h: two fp16
// planes, a0 b1 + a1 b0 + a0 b0) = 48 MFMAs fed by 16 ds_read_b128 (V fragments) + 16 buffer loads (U fragments in MFMA order from L2).
// Pipeline per chunk: raw fp32 patch (18 x 18 px x 16 ch) global -> registers -> LDS (one copy per workgroup; a transform thread needs a
// 4 x 4 neighbourhood) -> each thread transforms ONE (tile, channel quad): 16 float4 in, 16 float4 out, split into the two fp16 planes
// AFTER the transform (the sums of four fp16-exact values are not fp16-exact) -> V[plane][xi][wt][16 ch] in LDS, double buffered (2 x 64 KB),
// 32-byte rows with the 16-byte halves of odd tile rows swapped (conflict-free ds_read_b128 fragments).  The prefetch pipeline runs across
// tile boundaries; two barriers per chunk (raw patch single-buffered: 160 KB of LDS are exactly V x 2 + raw).
// Epilogue (not deferred: the accumulators ARE the register file): Z[r][j] = sum_c M[r][c] A[c][j] in registers, then the sum over r across
// the four waves through the V buffer that the last chunk has just freed, one output column parity j at a time (4 x 64 x 64 floats = 64 KB);
// 256-byte runs of float4 stores, bias and the power-of-two output scale fused.
//
// Numerics: transforms in fp32; both operands of every product carry 22 mantissa bits; fp32 accumulation.  Measured against fp64 in
// tests/test_winograd_gpu.py (gate 1e-5 of max |y|, the MATH_TOL of the direct kernels).
#include <stdlib.h>

#include "common.h"
#include "conv3x3p.h"
#include "winograd.h"

#define WN_VBUF 65536u                  // bytes per V buffer: [plane 2][xi 16][wt 64][16 ch] fp16
#define WN_PLANE 32768u
#define WN_XI 2048u                     // 64 wt x 32 bytes
#define WN_RAWP 96u                     // bytes per raw pixel: 16 floats + 8 pad (2-pixel strides cover all 64 banks: conflict-free ds_read_b128)
#define WN_RAW0 (2u * WN_VBUF)
#define WN_RAWB (18u * 18u * WN_RAWP)
#define WN_LDS (WN_RAW0 + WN_RAWB)      // 162176 <= 163840
#define WN_OOB 0xFFFFFFF0u
#define WN_RALL 0xFFFFFFEFu

typedef unsigned wn_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wn_u32x2 __attribute__((ext_vector_type(2)));
typedef float wn_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const wn_u32x4* wn_lds_u4;
typedef __attribute__((address_space(3))) wn_u32x2* wn_lds_u2;
typedef __attribute__((address_space(3))) wn_f32x4* wn_lds_f4;
typedef __attribute__((address_space(3))) float* wn_lds_f;

struct WinoParams {
  const float* x; int N, H, W, C;
  const unsigned short* wp; int NT;     // transformed weights [plane][C/16][xi][NT][64][8] fp16, NT = Nout / 32
  int Nout; float* y; const float* bias;
  float woscale;                        // 1 / (power-of-two scale of the prepared weights)
  unsigned int* sat;
  int tiles_x, tiles_y, tiles_n;
};

// SCHED: 0 = two regions per chunk, scheduling left to the compiler; 1 = the same with an issue pattern (an MFMA, then up to NV ALU instructions and LDS /
// vector-memory slots); 2 = twelve hand-placed units of 4 MFMAs per chunk (PDAE_WINO_SCHED, default 1)
#define WN_PATTERN(NM, NV)                                                                                    \
  if constexpr (SCHED >= 1) {                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < (NM); ++i_) {                                                     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                      \
      __builtin_amdgcn_sched_group_barrier(0x006, NV, 0);                                                     \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                      \
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);                                                      \
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                                                      \
      __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);                                                      \
    }                                                                                                         \
  }

// timing probes (tools/probe_build.py, WRONG RESULTS by design): pieces of the kernel compiled out
#ifdef PDAE_WN_PROBE_NOXF
#define WN_PXF(X)
#else
#define WN_PXF(X) X
#endif
#ifdef PDAE_WN_PROBE_NORAW
#define WN_PRAW(X)
#else
#define WN_PRAW(X) X
#endif
#ifdef PDAE_WN_PROBE_NOB
#define WN_PB(X)
#else
#define WN_PB(X) X
#endif
#ifdef PDAE_WN_PROBE_NOA
#define WN_PA(X)
#else
#define WN_PA(X) X
#endif
#ifdef PDAE_WN_PROBE_NOBARA
#define WN_PBARA(X)
#else
#define WN_PBARA(X) X
#endif

template <int SCHED>
__global__ void __launch_bounds__(256, 1) wino_kernel(const WinoParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, h = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)wsm;
  const int nch = P.C >> 4, NT = P.NT;
  const int ntiles = P.N * P.tiles_y * P.tiles_x * P.tiles_n, G = gridDim.x;
  // workgroups of one XCD (blockIdx % 8) take a contiguous range of tile indices: the two channel halves of a pixel tile share their input in one L2
  const int vb = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const float ascale = PASCALE, oscale = P.woscale / ascale;      // exact: powers of two
  float sat_hit = 0.f;

#define WN_DECODE(TILE, IMG, Y0, X0, N0)                                                                      \
  { int tl_ = (TILE); const int tn_ = tl_ % P.tiles_n; tl_ /= P.tiles_n; const int tx_ = tl_ % P.tiles_x; tl_ /= P.tiles_x;   \
    const int ty_ = tl_ % P.tiles_y; tl_ /= P.tiles_y; IMG = tl_; Y0 = ty_ * 16; X0 = tx_ * 16; N0 = tn_ * 64; }

  // ---- raw patch: global -> registers -> LDS.  Slot i of thread t: patch pixel (t >> 2) + 64 i, channel quad t & 3
  const int tq = t & 3;
  const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x), 0, (int)WN_RALL, 0x00020000);
  unsigned ld_off[6];
  int ld_tile = vb, ld_ch = 0;
  float4 rawreg[6] = {};
#define WN_LD_SETUP()                                                                                         \
  {                                                                                                           \
    int img_, y0_, x0_, n0_;                                                                                  \
    const bool live_ = ld_tile < ntiles;                                                                      \
    WN_DECODE(live_ ? ld_tile : 0, img_, y0_, x0_, n0_)                                                       \
    (void)n0_;                                                                                                \
    int t4_ = t >> 2;                                                                                         \
    asm volatile("" : "+v"(t4_));      /* patch coordinates re-derived per tile: six registers less across the chunk loop */ \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                           \
      const int px_ = t4_ + 64 * i, py_ = (px_ * 3641) >> 16, pxx_ = px_ - py_ * 18;      /* px / 18 for px < 384 */ \
      const int ly = y0_ - 1 + py_, lx = x0_ - 1 + pxx_;                                                      \
      const bool ok = live_ & (px_ < 324) & ((unsigned)ly < (unsigned)P.H) & ((unsigned)lx < (unsigned)P.W);  \
      const unsigned off = (unsigned)(((img_ * P.H + ly) * P.W + lx) * P.C + tq * 4) * 4u;                    \
      ld_off[i] = ok ? off : WN_OOB;                                                                          \
    }                                                                                                         \
  }
  auto gload = [&]() {
#pragma unroll
    for (int i = 0; i < 6; ++i)
      rawreg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, (int)ld_off[i], ld_ch * 64, 0));
  };
#define WN_LD_ADVANCE()                                                                                       \
  { ++ld_ch; if (ld_ch == nch) { ld_ch = 0; ld_tile += G; WN_LD_SETUP() } }
  const unsigned rs_base = lds0 + WN_RAW0 + (unsigned)((t >> 2) * WN_RAWP + tq * 16);
  // fp16-window guard (common.h): every transformed value is a signed sum of four inputs, so 4 x max |x| bounds it -- tracked here on the six raw
  // float4 a thread stages per chunk instead of on its sixteen transformed ones (64 ALU instructions per chunk saved; conservative by <= 4x)
  auto raw_store = [&]() {
#pragma unroll
    for (int i = 0; i < 6; ++i) pdae_f16_amax4(rawreg[i], 4.0f * ascale, sat_hit);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const wn_f32x4 v = {rawreg[i].x, rawreg[i].y, rawreg[i].z, rawreg[i].w};
      *(wn_lds_f4)(size_t)(rs_base + (unsigned)(i * 64 * WN_RAWP)) = v;
    }
    if (t < 16) {
      const wn_f32x4 v = {rawreg[5].x, rawreg[5].y, rawreg[5].z, rawreg[5].w};
      *(wn_lds_f4)(size_t)(rs_base + (unsigned)(5 * 64 * WN_RAWP)) = v;
    }
  };

  // ---- input transform: this thread's (tile twt = (twy, twx), channel quad tq)
  const int twt = t >> 2, twy = twt >> 3, twx = twt & 7;
  const unsigned rd_base = lds0 + WN_RAW0 + (unsigned)(((2 * twy) * 18 + 2 * twx) * WN_RAWP + tq * 16);
  const unsigned vw_base = lds0 + (unsigned)(twt * 32 + (((tq >> 1) ^ (twy & 1)) * 16 + (tq & 1) * 8));
  unsigned cur = 0;                      // V buffer holding the chunk being multiplied; the transform writes cur ^ 1
  unsigned vwb = vw_base;                // vw_base + (cur ^ 1) * WN_VBUF, refreshed per chunk, opaque: every V store is base + immediate
  auto raw_row = [&](int a, float4 (&d)[4]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const wn_f32x4 v = *(wn_lds_f4)(size_t)(rd_base + (unsigned)((a * 18 + b) * WN_RAWP));
      d[b] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
#define WN_F4(OP, A_, B_) make_float4(A_.x OP B_.x, A_.y OP B_.y, A_.z OP B_.z, A_.w OP B_.w)
  // a patch row through B along x:  s[0] = d0 - d2, s[1] = d1 + d2, s[2] = d2 - d1, s[3] = d1 - d3
  auto row_x = [&](const float4 (&d)[4], float4 (&s)[4]) {
    s[0] = WN_F4(-, d[0], d[2]); s[1] = WN_F4(+, d[1], d[2]); s[2] = WN_F4(-, d[2], d[1]); s[3] = WN_F4(-, d[1], d[3]);
  };
  // one transformed value quad -> the two fp16 planes of V[xi] in the buffer being written
  auto emit = [&](int xi, const float4& v) {
    unsigned a0, a1, b0, b1;
    pdae_f16_split2s(v.x, v.y, ascale, a0, a1);
    pdae_f16_split2s(v.z, v.w, ascale, b0, b1);
    const unsigned dst = vwb + (unsigned)xi * WN_XI;
    const wn_u32x2 hi = {a0, b0}, lo = {a1, b1};
    *(wn_lds_u2)(size_t)dst = hi;
    *(wn_lds_u2)(size_t)(dst + WN_PLANE) = lo;
  };
  // The transform in pieces that the chunk body places between its MFMAs (one piece per 4-MFMA unit): patch rows 1, 2 -> s1, s2 -> rows r = 1, 2 of
  // the transform domain; patch rows 0, 3 -> s0, s3 -> rows r = 0, 3 (ve), which cross the barrier that releases the raw patch as eight
  // float4 and are split / stored behind it.
  float4 ve[2][4], d1[4], d2[4], s1[4], s2[4], vm[4];
  // the same transform in two pieces for the compiler-scheduled chunk body (SCHED 0 / 1): everything that reads the raw patch, then the rest
  auto xform_head = [&]() {
    float4 d[4], t1[4], t2[4], t0[4];
    raw_row(1, d); row_x(d, t1);
    raw_row(2, d); row_x(d, t2);
#pragma unroll
    for (int c = 0; c < 4; ++c) { emit(4 + c, WN_F4(+, t1[c], t2[c])); emit(8 + c, WN_F4(-, t2[c], t1[c])); }
    raw_row(0, d); row_x(d, t0);
#pragma unroll
    for (int c = 0; c < 4; ++c) ve[0][c] = WN_F4(-, t0[c], t2[c]);
    raw_row(3, d); row_x(d, t0);
#pragma unroll
    for (int c = 0; c < 4; ++c) ve[1][c] = WN_F4(-, t1[c], t0[c]);
  };
  auto xform_tail = [&]() {
#pragma unroll
    for (int c = 0; c < 4; ++c) { emit(c, ve[0][c]); emit(12 + c, ve[1][c]); }
  };
  auto xform_all = [&]() {                 // prologue: the whole transform of one step
    float4 d[4], s0[4];
    raw_row(1, d); row_x(d, s1);
    raw_row(2, d); row_x(d, s2);
#pragma unroll
    for (int c = 0; c < 4; ++c) { emit(4 + c, WN_F4(+, s1[c], s2[c])); emit(8 + c, WN_F4(-, s2[c], s1[c])); }
    raw_row(0, d); row_x(d, s0);
#pragma unroll
    for (int c = 0; c < 4; ++c) emit(c, WN_F4(-, s0[c], s2[c]));
    raw_row(3, d); row_x(d, s0);
#pragma unroll
    for (int c = 0; c < 4; ++c) emit(12 + c, WN_F4(-, s1[c], s0[c]));
  };

  // ---- MFMA operands.  A: V fragments of transform position xi = 4 wv + c, tile half mb: lane (li, h) reads 16 bytes of row mb * 32 + li
  const unsigned a_lane = lds0 + (unsigned)(li * 32 + ((h ^ ((li >> 3) & 1)) * 16)) + (unsigned)wv * 4u * WN_XI;
  unsigned abase = a_lane;
  uint4 fa[2][2][2] = {};                // [ring][mb][plane]
  auto lda = [&](uint4 (&af)[2][2], int c) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        af[mb][p] = __builtin_bit_cast(uint4, *(wn_lds_u4)(size_t)(abase + (unsigned)c * WN_XI + (unsigned)(mb * 1024) + (unsigned)p * WN_PLANE));
  };
  // B: U fragments [plane][chunk][xi][nt][lane][8]: one coalesced 1 KB load per (plane, channel block)
  const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.wp), 0, 0x7fffffff, 0x00020000);
  const unsigned ps_b = (unsigned)nch * 16u * (unsigned)NT * 1024u;
  const int lane16 = lane * 16;
  uint4 qb[4][2][2] = {};                // [c][cb][plane]
  auto ldb = [&](uint4 (&bq)[2][2], int chunk, int c, int nt0) {
    const unsigned soff = (unsigned)((chunk * 16 + wv * 4 + c) * NT + nt0) * 1024u;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        bq[cb][p] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srd_w, lane16, (int)(soff + (unsigned)cb * 1024u + (unsigned)p * ps_b), 0));
  };

  f32x16 acc[4][2][2];                   // [c][mb][cb]
  // the four MFMAs of product pass PASS (0: a0 b1, 1: a1 b0, 2: a0 b0 -- cross terms first, as in conv3x3p) of transform column c
  auto mma4 = [&](const uint4 (&af)[2][2], const uint4 (&bq)[2][2], int c, int pass, bool zc) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int pa = pass == 1 ? 1 : 0, pb = pass == 0 ? 1 : 0;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        acc[c][mb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[mb][pa]), __builtin_bit_cast(f16x8, bq[cb][pb]),
                                                                (zc && pass == 0) ? zero : acc[c][mb][cb], 0, 0, 0);
  };

  // ---- prologue: step 0 transformed into V[0], step 1 raw in LDS, step 2 in flight, U fragments of step 0 in flight
  int tile = vb;
  if (tile >= ntiles) return;
  int c_img, c_y0, c_x0, c_n0;
  WN_DECODE(tile, c_img, c_y0, c_x0, c_n0)
  WN_LD_SETUP()
  gload();
  WN_LD_ADVANCE()
  raw_store();
  gload();
  WN_LD_ADVANCE()
  __syncthreads();
  cur = 1;                               // emit writes buffer cur ^ 1 = 0
  vwb = vw_base;
  xform_all();
  __syncthreads();
  raw_store();
#pragma unroll
  for (int c = 0; c < 4; ++c) ldb(qb[c], 0, c, c_n0 >> 5);
  gload();
  WN_LD_ADVANCE()
  cur = 0;
  __syncthreads();

  // One chunk = twelve units of 4 MFMAs (transform column c = unit / 3, product pass = unit % 3), each fenced by sched_barrier(0) and carrying one
  // piece of the NEXT step's transform: in this one-wave-per-SIMD kernel a matrix instruction only overlaps the vector / LDS instructions that sit
  // right behind it in program order (left to itself the scheduler issued the MFMAs first and the ~280 transform instructions after them: 0.21 MFMA
  // utilisation).  U fragments: ring of four slots, slot c reloaded for the next step right behind its last MFMAs; the raw-patch loads from HBM are
  // the LAST vector-memory instructions of a chunk, so that no U-fragment wait of the next chunk queues behind them (in-order return).
  //   units 0..7 (before the barrier that releases the raw patch): raw rows 1, 2 -> s1, s2 -> V rows 1, 2; raw rows 0, 3 -> ve
  //   units 8..11: V rows 0, 3 from ve; raw patch of step + 2 to LDS; HBM loads of step + 3
#define WN_UNIT_END(NV) WN_PATTERN(4, NV) __builtin_amdgcn_sched_barrier(0);
#define WN_ITER(FIRST, NCH, NNT)                                                                              \
  {                                                                                                           \
    abase = a_lane + cur * WN_VBUF;                                                                           \
    vwb = vw_base + (cur ^ 1u) * WN_VBUF;                                                                     \
    asm volatile("" : "+v"(abase), "+v"(vwb));                                                                \
    WN_PA(lda(fa[0], 0);)                                                                                     \
    WN_PXF(raw_row(1, d1); raw_row(2, d2);)                                                                   \
    WN_PA(lda(fa[1], 1);)                                                                                     \
    /* c = 0 */                                                                                               \
    mma4(fa[0], qb[0], 0, 0, FIRST); WN_UNIT_END(8)                                                           \
    mma4(fa[0], qb[0], 0, 1, FIRST); WN_PXF(row_x(d1, s1); raw_row(0, d1);) WN_UNIT_END(8)                    \
    mma4(fa[0], qb[0], 0, 2, FIRST); WN_PXF(row_x(d2, s2); raw_row(3, d2);) WN_PB(ldb(qb[0], NCH, 0, NNT);) WN_PA(lda(fa[0], 2);) WN_UNIT_END(8) \
    /* c = 1 */                                                                                               \
    mma4(fa[1], qb[1], 1, 0, FIRST);                                                                          \
    WN_PXF(_Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) vm[c_] = WN_F4(+, s1[c_], s2[c_]); emit(4, vm[0]); emit(5, vm[1]);) WN_UNIT_END(8) \
    mma4(fa[1], qb[1], 1, 1, FIRST);                                                                          \
    WN_PXF(emit(6, vm[2]); emit(7, vm[3]); _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) vm[c_] = WN_F4(-, s2[c_], s1[c_]);) WN_UNIT_END(8) \
    mma4(fa[1], qb[1], 1, 2, FIRST);                                                                          \
    WN_PXF(_Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) emit(8 + c_, vm[c_]);) WN_PB(ldb(qb[1], NCH, 1, NNT);) WN_PA(lda(fa[1], 3);) WN_UNIT_END(8) \
    /* c = 2 */                                                                                               \
    mma4(fa[0], qb[2], 2, 0, FIRST);                                                                          \
    WN_PXF({ float4 s0_[4]; row_x(d1, s0_); _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) ve[0][c_] = WN_F4(-, s0_[c_], s2[c_]); }) WN_UNIT_END(8) \
    mma4(fa[0], qb[2], 2, 1, FIRST);                                                                          \
    WN_PXF({ float4 s3_[4]; row_x(d2, s3_); _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) ve[1][c_] = WN_F4(-, s1[c_], s3_[c_]); }) WN_UNIT_END(8) \
    WN_PBARA(__syncthreads();) /* every raw read of this step is done: the raw patch may be overwritten */    \
    mma4(fa[0], qb[2], 2, 2, FIRST);                                                                          \
    WN_PXF(_Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) emit(c_, ve[0][c_]);) WN_PB(ldb(qb[2], NCH, 2, NNT);) WN_UNIT_END(8) \
    /* c = 3 */                                                                                               \
    mma4(fa[1], qb[3], 3, 0, FIRST);                                                                          \
    WN_PXF(_Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) emit(12 + c_, ve[1][c_]);) WN_UNIT_END(8)         \
    mma4(fa[1], qb[3], 3, 1, FIRST);                                                                          \
    WN_PRAW(raw_store();) WN_UNIT_END(8)                                                                      \
    mma4(fa[1], qb[3], 3, 2, FIRST);                                                                          \
    WN_PB(ldb(qb[3], NCH, 3, NNT);)                                                                           \
    WN_PRAW(gload();)                                                                                         \
    WN_LD_ADVANCE()                                                                                           \
    WN_UNIT_END(8)                                                                                            \
    __syncthreads();          /* V[cur ^ 1] and the raw patch of the step after it are complete */            \
    cur ^= 1u;                                                                                                \
  }

  // The compiler-scheduled form of a chunk (SCHED 0: no hints, 1: an issue pattern per region): two regions of 24 MFMAs around the barrier that
  // releases the raw patch.  Measured FASTER than the twelve hand-placed units above (0.43 vs 0.50 ms on 128x128 128->128, B = 32): kept as the default.
#define WN_MMA12(AF, BQ, C, FIRST) mma4(AF, BQ, C, 0, FIRST); mma4(AF, BQ, C, 1, FIRST); mma4(AF, BQ, C, 2, FIRST);
#define WN_ITER_C(FIRST, NCH, NNT)                                                                            \
  {                                                                                                           \
    abase = a_lane + cur * WN_VBUF;                                                                           \
    vwb = vw_base + (cur ^ 1u) * WN_VBUF;                                                                     \
    asm volatile("" : "+v"(abase), "+v"(vwb));                                                                \
    WN_PA(lda(fa[0], 0); lda(fa[1], 1);)                                                                      \
    WN_MMA12(fa[0], qb[0], 0, FIRST)                                                                          \
    WN_PB(ldb(qb[0], NCH, 0, NNT);)                                                                           \
    WN_PXF(xform_head();)                                                                                     \
    WN_PA(lda(fa[0], 2);)                                                                                     \
    WN_MMA12(fa[1], qb[1], 1, FIRST)                                                                          \
    WN_PB(ldb(qb[1], NCH, 1, NNT);)                                                                           \
    WN_PATTERN(24, 8)                                                                                         \
    WN_PBARA(__syncthreads();)                                                                                \
    WN_PA(lda(fa[1], 3);)                                                                                     \
    WN_MMA12(fa[0], qb[2], 2, FIRST)                                                                          \
    WN_PB(ldb(qb[2], NCH, 2, NNT);)                                                                           \
    WN_PXF(xform_tail();)                                                                                     \
    WN_PRAW(raw_store();)                                                                                     \
    WN_MMA12(fa[1], qb[3], 3, FIRST)                                                                          \
    WN_PB(ldb(qb[3], NCH, 3, NNT);)                                                                           \
    WN_PRAW(gload();)                                                                                         \
    WN_LD_ADVANCE()                                                                                           \
    WN_PATTERN(24, 6)                                                                                         \
    __syncthreads();                                                                                          \
    cur ^= 1u;                                                                                                \
  }
#define WN_ITER_ANY(FIRST, NCH, NNT) { if constexpr (SCHED == 2) WN_ITER(FIRST, NCH, NNT) else WN_ITER_C(FIRST, NCH, NNT) }

  for (; tile < ntiles; tile += G) {
    int x_img, x_y0, x_x0, x_n0;
    const int ntile = tile + G < ntiles ? tile + G : tile;
    WN_DECODE(ntile, x_img, x_y0, x_x0, x_n0)
    const int c_nt = c_n0 >> 5, x_nt = x_n0 >> 5;
    WN_ITER_ANY(true, nch > 1 ? 1 : 0, nch > 1 ? c_nt : x_nt)
    for (int ch = 1; ch < nch; ++ch) {
      const bool lastc = ch + 1 == nch;
      WN_ITER_ANY(false, lastc ? 0 : ch + 1, lastc ? x_nt : c_nt)
    }
#ifndef PDAE_WN_PROBE_NOEPI
    // ---- output transform.  Along c in registers: Z[.][0] = M0 + M1 + M2, Z[.][1] = M1 - M2 - M3; along r (= wave) through LDS
    // (addresses formed here, per tile, from t: hoisted out of the tile loop they were spilled to scratch around the chunk loop)
    int tt = t;
    asm volatile("" : "+v"(tt));
    const unsigned ex_w = (unsigned)(wv * 16384 + (4 * (tt >> 5 & 1) * 64 + (tt & 31)) * 4);
    const unsigned ex_r = (unsigned)((tt >> 4) * 256 + (tt & 15) * 16);
    const int e_wy = tt >> 7, e_wx = (tt >> 4) & 7, e_c4 = (tt & 15) * 4;
    const unsigned exb = lds0 + (cur ^ 1u) * WN_VBUF;          // the buffer the last chunk has just freed (cur already toggled)
    const float4 bias4 = P.bias ? *reinterpret_cast<const float4*>(P.bias + c_n0 + e_c4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float z = j == 0 ? (acc[0][mb][cb][r] + acc[1][mb][cb][r]) + acc[2][mb][cb][r]
                                   : (acc[1][mb][cb][r] - acc[2][mb][cb][r]) - acc[3][mb][cb][r];
            *(wn_lds_f)(size_t)(exb + ex_w + (unsigned)(((mb * 32 + (r & 3) + 8 * (r >> 2)) * 64 + cb * 32) * 4)) = z;
          }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        wn_f32x4 z[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = *(wn_lds_f4)(size_t)(exb + ex_r + (unsigned)(k * 4096 + r * 16384));
        const int oy = c_y0 + 2 * (e_wy + 2 * k), ox = c_x0 + 2 * e_wx + j;
        float* dst = P.y + ((size_t)(c_img * P.H + oy) * P.W + ox) * P.Nout + c_n0 + e_c4;
        float4 y0v, y1v;
        y0v.x = fmaf((z[0][0] + z[1][0]) + z[2][0], oscale, bias4.x); y0v.y = fmaf((z[0][1] + z[1][1]) + z[2][1], oscale, bias4.y);
        y0v.z = fmaf((z[0][2] + z[1][2]) + z[2][2], oscale, bias4.z); y0v.w = fmaf((z[0][3] + z[1][3]) + z[2][3], oscale, bias4.w);
        y1v.x = fmaf((z[1][0] - z[2][0]) - z[3][0], oscale, bias4.x); y1v.y = fmaf((z[1][1] - z[2][1]) - z[3][1], oscale, bias4.y);
        y1v.z = fmaf((z[1][2] - z[2][2]) - z[3][2], oscale, bias4.z); y1v.w = fmaf((z[1][3] - z[2][3]) - z[3][3], oscale, bias4.w);
#ifdef PDAE_WN_PROBE_NOSTORE
        if (y0v.x == 123.456f && y1v.y == 654.321f)
#endif
        {
          *reinterpret_cast<float4*>(dst) = y0v;
          *reinterpret_cast<float4*>(dst + (size_t)P.W * P.Nout) = y1v;
        }
      }
      __syncthreads();
    }
#else
    {                                                          // keeps every accumulator alive (256 adds per tile)
      float sum_ = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum_ += acc[c][mb][cb][r];
      if (sum_ == 123.456f) P.y[t] = sum_;
    }
#endif
    c_img = x_img; c_y0 = x_y0; c_x0 = x_x0; c_n0 = x_n0;
  }
  pdae_sat_report(P.sat, sat_hit);
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// Eight-wave form (PDAE_WINO_SCHED=8): the same tile, LDS layout, pipeline and arithmetic with TWO waves per SIMD (256 registers each), so
// that the hardware overlaps one wave's transform / memory waits with the other's MFMAs instead of a hand-placed instruction stream (the four-
// wave probes: transform 0.15-0.19 ms and raw-patch HBM latency 0.12-0.17 ms of a 0.43-0.50 ms launch, all in series with the MFMAs).
//   wave wv = (grp = wv >> 2, r = wv & 3):  MFMA role: transform row r, channel half cb = grp: acc[c][mb] = 8 tiles of 32 x 32 (128 AGPRs), per chunk
//   24 MFMAs, A fragments 16 ds_read_b128 (both tile halves), U fragments 8 loads.  Transform role: thread tl = t & 255 owns (tile tl >> 2, quad
//   tl & 3) as before; group 0 produces transform rows 1, 2 (patch rows 1, 2), group 1 rows 0, 3 (all four patch rows).  SIMD k hosts waves
//   k and k + 4: one of each group.
template <int SKEW>
__global__ void __launch_bounds__(512, 2) wino8_kernel(const WinoParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, h = lane >> 5;
  const int grp = wv >> 2, wr = wv & 3;
  const unsigned lds0 = (unsigned)(size_t)wsm;
  const int nch = P.C >> 4, NT = P.NT;
  const int ntiles = P.N * P.tiles_y * P.tiles_x * P.tiles_n, G = gridDim.x;
  const int vb = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const float ascale = PASCALE, oscale = P.woscale / ascale;
  float sat_hit = 0.f;

  // ---- raw patch: slot i of thread t: patch pixel (t >> 2) + 128 i, channel quad t & 3 (i < 3; 1296 slots)
  const int tq = t & 3;
  const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x), 0, (int)WN_RALL, 0x00020000);
  unsigned ld_off[3];
  int ld_tile = vb, ld_ch = 0;
  float4 rawreg[3] = {};
#define W8_LD_SETUP()                                                                                         \
  {                                                                                                           \
    int img_, y0_, x0_, n0_;                                                                                  \
    const bool live_ = ld_tile < ntiles;                                                                      \
    WN_DECODE(live_ ? ld_tile : 0, img_, y0_, x0_, n0_)                                                       \
    (void)n0_;                                                                                                \
    int t4_ = t >> 2;                                                                                         \
    asm volatile("" : "+v"(t4_));                                                                             \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                           \
      const int px_ = t4_ + 128 * i, py_ = (px_ * 3641) >> 16, pxx_ = px_ - py_ * 18;                         \
      const int ly = y0_ - 1 + py_, lx = x0_ - 1 + pxx_;                                                      \
      const bool ok = live_ & (px_ < 324) & ((unsigned)ly < (unsigned)P.H) & ((unsigned)lx < (unsigned)P.W);  \
      const unsigned off = (unsigned)(((img_ * P.H + ly) * P.W + lx) * P.C + tq * 4) * 4u;                    \
      ld_off[i] = ok ? off : WN_OOB;                                                                          \
    }                                                                                                         \
  }
  auto gload = [&]() {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      rawreg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, (int)ld_off[i], ld_ch * 64, 0));
  };
#define W8_LD_ADVANCE()                                                                                       \
  { ++ld_ch; if (ld_ch == nch) { ld_ch = 0; ld_tile += G; W8_LD_SETUP() } }
  const unsigned rs_base = lds0 + WN_RAW0 + (unsigned)((t >> 2) * WN_RAWP + tq * 16);
  auto raw_store = [&]() {
#pragma unroll
    for (int i = 0; i < 3; ++i) pdae_f16_amax4(rawreg[i], 4.0f * ascale, sat_hit);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const wn_f32x4 v = {rawreg[i].x, rawreg[i].y, rawreg[i].z, rawreg[i].w};
      *(wn_lds_f4)(size_t)(rs_base + (unsigned)(i * 128 * WN_RAWP)) = v;
    }
    if (t < 272) {
      const wn_f32x4 v = {rawreg[2].x, rawreg[2].y, rawreg[2].z, rawreg[2].w};
      *(wn_lds_f4)(size_t)(rs_base + (unsigned)(2 * 128 * WN_RAWP)) = v;
    }
  };

  // ---- input transform
  const int tl = t & 255, twt = tl >> 2, twy = twt >> 3, twx = twt & 7;
  const unsigned rd_base = lds0 + WN_RAW0 + (unsigned)(((2 * twy) * 18 + 2 * twx) * WN_RAWP + tq * 16);
  const unsigned vw_base = lds0 + (unsigned)(twt * 32 + (((tq >> 1) ^ (twy & 1)) * 16 + (tq & 1) * 8));
  unsigned cur = 0, vwb = vw_base;
  auto raw_row = [&](int a, float4 (&d)[4]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const wn_f32x4 v = *(wn_lds_f4)(size_t)(rd_base + (unsigned)((a * 18 + b) * WN_RAWP));
      d[b] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto row_x = [&](const float4 (&d)[4], float4 (&sx)[4]) {
    sx[0] = WN_F4(-, d[0], d[2]); sx[1] = WN_F4(+, d[1], d[2]); sx[2] = WN_F4(-, d[2], d[1]); sx[3] = WN_F4(-, d[1], d[3]);
  };
  auto emit = [&](int xi, const float4& v) {
    unsigned a0, a1, b0, b1;
    pdae_f16_split2s(v.x, v.y, ascale, a0, a1);
    pdae_f16_split2s(v.z, v.w, ascale, b0, b1);
    const unsigned dst = vwb + (unsigned)xi * WN_XI;
    const wn_u32x2 hi = {a0, b0}, lo = {a1, b1};
    *(wn_lds_u2)(size_t)dst = hi;
    *(wn_lds_u2)(size_t)(dst + WN_PLANE) = lo;
  };
  // transform rows (ra, rb) = (1, 2) for group 0: V[1] = s1 + s2, V[2] = s2 - s1;  (0, 3) for group 1: V[0] = s0 - s2, V[3] = s1 - s3
  auto xform_g0 = [&]() {
    float4 d[4], sa[4], sb[4];
    raw_row(1, d); row_x(d, sa);
    raw_row(2, d); row_x(d, sb);
#pragma unroll
    for (int c = 0; c < 4; ++c) { emit(4 + c, WN_F4(+, sa[c], sb[c])); emit(8 + c, WN_F4(-, sb[c], sa[c])); }
  };
  auto xform_g1 = [&]() {
    float4 d[4], sa[4], sb[4];
    raw_row(0, d); row_x(d, sa);
    raw_row(2, d); row_x(d, sb);
#pragma unroll
    for (int c = 0; c < 4; ++c) emit(c, WN_F4(-, sa[c], sb[c]));
    raw_row(1, d); row_x(d, sa);
    raw_row(3, d); row_x(d, sb);
#pragma unroll
    for (int c = 0; c < 4; ++c) emit(12 + c, WN_F4(-, sa[c], sb[c]));
  };
  auto xform = [&]() { if (grp == 0) xform_g0(); else xform_g1(); };

  // ---- MFMA operands: transform row wr, channel block grp
  const unsigned a_lane = lds0 + (unsigned)(li * 32 + ((h ^ ((li >> 3) & 1)) * 16)) + (unsigned)wr * 4u * WN_XI;
  unsigned abase = a_lane;
  uint4 fa[2][2] = {};                   // [mb][plane]
  auto lda = [&](int c) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        fa[mb][p] = __builtin_bit_cast(uint4, *(wn_lds_u4)(size_t)(abase + (unsigned)c * WN_XI + (unsigned)(mb * 1024) + (unsigned)p * WN_PLANE));
  };
  const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.wp), 0, 0x7fffffff, 0x00020000);
  const unsigned ps_b = (unsigned)nch * 16u * (unsigned)NT * 1024u;
  const int lane16 = lane * 16;
  uint4 qb[2][2] = {};                   // [c & 1][plane]: U fragments are requested two columns ahead (two waves per SIMD cover the rest of the L2 latency)
  auto ldb = [&](uint4 (&bq)[2], int chunk, int c, int nt0) {
    const unsigned soff = (unsigned)((chunk * 16 + wr * 4 + c) * NT + nt0 + grp) * 1024u;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      bq[p] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srd_w, lane16, (int)(soff + (unsigned)p * ps_b), 0));
  };
  f32x16 acc[4][2];                      // [c][mb]
  auto mma6 = [&](const uint4 (&bq)[2], int c, bool zc) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
      acc[c][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[mb][0]), __builtin_bit_cast(f16x8, bq[1]), zc ? zero : acc[c][mb], 0, 0, 0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
      acc[c][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[mb][1]), __builtin_bit_cast(f16x8, bq[0]), acc[c][mb], 0, 0, 0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
      acc[c][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[mb][0]), __builtin_bit_cast(f16x8, bq[0]), acc[c][mb], 0, 0, 0);
  };

  // ---- prologue
  int tile = vb;
  if (tile >= ntiles) return;
  int c_img, c_y0, c_x0, c_n0;
  WN_DECODE(tile, c_img, c_y0, c_x0, c_n0)
  W8_LD_SETUP()
  gload();
  W8_LD_ADVANCE()
  raw_store();
  gload();
  W8_LD_ADVANCE()
  __syncthreads();
  cur = 1;
  vwb = vw_base;
  xform();
  __syncthreads();
  raw_store();
  ldb(qb[0], 0, 0, c_n0 >> 5);
  ldb(qb[1], 0, 1, c_n0 >> 5);
  gload();
  W8_LD_ADVANCE()
  cur = 0;
  __syncthreads();

// The two waves of a SIMD (one of each group) run the first half of a chunk in OPPOSITE order -- group 0: transform, then its 12 MFMAs; group 1: its 12
// MFMAs, then the transform -- so that one wave's vector / LDS work sits beside the other's matrix work (both waves in the same phase waited at the
// same s_waitcnt: 0.41 of the wave cycles parked, MFMA utilisation 0.24).  PDAE_WINO_SKEW=0 (template SKEW) keeps the common order (A-B aid).
#define W8_MM(C_, FIRST, CH, CNT, NCH, NNT) WN_PA(lda(C_);) mma6(qb[(C_) & 1], C_, FIRST);                  \
    WN_PB(if ((C_) < 2) ldb(qb[(C_) & 1], CH, (C_) + 2, CNT); else ldb(qb[(C_) & 1], NCH, (C_) - 2, NNT);)
#define W8_ITER(FIRST, CH, CNT, NCH, NNT)                                                                              \
  {                                                                                                           \
    abase = a_lane + cur * WN_VBUF;                                                                           \
    vwb = vw_base + (cur ^ 1u) * WN_VBUF;                                                                     \
    asm volatile("" : "+v"(abase), "+v"(vwb));                                                                \
    /* the matrix code is common to both groups (a branch around it cost 112 spilled accumulator copies): only the transforms sit in branches */ \
    if (SKEW && grp == 0) { WN_PXF(xform_g0();) }                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    W8_MM(0, FIRST, CH, CNT, NCH, NNT)                                                                        \
    if (!SKEW) { WN_PXF(xform();) }                                                                           \
    W8_MM(1, FIRST, CH, CNT, NCH, NNT)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    if (SKEW && grp != 0) { WN_PXF(xform_g1();) }                                                             \
    WN_PBARA(__syncthreads();) /* every raw read of this step is done */                                      \
    W8_MM(2, FIRST, CH, CNT, NCH, NNT)                                                                                 \
    WN_PRAW(raw_store();)                                                                                     \
    W8_MM(3, FIRST, CH, CNT, NCH, NNT)                                                                                 \
    WN_PRAW(gload();)                                                                                         \
    W8_LD_ADVANCE()                                                                                           \
    __syncthreads();                                                                                          \
    cur ^= 1u;                                                                                                \
  }

  for (; tile < ntiles; tile += G) {
    int x_img, x_y0, x_x0, x_n0;
    const int ntile = tile + G < ntiles ? tile + G : tile;
    WN_DECODE(ntile, x_img, x_y0, x_x0, x_n0)
    const int c_nt = c_n0 >> 5, x_nt = x_n0 >> 5;
    W8_ITER(true, 0, c_nt, nch > 1 ? 1 : 0, nch > 1 ? c_nt : x_nt)
    for (int ch = 1; ch < nch; ++ch) {
      const bool lastc = ch + 1 == nch;
      W8_ITER(false, ch, c_nt, lastc ? 0 : ch + 1, lastc ? x_nt : c_nt)
    }
#ifndef PDAE_WN_PROBE_NOEPI
    // ---- output transform: along c in registers, along r through LDS (four waves of the same channel block)
    int tt = t;
    asm volatile("" : "+v"(tt));
    const unsigned ex_w = (unsigned)(wr * 16384 + (4 * (tt >> 5 & 1) * 64 + grp * 32 + (tt & 31)) * 4);
    const unsigned ex_r = (unsigned)((tt >> 4) * 256 + (tt & 15) * 16);
    const int e_wy = tt >> 7, e_wx = (tt >> 4) & 7, e_c4 = (tt & 15) * 4;
    const unsigned exb = lds0 + (cur ^ 1u) * WN_VBUF;
    const float4 bias4 = P.bias ? *reinterpret_cast<const float4*>(P.bias + c_n0 + e_c4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float z = j == 0 ? (acc[0][mb][r] + acc[1][mb][r]) + acc[2][mb][r] : (acc[1][mb][r] - acc[2][mb][r]) - acc[3][mb][r];
          *(wn_lds_f)(size_t)(exb + ex_w + (unsigned)(((mb * 32 + (r & 3) + 8 * (r >> 2)) * 64) * 4)) = z;
        }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        wn_f32x4 z[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = *(wn_lds_f4)(size_t)(exb + ex_r + (unsigned)(k * 8192 + r * 16384));
        const int oy = c_y0 + 2 * (e_wy + 4 * k), ox = c_x0 + 2 * e_wx + j;
        float* dst = P.y + ((size_t)(c_img * P.H + oy) * P.W + ox) * P.Nout + c_n0 + e_c4;
        float4 y0v, y1v;
        y0v.x = fmaf((z[0][0] + z[1][0]) + z[2][0], oscale, bias4.x); y0v.y = fmaf((z[0][1] + z[1][1]) + z[2][1], oscale, bias4.y);
        y0v.z = fmaf((z[0][2] + z[1][2]) + z[2][2], oscale, bias4.z); y0v.w = fmaf((z[0][3] + z[1][3]) + z[2][3], oscale, bias4.w);
        y1v.x = fmaf((z[1][0] - z[2][0]) - z[3][0], oscale, bias4.x); y1v.y = fmaf((z[1][1] - z[2][1]) - z[3][1], oscale, bias4.y);
        y1v.z = fmaf((z[1][2] - z[2][2]) - z[3][2], oscale, bias4.z); y1v.w = fmaf((z[1][3] - z[2][3]) - z[3][3], oscale, bias4.w);
#ifdef PDAE_WN_PROBE_NOSTORE
        if (y0v.x == 123.456f && y1v.y == 654.321f)
#endif
        {
          *reinterpret_cast<float4*>(dst) = y0v;
          *reinterpret_cast<float4*>(dst + (size_t)P.W * P.Nout) = y1v;
        }
      }
      __syncthreads();
    }
#else
    {
      float sum_ = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum_ += acc[c][mb][r];
      if (sum_ == 123.456f) P.y[t] = sum_;
    }
#endif
    c_img = x_img; c_y0 = x_y0; c_x0 = x_x0; c_n0 = x_n0;
  }
  pdae_sat_report(P.sat, sat_hit);
}

// ---- weight preparation: U[xi = (r, c)] = sum_{ky, kx} G[r][ky] G[c][kx] g[ky][kx], scaled by a power of two, split into two fp16 planes, in
// MFMA B-fragment order [plane][C/16][xi][NT][lane][8]: lane (n = lane % 32, k half = lane / 32) holds input channels chunk * 16 + 8 * half + 0..7
// of output channel nt * 32 + n.  w is [Cout][3][3][C] (reference shape (Cout, C, 3, 3) in channels-last memory).
__global__ void __launch_bounds__(256) wino_wprep_kernel(const float* __restrict__ w, int Nout, int C, int NT, float wscale, unsigned short* __restrict__ wp) {
  const size_t nslot = (size_t)(C >> 4) * 16 * NT * 64, plane_stride = nslot * 8;
  const float Gm[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nslot; i += (size_t)gridDim.x * 256) {
    const int lane = (int)(i & 63); size_t r = i >> 6;
    const int nt = (int)(r % NT); r /= NT;
    const int xi = (int)(r & 15); const int chunk = (int)(r >> 4);
    const int n = nt * 32 + (lane & 31), c0 = chunk * 16 + (lane >> 5) * 8;
    const int tr = xi >> 2, tc = xi & 3;
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = 0.f;
    if (n < Nout) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float g2 = Gm[tr][ky] * Gm[tc][kx];                    // exact: 0, +-1, +-1/2, +-1/4
          const float* src = w + ((size_t)n * 9 + ky * 3 + kx) * C + c0;
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = fmaf(g2, src[j], e[j]);
        }
    }
    wprep_store_slot<4>(e, wscale, wp, plane_stride, i);
  }
}

bool wino_ok(int math, int KH, int KW, int stride, int pad, int up, int C0, int C1, int H, int W, int N, int Nout) {
  if (math != 4 || KH != 3 || KW != 3 || stride != 1 || pad != 1 || up || C1 != 0) return false;
  if ((C0 & 15) || (H & 15) || (W & 15) || (Nout & 63) || H >= 2048 || W >= 2048) return false;
  const unsigned long long lim = 0xFFFFFFE0ull;
  if ((unsigned long long)N * H * W * C0 * 4ull >= lim || (unsigned long long)N * H * W * Nout * 4ull >= lim) return false;
  return true;
}

size_t wino_wprep_bytes(int Nout, int C) { return (size_t)2 * (C >> 4) * 16 * (Nout >> 5) * 64 * 8 * sizeof(unsigned short); }

// the weights' power-of-two scale: |U| <= 2.25 max |g|, fan-in C per transform position
float wino_wscale(int C) { int k = 0; while ((1 << (2 * k)) < 4 * C) ++k; return (float)(1 << k); }

int wino_wprep(const float* w, int Nout, int C, unsigned short* wp, hipStream_t s) {
  const size_t nslot = (size_t)(C >> 4) * 16 * (Nout >> 5) * 64;
  const int grid = (int)((nslot + 255) / 256 < 2048 ? (nslot + 255) / 256 : 2048);
  hipLaunchKernelGGL(wino_wprep_kernel, dim3(grid), dim3(256), 0, s, w, Nout, C, Nout >> 5, wino_wscale(C), wp);
  return pdae_launch_status("wino_wprep");
}

static int wn_sched() { const char* e = getenv("PDAE_WINO_SCHED"); return e ? atoi(e) : 8; }

template <int SCHED> static int wino_launch_t(const WinoParams& P, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wino_kernel<SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WN_LDS);
    if (e != hipSuccess) { pdae_set_error("wino: cannot raise dynamic LDS to %u: %s", WN_LDS, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  const long long ntiles = (long long)P.N * P.tiles_y * P.tiles_x * P.tiles_n;
  dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256));
  hipLaunchKernelGGL((wino_kernel<SCHED>), grid, dim3(256), WN_LDS, s, P);
  return pdae_launch_status("wino_fwd");
}

template <int SKEW> static int wino8_launch_t(const WinoParams& P, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wino8_kernel<SKEW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WN_LDS);
    if (e != hipSuccess) { pdae_set_error("wino8: cannot raise dynamic LDS to %u: %s", WN_LDS, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  const long long ntiles = (long long)P.N * P.tiles_y * P.tiles_x * P.tiles_n;
  hipLaunchKernelGGL((wino8_kernel<SKEW>), dim3((unsigned)(ntiles < 256 ? ntiles : 256)), dim3(512), WN_LDS, s, P);
  return pdae_launch_status("wino8_fwd");
}

int wino_fwd(const float* x, int N, int H, int W, int C, const unsigned short* wp, int Nout, const float* bias, float* y, hipStream_t s) {
  WinoParams P;
  P.x = x; P.N = N; P.H = H; P.W = W; P.C = C; P.wp = wp; P.NT = Nout >> 5; P.Nout = Nout; P.y = y; P.bias = bias;
  P.woscale = 1.0f / wino_wscale(C); P.sat = pdae_sat_counter();
  P.tiles_x = W / 16; P.tiles_y = H / 16; P.tiles_n = Nout / 64;
  const int sc = wn_sched();
  if (sc == 8 || sc == 9) return sc == 8 ? wino8_launch_t<1>(P, s) : wino8_launch_t<0>(P, s);      // 8: phase-skewed wave pairs (default form), 9: common order
  return sc == 0 ? wino_launch_t<0>(P, s) : (sc == 2 ? wino_launch_t<2>(P, s) : wino_launch_t<1>(P, s));
}
