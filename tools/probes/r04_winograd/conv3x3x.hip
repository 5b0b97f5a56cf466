// 3x3 / stride-1 / pad-1 convolution (forward and data gradient) for gfx950 with the Winograd F(2, 3) transform ALONG X ONLY -- the patch kernel
// (conv3x3p.hip: same operand formats, prepared-weight fragments, fused GroupNorm input, fused 1x1 skip chunks, epilogue) with two thirds of its
// matrix work.
//
//   y[i][2m + j] = sum_ky sum_c AT[j][c] * ( s[i + ky][m][c] . U[ky][c] ),   s[.][m][c] = sum_b BT[c][b] x[.][2m - 1 + b],   U[ky][c] = sum_kx G[c][kx] w[ky][kx]
//
// i.e. the three vertical taps stay direct, the three horizontal taps become four transform positions per PAIR of output pixels: 12 instead of 18
// products per pixel pair.  Why this form and not F(2x2, 3x3) (csrc/winograd.hip, profiles/r04_winograd_probe.txt): the accumulators only double
// (4 positions per 2 pixels) -- a 16 x 16 pixel x 128 channel tile still fits 8 waves x 128 registers, so every input element is transformed ONCE
// per pixel tile --, the transformed input has 2x (not 4x) the elements at one addition each, and the output transform is lane-local (the four
// positions of a pixel pair live in the same lane of four accumulators of one wave): no cross-wave exchange.
//
// Geometry: 512 threads = 8 waves (two per SIMD), tile 16 x 16 pixels x 128 output channels; wave (wm, wn) = 8 rows x 32 channels, accumulators
// acc[c][a2] = transform position c, tile half a2 (MFMA row i <-> image row i / 4, pixel pair a2 * 4 + i % 4).  LDS patch in the transform
// domain: 18 rows x 36 positions (c * 8 + pair; 32 used) x 32 channels per plane, 80-byte rows: 36 * 80 = 64 (mod 256), so the fragment reads
// are conflict-free like conv3x3p's 20-pixel pitch.  Staging item = (patch row, pixel pair, channel quad): four float4 in, (GroupNorm map,) four
// additions, operand split, eight 8-byte LDS stores; a wave stages whole patch rows (row = wave + 8 l).  K loop: per chunk 3 x 4 x 2 steps of
// 6 MFMAs, software-pipelined one step deep exactly like conv3x3p.  Prepared weights: [plane][chunk][12 = ky * 4 + c][k half][32-channel tile]
// [lane][8] (wprepx_slot, conv3x3p.h); fused 1x1 skip chunks are the centre tap: positions c = 1, 2 of ky = 1 with weights +-w / 2.
// Replaces F.conv2d(k=3, padding=1) of model/module.py:242,265 (+ nearest upsample :169) and its input gradient on the large layers.
#include <stdlib.h>

#include "common.h"
#include "igemm.h"
#include "conv3x3p.h"

#define XTH 16
#define XPW 36                                   // positions per patch row (32 used)
#define XNPOS ((XTH + 2) * XPW)                  // 648 position rows per plane
#define XTHREADS 512
#define XLD 3                                    // staging items per thread: 18 rows x 8 pairs x 8 quads = 1152 = 2.25 x 512
#define XPLANE_B (PPLANE(XNPOS) * 2)             // bytes per plane (51840)
#define XOOB 0xFFFFFFF0u
#define XALL 0xFFFFFFEFu

typedef unsigned x_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned x_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const x_u32x4* x_lds_u4;
typedef __attribute__((address_space(3))) x_u32x2* x_lds_u2;

#define PDAE_X_PATTERN                                                                                     \
  _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                                       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
    __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);                                                     \
    if (i_ < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                \
  }

template <int NS, bool GN>
__global__ void __launch_bounds__(XTHREADS, 2) conv3x3x_kernel(const PatchParams P) {
  constexpr int NP = NPL(NS);
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* sA = smem;
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, h = lane >> 5;
  const int wm = wv >> 2, wn = wv & 3;

  // block -> (image, tile_y, tile_x, n-tile); n-tile fastest, workgroups of one XCD take a contiguous range (conv3x3p)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  int tid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int tn_i = tid % P.tiles_n; tid /= P.tiles_n;
  const int tx_i = tid % P.tiles_x; tid /= P.tiles_x;
  const int ty_i = tid % P.tiles_y; tid /= P.tiles_y;
  const int img = tid;
  const int y0 = ty_i * XTH, x0 = tx_i * PTW, n0 = tn_i * PBN;
  const int C = P.C;

  // ---- staging items of this thread: (patch row py = wv + 8 l, pixel pair wt, channel quad qd).  The item needs image columns x0 - 1 + 2 wt + b,
  // b = 0..3; the thread LOADS its own two (b = 1, 2) and, as first / last pair of the row, the edge column (b = 0 of wt = 0, b = 3 of wt = 7);
  // b = 0 / b = 3 of the other pairs are the neighbours' own pixels, fetched across lanes (the eight pairs of a (row, quad) sit 8 lanes apart in
  // one wave).  36 prefetch registers instead of 48, and the GroupNorm map runs once per pixel, not twice.
  const int qd = t & 7, wt = (t >> 3) & 7;
  int pb[XLD];                                   // source pixel index of b = 1
  unsigned vmask = 0u;                           // per item: bit 0 row valid (own pixels always lie inside the image then), bit 1 edge pixel valid
  const int eb = wt == 0 ? 0 : 3;                // which b the edge load fetches (only lanes of the first / last pair use it)
#pragma unroll
  for (int l = 0; l < XLD; ++l) {
    const int py = wv + 8 * l, ly = y0 - 1 + py, lx1 = x0 + 2 * wt;
    const bool rok = py < XTH + 2 && (unsigned)ly < (unsigned)P.H;
    const int sy = P.up ? ly >> 1 : ly, sx1 = P.up ? lx1 >> 1 : lx1;
    pb[l] = (img * P.Hs + sy) * P.Ws + sx1;
    const int lxe = lx1 - 1 + eb;
    if (rok) vmask |= 1u << (2 * l);
    if (rok && (wt == 0 || wt == 7) && (unsigned)lxe < (unsigned)P.W) vmask |= 2u << (2 * l);
  }
  // source offsets relative to b = 1: b = 2 -> +1 (same source pixel when upsampled), edge b = 0 -> -1, b = 3 -> +2 (+1 upsampled)
  const int ob2 = P.up ? 0 : 1, obe = wt == 0 ? -1 : (P.up ? 1 : 2);
  const float ascale = NS == 4 ? (P.amax ? p_pow2_scale(*P.amax) : PASCALE) : 1.0f;
  float4 apre[XLD][3];                           // [item][own b = 1, own b = 2, edge]
  float4 gmu, gsc, gsh;
  float sat_hit = 0.f;
  bool pre_raw = false;
  const int nmain = C >> 5, nchunk = nmain + P.nx;
  auto a_gload = [&](int chunk) {
    const float* src; unsigned ldb4, cb4;
    pre_raw = chunk >= nmain;
    if (pre_raw) {
      const int c = (chunk - nmain) << 5;
      const bool first = c < P.Cs0;
      src = first ? P.s0 : P.s1; ldb4 = (unsigned)(first ? P.Cs0 : P.Cs1) * 4u; cb4 = (unsigned)(first ? c : c - P.Cs0) * 4u;
    } else {
      const int c = chunk << 5;
      const bool first = c < P.C0;
      src = first ? P.x : P.x1; ldb4 = (unsigned)(first ? P.C0 : C - P.C0) * 4u; cb4 = (unsigned)(first ? c : c - P.C0) * 4u;
      if constexpr (GN) {
        const size_t NC = (size_t)P.N * C;
        const float* cf = P.coef + (size_t)img * C + c + qd * 4;
        gmu = *reinterpret_cast<const float4*>(cf); gsc = *reinterpret_cast<const float4*>(cf + NC); gsh = *reinterpret_cast<const float4*>(cf + 2 * NC);
      }
    }
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)XALL, 0x00020000);
#pragma unroll
    for (int l = 0; l < XLD; ++l) {
      const unsigned v1 = (unsigned)pb[l] * ldb4 + (unsigned)(qd * 16);
      const bool rok = (vmask >> (2 * l)) & 1u, eok = (vmask >> (2 * l + 1)) & 1u;
      apre[l][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)(rok ? v1 : XOOB), (int)cb4, 0));
      apre[l][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)(rok ? v1 + (unsigned)ob2 * ldb4 : XOOB), (int)cb4, 0));
      apre[l][2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)(eok ? v1 + (unsigned)obe * ldb4 : XOOB), (int)cb4, 0));
    }
  };
  const unsigned w_lane = (unsigned)(size_t)sA + (unsigned)((PSLOT(wv * XPW + wt, qd >> 1) + (qd & 1) * 4) * 2);
  auto gn_map = [&](float4 v, bool on) {
    if (on) {                                    // padding pixels stay zero AFTER the map
      v.x = gsc.x * (v.x - gmu.x) + gsh.x; v.y = gsc.y * (v.y - gmu.y) + gsh.y;
      v.z = gsc.z * (v.z - gmu.z) + gsh.z; v.w = gsc.w * (v.w - gmu.w) + gsh.w;
      if (P.act) { v.x = p_silu(v.x); v.y = p_silu(v.y); v.z = p_silu(v.z); v.w = p_silu(v.w); }
    }
    return v;
  };
  auto shfl4 = [&](const float4& v, int src_lane) {
    return make_float4(__shfl(v.x, src_lane), __shfl(v.y, src_lane), __shfl(v.z, src_lane), __shfl(v.w, src_lane));
  };
  auto a_lstore = [&]() {
    const float sc = pre_raw ? 1.0f : ascale;      // skip chunks carry the RAW residual stream: unit scale, the 2^4 sits in their weights
#pragma unroll
    for (int l = 0; l < XLD; ++l) {
      if (wv + 8 * l < XTH + 2) {                  // wave-uniform
        float4 d1 = apre[l][0], d2 = apre[l][1], de = apre[l][2];
        if constexpr (GN) {
          const bool rok = (vmask >> (2 * l)) & 1u, eok = (vmask >> (2 * l + 1)) & 1u;
          d1 = gn_map(d1, rok && !pre_raw); d2 = gn_map(d2, rok && !pre_raw); de = gn_map(de, eok && !pre_raw);
        }
        if constexpr (NS == 4) {                    // a transformed value is a sum / difference of two inputs
          pdae_f16_amax4(d1, 2.0f * sc, sat_hit); pdae_f16_amax4(d2, 2.0f * sc, sat_hit); pdae_f16_amax4(de, 2.0f * sc, sat_hit);
        }
        // b = 0 is the previous pair's b = 2 pixel, b = 3 the next pair's b = 1 pixel (8 lanes away); the row's ends use the edge load
        float4 d0 = shfl4(d2, lane - 8), d3 = shfl4(d1, lane + 8);
        if (wt == 0) d0 = de;
        if (wt == 7) d3 = de;
#define X_F4(OP, A_, B_) make_float4(A_.x OP B_.x, A_.y OP B_.y, A_.z OP B_.z, A_.w OP B_.w)
        const float4 s[4] = {X_F4(-, d0, d2), X_F4(+, d1, d2), X_F4(-, d2, d1), X_F4(-, d1, d3)};
#undef X_F4
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          unsigned a[NP], b2[NP];
          p_split2<NS>(s[c].x, s[c].y, a, sc);
          p_split2<NS>(s[c].z, s[c].w, b2, sc);
          const unsigned dst = w_lane + (unsigned)((l * 8 * XPW + c * 8) * PLDH * 2);
#pragma unroll
          for (int p = 0; p < NP; ++p) { const x_u32x2 w2 = {a[p], b2[p]}; *(x_lds_u2)(size_t)(dst + (unsigned)(p * XPLANE_B)) = w2; }
        }
      }
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][a][r] = 0.f;

  // ---- operands of one step (tp = ky * 4 + c, k half kc)
  const int nt0 = min((n0 >> 5) + wn, P.NT - 1);
  const size_t plane_main = (size_t)nmain * 24 * P.NT * 512, plane_skip = (size_t)P.nx * 4 * P.NT * 512;      // 16-bit elements per plane
  const __amdgpu_buffer_rsrc_t srd_main = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.wp), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_skip = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.wps ? P.wps : P.wp), 0, 0x7fffffff, 0x00020000);
  const int lane16 = lane * 16;
  auto ldb = [&](uint4 (&bq)[NP], int chunk, int tp, int kc) {
    const bool raw = chunk >= nmain;
    // skip chunks: [chunk - nmain][c - 1 in {0, 1}][kc]; tp = 5, 6 there
    const unsigned soff = (unsigned)((raw ? ((((chunk - nmain) * 2 + (tp - 5)) << 1) + kc) * P.NT + nt0 : (((chunk * 12 + tp) << 1) + kc) * P.NT + nt0) * 1024);
    const unsigned ps2 = (unsigned)((raw ? plane_skip : plane_main) * 2);
#pragma unroll
    for (int p = 0; p < NP; ++p)
      bq[p] = __builtin_bit_cast(uint4, raw ? __builtin_amdgcn_raw_buffer_load_b128(srd_skip, lane16, (int)(soff + p * ps2), 0)
                                            : __builtin_amdgcn_raw_buffer_load_b128(srd_main, lane16, (int)(soff + p * ps2), 0));
  };
  const unsigned a_lane = (unsigned)(size_t)sA + (unsigned)(PSLOT((wm * 8 + (li >> 2)) * XPW + (li & 3), h) * 2);
  auto lda = [&](uint4 (&af)[2][NP], int tp, int kc) {
    const int ky = tp >> 2, c = tp & 3;
    unsigned ab = a_lane + (unsigned)((ky * XPW + c * 8) * (PLDH * 2) + kc * 32);
    asm volatile("" : "+v"(ab));
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int p = 0; p < NP; ++p) af[a][p] = __builtin_bit_cast(uint4, *(x_lds_u4)(size_t)(ab + (unsigned)(p * XPLANE_B + a * 4 * PLDH * 2)));
  };
  auto mma = [&](const uint4 (&af)[2][NP], const uint4 (&bq)[NP], int c) {
#define PDAE_XA(P_) __builtin_bit_cast(bf16x8, af[a][P_])
#define PDAE_XB(P_) __builtin_bit_cast(bf16x8, bq[P_])
#define PDAE_XAH(P_) __builtin_bit_cast(f16x8, af[a][P_])
#define PDAE_XBH(P_) __builtin_bit_cast(f16x8, bq[P_])
#define PDAE_X_EACH(STMT) _Pragma("unroll") for (int a = 0; a < 2; ++a) { STMT; }
    if constexpr (NS == 4) {
      PDAE_X_EACH(acc[c][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_XAH(0), PDAE_XBH(1), acc[c][a], 0, 0, 0))
      PDAE_X_EACH(acc[c][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_XAH(1), PDAE_XBH(0), acc[c][a], 0, 0, 0))
      PDAE_X_EACH(acc[c][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_XAH(0), PDAE_XBH(0), acc[c][a], 0, 0, 0))
    } else {
      if constexpr (NS == 2) {
        PDAE_X_EACH(acc[c][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_XA(0), PDAE_XB(1), acc[c][a], 0, 0, 0))
        PDAE_X_EACH(acc[c][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_XA(1), PDAE_XB(0), acc[c][a], 0, 0, 0))
      }
      PDAE_X_EACH(acc[c][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_XA(0), PDAE_XB(0), acc[c][a], 0, 0, 0))
    }
#undef PDAE_X_EACH
#undef PDAE_XA
#undef PDAE_XB
#undef PDAE_XAH
#undef PDAE_XBH
  };

  // ---- K loop.  Steps of a main chunk: (ky, c, kc) in lexicographic order; of a skip chunk: (1, c in {1, 2}, kc).  Buffer set 0 serves kc = 0,
  // set 1 serves kc = 1; every step requests the operands of the step after it before its own MFMAs.
  // (A fragments single-buffered: with 128 accumulator registers per wave there is no room for a second set; the other wave of the SIMD
  // covers the LDS latency)
  uint4 q0[NP] = {}, q1[NP] = {}, f0[2][NP] = {};
  // one transform column: steps (tp, 0) and (tp, 1); NTP = tp of the step after (tp, 1) in chunk NCHUNK
#ifdef PDAE_X_PROBE_NOA
#define X_PA(X)
#else
#define X_PA(X) X
#endif
#ifdef PDAE_X_PROBE_NOB
#define X_PB(X)
#else
#define X_PB(X) X
#endif
#define PDAE_X_COL(C_, TP_, NCHUNK_, NTP_)                                                                    \
  {                                                                                                           \
    X_PB(ldb(q1, chunk, TP_, 1);)                                                                             \
    mma(f0, q0, C_);                                                                                          \
    PDAE_X_PATTERN                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    X_PA(lda(f0, TP_, 1);)                                                                                    \
    X_PB(ldb(q0, NCHUNK_, NTP_, 0);)                                                                          \
    mma(f0, q1, C_);                                                                                          \
    PDAE_X_PATTERN                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    X_PA(lda(f0, NTP_, 0);)                                                                                   \
  }
  const int c_end = nchunk;
  a_gload(0);
  ldb(q0, 0, 0, 0);                              // (a launch always has at least one main chunk)
  a_lstore();
  __syncthreads();
  if (1 < c_end) a_gload(1);
  for (int chunk = 0; chunk < c_end; ++chunk) {
    const bool raw = chunk >= nmain;
    const bool nextc = chunk + 1 < c_end;
    const int nchunk_i = nextc ? chunk + 1 : chunk;
    const int first_next = nextc ? (chunk + 1 >= nmain ? 5 : 0) : (raw ? 5 : 0);      // first step of the next chunk (or a harmless repeat at the very end)
    if (!raw) {
      lda(f0, 0, 0);
      for (int ky = 0; ky < 3; ++ky) {
        const int tb = ky * 4;
        const bool more = ky < 2;
        PDAE_X_COL(0, tb + 0, chunk, tb + 1)
        PDAE_X_COL(1, tb + 1, chunk, tb + 2)
        PDAE_X_COL(2, tb + 2, chunk, tb + 3)
        PDAE_X_COL(3, tb + 3, more ? chunk : nchunk_i, more ? tb + 4 : first_next)
      }
    } else {
      lda(f0, 5, 0);
      PDAE_X_COL(1, 5, chunk, 6)
      PDAE_X_COL(2, 6, nchunk_i, first_next)
    }
#ifndef PDAE_X_PROBE_NOSTAGE
    if (nextc) {
      __syncthreads();
      a_lstore();
      __syncthreads();
      if (chunk + 2 < c_end) a_gload(chunk + 2);
    }
#endif
  }
#undef PDAE_X_COL

  // ---- epilogue: output transform (lane-local), then conv3x3p's transposition through a private LDS tile: float4 per lane, 8 lanes per pixel
  const float oscale = NS == 4 ? P.woscale / ascale : 1.0f;
  if constexpr (NS == 4) pdae_sat_report(P.sat, sat_hit);
  __syncthreads();
  float* tw = reinterpret_cast<float*>(smem) + wv * (32 * EPW);
  const int er = lane >> 3, ec = (lane & 7) * 4;
  const int colb = n0 + wn * 32 + ec;
  const int colc = colb < P.Nout ? colb : 0;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (colb < P.Nout) {
    if (P.bias) bias4 = *reinterpret_cast<const float4*>(P.bias + colb);
    if (P.bias_x) { const float4 u = *reinterpret_cast<const float4*>(P.bias_x + colb); bias4.x += u.x; bias4.y += u.y; bias4.z += u.z; bias4.w += u.w; }
  }
  const bool col_ok = colb < P.Nout;
  const unsigned lane_d2 = (unsigned)((er & 3) * P.Nout + colc);                   // half-resolution residual: pixel (0, er & 3) of the 1 x 4 sub-block
  const size_t row_pair = (size_t)2 * P.W * P.Nout;
  const bool want_stat = P.stat_part != nullptr;
  float st1 = 0.f, st2 = 0.f;
  const int y0a = y0 + wm * 8;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int a2 = g >> 1, j = g & 1;            // tile half, pixel parity: MFMA row i -> pixel (y0a + i / 4, x0 + a2 * 8 + 2 (i % 4) + j)
    const int x0a = x0 + a2 * 8;
    const unsigned lane_d = (unsigned)(((er >> 2) * P.W + 2 * (er & 3) + j) * P.Nout + colc);
    const size_t rb = (((size_t)img * P.H + y0a) * P.W + x0a) * P.Nout;
    float4 rv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) rv[it] = bias4;
    if (P.res_mode) {
      const size_t rb2 = (((size_t)img * (P.H >> 1) + (y0a >> 1)) * (P.W >> 1) + (x0a >> 1)) * P.Nout;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float* src = P.res_mode == 2 ? P.res + rb2 + (size_t)it * (P.W >> 1) * P.Nout + lane_d2 : P.res + rb + it * row_pair + lane_d;
        const float4 u = *reinterpret_cast<const float4*>(src);
        rv[it].x += u.x; rv[it].y += u.y; rv[it].z += u.z; rv[it].w += u.w;
      }
    }
    if (P.accumulate) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float4 u = *reinterpret_cast<const float4*>(P.y + rb + it * row_pair + lane_d);
        rv[it].x += u.x; rv[it].y += u.y; rv[it].z += u.z; rv[it].w += u.w;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float z = j == 0 ? (acc[0][a2][r] + acc[1][a2][r]) + acc[2][a2][r] : (acc[1][a2][r] - acc[2][a2][r]) - acc[3][a2][r];
      tw[((r & 3) + 8 * (r >> 2) + 4 * h) * EPW + li] = z;
    }
    float* dst = P.y + rb + lane_d;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float4 v = *reinterpret_cast<const float4*>(&tw[(it * 8 + er) * EPW + ec]);
      v.x = fmaf(v.x, oscale, rv[it].x); v.y = fmaf(v.y, oscale, rv[it].y); v.z = fmaf(v.z, oscale, rv[it].z); v.w = fmaf(v.w, oscale, rv[it].w);
      if (col_ok) *reinterpret_cast<float4*>(dst + it * row_pair) = v;
      if (want_stat) {
        st1 += (v.x + v.y) + (v.z + v.w);
        st2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, st2))));
      }
    }
  }
  if (want_stat) {      // (sum, sum of squares) of the wave's 128 pixels per channel quad, in the layout of conv3x3p's 16-row tiles
    st1 += __shfl_xor(st1, 8); st2 += __shfl_xor(st2, 8);
    st1 += __shfl_xor(st1, 16); st2 += __shfl_xor(st2, 16);
    st1 += __shfl_xor(st1, 32); st2 += __shfl_xor(st2, 32);
    if (lane < 8 && col_ok) {
      const int wtile = (ty_i * P.tiles_x + tx_i) * 2 + wm;
      reinterpret_cast<float2*>(P.stat_part)[((size_t)img * P.stat_tpi + wtile) * (P.Nout >> 2) + (colb >> 2)] = make_float2(st1, st2);
    }
  }
}

// ---- host side
static int x_mode() { const char* e = getenv("PDAE_W1"); return e ? atoi(e) : 1; }      // 0: off; 1 (default): layers with at least a chip-full of tiles; 2: every eligible shape (tests)

// Form of the prepared weights AND of the launch of a 3x3 convolution with these launch-side dimensions (C input channels, H x W output grid,
// Nout output channels): decided from the shape alone so that weight preparation and launch agree (the fused skip chunks follow the main
// convolution).  Read per call: do not change PDAE_W1 between preparing a convolution's weights and launching it.
bool conv3x3x_ok(int math, int C, int H, int W, int N, int Nout) {
  const int m = x_mode();
  if (m == 0) return false;
  if (!(math == 1 || math == 2 || math == 4)) return false;
  if ((H % XTH) || (W % PTW) || (Nout % PBN) || (C & 31) || H >= 2048 || W >= 2048) return false;
  const unsigned long long lim = 0xFFFFFFE0ull;
  if ((unsigned long long)N * H * W * (unsigned long long)(C > Nout ? C : Nout) * 4ull >= lim) return false;
  if (m == 2) return true;
  const long long tiles = (long long)N * (H / XTH) * (W / PTW) * (Nout / PBN);
  const long long rounds = (tiles + 255) / 256;
  static int min_eff = -1;                                    // PDAE_W1_EFF: minimum % of the CUs busy in the last round (tuning aid)
  if (min_eff < 0) { const char* e = getenv("PDAE_W1_EFF"); min_eff = e ? atoi(e) : 85; }
  return tiles >= 256 && tiles * 100 >= rounds * 256 * min_eff;      // persistent workgroups: the last round must not leave the chip idle
}

template <int NS, bool GN> static int launch_x(const PatchParams& P, hipStream_t s) {
  size_t smem = (size_t)NPL(NS) * XPLANE_B;
  const size_t epi = (size_t)8 * 32 * EPW * sizeof(float);
  if (smem < epi) smem = epi;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3x_kernel<NS, GN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { pdae_set_error("conv3x3x: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  dim3 grid((unsigned)(P.N * P.tiles_y * P.tiles_x * P.tiles_n));
  hipLaunchKernelGGL((conv3x3x_kernel<NS, GN>), grid, dim3(XTHREADS), smem, s, P);
  return pdae_launch_status("conv3x3x");
}

int conv3x3x_launch(int math, const PatchParams& P0, hipStream_t s) {
  PatchParams P = P0;
  P.tiles_x = P.W / PTW; P.tiles_y = P.H / XTH; P.tiles_n = P.Nout / PBN; P.splits = 1; P.cps = (P.C >> 5) + P.nx;
  const unsigned long long lim = 0xFFFFFFE0ull;
  const int smax = P.Cs0 > P.Cs1 ? P.Cs0 : P.Cs1;
  if ((unsigned long long)P.N * P.H * P.W * (unsigned long long)smax * 4ull >= lim) { pdae_set_error("conv3x3x: skip tensor beyond 4 GB"); return PDAE_EINVAL; }
  if (P.x1 && (P.C0 & 31)) { pdae_set_error("conv3x3x: two-source input needs C0 %% 32 == 0"); return PDAE_EINVAL; }
  // launches without fused skip chunks: the one-wave-per-SIMD persistent form (conv3x3y.hip) unless PDAE_W1_KERNEL=x
  {
    const char* e = getenv("PDAE_W1_KERNEL");
    if (P.nx == 0 && !(P.res_mode && P.accumulate) && !(e && e[0] == 'x')) return conv3x3y_launch(math, P, s);
  }
#define PDAE_X3(NS_) (P.coef ? launch_x<NS_, true>(P, s) : launch_x<NS_, false>(P, s))
  if (math == 1) return PDAE_X3(1);
  if (math == 2) return PDAE_X3(2);
  return PDAE_X3(4);
#undef PDAE_X3
}
