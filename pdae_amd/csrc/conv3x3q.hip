// 3x3 / stride-1 / pad-1 convolution (forward and data-gradient) for gfx950 -- the ONE-WAVE-PER-SIMD form of the patch kernel for the large
// layers (>= 64 x 64 pixels per image, output channels a multiple of 128), where conv3x3p.hip left 40 % of the matrix pipe idle.
//
// Why another kernel.  conv3x3p runs two waves per SIMD, each 128 pixels x 32 output channels (64 accumulator registers): per 12 MFMAs a wave
// issues 8 ds_read_b128 + 2 buffer_load_dwordx4 + ~3 VALU, and two such streams share one SIMD's issue port.  Measured there (profiles/r02_pmc_sq.txt):
// MFMA pipe busy 0.60, probe builds attribute ~28 % of the time to the weight-fragment loads alone.  The pipe issues one 32x32x16 MFMA per 32 cycles
// and hides at most ~5 other instructions in that gap (MI355X_MICROARCH.md, per-instruction constants), so the lever is work per MFMA:
//   wave tile 256 pixels x 64 output channels = 16 accumulators of 32x32 (256 registers, the AGPR half of the 512-entry file at one wave per SIMD):
//   per 48 MFMAs a wave issues 16 ds_read_b128 (each patch fragment now feeds two channel tiles) + 4 buffer loads (each weight fragment feeds
//   eight pixel groups): 0.42 loads per MFMA instead of 0.83, weight traffic per MFMA halved, no second wave competing for the issue port.
// Block = 4 waves = 32 x 16 pixels x 128 output channels (wave = pixel half x channel half); the 34 x 18 halo patch of a 32-channel chunk is staged
// once per block (same LDS layout as conv3x3p: 80-byte pixel rows, 20-pixel pitch, conflict-free ds_read_b128 for all 9 taps), 108.8 KB for two
// fp16 planes.  With a single wave per SIMD nothing else covers a stall, so the wave pipelines itself:
//   * patch fragments: ring of two "units" (2 pixel groups x planes), unit i+1 read from LDS under the 12 MFMAs of unit i;
//   * weight fragments: double buffer, k-step s+1 fetched from L2 under the 48 MFMAs of k-step s;
//   * next chunk's patch: ONE 16-byte buffer load per unit over the first 22 units of a chunk (out-of-image pixels are buffer-range misses: the
//     hardware returns zeros, no select), converted (GroupNorm / SiLU / fp16 split) one load per unit later in the chunk, so the hand-over at
//     the chunk boundary is barrier + 44 ds_write_b64 + barrier.
// Every tap / k-half / unit index is a compile-time constant of the fully unrolled chunk body: LDS addresses are one opaque base register plus
// an immediate, weight addresses one scalar offset -- no vector address arithmetic in the loop.
// Same operand formats, prepared-weight layout, fused GroupNorm input, fused 1x1 skip chunks, epilogue (bias / residual / accumulate / output
// statistics) and numerics as conv3x3p (bit-identical accumulation order per output element: chunks, taps, k-halves, products).
// Replaces F.conv2d(k=3, padding=1) of model/module.py:242,265 (+ nearest upsample :169) and its input gradient on the 128^2 / 64^2 layers.
#include <stdlib.h>

#include "common.h"
#include "igemm.h"
#include "conv3x3p.h"

#define QTH 32                                              // tile rows
#define QNPIX ((QTH + 2) * PPW)                             // 680 patch pixels (34 rows x 20-pixel pitch)
#define QTHREADS 256
#define QLD ((QNPIX * 8 + QTHREADS - 1) / QTHREADS)         // 22 float4 of the patch per thread and chunk
#define QBN 128
#define QPLANE_B (PPLANE(QNPIX) * 2)                        // bytes per LDS plane
#define QCV0 40                                             // first unit of a chunk that converts a prefetched float4 (unit l issues load l)

typedef unsigned q_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const q_u32x4* q_lds_u4;

// issue pattern of one unit: behind each of the 12 MFMAs up to NV scalar / vector ALU instructions and one load of the NEXT unit / k-step / chunk
#define PDAE_Q_PATTERN(NV)                                                                                  \
  _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
    __builtin_amdgcn_sched_group_barrier(0x006, NV, 0);                                                    \
    if (i_ < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    else if (i_ < 9) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                    \
  }

template <int NS, bool GN>
__global__ void __launch_bounds__(QTHREADS, 1) conv3x3q_kernel(const PatchParams P) {
  constexpr int NP = NPL(NS);
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* sA = smem;

  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, h = lane >> 5;
  const int wm = wv >> 1, wn = wv & 1;                      // pixel half (rows 16 wm .. 16 wm + 15), channel half (64 wn .. 64 wn + 63)

  // block -> (image, tile_y, tile_x, n-tile), XCD-aware like conv3x3p: n-tile fastest, blocks sharing a patch are co-scheduled
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  int tid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int tn_i = tid % P.tiles_n; tid /= P.tiles_n;
  const int tx_i = tid % P.tiles_x; tid /= P.tiles_x;
  const int ty_i = tid % P.tiles_y; tid /= P.tiles_y;
  const int img = tid;
  const int y0 = ty_i * QTH, x0 = tx_i * PTW, n0 = tn_i * QBN;
  const int C = P.C;
  const int qd = t & 7;                                     // this thread's channel quad of every staged pixel (256 % 8 == 0)

  // ---- patch staging: pixel index in the stored tensor per prefetch slot (fixed across chunks), -1 = outside the image / the patch
  int aoff[QLD];
#pragma unroll
  for (int l = 0; l < QLD; ++l) {
    const int pix = (t >> 3) + 32 * l;
    aoff[l] = -1;
    if (pix < QNPIX) {
      const int py = pix / PPW, px = pix - py * PPW;
      const int ly = y0 - 1 + py, lx = x0 - 1 + px;
      if (px < PTW + 2 && (unsigned)ly < (unsigned)P.H && (unsigned)lx < (unsigned)P.W) {
        const int sy = P.up ? ly >> 1 : ly, sx = P.up ? lx >> 1 : lx;
        aoff[l] = (img * P.Hs + sy) * P.Ws + sx;
      }
    }
  }
  const float ascale = NS == 4 ? (P.amax ? p_pow2_scale(*P.amax) : PASCALE) : 1.0f;
  float sat_hit = 0.f;
  const int nmain = C >> 5, nchunk = nmain + P.nx;          // virtual chunk list: main chunks (9 taps), then skip chunks (centre tap)

  // sources as buffer resources: a prefetch is ONE buffer load whose per-lane offset is (pixel * row bytes + quad * 16) -- or an offset beyond
  // the resource for padding pixels, which the hardware answers with zeros -- plus a scalar chunk offset
  const unsigned npix_in = (unsigned)P.N * P.Hs * P.Ws, npix_out = (unsigned)P.N * P.H * P.W;
  const int C1 = C - P.C0;
  const __amdgpu_buffer_rsrc_t srd_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x), 0, (int)(npix_in * (unsigned)P.C0 * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x1 ? P.x1 : P.x), 0, (int)(npix_in * (unsigned)(P.x1 ? C1 : P.C0) * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_s0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.s0 ? P.s0 : P.x), 0, (int)(P.s0 ? npix_out * (unsigned)P.Cs0 * 4u : 16u), 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_s1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.s1 ? P.s1 : P.x), 0, (int)(P.s1 ? npix_out * (unsigned)P.Cs1 * 4u : 16u), 0x00020000);

  float4 apre[QLD];                                         // prefetched patch quads of the NEXT chunk: raw fp32, later (in place) their split planes
  float4 gmu, gsc, gsh;                                     // GN: coefficients of this thread's 4 channels in the chunk being prefetched
  // per-chunk source of the prefetch in flight
  __amdgpu_buffer_rsrc_t n_srd = srd_x0;
  unsigned n_ldb = 0, n_cb = 0;
  bool n_raw = false;
  auto chunk_src = [&](int chunk) {
    n_raw = chunk >= nmain;
    if (n_raw) {
      const int c = (chunk - nmain) << 5;
      const bool first = c < P.Cs0;
      n_srd = first ? srd_s0 : srd_s1; n_ldb = (unsigned)(first ? P.Cs0 : P.Cs1) * 4u; n_cb = (unsigned)(first ? c : c - P.Cs0) * 4u;
    } else {
      const int c = chunk << 5;
      const bool first = c < P.C0;                          // C0 == C for a single source; C0 % 32 == 0 otherwise (launch check)
      n_srd = first ? srd_x0 : srd_x1; n_ldb = (unsigned)(first ? P.C0 : C1) * 4u; n_cb = (unsigned)(first ? c : c - P.C0) * 4u;
      if constexpr (GN) {
        const size_t NC = (size_t)P.N * C;
        const float* cf = P.coef + (size_t)img * C + c + qd * 4;
        gmu = *reinterpret_cast<const float4*>(cf); gsc = *reinterpret_cast<const float4*>(cf + NC); gsh = *reinterpret_cast<const float4*>(cf + 2 * NC);
      }
    }
  };
  auto gload_one = [&](int l) {
    const int po = aoff[l];
    const unsigned voff = po < 0 ? 0xFFFFFFF0u : (unsigned)po * n_ldb + (unsigned)(qd * 16);
    apre[l] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(n_srd, (int)voff, (int)n_cb, 0));
  };
  // GroupNorm / AdaGN + SiLU map (GN), fp16-window tracking and operand split of prefetch slot l, in place: apre[l] = {plane0 lo, plane0 hi, plane1 lo, plane1 hi}
  auto convert_one = [&](int l) {
    float4 v = apre[l];
    const float sc = n_raw ? 1.0f : ascale;                 // skip chunks carry the RAW residual stream: unit scale, the 2^4 sits in their weights
    if constexpr (GN) {
      // branch-free (the unit must stay one scheduling region): skip chunks and padding pixels bypass the map through selects
      const bool on = aoff[l] >= 0 && !n_raw;
      float4 m;
      m.x = gsc.x * (v.x - gmu.x) + gsh.x; m.y = gsc.y * (v.y - gmu.y) + gsh.y;
      m.z = gsc.z * (v.z - gmu.z) + gsh.z; m.w = gsc.w * (v.w - gmu.w) + gsh.w;
      m.x = p_silu(m.x); m.y = p_silu(m.y); m.z = p_silu(m.z); m.w = p_silu(m.w);      // act == 1 (launch check): a uniform branch would split the unit
      v.x = on ? m.x : v.x; v.y = on ? m.y : v.y; v.z = on ? m.z : v.z; v.w = on ? m.w : v.w;
    }
    if constexpr (NS == 4) pdae_f16_amax4(v, sc, sat_hit);
    unsigned a[NP], b[NP];
    p_split2<NS>(v.x, v.y, a, sc);
    p_split2<NS>(v.z, v.w, b, sc);
    float4 o;
    o.x = __uint_as_float(a[0]); o.y = __uint_as_float(b[0]);
    o.z = __uint_as_float(NP > 1 ? a[NP > 1 ? 1 : 0] : 0u); o.w = __uint_as_float(NP > 1 ? b[NP > 1 ? 1 : 0] : 0u);
    apre[l] = o;
  };
  auto lstore_all = [&]() {
#pragma unroll
    for (int l = 0; l < QLD; ++l) {
      const int pix = (t >> 3) + 32 * l;
      if (pix < QNPIX) {
        unsigned short* d = &sA[PSLOT(pix, qd >> 1) + (qd & 1) * 4];
        *reinterpret_cast<uint2*>(d) = make_uint2(__float_as_uint(apre[l].x), __float_as_uint(apre[l].y));
        if constexpr (NP > 1) *reinterpret_cast<uint2*>(d + PPLANE(QNPIX)) = make_uint2(__float_as_uint(apre[l].z), __float_as_uint(apre[l].w));
      }
    }
  };

  // ---- fragments
  // MFMA row i of pixel group g <-> pixel (by*8 + i/4, bx*4 + i%4), (bx, by) = (g & 3, g >> 2); the wave owns groups 8 wm .. 8 wm + 7
  const unsigned apix0 = (unsigned)(((2 * wm) * 8 + (li >> 2)) * PPW + (li & 3));
  unsigned abase0 = (unsigned)(size_t)sA + (unsigned)(PSLOT(apix0, h) * 2);
  unsigned abase1 = abase0 + (unsigned)QPLANE_B;
  asm volatile("" : "+v"(abase0));
  asm volatile("" : "+v"(abase1));
  uint4 fa[2][2][NP];                                       // [ring slot][group of the unit][plane]
  auto lda = [&](uint4 (&af)[2][NP], int tap, int kc, int u) {
    const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int a = u * 2 + j;
      const unsigned off = (unsigned)(((dy + (a >> 2) * 8) * PPW + dx + (a & 3) * 4) * (PLDH * 2) + kc * 32);
      af[j][0] = __builtin_bit_cast(uint4, *(q_lds_u4)(size_t)(abase0 + off));
      if constexpr (NP > 1) af[j][NP > 1 ? 1 : 0] = __builtin_bit_cast(uint4, *(q_lds_u4)(size_t)(abase1 + off));
    }
  };
  // weight fragments of one k-step (16 channels of one tap): one uint4 per lane, channel tile (2 per wave) and plane, straight from L2;
  //   main chunks: wp [p][chunk][tap][kc][nt][lane],  skip chunks: wps [p][chunk - nmain][kc][nt][lane]
  const int nt0 = (n0 >> 5) + wn * 2;                       // Nout % 128 == 0: both tiles exist
  const size_t plane_main = (size_t)nmain * 18 * P.NT * 512, plane_skip = (size_t)P.nx * 2 * P.NT * 512;      // bf16 elements per plane
  const __amdgpu_buffer_rsrc_t srd_main = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.wp), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_skip = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.wps ? P.wps : P.wp), 0, 0x7fffffff, 0x00020000);
  const int lane16 = lane * 16;
  uint4 qb[2][2][NP];                                       // [k-half parity][channel tile][plane]
  auto ldb = [&](uint4 (&bq)[2][NP], int chunk, int tap, int kc) {
    const bool raw = chunk >= nmain;
    const unsigned soff = (unsigned)((raw ? (((chunk - nmain) << 1) + kc) * P.NT + nt0 : (((chunk * 9 + tap) << 1) + kc) * P.NT + nt0) * 1024);   // bytes
    const unsigned ps2 = (unsigned)((raw ? plane_skip : plane_main) * 2);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int p = 0; p < NP; ++p)
        bq[ct][p] = __builtin_bit_cast(uint4, raw ? __builtin_amdgcn_raw_buffer_load_b128(srd_skip, lane16, (int)(soff + ct * 1024 + p * ps2), 0)
                                                  : __builtin_amdgcn_raw_buffer_load_b128(srd_main, lane16, (int)(soff + ct * 1024 + p * ps2), 0));
  };

  f32x16 acc[8][2];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][ct][r] = 0.f;

  // the 12 (6 / 2) MFMAs of one unit: product-major, a dependent pair is 4 issues apart
  auto mma = [&](const uint4 (&af)[2][NP], const uint4 (&bq)[2][NP], int u) {
#define PDAE_QA(P_) __builtin_bit_cast(bf16x8, af[j][P_])
#define PDAE_QB(P_) __builtin_bit_cast(bf16x8, bq[ct][P_])
#define PDAE_QAH(P_) __builtin_bit_cast(f16x8, af[j][P_])
#define PDAE_QBH(P_) __builtin_bit_cast(f16x8, bq[ct][P_])
#define PDAE_Q_EACH(STMT) _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) { STMT; }
    if constexpr (NS == 4) {                  // fp16 planes: cross terms first, leading term last (conv3x3p order)
      PDAE_Q_EACH(acc[u * 2 + j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_QAH(0), PDAE_QBH(1), acc[u * 2 + j][ct], 0, 0, 0))
      PDAE_Q_EACH(acc[u * 2 + j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_QAH(1), PDAE_QBH(0), acc[u * 2 + j][ct], 0, 0, 0))
      PDAE_Q_EACH(acc[u * 2 + j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_QAH(0), PDAE_QBH(0), acc[u * 2 + j][ct], 0, 0, 0))
    } else {
      if constexpr (NS == 2) {
        PDAE_Q_EACH(acc[u * 2 + j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_QA(0), PDAE_QB(1), acc[u * 2 + j][ct], 0, 0, 0))
        PDAE_Q_EACH(acc[u * 2 + j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_QA(1), PDAE_QB(0), acc[u * 2 + j][ct], 0, 0, 0))
      }
      PDAE_Q_EACH(acc[u * 2 + j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_QA(0), PDAE_QB(0), acc[u * 2 + j][ct], 0, 0, 0))
    }
#undef PDAE_Q_EACH
#undef PDAE_QA
#undef PDAE_QB
#undef PDAE_QAH
#undef PDAE_QBH
  };

  // ---- prologue: first chunk staged, its first weight fragments in flight
  chunk_src(0);
#pragma unroll
  for (int l = 0; l < QLD; ++l) gload_one(l);
  ldb(qb[0], 0, nmain > 0 ? 0 : 4, 0);
#pragma unroll
  for (int l = 0; l < QLD; ++l) convert_one(l);
  lstore_all();
  __syncthreads();

  // timing probes (tools/probe_build.py, WRONG RESULTS by design): pieces of the unit compiled out to see what the single wave waits for
#ifdef PDAE_Q_PROBE_NOGLOAD
#define PDAE_Q_GLOADS(U, NG)
#else
#ifdef PDAE_Q_BURST
// the vector-memory path returns data in order: a weight-fragment load (L2 hit) queued behind a patch prefetch (HBM) inherits its latency.  Bursts
// right behind the weight loads of every 4th k-step leave each weight batch at least two k-steps between the youngest older prefetch and its use
#define PDAE_Q_GLOADS(U, NG)                                                                                 \
      if ((NG) == 1) { if ((U) % 16 == 0) { _Pragma("unroll") for (int g_ = 0; g_ < 6; ++g_) if (((U) / 16) * 6 + g_ < QLD) gload_one(((U) / 16) * 6 + g_); } } \
      else { _Pragma("unroll") for (int g_ = 0; g_ < (NG); ++g_) if ((U) * (NG) + g_ < QLD) gload_one((U) * (NG) + g_); }
#else
#define PDAE_Q_GLOADS(U, NG) _Pragma("unroll") for (int g_ = 0; g_ < (NG); ++g_) if ((U) * (NG) + g_ < QLD) gload_one((U) * (NG) + g_);
#endif
#endif
#ifdef PDAE_Q_PROBE_NOCONV
#define PDAE_Q_CONVERT(U, NG)
#else
#define PDAE_Q_CONVERT(U, NG) if ((NG) == 1 && (U) >= QCV0 && (U) < QCV0 + QLD) convert_one((U) - QCV0);
#endif
#ifdef PDAE_Q_PROBE_NOA
#define PDAE_Q_LDA(TAP, I, LAST_TAP)
#else
#define PDAE_Q_LDA(TAP, I, LAST_TAP)                                                                         \
      if ((I) < 7) lda(fa[((I) + 1) & 1], TAP, ((I) + 1) >> 2, ((I) + 1) & 3);                              \
      else lda(fa[0], (LAST_TAP) ? (TAP) : (TAP) + 1, 0, 0);
#endif
#ifdef PDAE_Q_PROBE_NOB
#define PDAE_Q_LDB(TAP, LAST_TAP)
#else
#define PDAE_Q_LDB(TAP, LAST_TAP)                                                                            \
      if (u_ == 0) {                                                                                        \
        if (kc_ == 0) ldb(qb[1], chunk, TAP, 1);                                                            \
        else if (!(LAST_TAP)) ldb(qb[0], chunk, (TAP) + 1, 0);                                              \
        else ldb(qb[0], nxt, nxt_tap0, 0);                                                                  \
      }
#endif
  // one unit = the 12 MFMAs of 2 pixel groups x 2 channel tiles x 3 products; TAP / I (k-half * 4 + unit) / U (unit of the chunk) are constants
#define PDAE_Q_UNIT(TAP, I, U, LAST_TAP, NG)                                                                 \
    {                                                                                                       \
      const int kc_ = (I) >> 2, u_ = (I) & 3;                                                               \
      PDAE_Q_LDA(TAP, I, LAST_TAP)                                                                          \
      PDAE_Q_LDB(TAP, LAST_TAP)                                                                             \
      PDAE_Q_GLOADS(U, NG)                                                                                  \
      PDAE_Q_CONVERT(U, NG)                                                                                 \
      mma(fa[(I) & 1], qb[kc_], u_);                                                                        \
      PDAE_Q_PATTERN(4)                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
  // Two loops, not one loop with a branch on the chunk kind: the accumulators must be loop-carried in ONE register assignment (a join of two
  // 864-MFMA bodies made the allocator route them through VGPRs and spill)
  for (int chunk = 0; chunk < nmain; ++chunk) {             // main chunks: 9 taps x 2 k-halves x 4 units
    const bool has_next = chunk + 1 < nchunk;
    const int nxt = has_next ? chunk + 1 : chunk;           // no next chunk: the prefetch re-reads this one (branch-free body), nothing stores it
    const int nxt_tap0 = nxt >= nmain ? 4 : 0;
    chunk_src(nxt);
    lda(fa[0], 0, 0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int i = 0; i < 8; ++i) PDAE_Q_UNIT(tap, i, tap * 8 + i, tap == 8, 1)
    if (has_next) {
      __syncthreads();                                      // every wave is done with the current patch
      lstore_all();
      __syncthreads();
    }
  }
  for (int chunk = nmain; chunk < nchunk; ++chunk) {        // skip chunks: centre tap only = 8 units; three prefetches per unit, conversion at the hand-over
    const bool has_next = chunk + 1 < nchunk;
    const int nxt = has_next ? chunk + 1 : chunk;
    const int nxt_tap0 = 4;
    chunk_src(nxt);
    lda(fa[0], 4, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) PDAE_Q_UNIT(4, i, i, true, 3)
    if (has_next) {
#pragma unroll
      for (int l = 0; l < QLD; ++l) convert_one(l);
      __syncthreads();
      lstore_all();
      __syncthreads();
    }
  }
#undef PDAE_Q_UNIT
#if defined(PDAE_Q_PROBE_NOA) || defined(PDAE_Q_PROBE_NOB)
  lda(fa[1], 0, 0, 0); ldb(qb[1], 0, 0, 0);              // probes: keep every buffer defined
#endif

  // ---- epilogue: every wave transposes its sixteen 32-pixel x 32-channel accumulator tiles through a private LDS region (the patch is dead)
  // so that global traffic is float4 per lane in 128-byte runs; bias / residual / accumulate / output statistics as in conv3x3p
  const float oscale = NS == 4 ? P.woscale / ascale : 1.0f;      // exact: powers of two
  if constexpr (NS == 4) pdae_sat_report(P.sat, sat_hit);
  __syncthreads();
  float* tw = reinterpret_cast<float*>(smem) + wv * (32 * EPW);
  const int er = lane >> 3, ec = (lane & 7) * 4;
  const unsigned lane_d = (unsigned)(((er >> 2) * P.W + (er & 3)) * P.Nout);
  const unsigned lane_d2 = (unsigned)(((er & 3) >> 1) * P.Nout);          // half-resolution residual: pixel (0, (er & 3) >> 1) of the 1 x 2 sub-block
  const size_t row_pair = (size_t)2 * P.W * P.Nout;
  const bool want_stat = P.stat_part != nullptr;
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) {                          // the wave's two 8-row bands
    const int y0a = y0 + (2 * wm + bb) * 8;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int colb = n0 + wn * 64 + ct * 32 + ec;
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (P.bias) bias4 = *reinterpret_cast<const float4*>(P.bias + colb);
      if (P.bias_x) { const float4 u = *reinterpret_cast<const float4*>(P.bias_x + colb); bias4.x += u.x; bias4.y += u.y; bias4.z += u.z; bias4.w += u.w; }
      float st1 = 0.f, st2 = 0.f;
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4) {
        const int a = bb * 4 + a4, x0a = x0 + a4 * 4;
        const size_t rb = (((size_t)img * P.H + y0a) * P.W + x0a) * P.Nout + colb;
        float4 rv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) rv[it] = bias4;
        if (P.res_mode) {
          const size_t rb2 = (((size_t)img * (P.H >> 1) + (y0a >> 1)) * (P.W >> 1) + (x0a >> 1)) * P.Nout + colb;
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
        for (int r = 0; r < 16; ++r) tw[((r & 3) + 8 * (r >> 2) + 4 * h) * EPW + li] = acc[a][ct][r];
        float* dst = P.y + rb + lane_d;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          float4 v = *reinterpret_cast<const float4*>(&tw[(it * 8 + er) * EPW + ec]);
          v.x = fmaf(v.x, oscale, rv[it].x); v.y = fmaf(v.y, oscale, rv[it].y); v.z = fmaf(v.z, oscale, rv[it].z); v.w = fmaf(v.w, oscale, rv[it].w);
          *reinterpret_cast<float4*>(dst + it * row_pair) = v;
          if (want_stat) {
            st1 += (v.x + v.y) + (v.z + v.w);
            st2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, st2))));
          }
        }
      }
      if (want_stat) {      // (sum, sum of squares) of this band's 128 pixels per channel quad: the eight lanes holding a quad combine, lane er == 0 writes
        st1 += __shfl_xor(st1, 8); st2 += __shfl_xor(st2, 8);
        st1 += __shfl_xor(st1, 16); st2 += __shfl_xor(st2, 16);
        st1 += __shfl_xor(st1, 32); st2 += __shfl_xor(st2, 32);
        if (lane < 8) {
          const int wt = ((y0a >> 3) * P.tiles_x) + tx_i;   // any bijection onto the image's 8 x 16 bands: the reader sums all of them
          reinterpret_cast<float2*>(P.stat_part)[((size_t)img * P.stat_tpi + wt) * (P.Nout >> 2) + (colb >> 2)] = make_float2(st1, st2);
        }
      }
    }
  }
}

template <int NS, bool GN> static int launch_q(const PatchParams& P, hipStream_t s) {
  size_t smem = (size_t)(NPL(NS) * PPLANE(QNPIX)) * sizeof(unsigned short);
  const size_t epi = (size_t)4 * 32 * EPW * sizeof(float);
  if (smem < epi) smem = epi;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3q_kernel<NS, GN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { pdae_set_error("conv3x3q: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  dim3 grid(P.N * P.tiles_y * P.tiles_x * P.tiles_n);
  hipLaunchKernelGGL((conv3x3q_kernel<NS, GN>), grid, dim3(QTHREADS), smem, s, P);
  return pdae_launch_status("conv3x3q");
}

// PDAE_P3Q = 0 routes everything to conv3x3p (A-B aid), 2 ignores the fill heuristic (tests: small shapes through this kernel).  Read per
// launch -- a getenv costs far less than the launch -- so a test can switch between the two kernels inside one process.
static int q_mode() { const char* e = getenv("PDAE_P3Q"); return e ? atoi(e) : 0; }      // opt-in: conv3x3r supersedes it on every shape measured (profiles/r03_patch_kernels.txt)

// eligibility of a launch conv3x3p_launch has already planned WITHOUT split-K: fp16 / bf16 formats of at most two planes, 32 x 16 tiles, whole
// 128-channel output tiles, buffer-addressable sources, and enough tiles that whole rounds of 256 one-block-per-CU workgroups waste little
bool conv3x3q_ok(int math, int C, int H, int W, int N, int Nout, int Hs, int Ws, int C0, int Cs0, int Cs1) {
  if (q_mode() == 0) return false;
  if (!(math == 1 || math == 2 || math == 4)) return false;
  if ((H % QTH) || (W % PTW) || (Nout % QBN) || (C & 31)) return false;
  const unsigned long long lim = 0xFFFFFFF0ull;
  const int cmax = C0 > C - C0 ? C0 : C - C0;
  if ((unsigned long long)N * Hs * Ws * cmax * 4ull >= lim) return false;
  const int smax = Cs0 > Cs1 ? Cs0 : Cs1;
  if ((unsigned long long)N * H * W * smax * 4ull >= lim) return false;
  const long long blocks = (long long)N * (H / QTH) * (W / PTW) * (Nout / QBN);
  if (q_mode() == 2) return true;
  const long long rounds = (blocks + 255) / 256;
  return blocks >= 224 && blocks * 100 >= rounds * 256 * 85;          // >= 85 % of the last round's CUs busy
}

int conv3x3q_launch(int math, const PatchParams& P0, hipStream_t s) {
  PatchParams P = P0;
  P.tiles_x = P.W / PTW; P.tiles_y = P.H / QTH; P.tiles_n = P.Nout / QBN; P.splits = 1; P.cps = (P.C >> 5) + P.nx;
#define PDAE_Q3(NS_) (P.coef ? launch_q<NS_, true>(P, s) : launch_q<NS_, false>(P, s))
  if (math == 1) return PDAE_Q3(1);
  if (math == 2) return PDAE_Q3(2);
  return PDAE_Q3(4);
#undef PDAE_Q3
}
