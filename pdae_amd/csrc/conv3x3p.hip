// 3x3 / stride-1 / pad-1 convolution (forward and data-gradient) for gfx950 on the bf16 MFMA pipe with split fp32
// operands -- the "patch" kernel.
//
// Block = 256 threads (4 waves, each 128 pixels x 32 output channels), output tile = 8x16 pixels x 128 output channels (PTH 16: 8 waves, 16x16 pixels).
// For every 32-channel chunk of the input the 18x18-pixel halo patch is staged in LDS ONCE (as NS bf16 planes, see
// igemm.hip) and all 9 taps read their shifted A fragments straight out of it, so the input crosses L2->CU ~1.3x
// instead of 9x.  The weights are pre-split into bf16 planes in MFMA-FRAGMENT ORDER by conv3x3p_wprep (one 1 KB
// contiguous block per (chunk, tap, k-half, 32-channel tile)), so every wave loads its B fragments straight from L2 with
// fully coalesced 16-byte loads, prefetched one k-step ahead: no weight traffic through LDS and NO barrier inside a
// chunk -- the waves drift apart and hide each other's LDS / VALU phases (barriers only when the patch is restaged).
//
// Replaces F.conv2d(k=3, padding=1) of model/module.py:242,265 (+ nearest upsample :169) and its input gradient.
#include <stdlib.h>

#include "common.h"
#include "igemm.h"

#include "conv3x3p.h"

// issue pattern of one k-step: one LDS / vector-memory load of the NEXT step behind each MFMA of this one (measured +2..4 % over "all loads,
// then the 12 MFMAs"; -DPDAE_P3_CLUSTERED restores that form for tools/probe_build.py)
#define PDAE_ILV_PATTERN                                                                                   \
  _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
    __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);                                                     \
    if (i_ < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    else if (i_ < 10) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                   \
  }
template <int NS, int PTH, bool W8, bool GN = false>
__global__ void __launch_bounds__(PTH * 32, 2) conv3x3p_kernel(const PatchParams P) {      // 2 waves per SIMD: two 8x16 blocks (or one 16x16 block) per CU
  static_assert(!(GN && W8), "fused GroupNorm input is not built for the image-pair geometry");
  constexpr int PTHREADS = PTH * 32;                      // 512 | 256
  constexpr int PNPIX = (PTH + 2) * PPW;                  // 324 | 180
  constexpr int PA_LD = (PNPIX * 8 + PTHREADS - 1) / PTHREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  constexpr int SA = NPL(NS) * PPLANE(PNPIX);   // A patch planes
  unsigned short* sA = smem;

  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, h = lane >> 5;   // wave id in an SGPR: everything derived from it is scalar
  // wave tile = 128 pixels x 32 output channels: the 4 waves (PTH 8) / 2 x 4 waves (PTH 16) that share a pixel set each own ONE 32-channel
  // weight tile, so no two waves of a block load the same B fragment (the 64 x 64 wave tile made pairs of waves fetch identical fragments:
  // probe timing put ~20 % of the kernel on the weight loads through the vector-memory path; A fragments come from LDS, which has headroom)
  const int wm = wv >> 2, wn = wv & 3;

  // block -> (image, tile_y, tile_x, n-tile); n-tile fastest so the blocks sharing a patch are co-scheduled
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  int tid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int tn_i = tid % P.tiles_n; tid /= P.tiles_n;
  const int tx_i = tid % P.tiles_x; tid /= P.tiles_x;
  const int ty_i = tid % P.tiles_y; tid /= P.tiles_y;
  const int nimg = W8 ? (P.N + 1) >> 1 : P.N;
  const int img = (tid % nimg) * (W8 ? 2 : 1), sp = tid / nimg;         // W8: img = first image of the pair
  const int y0 = ty_i * PTH, x0 = tx_i * PTW, n0 = tn_i * PBN;
  const int C = P.C;

  // ---- A patch: per-thread source offsets (fixed across chunks)
  long long aoff[PA_LD];
#pragma unroll
  for (int l = 0; l < PA_LD; ++l) {
    int idx = t + PTHREADS * l;
    int pix = idx >> 3, qd = idx & 7;
    aoff[l] = -1;
    if (pix < PNPIX) {
      int py = pix / PPW, px = pix - py * PPW;
      int ly = y0 - 1 + py, lx = x0 - 1 + px, im = img;
      bool okx = px < PTW + 2;
      if constexpr (W8) { const int sub = px >= 10; im = img + sub; lx = px - 1 - 10 * sub; okx = im < P.N; }
      if (okx && (unsigned)ly < (unsigned)P.H && (unsigned)lx < (unsigned)P.W) {
        int sy = P.up ? ly >> 1 : ly, sx = P.up ? lx >> 1 : lx;
        aoff[l] = (long long)(im * P.Hs + sy) * P.Ws + sx;          // pixel index in the stored tensor(s)
      }
    }
  }
  const float ascale = NS == 4 ? (P.amax ? p_pow2_scale(*P.amax) : PASCALE) : 1.0f;
  float4 apre[PA_LD];
  float4 gmu, gsc, gsh;                          // GN: coefficients of this thread's 4 channels in the chunk being loaded
  float sat_hit = 0.f;                          // fp16 format: lanes that had a value clamped into the window (wave-uniform)
  bool pre_raw = false;                          // the registers hold a skip chunk (raw input: no GroupNorm map)
  const int nmain = C >> 5, nchunk = nmain + P.nx;            // virtual chunk list: main chunks (9 taps), then skip chunks (centre tap)
  auto a_gload = [&](int chunk) {
    const int qd = t & 7;
    const float* src; int ld, cc;
    pre_raw = chunk >= nmain;
    if (pre_raw) {                               // skip chunk: raw [s0 | s1] at the output resolution
      const int c = ((chunk - nmain) << 5) + qd * 4;
      const bool first = c < P.Cs0;
      src = first ? P.s0 : P.s1; ld = first ? P.Cs0 : P.Cs1; cc = first ? c : c - P.Cs0;
    } else {
      const int c = (chunk << 5) + qd * 4;
      const bool first = c < P.C0;               // C0 == C for a single source
      src = first ? P.x : P.x1; ld = first ? P.C0 : C - P.C0; cc = first ? c : c - P.C0;
      if constexpr (GN) {
        const size_t NC = (size_t)P.N * C;
        const float* cf = P.coef + (size_t)img * C + c;
        gmu = *reinterpret_cast<const float4*>(cf); gsc = *reinterpret_cast<const float4*>(cf + NC); gsh = *reinterpret_cast<const float4*>(cf + 2 * NC);
      }
    }
    // unconditional loads from a clamped address, zeroed afterwards where the patch pixel lies outside the image (see ldb)
#pragma unroll
    for (int l = 0; l < PA_LD; ++l) {
      const float4 v = *reinterpret_cast<const float4*>(src + (aoff[l] < 0 ? 0ll : aoff[l]) * ld + cc);
      apre[l] = aoff[l] >= 0 ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto a_lstore = [&]() {
#pragma unroll
    for (int l = 0; l < PA_LD; ++l) {
      int idx = t + PTHREADS * l;
      int pix = idx >> 3, qd = idx & 7;
      if (pix < PNPIX) {
        if constexpr (GN) {
          if (aoff[l] >= 0 && !pre_raw) {
            float4 v = apre[l];
            v.x = gsc.x * (v.x - gmu.x) + gsh.x; v.y = gsc.y * (v.y - gmu.y) + gsh.y;
            v.z = gsc.z * (v.z - gmu.z) + gsh.z; v.w = gsc.w * (v.w - gmu.w) + gsh.w;
            if (P.act) { v.x = p_silu(v.x); v.y = p_silu(v.y); v.z = p_silu(v.z); v.w = p_silu(v.w); }
            apre[l] = v;
          }
        }
        if constexpr (NS == 4) {                // fp16 format: exact power-of-two pre-scale, window [2^-7, 4094] keeps both planes normal
          // a value outside the window (|x| > 3750 cannot occur behind GroupNorm unless the network has diverged) overflows to Inf / NaN
          // like any fp16 overflow and is counted: the step is then discarded and re-run in the range-free bf16x6 split (common.h)
          // skip chunks carry the RAW residual stream (no GroupNorm in front): unit activation scale, the 2^4 sits in their weights instead
          pdae_f16_amax4(apre[l], pre_raw ? 1.0f : ascale, sat_hit);
        }
        unsigned a[NPL(NS)], b[NPL(NS)];
        p_split2<NS>(apre[l].x, apre[l].y, a, pre_raw ? 1.0f : ascale);
        p_split2<NS>(apre[l].z, apre[l].w, b, pre_raw ? 1.0f : ascale);
#pragma unroll
        for (int p = 0; p < NPL(NS); ++p) *reinterpret_cast<uint2*>(&sA[p * PPLANE(PNPIX) + PSLOT(pix, qd >> 1) + (qd & 1) * 4]) = make_uint2(a[p], b[p]);
      }
    }
  };

  // MFMA row i of 32-row group g = wm*4+a <-> pixel (by*8 + i/4, bx*4 + i%4), (bx, by) = (g & 3, g >> 2): with the 20-pixel
  // pitch and 80-byte rows the 16 lanes of every ds_read_b128 group hit 16 distinct 16-byte slots for all 9 tap shifts
  int apix[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int g = wm * 4 + a, bx = g & 3;
    apix[a] = ((g >> 2) * 8 + (li >> 2)) * PPW + (W8 ? (bx >> 1) * 10 + (bx & 1) * 4 : bx * 4) + (li & 3);
  }

  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  const int c_begin = sp * P.cps, c_end = min(nchunk, c_begin + P.cps);
  // B fragments of one k-step (16 channels of one tap): one uint4 (8 bf16) per lane, 32-channel tile and plane;
  //   main chunks: wp [p][chunk][tap][kc][nt][lane],  skip chunks: wps [p][chunk - nmain][kc][nt][lane]
  const int nt0 = min((n0 >> 5) + wn, P.NT - 1);
  const size_t plane_main = (size_t)nmain * 18 * P.NT * 512, plane_skip = (size_t)P.nx * 2 * P.NT * 512;      // bf16 elements per plane
  // Buffer loads: descriptor (SGPRs) + ONE constant per-lane byte offset (lane * 16) + a scalar byte offset per step and plane, so a weight
  // fragment load needs no vector arithmetic at all (the flat form cost two 64-bit v_lshl_add per load, most of what was left of the loop's VALU)
  typedef unsigned pw_u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t srd_main = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.wp), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_skip = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.wps ? P.wps : P.wp), 0, 0x7fffffff, 0x00020000);
  const int lane16 = lane * 16;
  auto ldb = [&](uint4 (&bq)[NPL(NS)], int chunk, int tap, int kc) {
    const bool raw = chunk >= nmain;
    // branch-free: a tile index beyond the last one is clamped onto it (those output columns are masked in the epilogue), so the loads are
    // unconditional straight-line code and the compiler can wait for them with a COUNTED vmcnt instead of draining everything in flight
    const unsigned soff = (unsigned)((raw ? (((chunk - nmain) << 1) + kc) * P.NT + nt0 : (((chunk * 9 + tap) << 1) + kc) * P.NT + nt0) * 1024);   // bytes
    const unsigned ps2 = (unsigned)((raw ? plane_skip : plane_main) * 2);
#pragma unroll
    for (int p = 0; p < NPL(NS); ++p)
      bq[p] = __builtin_bit_cast(uint4, raw ? __builtin_amdgcn_raw_buffer_load_b128(srd_skip, lane16, (int)(soff + p * ps2), 0)
                                            : __builtin_amdgcn_raw_buffer_load_b128(srd_main, lane16, (int)(soff + p * ps2), 0));
  };
  // A fragments of one k-step (16 channels = half kc of the staged chunk, one tap): 4 pixel groups x planes, 16 bytes per lane each.  All eight
  // reads share ONE address register -- group 0 / plane 0 of this tap and k-half, made opaque so that hipcc keeps it as the base -- and reach
  // their slot through the instruction's immediate offset: the pixel groups of a wave and the planes lie at compile-time distances.  (Left to
  // itself the compiler picked a base in the middle and formed the other seven addresses with a v_add_u32 each, 8 VALU instructions per 12 MFMAs.)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) const u32x4* lds_u4;
  const unsigned a_lane = (unsigned)(size_t)sA + (unsigned)(PSLOT(apix[0], h) * 2);
  auto lda = [&](uint4 (&af)[4][NPL(NS)], int tap, int kc) {
    const int dy = tap / 3, dx = tap - dy * 3;
    unsigned ab = a_lane + (unsigned)((dy * PPW + dx) * (PLDH * 2) + kc * 32);
    asm volatile("" : "+v"(ab));
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int dpix = W8 ? (a >> 1) * 10 + (a & 1) * 4 : a * 4;             // apix[a] - apix[0]
#pragma unroll
      for (int p = 0; p < NPL(NS); ++p) af[a][p] = __builtin_bit_cast(uint4, *(lds_u4)(size_t)(ab + (unsigned)((p * PPLANE(PNPIX) + dpix * PLDH) * 2)));
    }
  };
  // the MFMAs of one k-step.  Product-major order: consecutive MFMAs target different accumulators (a dependent pair is 4 issues apart)
  auto mma = [&](const uint4 (&af)[4][NPL(NS)], const uint4 (&bq)[NPL(NS)]) {
#define PDAE_A(P_) __builtin_bit_cast(bf16x8, af[a][P_])
#define PDAE_B(P_) __builtin_bit_cast(bf16x8, bq[P_])
#define PDAE_AH(P_) __builtin_bit_cast(f16x8, af[a][P_])
#define PDAE_BH(P_) __builtin_bit_cast(f16x8, bq[P_])
#define PDAE_EACH_ACC(STMT) _Pragma("unroll") for (int a = 0; a < 4; ++a) { STMT; }
    if constexpr (NS == 4) {                  // fp16 planes: cross terms first, leading term last
      PDAE_EACH_ACC(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_AH(0), PDAE_BH(1), acc[a], 0, 0, 0))
      PDAE_EACH_ACC(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_AH(1), PDAE_BH(0), acc[a], 0, 0, 0))
      PDAE_EACH_ACC(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_AH(0), PDAE_BH(0), acc[a], 0, 0, 0))
    } else {
      if constexpr (NS == 3) {
        PDAE_EACH_ACC(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_A(1), PDAE_B(1), acc[a], 0, 0, 0))
        PDAE_EACH_ACC(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_A(0), PDAE_B(2), acc[a], 0, 0, 0))
        PDAE_EACH_ACC(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_A(2), PDAE_B(0), acc[a], 0, 0, 0))
      }
      if constexpr (NS >= 2) {
        PDAE_EACH_ACC(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_A(0), PDAE_B(1), acc[a], 0, 0, 0))
        PDAE_EACH_ACC(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_A(1), PDAE_B(0), acc[a], 0, 0, 0))
      }
      PDAE_EACH_ACC(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_A(0), PDAE_B(0), acc[a], 0, 0, 0))
    }
#undef PDAE_EACH_ACC
#undef PDAE_A
#undef PDAE_AH
#undef PDAE_BH
#undef PDAE_B
  };
  // Software pipeline, one k-step deep for BOTH operands: while the 12 (6 / 3 / 1 x 4) MFMAs of step i run, the weight fragments (L2, ~200-500
  // cycles) and the patch fragments (LDS, ~64+ cycles) of step i+1 are already in flight.  sched_barrier pins "issue the loads, THEN the MFMA
  // cluster": left alone the scheduler sinks the loads to the end of the cluster and the next cluster starts with an exposed vmcnt / lgkmcnt wait.
  uint4 q0[NPL(NS)], q1[NPL(NS)], f0[4][NPL(NS)], f1[4][NPL(NS)];
  if (c_begin < c_end) {
    a_gload(c_begin);
    ldb(q0, c_begin, c_begin >= nmain ? 4 : 0, 0);
    a_lstore();
    __syncthreads();
    if (c_begin + 1 < c_end) a_gload(c_begin + 1);
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
      const bool raw = chunk >= nmain;
      const int t_lo = raw ? 4 : 0, t_hi = raw ? 5 : 9;                   // skip chunks: centre tap only
      lda(f0, t_lo, 0);
      for (int tap = t_lo; tap < t_hi; ++tap) {
#ifndef PDAE_PROBE_NOB
        ldb(q1, chunk, tap, 1);
#endif
#ifndef PDAE_PROBE_NOA
        lda(f1, tap, 1);
#endif
#ifndef PDAE_P3_CLUSTERED
        mma(f0, q0);
        PDAE_ILV_PATTERN
#else
        __builtin_amdgcn_sched_barrier(0);
        mma(f0, q0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        {   // next k-step, ALWAYS issued (unconditional straight-line loads keep the vmcnt / lgkmcnt waits counted): the next tap of this
            // chunk, else the first tap of the next chunk (its patch fragments are re-read after the hand-over below), else a harmless
            // repeat of the current step at the very end
          const bool more = tap + 1 < t_hi, nextc = !more && chunk + 1 < c_end;
          const int nchunk_i = nextc ? chunk + 1 : chunk;
          const int ntap = more ? tap + 1 : (nextc ? (chunk + 1 >= nmain ? 4 : 0) : tap);
#ifndef PDAE_PROBE_NOB
          ldb(q0, nchunk_i, ntap, 0);
#endif
#ifndef PDAE_PROBE_NOA
          lda(f0, ntap, 0);
#endif
        }
#ifndef PDAE_P3_CLUSTERED
        mma(f1, q1);
        PDAE_ILV_PATTERN
#else
        __builtin_amdgcn_sched_barrier(0);
        mma(f1, q1);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      // patch hand-over at the end of a chunk: the only barriers of the kernel
#ifndef PDAE_PROBE_NOSTAGE
      // (a second LDS patch buffer filled in the middle of the chunk -- one barrier per chunk -- was measured: no gain, 2x LDS, +30 VGPRs)
      if (chunk + 1 < c_end) {
        __syncthreads();                          // every wave is done with the current patch
        a_lstore();
        __syncthreads();
        if (chunk + 2 < c_end) a_gload(chunk + 2);
      }
#endif
    }
  }
#if defined(PDAE_PROBE_NOB) || defined(PDAE_PROBE_NOA)
  lda(f1, 4, 1); ldb(q1, c_begin, 0, 1);        // probes: keep every buffer defined
#endif

  // ---- epilogue: every wave transposes its four 32-pixel x 32-channel accumulator groups through a private LDS region (the patch is
  // dead by now) so that global traffic is float4 per lane, 8 lanes per pixel row: 128-byte contiguous runs (one cache line), 4x fewer
  // store instructions than storing the MFMA layout directly (the dword-per-lane form is store-issue bound)
  const long long Mtot = (long long)P.N * P.H * P.W;
  const float oscale = NS == 4 ? P.woscale / ascale : 1.0f;      // exact: powers of two
  if constexpr (NS == 4) pdae_sat_report(P.sat, sat_hit);
  __syncthreads();                                        // all waves are done reading the patch
  float* tw = reinterpret_cast<float*>(smem) + wv * (32 * EPW);
  const int er = lane >> 3, ec = (lane & 7) * 4;          // read side: row within a group of 8, first of 4 channels
  const int colb = n0 + wn * 32 + ec;
  const int colc = colb < P.Nout ? colb : 0;              // clamped column for the operand loads of lanes that store nothing
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (P.splits == 1 && colb < P.Nout) {          // Nout % 4 == 0: the four columns are valid together
    if (P.bias) bias4 = *reinterpret_cast<const float4*>(P.bias + colb);
    if (P.bias_x) { const float4 u = *reinterpret_cast<const float4*>(P.bias_x + colb); bias4.x += u.x; bias4.y += u.y; bias4.z += u.z; bias4.w += u.w; }
  }
  // Addresses: a wave-uniform 64-bit base per (pixel group a, row pair it) -- scalar arithmetic -- plus ONE 32-bit per-lane offset for the whole
  // epilogue (pixel (er >> 2, er & 3) of the 2 x 4 sub-block, channel quad ec).  The per-row 64-bit index / bounds arithmetic this replaces was
  // ~800 VALU instructions per wave, as much as five chunks of the main loop.  Tiles are full in x (W % 16 == 0 or the image-pair form) and
  // whole 8-row groups in y, so validity is the lane's column test plus a uniform test per pixel group.
  const bool col_ok = colb < P.Nout;
  const unsigned lane_d = (unsigned)(((er >> 2) * P.W + (er & 3)) * P.Nout + colc);
  const unsigned lane_d2 = (unsigned)(((er & 3) >> 1) * P.Nout + colc);          // half-resolution residual: pixel (0, (er & 3) >> 1) of the 1 x 2 sub-block
  const size_t row_pair = (size_t)2 * P.W * P.Nout;                                // two output rows
  const bool want_stat = P.stat_part != nullptr && P.splits == 1;                                  // wave-uniform
  float st1 = 0.f, st2 = 0.f;                                                      // this lane's share of the wave's output statistics
  // the eight lanes that hold the same channel quad (er = lane >> 3) combine; lane er == 0 writes.  Wave-tile index inside its image:
  // ((ty, tx), 128-pixel half wm); the image-pair form flushes after each image of the pair.
  auto stat_flush = [&](int im) {
    st1 += __shfl_xor(st1, 8); st2 += __shfl_xor(st2, 8);
    st1 += __shfl_xor(st1, 16); st2 += __shfl_xor(st2, 16);
    st1 += __shfl_xor(st1, 32); st2 += __shfl_xor(st2, 32);
    if (lane < 8 && col_ok && im < P.N) {
      const int wt = (ty_i * P.tiles_x + tx_i) * (PTH / 8) + wm;
      reinterpret_cast<float2*>(P.stat_part)[((size_t)im * P.stat_tpi + wt) * (P.Nout >> 2) + (colb >> 2)] = make_float2(st1, st2);
    }
    st1 = st2 = 0.f;
  };
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int im_a = W8 ? img + (a >> 1) : img, x0a = W8 ? (a & 1) * 4 : x0 + a * 4, y0a = y0 + wm * 8;
    const bool grp_ok = y0a < P.H && im_a < P.N;                                   // wave-uniform
    const size_t rb = (((size_t)im_a * P.H + y0a) * P.W + x0a) * P.Nout;           // element offset of the group's first pixel
    float4 rv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) rv[it] = bias4;
    // residual / accumulate operands: all loads are issued up front (clamped column for lanes that store nothing) so that their latency runs
    // under the LDS transposition instead of serialising the stores
    if (grp_ok && P.splits == 1 && P.res_mode) {
      const size_t rb2 = (((size_t)im_a * (P.H >> 1) + (y0a >> 1)) * (P.W >> 1) + (x0a >> 1)) * P.Nout;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float* src = P.res_mode == 2 ? P.res + rb2 + (size_t)it * (P.W >> 1) * P.Nout + lane_d2 : P.res + rb + it * row_pair + lane_d;
        const float4 u = *reinterpret_cast<const float4*>(src);
        rv[it].x += u.x; rv[it].y += u.y; rv[it].z += u.z; rv[it].w += u.w;
      }
    }
    if (grp_ok && P.splits == 1 && P.accumulate) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float4 u = *reinterpret_cast<const float4*>(P.y + rb + it * row_pair + lane_d);
        rv[it].x += u.x; rv[it].y += u.y; rv[it].z += u.z; rv[it].w += u.w;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tw[((r & 3) + 8 * (r >> 2) + 4 * h) * EPW + li] = acc[a][r];
    if (!grp_ok) { if (W8 && want_stat && a == 1) stat_flush(img); continue; }
    float* dst = P.splits > 1 ? P.slab + (size_t)sp * Mtot * P.Nout + rb + lane_d : P.y + rb + lane_d;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float4 v = *reinterpret_cast<const float4*>(&tw[(it * 8 + er) * EPW + ec]);
      // the operand scales are powers of two: scaling after the transpose, fused with the bias / residual add, is exact
      if (P.splits > 1) { v.x *= oscale; v.y *= oscale; v.z *= oscale; v.w *= oscale; }
      else { v.x = fmaf(v.x, oscale, rv[it].x); v.y = fmaf(v.y, oscale, rv[it].y); v.z = fmaf(v.z, oscale, rv[it].z); v.w = fmaf(v.w, oscale, rv[it].w); }
      if (col_ok) *reinterpret_cast<float4*>(dst + it * row_pair) = v;
      if (want_stat) {
        st1 += (v.x + v.y) + (v.z + v.w);
        st2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, st2))));
      }
    }
    if (W8 && want_stat && a == 1) stat_flush(img);
  }
  if (want_stat) stat_flush(W8 ? img + 1 : img);
}

// y = sum of the split slabs (fixed order) + bias + residual (+ y)
// one output float4 of a split-K launch: slabs summed in slab order, then the epilogue of the unsplit kernel (bias, residual, accumulate)
__device__ __forceinline__ float4 reduce_one(const PatchParams& P, long long Mtot, long long row, int col) {
  float4 v = *reinterpret_cast<const float4*>(P.slab + row * P.Nout + col);
  const float* sl = P.slab + row * P.Nout + col; const long long ss = Mtot * P.Nout;
  int k = 1;                                  // four slabs in flight, added in slab order (see splitk_reduce_kernel)
  for (; k + 4 <= P.splits; k += 4) {
    const float4 u0 = *reinterpret_cast<const float4*>(sl + k * ss), u1 = *reinterpret_cast<const float4*>(sl + (k + 1) * ss);
    const float4 u2 = *reinterpret_cast<const float4*>(sl + (k + 2) * ss), u3 = *reinterpret_cast<const float4*>(sl + (k + 3) * ss);
    v.x += u0.x; v.y += u0.y; v.z += u0.z; v.w += u0.w;
    v.x += u1.x; v.y += u1.y; v.z += u1.z; v.w += u1.w;
    v.x += u2.x; v.y += u2.y; v.z += u2.z; v.w += u2.w;
    v.x += u3.x; v.y += u3.y; v.z += u3.z; v.w += u3.w;
  }
  for (; k < P.splits; ++k) {
    const float4 u = *reinterpret_cast<const float4*>(sl + k * ss);
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  if (P.bias) { const float4 u = *reinterpret_cast<const float4*>(P.bias + col); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
  if (P.bias_x) { const float4 u = *reinterpret_cast<const float4*>(P.bias_x + col); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
  if (P.res_mode) {
    long long rrow = row;
    if (P.res_mode == 2) {
      const int ox = (int)(row % P.W); const long long t2 = row / P.W; const int oy = (int)(t2 % P.H); const long long im = t2 / P.H;
      rrow = (im * (P.H >> 1) + (oy >> 1)) * (P.W >> 1) + (ox >> 1);
    }
    const float4 u = *reinterpret_cast<const float4*>(P.res + rrow * P.Nout + col); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  float4* dst = reinterpret_cast<float4*>(P.y + row * P.Nout + col);
  if (P.accumulate) { const float4 u = *dst; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
  *dst = v;
  return v;
}

__global__ void __launch_bounds__(256) conv3x3p_reduce_kernel(const PatchParams P) {
  const int n4 = P.Nout >> 2;
  const long long Mtot = (long long)P.N * P.H * P.W, total = Mtot * n4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long row = i / n4; const int col = (int)(i - row * n4) * 4;
    reduce_one(P, Mtot, row, col);
  }
}

// The same reduction, leaving the GroupNorm partial statistics of the stored tensor behind like the unsplit kernel does: block = (image, run of
// tp pixels), threads = channel quads x pixel lanes; stat_part[image][stat_tpi runs][Nout / 4] x (sum, sum of squares) over tp pixels x 4
// channels, unshifted fp32 (at most 64 values each), combined in fp64 by pdae_gn_coef_from_conv_stats.  The split layers are the small ones
// (8^2 .. 32^2): a statistics pass over their output cost a launch of its own plus a finalize launch -- 10 + 5 us, ~70 times per step.
__global__ void __launch_bounds__(256) conv3x3p_reduce_stats_kernel(const PatchParams P, int tp) {
  __shared__ float2 red[256];
  const int NQ = P.Nout >> 2, PL = 256 / NQ, HW = P.H * P.W, t = threadIdx.x, q = t % NQ, pl = t / NQ;
  const int n = blockIdx.x / P.stat_tpi, run = blockIdx.x - n * P.stat_tpi;
  const long long Mtot = (long long)P.N * HW;
  const int p1 = min(HW, (run + 1) * tp);
  float s1 = 0.f, s2 = 0.f;
  if (pl < PL) {
    for (int p = run * tp + pl; p < p1; p += PL) {
      const float4 v = reduce_one(P, Mtot, (long long)n * HW + p, q * 4);
      s1 += (v.x + v.y) + (v.z + v.w);
      s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    red[pl * NQ + q] = make_float2(s1, s2);
  }
  __syncthreads();
  if (t < NQ) {
    float a = 0.f, b = 0.f;
    for (int l = 0; l < PL; ++l) { const float2 v = red[l * NQ + t]; a += v.x; b += v.y; }
    reinterpret_cast<float2*>(P.stat_part)[((size_t)n * P.stat_tpi + run) * NQ + t] = make_float2(a, b);
  }
}

// pixels per block of conv3x3p_reduce_stats_kernel: 16, or 8 when 16 would leave the chip half empty
static int reduce_stats_tp(int N, int HW) { return (long long)N * ((HW + 15) / 16) >= 512 ? 16 : 8; }

template <int NS, int PTH, bool W8, bool GN = false> static int launch_ns(const PatchParams& P, hipStream_t s) {
  constexpr int NPIX = (PTH + 2) * PPW;
  size_t smem = (size_t)(NPL(NS) * PPLANE(NPIX)) * sizeof(unsigned short);
  const size_t epi = (size_t)(PTH / 2) * 32 * EPW * sizeof(float);        // one 32 x 36 fp32 tile per wave
  if (smem < epi) smem = epi;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3p_kernel<NS, PTH, W8, GN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { pdae_set_error("conv3x3p: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  const int nimg = W8 ? (P.N + 1) / 2 : P.N;
  dim3 grid(nimg * P.tiles_y * P.tiles_x * P.tiles_n * P.splits);
  hipLaunchKernelGGL((conv3x3p_kernel<NS, PTH, W8, GN>), grid, dim3(PTH * 32), smem, s, P);
  if (P.splits > 1) {
    const long long total = (long long)P.N * P.H * P.W * (P.Nout >> 2);
    long long nb = (total + 255) / 256; if (nb > 4096) nb = 4096;
    if (P.stat_part) hipLaunchKernelGGL(conv3x3p_reduce_stats_kernel, dim3(P.N * P.stat_tpi), dim3(256), 0, s, P, reduce_stats_tp(P.N, P.H * P.W));
    else hipLaunchKernelGGL(conv3x3p_reduce_kernel, dim3((int)nb), dim3(256), 0, s, P);
  }
  return pdae_launch_status("conv3x3p");
}

static int patch_th() {                        // PDAE_P3_TH = 8 | 16 overrides the tile height (tuning aid)
  return pdae_knob(KNOB_P3_TH);
}

// launch plan of one convolution: tile geometry and split-K factor, a pure function of the shape (shared by the workspace
// query and the launch)
struct PatchPlan { int th, w8, tiles_x, tiles_y, tiles_n, splits, cps; long long blocks; };
static PatchPlan patch_plan(int C, int H, int W, int N, int Nout) {
  PatchPlan q;
  q.w8 = W == 8;
  q.th = patch_th();
  if (q.th != 8 && q.th != 16) q.th = 8;       // 8x16 tiles: two independent 4-wave blocks per CU overlap each other's staging phases
  if (q.w8 || (H % q.th)) q.th = 8;
  q.tiles_x = q.w8 ? 1 : W / PTW; q.tiles_y = H / q.th; q.tiles_n = (Nout + PBN - 1) / PBN;
  const long long base = (long long)(q.w8 ? (N + 1) / 2 : N) * q.tiles_x * q.tiles_y * q.tiles_n;
  const int nchunk = C >> 5, slots = q.th == 8 ? 512 : 256;            // resident blocks of a full chip
  // split-K over chunk ranges: minimise rounds(grid) x chunks-per-block (+1 chunk-time of prologue / epilogue per block)
  long long best = -1; int best_s = 1;
  for (int sN = 1; sN <= nchunk && sN <= 16; ++sN) {
    const int cps = (nchunk + sN - 1) / sN, sp = (nchunk + cps - 1) / cps;
    if (sp != sN) continue;
    const long long rounds = (base * sp + slots - 1) / slots;
    const long long cost = rounds * (cps + 1);      // (a slab-traffic term like conv1x1's point_plan was measured: +0.6 ... +3.6 ms on the step)
    if (best < 0 || cost < best) { best = cost; best_s = sN; }
  }
  q.cps = (nchunk + best_s - 1) / best_s;
  q.splits = (nchunk + q.cps - 1) / q.cps;
  q.blocks = base * q.splits;
  return q;
}

// fused 1x1 skip convolution: the main conv must be eligible without the image-pair geometry / upsample and the combined K loop must not
// need split-K (its slabs are sized for the main convolution alone)
bool conv3x3p_skip_ok(int math, int C, int H, int W, int N, int Nout, int up, int Cs0, int Cs1) {
  if (conv3x3p_form(math, C, H, W, N, Nout)) return false;      // Winograd-along-x form of the main convolution: no skip chunks there (math may carry the direct bit)
  math &= ~PDAE_MATH_DIRECT_BIT;
  if (math < 1 || up || (C & 31) || (Cs0 & 31) || (Cs1 & 31) || (H % 8) || (W % PTW) || (Nout & 3) || Nout < 32) return false;
  const PatchPlan q = patch_plan(C + Cs0 + Cs1, H, W, N, Nout);
  return q.blocks / q.splits >= 256;              // enough tiles without split-K (the fused launch never splits)
}

// eligibility: 3x3, stride 1, pad 1, one source, channels % 32, spatial tile-aligned (W % 16, or W == 8 for image pairs);
// fill: also enough blocks to occupy the chip (fill = false: shape eligibility only)
bool conv3x3p_ok(int math, int KH, int KW, int stride, int pad, int C1, int C, int H, int W, int N, int Nout, bool fill) {
  if (math < 1 || KH != 3 || KW != 3 || stride != 1 || pad != 1 || C1 != 0) return false;
  if ((C & 31) || (H % 8) || ((W % PTW) && W != 8) || (Nout & 3)) return false;
  if (!fill) return true;
  return Nout >= 32 && patch_plan(C, H, W, N, Nout).blocks >= 256;
}

static size_t slab_bytes(int C, int H, int W, int N, int Nout) {
  const PatchPlan q = patch_plan(C, H, W, N, Nout);
  return q.splits > 1 ? (size_t)q.splits * N * H * W * Nout * sizeof(float) : 0;
}

// k-halves per 32-channel chunk: 2 x 12 transform taps in the Winograd-along-x layout, 2 x 9 in the direct one (ABI 7 / 8 reserved 24 for every
// copy: +33 % on every direct-form layer and data-gradient copy).  `form` = conv3x3p_form of the same arguments: size query, preparation and
// launch all derive it from the shape, so they agree on where the split-K slabs start (a Winograd-form launch never has slabs).
static size_t prep_bytes(int math, int Nout, int C, int form) {
  const int NS = math < 1 ? 1 : (math == 4 ? 2 : (math > 3 ? 3 : math));       // planes: math 4 = two fp16 planes
  const size_t b = (size_t)NS * (C >> 5) * (form ? 24 : 18) * ((Nout + 31) / 32) * 512 * sizeof(unsigned short);
  return (b + 255) & ~(size_t)255;
}

// ---- form tags of prepared 3x3 weights.  A prepared copy is written either in the direct layout (9 taps) or in the Winograd-along-x layout (12
// transform taps); a launch in the other form would read garbage and still return 0.  Every preparation (pdae_conv_wprep, or the description of
// a job of the grouped launch) therefore records the layout it writes under the copy's address, and every launch that takes a copy checks it: a
// mismatch -- PDAE_W1 changed through pdae_set_knob, or PDAE_MATH_DIRECT present on one side only -- fails with PDAE_EINVAL instead of computing
// wrong numbers.  Host-side only (no device traffic, no synchronisation: a tag INSIDE the buffer could only be checked by a device-to-host copy on
// the launch path or asynchronously by the kernel).  Round 6: the tag is signed with the convolution it was written for
// (arithmetic, Nout, C) -- a buffer that was freed and re-allocated for ANOTHER convolution no longer inherits a stale tag; only a tag of the same
// convolution can refuse a launch -- and the table is bounded (oldest half dropped beyond 64 K entries).  Copies of unknown provenance pass.
#include <mutex>
#include <unordered_map>
struct FormTag { int form; unsigned sig; unsigned long long age; };
static std::mutex g_form_mu;
static std::unordered_map<const void*, FormTag> g_form_tag;
static unsigned long long g_form_age = 0;
static unsigned form_sig(int math, int Nout, int C) { return ((unsigned)math * 2654435761u) ^ ((unsigned)Nout * 40503u) ^ ((unsigned)C << 16) ^ 0x9e3779b9u; }
static void form_note(const void* wp, int form, unsigned sig) {
  std::lock_guard<std::mutex> lk(g_form_mu);
  if (g_form_tag.size() > 65536) {
    const unsigned long long cut = g_form_age - 32768;
    for (auto it = g_form_tag.begin(); it != g_form_tag.end();) it = it->second.age < cut ? g_form_tag.erase(it) : ++it;
  }
  g_form_tag[wp] = FormTag{form, sig, g_form_age++};
}
static int form_check(const void* wp, int form, unsigned sig) {
  std::lock_guard<std::mutex> lk(g_form_mu);
  auto it = g_form_tag.find(wp);
  if (it == g_form_tag.end() || it->second.sig != sig || it->second.form == form) return PDAE_OK;
  pdae_set_error("conv3x3p: the prepared weights at %p were written in the %s layout but this launch takes the %s form (PDAE_W1 changed between "
                 "pdae_conv_wprep and the launch, or PDAE_MATH_DIRECT is set on one of the two descriptors only)", wp,
                 it->second.form ? "Winograd-along-x" : "direct", form ? "Winograd-along-x" : "direct");
  return PDAE_EINVAL;
}

// 1: this convolution (launch-side dimensions) is prepared and launched in the Winograd-along-x form (conv3x3x.hip); 0: direct patch kernels.
// A pure function of the shape and of PDAE_W1 -- weight preparation and launch call it with the same arguments (fused skip chunks follow
// A launch the direct plan would split over K keeps the direct form.  Fused 1x1 skip chunks exist in the direct form only: conv3x3p_skip_ok
// says no where this function says 1, and the caller computes the skip convolution separately (it then enters as the residual).
int conv3x3p_form(int math, int C, int H, int W, int N, int Nout) {
  if (math & PDAE_MATH_DIRECT_BIT) return 0;              // the caller pinned the direct form (pdae_conv_desc.math | PDAE_MATH_DIRECT)
  if (!conv3x3x_ok(math, C, H, W, N, Nout)) return 0;
  // PDAE_W1 = 2 (tests of the 16-row kernel on small shapes): a launch the direct plan would split over K keeps the direct form, as in round 4
  if (pdae_knob(KNOB_W1) == 2) return patch_plan(C, H, W, N, Nout).splits == 1 ? 1 : 0;
  return 1;
}

// Tiles per image of the GroupNorm-backward partial sums a data gradient with these LAUNCH-side dimensions (C = dY channels, Nout = the GroupNorm's
// channels = C0 + C1) can leave in its epilogue: only the Winograd-form launches (conv3x3y) build it; at most 64 tiles per image (the finalize
// kernel's workspace, k_gn_workspace_floats); both sources whole 32-channel runs.  0: not available.
int conv3x3p_gnb_tiles(int math, int C, int H, int W, int N, int Nout, int C0, int C1) {
  if (C0 + C1 != Nout || (C0 & 31) || (C1 & 31) || !conv3x3p_form(math, C, H, W, N, Nout)) return 0;
  const int rows = conv3x3x_rows(math & ~PDAE_MATH_DIRECT_BIT, C, H, W, N, Nout);
  const int t = (H / (8 * rows)) * (W / 16);
  return t <= 64 ? t : 0;
}

// power-of-two scale of the fp16-format prepared weights: trained conv weights are ~ 1/sqrt(fan_in), which would put their low plane
// into the fp16 subnormals; scaled to O(1) both planes keep full precision, the kernel multiplies the accumulators by the inverse (exact)
float conv3x3p_wscale(int C) { int k = 0; while ((1 << (2 * k)) < 9 * C) ++k; return (float)(1 << k); }

// prepared weights + split-K slabs of the convolution (one buffer: [planes | slabs])
// (math may carry the direct bit, exactly as for the preparation and the launch: same arguments, same layout, same size)
size_t conv3x3p_wprep_bytes(int math, int Nout, int C, int H, int W, int N) {
  const int form = conv3x3p_form(math, C, H, W, N, Nout);
  return prep_bytes(math & ~PDAE_MATH_DIRECT_BIT, Nout, C, form) + (form ? 0 : slab_bytes(C, H, W, N, Nout));      // (the Winograd form never splits over K)
}

// One-shot request (pdae_conv_stats_arm): the next forward convolution entry point on this host thread TAKES it -- on entry, before any argument
// check, so that no return path leaves it armed for a later launch on another tensor -- and hands it to conv3x3p_launch explicitly, which
// then also writes the GroupNorm partial statistics of its output (the caller sized `part` with conv3x3p_stats_bytes; a split-K launch
// leaves them from its slab reduction, conv3x3p_reduce_stats_kernel).
static thread_local float* g_stat_arm = nullptr;
void conv3x3p_arm_stats(float* part) { g_stat_arm = part; }
float* conv3x3p_take_stats() { float* p = g_stat_arm; g_stat_arm = nullptr; return p; }
size_t conv3x3p_stats_bytes(int math, int C, int H, int W, int N, int Nout, int fused_skip_chunks, int* tpi) {
  if (Nout & 3) return 0;
  if (!fused_skip_chunks && conv3x3p_form(math, C, H, W, N, Nout)) {      // Winograd form: one entry per 8 x 16 pixels and wave, whatever the tile height
    const int t = (H / 8) * (W / PTW);
    if (tpi) *tpi = t;
    return (size_t)N * t * (Nout >> 2) * 2 * sizeof(float);
  }
  PatchPlan q = patch_plan(C + 32 * fused_skip_chunks, H, W, N, Nout);
  if (fused_skip_chunks) q.splits = 1;
  const int split_stats = pdae_knob(KNOB_SPLIT_STATS);       // 0: split launches leave none (A/B aid)
  if (q.splits != 1 && (Nout > 1024 || !split_stats)) return 0;
  const int t = q.splits != 1 ? (H * W + reduce_stats_tp(N, H * W) - 1) / reduce_stats_tp(N, H * W) : q.tiles_x * q.tiles_y * (q.th / 8);
  if (tpi) *tpi = t;
  return (size_t)N * t * (Nout >> 2) * 2 * sizeof(float);
}

int conv3x3p_launch(int math, const float* x, int N, int Hs, int Ws, int C, int H, int W, int up, const unsigned short* wp, int Nout,
                    float* y, const float* bias, const float* res, int res_mode, int accumulate, hipStream_t s, const float* x1, int C0,
                    const float* coef, int act, const PatchSkip* sk, const float* amax, float* stat_part, const PatchGnb* gb) {
  const int math_form = math;                             // may carry the direct bit (forward launches): only conv3x3p_form looks at it
  math &= ~PDAE_MATH_DIRECT_BIT;
  PatchParams P;
  P.x1 = x1; P.C0 = x1 ? C0 : C; P.coef = coef; P.act = act;
  P.woscale = 1.0f / conv3x3p_wscale(C); P.amax = amax; P.sat = pdae_sat_counter();
  P.nx = 0; P.s0 = P.s1 = nullptr; P.Cs0 = P.Cs1 = 0; P.wps = nullptr; P.bias_x = nullptr;
  P.gb_x0 = P.gb_x1 = nullptr; P.gb_C0 = 0; P.gb_coef = nullptr; P.gb_part = nullptr; P.gb_tpi = 0;
  if (sk) { P.nx = (sk->C0 + sk->C1) >> 5; P.s0 = sk->s0; P.s1 = sk->s1; P.Cs0 = sk->C0; P.Cs1 = sk->C1; P.wps = sk->wps; P.bias_x = sk->bias; }
  P.x = x; P.N = N; P.Hs = Hs; P.Ws = Ws; P.C = C; P.H = H; P.W = W; P.up = up; P.wp = wp; P.NT = (Nout + 31) / 32; P.Nout = Nout;
  P.y = y; P.bias = bias; P.res = res; P.res_mode = res_mode; P.accumulate = accumulate;
  PatchPlan q = patch_plan(C + 32 * P.nx, H, W, N, Nout);
  if (sk) { q.splits = 1; q.cps = (C >> 5) + P.nx; }          // the slabs behind wp are sized for the main convolution alone
  if (sk && (q.w8 || up || (sk->C0 & 31) || (sk->C1 & 31))) { pdae_set_error("conv3x3p: fused skip convolution not eligible for this shape"); return PDAE_EINVAL; }
  if (coef && (q.w8 || (x1 && (C0 & 31)))) { pdae_set_error("conv3x3p: fused GroupNorm input needs W %% 16 == 0 and C0 %% 32 == 0"); return PDAE_EINVAL; }
  P.tiles_x = q.tiles_x; P.tiles_y = q.tiles_y; P.tiles_n = q.tiles_n; P.splits = q.splits; P.cps = q.cps;
  const int form = (!q.w8 && conv3x3p_form(math_form, C, H, W, N, Nout)) ? 1 : 0;
  if (int e = form_check(wp, form, form_sig(math, Nout, C))) return e;
  P.slab = (float*)((char*)wp + prep_bytes(math, Nout, C, form));
  P.stat_part = stat_part;
  P.stat_tpi = q.splits != 1 ? (H * W + reduce_stats_tp(N, H * W) - 1) / reduce_stats_tp(N, H * W) : q.tiles_x * q.tiles_y * (q.th / 8);
  if (P.stat_part && ((Nout & 3) || (q.splits != 1 && Nout > 1024))) { pdae_set_error("conv3x3p: output statistics requested for Nout = %d", Nout); return PDAE_EINVAL; }
  // Winograd F(2, 3) along x (conv3x3x.hip): two thirds of the MFMAs; the prepared weights are in that form iff conv3x3p_form says so
  if (gb) {
    const int tpi = conv3x3p_gnb_tiles(math_form, C, H, W, N, Nout, gb->C0, gb->C1);
    if (!tpi || !form || sk || coef || res_mode || accumulate || stat_part || up) {
      pdae_set_error("conv3x3p: GroupNorm-backward sums were requested (pdae_conv_gnbwd_arm) but this data gradient cannot leave them (pdae_conv_gnbwd_bytes == 0)");
      return PDAE_EINVAL;
    }
    P.gb_x0 = gb->x0; P.gb_x1 = gb->C1 ? gb->x1 : nullptr; P.gb_C0 = gb->C1 ? gb->C0 : Nout; P.gb_coef = gb->coef; P.gb_part = gb->part; P.gb_tpi = tpi;
  }
  if (form) {
    if (sk) { pdae_set_error("conv3x3p: fused skip chunks are not built for the Winograd form (pdae_conv2d_fwd_skip_ok == 0 for this shape)"); return PDAE_EINVAL; }
    if (coef && !act) { pdae_set_error("conv3x3p: fused GroupNorm input without SiLU is not built for the Winograd form"); return PDAE_EINVAL; }
    P.stat_tpi = (H / 8) * (W / 16);
    return conv3x3x_launch(math, P, s);
  }
  // large layers: persistent workgroups, one wave per SIMD, epilogue of tile i inside tile i+1 (conv3x3r.hip); same parameters, same results
  if (q.splits == 1 && !q.w8 && (!coef || act) && conv3x3r_ok(math, C, H, W, N, Nout, Hs, Ws, P.C0, P.Cs0, P.Cs1)) return conv3x3r_launch(math, P, s);
#define PDAE_P3(NS_)                                                                                                    \
  (coef ? (q.th == 16 ? launch_ns<NS_, 16, false, true>(P, s) : launch_ns<NS_, 8, false, true>(P, s))                    \
        : (q.w8 ? launch_ns<NS_, 8, true>(P, s) : q.th == 16 ? launch_ns<NS_, 16, false>(P, s) : launch_ns<NS_, 8, false>(P, s)))
  if (math == 1) return PDAE_P3(1);
  if (math == 2) return PDAE_P3(2);
  if (math == 4) return PDAE_P3(4);
  return PDAE_P3(3);
#undef PDAE_P3
}


// ---------------------------------------------------------------------------------------------
// weight preparation: w [Nout][9][C] fp32 (or, transposed: the data-gradient weights of w [C][9][Nout])  ->  wp [NS][C/32][9][2][NT][64][8] bf16 planes in MFMA B-fragment order
// (lane l of a wave holds k = (l>>5)*8 + j, j = 0..7, of output channel nt*32 + (l&31)).  One thread per fragment slot.
// ---------------------------------------------------------------------------------------------
template <int NS>
__global__ void __launch_bounds__(256) conv3x3p_wprep_kernel(const float* __restrict__ w, int Nout, int C, int NT, int transposed, float wscale,
                                                             int T, unsigned short* __restrict__ wp) {
  const size_t nslot = (size_t)(C >> 5) * 2 * T * NT * 64;          // T taps: 9 (3x3) or 1 (fused 1x1 skip chunks); 12 / 2 in the Winograd-along-x form
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nslot; i += (size_t)gridDim.x * 256) {
    if (transposed & PDAE_WPREP_FORM_X) wprepx_slot<NS>(w, Nout, C, NT, transposed & 1, wscale, T, wp, i);
    else wprep3_slot<NS>(w, Nout, C, NT, transposed, wscale, T, wp, i);
  }
}

static int wprep_launch(int math, const float* w, int Nout, int C, int transposed, float wscale, int T, unsigned short* wp, hipStream_t s) {
  const int NT = (Nout + 31) / 32;
  const size_t nslot = (size_t)(C >> 5) * 2 * T * NT * 64;
  int grid = (int)((nslot + 255) / 256); if (grid > 4096) grid = 4096;
  if (math == 1) hipLaunchKernelGGL(conv3x3p_wprep_kernel<1>, dim3(grid), dim3(256), 0, s, w, Nout, C, NT, transposed, wscale, T, wp);
  else if (math == 2) hipLaunchKernelGGL(conv3x3p_wprep_kernel<2>, dim3(grid), dim3(256), 0, s, w, Nout, C, NT, transposed, wscale, T, wp);
  else if (math == 4) hipLaunchKernelGGL(conv3x3p_wprep_kernel<4>, dim3(grid), dim3(256), 0, s, w, Nout, C, NT, transposed, wscale, T, wp);
  else hipLaunchKernelGGL(conv3x3p_wprep_kernel<3>, dim3(grid), dim3(256), 0, s, w, Nout, C, NT, transposed, wscale, T, wp);
  return pdae_launch_status("conv3x3p_wprep");
}

int conv3x3p_wprep(int math, const float* w, int Nout, int C, int transposed, unsigned short* wp, hipStream_t s, int H, int W, int N) {
  const int form = conv3x3p_form(math, C, H, W, N, Nout);
  math &= ~PDAE_MATH_DIRECT_BIT;
  form_note(wp, form, form_sig(math, Nout, C));
  if (form) return wprep_launch(math, w, Nout, C, transposed | PDAE_WPREP_FORM_X, conv3x3p_wscale(C), 12, wp, s);
  return wprep_launch(math, w, Nout, C, transposed, conv3x3p_wscale(C), 9, wp, s);
}
static void fill_job3(int math, const float* w, int Nout, int C, int transposed, float wscale, int T, unsigned short* wp, WprepJob* j) {
  j->w = w; j->wp = wp; j->Nout = Nout; j->C = C; j->NT = (Nout + 31) / 32; j->transposed = transposed; j->T = T;
  j->ns = math < 1 ? 3 : (math > 4 ? 3 : math); j->wscale = wscale;
  j->nblocks = (int)(((size_t)(C >> 5) * 2 * T * j->NT * 64 + 255) / 256);
}
void conv3x3p_wprep_job(int math, const float* w, int Nout, int C, int transposed, unsigned short* wp, WprepJob* j, int H, int W, int N) {
  const int form = conv3x3p_form(math, C, H, W, N, Nout);
  math &= ~PDAE_MATH_DIRECT_BIT;
  form_note(wp, form, form_sig(math, Nout, C));
  if (form) fill_job3(math, w, Nout, C, transposed | PDAE_WPREP_FORM_X, conv3x3p_wscale(C), 12, wp, j);
  else fill_job3(math, w, Nout, C, transposed, conv3x3p_wscale(C), 9, wp, j);
}
void conv3x3p_skip_wprep_job(int math, const float* w, int Nout, int Cs, int Cmain, unsigned short* wp, WprepJob* j, int H, int W, int N) {
  (void)H; (void)W; (void)N;                               // launches with fused skip chunks exist in the direct form only
  fill_job3(math, w, Nout, Cs, 0, conv3x3p_wscale(Cmain) * PASCALE, 1, wp, j);
}

// weights of a 1x1 skip convolution [Nout][Cs] for the skip chunks of a 3x3 launch whose main input has Cmain channels: same plane
// format and (fp16 format) the same power-of-two scale as the main weights
size_t conv3x3p_skip_wprep_bytes(int math, int Nout, int Cs) {      // one centre tap = 2 k-halves per chunk (fused skip chunks exist in the direct form only)
  const int NS = math < 1 ? 1 : (math == 4 ? 2 : (math > 3 ? 3 : math));
  return (((size_t)NS * (Cs >> 5) * 2 * ((Nout + 31) / 32) * 512 * sizeof(unsigned short)) + 255) & ~(size_t)255;
}
int conv3x3p_skip_wprep(int math, const float* w, int Nout, int Cs, int Cmain, unsigned short* wp, hipStream_t s, int H, int W, int N) {
  (void)H; (void)W; (void)N;
  return wprep_launch(math, w, Nout, Cs, 0, conv3x3p_wscale(Cmain) * PASCALE, 1, wp, s);      // x PASCALE: the skip chunks' activations are unscaled
}
