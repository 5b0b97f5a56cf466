// Weight gradient of a 3x3 / stride-1 / pad-1 convolution on the bf16 MFMA pipe with split fp32 operands.
//
//   dW[co][tap][ci] = sum over pixels p of  dY[p][co] * X[p + tap][ci]
//
// The contraction runs over PIXELS while both tensors are pixel-major in memory ([pixel][channel], NHWC), i.e. both
// GEMM operands arrive "transposed".  Instead of transposing through LDS stores (igemm.hip wgrad path: 16 KB gathers and
// two barriers per 32 pixels, every tap re-reading dY and X), this kernel keeps the natural layout in LDS and lets the
// gfx950 transposing LDS read build the MFMA fragments:
//   ds_read_b64_tr_b16: within a 16-lane group lane i passes the address of row (i>>2), columns (i&3)*4..+3 and receives
//   the 4-row column i (verified on MI355X by tools/probes/tr_probe.hip for arbitrary row strides).
// Per 8x16-pixel tile the block stages dY[128 px][64 co] and the haloed X patch [10x18 px][32 ci] once (NS planes each,
// see igemm.hip) and all nine taps read shifted pixel rows of the same patch.  Block = 4 waves: wave (a, th) owns output
// channels a*32..+31 and half of the taps -- th 0: taps 0..3 plus the centre tap on tile rows 0..3, th 1: taps 5..8 plus the
// centre tap on rows 4..7 (108 MFMAs per tile either way; the two centre halves meet through LDS once, in the epilogue) --
// i.e. 5 accumulator tiles.  60 KB of LDS and ~170 VGPRs put TWO blocks on a CU, so one block's staging phase (global ->
// split -> LDS, fenced by two barriers) runs under the other block's MFMA phase.  The earlier layout (128 co per block, 8 waves
// x 9 accumulators, one block per CU) idled the matrix pipe during every staging phase: MFMA busy 0.42 (profiles/r02_pmc_sq.txt).
// Split-K over pixel tiles: one slab per split, reduced in fixed order by splitk_reduce_kernel.
//
// Replaces the weight-gradient of F.conv2d(k=3, padding=1) (model/module.py:242,265).
#include "common.h"
#include "igemm.h"
#include "conv3x3w.h"
#include <type_traits>
#include <cstdlib>


// Two blocks share a CU so that one stages while the other multiplies -- but blocks launched together run in lock step (same phase lengths),
// stage at the same time and fight for the matrix pipe at the same time.  Every block therefore counts its arrival on its CU (hardware id
// registers -> one counter per CU, never reset: only the parity matters) and every second arrival starts half a tile late; the offset then
// persists because both blocks have the same period.  Timing only: results do not depend on it.
__device__ unsigned int w3_cu_arrivals[4096];
__device__ __forceinline__ void w3_phase_offset(int stagger, unsigned* lds_word) {
  if (stagger <= 0) return;
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID [3:0]
    *lds_word = atomicAdd(&w3_cu_arrivals[((hw >> 8) & 0xffu) | ((xcc & 0xfu) << 8)], 1u);
  }
  __syncthreads();
  const unsigned late = *lds_word & 1u;
  __syncthreads();
  if (late) for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);
}

template <int NS, bool W8 = false, bool GN = false>
__global__ void __launch_bounds__(WTHREADS, 2) conv3x3w_kernel(const WgradParams P) {
  static_assert(!(GN && W8), "the fused GroupNorm input is not built for image-pair tiles");
  constexpr int WPW = WPW_(W8), WNPIX = WNPIX_(W8), WX_LD = WX_LD_(W8);
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  constexpr int SX = WNPL(NS) * WNPIX * WSX;
  unsigned short* sX = smem;                 // [NS][180][WSX]
  unsigned short* sY = smem + SX;            // [NS][128][WSY]

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int a = wv & 1, th = wv >> 1;        // output-channel tile, tap half
  const int li = lane & 31, h = lane >> 5, i16 = lane & 15, g16 = (lane >> 4) & 1;

  // XCD-aware bijective block order (block b runs on XCD b % 8, each XCD has its own L2): consecutive LOGICAL ids share an XCD, and the
  // ci_chunk index runs fastest, so the C/32 blocks that stream the same dY tiles (and the Cout/128 blocks that stream the same X patches)
  // hit in one L2 instead of fetching the tensor once per XCD
  const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  const int ci_chunk = bid % P.ci_chunks; bid /= P.ci_chunks;
  const int co_tile = bid % P.co_tiles; const int split = bid / P.co_tiles;
  const int ci0 = ci_chunk * 32, co0 = co_tile * WCO;
  const int Cout = P.Cout;
  // the block's 32 input channels lie in ONE tensor of the virtual concat (C0 % 32 == 0): source pointer, its channel count, first channel in it
  const bool src_first = !GN || ci0 < P.C0;
  const float* const xsrc = src_first ? P.x : P.x1;
  const int C = GN ? (src_first ? P.C0 : P.C - P.C0) : P.C;          // channels per pixel of the tensor this block READS (P.C: of the weight)
  const int cis = GN ? (src_first ? ci0 : ci0 - P.C0) : ci0;

  f32x16 acc[5];                            // th 0: taps 0..3, th 1: taps 5..8; acc[4]: this wave's half of the centre tap
#pragma unroll
  for (int tp = 0; tp < 5; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

  // lane-constant parts of the transposing-read addresses (bytes)
  const unsigned y_lane = (unsigned)(((h * 8 + (i16 >> 2)) * WSY + ((a * 32 + g16 * 16 + (i16 & 3) * 4) ^ WSWZ(i16 >> 2))) * 2);   // rows h*8 + (i16>>2) [+4]: low 2 bits = i16>>2
  const unsigned x_lane = (unsigned)(((h * (W8 ? 10 : 8) + (i16 >> 2)) * WSX + g16 * 16 + (i16 & 3) * 4) * 2);   // W8: columns 8..15 = second image
  const unsigned sX_base = 0u, sY_base = (unsigned)(SX * 2);   // LDS byte offsets: the dynamic segment is the only LDS of this kernel

  float4 xpre[WX_LD], ypre[WY_LD];
  float sat_hit = 0.f;                      // fp16 format: lanes with an activation clamped into the window (common.h)
  const bool want_db = P.db_part != nullptr && ci_chunk == 0;       // this block also owns the column sums of its dY tiles
  // per-thread running column sums live in LDS behind the operand planes (thread-private slots: deterministic, no registers held
  // across the MFMA phase); thread: output channels (t & 15) * 4 .. +3, pixels idx >> 4
  float4* bred = reinterpret_cast<float4*>(smem + SX + WNPL(NS) * WTPIX * WSY);
  const float yscale = NS == 4 ? w_pow2_scale(*P.dy_amax) : 1.0f;
  if (want_db) bred[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  // Tile-invariant halves of the operand addresses, computed once per block (a VALU instruction costs ~5 issue cycles of a two-wave SIMD during
  // which the matrix pipe idles, and the per-tile address arithmetic used to be a third of this kernel's VALU stream):
  //   X patch slot l of this thread = pixel (py, px) of the haloed patch, channel quad qd:  xw[l] = (element offset from the tile's origin pixel) * 64
  //   + border bits {1: top halo row, 2: bottom halo row, 4: left halo column, 8: right halo column, 16: never valid, 32: second image of a pair};
  //   a tile contributes the set of bits that fall outside the image (scalar), so validity is one AND per load.
  //   dY slot l = tile row l, column t >> 4, channel quad t & 15: one thread offset plus l * row stride.
  int xw[WX_LD];
  const int ush = P.up ? 1 : 0;              // nearest-neighbour 2x upsampling in front of the convolution: stored pixel = logical >> 1 (tile origins are even)
#pragma unroll
  for (int l = 0; l < WX_LD; ++l) {
    const int idx = t + WTHREADS * l, pix = idx >> 3, qd = idx & 7;
    int bm = 16, rel = 0;
    if (pix < WNPIX) {
      const int py = pix / WPW, px = pix - py * WPW;
      int lx = px - 1, sub = 0;
      bm = (py == 0 ? 1 : 0) | (py == WTH + 1 ? 2 : 0);
      if constexpr (W8) { sub = px >= 10; lx = px - 1 - 10 * sub; bm |= ((unsigned)lx >= 8u ? 16 : 0) | (sub ? 32 : 0); }
      else bm |= (px == 0 ? 4 : 0) | (px == WTW + 1 ? 8 : 0);
      rel = ((sub * P.Hs + ((py - 1) >> ush)) * P.Ws + (lx >> ush)) * C + qd * 4;
    }
    xw[l] = rel * 64 + bm;
  }
  const int ycol = t >> 4, yc4 = t & 15;
  const bool co_ok = co0 + yc4 * 4 < Cout;
  const int ythr_a = ((W8 ? (ycol & 7) : ycol) * Cout) + (co_ok ? yc4 * 4 : 0);                    // first (only) image of the tile
  const int ythr = W8 ? ythr_a + (ycol >> 3) * P.H * P.W * Cout : ythr_a;                          // W8: columns 8..15 = second image of the pair
  const int yrow = P.W * Cout;
  bool y_ok = co_ok;                          // W8: also false for the missing second image of an odd batch (set per tile)
  const float yscale_ok = (W8 || co_ok) ? yscale : 0.f;      // fp16 format, 16-pixel-wide tiles: lanes beyond Cout load a valid address and are scaled to zero
  float4 gmu, gsc, gsh;                      // GN: coefficients of the tile being staged (loaded with it, used by lstore)
  unsigned xokm = 0u;                        // GN: validity bit per patch slot (padding is zero AFTER the map)
  auto gload = [&](int tile) {
    int img = tile / (P.tiles_y * P.tiles_x); int rem = tile - img * P.tiles_y * P.tiles_x;
    int ty = rem / P.tiles_x, tx = rem - ty * P.tiles_x;
    if constexpr (W8) img *= 2;                 // first image of the pair
    const int y0 = ty * WTH, x0 = tx * WTW;
    const bool pair_missing = W8 && img + 1 >= P.N;
    const int tmask = 16 | (y0 == 0 ? 1 : 0) | (y0 + WTH >= P.H ? 2 : 0) | (!W8 && x0 == 0 ? 4 : 0) | (!W8 && x0 + WTW >= P.W ? 8 : 0) | (pair_missing ? 32 : 0);
    const float* xb = xsrc + ((size_t)(img * P.Hs + (y0 >> ush)) * P.Ws + (x0 >> ush)) * C + cis;
    if constexpr (GN) {                          // this thread's channel quad (t & 7 in every slot) of the tile's image: [mu | a | b]
      const float* cf = P.coef + (size_t)img * P.C + ci0 + (t & 7) * 4;
      const size_t NC = (size_t)P.N * P.C;
      gmu = *reinterpret_cast<const float4*>(cf); gsc = *reinterpret_cast<const float4*>(cf + NC); gsh = *reinterpret_cast<const float4*>(cf + 2 * NC);
      xokm = 0u;
    }
#pragma unroll
    for (int l = 0; l < WX_LD; ++l) {            // unconditional loads from clamped addresses, zeroed afterwards
      const bool ok = (xw[l] & tmask) == 0;
      const float4 v = *reinterpret_cast<const float4*>(xb + (ok ? xw[l] >> 6 : 0));
      if constexpr (GN) { xpre[l] = v; xokm |= (ok ? 1u : 0u) << l; }      // zeroed AFTER the map (lstore)
      else xpre[l] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* yb = P.dy + ((size_t)(img * P.H + y0) * P.W + x0) * Cout + co0;
    if constexpr (W8) y_ok = co_ok && !(pair_missing && ycol >= 8);
    const int yo = (W8 && pair_missing) ? ythr_a : ythr;
#pragma unroll
    for (int l = 0; l < WY_LD; ++l) {
      const float4 v = *reinterpret_cast<const float4*>(yb + yo + l * yrow);
      if constexpr (W8 || NS != 4) ypre[l] = y_ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      else ypre[l] = v;                          // fp16 format: a lane beyond Cout is zeroed by its operand scale instead (lstore)
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int l = 0; l < WX_LD; ++l) {
      int idx = t + WTHREADS * l; int pix = idx >> 3, qd = idx & 7;
      if (pix < WNPIX) {
        if constexpr (GN) xpre[l] = w_gn_map(xpre[l], (xokm >> l) & 1u, gmu, gsc, gsh, P.act);
        if constexpr (NS == 4) pdae_f16_amax4(xpre[l], WXSCALE, sat_hit);
        unsigned u[WNPL(NS)], v[WNPL(NS)];
        w_split2<NS>(xpre[l].x, xpre[l].y, u, WXSCALE); w_split2<NS>(xpre[l].z, xpre[l].w, v, WXSCALE);
#pragma unroll
        for (int p = 0; p < WNPL(NS); ++p) *reinterpret_cast<uint2*>(&sX[(p * WNPIX + pix) * WSX + qd * 4]) = make_uint2(u[p], v[p]);
      }
    }
    if (want_db) {
      float4 b4 = bred[t];
#pragma unroll
      for (int l = 0; l < WY_LD; ++l) { b4.x += ypre[l].x; b4.y += ypre[l].y; b4.z += ypre[l].z; b4.w += ypre[l].w; }
      bred[t] = b4;
    }
#pragma unroll
    for (int l = 0; l < WY_LD; ++l) {
      int idx = t + WTHREADS * l; int pix = idx >> 4, c4 = idx & 15;
      unsigned u[WNPL(NS)], v[WNPL(NS)];
      w_split2<NS>(ypre[l].x, ypre[l].y, u, yscale_ok); w_split2<NS>(ypre[l].z, ypre[l].w, v, yscale_ok);
#pragma unroll
      for (int p = 0; p < WNPL(NS); ++p) *reinterpret_cast<uint2*>(&sY[(p * WTPIX + pix) * WSY + ((c4 * 4) ^ WSWZ(pix))]) = make_uint2(u[p], v[p]);
    }
  };

  // MFMA phase of one staged tile for tap half TH: 8 k-chunks (tile rows of 16 pixels) x (4 taps + the centre tap on this wave's 4 rows) = 36
  // tap steps of 3 MFMAs (2 fragment reads per plane each), fully unrolled and software-pipelined by hand: the fragments of step s+2 are
  // requested before the MFMAs of step s (ring of three buffers), so that the LDS latency of a step hides under two steps of matrix work
  // instead of being exposed once per k-chunk (hipcc clusters the reads of a loop iteration at its top and does not pipeline across iterations).
  auto mma_phase = [&](auto th_c) {
    constexpr int TH = decltype(th_c)::value;
    constexpr int PD = NS == 3 ? 1 : 2;          // prefetch distance in steps (three planes: registers only allow one)
    uint4 af[2][WNPL(NS)], bq[PD + 1][WNPL(NS)];
    auto lda = [&](uint4 (&f)[WNPL(NS)], int kc) {
#pragma unroll
      for (int p = 0; p < WNPL(NS); ++p) {
        const unsigned ad = sY_base + y_lane + (unsigned)((p * WTPIX + kc * 16) * WSY * 2);
        f[p] = tr_frag(ad, ad + 4 * WSY * 2);
      }
    };
    auto ldb = [&](uint4 (&f)[WNPL(NS)], int kc, int j) {
      const int tp = j == 4 ? 4 : TH * 5 + j;
      const int dy = tp / 3, dx = tp - dy * 3;
#pragma unroll
      for (int p = 0; p < WNPL(NS); ++p) {
        const unsigned ad = sX_base + x_lane + (unsigned)((p * WNPIX + (kc + dy) * WPW + dx) * WSX * 2);
        f[p] = tr_frag(ad, ad + 4 * WSX * 2);
      }
    };
    lda(af[0], 0); ldb(bq[0], 0, 0);
    if constexpr (PD == 2) ldb(bq[1], 0, 1);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
#ifdef PDAE_W3_PROBE_6TAPS       // timing probe (WRONG results): 24 of the 36 tap steps = the matrix work of a Winograd F(3, 2)-along-x weight gradient
      const int n = 3;
      const int before = 3 * kc;
#else
      const int n = (kc >> 2) == TH ? 5 : 4;                                   // taps of this k-chunk
      const int before = 4 * kc + (TH == 0 ? (kc < 4 ? kc : 4) : (kc > 4 ? kc - 4 : 0));   // steps before this k-chunk
#endif
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (j < n) {
          const int cur = (before + j) % (PD + 1), nxt = (before + j + PD) % (PD + 1);
          if (j + PD < n) ldb(bq[nxt], kc, j + PD);
          else if (kc + 1 < 8) {
            if (j + PD == n) lda(af[(kc + 1) & 1], kc + 1);
            ldb(bq[nxt], kc + 1, j + PD - n);
          }
#define WA(P_) __builtin_bit_cast(bf16x8, af[kc & 1][P_])
#define WB(P_) __builtin_bit_cast(bf16x8, bq[cur][P_])
#define WAH(P_) __builtin_bit_cast(f16x8, af[kc & 1][P_])
#define WBH(P_) __builtin_bit_cast(f16x8, bq[cur][P_])
          if constexpr (NS == 4) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WAH(0), WBH(1), acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WAH(1), WBH(0), acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WAH(0), WBH(0), acc[j], 0, 0, 0);
          } else {
            if constexpr (NS == 3) {
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA(1), WB(1), acc[j], 0, 0, 0);
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA(0), WB(2), acc[j], 0, 0, 0);
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA(2), WB(0), acc[j], 0, 0, 0);
            }
            if constexpr (NS >= 2) {
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA(0), WB(1), acc[j], 0, 0, 0);
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA(1), WB(0), acc[j], 0, 0, 0);
            }
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA(0), WB(0), acc[j], 0, 0, 0);
          }
#undef WA
#undef WB
#undef WAH
#undef WBH
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };

  const int t_beg = split * P.tiles_per_split;
  const int t_end = min(P.ntiles, t_beg + P.tiles_per_split);
  if (t_beg < t_end) gload(t_beg);
  w3_phase_offset(P.stagger, reinterpret_cast<unsigned*>(smem));
  for (int tile = t_beg; tile < t_end; ++tile) {
    __syncthreads();                          // previous tile's fragments have been consumed
#ifndef PDAE_W3_PROBE_NOSTAGE
    lstore();
#endif
    __syncthreads();
#ifndef PDAE_W3_PROBE_NOLOAD
    if (tile + 1 < t_end) gload(tile + 1);    // next tile in flight under the 108 MFMAs per wave
#endif
#ifndef PDAE_W3_PROBE_NOMMA            // timing probes (tools/probe_build.py): wrong results by design
    if (th == 0) mma_phase(std::integral_constant<int, 0>{}); else mma_phase(std::integral_constant<int, 1>{});
#endif
  }

  if (P.db_part != nullptr) {                 // bias gradient: 16 threads share each channel quad -> LDS -> fixed-order sum
    __syncthreads();
    if (want_db && t < 16) {
      float4 s4 = bred[t];
      for (int k = 1; k < 16; ++k) { const float4 u = bred[t + 16 * k]; s4.x += u.x; s4.y += u.y; s4.z += u.z; s4.w += u.w; }
      const int co = co0 + t * 4;
      if (co < Cout) *reinterpret_cast<float4*>(P.db_part + (size_t)split * Cout + co) = s4;      // Cout % 4 == 0
    }
  }
  const float oscale = NS == 4 ? 1.0f / (yscale * WXSCALE) : 1.0f;      // exact: both scales are powers of two
  if constexpr (NS == 4) pdae_sat_report(P.sat, sat_hit);
  // the two halves of the centre tap meet in LDS (operand planes are dead now): th 1 hands its acc[4] to the th 0 wave of the same channels
  __syncthreads();
  float* cen = reinterpret_cast<float*>(smem) + a * 16 * 64;
  if (th == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) cen[r * 64 + lane] = acc[4][r];
  }
  __syncthreads();
  if (th == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[4][r] += cen[r * 64 + lane];
  }
  // epilogue: slab `split` of the workspace, layout [Cout][9][C]
  float* slab = P.ws + (size_t)split * Cout * 9 * P.C;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    if (j == 4 && th == 1) break;
    const int tp = j == 4 ? 4 : th * 5 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (co < Cout) slab[((size_t)co * 9 + tp) * P.C + ci0 + li] = NS == 4 ? acc[j][r] * oscale : acc[j][r];
    }
  }
}

// split of the pixel tiles over blocks: two blocks per CU (60 KB of LDS in the default format), so the grid should fill 512 slots in whole
// rounds: minimise rounds(grid) x tiles-per-block (a 520-block grid costs two rounds for the work of one)
static void wgradp_plan(int N, int H, int W, int C, int Cout, int& splits, int& tiles_per_split) {
  const int ntiles = W == 8 ? ((N + 1) / 2) * (H / WTH) : N * (H / WTH) * (W / WTW);
  const int base = ((Cout + WCO - 1) / WCO) * (C / 32);
  int maxs = ntiles / 4; if (maxs < 1) maxs = 1;      // at least 4 tiles per block
  if (maxs > 128) maxs = 128;
  long long best = -1; int best_s = 1;
  for (int s = 1; s <= maxs; ++s) {
    const int tps = (ntiles + s - 1) / s, sp = (ntiles + tps - 1) / tps;
    if (sp != s) continue;
    const long long rounds = ((long long)base * sp + 511) / 512;
    const long long cost = rounds * (tps + 1);          // +1: per-block prologue / slab write
    if (best < 0 || cost < best) { best = cost; best_s = s; }
  }
  tiles_per_split = (ntiles + best_s - 1) / best_s;
  splits = (ntiles + tiles_per_split - 1) / tiles_per_split;
}

bool conv3x3w_ok(int math, int KH, int KW, int stride, int pad, int C1, int C, int H, int W, int N, int Cout) {
  if (math < 1 || KH != 3 || KW != 3 || stride != 1 || pad != 1 || C1 != 0) return false;
  if ((C & 31) || (H % WTH) || ((W % WTW) && W != 8) || (Cout & 3) || Cout < 32) return false;
  if (W == 8) return (long long)((N + 1) / 2) * (H / WTH) >= 8;       // image pairs: the 8x8 bottleneck layers
  return (long long)N * (H / WTH) * (W / WTW) >= 64;
}

// the GN instantiation (fused GroupNorm + SiLU input, two-source): both sources whole 32-channel chunks, 16-pixel-wide tiles
bool conv3x3w_gn_ok(int math, int KH, int KW, int stride, int pad, int C0, int C1, int H, int W, int N, int Cout) {
  if (W == 8 || (C0 & 31) || (C1 & 31)) return false;
  return conv3x3w_ok(math, KH, KW, stride, pad, 0, C0 + C1, H, W, N, Cout);
}

// slabs [splits][Cout][9][C] + bias-gradient partials [splits][Cout]
// (the larger of the two forms' plans where both apply: the knob may change between sizing and launch)
size_t conv3x3w_workspace_bytes(int N, int H, int W, int C, int Cout) {
  int splits, tps;
  wgradp_plan(N, H, W, C, Cout, splits, tps);
  size_t b = ((size_t)splits * Cout * 9 * C + (size_t)splits * Cout) * sizeof(float);
  if (conv3x3v_ok(4, C, H, W, N, Cout)) { const size_t bv = conv3x3v_workspace_bytes(N, H, W, C, Cout); if (bv > b) b = bv; }
  return b;
}

template <int NS, bool W8 = false, bool GN = false> static int launch_w(const WgradParams& P, hipStream_t s) {
  const size_t smem = (size_t)(WNPL(NS) * WNPIX_(W8) * WSX + WNPL(NS) * WTPIX * WSY) * sizeof(unsigned short) + WTHREADS * sizeof(float4);   // + bias-sum slots
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3w_kernel<NS, W8, GN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { pdae_set_error("conv3x3w: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3w_kernel<NS, W8, GN>), dim3(P.splits * P.co_tiles * P.ci_chunks), dim3(WTHREADS), smem, s, P);
  return pdae_launch_status("conv3x3w");
}

int conv3x3w_launch(int math, const float* x, int N, int Hs, int Ws, int C, int H, int W, int up, const float* dy, int Cout, float* dw,
                    int accumulate, float* ws, size_t ws_bytes, hipStream_t s, float** db_part, int* db_rows, const float* dy_amax, float* db,
                    const float* x1, int C0, const float* coef, int act) {
  WgradParams P;
  P.x1 = x1; P.C0 = x1 ? C0 : C; P.coef = coef; P.act = act;
  if (!coef && x1) { pdae_set_error("conv3x3w: a two-source input needs the fused GroupNorm form (coef)"); return PDAE_EINVAL; }
  if (coef && (W == 8 || (P.C0 & 31) || ((C - P.C0) & 31))) { pdae_set_error("conv3x3w: fused GroupNorm input needs W %% 16 == 0 and both sources whole 32-channel chunks"); return PDAE_EINVAL; }
  P.dy_amax = dy_amax; P.sat = pdae_sat_counter();
  const int stagger = pdae_knob(KNOB_W3_STAGGER);      // off by default: measured +-0 (the kernel is power-bound, not phase-bound)
  P.stagger = stagger;
  if (math == 4 && !dy_amax) math = 3;          // fp16 format needs the dY scale: without it the exact bf16 split runs
  if (pdae_knob(KNOB_W3V) && conv3x3v_ok(math, C, H, W, N, Cout))      // producer / consumer form (conv3x3v.hip)
    return conv3x3v_launch(math, x, N, Hs, Ws, C, H, W, up, dy, Cout, dw, accumulate, ws, ws_bytes, s, db_part, db_rows, dy_amax, db, x1, P.C0, coef, act);
  P.x = x; P.N = N; P.Hs = Hs; P.Ws = Ws; P.C = C; P.H = H; P.W = W; P.up = up; P.dy = dy; P.Cout = Cout; P.ws = ws;
  const bool w8 = W == 8;
  P.tiles_x = w8 ? 1 : W / WTW; P.tiles_y = H / WTH; P.ntiles = (w8 ? (N + 1) / 2 : N) * P.tiles_x * P.tiles_y;
  wgradp_plan(N, H, W, C, Cout, P.splits, P.tiles_per_split);
  P.co_tiles = (Cout + WCO - 1) / WCO; P.ci_chunks = C / 32;
  const size_t need = ((size_t)P.splits * Cout * 9 * C + (size_t)P.splits * Cout) * sizeof(float);
  P.db_part = db_part ? ws + (size_t)P.splits * Cout * 9 * C : nullptr;
  if (db_part) { *db_part = P.db_part; *db_rows = P.splits; }
  if (!ws || ws_bytes < need) { pdae_set_error("conv3x3w: workspace too small (%zu < %zu)", ws_bytes, need); return PDAE_EINVAL; }
  int e;
  if (w8) e = math == 1 ? launch_w<1, true>(P, s) : (math == 2 ? launch_w<2, true>(P, s) : (math == 4 ? launch_w<4, true>(P, s) : launch_w<3, true>(P, s)));
  else if (coef) e = math == 1 ? launch_w<1, false, true>(P, s) : (math == 2 ? launch_w<2, false, true>(P, s) : (math == 4 ? launch_w<4, false, true>(P, s) : launch_w<3, false, true>(P, s)));
  else e = math == 1 ? launch_w<1>(P, s) : (math == 2 ? launch_w<2>(P, s) : (math == 4 ? launch_w<4>(P, s) : launch_w<3>(P, s)));
  if (e) return e;
  // the bias gradient's final sum rides in the reduce launch when the caller gave its destination (db_part is then reported as consumed)
  if (db_part && db) *db_part = nullptr;
  return igemm_splitk_reduce(ws, dw, (long long)Cout * 9 * C, P.splits, accumulate, s, db ? P.db_part : nullptr, P.splits, Cout, db);
}


// =============================================================================================================================
// 1x1 weight gradient:  dW[co][ci] = sum over pixels p of dY[p][co] * X[p][ci],  X = [x0 | x1] (the concat of a skip convolution is never
// materialised).  Same fragment machinery as above -- pixel-major tiles in LDS, transposing reads -- but with one tap the arithmetic intensity
// is set by how many input channels a block covers per staged dY tile: a block owns 128 output x 128 input channels (4 chunks), wave (a, cj) =
// output channels a*32..+31 x input chunks 2cj, 2cj+1, and walks all eight 16-pixel k-chunks of a 128-pixel tile itself, so there is no
// k-parity split and one slab per pixel split.  (The generic implicit-GEMM kernel ran these launches at 35-85 TFLOP/s in 6-product arithmetic.)
// Replaces the weight gradient of the ResBlock skip_connection (module.py:276) and the attention qkv / proj_out convolutions (:412,420).
// =============================================================================================================================
#define W1SY 128                     // row stride (bf16) of both operands: 128 channels = 64 dwords, all four rows of a read on the same banks ...
#define W1SWZ(pix) ((((pix) & 3)) << 5)   // ... so the channel index is XOR-swizzled with (pixel & 3) * 32
#define W1THREADS 512
struct Wgrad1Params {
  const float* x0; const float* x1; int C0, C1, C;      // X rows of M pixels, C = C0 + C1
  long long M;                                          // pixels
  const float* dy; int Cout;
  float* ws;                                            // slabs [splits][Cout][C]
  int ntiles, tiles_per_split, splits, co_tiles, ci_blocks;
  const float* dy_amax; float* db_part; unsigned int* sat;
};

template <int NS>
__global__ void __launch_bounds__(W1THREADS) conv1x1w_kernel(const Wgrad1Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  constexpr int SPL = WTPIX * W1SY;                        // one plane of one operand: 128 pixels x (128 channels + 8 pad)
  unsigned short* sX = smem;                              // [NPL][128][W1SY]
  unsigned short* sY = smem + WNPL(NS) * SPL;             // [NPL][128][W1SY]
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int a = wv & 3, cj = wv >> 2;
  const int li = lane & 31, h = lane >> 5, i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;      // XCD-aware order, ci block fastest (see conv3x3w_kernel)
  const int ci_blk = bid % P.ci_blocks; bid /= P.ci_blocks;
  const int co_tile = bid % P.co_tiles; const int split = bid / P.co_tiles;
  const int ci0 = ci_blk * 128, co0 = co_tile * 128;
  const int C = P.C, Cout = P.Cout;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const unsigned lane_row = (unsigned)((h * 8 + (i16 >> 2)) * W1SY), lane_col = (unsigned)(g16 * 16 + (i16 & 3) * 4), lane_swz = (unsigned)W1SWZ(i16 >> 2);
  const unsigned sY_base = (unsigned)(WNPL(NS) * SPL * 2);

  float4 xpre[8], ypre[8];
  float sat_hit = 0.f;
  const bool want_db = P.db_part != nullptr && ci_blk == 0;
  float4* bred = reinterpret_cast<float4*>(smem + 2 * WNPL(NS) * SPL);
  const float yscale = NS == 4 ? w_pow2_scale(*P.dy_amax) : 1.0f;
  if (want_db) bred[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  // thread-constant halves of the operand addresses (see conv3x3w_kernel): slot l of this thread = pixel (t >> 5) + 16 l of the tile, channel quad t & 31
  const int c4t = t & 31, pixt = t >> 5;
  const bool cx_ok = ci0 + c4t * 4 < C, cy_ok = co0 + c4t * 4 < Cout;
  const int cic = cx_ok ? ci0 + c4t * 4 : 0;
  const bool first = cic < P.C0;                                   // which tensor of the (never materialised) concat this thread reads
  const float* xsrc = first ? P.x0 + cic : P.x1 + (cic - P.C0);
  const int xC = first ? P.C0 : P.C1;
  const float* ysrc = P.dy + (cy_ok ? co0 + c4t * 4 : 0);
  auto gload = [&](int tile) {
    const long long m0 = (long long)tile * WTPIX;
    if (m0 + WTPIX <= P.M) {                                       // whole tile (all but possibly the last): no per-row tests
      const float* xp = xsrc + (m0 + pixt) * xC;
      const float* yp = ysrc + (m0 + pixt) * Cout;
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        const float4 vx = *reinterpret_cast<const float4*>(xp + (long long)(16 * l) * xC);
        const float4 vy = *reinterpret_cast<const float4*>(yp + (long long)(16 * l) * Cout);
        if constexpr (NS == 4) { xpre[l] = vx; ypre[l] = vy; }      // fp16 format: channels beyond C / Cout are zeroed by their operand scale (lstore)
        else { xpre[l] = cx_ok ? vx : make_float4(0.f, 0.f, 0.f, 0.f); ypre[l] = cy_ok ? vy : make_float4(0.f, 0.f, 0.f, 0.f); }
      }
      return;
    }
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      const long long m = m0 + pixt + 16 * l;
      const bool inm = m < P.M;
      const long long mm = inm ? m : 0;                   // unconditional loads from clamped addresses, zeroed afterwards
      const float4 vx = *reinterpret_cast<const float4*>(xsrc + mm * xC);
      const float4 vy = *reinterpret_cast<const float4*>(ysrc + mm * Cout);
      xpre[l] = (inm && cx_ok) ? vx : make_float4(0.f, 0.f, 0.f, 0.f);
      ypre[l] = (inm && cy_ok) ? vy : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  const float xscale_ok = cx_ok ? WXSCALE : 0.f, yscale_ok = cy_ok ? yscale : 0.f;
  auto lstore = [&]() {
    if (want_db) {
      float4 b4 = bred[t];
#pragma unroll
      for (int l = 0; l < 8; ++l) { b4.x += ypre[l].x; b4.y += ypre[l].y; b4.z += ypre[l].z; b4.w += ypre[l].w; }
      bred[t] = b4;
    }
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      const int idx = t + W1THREADS * l, pix = idx >> 5, c4 = idx & 31;
      if constexpr (NS == 4) pdae_f16_amax4(xpre[l], xscale_ok, sat_hit);
      unsigned u[WNPL(NS)], v[WNPL(NS)];
      w_split2<NS>(xpre[l].x, xpre[l].y, u, xscale_ok); w_split2<NS>(xpre[l].z, xpre[l].w, v, xscale_ok);
#pragma unroll
      for (int p = 0; p < WNPL(NS); ++p) *reinterpret_cast<uint2*>(&sX[(p * WTPIX + pix) * W1SY + ((c4 * 4) ^ W1SWZ(pix))]) = make_uint2(u[p], v[p]);
      w_split2<NS>(ypre[l].x, ypre[l].y, u, yscale_ok); w_split2<NS>(ypre[l].z, ypre[l].w, v, yscale_ok);
#pragma unroll
      for (int p = 0; p < WNPL(NS); ++p) *reinterpret_cast<uint2*>(&sY[(p * WTPIX + pix) * W1SY + ((c4 * 4) ^ W1SWZ(pix))]) = make_uint2(u[p], v[p]);
    }
  };

  const int t_beg = split * P.tiles_per_split, t_end = min(P.ntiles, t_beg + P.tiles_per_split);
  if (t_beg < t_end) gload(t_beg);
  for (int tile = t_beg; tile < t_end; ++tile) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (tile + 1 < t_end) gload(tile + 1);
#pragma unroll 2
    for (int kc = 0; kc < 8; ++kc) {                      // 16 pixels per k-chunk
      uint4 af[WNPL(NS)], bfr[2][WNPL(NS)];
#pragma unroll
      for (int p = 0; p < WNPL(NS); ++p) {
        const unsigned ad = sY_base + (lane_row + (unsigned)((p * WTPIX + kc * 16) * W1SY) + ((lane_col + a * 32) ^ lane_swz)) * 2;
        af[p] = tr_frag(ad, ad + 4 * W1SY * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const unsigned bd = (lane_row + (unsigned)((p * WTPIX + kc * 16) * W1SY) + ((lane_col + (cj * 2 + j) * 32) ^ lane_swz)) * 2;
          bfr[j][p] = tr_frag(bd, bd + 4 * W1SY * 2);
        }
      }
#define WA(P_) __builtin_bit_cast(bf16x8, af[P_])
#define WB(P_) __builtin_bit_cast(bf16x8, bfr[j][P_])
#define WAH(P_) __builtin_bit_cast(f16x8, af[P_])
#define WBH(P_) __builtin_bit_cast(f16x8, bfr[j][P_])
#define W1_EACH(STMT) _Pragma("unroll") for (int j = 0; j < 2; ++j) { STMT; }
      if constexpr (NS == 4) {
        W1_EACH(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WAH(0), WBH(1), acc[j], 0, 0, 0))
        W1_EACH(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WAH(1), WBH(0), acc[j], 0, 0, 0))
        W1_EACH(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WAH(0), WBH(0), acc[j], 0, 0, 0))
      } else {
        if constexpr (NS == 3) {
          W1_EACH(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA(1), WB(1), acc[j], 0, 0, 0))
          W1_EACH(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA(0), WB(2), acc[j], 0, 0, 0))
          W1_EACH(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA(2), WB(0), acc[j], 0, 0, 0))
        }
        if constexpr (NS >= 2) {
          W1_EACH(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA(0), WB(1), acc[j], 0, 0, 0))
          W1_EACH(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA(1), WB(0), acc[j], 0, 0, 0))
        }
        W1_EACH(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA(0), WB(0), acc[j], 0, 0, 0))
      }
#undef W1_EACH
#undef WA
#undef WB
#undef WAH
#undef WBH
    }
  }
  if (P.db_part != nullptr) {                   // bias gradient: 16 threads share each channel quad -> LDS -> fixed-order sum
    __syncthreads();
    if (want_db && t < 32) {
      float4 s4 = bred[t];
      for (int k = 1; k < 16; ++k) { const float4 u = bred[t + 32 * k]; s4.x += u.x; s4.y += u.y; s4.z += u.z; s4.w += u.w; }
      const int co = co0 + t * 4;
      if (co < Cout) *reinterpret_cast<float4*>(P.db_part + (size_t)split * Cout + co) = s4;
    }
  }
  const float oscale = NS == 4 ? 1.0f / (yscale * WXSCALE) : 1.0f;
  if constexpr (NS == 4) pdae_sat_report(P.sat, sat_hit);
  float* slab = P.ws + (size_t)split * Cout * C;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ci = ci0 + (cj * 2 + j) * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (co < Cout && ci < C) slab[(size_t)co * C + ci] = NS == 4 ? acc[j][r] * oscale : acc[j][r];
    }
  }
}

static void wgrad1_plan(long long M, int C, int Cout, int& splits, int& tiles_per_split) {
  const int ntiles = (int)((M + WTPIX - 1) / WTPIX);
  const int base = ((Cout + 127) / 128) * ((C + 127) / 128);
  int maxs = ntiles / 4; if (maxs < 1) maxs = 1;
  if (maxs > 256) maxs = 256;
  long long best = -1; int best_s = 1;
  for (int s = 1; s <= maxs; ++s) {
    const int tps = (ntiles + s - 1) / s, sp = (ntiles + tps - 1) / tps;
    if (sp != s) continue;
    const long long rounds = ((long long)base * sp + 255) / 256;
    const long long cost = rounds * (tps + 2);
    if (best < 0 || cost < best) { best = cost; best_s = s; }
  }
  tiles_per_split = (ntiles + best_s - 1) / best_s;
  splits = (ntiles + tiles_per_split - 1) / tiles_per_split;
}

bool conv1x1w_ok(int math, int KH, int KW, int stride, int pad, int up, int C0, int C1, long long M, int Cout) {
  if (math < 1 || KH != 1 || KW != 1 || stride != 1 || pad != 0 || up) return false;
  if ((C0 & 31) || (C1 & 31) || (Cout & 3) || Cout < 32 || C0 + C1 < 32) return false;
  return M >= 16 * WTPIX;
}

size_t conv1x1w_workspace_bytes(long long M, int C, int Cout) {
  int splits, tps;
  wgrad1_plan(M, C, Cout, splits, tps);
  return ((size_t)splits * Cout * C + (size_t)splits * Cout) * sizeof(float);
}

template <int NS> static int launch_w1(const Wgrad1Params& P, hipStream_t s) {
  const size_t smem = (size_t)(2 * WNPL(NS) * WTPIX * W1SY) * sizeof(unsigned short) + W1THREADS * sizeof(float4);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv1x1w_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { pdae_set_error("conv1x1w: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv1x1w_kernel<NS>), dim3(P.splits * P.co_tiles * P.ci_blocks), dim3(W1THREADS), smem, s, P);
  return pdae_launch_status("conv1x1w");
}

int conv1x1w_launch(int math, const float* x0, int C0, const float* x1, int C1, long long M, const float* dy, int Cout, float* dw, int accumulate,
                    float* ws, size_t ws_bytes, hipStream_t s, float** db_part, int* db_rows, const float* dy_amax, float* db) {
  Wgrad1Params P;
  P.dy_amax = dy_amax; P.sat = pdae_sat_counter();
  if (math == 4 && !dy_amax) math = 3;
  P.x0 = x0; P.x1 = x1; P.C0 = C0; P.C1 = C1; P.C = C0 + C1; P.M = M; P.dy = dy; P.Cout = Cout; P.ws = ws;
  P.ntiles = (int)((M + WTPIX - 1) / WTPIX);
  wgrad1_plan(M, P.C, Cout, P.splits, P.tiles_per_split);
  P.co_tiles = (Cout + 127) / 128; P.ci_blocks = (P.C + 127) / 128;
  const size_t need = ((size_t)P.splits * Cout * P.C + (size_t)P.splits * Cout) * sizeof(float);
  P.db_part = db_part ? ws + (size_t)P.splits * Cout * P.C : nullptr;
  if (db_part) { *db_part = P.db_part; *db_rows = P.splits; }
  if (!ws || ws_bytes < need) { pdae_set_error("conv1x1w: workspace too small (%zu < %zu)", ws_bytes, need); return PDAE_EINVAL; }
  int e = math == 1 ? launch_w1<1>(P, s) : (math == 2 ? launch_w1<2>(P, s) : (math == 4 ? launch_w1<4>(P, s) : launch_w1<3>(P, s)));
  if (e) return e;
  if (db_part && db) *db_part = nullptr;
  return igemm_splitk_reduce(ws, dw, (long long)Cout * P.C, P.splits, accumulate, s, db ? P.db_part : nullptr, P.splits, Cout, db);
}
