// Weight gradient of a 3x3 / stride-1 / pad-1 convolution, producer / consumer form (round 6):
//
//   dW[co][tap][ci] = sum over pixels p of  dY[p][co] * X[p + tap][ci]
//
// ONE workgroup of eight waves per CU: waves 0..3 (one per SIMD) only multiply, waves 4..7 (their SIMD partners) only stage.  conv3x3w.hip
// (two workgroups of four waves per CU, every wave staging AND multiplying its 32 x 32 x 4.5-tap share) left the matrix pipe idle for half of
// the time (SQ_VALU_MFMA_BUSY 0.53, profiles/r05_pmc_sq.txt): each wave's matrix phase waits behind its own staging phase and two barriers, the
// overlap with the other workgroup is luck, and a 32 x 32 wave tile reads 0.81 KB of LDS fragments per MFMA.  Here
//   * a matrix wave owns 64 output x 32 input channels x half of the taps (tap half th: taps 0..3 + the centre tap on tile rows 0..3, or taps
//     5..8 + the centre tap on rows 4..7 -- conv3x3w's split): 10 accumulator tiles, every X fragment feeds TWO MFMA chains, every dY fragment
//     4.5 taps: 0.48 KB of LDS per MFMA, 216 MFMAs per 8 x 16-pixel tile and wave, issued back to back from a ring of fragments requested two
//     tap steps ahead; it never touches global memory before the epilogue and waits for nothing but its own LDS reads and ONE barrier per tile;
//   * the workgroup covers 64 output x 64 input channels (matrix wave = (input half b, tap half th)), so a pixel tile is staged for 864 MFMAs
//     instead of 432: 22.8 staged elements per MFMA against 32.3, and dY is re-read C / 64 instead of C / 32 times;
//   * the staging waves hold the NEXT tile's raw fp32 data in registers (up to 20 float4: they have no accumulators), split it into the fp16 /
//     bf16 planes of the other LDS buffer while the matrix waves run, then request the tile after that: global latency never meets an MFMA;
//   * LDS: two buffers of [2 planes][2 input halves][180 patch pixels][32 ci] + [2 planes][128 pixels][64 co] = 2 x 77.3 KB, the layouts of
//     conv3x3w.hip (pixel-major, fragments through the transposing read ds_read_b64_tr_b16, dY rows XOR-swizzled).
// Slabs, split plan and the reduce launch are conv3x3w's ([splits][Cout][9][C], fixed-order splitk_reduce): bit-for-bit the same reduction tree
// per slab element is NOT promised across the two kernels (different pixel-tile partition), the tests bound both against fp64.
// Formats: math 4 (two fp16 planes, default), 2 (two bf16 planes), 1 (one bf16 plane); math 3 (three planes) does not fit and stays on conv3x3w.
//
// Replaces the weight-gradient of F.conv2d(k=3, padding=1) (model/module.py:242,265).
#include "common.h"
#include "igemm.h"
#include "conv3x3w.h"
#include <type_traits>

#define VTHREADS 512
#define VSTG 256                       // staging threads (waves 4..7)
#define VCO 64
#define VCI 64
#define VPW (WTW + 2)                  // 18
#define VNPIX ((WTH + 2) * VPW)        // 180 patch pixels
#define VXPL (VNPIX * 32 + 32)         // ushorts per (plane, input half) X sub-plane: 180 rows of 32 ci + 64 B, so that the two halves of a pixel sit 16 banks apart
#define VX_LD ((VNPIX * 16 + VSTG - 1) / VSTG)      // 12 float4 slots per staging thread (11.25 used)
#define VY_LD (WTPIX * (VCO / 4) / VSTG)            // 8
#define VBUF_US(NS_) (WNPL(NS_) * 2 * VXPL + WNPL(NS_) * WTPIX * WSY)      // ushorts per buffer

template <int NS, bool GN>
__global__ void __launch_bounds__(VTHREADS, 2) conv3x3v_kernel(const WgradParams P) {
  constexpr int NP = WNPL(NS);
  constexpr unsigned XREG_B = (unsigned)(NP * 2 * VXPL * 2);          // bytes of the X region of a buffer
  constexpr unsigned BUF_B = (unsigned)(VBUF_US(NS) * 2);
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool is_mma = wv < 4;

  // XCD-aware bijective block order (conv3x3w.hip): consecutive LOGICAL ids share an XCD (own L2), the input-chunk index runs fastest
  const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  const int ci_chunk = bid % P.ci_chunks; bid /= P.ci_chunks;
  const int co_tile = bid % P.co_tiles; const int split = bid / P.co_tiles;
  const int ci0 = ci_chunk * VCI, co0 = co_tile * VCO;
  const int Cout = P.Cout;
  const int t_beg = split * P.tiles_per_split;
  const int t_end = min(P.ntiles, t_beg + P.tiles_per_split);
  const float yscale = NS == 4 ? w_pow2_scale(*P.dy_amax) : 1.0f;

  // ---------------------------------------------------------------- matrix waves
  const int b = wv & 1, th = (wv >> 1) & 1;          // input-channel half, tap half
  const int li = lane & 31, h = lane >> 5, i16 = lane & 15, g16 = (lane >> 4) & 1;
  // lane-constant parts of the transposing-read addresses (bytes from the start of a buffer); a = 1 is a = 0 with the 64-byte column bit flipped
  const unsigned y_lane0 = XREG_B + (unsigned)(((h * 8 + (i16 >> 2)) * WSY + ((g16 * 16 + (i16 & 3) * 4) ^ WSWZ(i16 >> 2))) * 2);
  const unsigned y_lane1 = XREG_B + (unsigned)(((h * 8 + (i16 >> 2)) * WSY + ((32 + g16 * 16 + (i16 & 3) * 4) ^ WSWZ(i16 >> 2))) * 2);
  const unsigned x_lane = (unsigned)((b * VXPL + (h * 8 + (i16 >> 2)) * WSX + g16 * 16 + (i16 & 3) * 4) * 2);

  auto mma_tile = [&](auto th_c, f32x16 (&acc)[5][2], const unsigned bufb) {
    constexpr int TH = decltype(th_c)::value;
    const unsigned ya0 = bufb + y_lane0, ya1 = bufb + y_lane1, xa = bufb + x_lane;
    uint4 af[2][2][NP], bq[3][NP];
    auto lda = [&](uint4 (&f)[2][NP], int kc) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const unsigned o = (unsigned)((p * WTPIX + kc * 16) * WSY * 2);
        f[0][p] = tr_frag(ya0 + o, ya0 + o + 4 * WSY * 2);
        f[1][p] = tr_frag(ya1 + o, ya1 + o + 4 * WSY * 2);
      }
    };
    auto ldb = [&](uint4 (&f)[NP], int kc, int j) {
      const int tp = j == 4 ? 4 : TH * 5 + j;
      const int dy = tp / 3, dx = tp - dy * 3;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const unsigned ad = xa + (unsigned)((p * 2 * VXPL + ((kc + dy) * VPW + dx) * WSX) * 2);
        f[p] = tr_frag(ad, ad + 4 * WSX * 2);
      }
    };
    lda(af[0], 0); ldb(bq[0], 0, 0); ldb(bq[1], 0, 1);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
      const int n = (kc >> 2) == TH ? 5 : 4;                                   // taps of this k-chunk (tile row)
      const int before = 4 * kc + (TH == 0 ? (kc < 4 ? kc : 4) : (kc > 4 ? kc - 4 : 0));   // steps before this k-chunk
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (j < n) {
          const int cur = (before + j) % 3, nxt = (before + j + 2) % 3;
          if (j + 2 < n) ldb(bq[nxt], kc, j + 2);
          else if (kc + 1 < 8) {
            if (j + 2 == n) lda(af[(kc + 1) & 1], kc + 1);
            ldb(bq[nxt], kc + 1, j + 2 - n);
          }
#define VA(A_, P_) __builtin_bit_cast(bf16x8, af[kc & 1][A_][P_])
#define VB(P_) __builtin_bit_cast(bf16x8, bq[cur][P_])
#define VAH(A_, P_) __builtin_bit_cast(f16x8, af[kc & 1][A_][P_])
#define VBH(P_) __builtin_bit_cast(f16x8, bq[cur][P_])
#define V_EACH(STMT) _Pragma("unroll") for (int a = 0; a < 2; ++a) { STMT; }
          if constexpr (NS == 4) {
            V_EACH(acc[j][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(VAH(a, 0), VBH(1), acc[j][a], 0, 0, 0))
            V_EACH(acc[j][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(VAH(a, 1), VBH(0), acc[j][a], 0, 0, 0))
            V_EACH(acc[j][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(VAH(a, 0), VBH(0), acc[j][a], 0, 0, 0))
          } else {
            if constexpr (NS == 2) {
              V_EACH(acc[j][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(VA(a, 0), VB(1), acc[j][a], 0, 0, 0))
              V_EACH(acc[j][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(VA(a, 1), VB(0), acc[j][a], 0, 0, 0))
            }
            V_EACH(acc[j][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(VA(a, 0), VB(0), acc[j][a], 0, 0, 0))
          }
#undef V_EACH
#undef VA
#undef VB
#undef VAH
#undef VBH
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };

  // ---------------------------------------------------------------- staging waves
  const int ts = t & (VSTG - 1);             // staging thread index (only meaningful in waves 4..7)
  const int qd = ts & 15;                    // channel quad of the workgroup's 64 input channels (the same in every patch slot of this thread)
  const int cthr = ci0 + qd * 4;             // its first channel in the virtual concat
  const bool src_first = !GN || cthr < P.C0;
  const int Cs = GN ? (src_first ? P.C0 : P.C - P.C0) : P.C;          // channels per pixel of the tensor this thread READS
  const float* const xthr = (src_first ? P.x : P.x1) + (GN ? (src_first ? cthr : cthr - P.C0) : cthr);
  const int ush = P.up ? 1 : 0;              // nearest-neighbour 2x upsampling in front of the convolution: stored pixel = logical >> 1 (tile origins are even)
  // tile-invariant halves of the operand addresses (conv3x3w.hip): slot l = patch pixel (t + 256 l) >> 4: offset from the tile's origin pixel * 64 +
  // border bits {1: top halo row, 2: bottom halo row, 4: left halo column, 8: right halo column, 16: never valid}
  int xw[VX_LD];
  float4 xpre[VX_LD], ypre[VY_LD];
  float sat_hit = 0.f;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);     // bias gradient: this thread's column sums (channel quad ts & 15, pixel column ts >> 4)
  const bool want_db = P.db_part != nullptr && ci_chunk == 0;
  {
#pragma unroll
    for (int l = 0; l < VX_LD; ++l) {
      const int pix = (ts + VSTG * l) >> 4;
      int bm = 16, rel = 0;
      if (pix < VNPIX) {
        const int py = pix / VPW, px = pix - py * VPW;
        bm = (py == 0 ? 1 : 0) | (py == WTH + 1 ? 2 : 0) | (px == 0 ? 4 : 0) | (px == WTW + 1 ? 8 : 0);
        rel = (((py - 1) >> ush) * P.Ws + ((px - 1) >> ush)) * Cs;
      }
      xw[l] = rel * 64 + bm;
    }
  }
  const int ycol = ts >> 4, yc4 = ts & 15;
  const int ythr = ycol * Cout + yc4 * 4;            // Cout % 64 == 0: every lane's quad exists
  const int yrow = P.W * Cout;
  float4 gmu, gsc, gsh;                      // GN: coefficients of the tile held in the registers
  unsigned xokm = 0u;                        // GN: validity bit per patch slot (padding is zero AFTER the map)
  auto gload = [&](int tile) {
    const int img = tile / (P.tiles_y * P.tiles_x); const int rem = tile - img * P.tiles_y * P.tiles_x;
    const int ty = rem / P.tiles_x, tx = rem - ty * P.tiles_x;
    const int y0 = ty * WTH, x0 = tx * WTW;
    const int tmask = 16 | (y0 == 0 ? 1 : 0) | (y0 + WTH >= P.H ? 2 : 0) | (x0 == 0 ? 4 : 0) | (x0 + WTW >= P.W ? 8 : 0);
    const float* xb = xthr + ((size_t)(img * P.Hs + (y0 >> ush)) * P.Ws + (x0 >> ush)) * Cs;
    if constexpr (GN) {
      const float* cf = P.coef + (size_t)img * P.C + cthr;
      const size_t NC = (size_t)P.N * P.C;
      gmu = *reinterpret_cast<const float4*>(cf); gsc = *reinterpret_cast<const float4*>(cf + NC); gsh = *reinterpret_cast<const float4*>(cf + 2 * NC);
      xokm = 0u;
    }
#pragma unroll
    for (int l = 0; l < VX_LD; ++l) {            // unconditional loads from clamped addresses, zeroed afterwards
      const bool ok = (xw[l] & tmask) == 0;
      const float4 v = *reinterpret_cast<const float4*>(xb + (ok ? xw[l] >> 6 : 0));
      if constexpr (GN) { xpre[l] = v; xokm |= (ok ? 1u : 0u) << l; }
      else xpre[l] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* yb = P.dy + ((size_t)(img * P.H + y0) * P.W + x0) * Cout + co0 + ythr;
#pragma unroll
    for (int l = 0; l < VY_LD; ++l) ypre[l] = *reinterpret_cast<const float4*>(yb + l * yrow);
  };
  auto lstore = [&](const unsigned bufb) {
    unsigned short* const sX = smem + (bufb >> 1);
    unsigned short* const sY = sX + NP * 2 * VXPL;
#pragma unroll
    for (int l = 0; l < VX_LD; ++l) {
      const int pix = (ts + VSTG * l) >> 4;
      if (pix < VNPIX) {
        if constexpr (GN) xpre[l] = w_gn_map(xpre[l], (xokm >> l) & 1u, gmu, gsc, gsh, P.act);
        if constexpr (NS == 4) pdae_f16_amax4(xpre[l], WXSCALE, sat_hit);
        unsigned u[NP], v[NP];
        w_split2<NS>(xpre[l].x, xpre[l].y, u, WXSCALE); w_split2<NS>(xpre[l].z, xpre[l].w, v, WXSCALE);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(&sX[(p * 2 + (qd >> 3)) * VXPL + pix * WSX + (qd & 7) * 4]) = make_uint2(u[p], v[p]);
      }
    }
    if (want_db) {
#pragma unroll
      for (int l = 0; l < VY_LD; ++l) { bsum.x += ypre[l].x; bsum.y += ypre[l].y; bsum.z += ypre[l].z; bsum.w += ypre[l].w; }
    }
#pragma unroll
    for (int l = 0; l < VY_LD; ++l) {
      const int pix = (ts + VSTG * l) >> 4;
      unsigned u[NP], v[NP];
      w_split2<NS>(ypre[l].x, ypre[l].y, u, yscale); w_split2<NS>(ypre[l].z, ypre[l].w, v, yscale);
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(&sY[(p * WTPIX + pix) * WSY + ((yc4 * 4) ^ WSWZ(pix))]) = make_uint2(u[p], v[p]);
    }
  };

  // ---------------------------------------------------------------- the pipeline: tile i is multiplied out of buffer i & 1 while tile i + 1 is
  // split into the other buffer and tile i + 2 is requested; one barrier per tile.  The two roles are two separate straight paths through the
  // kernel (each with its own copies of the barriers: same count on both), so that the accumulators of one and the raw-data registers of the
  // other are never live together.
  float* const fsm = reinterpret_cast<float*>(smem);
  float4* const bred = reinterpret_cast<float4*>(smem + 32768);      // epilogue scratch, 64 KB into the (dead) operand buffers: behind the centre-tap exchange (4 x 4 KB)
  if (!is_mma) {
    if (t_beg < t_end) { gload(t_beg); lstore(0u); }
    if (t_beg + 1 < t_end) gload(t_beg + 1);
    __syncthreads();
    for (int tile = t_beg; tile < t_end; ++tile) {
      const unsigned cur_b = ((tile - t_beg) & 1) ? BUF_B : 0u;
      if (tile + 1 < t_end) {
#ifndef PDAE_V_PROBE_NOSTAGE          // timing probes (tools/probe_build.py): wrong results by design
        lstore(cur_b ^ BUF_B);
#endif
#ifndef PDAE_V_PROBE_NOLOAD
        if (tile + 2 < t_end) gload(tile + 2);
#endif
      }
      __syncthreads();
    }
    if (want_db) bred[ts] = bsum;
    __syncthreads();
    if constexpr (NS == 4) pdae_sat_report(P.sat, sat_hit);
    if (want_db && ts < 16) {                 // bias gradient: 16 threads share each channel quad -> fixed-order sum
      float4 s4 = bred[ts];
      for (int k = 1; k < 16; ++k) { const float4 u = bred[ts + 16 * k]; s4.x += u.x; s4.y += u.y; s4.z += u.z; s4.w += u.w; }
      *reinterpret_cast<float4*>(P.db_part + (size_t)split * Cout + co0 + ts * 4) = s4;
    }
    return;
  }

  // the whole matrix path per tap half (loop AND epilogue inside the branch: a join of the two 216-MFMA bodies inside one loop would make the
  // register allocator shuffle the 160 accumulator registers at every tile)
  auto mma_path = [&](auto th_c) {
    constexpr int TH = decltype(th_c)::value;
    f32x16 acc[5][2];                          // [tap slot j][output-channel tile a]; j = 4: this wave's half of the centre tap
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][a][r] = 0.f;
    __builtin_amdgcn_s_setprio(1);            // the matrix wave's few issue slots come first: its SIMD partner only fills the gaps (MI355X_MICROARCH.md, two waves per SIMD)
    __syncthreads();
    for (int tile = t_beg; tile < t_end; ++tile) {
#ifndef PDAE_V_PROBE_NOMMA
      mma_tile(th_c, acc, ((tile - t_beg) & 1) ? BUF_B : 0u);
#endif
      __syncthreads();
    }
    // epilogue (the operand buffers are dead): the two halves of the centre tap meet in LDS: wave (b, th = 1) hands its acc[4] to wave (b, th = 0)
    if constexpr (TH == 1) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) fsm[((b * 2 + a) * 16 + r) * 64 + lane] = acc[4][a][r];
    }
    __syncthreads();
    if constexpr (TH == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[4][a][r] += fsm[((b * 2 + a) * 16 + r) * 64 + lane];
    }
    const float oscale = NS == 4 ? 1.0f / (yscale * WXSCALE) : 1.0f;      // exact: both scales are powers of two
    float* const slab = P.ws + (size_t)split * Cout * 9 * P.C;            // layout [Cout][9][C]
#pragma unroll
    for (int j = 0; j < 5 - TH; ++j) {
      const int tp = j == 4 ? 4 : TH * 5 + j;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          slab[((size_t)co * 9 + tp) * P.C + ci0 + b * 32 + li] = NS == 4 ? acc[j][a][r] * oscale : acc[j][a][r];
        }
    }
  };
  if (th == 0) mma_path(std::integral_constant<int, 0>{}); else mma_path(std::integral_constant<int, 1>{});
}

// split of the pixel tiles over workgroups: ONE per CU, so the grid should fill 256 slots in whole rounds: minimise rounds x (tiles + 2), the 2 =
// a workgroup's exposed first staging and its slab write
static void wgradv_plan(int N, int H, int W, int C, int Cout, int& splits, int& tiles_per_split) {
  const int ntiles = N * (H / WTH) * (W / WTW);
  const int base = (Cout / VCO) * (C / VCI);
  int maxs = ntiles / 4; if (maxs < 1) maxs = 1;      // at least 4 tiles per workgroup
  if (maxs > 128) maxs = 128;
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

// shapes of this form (the caller has checked conv3x3w_ok): whole 64-channel blocks on both sides, 16-pixel-wide tiles, a two-plane format
bool conv3x3v_ok(int math, int C, int H, int W, int N, int Cout) {
  if (math != 1 && math != 2 && math != 4) return false;
  if ((C % VCI) || (Cout % VCO) || (W % WTW) || (H % WTH)) return false;
  return (long long)N * (H / WTH) * (W / WTW) >= 64;
}

size_t conv3x3v_workspace_bytes(int N, int H, int W, int C, int Cout) {
  int splits, tps;
  wgradv_plan(N, H, W, C, Cout, splits, tps);
  return ((size_t)splits * Cout * 9 * C + (size_t)splits * Cout) * sizeof(float);
}

template <int NS, bool GN> static int launch_v(const WgradParams& P, hipStream_t s) {
  const size_t smem = (size_t)2 * VBUF_US(NS) * sizeof(unsigned short);
  static_assert(2 * VBUF_US(4) * 2 <= 163840, "two operand buffers must fit the CU's LDS");
  static_assert(2 * VBUF_US(4) * 2 >= 65536 + VSTG * 16 && 2 * VBUF_US(1) * 2 >= 65536 + VSTG * 16, "epilogue scratch lives inside the operand buffers");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3v_kernel<NS, GN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { pdae_set_error("conv3x3v: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3v_kernel<NS, GN>), dim3(P.splits * P.co_tiles * P.ci_chunks), dim3(VTHREADS), smem, s, P);
  return pdae_launch_status("conv3x3v");
}

// same contract as conv3x3w_launch (which routes here): slabs + bias partials in ws, the reduce launch finishes dw (and db)
int conv3x3v_launch(int math, const float* x, int N, int Hs, int Ws, int C, int H, int W, int up, const float* dy, int Cout, float* dw,
                    int accumulate, float* ws, size_t ws_bytes, hipStream_t s, float** db_part, int* db_rows, const float* dy_amax, float* db,
                    const float* x1, int C0, const float* coef, int act) {
  WgradParams P;
  P.x1 = x1; P.C0 = x1 ? C0 : C; P.coef = coef; P.act = act;
  P.dy_amax = dy_amax; P.sat = pdae_sat_counter(); P.stagger = 0;
  P.x = x; P.N = N; P.Hs = Hs; P.Ws = Ws; P.C = C; P.H = H; P.W = W; P.up = up; P.dy = dy; P.Cout = Cout; P.ws = ws;
  P.tiles_x = W / WTW; P.tiles_y = H / WTH; P.ntiles = N * P.tiles_x * P.tiles_y;
  wgradv_plan(N, H, W, C, Cout, P.splits, P.tiles_per_split);
  P.co_tiles = Cout / VCO; P.ci_chunks = C / VCI;
  const size_t need = ((size_t)P.splits * Cout * 9 * C + (size_t)P.splits * Cout) * sizeof(float);
  P.db_part = db_part ? ws + (size_t)P.splits * Cout * 9 * C : nullptr;
  if (db_part) { *db_part = P.db_part; *db_rows = P.splits; }
  if (!ws || ws_bytes < need) { pdae_set_error("conv3x3v: workspace too small (%zu < %zu)", ws_bytes, need); return PDAE_EINVAL; }
  int e;
  if (coef) e = math == 1 ? launch_v<1, true>(P, s) : (math == 2 ? launch_v<2, true>(P, s) : launch_v<4, true>(P, s));
  else e = math == 1 ? launch_v<1, false>(P, s) : (math == 2 ? launch_v<2, false>(P, s) : launch_v<4, false>(P, s));
  if (e) return e;
  if (db_part && db) *db_part = nullptr;      // the bias gradient's final sum rides in the reduce launch
  return igemm_splitk_reduce(ws, dw, (long long)Cout * 9 * C, P.splits, accumulate, s, db ? P.db_part : nullptr, P.splits, Cout, db);
}
