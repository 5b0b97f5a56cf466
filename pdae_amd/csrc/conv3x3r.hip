// 3x3 / stride-1 / pad-1 convolution (forward and data-gradient) for gfx950 -- PERSISTENT form of the patch kernel with a DEFERRED epilogue,
// for layers with at least a chip-full of 16 x 16-pixel x 128-channel tiles.
//
// What the counters said about one-tile-per-workgroup forms (profiles/r03_patch_kernels.txt): a launch is rounds of {prologue: first patch
// from HBM, main loop, epilogue: 64 MB of output per round}, every CU in the same phase at the same time.  While the main loops run HBM idles,
// while the epilogues run the matrix pipes idle and all 256 CUs queue on the write bandwidth: with loads, conversions and weights compiled
// out, a one-workgroup-per-CU prototype with 256-pixel x 64-channel wave tiles still needed 1.3x its MFMA time; conv3x3p hides part of it
// behind a second workgroup per CU and pays with two instruction streams per SIMD (MFMA pipe busy 0.60).  Here each CU keeps ONE 4-wave workgroup for the whole launch and walks a
// list of tiles; the epilogue of tile i runs INSIDE the main loop of tile i+1:
//   * wave tile 128 pixels x 64 output channels = 8 accumulators of 32 x 32 (128 registers); TWO sets live in the 256 AGPRs of a one-wave-per-SIMD
//     kernel: the set being accumulated and the set being drained (acc -> private LDS transpose tile -> float4 + bias / residual -> global,
//     plus the GroupNorm partial statistics), one accumulator tile every four 12-MFMA units of the next tile's first chunk;
//   * the first patch of tile i+1 is simply the next entry of the same prefetch pipeline that feeds chunk after chunk: no prologue after the first;
//   * the LDS patch is double buffered (2 x 57.6 KB for two fp16 planes): conversion results go straight into the other buffer, a chunk ends with
//     ONE barrier and no staging phase;
//   * weight fragments: ring of three k-steps (prefetch distance two: a k-step is only 24 MFMAs here), patch fragments: ring of two 2-group units;
//   * every HBM-latency load (patch prefetch, residual of the drain) is issued right behind a weight-fragment batch, in bursts, because the
//     vector-memory path returns in order and a weight load queued behind an HBM miss inherits its latency;
//   * straight-line units: unused operands (no residual, no bias, nothing to drain yet, padding pixels, padded skip chunks) are buffer loads /
//     stores against an EMPTY resource or an out-of-range offset -- the hardware returns zeros / drops the store -- instead of branches.
// Per 24 MFMAs a wave issues 8 ds_read_b128 + 4 buffer loads (0.5 per MFMA; conv3x3p: 0.83) and nothing competes for the SIMD's issue port.
// Operand formats, prepared-weight layout, fused GroupNorm input, fused 1x1 skip chunks, epilogue arithmetic and the accumulation order per
// output element are those of conv3x3p: results are bit-identical (tests/test_conv3x3r_gpu.py).
// Replaces F.conv2d(k=3, padding=1) of model/module.py:242,265 (+ nearest upsample :169) and its input gradient on the large layers.
#include <stdlib.h>

#include "common.h"
#include "igemm.h"
#include "conv3x3p.h"

#define RTH 16                                              // tile rows
#define RNPIX ((RTH + 2) * PPW)                             // 360 patch pixels (18 rows x 20-pixel pitch)
#define RTHREADS 256
#define RLD ((RNPIX * 8 + RTHREADS - 1) / RTHREADS)         // 12 float4 of the patch per thread and chunk
#define RBN 128
#define RROWS 384                                           // pixel rows allocated per LDS plane: every staging slot (12 x 32) has a home, no store guard
#define RPLANE_B (PPLANE(RROWS) * 2)                        // bytes per LDS plane (30720)
// units of a main chunk that convert + store a prefetched float4: the three loads issued at unit 6 g are converted at units 6 g + 10 .. + 12, so
// that at most two load groups (24 registers) are live at a time -- with all twelve converted behind the last load (units 22 .. 33) the 48
// registers of the whole patch were live at once and the fused-GroupNorm instantiation spilled 20 VGPRs to scratch (68 bytes per lane)
#define RCV_AT(U) ((U) >= 10 && ((U) - 10) % 6 < 3 && ((U) - 10) / 6 < 4)
#define RCV_SLOT(U) ((((U) - 10) / 6) * 3 + ((U) - 10) % 6)
#define ROOB 0xFFFFFFF0u                                    // per-lane buffer offset beyond every resource: load -> 0, store -> dropped

typedef unsigned r_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned r_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const r_u32x4* r_lds_u4;
typedef __attribute__((address_space(3))) float* r_lds_f;
typedef float r_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const r_f32x4* r_lds_f4;
typedef __attribute__((address_space(3))) r_u32x2* r_lds_u2;

// issue pattern of one unit: behind each of the 12 MFMAs up to NV scalar / vector ALU instructions, then LDS / vector-memory slots
#define PDAE_R_PATTERN(NV)                                                                                  \
  _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
    __builtin_amdgcn_sched_group_barrier(0x006, NV, 0);                                                    \
    if (i_ < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                     \
    if (i_ >= 4) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);                                        \
    __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);                                                     \
  }

// tile index -> (image, first row, first column, first output channel); channel tile fastest: workgroups that run at the same time share patches
__device__ __forceinline__ void r_decode(int tile, int tiles_n, int tiles_x, int tiles_y, int& img, int& y0, int& x0, int& n0) {
  const int tn = tile % tiles_n; tile /= tiles_n;
  const int tx = tile % tiles_x; tile /= tiles_x;
  const int ty = tile % tiles_y; tile /= tiles_y;
  img = tile; y0 = ty * RTH; x0 = tx * PTW; n0 = tn * RBN;
}

template <int NS, bool GN>
__global__ void __launch_bounds__(RTHREADS, 1) conv3x3r_kernel(const PatchParams P) {
  constexpr int NP = NPL(NS);
  constexpr unsigned BUF_B = NP * RPLANE_B;                 // one patch buffer
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];

  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, h = lane >> 5;
  const int wm = wv >> 1, wn = wv & 1;                      // 8-row band of the tile, 64-channel half
  const int qd = t & 7;                                     // this thread's channel quad of every staged pixel
  const int C = P.C, C1 = C - P.C0;
  const int nmain = C >> 5, nx = P.nx, nxp = ((nx + 2) / 3) * 3;      // skip chunks are walked in groups of three (weight-ring period): padded with empty ones
  const int ntiles = P.N * P.tiles_y * P.tiles_x * P.tiles_n, G = gridDim.x;

  // ---- patch staging: (patch row << 8 | patch column) of every prefetch slot; slots that are not patch pixels (pitch padding, beyond the
  // patch) carry row 4000: they fail the image-bounds test like any padding pixel (H < 2048: launch check), no separate validity bit
  int pyx[RLD];
#pragma unroll
  for (int l = 0; l < RLD; ++l) {
    const int pix = (t >> 3) + 32 * l, py = pix / PPW, px = pix - py * PPW;
    pyx[l] = (pix < RNPIX && px < PTW + 2) ? ((py << 8) | px) : (4000 << 8);
  }
  const float ascale = NS == 4 ? (P.amax ? p_pow2_scale(*P.amax) : PASCALE) : 1.0f;
  float sat_hit = 0.f;

  // Buffer resources are built where they are used, from a pointer and a byte count: twelve live 4-dword descriptors were most of the SGPR
  // file.  All tensors are < RALL bytes (launch check), so one bound serves every real tensor and an operand that is absent (no residual, no
  // bias, nothing to drain yet, a padded skip chunk) gets 0 bytes: loads return 0, stores are dropped.  Padding lanes use the offset ROOB >= RALL.
#define RALL 0xFFFFFFEFu
#define R_RS(PTR, BYTES) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>((const void*)(PTR)), 0, (int)(BYTES), 0x00020000)
  const float* const x1_ = P.x1 ? P.x1 : P.x;
  const float* const s0_ = P.s0 ? P.s0 : P.x;
  const float* const s1_ = P.s1 ? P.s1 : P.x;
  const float* const extra_ = P.res_mode ? P.res : P.y;     // residual OR (accumulate) the previous contents of y
  const unsigned extra_on = (P.res_mode || P.accumulate) ? RALL : 0u, stat_on = P.stat_part ? RALL : 0u;
  const unsigned bias_b = P.bias ? RALL : 0u, biasx_b = P.bias_x ? RALL : 0u;
  const float* const bias_ = P.bias ? P.bias : P.x;
  const float* const biasx_ = P.bias_x ? P.bias_x : P.x;
  const float* const stat_ = P.stat_part ? (const float*)P.stat_part : P.x;
  const unsigned short* const wp_ = P.wp;
  const unsigned short* const wps_ = P.wps ? P.wps : P.wp;

  // ---- the chunk being PREFETCHED: tile origin, source tensor, channel offset; and the tile sequence
  // slot of a tile: 0 .. nmain - 1 main chunks, nmain .. nmain + nxp - 1 skip chunks (>= nmain + nx: empty)
  int n_y0, n_x0, n_pixbase, n_img;
  const float* n_ptr = P.x;
  unsigned n_bytes = RALL;
  unsigned n_ldb = 0, n_cb = 0;
  bool n_raw = false;
  float4 apre[RLD];
  float4 gmu, gsc, gsh;
  // (closures must not capture other closures: the chain P <- closure <- closure survives SROA and drags every kernel argument through scratch)
#define PDAE_R_DECODE(TILE, IMG, Y0, X0, N0) r_decode(TILE, P.tiles_n, P.tiles_x, P.tiles_y, IMG, Y0, X0, N0)
  // source of prefetch position (tile, slot); a position behind the last tile re-reads the last one (branch-free bodies), nothing uses it.
  // A macro, not a closure: "cond ? captured_a : captured_b" inside a lambda becomes a load at a DYNAMIC offset of the closure object, which
  // survives SROA and then drags every captured variable -- all buffer resources, all kernel arguments -- through scratch and VGPRs
  int n_ws = 0, n_sh = 0;
#define PDAE_R_CHUNK_SRC(TILE, SLOT)                                                                         \
  {                                                                                                         \
    const int tile_ = (TILE), slot_ = (SLOT);                                                               \
    /* origin of the tile being prefetched: this tile or the next one (decoded once per tile: three divisions are ~130 scalar instructions) */ \
    const bool nxt_ = tile_ != tile;                                                                        \
    n_img = nxt_ ? x_img : c_img; n_y0 = nxt_ ? x_y0 : c_y0; n_x0 = nxt_ ? x_x0 : c_x0;                     \
    n_raw = slot_ >= nmain;                                                                                 \
    if (n_raw) {                                                                                            \
      const int c_ = (slot_ - nmain) << 5;                                                                  \
      const bool first_ = c_ < P.Cs0, live_ = slot_ < nmain + nx;                                           \
      n_ptr = first_ ? s0_ : s1_; n_bytes = live_ ? RALL : 0u;                                              \
      n_ldb = (unsigned)(first_ ? P.Cs0 : P.Cs1) * 4u; n_cb = (unsigned)(first_ ? c_ : c_ - P.Cs0) * 4u;    \
      n_pixbase = n_img * P.H * P.W;      /* skip tensors live at the output resolution (no up-sampling with a fused skip: launch check) */ \
      n_ws = P.W; n_sh = 0;                                                                                 \
    } else {                                                                                                \
      const int c_ = slot_ << 5;                                                                            \
      const bool first_ = c_ < P.C0;      /* C0 == C for a single source; C0 % 32 == 0 otherwise (launch check) */ \
      n_ptr = first_ ? P.x : x1_; n_bytes = RALL;                                                           \
      n_ldb = (unsigned)(first_ ? P.C0 : C1) * 4u; n_cb = (unsigned)(first_ ? c_ : c_ - P.C0) * 4u;         \
      n_pixbase = n_img * P.Hs * P.Ws;                                                                      \
      n_ws = P.Ws; n_sh = up_sh;                                                                            \
      if constexpr (GN) {                                                                                   \
        const size_t NC_ = (size_t)P.N * C;                                                                 \
        const float* cf_ = P.coef + (size_t)n_img * C + c_ + qd * 4;                                        \
        gmu = *reinterpret_cast<const float4*>(cf_); gsc = *reinterpret_cast<const float4*>(cf_ + NC_); gsh = *reinterpret_cast<const float4*>(cf_ + 2 * NC_); \
      }                                                                                                     \
    }                                                                                                       \
  }
  const int up_sh = P.up ? 1 : 0;
  auto gload_one = [&](int l) {
    const int v = pyx[l];
    const int ly = n_y0 - 1 + (v >> 8), lx = n_x0 - 1 + (v & 255);
    const bool ok = ((unsigned)ly < (unsigned)P.H) & ((unsigned)lx < (unsigned)P.W);       // "&": a short-circuit "&&" becomes an exec-masked block
    const unsigned pixel = (unsigned)(n_pixbase + (ly >> n_sh) * n_ws + (lx >> n_sh));
    unsigned vin = pixel * n_ldb + (unsigned)(qd * 16);
    asm("" : "+v"(vin));                                    // computed unconditionally: left to itself hipcc wraps the multiply into an exec-masked block, and a unit must stay ONE basic block
    const unsigned voff = ok ? vin : ROOB;
    apre[l] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(R_RS(n_ptr, n_bytes), (int)voff, (int)n_cb, 0));
  };
  // LDS: [patch buffer 0 | patch buffer 1 | 4 x 32 x EPW floats of drain transpose tiles]
  const unsigned lds0 = (unsigned)(size_t)smem;
  unsigned cur = 0;                                         // buffer holding the chunk being multiplied; the prefetched one goes to cur ^ 1
  const unsigned wr_lane = lds0 + (unsigned)((((t >> 3) * PLDH) + qd * 4) * 2);
  // GroupNorm / AdaGN + SiLU map (GN), fp16-window tracking, operand split of prefetch slot l and its store into the other patch buffer
  auto convert_store = [&](int l) {
    float4 v = apre[l];
    const float sc = n_raw ? 1.0f : ascale;                 // skip chunks carry the RAW residual stream: unit scale, the 2^4 sits in their weights
    if constexpr (GN) {
      const int pv = pyx[l];
      const int ly = n_y0 - 1 + (pv >> 8), lx = n_x0 - 1 + (pv & 255);
      const bool on = ((unsigned)ly < (unsigned)P.H) & ((unsigned)lx < (unsigned)P.W) & !n_raw;      // padding pixels stay zero AFTER the map
      float4 m;
      m.x = gsc.x * (v.x - gmu.x) + gsh.x; m.y = gsc.y * (v.y - gmu.y) + gsh.y;
      m.z = gsc.z * (v.z - gmu.z) + gsh.z; m.w = gsc.w * (v.w - gmu.w) + gsh.w;
      m.x = p_silu(m.x); m.y = p_silu(m.y); m.z = p_silu(m.z); m.w = p_silu(m.w);      // act == 1 (launch check)
      v.x = on ? m.x : v.x; v.y = on ? m.y : v.y; v.z = on ? m.z : v.z; v.w = on ? m.w : v.w;
    }
    if constexpr (NS == 4) pdae_f16_amax4(v, sc, sat_hit);
    unsigned a[NP], b[NP];
    p_split2<NS>(v.x, v.y, a, sc);
    p_split2<NS>(v.z, v.w, b, sc);
    const unsigned d = wr_lane + (cur ^ 1u) * BUF_B + (unsigned)(l * 32 * PLDH * 2);
#pragma unroll
    for (int p = 0; p < NP; ++p) { const r_u32x2 w2 = {a[p], b[p]}; *(r_lds_u2)(size_t)(d + (unsigned)(p * RPLANE_B)) = w2; }
  };

  // ---- fragments.  MFMA row i of pixel group a <-> pixel (band row i / 4, a * 4 + i % 4) of the wave's 8 x 16 band
  const unsigned a_lane = lds0 + (unsigned)(PSLOT((wm * 8 + (li >> 2)) * PPW + (li & 3), h) * 2);
  unsigned abase = a_lane;                                  // + cur * BUF_B, refreshed per chunk, opaque to the compiler: every fragment read is base + immediate
  uint4 fa[2][2][NP];                                       // [ring slot][group of the unit][plane]
  auto lda = [&](uint4 (&af)[2][NP], int tap, int kc, int u) {
    const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned off = (unsigned)((dy * PPW + dx + (u * 2 + j) * 4) * (PLDH * 2) + kc * 32);
#pragma unroll
      for (int p = 0; p < NP; ++p) af[j][p] = __builtin_bit_cast(uint4, *(r_lds_u4)(size_t)(abase + off + (unsigned)(p * RPLANE_B)));
    }
  };
  // weight fragments of one k-step: one uint4 per lane, channel tile (2 per wave) and plane, straight from L2
  //   main chunks: wp [p][chunk][tap][kc][nt][lane],  skip chunks: wps [p][chunk - nmain][kc][nt][lane]
  const size_t plane_main = (size_t)nmain * 18 * P.NT * 512, plane_skip = (size_t)nx * 2 * P.NT * 512;      // bf16 elements per plane
  const int lane16 = lane * 16;
  uint4 qb[3][2][NP];                                       // [k-step mod 3][channel tile][plane]
  int c_nt0 = 0, x_nt0 = 0;                                 // first channel tile of this wave in the tile being multiplied / the tile after it
  const unsigned ps_main2 = (unsigned)(plane_main * 2), ps_skip2 = (unsigned)(plane_skip * 2);      // bytes per plane
  auto ldb = [&](uint4 (&bq)[2][NP], const unsigned short* base, unsigned ps2, int tileidx) {        // tileidx: 1 KB fragment tiles from the plane start
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(base), 0, 0x7fffffff, 0x00020000);
    const unsigned soff = (unsigned)tileidx * 1024u;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int p = 0; p < NP; ++p) bq[ct][p] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srd, lane16, (int)(soff + ct * 1024 + p * ps2), 0));
  };
  // (source selection stays OUTSIDE closures: "cond ? captured_a : captured_b" in a lambda is a dynamic closure offset, see PDAE_R_CHUNK_SRC)
#define PDAE_R_LDB_MAIN(BQ, CHUNK, STEP, NT0) ldb(BQ, wp_, ps_main2, ((CHUNK) * 18 + (STEP)) * P.NT + (NT0))
#define PDAE_R_LDB_SKIP(BQ, R, KC, NT0) ldb(BQ, wps_, ps_skip2, ((((R) < nx ? (R) : nx - 1) << 1) + (KC)) * P.NT + (NT0))      /* padding chunks: last real weights, zero patch */
  // weight fragments of step S (0, 1) of whatever FOLLOWS main chunk CHUNK: the next main chunk, the first skip chunk, or the next tile -- branch-free
#define PDAE_R_LDB_AFTER_MAIN(BQ, CHUNK, S)                                                                  \
    { const bool more_ = (CHUNK) + 1 < nmain, sk_ = !more_ && nx > 0;                                       \
      ldb(BQ, sk_ ? wps_ : wp_, sk_ ? ps_skip2 : ps_main2, ((more_ ? ((CHUNK) + 1) * 18 : 0) + (S)) * P.NT + ((more_ || sk_) ? c_nt0 : x_nt0)); }
  // ... of group step G2 >= 6 behind the skip-chunk group at R0: the next group, or the next tile
#define PDAE_R_LDB_AFTER_SKIP(BQ, R0, G2)                                                                    \
    { const bool more_ = (R0) + 3 < nxp; const int r_ = (R0) + 3 < nx ? (R0) + 3 : nx - 1;                  \
      ldb(BQ, more_ ? wps_ : wp_, more_ ? ps_skip2 : ps_main2, ((more_ ? (r_ << 1) : 0) + ((G2) - 6)) * P.NT + (more_ ? c_nt0 : x_nt0)); }

  f32x16 acc[4][2], dacc[4][2];                             // the tile being accumulated | the tile being drained
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[a][ct][r] = 0.f; dacc[a][ct][r] = 0.f; }

  // the 12 (6 / 2) MFMAs of one unit: product-major, a dependent pair is 4 issues apart.  ZC: first MFMAs of a tile start from C = 0
  auto mma = [&](const uint4 (&af)[2][NP], const uint4 (&bq)[2][NP], int u, bool zc) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define PDAE_RA(P_) __builtin_bit_cast(bf16x8, af[j][P_])
#define PDAE_RB(P_) __builtin_bit_cast(bf16x8, bq[ct][P_])
#define PDAE_RAH(P_) __builtin_bit_cast(f16x8, af[j][P_])
#define PDAE_RBH(P_) __builtin_bit_cast(f16x8, bq[ct][P_])
#define PDAE_R_EACH(STMT) _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) { STMT; }
    if constexpr (NS == 4) {                  // fp16 planes: cross terms first, leading term last (conv3x3p order)
      PDAE_R_EACH(acc[u * 2 + j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_RAH(0), PDAE_RBH(1), zc ? zero : acc[u * 2 + j][ct], 0, 0, 0))
      PDAE_R_EACH(acc[u * 2 + j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_RAH(1), PDAE_RBH(0), acc[u * 2 + j][ct], 0, 0, 0))
      PDAE_R_EACH(acc[u * 2 + j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_RAH(0), PDAE_RBH(0), acc[u * 2 + j][ct], 0, 0, 0))
    } else if constexpr (NS == 2) {
      PDAE_R_EACH(acc[u * 2 + j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_RA(0), PDAE_RB(1), zc ? zero : acc[u * 2 + j][ct], 0, 0, 0))
      PDAE_R_EACH(acc[u * 2 + j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_RA(1), PDAE_RB(0), acc[u * 2 + j][ct], 0, 0, 0))
      PDAE_R_EACH(acc[u * 2 + j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_RA(0), PDAE_RB(0), acc[u * 2 + j][ct], 0, 0, 0))
    } else {
      PDAE_R_EACH(acc[u * 2 + j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_RA(0), PDAE_RB(0), zc ? zero : acc[u * 2 + j][ct], 0, 0, 0))
    }
#undef PDAE_R_EACH
#undef PDAE_RA
#undef PDAE_RB
#undef PDAE_RAH
#undef PDAE_RBH
  };

  // ---- deferred epilogue ("drain") of the previous tile: accumulator tile j = ct * 4 + a4 in three stages
  //   L(j): the 4 float4 of residual / previous contents, from HBM;  W(j): accumulators -> the wave's LDS transpose tile;
  //   S(j): float4 rows back, * scale + bias + residual, stored, statistics summed; after a4 == 3 the band's (sum, sum of squares) are written
  const float oscale = NS == 4 ? P.woscale / ascale : 1.0f;      // exact: powers of two
  const int er = lane >> 3, ec = (lane & 7) * 4;
  const unsigned tw = lds0 + 2u * BUF_B + (unsigned)(wv * 32 * EPW * 4);
  const unsigned tw_w = tw + (unsigned)((4 * h * EPW + li) * 4), tw_r = tw + (unsigned)((er * EPW + ec) * 4);
  const unsigned lane_y = (unsigned)((((er >> 2) * P.W + (er & 3)) * P.Nout + ec) * 4);        // byte offset of the lane's float4 in a 2-row x 4-pixel sub-block
  const unsigned lane_x = P.res_mode == 2 ? (unsigned)((((er & 3) >> 1) * P.Nout + ec) * 4) : lane_y;      // ... in the half-resolution residual
  const unsigned lane_st = lane < 8 ? (unsigned)(lane * 8) : ROOB;
  unsigned d_live = 0u;                                      // byte bound of the drain's tensors: 0 = nothing to drain yet (loads give 0, stores are dropped)
  // byte offsets of the drained tile (wave's band and channel half), set when a tile is handed over; per accumulator tile only constant steps
  // are added (no multiplications, no res_mode branches inside a unit)
  unsigned d_rb4 = 0u, d_xb4 = 0u, d_sb8 = 0u, d_col4 = 0u;
  const int rsh = P.res_mode == 2 ? 1 : 0;                  // half-resolution residual: rows / columns >> 1
  const unsigned y_it = (unsigned)(2 * P.W * P.Nout * 4), y_a4 = (unsigned)(4 * P.Nout * 4);                 // output: 2 rows per `it`, 4 pixels per group
  const unsigned x_it = rsh ? (unsigned)((P.W >> 1) * P.Nout * 4) : y_it, x_a4 = rsh ? (unsigned)(2 * P.Nout * 4) : y_a4;
  float4 rv[2][4], bias4 = make_float4(0.f, 0.f, 0.f, 0.f), bias_n = make_float4(0.f, 0.f, 0.f, 0.f);
  float st1 = 0.f, st2 = 0.f;
  auto drain_L = [&](int j) {
    const int ct = j >> 2, a4 = j & 3;
    if (a4 == 0) {                                          // bias of this channel tile (two tiny loads against possibly-empty resources)
      const float4 u0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(R_RS(bias_, bias_b), ec * 4, (int)(d_col4 + ct * 128), 0));
      const float4 u1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(R_RS(biasx_, biasx_b), ec * 4, (int)(d_col4 + ct * 128), 0));
      bias_n = make_float4(u0.x + u1.x, u0.y + u1.y, u0.z + u1.z, u0.w + u1.w);      // committed by S(j): S of the previous channel tile may still be pending
    }
#pragma unroll
    for (int it = 0; it < 4; ++it)
      rv[j & 1][it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(R_RS(extra_, d_live & extra_on), (int)lane_x,
                                                                                        (int)(d_xb4 + it * x_it + a4 * x_a4 + ct * 128), 0));
  };
  auto drain_W = [&](int j) {
    const int ct = j >> 2, a4 = j & 3;
#pragma unroll
    for (int r = 0; r < 16; ++r) *(r_lds_f)(size_t)(tw_w + (unsigned)((((r & 3) + 8 * (r >> 2)) * EPW) * 4)) = dacc[a4][ct][r];
  };
  auto drain_S = [&](int j) {
    const int ct = j >> 2, a4 = j & 3;
    if (a4 == 0) { st1 = 0.f; st2 = 0.f; bias4 = bias_n; }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const r_f32x4 v4 = *(r_lds_f4)(size_t)(tw_r + (unsigned)(it * 8 * EPW * 4));
      float4 v = make_float4(v4[0], v4[1], v4[2], v4[3]);
      const float4 u = rv[j & 1][it];
      // the operand scales are powers of two: scaling after the transpose, fused with the bias / residual add, is exact (conv3x3p expression)
      v.x = fmaf(v.x, oscale, bias4.x + u.x); v.y = fmaf(v.y, oscale, bias4.y + u.y); v.z = fmaf(v.z, oscale, bias4.z + u.z); v.w = fmaf(v.w, oscale, bias4.w + u.w);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(r_u32x4, v), R_RS(P.y, d_live), (int)lane_y, (int)(d_rb4 + it * y_it + a4 * y_a4 + ct * 128), 0);
      st1 += (v.x + v.y) + (v.z + v.w);
      st2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, st2))));
    }
    if (a4 == 3) {      // (sum, sum of squares) of the band's 128 pixels per channel quad: the eight lanes holding a quad combine, lanes 0..7 write
      float s1 = st1, s2 = st2;
      s1 += __shfl_xor(s1, 8); s2 += __shfl_xor(s2, 8);
      s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      r_u32x2 sv = {__float_as_uint(s1), __float_as_uint(s2)};
      __builtin_amdgcn_raw_buffer_store_b64(sv, R_RS(stat_, d_live & stat_on), (int)lane_st, (int)(d_sb8 + ct * 64), 0);
    }
  };

  // ---- units.  A main chunk = 18 k-steps (tap, k-half) of 2 units; U = unit of the chunk, S = step, all compile-time after unrolling
  // FIRST: first chunk of a tile -- accumulators start from zero, the previous tile drains underneath
  // timing probes (tools/probe_build.py, WRONG RESULTS by design)
#ifdef PDAE_R_PROBE_NOA
#define PDAE_R_PA(X)
#else
#define PDAE_R_PA(X) X
#endif
#ifdef PDAE_R_PROBE_NOB
#define PDAE_R_PB(X)
#else
#define PDAE_R_PB(X) X
#endif
#ifdef PDAE_R_PROBE_NOGLOAD
#define PDAE_R_PG(X)
#else
#define PDAE_R_PG(X) X
#endif
#ifdef PDAE_R_PROBE_NOCONV
#define PDAE_R_PC(X)
#else
#define PDAE_R_PC(X) X
#endif
#ifdef PDAE_R_PROBE_NODRAIN
#define PDAE_R_PD(X)
#else
#define PDAE_R_PD(X) X
#endif
#define PDAE_R_UNIT_MAIN(U, FIRST)                                                                           \
    {                                                                                                       \
      const int s_ = (U) >> 1, u_ = (U) & 1, tap_ = s_ >> 1, kc_ = s_ & 1;                                  \
      PDAE_R_PA(if ((U) < 35) lda(fa[((U) + 1) & 1], ((U) + 1) >> 2, (((U) + 1) >> 1) & 1, ((U) + 1) & 1);  \
      else lda(fa[0], tap_, kc_, u_);)                                                                      \
      PDAE_R_PB(if (u_ == 0) {                                                                              \
        if (s_ + 2 < 18) PDAE_R_LDB_MAIN(qb[(s_ + 2) % 3], chunk, s_ + 2, c_nt0);                           \
        else PDAE_R_LDB_AFTER_MAIN(qb[(s_ + 2) % 3], chunk, s_ + 2 - 18)                                    \
      })                                                                                                    \
      PDAE_R_PG(if ((U) % 6 == 0 && (U) < 24) { _Pragma("unroll") for (int g_ = 0; g_ < 3; ++g_) gload_one(((U) / 6) * 3 + g_); }) \
      PDAE_R_PD(if ((FIRST) && (U) % 4 == 0 && (U) < 32) drain_L((U) / 4);                                  \
      if ((FIRST) && (U) % 4 == 1 && (U) >= 5 && (U) < 37) drain_W(((U) - 5) / 4);                          \
      if ((FIRST) && (U) % 4 == 2 && (U) >= 6 && (U) < 38) drain_S(((U) - 6) / 4);)                         \
      PDAE_R_PC(if (RCV_AT(U)) convert_store(RCV_SLOT(U));)                            \
      mma(fa[(U) & 1], qb[s_ % 3], u_, (FIRST) && s_ == 0);                                                 \
      PDAE_R_PATTERN(5)                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
  // skip chunk K (0..2) of a group of three: centre tap, 2 k-steps = 4 units; group step g = 2 K + kc, ring slot g % 3
#define PDAE_R_UNIT_SKIP(K, U4)                                                                              \
    {                                                                                                       \
      const int kc_ = (U4) >> 1, u_ = (U4) & 1, g_s = 2 * (K) + kc_;                                        \
      if ((U4) < 3) lda(fa[((U4) + 1) & 1], 4, ((U4) + 1) >> 1, ((U4) + 1) & 1);                            \
      else lda(fa[0], 4, kc_, u_);                                                                          \
      if (u_ == 0) {                                                                                        \
        if (g_s + 2 < 6) PDAE_R_LDB_SKIP(qb[(g_s + 2) % 3], r0 + ((g_s + 2) >> 1), (g_s + 2) & 1, c_nt0);   \
        else PDAE_R_LDB_AFTER_SKIP(qb[(g_s + 2) % 3], r0, g_s + 2)                                          \
      }                                                                                                     \
      if ((U4) == 0) { _Pragma("unroll") for (int l_ = 0; l_ < RLD; ++l_) gload_one(l_); }                  \
      if ((U4) == 3) { _Pragma("unroll") for (int l_ = 0; l_ < RLD; ++l_) convert_store(l_); }              \
      mma(fa[(U4) & 1], qb[g_s % 3], u_, false);                                                            \
      PDAE_R_PATTERN(5)                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
  // the 36 units of a main chunk written out: LLVM refuses to fully unroll a loop of this size even under "#pragma unroll" (pragma-unroll-threshold)
#ifdef PDAE_R_PROBE_24U      // timing probe (WRONG RESULTS): 24 of the 36 units of a main chunk = the matrix work of a Winograd F(2, 3)-along-x form in this structure
#define PDAE_R_36(M, F) M(0, F) M(1, F) M(2, F) M(3, F) M(4, F) M(5, F) M(6, F) M(7, F) M(8, F) M(9, F) M(10, F) M(11, F) M(12, F) M(13, F) M(14, F) M(15, F) M(16, F) M(17, F) \
    M(18, F) M(19, F) M(20, F) M(21, F) M(22, F) M(23, F)
#else
#define PDAE_R_36(M, F) M(0, F) M(1, F) M(2, F) M(3, F) M(4, F) M(5, F) M(6, F) M(7, F) M(8, F) M(9, F) M(10, F) M(11, F) M(12, F) M(13, F) M(14, F) M(15, F) M(16, F) M(17, F) \
    M(18, F) M(19, F) M(20, F) M(21, F) M(22, F) M(23, F) M(24, F) M(25, F) M(26, F) M(27, F) M(28, F) M(29, F) M(30, F) M(31, F) M(32, F) M(33, F) M(34, F) M(35, F)
#endif
  // position of the prefetch pipeline after (tile, slot)
  auto advance = [&](int tile, int slot, int& ntile, int& nslot) {
    const bool last = slot + 1 >= nmain + nxp;
    ntile = last ? tile + G : tile; nslot = last ? 0 : slot + 1;
  };

  // ---- prologue: first chunk of the first tile staged, weight fragments of its first two k-steps in flight
  int tile = blockIdx.x;
  int c_img, c_y0, c_x0, c_n0, x_img, x_y0, x_x0, x_n0;     // origin of the tile being multiplied | of the tile after it (itself when there is none)
  PDAE_R_DECODE(tile, c_img, c_y0, c_x0, c_n0);
  x_img = c_img; x_y0 = c_y0; x_x0 = c_x0; x_n0 = c_n0;
  c_nt0 = (c_n0 >> 5) + wn * 2;
  PDAE_R_CHUNK_SRC(tile, 0)
  cur = 1;                                                  // convert_store writes buffer cur ^ 1 = 0 ...
#pragma unroll
  for (int l = 0; l < RLD; ++l) gload_one(l);
  PDAE_R_LDB_MAIN(qb[0], 0, 0, c_nt0);
  PDAE_R_LDB_MAIN(qb[1], 0, 1, c_nt0);
#pragma unroll
  for (int l = 0; l < RLD; ++l) convert_store(l);
  cur = 0;                                                  // ... which the first chunk reads
  __syncthreads();

  for (; tile < ntiles; tile += G) {
    PDAE_R_DECODE(tile + G < ntiles ? tile + G : tile, x_img, x_y0, x_x0, x_n0);
    x_nt0 = (x_n0 >> 5) + wn * 2;
    // ---- main chunks.  The first one (zero start, drain underneath) is its own straight-line copy IN FRONT of the loop over the others: a join
    // of two 432-MFMA bodies inside one loop made the register allocator route the accumulators through VGPRs (seen with the one-tile prototype)
#define PDAE_R_CHUNK(FIRST)                                                                                  \
    {                                                                                                       \
      int pt, ps;                                                                                           \
      advance(tile, chunk, pt, ps);                                                                         \
      PDAE_R_CHUNK_SRC(pt, ps)                                                                              \
      abase = a_lane + cur * BUF_B;                                                                         \
      asm volatile("" : "+v"(abase));                                                                       \
      lda(fa[0], 0, 0, 0);                                                                                  \
      PDAE_R_36(PDAE_R_UNIT_MAIN, FIRST)                                                                    \
      __syncthreads();      /* the next patch is complete in the other buffer, everyone is done with this one */ \
      cur ^= 1u;                                                                                            \
    }
    {
      const int chunk = 0;
      PDAE_R_CHUNK(true)
    }
    for (int chunk = 1; chunk < nmain; ++chunk) PDAE_R_CHUNK(false)
#undef PDAE_R_CHUNK
    // ---- skip chunks, three at a time (padding chunks: zero patch, real weights)
    for (int r0 = 0; r0 < nxp; r0 += 3) {
#define PDAE_R_SKIP_CHUNK(K)                                                                                 \
      {                                                                                                     \
        int pt, ps;                                                                                         \
        advance(tile, nmain + r0 + (K), pt, ps);                                                            \
        PDAE_R_CHUNK_SRC(pt, ps)                                                                            \
        abase = a_lane + cur * BUF_B;                                                                       \
        asm volatile("" : "+v"(abase));                                                                     \
        lda(fa[0], 4, 0, 0);                                                                                \
        PDAE_R_UNIT_SKIP(K, 0) PDAE_R_UNIT_SKIP(K, 1) PDAE_R_UNIT_SKIP(K, 2) PDAE_R_UNIT_SKIP(K, 3)         \
        __syncthreads();                                                                                    \
        cur ^= 1u;                                                                                          \
      }
      PDAE_R_SKIP_CHUNK(0) PDAE_R_SKIP_CHUNK(1) PDAE_R_SKIP_CHUNK(2)
#undef PDAE_R_SKIP_CHUNK
    }
    // ---- hand the finished tile to the drain
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) dacc[a][ct] = acc[a][ct];
    {
      const int ya = c_y0 + wm * 8, col = c_n0 + wn * 64;
      d_rb4 = (unsigned)((((c_img * P.H + ya) * P.W + c_x0) * P.Nout + col) * 4);
      d_xb4 = (unsigned)((((c_img * (P.H >> rsh) + (ya >> rsh)) * (P.W >> rsh) + (c_x0 >> rsh)) * P.Nout + col) * 4);
      d_sb8 = (unsigned)(((c_img * P.stat_tpi + (ya >> 3) * P.tiles_x + (c_x0 >> 4)) * (P.Nout >> 2) + (col >> 2)) * 8);
      d_col4 = (unsigned)(col * 4);
    }
    d_live = RALL;
    c_nt0 = x_nt0;
    c_img = x_img; c_y0 = x_y0; c_x0 = x_x0; c_n0 = x_n0;
  }
#undef PDAE_R_UNIT_MAIN
#undef PDAE_R_UNIT_SKIP
#undef PDAE_R_36

  // ---- the last tile of this workgroup drains in the open
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    drain_L(j);
    drain_W(j);
    drain_S(j);
  }
  if constexpr (NS == 4) pdae_sat_report(P.sat, sat_hit);
}

template <int NS, bool GN> static int launch_r(const PatchParams& P, hipStream_t s) {
  const size_t smem = (size_t)2 * NPL(NS) * RPLANE_B + (size_t)4 * 32 * EPW * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3r_kernel<NS, GN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { pdae_set_error("conv3x3r: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  const long long ntiles = (long long)P.N * P.tiles_y * P.tiles_x * P.tiles_n;
  const int cus = 256;                                       // one persistent workgroup per CU (133 KB of LDS, 512 registers per lane)
  dim3 grid((unsigned)(ntiles < cus ? ntiles : cus));
  hipLaunchKernelGGL((conv3x3r_kernel<NS, GN>), grid, dim3(RTHREADS), smem, s, P);
  return pdae_launch_status("conv3x3r");
}

// PDAE_P3R = 0 routes everything to the other patch kernels (A-B aid), 2 ignores the fill heuristic (tests: small shapes)
static int r_mode() { return pdae_knob(KNOB_P3R); }

// eligibility of a launch conv3x3p_launch has planned WITHOUT split-K: at most two operand planes, 16 x 16 tiles, whole 128-channel output
// tiles, buffer-addressable tensors (< 4 GB each), and enough tiles that the last round of the 256 persistent workgroups wastes little
bool conv3x3r_ok(int math, int C, int H, int W, int N, int Nout, int Hs, int Ws, int C0, int Cs0, int Cs1) {
  if (r_mode() == 0) return false;
  if (!(math == 1 || math == 2 || math == 4)) return false;
  if ((H % RTH) || (W % PTW) || (Nout % RBN) || (C & 31) || (C0 & 31) || H >= 2048 || W >= 2048) return false;
  const unsigned long long lim = 0xFFFFFFF0ull;
  const int cmax = C0 > C - C0 ? C0 : C - C0;
  if ((unsigned long long)N * Hs * Ws * cmax * 4ull >= lim) return false;
  const int smax = Cs0 > Cs1 ? Cs0 : Cs1;
  if ((unsigned long long)N * H * W * (smax > Nout ? smax : Nout) * 4ull >= lim) return false;
  if (r_mode() == 2) return true;
  const long long tiles = (long long)N * (H / RTH) * (W / PTW) * (Nout / RBN);
  const long long rounds = (tiles + 255) / 256;
  const int min_tiles = pdae_knob(KNOB_P3R_MIN), min_eff = pdae_knob(KNOB_P3R_EFF);      // tuning aids
  return tiles >= min_tiles && tiles * 100 >= rounds * 256 * min_eff;   // at least two tiles per CU (something to overlap), last round >= 85 % full
}

int conv3x3r_launch(int math, const PatchParams& P0, hipStream_t s) {
  PatchParams P = P0;
  P.tiles_x = P.W / PTW; P.tiles_y = P.H / RTH; P.tiles_n = P.Nout / RBN; P.splits = 1; P.cps = (P.C >> 5) + P.nx;
#define PDAE_R3(NS_) (P.coef ? launch_r<NS_, true>(P, s) : launch_r<NS_, false>(P, s))
  if (math == 1) return PDAE_R3(1);
  if (math == 2) return PDAE_R3(2);
  return PDAE_R3(4);
#undef PDAE_R3
}
