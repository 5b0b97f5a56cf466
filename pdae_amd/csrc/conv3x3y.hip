// Winograd F(2, 3)-along-x 3x3 convolution (forward / data gradient) for gfx950 in the PERSISTENT one-wave-per-SIMD structure of conv3x3r.hip.
//
// Per output pixel pair (x, x + 1) and row tap ky the three column taps become four transform positions c:
//   input   s = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)            of the pair's window d0..d3 = pixels 2p - 1 .. 2p + 2   (here, while staging, in fp32)
//   weights u = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2)  (wprepx_slot in conv3x3p.h, once per optimizer step)
//   M_c += s_c u_c on the MFMAs;   Y0 = M0 + M1 + M2,  Y1 = M1 - M2 - M3 in the epilogue
// i.e. 4 instead of 6 products per pair, row tap and channel: two thirds of the matrix instructions of the direct form.  History of the form:
// conv3x3x.hip (two waves per SIMD, 256 registers: 0.75-0.9x of the direct kernels), the probe build of conv3x3r with 24 of its 36 units per
// chunk (0.598 vs 0.857 ms on 128x128 256->128, B = 32: the bound that motivated this file), winograd.hip (the 2-D form, gated out);
// profiles/r04_winograd_probe.txt has the measurements, DESIGN.md section 7 the probe ladder of this kernel.
//
//   * 4 waves, one per SIMD, 512 registers: wave w = the whole 16 x 16-pixel tile x 32 output channels (32 w ..), accumulators acc[c][a] = 16 tiles
//     of 32 x 32 in the 256 AGPRs (transform position c; m-tile a = (row half a >> 1, pair half a & 1)).  Each weight fragment is fetched by ONE
//     wave (the earlier (8 rows x 64 channels) split fetched every fragment twice per CU: 50 B/clk through the vector cache at full matrix rate,
//     against 25 now) and used for four products; the patch fragments -- LDS, 256 B/clk -- are read by all four waves.  No second accumulator
//     set: the epilogue is NOT deferred, it runs between tiles (output transform lane-local, conv3x3r's transposition + float4 stores).
//   * step = 16 input channels = 12 units (ky, c) of 12 MFMAs (4 m-tiles x 3 products, a dependent pair 4 issues apart).  Weight fragments: ring
//     of YRB units straight from L2 (buffer loads, YRB - 1 units ahead); patch fragments: ring of YRA units from LDS.
//   * LDS patch in the transform domain, double buffered: 18 rows x 36 positions (c * 8 + pair) x 48-byte rows (16 channels + pad: 36 * 48 = 192
//     (mod 256) keeps every ds_read_b128 fragment conflict-free) x planes = 62 KB per buffer.  The raw input of step s + 2 is loaded while step s
//     is multiplied (8 - 11 units ahead of its use) and transformed / split / stored into the other buffer while step s + 1 is: per thread two
//     items (patch row, pair, channel quad: own pixel pair + one segment-edge pixel, the neighbours' pixels through DPP row shifts) and one
//     quarter item of patch rows 16, 17 (one position of a pair: two loads); the conversions are spread over units 0 - 8.  ONE barrier per step,
//     placed so that the first fragments of the next step are fetched behind it.
//   * everything inside a unit is branch-free (absent operands = empty buffer resources / out-of-range offsets, as in conv3x3r).
// Prepared weights: the Winograd-along-x layout (wprepx_slot, 12 taps x 2 k halves per 32-channel chunk).  Fused 1x1 skip chunks are not built for
// this form: pdae_conv2d_fwd_skip_ok says no where it applies (conv3x3p_skip_ok).
#include <stdlib.h>

#include "common.h"
#include "igemm.h"
#include "conv3x3p.h"

#define YTHREADS 256
#define YPW 36
#define YROWB 48u
#define YNPOS_(RH_) ((8 * (RH_) + 2) * YPW)          // patch rows: the tile's 8 RH rows + halo
#define YPLANE_B_(RH_) ((unsigned)YNPOS_(RH_) * YROWB)      // 31104 (RH = 2), 17280 (RH = 1)
#define YOOB 0xFFFFFFF0u
#define YALL 0xFFFFFFEFu

typedef unsigned y_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned y_u32x2 __attribute__((ext_vector_type(2)));
typedef float y_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const y_u32x4* y_lds_u4;
typedef __attribute__((address_space(3))) y_u32x2* y_lds_u2;
typedef __attribute__((address_space(3))) float* y_lds_f;
typedef __attribute__((address_space(3))) const y_f32x4* y_lds_f4;

#define PDAE_Y_PATTERN(NV)                                                                                  \
  _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                                                      \
    if (RH == 2 || (i_ & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      /* 12 (RH = 2) or 6 MFMAs per unit */ \
    __builtin_amdgcn_sched_group_barrier(0x006, NV, 0);                                                    \
    if (i_ < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                     \
    if (i_ >= 4) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);                                        \
    __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);                                                     \
  }

// tile index -> (image, tile row, tile column, channel tile) without integer divisions: q = (n * m) >> 32 with m = floor(2^32 / d) + 1 is n / d for
// n * d < 2^32 (launch check); d == 1 has no 32-bit m
struct YDiv { unsigned mn, mx, my; };

// EX: the launch has a residual or accumulates (an epilogue operand to load); without it the 32 loads per tile and their 64 registers do not exist
// (against an empty resource they still cost the vector-memory path ~35 cycles each: tools/micro/unit_pipe.hip)
// ST: the launch leaves the GroupNorm partial statistics of its output (pdae_conv_stats_arm); data gradients and convolutions not followed by a
// GroupNorm do not, and then the 64 VALU instructions per epilogue block that sum them do not exist either
// GB: the launch is a data gradient that leaves the GroupNorm-backward sums of its output (PatchParams::gb_*): EX's operand slots carry the
// GroupNorm's raw input x instead of a residual (the operand is NOT added), the sums replace ST's
// RH: row halves of the tile.  2 = 16 x 16 pixels (everything above); 1 = 8 x 16 pixels (round 5): the same pipeline with ONE staging item per thread
// (patch rows 0 .. 7; the quarter item takes rows 8, 9), 8 accumulator tiles, 6 MFMAs per unit and 2 epilogue blocks -- for layers whose 16-row
// tiles leave most of the chip idle (16^2 at B = 32: 96 tiles of 16 x 16 x 128 against 192 of 8 x 16 x 128), which otherwise run the direct form
// split over K with slab reductions (conv3x3p.hip) at 3 instead of 2 MFMAs per product.
template <int NS, bool GN, bool EX, bool ST, bool GB = false, int RH = 2>
__global__ void __launch_bounds__(YTHREADS, 1) conv3x3y_kernel(const PatchParams P, const int stagger, const YDiv D) {
  static_assert(!GB || (EX && !ST && !GN), "GB: a data gradient with the operand slots, no forward statistics, no fused GroupNorm input");
  static_assert(RH == 1 || RH == 2, "tiles of 8 or 16 rows");
  constexpr int NP = NPL(NS);
  constexpr unsigned YPLANE_B = YPLANE_B_(RH);
  constexpr int NA = 2 * RH;                       // m-tiles of a wave: (row half a >> 1, pair half a & 1)
  constexpr unsigned BUF_B = NP * YPLANE_B;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, h = lane >> 5;
  const int C = P.C, C1 = C - P.C0, nk = C >> 4;
  const int ntiles = P.N * P.tiles_y * P.tiles_x * P.tiles_n, G = gridDim.x;
  const unsigned lds0 = (unsigned)(size_t)smem;
  const float ascale = NS == 4 ? (P.amax ? p_pow2_scale(*P.amax) : PASCALE) : 1.0f;
  float sat_hit = 0.f;                             // fp16-window guard: running per-lane max |raw operand| (UNSCALED: two v_max3 per float4; the scale is applied once, at the end)
  auto sat_track = [&](const float4& v) {
    sat_hit = __builtin_fmaxf(__builtin_fmaxf(sat_hit, __builtin_fabsf(v.x)), __builtin_fabsf(v.y));
    sat_hit = __builtin_fmaxf(__builtin_fmaxf(sat_hit, __builtin_fabsf(v.z)), __builtin_fabsf(v.w));
  };
  const int up_sh = P.up ? 1 : 0;
  const float* const x1_ = P.x1 ? P.x1 : P.x;

#define Y_UDIV(N_, D_, M_) ((D_) == 1 ? (unsigned)(N_) : __umulhi((unsigned)(N_), (M_)))
#define Y_DECODE(TILE, IMG, Y0, X0, N0)                                                                       \
  { const unsigned tl_ = (unsigned)(TILE), q1_ = Y_UDIV(tl_, P.tiles_n, D.mn), q2_ = Y_UDIV(q1_, P.tiles_x, D.mx), q3_ = Y_UDIV(q2_, P.tiles_y, D.my);   \
    IMG = (int)q3_; Y0 = (int)(q2_ - q3_ * (unsigned)P.tiles_y) * (8 * RH); X0 = (int)(q1_ - q2_ * (unsigned)P.tiles_x) * 16; N0 = (int)(tl_ - q1_ * (unsigned)P.tiles_n) * 128; }
#define Y_DECODE_N0(TILE, N0) { const unsigned tl_ = (unsigned)(TILE), q1_ = Y_UDIV(tl_, P.tiles_n, D.mn); N0 = (int)(tl_ - q1_ * (unsigned)P.tiles_n) * 128; }

  // ---- staging roles.  Items l = 0, 1: (patch row (t >> 5) + 8 l, pair (t >> 2) & 7, channel quad t & 3): own pixels b = 1, 2 and, for the first / last
  // pair of the row, the edge pixel.  Quarter item (patch rows 16, 17): (row 16 + (t >> 7), pair (t >> 4) & 7, quad (t >> 2) & 3, position t & 3): the
  // two pixels that position needs (c = 0: b = 0, 2; c = 1, 2: b = 1, 2; c = 3: b = 1, 3).
  const int qd = t & 3, wt = (t >> 2) & 7, r8 = t >> 5;
  const int q_c = t & 3, q_qd = (t >> 2) & 3, q_wt = (t >> 4) & 7, q_row = 8 * RH + (t >> 7);
  const int q_ba = q_c == 0 ? 0 : 1, q_bb = q_c == 3 ? 3 : 2;
  // per-tile pixel bookkeeping of the tile whose raw data is being LOADED (ld_*): source pixel index of b = 1 per item + validity bits
  int ld_pb[RH], ld_qa = 0, ld_qb = 0;
  unsigned ld_vm = 0u;                           // bits 0, 1: items' rows valid; 2, 3: items' edge pixels valid; 4, 5: quarter item's two pixels valid
  // the neighbours' pixels come across lanes with DPP row shifts (16-lane rows = 4 pairs x 4 quads): the first / last pair of each 4-pair segment
  // loads its outer pixel itself (the tile's edge pixel for pairs 0 and 7, a pixel of the neighbouring segment for pairs 3 and 4)
  const bool seg_lo = (wt & 3) == 0, seg_hi = (wt & 3) == 3;
  const int eb = seg_lo ? 0 : 3;
  const int ob2 = P.up ? 0 : 1, obe = seg_lo ? -1 : (P.up ? 1 : 2);
#define Y_LD_TILE(IMG, Y0, X0, LIVE)                                                                          \
  {                                                                                                           \
    /* the lane's roles are re-derived from the thread index here: kept live across the step they were spilled in the fused-GroupNorm instantiation */ \
    int t_ = t;                                                                                               \
    asm volatile("" : "+v"(t_));                                                                              \
    const int wt_ = (t_ >> 2) & 7, r8_ = t_ >> 5, qc_ = t_ & 3, qwt_ = (t_ >> 4) & 7, qrow_ = 8 * RH + (t_ >> 7); \
    const int qba_ = qc_ == 0 ? 0 : 1, qbb_ = qc_ == 3 ? 3 : 2;                                               \
    const bool sl_ = (wt_ & 3) == 0, sh_ = (wt_ & 3) == 3;                                                    \
    ld_vm = 0u;                                                                                               \
    _Pragma("unroll") for (int l = 0; l < RH; ++l) {                                                          \
      const int ly = (Y0) - 1 + r8_ + 8 * l, lx1 = (X0) + 2 * wt_;                                            \
      const bool rok = (LIVE) && (unsigned)ly < (unsigned)P.H;                                                \
      ld_pb[l] = ((IMG) * P.Hs + (ly >> up_sh)) * P.Ws + (lx1 >> up_sh);                                      \
      ld_vm |= (rok ? 1u : 0u) << l;                                                                          \
      ld_vm |= ((rok && (sl_ || sh_) && (unsigned)(lx1 - 1 + (sl_ ? 0 : 3)) < (unsigned)P.W) ? 4u : 0u) << l; \
    }                                                                                                         \
    {                                                                                                         \
      const int ly = (Y0) - 1 + qrow_, lxa = (X0) - 1 + 2 * qwt_ + qba_, lxb = (X0) - 1 + 2 * qwt_ + qbb_;    \
      const bool rok = (LIVE) && (unsigned)ly < (unsigned)P.H;                                                \
      ld_qa = ((IMG) * P.Hs + (ly >> up_sh)) * P.Ws + (lxa >> up_sh);                                         \
      ld_qb = ((IMG) * P.Hs + (ly >> up_sh)) * P.Ws + (lxb >> up_sh);                                         \
      ld_vm |= (rok && (unsigned)lxa < (unsigned)P.W) ? 16u : 0u;                                             \
      ld_vm |= (rok && (unsigned)lxb < (unsigned)P.W) ? 32u : 0u;                                             \
    }                                                                                                         \
  }
  // validity of the data being CONVERTED (one step behind the loads): a copy taken when the load position moves on
  unsigned cv_vm = 0u;
  // source of the chunk being loaded: pointer, bytes per pixel, byte offset of the chunk's channels
  const float* ld_ptr = P.x; unsigned ld_ldb = 0, ld_cb = 0;
#define Y_LD_SRC(K)                                                                                           \
  { const int c_ = (K) << 4; const bool first_ = c_ < P.C0;                                                   \
    ld_ptr = first_ ? P.x : x1_; ld_ldb = (unsigned)(first_ ? P.C0 : C1) * 4u; ld_cb = (unsigned)(first_ ? c_ : c_ - P.C0) * 4u; }
#define Y_RS(PTR) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>((const void*)(PTR)), 0, (int)YALL, 0x00020000)
  float4 apre[RH][3], qpre[2];
  auto gload_item = [&](int l) {
    const unsigned v1 = (unsigned)ld_pb[l] * ld_ldb + (unsigned)(qd * 16);
    const bool rok = (ld_vm >> l) & 1u, eok = (ld_vm >> (2 + l)) & 1u;
    apre[l][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(Y_RS(ld_ptr), (int)(rok ? v1 : YOOB), (int)ld_cb, 0));
    apre[l][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(Y_RS(ld_ptr), (int)(rok ? v1 + (unsigned)ob2 * ld_ldb : YOOB), (int)ld_cb, 0));
    apre[l][2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(Y_RS(ld_ptr), (int)(eok ? v1 + (unsigned)obe * ld_ldb : YOOB), (int)ld_cb, 0));
  };
  auto gload_quarter = [&]() {
    const unsigned va = (unsigned)ld_qa * ld_ldb + (unsigned)(q_qd * 16), vb = (unsigned)ld_qb * ld_ldb + (unsigned)(q_qd * 16);
    qpre[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(Y_RS(ld_ptr), (int)((ld_vm & 16u) ? va : YOOB), (int)ld_cb, 0));
    qpre[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(Y_RS(ld_ptr), (int)((ld_vm & 32u) ? vb : YOOB), (int)ld_cb, 0));
  };
  // GroupNorm coefficients of the chunk being converted (this thread's quad for the items, and for the quarter item)
  float4 gmu, gsc, gsh, hmu, hsc, hsh;
  // GroupNorm coefficients of a step = 12 float4 ([mu | scale | shift] x 4 channel quads): ONE buffer load per wave fetches them (lane i < 12 takes
  // float4 i), they are parked in a private LDS slot and read back per quad.  The direct form (six global loads per thread and step) was 6 of the
  // 38 vector-memory instructions of a fused-GroupNorm step, each ~35 cycles of the CU's vector-memory path (tools/micro/unit_pipe.hip).
  float4 cf_raw = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned cf_lds = lds0 + 2u * BUF_B + 4u * 2u * 32u * EPW * 4u + (unsigned)wv * 256u;
  const unsigned cf_voff = lane < 12 ? (unsigned)(((lane >> 2) * P.N * C + (lane & 3) * 4) * 4) : YOOB;
  const unsigned cf_slot = cf_lds + (unsigned)(lane < 12 ? lane : 12) * 16u;
  auto coef_fetch = [&](int img, int k) {
    if constexpr (GN)
      cf_raw = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(Y_RS(P.coef), (int)cf_voff, (int)((img * C + (k << 4)) * 4), 0));
  };
  auto coef_stash = [&]() {
    if constexpr (GN) *(__attribute__((address_space(3))) y_f32x4*)(size_t)cf_slot = y_f32x4{cf_raw.x, cf_raw.y, cf_raw.z, cf_raw.w};
  };
  auto cf_read = [&](int kind, int quad) {
    const y_f32x4 v = *(y_lds_f4)(size_t)(cf_lds + (unsigned)((kind * 4 + quad) * 16));
    return make_float4(v[0], v[1], v[2], v[3]);
  };
  auto coef_load = [&]() {                         // the items' quad, for the conversions of the next step
    if constexpr (GN) { gmu = cf_read(0, qd); gsc = cf_read(1, qd); gsh = cf_read(2, qd); }
  };
  auto coef_load_q = [&]() {                       // the quarter item's quad: two units before its conversion (twelve registers less across the step)
    if constexpr (GN) { hmu = cf_read(0, q_qd); hsc = cf_read(1, q_qd); hsh = cf_read(2, q_qd); }
  };
  // silu(scale * (x - mu) + shift) on a channel quad, written on float pairs so that the subtract / fma / the three multiply-adds of SiLU are packed
  // instructions (v_pk_add_f32, v_pk_fma_f32, v_pk_mul_f32: half the VALU issue slots; the fused-GroupNorm instantiation spends 3.9 VALU instructions
  // per MFMA on this map); padding pixels are multiplied by 0 AFTER the map (their raw value is the 0 of an out-of-range load)
  typedef float y_f32x2 __attribute__((ext_vector_type(2)));
  auto gn_map = [&](float4 v, bool on, const float4& mu, const float4& sc_, const float4& sh_) {
    const float onf = on ? 1.0f : 0.0f;
    y_f32x2 r[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const y_f32x2 x2 = hh ? y_f32x2{v.z, v.w} : y_f32x2{v.x, v.y};
      const y_f32x2 mu2 = hh ? y_f32x2{mu.z, mu.w} : y_f32x2{mu.x, mu.y};
      const y_f32x2 sc2 = hh ? y_f32x2{sc_.z, sc_.w} : y_f32x2{sc_.x, sc_.y};
      const y_f32x2 sh2 = hh ? y_f32x2{sh_.z, sh_.w} : y_f32x2{sh_.x, sh_.y};
      const y_f32x2 m = __builtin_elementwise_fma(sc2, x2 - mu2, sh2);
      const y_f32x2 a = m * -1.4426950408889634f;
      const y_f32x2 d = y_f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + 1.0f;
      r[hh] = m * y_f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])} * onf;      // act == 1 (launch check)
    }
    return make_float4(r[0][0], r[0][1], r[1][0], r[1][1]);
  };
  unsigned cur = 0;
  // LDS store bases (into buffer cur ^ 1): items: row r8 + 8 l, position c * 8 + wt, quad qd; quarter item: row q_row, position q_c * 8 + q_wt
  const unsigned w_item = lds0 + (unsigned)((r8 * YPW + wt) * YROWB + qd * 8);
  const unsigned w_quar = lds0 + (unsigned)((q_row * YPW + q_c * 8 + q_wt) * YROWB + q_qd * 8);
  // conversion of an item in three parts (one per unit): A = map + window tracking of the own pixels, B = edge pixel, neighbours, transform, C = split + store
  float4 sv[4];
  auto conv_A = [&](int l) {
    if constexpr (GN) {
      const bool rok = (cv_vm >> l) & 1u;
      apre[l][0] = gn_map(apre[l][0], rok, gmu, gsc, gsh); apre[l][1] = gn_map(apre[l][1], rok, gmu, gsc, gsh);
    }
    if constexpr (NS == 4) { sat_track(apre[l][0]); sat_track(apre[l][1]); }
  };
  auto conv_B = [&](int l) {
    float4 de = apre[l][2];
    if constexpr (GN) de = gn_map(de, (cv_vm >> (2 + l)) & 1u, gmu, gsc, gsh);
    if constexpr (NS == 4) sat_track(de);
    const float4 d1 = apre[l][0], d2 = apre[l][1];
#define Y_SHR4(V) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), 0x114, 0xf, 0xf, true))      /* row_shr:4: lane - 4 */
#define Y_SHL4(V) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), 0x104, 0xf, 0xf, true))      /* row_shl:4: lane + 4 */
    float4 d0 = make_float4(Y_SHR4(d2.x), Y_SHR4(d2.y), Y_SHR4(d2.z), Y_SHR4(d2.w));
    float4 d3 = make_float4(Y_SHL4(d1.x), Y_SHL4(d1.y), Y_SHL4(d1.z), Y_SHL4(d1.w));
#undef Y_SHR4
#undef Y_SHL4
    if (seg_lo) d0 = de;
    if (seg_hi) d3 = de;
#define Y_F4(OP, A_, B_) make_float4(A_.x OP B_.x, A_.y OP B_.y, A_.z OP B_.z, A_.w OP B_.w)
    sv[0] = Y_F4(-, d0, d2); sv[1] = Y_F4(+, d1, d2); sv[2] = Y_F4(-, d2, d1); sv[3] = Y_F4(-, d1, d3);
  };
  auto conv_C = [&](int l, int c0) {               // positions c0, c0 + 1
    const unsigned dst0 = w_item + (cur ^ 1u) * BUF_B + (unsigned)(l * 8 * YPW) * YROWB;
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c) {
      unsigned a[NP], b2[NP];
      p_split2<NS>(sv[c].x, sv[c].y, a, ascale);
      p_split2<NS>(sv[c].z, sv[c].w, b2, ascale);
#pragma unroll
      for (int p = 0; p < NP; ++p) { const y_u32x2 w2 = {a[p], b2[p]}; *(y_lds_u2)(size_t)(dst0 + (unsigned)(c * 8) * YROWB + (unsigned)p * YPLANE_B) = w2; }
    }
  };
  auto conv_quarter = [&]() {
    float4 da = qpre[0], db = qpre[1];
    if constexpr (GN) { da = gn_map(da, (cv_vm >> 4) & 1u, hmu, hsc, hsh); db = gn_map(db, (cv_vm >> 5) & 1u, hmu, hsc, hsh); }
    if constexpr (NS == 4) { sat_track(da); sat_track(db); }
    // c = 0: d0 - d2; c = 1: d1 + d2; c = 2: d2 - d1; c = 3: d1 - d3   (da = first, db = second pixel of the position)
    const float sg = q_c == 1 ? 1.0f : -1.0f;
    float4 s = q_c == 2 ? Y_F4(-, db, da) : make_float4(fmaf(sg, db.x, da.x), fmaf(sg, db.y, da.y), fmaf(sg, db.z, da.z), fmaf(sg, db.w, da.w));
#undef Y_F4
    unsigned a[NP], b2[NP];
    p_split2<NS>(s.x, s.y, a, ascale);
    p_split2<NS>(s.z, s.w, b2, ascale);
    const unsigned dst = w_quar + (cur ^ 1u) * BUF_B;
#pragma unroll
    for (int p = 0; p < NP; ++p) { const y_u32x2 w2 = {a[p], b2[p]}; *(y_lds_u2)(size_t)(dst + (unsigned)p * YPLANE_B) = w2; }
  };

  // ---- fragments
  const unsigned a_lane = lds0 + (unsigned)(((li >> 2) * YPW + (li & 3)) * YROWB + h * 16);
  unsigned abase = a_lane;
  // patch fragments: ring of YRA units (requested YRA - 1 units ahead); the fused-GroupNorm instantiation has no registers for a third slot
  constexpr int YRA = GN ? 2 : 3;
  uint4 fa[YRA][NA][NP];                           // [unit mod YRA][m-tile a][plane]
  unsigned abase_n = a_lane;                       // the other buffer: unit 0 of the NEXT step is fetched during unit 11, behind the step's barrier
  auto lda = [&](uint4 (&af)[NA][NP], unsigned base, int u) {      // unit u: ky = u >> 2, c = u & 3
    const unsigned off = (unsigned)(((u >> 2) * YPW + (u & 3) * 8) * YROWB);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int p = 0; p < NP; ++p)
        af[a][p] = __builtin_bit_cast(uint4, *(y_lds_u4)(size_t)(base + off + (unsigned)(((a >> 1) * 8 * YPW + (a & 1) * 4) * YROWB) + (unsigned)p * YPLANE_B));
  };
  // U fragments of unit u of 16-channel chunk k: prepared layout [p][k >> 1][tp = u][kc = k & 1][nt][lane][8]
  const size_t plane_main = (size_t)(C >> 5) * 24 * P.NT * 512;
  const unsigned ps2 = (unsigned)(plane_main * 2);
  const int lane16 = lane * 16;
  const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.wp), 0, 0x7fffffff, 0x00020000);
  // weight fragments: ring of YRB units, requested YRB - 1 units (1920 cycles of MFMAs) ahead -- the vector-memory path returns in order, and with
  // a distance of three a fragment queued behind the raw-data loads of the same unit (HBM latency) arrived late three times per step
  constexpr int YRB = GN ? 6 : 4;
  uint4 qb[YRB][NP];                               // [unit mod YRB][plane]
  // byte offset of unit 0 of chunk k, channel tile nt0 (one scalar chain per STEP); a unit adds u * b_ustep -- the per-load form
  // ((((k >> 1) * 12 + u) << 1) + (k & 1)) * NT + nt0) * 1024 was five scalar instructions per load, 60 per step, in a wave whose issue port is full
  const unsigned b_ustep = (unsigned)(2 * P.NT * 1024);
  auto b_base = [&](int k, int nt0) { return (unsigned)((((k >> 1) * 24 + (k & 1)) * P.NT + nt0) * 1024); };
  auto ldb = [&](uint4 (&bq)[NP], unsigned base, int u) {
    const unsigned soff = base + (unsigned)u * b_ustep;
#pragma unroll
    for (int p = 0; p < NP; ++p) bq[p] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srd_w, lane16, (int)(soff + p * ps2), 0));
  };
  f32x16 acc[4][NA];                               // [c][a]
  auto mma = [&](const uint4 (&af)[NA][NP], const uint4 (&bq)[NP], int c, bool zc) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define PDAE_YA(P_) __builtin_bit_cast(bf16x8, af[a][P_])
#define PDAE_YB(P_) __builtin_bit_cast(bf16x8, bq[P_])
#define PDAE_YAH(P_) __builtin_bit_cast(f16x8, af[a][P_])
#define PDAE_YBH(P_) __builtin_bit_cast(f16x8, bq[P_])
#define PDAE_Y_EACH(STMT) _Pragma("unroll") for (int a = 0; a < NA; ++a) { STMT; }
    if constexpr (NS == 4) {
      PDAE_Y_EACH(acc[c][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_YAH(0), PDAE_YBH(1), zc ? zero : acc[c][a], 0, 0, 0))
      PDAE_Y_EACH(acc[c][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_YAH(1), PDAE_YBH(0), acc[c][a], 0, 0, 0))
      PDAE_Y_EACH(acc[c][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_YAH(0), PDAE_YBH(0), acc[c][a], 0, 0, 0))
    } else if constexpr (NS == 2) {
      PDAE_Y_EACH(acc[c][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_YA(0), PDAE_YB(1), zc ? zero : acc[c][a], 0, 0, 0))
      PDAE_Y_EACH(acc[c][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_YA(1), PDAE_YB(0), acc[c][a], 0, 0, 0))
      PDAE_Y_EACH(acc[c][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_YA(0), PDAE_YB(0), acc[c][a], 0, 0, 0))
    } else {
      PDAE_Y_EACH(acc[c][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_YA(0), PDAE_YB(0), zc ? zero : acc[c][a], 0, 0, 0))
    }
#undef PDAE_Y_EACH
#undef PDAE_YA
#undef PDAE_YB
#undef PDAE_YAH
#undef PDAE_YBH
  };

  // ---- the pipeline.  Steps of this workgroup in order: step = (tile, k), k = 16-channel chunk.  At the top of an iteration the LDS buffer `cur`
  // holds the CONVERTED step m (being multiplied), the registers hold the RAW data of step m + 1 (validity bits cv_vm, GroupNorm coefficients
  // loaded), qb[0 .. YRB - 2] hold the U fragments of the first YRB - 1 units of step m.  During the iteration: step m + 1 is converted into the other buffer, step
  // m + 2 (the load position l_*) is loaded into the freed registers, its coefficients at unit 11.
  // XCD-aware order (round 6): block b runs on XCD b % 8 and each XCD has its own L2, so with tile = block index the two (or more) 128-channel
  // output tiles of ONE pixel tile -- consecutive tile indices, which read the same input patch -- sat on different XCDs and every input byte was
  // fetched from HBM once per output tile (profiles/r05_pmc_traffic: 1043 MB per 128 -> 256 data gradient against ~540 MB once).  The bijection
  // below (conv3x3w's) gives the blocks of an XCD CONSECUTIVE tile indices: output tiles of a pixel tile, and pixel tiles that share halo
  // columns, meet in one L2.  Bit 16 of `stagger` switches it (knob PDAE_Y_XCD, default on).
  int m_tile = blockIdx.x;
  if (stagger & 0x10000) {
    const int q8 = G >> 3, r8 = G & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    m_tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  }
  if (m_tile >= ntiles) return;
  // Phase stagger: every workgroup has the same work, so without it all 256 reach their (exposed) epilogues together and the output of a whole
  // round -- 16 x 16 x 128 floats per CU, 33 MB -- queues on the HBM write path while every matrix pipe idles.  Four phase groups per XCD, each
  // `stagger` x 1024 cycles behind the previous one: a group drains its quarter of the round while the other three multiply.
  if ((stagger & 0xffff) > 0) {
    const int g = (blockIdx.x >> 3) & 3;
    for (int i = 0; i < g * (stagger & 0xffff); ++i) __builtin_amdgcn_s_sleep(16);
  }
  int m_img, m_y0, m_x0, m_n0, m_k = 0;
  Y_DECODE(m_tile, m_img, m_y0, m_x0, m_n0)
  auto advance = [&](int tile, int k, int& tile2, int& k2) { const bool last = k + 1 >= nk; tile2 = last ? tile + G : tile; k2 = last ? 0 : k + 1; };
  int l_tile = m_tile, l_k = 0, l_img = m_img, l_y0 = m_y0, l_x0 = m_x0, l_n0 = m_n0;
  // the load position moves one step ahead: new source, and at a tile boundary the pixel bookkeeping of the new tile (beyond the last step: nothing
  // valid).  (Branch-free under the MFMAs of unit 11 instead of at the top of the step it measured 3 % SLOWER: the unit's issue slots overflow.)
#define PDAE_Y_LD_NEXT()                                                                                      \
  {                                                                                                           \
    int t2_, k2_;                                                                                             \
    advance(l_tile, l_k, t2_, k2_);                                                                           \
    if (t2_ != l_tile) {                                                                                      \
      const bool live_ = t2_ < ntiles;                                                                        \
      Y_DECODE(live_ ? t2_ : 0, l_img, l_y0, l_x0, l_n0)                                                      \
      Y_LD_TILE(l_img, l_y0, l_x0, live_)                                                                     \
    }                                                                                                         \
    l_tile = t2_; l_k = k2_;                                                                                  \
    Y_LD_SRC(l_k)                                                                                             \
  }
  // prologue: step 0 converted into buffer 0, step 1 raw in the registers with its coefficients
  Y_LD_TILE(l_img, l_y0, l_x0, true)
  Y_LD_SRC(0)
#define Y_IF2(...) if constexpr (RH == 2) { __VA_ARGS__ }      /* the second staging item exists for 16-row tiles only */
  gload_item(0); Y_IF2(gload_item(1);) gload_quarter();
  coef_fetch(l_img, 0); coef_stash(); coef_load(); coef_load_q();
  cv_vm = ld_vm;
  cur = 1;                                         // the conversions write buffer cur ^ 1 = 0
  conv_A(0); conv_B(0); conv_C(0, 0); conv_C(0, 2);
  Y_IF2(conv_A(1); conv_B(1); conv_C(1, 0); conv_C(1, 2);)
  conv_quarter();
  cur = 0;
  PDAE_Y_LD_NEXT()
  gload_item(0); Y_IF2(gload_item(1);) gload_quarter();
  coef_fetch(l_img, l_k); coef_stash(); coef_load();      // (the quarter item's quad is read at unit 6 of the first step)
  {
    const int nt0 = (m_n0 >> 5) + wv;
#pragma unroll
    for (int u = 0; u < YRB - 1; ++u) ldb(qb[u], b_base(0, nt0), u);
  }
  int n_k, n_n0;                                   // chunk / first output channel of step m + 1 (its U fragments are requested YRB - 1 units ahead, from this step)
  {
    int t2;
    advance(m_tile, 0, t2, n_k);
    Y_DECODE_N0(t2 < ntiles ? t2 : m_tile, n_n0)
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < YRA - 1; ++u) lda(fa[u], a_lane, u);

  // timing probes (tools/probe_build.py; WRONG RESULTS by design)
#ifdef PDAE_Y_PROBE_NOA
#define PDAE_Y_DO_A(...)
#else
#define PDAE_Y_DO_A(...) __VA_ARGS__
#endif
#ifdef PDAE_Y_PROBE_NOB
#define PDAE_Y_DO_B(...)
#else
#define PDAE_Y_DO_B(...) __VA_ARGS__
#endif
#ifdef PDAE_Y_PROBE_NOCONV
#define PDAE_Y_CV(...) asm volatile("" :: "v"(apre[0][0].x), "v"(apre[RH - 1][0].x), "v"(qpre[0].x), "v"(apre[0][2].w), "v"(apre[RH - 1][2].w), "v"(qpre[1].w), "v"(apre[0][1].y), "v"(apre[RH - 1][1].y));
#else
#define PDAE_Y_CV(...) __VA_ARGS__
#endif
#ifdef PDAE_Y_PROBE_NOGLOAD
#define PDAE_Y_GL(...)
#else
#define PDAE_Y_GL(...) __VA_ARGS__
#endif
  // One unit: patch fragments of the next unit, U fragments of the unit three ahead (into the ring slot the previous unit has freed), one piece
  // of staging work, the unit's 12 MFMAs.
#define PDAE_Y_UNIT(U, FIRST, WORK)                                                                           \
  {                                                                                                           \
    PDAE_Y_DO_A(if ((U) + YRA - 1 < 12) lda(fa[((U) + YRA - 1) % YRA], abase, (U) + YRA - 1);                 \
                else lda(fa[((U) + YRA - 1) % YRA], abase_n, (U) + YRA - 1 - 12);)                            \
    PDAE_Y_DO_B(if ((U) + YRB - 1 < 12) ldb(qb[((U) + YRB - 1) % YRB], m_bb, (U) + YRB - 1);                  \
                else ldb(qb[((U) + YRB - 1) % YRB], n_bb, (U) + YRB - 1 - 12);)                               \
    WORK                                                                                                      \
    mma(fa[(U) % YRA], qb[(U) % YRB], (U) & 3, (FIRST) && (U) < 4);                                           \
    PDAE_Y_PATTERN(5)                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  }
#define PDAE_Y_STEP(FIRST)                                                                                    \
  {                                                                                                           \
    const unsigned m_bb = b_base(m_k, (m_n0 >> 5) + wv), n_bb = b_base(n_k, (n_n0 >> 5) + wv);                \
    cv_vm = ld_vm;                                  /* validity of the raw data in the registers (step m + 1) */ \
    PDAE_Y_LD_NEXT()                                /* the loads of this iteration: step m + 2 */              \
    abase = a_lane + cur * BUF_B; abase_n = a_lane + (cur ^ 1u) * BUF_B;                                      \
    asm volatile("" : "+v"(abase), "+v"(abase_n));                                                            \
    PDAE_Y_UNIT(0, FIRST, PDAE_Y_CV(conv_A(0);))                                                              \
    PDAE_Y_UNIT(1, FIRST, PDAE_Y_CV(conv_B(0);))                                                              \
    PDAE_Y_UNIT(2, FIRST, PDAE_Y_CV(conv_C(0, 0);))                                                           \
    PDAE_Y_UNIT(3, FIRST, PDAE_Y_CV(conv_C(0, 2);))                                                           \
    PDAE_Y_UNIT(4, FIRST, PDAE_Y_CV(Y_IF2(conv_A(1);)) PDAE_Y_GL(gload_item(0);) coef_fetch(l_img, l_k);)      /* eight units ahead of its conversion */ \
    PDAE_Y_UNIT(5, FIRST, PDAE_Y_CV(Y_IF2(conv_B(1);)))                                                       \
    PDAE_Y_UNIT(6, FIRST, PDAE_Y_CV(Y_IF2(conv_C(1, 0);)) coef_load_q();)                                     \
    PDAE_Y_UNIT(7, FIRST, PDAE_Y_CV(Y_IF2(conv_C(1, 2);)))                                                    \
    PDAE_Y_UNIT(8, FIRST, PDAE_Y_CV(conv_quarter();) PDAE_Y_GL(Y_IF2(gload_item(1);)))                        \
    PDAE_Y_UNIT(9, FIRST, PDAE_Y_GL(gload_quarter();))                                                        \
    /* the step's barrier: every conversion into the other buffer is done (unit 8), every read of this one is issued (unit 12 - YRA); */ \
    /* behind it the first fragments of the NEXT step are fetched from the other buffer */                     \
    if (YRA == 3) __syncthreads();                                                                            \
    PDAE_Y_UNIT(10, FIRST, coef_stash();)                                                                     \
    if (YRA == 2) __syncthreads();                                                                            \
    PDAE_Y_UNIT(11, FIRST, coef_load();)                                                                      \
    cur ^= 1u;                                                                                                \
  }

  // after a step: the next step of this tile becomes the one being multiplied (the tile change is handled behind the epilogue)
#define PDAE_Y_NEXT_K()                                                                                       \
  {                                                                                                           \
    m_k += 1;                                                                                                 \
    int t3_;                                                                                                  \
    advance(m_tile, m_k, t3_, n_k);                                                                           \
    Y_DECODE_N0(t3_ < ntiles ? t3_ : m_tile, n_n0)                                                            \
  }
  // The first chunk of a tile (accumulators start from zero) is its own straight-line copy IN FRONT of the loop over the others: a join of two
  // 144-MFMA bodies inside one loop makes the register allocator route the accumulators through VGPRs (conv3x3r.hip)
  for (;;) {
    PDAE_Y_STEP(true)
    for (int kk = 1; kk < nk; ++kk) {
      PDAE_Y_NEXT_K()
      PDAE_Y_STEP(false)
    }
#ifdef PDAE_Y_PROBE_NOEPI
    {
      f32x16 sm = acc[0][0];
#pragma unroll
      for (int i = 1; i < 4 * NA; ++i) sm += acc[i / NA][i % NA];
      float z = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) z += sm[r];
      if (z == 123.456f) P.y[t] = z;
    }
#else
    {
      // ---- epilogue of the finished tile, straight-line (conv3x3r's drain, not deferred): block b = m-tile a (row half b >> 1, pair half b & 1) in three stages
      //   L(b): residual / previous contents of both output columns (8 float4; a possibly EMPTY resource: zeros), two blocks ahead, because the
      //         vector-memory path returns in order and a load queued behind the previous block's stores waits for their acknowledgements;
      //   W(b): the four transform-domain accumulators are read once, both output columns (j = 0: M0 + M1 + M2, j = 1: M1 - M2 - M3) go into the
      //         wave's two private transposition tiles (a dedicated LDS region behind the patch buffers: no barrier on either side of the epilogue);
      //   S(b): float4 rows back, * scale + bias + residual, stored, statistics summed.
      const float oscale = NS == 4 ? P.woscale / ascale : 1.0f;
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));               // nothing of the epilogue's per-lane addressing may be hoisted in front of the chunk loop (it was: 34 spilled registers per tile)
      constexpr unsigned TWB = 32u * EPW * 4u;
      const unsigned tw = lds0 + 2u * BUF_B + (unsigned)wv * 2u * TWB;
      const int er = lane_e >> 3, ec = (lane_e & 7) * 4;
      const unsigned tw_w = tw + (unsigned)((4 * (lane_e >> 5) * EPW + (lane_e & 31)) * 4), tw_r = tw + (unsigned)((er * EPW + ec) * 4);
      const int n0w = m_n0 + wv * 32;
      const int rsh = (!GB && P.res_mode == 2) ? 1 : 0;    // half-resolution residual (nearest upsample): rows / columns >> 1, both columns of a pair read one pixel
      // GB: the operand is the GroupNorm input [x0 | x1]: this wave's 32 channels lie in one of the two tensors (C0 % 32 == 0), whose pixel stride
      // is ITS channel count
      const bool g_first = !GB || n0w < P.gb_C0;
      const int xc = GB ? (g_first ? P.gb_C0 : P.Nout - P.gb_C0) : P.Nout, xn0 = GB ? (g_first ? n0w : n0w - P.gb_C0) : n0w;
      const unsigned lane_y = (unsigned)((((er >> 2) * P.W + 2 * (er & 3)) * P.Nout + ec) * 4);
      const unsigned lane_x = rsh ? (unsigned)(((er & 3) * xc + ec) * 4) : (unsigned)((((er >> 2) * P.W + 2 * (er & 3)) * xc + ec) * 4);
      const unsigned lane_st = lane_e < 8 ? (unsigned)(lane_e * 8) : YOOB;
      // byte steps of the output: `it` = 2 rows, j = one pixel, a & 1 = 8 pixels, a >> 1 = 8 rows; of the residual likewise (half resolution: halved)
      const unsigned y_it = (unsigned)(2 * P.W * P.Nout * 4), y_j = (unsigned)(P.Nout * 4), y_a2 = (unsigned)(8 * P.Nout * 4), y_ar = (unsigned)(8 * P.W * P.Nout * 4);
      const unsigned x_it = rsh ? (unsigned)((P.W >> 1) * xc * 4) : (unsigned)(2 * P.W * xc * 4), x_j = rsh ? 0u : (unsigned)(xc * 4);
      const unsigned x_a2 = rsh ? (unsigned)(4 * xc * 4) : (unsigned)(8 * xc * 4);
      const unsigned x_ar = rsh ? (unsigned)(4 * (P.W >> 1) * xc * 4) : (unsigned)(8 * P.W * xc * 4);
      const unsigned d_rb4 = (unsigned)((((m_img * P.H + m_y0) * P.W + m_x0) * P.Nout + n0w) * 4);
      const unsigned d_xb4 = rsh ? (unsigned)((((m_img * (P.H >> 1) + (m_y0 >> 1)) * (P.W >> 1) + (m_x0 >> 1)) * xc + xn0) * 4)
                                 : (unsigned)((((m_img * P.H + m_y0) * P.W + m_x0) * xc + xn0) * 4);
      const unsigned d_sb8 = (unsigned)(((m_img * P.stat_tpi + (m_y0 / (8 * RH) * P.tiles_x + (m_x0 >> 4)) * RH) * (P.Nout >> 2) + (n0w >> 2)) * 8);      // + (a >> 1): the row half's entry
      const unsigned s_ar = (unsigned)((P.Nout >> 2) * 8);
      const float* const extra_ = GB ? (g_first ? P.gb_x0 : P.gb_x1) : (P.res_mode ? P.res : P.y);      // residual OR (accumulate) the previous contents of y (both: conv3x3y_launch falls back)
      const unsigned extra_on = (GB || P.res_mode || P.accumulate) ? YALL : 0u, stat_on = P.stat_part ? YALL : 0u, bias_on = P.bias ? YALL : 0u;
#define Y_RSB(PTR, BYTES) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>((const void*)(PTR)), 0, (int)(BYTES), 0x00020000)
      float4 rv[EX ? 2 : 1][2][EX ? 4 : 1];          // (RH = 1: two blocks, both requested up front)
      float st1 = 0.f, st2 = 0.f;
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);      // (GB: a data gradient has no bias -- four registers the sums need)
      if constexpr (!GB) bias4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(Y_RSB(P.bias, bias_on), ec * 4, (int)(n0w * 4), 0));
      // GB: coefficients of this lane's channel quad in the tile's image as  z = a x + b',  b' = b - a mu  (eight registers instead of twelve: the
      // kernel is at its register limit), and the tile's sums S0 = sum dv, S1x = sum dv x as float pairs; sum dv (x - mu) = S1x - mu S0 is formed per
      // lane at the end of the tile (mu is loaded again there -- the operand slots are free by then), i.e. the cancellation acts on 32-pixel partials
      // and costs the same rounding as a per-element subtraction
      float4 gba = make_float4(0.f, 0.f, 0.f, 0.f), gbb = gba;
      y_f32x2 gs0[2] = {y_f32x2{0.f, 0.f}, y_f32x2{0.f, 0.f}}, gs1[2] = {y_f32x2{0.f, 0.f}, y_f32x2{0.f, 0.f}};
      const unsigned g_nc4 = (unsigned)(P.N * P.Nout * 4), g_co4 = (unsigned)((m_img * P.Nout + n0w) * 4);
      if constexpr (GB) {
        const float4 m_ = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(Y_RSB(P.gb_coef, YALL), ec * 4, (int)g_co4, 0));
        gba = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(Y_RSB(P.gb_coef, YALL), ec * 4, (int)(g_co4 + g_nc4), 0));
        gbb = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(Y_RSB(P.gb_coef, YALL), ec * 4, (int)(g_co4 + 2u * g_nc4), 0));
        gbb.x = fmaf(-gba.x, m_.x, gbb.x); gbb.y = fmaf(-gba.y, m_.y, gbb.y); gbb.z = fmaf(-gba.z, m_.z, gbb.z); gbb.w = fmaf(-gba.w, m_.w, gbb.w);
      }
      auto epi_L = [&](int b) {                       // block b = m-tile a
        if constexpr (EX)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int it = 0; it < 4; ++it)
            rv[b & 1][j][it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(Y_RSB(extra_, extra_on), (int)lane_x,
                                                                                                 (int)(d_xb4 + it * x_it + j * x_j + (b & 1) * x_a2 + (b >> 1) * x_ar), 0));
      };
      auto epi_W = [&](int b) {
        // the accumulators are read HERE: without the opaque redefinition the sixteen-float extractions of all four blocks were hoisted to the top
        // of the epilogue (192 v_accvgpr_read up front, 42 spilled registers)
        asm volatile("" : "+a"(acc[0][b]), "+a"(acc[1][b]), "+a"(acc[2][b]), "+a"(acc[3][b]));
        // LDS STORE SOURCE HAZARD (found here in round 4): `ds_write2_b32 v40, v41, v42` followed one instruction later by
        // `v_accvgpr_read_b32 v42, a98` stored the NEW v42 in lanes 12-15 of every 16 -- a store's operands leave the register file over several
        // cycles (longer when two waves of a SIMD pair store at once) and the accumulator read is not interlocked against that.  So: all values of a
        // half block are computed first (no accumulator read between the stores) and a spacer separates the stores from the next reads;
        // tools/isa_hazard.py finds the pattern in the assembly, tests/test_kernel_resources_cpu.py asserts it is absent.
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          float z0[8], z1[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = hf * 8 + q;
            const float m0 = acc[0][b][r], m1 = acc[1][b][r], m2 = acc[2][b][r], m3 = acc[3][b][r];
            z0[q] = (m0 + m1) + m2;
            z1[q] = (m1 - m2) - m3;
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = hf * 8 + q;
            const unsigned o = tw_w + (unsigned)((((r & 3) + 8 * (r >> 2)) * EPW) * 4);
            *(y_lds_f)(size_t)(o) = z0[q];
            *(y_lds_f)(size_t)(o + TWB) = z1[q];
          }
          __builtin_amdgcn_sched_barrier(0);
          // (round 5) the spacer is now `s_waitcnt lgkmcnt(0)` -- the stores have COMPLETED, so their operands have certainly left the register file --
          // and it takes the sixteen stored values as inputs, so their registers stay allocated up to it.  The 16 idle cycles of round 4 were enough for
          // the 16-row instantiations; the 8-row ones with an epilogue operand (16 buffer loads issued right in front of these stores compete for the
          // same operand path) still stored the NEXT half block's values in lanes 12-15 of every 16, ~1.9 % of the outputs, varying from run to run.
          // Before that the same instantiation had formed the next store's ADDRESS in a dead data register one instruction behind the store
          // (`ds_write2_b32 v157, v174, v176` / `v_add_u32 v176, 0x400, v83`): an ordinary VALU write is not interlocked against the operand transfer
          // either (tools/isa_hazard.py flags any vector write now).
          asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(z0[0]), "v"(z0[1]), "v"(z0[2]), "v"(z0[3]), "v"(z0[4]), "v"(z0[5]), "v"(z0[6]), "v"(z0[7]),
                       "v"(z1[0]), "v"(z1[1]), "v"(z1[2]), "v"(z1[3]), "v"(z1[4]), "v"(z1[5]), "v"(z1[6]), "v"(z1[7]) : "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto epi_S = [&](int b) {
        float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);      // the value stored last (see OUTPUT STORE DATA HAZARD below)
        if ((b & 1) == 0) { st1 = 0.f; st2 = 0.f; }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const y_f32x4 v4 = *(y_lds_f4)(size_t)(tw_r + (unsigned)j * TWB + (unsigned)(it * 8 * EPW * 4));
            float4 v = make_float4(v4[0], v4[1], v4[2], v4[3]);
            float4 bb = bias4;
            if constexpr (EX && !GB) { const float4 u = rv[b & 1][j][it]; bb.x += u.x; bb.y += u.y; bb.z += u.z; bb.w += u.w; }
            if constexpr (GB) { v.x *= oscale; v.y *= oscale; v.z *= oscale; v.w *= oscale; }
            else { v.x = fmaf(v.x, oscale, bb.x); v.y = fmaf(v.y, oscale, bb.y); v.z = fmaf(v.z, oscale, bb.z); v.w = fmaf(v.w, oscale, bb.w); }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(y_u32x4, v), Y_RSB(P.y, YALL), (int)lane_y, (int)(d_rb4 + it * y_it + j * y_j + (b & 1) * y_a2 + (b >> 1) * y_ar), 0);
            // OUTPUT STORE DATA HAZARD (round 5).  The 8-row instantiation with an epilogue operand stored wrong values in lanes 12-15 of every 16 for
            // ~1.9 % of its outputs, varying from run to run: `buffer_store_dwordx4 v[162:165]` and, two instructions later, `v_pk_add_f32 v[162:163], ..`
            // -- the next value formed in the store's data registers while the store, queued behind the sixteen operand loads issued just before, had
            // not yet moved its data out of the register file (the lanes transferred last got the new value; the same signature as the LDS store
            // hazard of round 4).  A debug build with a 32-cycle spacer behind every store was correct.  The rule here: the
            // registers of a stored value stay allocated until the NEXT store has been issued (`pv`: its computation contains an LDS round trip,
            // >= 60 cycles), and the last one of a block is followed by an explicit spacer.  tests/test_kernel_resources_cpu.py scans for the pattern.
            // (GB: the stored value is read again by ~30 instructions of the sums below, which keeps its registers busy for longer than that)
            if constexpr (!GB) { asm volatile("" :: "v"(pv.x), "v"(pv.y), "v"(pv.z), "v"(pv.w)); pv = v; }
            if constexpr (GB) {
              // dv = dA * silu'(z), z = a x + b', silu'(z) = s (1 + z (1 - s)), s = 1 / (1 + 2^(-z log2 e)): the arithmetic of compute_dv (norm.hip)
              // on float pairs (packed VALU); act == 1, no dropout (launch check)
              const float4 u = rv[b & 1][j][it];
#pragma unroll
              for (int hh = 0; hh < 2; ++hh) {
                const y_f32x2 x2 = hh ? y_f32x2{u.z, u.w} : y_f32x2{u.x, u.y}, d2 = hh ? y_f32x2{v.z, v.w} : y_f32x2{v.x, v.y};
                const y_f32x2 a2 = hh ? y_f32x2{gba.z, gba.w} : y_f32x2{gba.x, gba.y}, b2 = hh ? y_f32x2{gbb.z, gbb.w} : y_f32x2{gbb.x, gbb.y};
                const y_f32x2 z = __builtin_elementwise_fma(a2, x2, b2);
                const y_f32x2 e = z * -1.4426950408889634f;
                const y_f32x2 p1 = y_f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + 1.0f;
                const y_f32x2 sg = y_f32x2{__builtin_amdgcn_rcpf(p1[0]), __builtin_amdgcn_rcpf(p1[1])};
                const y_f32x2 ds = sg * (1.0f + z * (1.0f - sg));
                const y_f32x2 dv = d2 * ds;
                gs0[hh] += dv;
                gs1[hh] = __builtin_elementwise_fma(dv, x2, gs1[hh]);
              }
            }
            if constexpr (ST) {
              st1 += (v.x + v.y) + (v.z + v.w);
              st2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, st2))));
            }
          }
        if constexpr (!GB) asm volatile("s_nop 7\n\ts_nop 7" :: "v"(pv.x), "v"(pv.y), "v"(pv.z), "v"(pv.w));      // the block's last store
        if (ST && (b & 1) == 1) {      // (sum, sum of squares) of a row half's 8 x 16 pixels per channel quad: the eight lanes holding a quad combine, lanes 0..7 write
          float s1 = st1, s2 = st2;
          s1 += __shfl_xor(s1, 8); s2 += __shfl_xor(s2, 8);
          s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
          s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
          const y_u32x2 sv2 = {__float_as_uint(s1), __float_as_uint(s2)};
          __builtin_amdgcn_raw_buffer_store_b64(sv2, Y_RSB(P.stat_part, stat_on), (int)lane_st, (int)(d_sb8 + (b >> 1) * s_ar), 0);
          asm volatile("s_nop 7\n\ts_nop 7" :: "v"(sv2[0]), "v"(sv2[1]));      // (the next block's accumulator reads took these registers two instructions later: output store data hazard)
        }
      };
#define Y_EPI_FENCE __builtin_amdgcn_sched_barrier(0);
      epi_L(0); epi_L(1); Y_EPI_FENCE
      epi_W(0); Y_EPI_FENCE epi_S(0); Y_EPI_FENCE Y_IF2(epi_L(2);) Y_EPI_FENCE
      epi_W(1); Y_EPI_FENCE epi_S(1); Y_EPI_FENCE Y_IF2(epi_L(3);) Y_EPI_FENCE
      Y_IF2(epi_W(2); Y_EPI_FENCE epi_S(2); Y_EPI_FENCE
            epi_W(3); Y_EPI_FENCE epi_S(3); Y_EPI_FENCE)
      if constexpr (GB) {      // the eight lanes that hold a channel quad combine (fixed order), lanes 0..7 write (S0, S1) of four channels: 32 bytes
        const float4 m_ = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(Y_RSB(P.gb_coef, YALL), ec * 4, (int)g_co4, 0));
        float sv8[8] = {gs0[0][0], fmaf(-m_.x, gs0[0][0], gs1[0][0]), gs0[0][1], fmaf(-m_.y, gs0[0][1], gs1[0][1]),
                        gs0[1][0], fmaf(-m_.z, gs0[1][0], gs1[1][0]), gs0[1][1], fmaf(-m_.w, gs0[1][1], gs1[1][1])};
#pragma unroll
        for (int k = 0; k < 8; ++k) { sv8[k] += __shfl_xor(sv8[k], 8); sv8[k] += __shfl_xor(sv8[k], 16); sv8[k] += __shfl_xor(sv8[k], 32); }
        const unsigned g_lane = lane_e < 8 ? (unsigned)(lane_e * 32) : YOOB;
        const unsigned g_off = (unsigned)((((m_img * P.gb_tpi + m_y0 / (8 * RH) * P.tiles_x + (m_x0 >> 4)) * P.Nout) + n0w) * 8);
        const y_u32x4 w0 = {__float_as_uint(sv8[0]), __float_as_uint(sv8[1]), __float_as_uint(sv8[2]), __float_as_uint(sv8[3])};
        const y_u32x4 w1 = {__float_as_uint(sv8[4]), __float_as_uint(sv8[5]), __float_as_uint(sv8[6]), __float_as_uint(sv8[7])};
        __builtin_amdgcn_raw_buffer_store_b128(w0, Y_RSB(P.gb_part, YALL), (int)g_lane, (int)g_off, 0);
        __builtin_amdgcn_raw_buffer_store_b128(w1, Y_RSB(P.gb_part, YALL), (int)g_lane, (int)(g_off + 16u), 0);
      }
#undef Y_EPI_FENCE
#undef Y_RSB
    }
#endif
    // the first step of the next tile becomes the one being multiplied
    const int t2 = m_tile + G;
    if (t2 >= ntiles) break;
    Y_DECODE(t2, m_img, m_y0, m_x0, m_n0)
    m_tile = t2; m_k = 0;
    {
      int t3;
      advance(m_tile, 0, t3, n_k);
      Y_DECODE_N0(t3 < ntiles ? t3 : m_tile, n_n0)
    }
  }
  if constexpr (NS == 4) pdae_sat_report(P.sat, sat_hit * (2.0f * ascale));      // |s| <= 2 max|d| after the transform
}

template <int NS, bool GN, bool EX, bool ST, bool GB = false, int RH = 2> static int launch_y(const PatchParams& P, hipStream_t s) {
  const size_t smem = (size_t)2 * NPL(NS) * YPLANE_B_(RH) + (size_t)4 * 2 * 32 * EPW * 4 + 4 * 256;      // two patch buffers + two transposition tiles per wave + the coefficient slots (f16x3, RH = 2: 162304 of 163840 bytes)
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3y_kernel<NS, GN, EX, ST, GB, RH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { pdae_set_error("conv3x3y: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  const long long ntiles = (long long)P.N * P.tiles_y * P.tiles_x * P.tiles_n;
  dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256));
  const int stagger = pdae_knob(KNOB_Y_STAGGER);
  auto magic = [](int d) { return (unsigned)((0x100000000ull / (unsigned long long)d) + 1ull); };      // unused for d == 1
  const YDiv D{magic(P.tiles_n), magic(P.tiles_x), magic(P.tiles_y)};
  if (ntiles >= (1ll << 20) || P.tiles_n >= 4096 || P.tiles_x >= 4096 || P.tiles_y >= 4096) { pdae_set_error("conv3x3y: %lld tiles", ntiles); return 1; }
  // Multi-GPU remedy for the LDS displacement measured in round 5: a co-resident communication kernel that holds LDS keeps a
  // persistent workgroup off its CU for the whole launch; with PDAE_Y_GRID_TRIM = k the launch uses 256 - k workgroups, leaving k CUs to the
  // collective's channels (default 0; bench.py records it in `comm`; only meaningful at world size > 1)
  const int trim = pdae_knob(KNOB_Y_GRID_TRIM);
  if (trim > 0 && trim < 128 && ntiles > 256 - trim) grid = dim3((unsigned)(256 - trim));
  hipLaunchKernelGGL((conv3x3y_kernel<NS, GN, EX, ST, GB, RH>), grid, dim3(YTHREADS), smem, s, P,
                     (ntiles >= 2 * 256 ? (stagger & 0xffff) : 0) | ((pdae_knob(KNOB_Y_XCD) == 2 || (pdae_knob(KNOB_Y_XCD) == 1 && P.tiles_n >= 2)) ? 0x10000 : 0), D);
  return pdae_launch_status("conv3x3y");
}

// ---- host side of the Winograd-along-x form (the decision used to live next to the two-waves-per-SIMD probe kernel conv3x3x, which is now under
// tools/probes/r04_winograd/ together with the 2-D F(2x2, 3x3) probe: neither was launched by any plan)

// Form of the prepared weights AND of the launch of a 3x3 convolution with these launch-side dimensions (C input channels, H x W output grid,
// Nout output channels): decided from the shape and the knob PDAE_W1 alone, so that weight preparation and launch agree.  The knob is read once
// (common.h); a prepared buffer is tagged with the form it was written in and a launch that expects the other form is refused (conv3x3p.hip).
int conv3x3x_rows(int math, int C, int H, int W, int N, int Nout) {
  // PDAE_W1: 0 off; 1 (default) by the fill rules below; 2: 16-row tiles on every eligible shape (8-row where H % 16 != 0); 3: 8-row tiles on every
  // eligible shape (2, 3: tests)
  const int m = pdae_knob(KNOB_W1);
  if (m == 0) return 0;
  if (!(math == 1 || math == 2 || math == 4)) return 0;
  if ((H % 8) || (W % PTW) || (Nout % PBN) || (C & 31) || H >= 2048 || W >= 2048) return 0;
  const unsigned long long lim = 0xFFFFFFE0ull;
  if ((unsigned long long)N * H * W * (unsigned long long)(C > Nout ? C : Nout) * 4ull >= lim) return 0;
  const bool h16 = (H % 16) == 0;
  if (m == 2) return h16 ? 2 : 1;
  if (m >= 3) return 1;
  const long long per8 = (long long)N * (W / PTW) * (Nout / PBN), tiles8 = per8 * (H / 8), tiles16 = h16 ? per8 * (H / 16) : 0;
  // persistent workgroups, one per CU: the last round of tiles must not leave the chip idle
  if (tiles16 >= 256 && tiles16 * 100 >= ((tiles16 + 255) / 256) * 256 * pdae_knob(KNOB_W1_EFF)) return 2;
  // 8-row tiles for what is left (round 5): at least PDAE_W1_MIN8 (160 = 5/8 of the CUs) tiles in a single round, or rounds filled to PDAE_W1_EFF8 %
  if (pdae_knob(KNOB_W1_ROWS8) && tiles8 >= pdae_knob(KNOB_W1_MIN8) && (tiles8 <= 256 || tiles8 * 100 >= ((tiles8 + 255) / 256) * 256 * pdae_knob(KNOB_W1_EFF8))) return 1;
  return 0;
}
bool conv3x3x_ok(int math, int C, int H, int W, int N, int Nout) { return conv3x3x_rows(math, C, H, W, N, Nout) != 0; }

// P: as conv3x3p_launch fills it; tiles of 16 x 16 (or 8 x 16) pixels x 128 channels, Winograd-along-x weights; no fused skip chunks in this form
int conv3x3x_launch(int math, const PatchParams& P0, hipStream_t s) {
  PatchParams P = P0;
  const int rows = conv3x3x_rows(math, P.C, P.H, P.W, P.N, P.Nout);
  if (!rows) { pdae_set_error("conv3x3y: shape not eligible for the Winograd form"); return PDAE_EINVAL; }
  P.tiles_x = P.W / PTW; P.tiles_y = P.H / (8 * rows); P.tiles_n = P.Nout / PBN; P.splits = 1; P.cps = P.C >> 5;
  if (P.nx) { pdae_set_error("conv3x3y: fused skip chunks are not built for the Winograd form"); return PDAE_EINVAL; }
  if (P.x1 && (P.C0 & 31)) { pdae_set_error("conv3x3y: two-source input needs C0 %% 32 == 0"); return PDAE_EINVAL; }
  return conv3x3y_launch(math, P, s, rows);
}

// P: as prepared by conv3x3x_launch (tiles of 16 x 16 pixels x 128 channels, Winograd-along-x weights); no fused skip chunks
int conv3x3y_launch(int math, const PatchParams& P, hipStream_t s, int rows) {
  if (P.res_mode && P.accumulate) { pdae_set_error("conv3x3y: residual and accumulate in one launch"); return 1; }
  if (P.gb_part) {      // data gradient + GroupNorm-backward sums: the operand slots carry the GroupNorm input
    if (P.res_mode || P.accumulate || P.coef || P.stat_part || (P.gb_C0 & 31) || ((P.Nout - P.gb_C0) & 31) || P.gb_tpi != P.tiles_x * P.tiles_y) {
      pdae_set_error("conv3x3y: GroupNorm-backward sums need a plain data gradient (no residual / accumulate / fused input) and 32-channel-aligned sources");
      return PDAE_EINVAL;
    }
#define PDAE_YG(NS_) (rows == 1 ? launch_y<NS_, false, true, false, true, 1>(P, s) : launch_y<NS_, false, true, false, true, 2>(P, s))
    if (math == 1) return PDAE_YG(1);
    if (math == 2) return PDAE_YG(2);
    return PDAE_YG(4);
#undef PDAE_YG
  }
#define PDAE_Y1(NS_, GN_, EX_, ST_) (rows == 1 ? launch_y<NS_, GN_, EX_, ST_, false, 1>(P, s) : launch_y<NS_, GN_, EX_, ST_, false, 2>(P, s))
#define PDAE_Y2(NS_, GN_) (ex ? (st ? PDAE_Y1(NS_, GN_, true, true) : PDAE_Y1(NS_, GN_, true, false))       \
                              : (st ? PDAE_Y1(NS_, GN_, false, true) : PDAE_Y1(NS_, GN_, false, false)))
#define PDAE_Y3(NS_) (P.coef ? PDAE_Y2(NS_, true) : PDAE_Y2(NS_, false))
  const bool ex = P.res_mode || P.accumulate, st = P.stat_part != nullptr;
  if (math == 1) return PDAE_Y3(1);
  if (math == 2) return PDAE_Y3(2);
  return PDAE_Y3(4);
#undef PDAE_Y3
#undef PDAE_Y2
#undef PDAE_Y1
}
