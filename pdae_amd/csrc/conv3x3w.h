// Shared device helpers and the parameter block of the 3x3 weight-gradient kernels (conv3x3w.hip: two workgroups per CU, every wave stages and
// multiplies; conv3x3v.hip: one workgroup per CU, four matrix waves fed by four staging waves).  Internal; the C ABI is include/pdae_hip.h.
#pragma once
#include "common.h"
#include "igemm.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// NS = 4: two fp16 planes, 3 products (see conv3x3p.hip); dY is scaled by a per-launch power of two derived from its abs-max
#define WNPL(NS_) ((NS_) == 4 ? 2 : (NS_))
#define WXSCALE 16.0f

#define WTH 8
#define WTW 16
#define WPW_(W8_) ((W8_) ? 20 : WTW + 2)          // 8-pixel-wide images: two images side by side, their 10-pixel halo rows = pitch 20
#define WNPIX_(W8_) ((WTH + 2) * WPW_(W8_))     // 180 (200) patch pixels
#define WTPIX (WTH * WTW)            // 128 tile pixels
// LDS row strides chosen for the transposing read: the 32 lanes of one ds_read_b64_tr_b16 group address 4 pixel rows x 16 channels (2 dwords
// per lane); they are conflict-free when the 4 rows start 16 banks (dwords) apart.  X rows are exactly 32 bf16 = 16 dwords: no padding needed.
// 64-channel dY rows (32 dwords) put rows r and r+2 on the same banks: their channel index is XOR-swizzled with bit 1 of the pixel row * 32,
// so that rows 0..3 of a read land on bank groups {a, 2+a, a^1, 2+(a^1)} (a = the wave's 32-channel half) instead of being padded
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 0.50 with padded 40 / 136 strides, profiles/r02_pmc_sq.txt).
#define WCO 64                       // output channels per block
#define WSX 32                       // X row stride (bf16): 32 ci
#define WSY WCO                      // dY row stride (bf16): 64 co, swizzled
#define WSWZ(pix) ((((pix) >> 1) & 1) << 5)   // bf16 column XOR of pixel row `pix`
#define WTHREADS 256
#define WX_LD_(W8_) ((WNPIX_(W8_) * 8 + WTHREADS - 1) / WTHREADS)    // 6 (7) float4 per thread
#define WY_LD (WTPIX * (WCO / 4) / WTHREADS)             // 8 float4 per thread

__device__ __forceinline__ float w_trunc(float a) { return __uint_as_float(__float_as_uint(a) & 0xffff0000u); }
__device__ __forceinline__ unsigned w_hi16(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }
__device__ __forceinline__ unsigned w_rn(float a, float b) {
  unsigned short x = __builtin_bit_cast(unsigned short, (__bf16)a), y = __builtin_bit_cast(unsigned short, (__bf16)b);
  return (unsigned)x | ((unsigned)y << 16);
}
// 2^(10 - floor(log2(amax))): amax * scale in [1024, 2048)  (amax == 0 or non-finite: 1)
__device__ __forceinline__ float w_pow2_scale(float amax) {
  const int ex = (__float_as_int(amax) >> 23) & 0xff;
  if (ex == 0 || ex == 255) return 1.0f;
  int sb = 127 + 10 - (ex - 127);
  sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
  return __int_as_float(sb << 23);
}
template <int NS> __device__ __forceinline__ void w_split2(float e0, float e1, unsigned (&w)[WNPL(NS)], float sc = 1.0f) {
  if constexpr (NS == 4) {          // fp16 planes of e * sc (the other formats take no scale)
    pdae_f16_split2s(e0, e1, sc, w[0], w[1]);
  } else if constexpr (NS == 1) { w[0] = w_rn(e0, e1); }
  else {
    float h0 = w_trunc(e0), h1 = w_trunc(e1);
    float r0 = e0 - h0, r1 = e1 - h1;
    w[0] = w_hi16(h0, h1);
    if constexpr (NS == 2) { w[1] = w_rn(r0, r1); }
    else {
      float m0 = w_trunc(r0), m1 = w_trunc(r1);
      w[1] = w_hi16(m0, m1);
      w[2] = w_hi16(r0 - m0, r1 - m1);
    }
  }
}

// act(a * (x - mu) + b) on a channel quad, on float pairs (v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32: half the VALU issue slots), SiLU through
// v_exp_f32 / v_rcp_f32 -- the arithmetic of gn_apply_stream_kernel (norm.hip) and of conv3x3y's gn_map; `on` = 0: a padding pixel (zero AFTER the map)
typedef float w_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 w_gn_map(const float4& v, unsigned on, const float4& mu, const float4& sc, const float4& sh, int act) {
  const float onf = on ? 1.0f : 0.0f;
  w_f32x2 r[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const w_f32x2 x2 = hh ? w_f32x2{v.z, v.w} : w_f32x2{v.x, v.y};
    const w_f32x2 mu2 = hh ? w_f32x2{mu.z, mu.w} : w_f32x2{mu.x, mu.y};
    const w_f32x2 sc2 = hh ? w_f32x2{sc.z, sc.w} : w_f32x2{sc.x, sc.y};
    const w_f32x2 sh2 = hh ? w_f32x2{sh.z, sh.w} : w_f32x2{sh.x, sh.y};
    const w_f32x2 m = __builtin_elementwise_fma(sc2, x2 - mu2, sh2);
    if (act) {
      const w_f32x2 a = m * -1.4426950408889634f;
      const w_f32x2 d = w_f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + 1.0f;
      r[hh] = m * w_f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])} * onf;
    } else r[hh] = m * onf;
  }
  return make_float4(r[0][0], r[0][1], r[1][0], r[1][1]);
}

// two transposing reads -> 8 consecutive k (pixel rows r0..r0+7 as seen by this lane's half) of this lane's column.  The compiler builtin
// (not inline asm) so that hipcc tracks the LDS counter itself and can keep the next tap's fragments in flight under this tap's MFMAs
// (an asm statement needs its own "s_waitcnt lgkmcnt(0)", which exposes the LDS latency once per tap).
typedef short w_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 tr_frag(unsigned addr_lo, unsigned addr_hi) {
  typedef __attribute__((address_space(3))) w_s16x4* lds_ptr;
  const w_s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(size_t)addr_lo);
  const w_s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(size_t)addr_hi);
  const uint2 a = __builtin_bit_cast(uint2, v0), b = __builtin_bit_cast(uint2, v1);
  return make_uint4(a.x, a.y, b.x, b.y);
}

struct WgradParams {
  const float* x; int N, Hs, Ws, C;      // stored input [N,Hs,Ws,C] (C = Cin)
  int H, W, up;                          // conv grid (output size == logical input size)
  const float* dy; int Cout;             // dY [N,H,W,Cout]
  float* ws;                             // split-K slabs [splits][Cout][9][C]
  int tiles_x, tiles_y, ntiles;          // pixel tiles per image / total
  int tiles_per_split, splits;
  int co_tiles, ci_chunks;
  const float* dy_amax;                  // fp16 format: device scalar max|dY| (pdae_amax) -> power-of-two dY scale
  float* db_part;                        // optional bias-gradient partials [splits][Cout] (column sums of dY, written by the ci_chunk 0 blocks)
  unsigned int* sat;                     // fp16 format: saturation counter (common.h) or NULL
  int stagger;                           // start delay (units of 64 clocks) of every second block arriving on a CU, see w3_phase_offset
  // GN instantiation: X is act(a[n,c] * (x - mu[n,c]) + b[n,c]) of the RAW virtual concat [x | x1] (C0 channels in x), recomputed while the
  // patch is staged -- the activated tensor of the forward pass (module.py:241,279-284: in_layers GroupNorm + SiLU) is then never written,
  // saved or re-read; coef = [mu | a | b] each [N][C] as pdae_gn_coef leaves them; zero padding applies AFTER the map.
  const float* x1; int C0; const float* coef; int act;
};
