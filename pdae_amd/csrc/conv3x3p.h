// Shared pieces of the 3x3 "patch" convolution kernels (conv3x3p.hip: 4 / 8 waves of 128 px x 32 ch at two waves per SIMD, one tile per
// workgroup; conv3x3r.hip: persistent workgroups of 4 waves of 128 px x 64 ch at one wave per SIMD, deferred epilogue): operand formats,
// LDS patch geometry, launch parameters.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// NS encodes the operand format: 1..3 = that many bf16 planes; 4 = TWO fp16 planes (11 + 11 mantissa bits, 3 products ~2^-21:
// fp32-grade at half the MFMAs of bf16x6, for operands inside the fp16 range -- forward activations / weights)
#define NPL(NS_) ((NS_) == 4 ? 2 : (NS_))
#define PASCALE 16.0f            // fp16 format: activations are multiplied by 2^4 before the split (undone exactly in the epilogue)

// Geometries (template PTH, W8):
//   PTH = 16      : 16x16-pixel tile, 512 threads (4x2 waves), 86 KB LDS at NS=3 (one block per CU)
//   PTH =  8      :  8x16-pixel tile, 256 threads (2x2 waves), 48 KB LDS (several independent blocks per CU)
//   PTH =  8, W8  :  8-pixel-wide images: the tile is 8 rows of TWO images side by side (their 10-pixel halo rows fill the
//                    20-pixel patch pitch exactly), so the 8x8 bottleneck layers run on the same kernel
// Small layers (few tiles) are additionally split over ranges of input-channel chunks (split-K): every split writes its
// partial tile to a slab behind the prepared weights and conv3x3p_reduce adds the slabs, bias and residual in fixed order.
#define PLDH 40                 // bf16 per LDS row (32 + 8 pad): 80-byte rows
#define PTW 16
#define PPW 20                  // patch pitch in pixels (18 used): with the 4x8 fragment blocks below every ds_read_b128 is conflict-free
#define PBN 128
#define PSLOT(row, slot) ((row) * PLDH + ((slot) << 3))       // bf16 offset of 16-byte k-slot `slot` of `row`
#define PPLANE(rows) ((rows) * PLDH)
#define EPW 36                  // floats per row of the epilogue transpose tile (32 + 4 pad)

__device__ __forceinline__ float p_trunc(float a) { return __uint_as_float(__float_as_uint(a) & 0xffff0000u); }
__device__ __forceinline__ unsigned p_hi16(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }
__device__ __forceinline__ unsigned p_rn(float a, float b) {
  unsigned short x = __builtin_bit_cast(unsigned short, (__bf16)a), y = __builtin_bit_cast(unsigned short, (__bf16)b);
  return (unsigned)x | ((unsigned)y << 16);
}
template <int NS> __device__ __forceinline__ void p_split2(float e0, float e1, unsigned (&w)[NPL(NS)], float sc = 1.0f) {
  if constexpr (NS == 4) {          // fp16 planes of e * sc (the other formats take no scale)
    pdae_f16_split2s(e0, e1, sc, w[0], w[1]);
  } else if constexpr (NS == 1) { w[0] = p_rn(e0, e1); }
  else {
    float h0 = p_trunc(e0), h1 = p_trunc(e1);
    float r0 = e0 - h0, r1 = e1 - h1;
    w[0] = p_hi16(h0, h1);
    if constexpr (NS == 2) { w[1] = p_rn(r0, r1); }
    else {
      float m0 = p_trunc(r0), m1 = p_trunc(r1);
      w[1] = p_hi16(m0, m1);
      w[2] = p_hi16(r0 - m0, r1 - m1);
    }
  }
}

struct PatchParams {
  const float* x; int N, Hs, Ws, C;     // stored input [N,Hs,Ws,C]
  int H, W, up;                         // output (= logical input) size; up: stored = logical >> 1
  const unsigned short* wp;             // pre-split weights [NS][C/32][9][2][NT][64][8] bf16 (conv3x3p_wprep)
  int NT;                               // 32-channel output tiles in wp (= ceil(Nout/32))
  int Nout;                             // GEMM N
  float* y; const float* bias; const float* res; int res_mode; int accumulate;
  int tiles_x, tiles_y, tiles_n;
  int splits, cps;                      // split-K: `splits` ranges of `cps` chunks; splits > 1 => raw partials to slab[split][M][Nout]
  float* slab;
  // fused GroupNorm / AdaGN + SiLU on the input (GN instantiation): the conv reads act(a[n,c] * (x - mu[n,c]) + b[n,c]) of the virtual
  // concat [x | x1] (C0 channels in x), coefficients coef = [mu | a | b] each [N][C] from pdae_gn_coef; zero padding applies AFTER the map
  const float* x1; int C0; const float* coef; int act;
  // fused 1x1 skip convolution (ResBlock skip_connection, module.py:276,297): nx extra 32-channel chunks of the raw two-source tensor
  // [s0 | s1] enter the K loop with the centre tap only, weights wps = conv1x1_wprep layout, bias_x added in the epilogue
  int nx; const float* s0; const float* s1; int Cs0, Cs1; const unsigned short* wps; const float* bias_x;
  float woscale;                        // fp16 format: 1 / (power-of-two scale of the prepared weights, conv3x3p_wscale)
  const float* amax;                    // fp16 format: NULL = activations (static 2^4 pre-scale); else device scalar max|input| (pdae_amax)
                                        // -> power-of-two scale putting the input's abs-max into [1024, 2048): gradients (dY) as input
  unsigned int* sat;                    // fp16 format: saturation counter (common.h) or NULL
  // GroupNorm statistics of the OUTPUT, fused into the epilogue (splits == 1 only): every wave writes (sum, sum of squares) of its 128 pixels
  // for each of its eight channel quads to stat_part[image][stat_tpi wave-tiles][Nout / 4] as float2; pdae_gn_coef_from_conv_stats sums the
  // wave-tiles in fp64.  The next GroupNorm then needs no pass over this tensor (it was 9 % of a sampling step).
  float* stat_part; int stat_tpi;
  // GroupNorm-BACKWARD sums of a data gradient's output dA (conv3x3y, GB instantiation; PatchGnb in igemm.h): the launch loads the GroupNorm's raw
  // input x as its epilogue operand, forms dv = dA * silu'(a (x - mu) + b) for every value it stores and leaves (sum dv, sum dv (x - mu)) per channel
  // and 16 x 16 tile in gb_part[image][gb_tpi][Nout][2] -- the layout gn_bwd_reduce_kernel writes, so gn_bwd_finalize / _apply run unchanged and the
  // separate reduction pass over (x, dA) (2 of the 5 tensor passes of a GroupNorm backward) does not exist.
  const float* gb_x0; const float* gb_x1; int gb_C0; const float* gb_coef; float* gb_part; int gb_tpi;
};

// 2^(10 - floor(log2(amax))): amax * scale in [1024, 2048)  (amax == 0 or non-finite: 1)
__device__ __forceinline__ float p_pow2_scale(float amax) {
  const int ex = (__float_as_int(amax) >> 23) & 0xff;
  if (ex == 0 || ex == 255) return 1.0f;
  int sb = 127 + 10 - (ex - 127);
  sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
  return __int_as_float(sb << 23);
}
// SiLU in 5 VALU instructions (v_mul, v_exp_f32, v_add, v_rcp_f32, v_mul; ~2 ulp) instead of ~17 for expf + IEEE division: the fused-GroupNorm
// variant evaluates it once per staged element per block, and under the power cap every VALU instruction is paid for in matrix throughput
__device__ __forceinline__ float p_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }


// ---- weight preparation, one fragment slot (8 consecutive k of one output channel) per call; shared by the per-convolution kernels and the
// grouped launch (wprep.hip).  3x3 layout: wp [NS][C/32][T][2][NT][64][8] (T taps: 9, or 1 for fused 1x1 skip chunks); 1x1 layout:
// wp [NS][C/16][NT][64][8].  transposed: the data-gradient weights (3x3: w'[n][tap][c] = w[c][T-1-tap][n], w stored [C][T][Nout]).
template <int NS> __device__ __forceinline__ void wprep_store_slot(float (&e)[8], float wscale, unsigned short* __restrict__ wp, size_t plane_stride, size_t i) {
  if constexpr (NS == 4) {
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] *= wscale;
  }
  unsigned a[NPL(NS)], b[NPL(NS)], cc[NPL(NS)], d[NPL(NS)];
  p_split2<NS>(e[0], e[1], a); p_split2<NS>(e[2], e[3], b); p_split2<NS>(e[4], e[5], cc); p_split2<NS>(e[6], e[7], d);
#pragma unroll
  for (int p = 0; p < NPL(NS); ++p) *reinterpret_cast<uint4*>(wp + p * plane_stride + i * 8) = make_uint4(a[p], b[p], cc[p], d[p]);
}
template <int NS> __device__ __forceinline__ void wprep3_slot(const float* __restrict__ w, int Nout, int C, int NT, int transposed, float wscale, int T,
                                                              unsigned short* __restrict__ wp, size_t i) {
  const size_t plane_stride = (size_t)(C >> 5) * 2 * T * NT * 64 * 8;
  const int lane = (int)(i & 63); size_t r = i >> 6;
  const int nt = (int)(r % NT); r /= NT;
  const int kc = (int)(r & 1); r >>= 1;
  const int tap = (int)(r % T); const int chunk = (int)(r / T);
  const int n = nt * 32 + (lane & 31), c = (chunk << 5) + kc * 16 + (lane >> 5) * 8;
  float e[8];
  if (n < Nout && transposed) {
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = w[((size_t)(c + j) * T + (T - 1 - tap)) * Nout + n];
  } else if (n < Nout) {
    const float4 v0 = *reinterpret_cast<const float4*>(w + ((size_t)n * T + tap) * C + c);
    const float4 v1 = *reinterpret_cast<const float4*>(w + ((size_t)n * T + tap) * C + c + 4);
    e[0] = v0.x; e[1] = v0.y; e[2] = v0.z; e[3] = v0.w; e[4] = v1.x; e[5] = v1.y; e[6] = v1.z; e[7] = v1.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = 0.f;
  }
  wprep_store_slot<NS>(e, wscale, wp, plane_stride, i);
}
template <int NS> __device__ __forceinline__ void wprep1_slot(const float* __restrict__ w, int Nout, int C, int NT, int transposed, float wscale,
                                                              unsigned short* __restrict__ wp, size_t i) {
  const size_t plane_stride = (size_t)(C >> 4) * NT * 64 * 8;
  const int lane = (int)(i & 63); const size_t r = i >> 6;
  const int nt = (int)(r % NT); const int s = (int)(r / NT);
  const int n = nt * 32 + (lane & 31), c = (s << 4) + (lane >> 5) * 8;
  float e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = 0.f;
  if (n < Nout) {
    if (transposed) {
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = w[(size_t)(c + j) * Nout + n];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = w[(size_t)n * C + c + j];
    }
  }
  wprep_store_slot<NS>(e, wscale, wp, plane_stride, i);
}
// Prepared weights of the Winograd-along-x form (conv3x3x.hip): wp [NS][C/32][T][2][NT][64][8] with T = 12 transform taps tp = ky * 4 + c,
// U[ky][c] = sum_kx G[c][kx] w[ky][kx], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] (exact factors); or T = 2 for fused 1x1 skip chunks: the
// centre tap's positions c = 1, 2 with G[c][1] = +-1/2.  transposed: the data-gradient weights (taps flipped), as wprep3_slot.
template <int NS> __device__ __forceinline__ void wprepx_slot(const float* __restrict__ w, int Nout, int C, int NT, int transposed, float wscale, int T,
                                                              unsigned short* __restrict__ wp, size_t i) {
  const size_t plane_stride = (size_t)(C >> 5) * 2 * T * NT * 64 * 8;
  const int lane = (int)(i & 63); size_t r = i >> 6;
  const int nt = (int)(r % NT); r /= NT;
  const int kc = (int)(r & 1); r >>= 1;
  const int tp = (int)(r % T); const int chunk = (int)(r / T);
  const int n = nt * 32 + (lane & 31), c0 = (chunk << 5) + kc * 16 + (lane >> 5) * 8;
  float e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = 0.f;
  if (n < Nout) {
    if (T == 2) {                                 // skip chunk: w [Nout][C] (1x1)
      const float g = tp == 0 ? 0.5f : -0.5f;
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = g * w[(size_t)n * C + c0 + j];
    } else {
      const int ky = tp >> 2, cc = tp & 3;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float g = cc == 0 ? (kx == 0 ? 1.f : 0.f) : (cc == 3 ? (kx == 2 ? 1.f : 0.f) : ((cc == 2 && kx == 1) ? -0.5f : 0.5f));
        if (g == 0.f) continue;
        const int tap = ky * 3 + kx;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = transposed ? w[((size_t)(c0 + j) * 9 + (8 - tap)) * Nout + n] : w[((size_t)n * 9 + tap) * C + c0 + j];
          e[j] = fmaf(g, v, e[j]);
        }
      }
    }
  }
  wprep_store_slot<NS>(e, wscale, wp, plane_stride, i);
}
#define PDAE_MATH_DIRECT_BIT 0x100   // = PDAE_MATH_DIRECT of the C ABI, carried in the `math` argument of conv3x3p_form / _launch / _wprep(_job) / _skip_ok
#define PDAE_WPREP_FORM_X 8        // WprepJob.transposed bit: Winograd-along-x layout (wprepx_slot)

// one job of the grouped launch (= include/pdae_hip.h: pdae_wprep_job) and how the existing entry points describe theirs
struct WprepJob { const float* w; unsigned short* wp; int Nout, C, NT, transposed, T, ns; float wscale; int nblocks; };
// (H, W, N: the launch-side output grid and batch -- they decide between the direct and the Winograd-along-x layout, conv3x3p_form)
void conv3x3p_wprep_job(int math, const float* w, int Nout, int C, int transposed, unsigned short* wp, WprepJob* j, int H, int W, int N);
void conv3x3p_skip_wprep_job(int math, const float* w, int Nout, int Cs, int Cmain, unsigned short* wp, WprepJob* j, int H, int W, int N);
void conv1x1_wprep_job(int math, const float* w, int Nrows, int C, int transposed, unsigned short* wp, WprepJob* j);
int wprep_group_launch(const WprepJob* jobs_dev, const int* first_block_dev, int njobs, int total_blocks, hipStream_t s);

// conv3x3y.hip: Winograd F(2, 3) along x (two thirds of the matrix work) on the large layers; conv3x3p_form = 1 when a convolution with these
// launch-side dimensions is prepared AND launched in that form
bool conv3x3x_ok(int math, int C, int H, int W, int N, int Nout);
int conv3x3x_rows(int math, int C, int H, int W, int N, int Nout);       // 0: direct form; 2: tiles of 16 rows; 1: tiles of 8 rows
int conv3x3x_launch(int math, const PatchParams& P, hipStream_t s);
int conv3x3y_launch(int math, const PatchParams& P, hipStream_t s, int rows);      // conv3x3y.hip: persistent, one wave per SIMD (no fused skip chunks)
int conv3x3p_form(int math, int C, int H, int W, int N, int Nout);
// conv3x3r.hip: persistent workgroups with a deferred epilogue for layers with at least two 16 x 16 x 128-channel tiles per CU
bool conv3x3r_ok(int math, int C, int H, int W, int N, int Nout, int Hs, int Ws, int C0, int Cs0, int Cs1);
int conv3x3r_launch(int math, const PatchParams& P, hipStream_t s);
