// 3x3 / stride-1 / pad-1 convolution (forward and data-gradient) for gfx950 on the bf16 MFMA pipe with split fp32
// operands -- the "patch" kernel.
//
// Block = 512 threads (8 waves as 4(M) x 2(N)), output tile = 16x16 pixels x 128 output channels.
// For every 32-channel chunk of the input the 18x18-pixel halo patch is staged in LDS ONCE (as NS bf16 planes, see
// igemm.hip) and all 9 taps read their shifted A fragments straight out of it, so the input crosses L2->CU ~1.3x
// instead of 9x; the weight panel of one tap (32 x 128) is double-buffered in LDS with its global load in flight under
// the previous tap's MFMAs: one barrier per tap (~48 MFMAs per wave in the 6-product mode).
//
// Replaces F.conv2d(k=3, padding=1) of model/module.py:242,265 (+ nearest upsample :169) and its input gradient.
#include <stdlib.h>

#include "common.h"
#include "igemm.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Two geometries (template PTH):
//   PTH = 16 : 16x16-pixel tile, 512 threads (4x2 waves), weight panel double-buffered, one block per CU (148 KB LDS at NS=3)
//   PTH =  8 :  8x16-pixel tile, 256 threads (2x2 waves), weight panel single-buffered, 74 KB LDS -> TWO independent blocks
//              per CU, so one block's staging / barrier phases run under the other block's MFMAs
#define PLDH 40                 // bf16 per LDS row (32 + 8 pad): 80-byte rows
#define PTW 16
#define PPW 20                  // patch pitch in pixels (18 used): with the 4x8 fragment blocks below every ds_read_b128 is conflict-free
#define PBN 128
#define PSLOT(row, slot) ((row) * PLDH + ((slot) << 3))       // bf16 offset of 16-byte k-slot `slot` of `row`
#define PPLANE(rows) ((rows) * PLDH)

__device__ __forceinline__ float p_trunc(float a) { return __uint_as_float(__float_as_uint(a) & 0xffff0000u); }
__device__ __forceinline__ unsigned p_hi16(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }
__device__ __forceinline__ unsigned p_rn(float a, float b) {
  unsigned short x = __builtin_bit_cast(unsigned short, (__bf16)a), y = __builtin_bit_cast(unsigned short, (__bf16)b);
  return (unsigned)x | ((unsigned)y << 16);
}
template <int NS> __device__ __forceinline__ void p_split2(float e0, float e1, unsigned (&w)[NS]) {
  if constexpr (NS == 1) { w[0] = p_rn(e0, e1); }
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
  const float* w;                       // weights [Nout][9][C] (for dgrad: the tap-flipped transposed copy, pdae_conv3x3_wtranspose)
  int Nout;                             // GEMM N
  float* y; const float* bias; const float* res; int res_mode; int accumulate;
  int tiles_x, tiles_y, tiles_n;
};

template <int NS, int PTH>
__global__ void __launch_bounds__(PTH * 32) conv3x3p_kernel(const PatchParams P) {
  constexpr int PTHREADS = PTH * 32;                      // 512 | 256
  constexpr int PNPIX = (PTH + 2) * PPW;                  // 324 | 180
  constexpr int PA_LD = (PNPIX * 8 + PTHREADS - 1) / PTHREADS;
  constexpr int BREP = 512 / PTHREADS;                    // weight-panel passes per thread
  constexpr bool DBUF = PTH == 16;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  constexpr int SA = NS * PPLANE(PNPIX);        // A patch planes
  constexpr int SB = NS * PPLANE(PBN);          // one B buffer
  unsigned short* sA = smem;
  unsigned short* sB = smem + SA;                // two buffers: sB, sB + SB

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, h = lane >> 5;
  const int wm = wv >> 1, wn = wv & 1;           // 4 x 2 waves; wave tile = 64 pixels x 64 channels

  // block -> (image, tile_y, tile_x, n-tile); n-tile fastest so the blocks sharing a patch are co-scheduled
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  int tid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int tn_i = tid % P.tiles_n; tid /= P.tiles_n;
  const int tx_i = tid % P.tiles_x; tid /= P.tiles_x;
  const int ty_i = tid % P.tiles_y; const int img = tid / P.tiles_y;
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
      int ly = y0 - 1 + py, lx = x0 - 1 + px;
      if (px < PTW + 2 && (unsigned)ly < (unsigned)P.H && (unsigned)lx < (unsigned)P.W) {
        int sy = P.up ? ly >> 1 : ly, sx = P.up ? lx >> 1 : lx;
        aoff[l] = ((long long)(img * P.Hs + sy) * P.Ws + sx) * C + qd * 4;
      }
    }
  }
  float4 apre[PA_LD];
  auto a_gload = [&](int c0) {
#pragma unroll
    for (int l = 0; l < PA_LD; ++l)
      apre[l] = aoff[l] >= 0 ? *reinterpret_cast<const float4*>(P.x + aoff[l] + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto a_lstore = [&]() {
#pragma unroll
    for (int l = 0; l < PA_LD; ++l) {
      int idx = t + PTHREADS * l;
      int pix = idx >> 3, qd = idx & 7;
      if (pix < PNPIX) {
        unsigned a[NS], b[NS];
        p_split2<NS>(apre[l].x, apre[l].y, a);
        p_split2<NS>(apre[l].z, apre[l].w, b);
#pragma unroll
        for (int p = 0; p < NS; ++p) *reinterpret_cast<uint2*>(&sA[p * PPLANE(PNPIX) + PSLOT(pix, qd >> 1) + (qd & 1) * 4]) = make_uint2(a[p], b[p]);
      }
    }
  };

  // ---- B panel of one (chunk, tap): 32 k x 128 n
  const int T9 = 9;
  // weight panel of one (chunk, tap): 128 rows (output channels) x 32 k, K-contiguous in memory ([Nout][9][C])
  auto b_gload = [&](float4 (&bpre)[2 * BREP], int chunk, int tap) {
#pragma unroll
    for (int rep = 0; rep < BREP; ++rep) {
      const int tt = t + PTHREADS * rep;
      int n = n0 + (tt >> 2), k8 = (tt & 3) * 8;
      if (n < P.Nout) {
        const float* src = P.w + ((size_t)n * T9 + tap) * C + (chunk << 5) + k8;
        bpre[2 * rep] = *reinterpret_cast<const float4*>(src);
        bpre[2 * rep + 1] = *reinterpret_cast<const float4*>(src + 4);
      } else { bpre[2 * rep] = bpre[2 * rep + 1] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
  };
  auto b_lstore = [&](const float4 (&bpre)[2 * BREP], unsigned short* sb) {
#pragma unroll
    for (int rep = 0; rep < BREP; ++rep) {
      const int tt = t + PTHREADS * rep;
      const float4 v0 = bpre[2 * rep], v1 = bpre[2 * rep + 1];
      int nl = tt >> 2, k8 = (tt & 3) * 8;
      unsigned a[NS], b[NS], c[NS], d[NS];
      p_split2<NS>(v0.x, v0.y, a); p_split2<NS>(v0.z, v0.w, b);
      p_split2<NS>(v1.x, v1.y, c); p_split2<NS>(v1.z, v1.w, d);
#pragma unroll
      for (int p = 0; p < NS; ++p) *reinterpret_cast<uint4*>(&sb[p * PPLANE(PBN) + PSLOT(nl, k8 >> 3)]) = make_uint4(a[p], b[p], c[p], d[p]);
    }
  };

  // ---- A fragment rows of this wave
  // MFMA row i of 32-row group g = wm*2+a <-> pixel (by*8 + i/4, bx*4 + i%4), (bx, by) = (g & 3, g >> 2): with the 20-pixel
  // pitch and 80-byte rows the 16 lanes of every ds_read_b128 group hit 16 distinct 16-byte slots for all 9 tap shifts
  int apix[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) { int g = wm * 2 + a; apix[a] = ((g >> 2) * 8 + (li >> 2)) * PPW + (g & 3) * 4 + (li & 3); }

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nchunk = C >> 5;
  const int total = 9 * nchunk;                   // (chunk, tap) steps
  float4 r0[2 * BREP], r1[2 * BREP];              // weight panels in flight: the panel of step s+2 is loaded during step s,
                                                  // stored to LDS during step s+1 -- its L2 latency is off the critical path
  auto compute = [&](int tap, const unsigned short* sb) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const int ashift = dy * PPW + dx;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      bf16x8 af[2][NS], bfr[2][NS];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int p = 0; p < NS; ++p)
          af[a][p] = *reinterpret_cast<const bf16x8*>(&sA[p * PPLANE(PNPIX) + PSLOT(apix[a] + ashift, kc * 2 + h)]);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int p = 0; p < NS; ++p)
          bfr[b][p] = *reinterpret_cast<const bf16x8*>(&sb[p * PPLANE(PBN) + PSLOT((wn * 2 + b) * 32 + li, kc * 2 + h)]);
#define PDAE_MMA(PA, PB)                                                                                          \
  _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)                    \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA], bfr[b][PB], acc[a][b], 0, 0, 0);
      if constexpr (NS == 3) { PDAE_MMA(1, 1) PDAE_MMA(0, 2) PDAE_MMA(2, 0) }
      if constexpr (NS >= 2) { PDAE_MMA(0, 1) PDAE_MMA(1, 0) }
      PDAE_MMA(0, 0)
#undef PDAE_MMA
    }
  };
  int buf = 0;
  // one (chunk, tap) step.  rl: register set that receives the panel of step s+2; rs: set holding the panel of step s+1
  auto step = [&](int s, float4 (&rl)[2 * BREP], const float4 (&rs)[2 * BREP]) {
    const int chunk = s / 9, tap = s - chunk * 9;
    if (s + 2 < total) { const int c2 = (s + 2) / 9; b_gload(rl, c2, s + 2 - c2 * 9); }
    if (tap == 0 && chunk + 1 < nchunk) a_gload((chunk + 1) << 5);          // next patch in flight during the 9 taps
    compute(tap, DBUF ? sB + buf * SB : sB);
    if constexpr (DBUF) {
      if (s + 1 < total) b_lstore(rs, sB + (buf ^ 1) * SB);                  // other buffer: nobody reads it during this step
      if (tap == 8 && chunk + 1 < nchunk) {
        __syncthreads();                                                     // every wave is done with the current patch
        a_lstore();
      }
      __syncthreads();
      buf ^= 1;
    } else {
      if (s + 1 < total) {
        __syncthreads();                                                     // every wave is done reading the panel (and the patch)
        b_lstore(rs, sB);
        if (tap == 8) a_lstore();
        __syncthreads();
      }
    }
  };
  a_gload(0);
  b_gload(r0, 0, 0);
  if (total > 1) b_gload(r1, 0, 1);
  a_lstore();
  b_lstore(r0, sB);
  __syncthreads();
  for (int s = 0; s < total; s += 2) {
    step(s, r0, r1);
    if (s + 1 < total) step(s + 1, r1, r0);
  }

  // ---- epilogue
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * h, g = wm * 2 + a;
      const int oy = y0 + (g >> 2) * 8 + (i >> 2), ox = x0 + (g & 3) * 4 + (i & 3);
      if (oy >= P.H || ox >= P.W) continue;
      const long long row = ((long long)img * P.H + oy) * P.W + ox;
      long long rrow = row;
      if (P.res_mode == 2) rrow = ((long long)img * (P.H >> 1) + (oy >> 1)) * (P.W >> 1) + (ox >> 1);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int col = n0 + (wn * 2 + b) * 32 + li;
        if (col >= P.Nout) continue;
        float val = acc[a][b][r];
        if (P.bias) val += P.bias[col];
        if (P.res_mode) val += P.res[rrow * P.Nout + col];
        float* dst = P.y + row * P.Nout + col;
        if (P.accumulate) val += *dst;
        *dst = val;
      }
    }
  }
}

template <int NS, int PTH> static int launch_ns(const PatchParams& P, hipStream_t s) {
  constexpr int NPIX = (PTH + 2) * PPW;
  const size_t smem = (size_t)(NS * PPLANE(NPIX) + (PTH == 16 ? 2 : 1) * NS * PPLANE(PBN)) * sizeof(unsigned short);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3p_kernel<NS, PTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { pdae_set_error("conv3x3p: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  dim3 grid(P.N * P.tiles_y * P.tiles_x * P.tiles_n);
  hipLaunchKernelGGL((conv3x3p_kernel<NS, PTH>), grid, dim3(PTH * 32), smem, s, P);
  return pdae_launch_status("conv3x3p");
}

static int patch_th() {                        // PDAE_P3_TH = 8 | 16 overrides the tile height (tuning aid)
  static int th = -1;
  if (th < 0) { const char* e = getenv("PDAE_P3_TH"); th = e ? atoi(e) : 0; }
  return th;
}

// eligibility: 3x3, stride 1, pad 1, one source, channels % 32, spatial tile-aligned and enough tiles to fill the chip
bool conv3x3p_ok(int math, int KH, int KW, int stride, int pad, int C1, int C, int H, int W, int N, int Nout) {
  if (math < 1 || KH != 3 || KW != 3 || stride != 1 || pad != 1 || C1 != 0) return false;
  if ((C & 31) || (H % 8) || (W % PTW) || (Nout & 3) || Nout < 32) return false;
  long long blocks = (long long)N * (H / 8) * (W / PTW) * ((Nout + PBN - 1) / PBN);
  return blocks >= 256;
}

int conv3x3p_launch(int math, const float* x, int N, int Hs, int Ws, int C, int H, int W, int up, const float* w, int Nout,
                    float* y, const float* bias, const float* res, int res_mode, int accumulate, hipStream_t s) {
  PatchParams P;
  P.x = x; P.N = N; P.Hs = Hs; P.Ws = Ws; P.C = C; P.H = H; P.W = W; P.up = up; P.w = w; P.Nout = Nout;
  P.y = y; P.bias = bias; P.res = res; P.res_mode = res_mode; P.accumulate = accumulate;
  int th = patch_th();
  if (th != 8 && th != 16) th = 16;
  if (H % th) th = 8;
  P.tiles_x = W / PTW; P.tiles_y = H / th; P.tiles_n = (Nout + PBN - 1) / PBN;
#define PDAE_P3(NS_) (th == 16 ? launch_ns<NS_, 16>(P, s) : launch_ns<NS_, 8>(P, s))
  if (math == 1) return PDAE_P3(1);
  if (math == 2) return PDAE_P3(2);
  return PDAE_P3(3);
#undef PDAE_P3
}
