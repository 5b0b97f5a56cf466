// The 3-channel edges of the U-Nets on the matrix cores: the image heads `out.2` / `shift_out.2` (unet.py:174, shift_unet.py:248: 128 -> 3
// channels), the stem `input_blocks.0` (unet.py:62: 3 -> 128) and the heads' data gradient (3 -> 128).  All three are HBM-bound layers --
// one pass over a 128-channel tensor (268 MB at 128^2, B=32: ~50 us) next to a few GFLOP -- that the fp32 FMA kernels of convhead.hip and the
// generic implicit GEMM ran 3-4x above that time (185 / 182 / 153 us; VALU floor of the FMA form alone: 55 us).
//
//   head forward (Cout <= 3):  y[q][co] = b[co] + sum_tap Z[q + d(tap)][tap, co],   Z[p][tap, co] = sum_c x[p][c] w[co][tap][c]
//     i.e. a 1x1 convolution to 9 Cout <= 27 virtual channels on the MFMAs (K = C: NINE times fewer products than the im2col form, whose
//     K is 9 C) followed by a 9-term shifted sum out of LDS.  One workgroup = 16 x 16 output pixels: Z of the 18 x 18 halo pixels as 11
//     row tiles of 32 pixels; the A fragments (pixel x 16 channels) come straight from global memory in fragment order (two 16-byte loads
//     per lane and k-step, next tile prefetched into registers under the MFMAs), the weights once per workgroup as bf16 planes in LDS.
//     Arithmetic: 3 bf16 planes x 6 products, fp32 accumulate (the range-free fp32-grade format of the 1x1 kernels).
#include "common.h"
#include "kernels.h"
#include "igemm.h"
#include "conv3x3p.h"

#define ET 16                    // 16 x 16 output pixels per workgroup
#define EH (ET + 2)              // halo edge
#define ENH (EH * EH)            // 324 halo pixels
#define ETILES ((ENH + 31) / 32) // 11 row tiles of 32 pixels
#define EZP (ETILES * 32 + 1)    // floats per Z row (odd pitch: the 32 virtual channels of an accumulator column hit 32 different banks)

struct EdgeParams {
  const float* x; int N, H, W, C;
  const float* w;               // [COUT][9][C]
  const float* bias; float* y;
  int tiles_x, tiles_y;
};

typedef unsigned e_u32x4 __attribute__((ext_vector_type(4)));

// position in the tile list of workgroup b of n: XCD b % 8 owns the contiguous range [x * n / 8, (x + 1) * n / 8) (n % 8 != 0: plain order)
__device__ __forceinline__ int edge_xcd_tile(int b, int n) { return (n & 7) ? b : (b & 7) * (n >> 3) + (b >> 3); }

// KS = C / 16 k-steps
template <int COUT, int KS>
__global__ void __launch_bounds__(256, 2) edge_head_fwd_kernel(const EdgeParams P) {
  __shared__ __attribute__((aligned(16))) e_u32x4 sB[KS * 3 * 64];       // weight fragments [k-step][plane][lane]
  __shared__ float sZ[9 * COUT * EZP];
  const int t = threadIdx.x, l = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), ln = l & 31, kh = l >> 5;
  // workgroups are dealt round-robin to the 8 XCDs, each with its own L2: XCD k takes the k-th eighth of the tile list in order, so that the
  // workgroups running side by side on one L2 are neighbouring tiles and share their halo rows / columns there (27 % of the reads)
  int b = edge_xcd_tile(blockIdx.x, gridDim.x);
  const int bx = b % P.tiles_x; b /= P.tiles_x;
  const int by = b % P.tiles_y; const int img = b / P.tiles_y;
  const int y0 = by * ET, x0 = bx * ET, C = P.C;

  // weights -> bf16 planes in B-fragment order: lane (n = virtual channel tap * COUT + co, 8 consecutive input channels)
  for (int ks = wv; ks < KS; ks += 4) {
    float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0;
    if (ln < 9 * COUT) {
      const int tap = ln / COUT, co = ln - tap * COUT;
      const float* s = P.w + ((size_t)co * 9 + tap) * C + ks * 16 + kh * 8;
      u0 = *reinterpret_cast<const float4*>(s); u1 = *reinterpret_cast<const float4*>(s + 4);
    }
    unsigned w0[3], w1[3], w2[3], w3[3];
    p_split2<3>(u0.x, u0.y, w0); p_split2<3>(u0.z, u0.w, w1); p_split2<3>(u1.x, u1.y, w2); p_split2<3>(u1.z, u1.w, w3);
#pragma unroll
    for (int p = 0; p < 3; ++p) { e_u32x4 v = {w0[p], w1[p], w2[p], w3[p]}; sB[(ks * 3 + p) * 64 + l] = v; }
  }

  float4 raw[2][KS * 2];
  auto issue = [&](int tile, float4 (&r)[KS * 2]) {
    const int h = tile * 32 + ln, hy = h / EH, hx = h - hy * EH, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = h < ENH && (unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W;
    const float* s = P.x + (((long long)img * P.H + gy) * P.W + gx) * C + kh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      r[ks * 2] = make_float4(0.f, 0.f, 0.f, 0.f); r[ks * 2 + 1] = r[ks * 2];
      if (ok) { r[ks * 2] = *reinterpret_cast<const float4*>(s + ks * 16); r[ks * 2 + 1] = *reinterpret_cast<const float4*>(s + ks * 16 + 4); }
    }
  };
  issue(wv, raw[0]);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int tile = wv + 4 * it;
    if (tile >= ETILES) break;
    if (it < 2 && tile + 4 < ETILES) issue(tile + 4, raw[(it + 1) & 1]);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 u0 = raw[it & 1][ks * 2], u1 = raw[it & 1][ks * 2 + 1];
      unsigned w0[3], w1[3], w2[3], w3[3];
      p_split2<3>(u0.x, u0.y, w0); p_split2<3>(u0.z, u0.w, w1); p_split2<3>(u1.x, u1.y, w2); p_split2<3>(u1.z, u1.w, w3);
      bf16x8 A[3], B[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const e_u32x4 a = {w0[p], w1[p], w2[p], w3[p]};
        A[p] = __builtin_bit_cast(bf16x8, a);
        B[p] = __builtin_bit_cast(bf16x8, sB[(ks * 3 + p) * 64 + l]);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], acc, 0, 0, 0);           // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], acc, 0, 0, 0);
    }
    if (ln < 9 * COUT) {                         // accumulator column = virtual channel, rows = halo pixels
#pragma unroll
      for (int r = 0; r < 16; ++r) sZ[ln * EZP + tile * 32 + (r >> 2) * 8 + kh * 4 + (r & 3)] = acc[r];
    }
  }
  __syncthreads();
  const int ty = t >> 4, tx = t & 15, oy = y0 + ty, ox = x0 + tx;
  if (oy < P.H && ox < P.W) {
    float* dst = P.y + (((long long)img * P.H + oy) * P.W + ox) * COUT;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float s = P.bias ? P.bias[co] : 0.f;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) s += sZ[(tap * COUT + co) * EZP + (ty + tap / 3) * EH + tx + tap % 3];
      dst[co] = s;
    }
  }
}

bool edge_head_ok(int KH, int KW, int stride, int pad, int up, int C1, int C, int Cout) {
  return KH == 3 && KW == 3 && stride == 1 && pad == 1 && !up && C1 == 0 && (C == 32 || C == 64 || C == 128) && Cout >= 1 && Cout <= 3;
}

template <int COUT> static void edge_head_launch(const EdgeParams& P, dim3 grid, hipStream_t s) {
  switch (P.C) {
    case 32: hipLaunchKernelGGL((edge_head_fwd_kernel<COUT, 2>), grid, dim3(256), 0, s, P); break;
    case 64: hipLaunchKernelGGL((edge_head_fwd_kernel<COUT, 4>), grid, dim3(256), 0, s, P); break;
    default: hipLaunchKernelGGL((edge_head_fwd_kernel<COUT, 8>), grid, dim3(256), 0, s, P); break;
  }
}

int edge_head_fwd(const float* x, int N, int H, int W, int C, const float* w, int Cout, const float* bias, float* y, hipStream_t s) {
  EdgeParams P;
  P.x = x; P.N = N; P.H = H; P.W = W; P.C = C; P.w = w; P.bias = bias; P.y = y;
  P.tiles_x = (W + ET - 1) / ET; P.tiles_y = (H + ET - 1) / ET;
  dim3 grid(N * P.tiles_x * P.tiles_y);
  switch (Cout) {
    case 1: edge_head_launch<1>(P, grid, s); break;
    case 2: edge_head_launch<2>(P, grid, s); break;
    default: edge_head_launch<3>(P, grid, s); break;
  }
  return pdae_launch_status("edge_head_fwd");
}

// ----------------------------------------------------------------------------------------------
//   3 -> Nout channels (stem forward; data gradient of the heads): im2col GEMM with K = 9 Cin <= 27 padded to 32 = two k-steps.  One
//     workgroup = 16 x 16 output pixels x NT * 32 channels: the 18 x 18 x Cin input patch in LDS as fp32 (4 KB), A fragments gathered from
//     it (lane = pixel, 8 of the 32 (tap, channel) slots), weights as bf16 planes in LDS in B-fragment order, accumulators stored straight
//     from the MFMA layout: 32 lanes = 32 consecutive channels of one pixel = one 128-byte line per half wave.  HBM-bound on the output
//     write (268 MB at 128^2 x 128, B=32).
//     transposed = 1: w is the FORWARD weight [Cin][9][Nout] of the convolution whose data gradient this is (taps flipped).
// ----------------------------------------------------------------------------------------------
struct EdgeInParams {
  const float* x; int N, H, W;
  const float* w; int transposed;
  const float* bias; float* y; int Nout, accumulate;
  int tiles_x, tiles_y;
};

template <int CIN, int NT>
__global__ void __launch_bounds__(256, 2) edge_in_kernel(const EdgeInParams P) {
  __shared__ __attribute__((aligned(16))) e_u32x4 sB[NT * 2 * 3 * 64];     // [channel tile][k-step][plane][lane]
  __shared__ float patch[2][ENH * CIN];
  constexpr int K = 9 * CIN, NPT = (ENH * CIN + 255) / 256;                // patch floats per thread
  const int t = threadIdx.x, l = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), ln = l & 31, kh = l >> 5;
  const int n0 = blockIdx.y * NT * 32, Nout = P.Nout, ntiles = P.N * P.tiles_x * P.tiles_y;

  // persistent workgroups (two per CU): the weights are prepared once, the patch of tile i+1 is fetched into registers under the MFMAs and
  // stores of tile i, so that after the first tile nothing waits for HBM but the store queue
  for (int item = wv; item < NT * 2; item += 4) {
    const int nt = item >> 1, s = item & 1, n = n0 + nt * 32 + ln;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = s * 16 + kh * 8 + j, tap = k / CIN, c = k - tap * CIN;
      v[j] = 0.f;
      if (k < K) v[j] = P.transposed ? P.w[((size_t)c * 9 + (8 - tap)) * Nout + n] : P.w[((size_t)n * 9 + tap) * CIN + c];
    }
    unsigned w0[3], w1[3], w2[3], w3[3];
    p_split2<3>(v[0], v[1], w0); p_split2<3>(v[2], v[3], w1); p_split2<3>(v[4], v[5], w2); p_split2<3>(v[6], v[7], w3);
#pragma unroll
    for (int p = 0; p < 3; ++p) { e_u32x4 q = {w0[p], w1[p], w2[p], w3[p]}; sB[((nt * 2 + s) * 3 + p) * 64 + l] = q; }
  }
  float pre[NPT];
  auto fetch = [&](int tile) {
    const int bx = tile % P.tiles_x, t2 = tile / P.tiles_x, by = t2 % P.tiles_y, img = t2 / P.tiles_y;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int idx = t + i * 256, pix = idx / CIN, c = idx - pix * CIN, py = pix / EH, px = pix - py * EH;
      const int gy = by * ET - 1 + py, gx = bx * ET - 1 + px;
      pre[i] = 0.f;
      if (idx < ENH * CIN && (unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W)
        pre[i] = P.x[(((long long)img * P.H + gy) * P.W + gx) * CIN + c];
    }
  };
  auto commit = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) if (t + i * 256 < ENH * CIN) dst[t + i * 256] = pre[i];
  };
  int tile = blockIdx.x, cur = 0;
  if (tile < ntiles) { fetch(tile); commit(patch[0]); }
  __syncthreads();
#pragma unroll 1
  for (; tile < ntiles; tile += gridDim.x) {
    const bool more = tile + (int)gridDim.x < ntiles;
    if (more) fetch(tile + gridDim.x);
    const int bx = tile % P.tiles_x, t2 = tile / P.tiles_x, by = t2 % P.tiles_y, img = t2 / P.tiles_y;
    const int y0 = by * ET, x0 = bx * ET;
    const float* pt = patch[cur];
#pragma unroll 1
    for (int rt = 0; rt < 2; ++rt) {
      const int rtile = wv * 2 + rt;                       // row tile = two image rows of 16 pixels
      const int base = ((rtile * 2 + (ln >> 4)) * EH + (ln & 15)) * CIN;
      bf16x8 A[2][3];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = s * 16 + kh * 8 + j, tap = k / CIN, c = k - tap * CIN;
          v[j] = k < K ? pt[base + ((tap / 3) * EH + tap % 3) * CIN + c] : 0.f;
        }
        unsigned w0[3], w1[3], w2[3], w3[3];
        p_split2<3>(v[0], v[1], w0); p_split2<3>(v[2], v[3], w1); p_split2<3>(v[4], v[5], w2); p_split2<3>(v[6], v[7], w3);
#pragma unroll
        for (int p = 0; p < 3; ++p) { const e_u32x4 q = {w0[p], w1[p], w2[p], w3[p]}; A[s][p] = __builtin_bit_cast(bf16x8, q); }
      }
      // channel tiles in groups of two: accumulate, then store (four live accumulator tiles next to the prefetched patch spilled two VGPRs
      // in the <3, 4> instantiation; the sums are the same)
      constexpr int NTG = NT < 2 ? NT : 2;
#pragma unroll 1
      for (int ng = 0; ng < NT; ng += NTG) {
      f32x16 acc[NTG];
#pragma unroll
      for (int ni = 0; ni < NTG; ++ni) {
        const int nt = ng + ni;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 B[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) B[p] = __builtin_bit_cast(bf16x8, sB[((nt * 2 + s) * 3 + p) * 64 + l]);
          acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s][1], B[1], acc[ni], 0, 0, 0);
          acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s][0], B[2], acc[ni], 0, 0, 0);
          acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s][2], B[0], acc[ni], 0, 0, 0);
          acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s][0], B[1], acc[ni], 0, 0, 0);
          acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s][1], B[0], acc[ni], 0, 0, 0);
          acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s][0], B[0], acc[ni], 0, 0, 0);
        }
      }
      // accumulator row r of this lane = pixel (rtile * 2 + row / 16, row % 16), column = channel: 32 lanes = one 128-byte line.
      // Addresses = uniform base (scalar arithmetic) + one 32-bit lane offset.
      char* const ub = reinterpret_cast<char*>(P.y + (((long long)img * P.H + y0 + rtile * 2) * P.W + x0) * Nout + n0);
      const unsigned lane_off = (unsigned)((kh * 4 * Nout + ln) * 4);
      const bool full = y0 + ET <= P.H && x0 + ET <= P.W;
#pragma unroll
      for (int ni = 0; ni < NTG; ++ni) {
        const int nt = ng + ni;
        const float bv = P.bias ? P.bias[n0 + nt * 32 + ln] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dyy = r >> 3, dxu = 8 * ((r >> 2) & 1) + (r & 3);                 // pixel column = dxu + kh * 4
          float* dst = reinterpret_cast<float*>(ub + ((size_t)(dyy * P.W + dxu) * Nout + nt * 32) * 4 + lane_off);
          if (full || (y0 + rtile * 2 + dyy < P.H && x0 + dxu + kh * 4 < P.W)) {
            float v = acc[ni][r] + bv;
            if (P.accumulate) v += *dst;
            *dst = v;
          }
        }
      }
      }
    }
    if (more) commit(patch[cur ^ 1]);
    __syncthreads();
    cur ^= 1;
  }
}

bool edge_in_ok(int KH, int KW, int stride, int pad, int up, int C1, int Cin, int Nout) {
  return KH == 3 && KW == 3 && stride == 1 && pad == 1 && !up && C1 == 0 && Cin >= 1 && Cin <= 3 && Nout >= 32 && (Nout & 31) == 0;
}

template <int CIN> static void edge_in_launch(const EdgeInParams& P, hipStream_t s) {
  const int nt = (P.Nout % 128) == 0 ? 4 : (P.Nout % 64) == 0 ? 2 : 1;
  const int ntiles = P.N * P.tiles_x * P.tiles_y;
  dim3 grid(ntiles < 512 ? ntiles : 512, P.Nout / (32 * nt));              // two persistent workgroups per CU
  if (nt == 4) hipLaunchKernelGGL((edge_in_kernel<CIN, 4>), grid, dim3(256), 0, s, P);
  else if (nt == 2) hipLaunchKernelGGL((edge_in_kernel<CIN, 2>), grid, dim3(256), 0, s, P);
  else hipLaunchKernelGGL((edge_in_kernel<CIN, 1>), grid, dim3(256), 0, s, P);
}

int edge_in_conv(const float* x, int N, int H, int W, int Cin, const float* w, int transposed, int Nout, const float* bias, float* y, int accumulate,
                 hipStream_t s) {
  EdgeInParams P;
  P.x = x; P.N = N; P.H = H; P.W = W; P.w = w; P.transposed = transposed; P.bias = bias; P.y = y; P.Nout = Nout; P.accumulate = accumulate;
  P.tiles_x = (W + ET - 1) / ET; P.tiles_y = (H + ET - 1) / ET;
  switch (Cin) {
    case 1: edge_in_launch<1>(P, s); break;
    case 2: edge_in_launch<2>(P, s); break;
    default: edge_in_launch<3>(P, s); break;
  }
  return pdae_launch_status("edge_in_conv");
}

// ----------------------------------------------------------------------------------------------
//   head weight gradient (Cout <= 3):  dW[co][tap][c] = sum_p x[p][c] dY[p - d(tap)][co]  as a GEMM with M = 9 Cout <= 27 virtual rows
//     (tap, co), N = C, K = pixels: A fragments (row = (tap, co), 8 consecutive pixels of one tile row) gathered from an 18 x 18 x Cout
//     patch of dY in LDS, B fragments (channel, 8 consecutive pixels) straight from global memory -- for each of the 8 pixels the 32 lanes of
//     a half wave read 32 consecutive channels, one 128-byte line -- through a ring of four k-steps that runs across tile boundaries (eight: register spills, 105 -> 144 us).
//     Persistent workgroups: every wave keeps ONE accumulator tile (27 x 32 channels) over all its pixel tiles, so x is read exactly once
//     and the only output is one partial dW per wave group, summed in fixed order by the column-sum kernel.
//     NTC = C / 32 channel tiles; the 4 waves = NTC channel tiles x KSPLIT = 4 / NTC interleaved subsets of the 16 tile rows.
// ----------------------------------------------------------------------------------------------
struct EdgeWgParams {
  const float* x; const float* dy; float* part;
  int N, H, W, C, tiles_x, tiles_y;
};

template <int COUT, int NTC>
__global__ void __launch_bounds__(256, 2) edge_head_wgrad_kernel(const EdgeWgParams P) {
  constexpr int KSPLIT = 4 / NTC, SPT = 16 / KSPLIT, D = 4, NPT = (ENH * COUT + 255) / 256;
  __shared__ float dpatch[2][ENH * COUT];
  const int t = threadIdx.x, l = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), ln = l & 31, kh = l >> 5;
  const int nt = wv % NTC, ksub = wv / NTC, C = P.C, ntiles = P.N * P.tiles_x * P.tiles_y, G = gridDim.x;
  const bool mrow = ln < 9 * COUT;
  const int mtap = mrow ? ln / COUT : 0, mco = mrow ? ln - mtap * COUT : 0;
  const int abase = ((2 - mtap / 3) * EH + (kh * 8 + 2 - mtap % 3)) * COUT + mco;

  float pre[NPT];
  auto fetch = [&](int tile) {
    const int bx = tile % P.tiles_x, t2 = tile / P.tiles_x, by = t2 % P.tiles_y, img = t2 / P.tiles_y;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int idx = t + i * 256, pix = idx / COUT, c = idx - pix * COUT, py = pix / EH, px = pix - py * EH;
      const int gy = by * ET - 1 + py, gx = bx * ET - 1 + px;
      pre[i] = 0.f;
      if (idx < ENH * COUT && (unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W)
        pre[i] = P.dy[(((long long)img * P.H + gy) * P.W + gx) * COUT + c];
    }
  };
  auto commit = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) if (t + i * 256 < ENH * COUT) dst[t + i * 256] = pre[i];
  };
  float bq[D][8];
  auto loadB = [&](int tile, int s, float (&q)[8]) {            // tile row s: 8 pixels (kh * 8 + j) x this lane's channel
    const int bx = tile % P.tiles_x, t2 = tile / P.tiles_x, by = t2 % P.tiles_y, img = t2 / P.tiles_y;
    const int gy = by * ET + s, gx = bx * ET + kh * 8;
    const float* src = P.x + (((long long)img * P.H + gy) * P.W + gx) * C + nt * 32 + ln;
#pragma unroll
    for (int j = 0; j < 8; ++j) {                                // branch-free: pixels outside the image read x[0] and are zeroed
      const bool ok = gy < P.H && gx + j < P.W;
      const float v = *(ok ? src + (size_t)j * C : P.x);
      q[j] = ok ? v : 0.f;
    }
  };
  int tile = blockIdx.x, cur = 0;
  fetch(tile); commit(dpatch[0]);
#pragma unroll
  for (int i = 0; i < D; ++i) loadB(tile, ksub + KSPLIT * i, bq[i]);
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 1
  for (; tile < ntiles; tile += G) {
    const bool more = tile + G < ntiles;
    if (more) fetch(tile + G);
    const float* dp = dpatch[cur] + abase;
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
      const int s = ksub + KSPLIT * i;
      float av[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = dp[(s * EH + j) * COUT];
      unsigned a0[3], a1[3], a2[3], a3[3], b0[3], b1[3], b2[3], b3[3];
      if (!mrow) {
#pragma unroll
        for (int j = 0; j < 8; ++j) av[j] = 0.f;
      }
      p_split2<3>(av[0], av[1], a0); p_split2<3>(av[2], av[3], a1); p_split2<3>(av[4], av[5], a2); p_split2<3>(av[6], av[7], a3);
      float (&q)[8] = bq[i % D];
      p_split2<3>(q[0], q[1], b0); p_split2<3>(q[2], q[3], b1); p_split2<3>(q[4], q[5], b2); p_split2<3>(q[6], q[7], b3);
      bf16x8 A[3], B[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const e_u32x4 a = {a0[p], a1[p], a2[p], a3[p]}, bb = {b0[p], b1[p], b2[p], b3[p]};
        A[p] = __builtin_bit_cast(bf16x8, a); B[p] = __builtin_bit_cast(bf16x8, bb);
      }
      if (i + D < SPT) loadB(tile, ksub + KSPLIT * (i + D), q);
      else if (more) loadB(tile + G, ksub + KSPLIT * (i + D - SPT), q);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);                         // keep the ring: the scheduler would hoist every load of the tile to its top
    }
    if (more) commit(dpatch[cur ^ 1]);
    __syncthreads();
    cur ^= 1;
  }
  float* outp = P.part + ((size_t)blockIdx.x * KSPLIT + ksub) * (COUT * 9 * C) + nt * 32 + ln;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = (r >> 2) * 8 + kh * 4 + (r & 3);
    if (m < 9 * COUT) { const int tap = m / COUT, co = m - tap * COUT; outp[(size_t)(co * 9 + tap) * C] = acc[r]; }
  }
}

bool edge_head_wgrad_ok(int KH, int KW, int stride, int pad, int up, int C1, int C, int Cout) { return edge_head_ok(KH, KW, stride, pad, up, C1, C, Cout); }

template <int COUT> static void edge_wg_launch(const EdgeWgParams& P, int nblocks, hipStream_t s) {
  switch (P.C) {
    case 32: hipLaunchKernelGGL((edge_head_wgrad_kernel<COUT, 1>), dim3(nblocks), dim3(256), 0, s, P); break;
    case 64: hipLaunchKernelGGL((edge_head_wgrad_kernel<COUT, 2>), dim3(nblocks), dim3(256), 0, s, P); break;
    default: hipLaunchKernelGGL((edge_head_wgrad_kernel<COUT, 4>), dim3(nblocks), dim3(256), 0, s, P); break;
  }
}

// workspace: convhead_wgrad_workspace_bytes (at most as many partial rows as that kernel's one-per-tile)
int edge_head_wgrad(const float* x, int N, int H, int W, int C, const float* dy, int Cout, float* dw, int accumulate, float* ws, size_t ws_bytes,
                    hipStream_t s) {
  EdgeWgParams P;
  P.x = x; P.dy = dy; P.part = ws; P.N = N; P.H = H; P.W = W; P.C = C;
  P.tiles_x = (W + ET - 1) / ET; P.tiles_y = (H + ET - 1) / ET;
  const int ntiles = N * P.tiles_x * P.tiles_y, ksplit = 4 / (C / 32);
  int nblocks = ntiles / ksplit; if (nblocks < 1) nblocks = 1; if (nblocks > 512) nblocks = 512;      // two persistent workgroups per CU
  const long long rows = (long long)nblocks * ksplit;
  if (!ws || ws_bytes < convhead_wgrad_workspace_bytes(N, H, W, C, Cout)) { pdae_set_error("edge_head_wgrad: workspace too small"); return PDAE_EINVAL; }
  switch (Cout) {
    case 1: edge_wg_launch<1>(P, nblocks, s); break;
    case 2: edge_wg_launch<2>(P, nblocks, s); break;
    default: edge_wg_launch<3>(P, nblocks, s); break;
  }
  if (int e = pdae_launch_status("edge_head_wgrad")) return e;
  return k_colsum(ws, rows, Cout * 9 * C, dw, accumulate, ws + (size_t)rows * Cout * 9 * C, s);
}
