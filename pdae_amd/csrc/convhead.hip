// 3x3 / stride-1 / pad-1 convolution with a handful of output channels (Cout <= 4): the image heads `out.2` / `shift_out.2`
// (unet.py:174, shift_unet.py:248: 128 -> 3 channels).  On a 128-wide MFMA tile 97% of the work would be padding; these are
// HBM-bound layers (read the 128-channel activation once), so they run as exact fp32 FMA kernels out of an LDS halo patch:
//   forward : thread = output pixel, COUT accumulators, weights through the scalar cache (uniform addresses);
//   wgrad   : thread = (tap, 4 input channels, pixel third), 12 accumulators, dY read as LDS broadcasts; per-block partials are
//             summed in fixed order by the column-sum kernel.
#include "common.h"
#include "kernels.h"
#include "igemm.h"

#define HT 16                    // 16 x 16 pixel tile
#define HP (HT + 2)              // patch edge
#define HPITCH 36                // floats per patch pixel: 32 channels + 4 pad (144 bytes: conflict-free ds_read_b128 across pixels)

struct HeadParams {
  const float* x; int N, H, W, C;
  const float* w;               // [COUT][9][C]
  const float* bias; float* y;  // forward
  const float* dy; float* part; // wgrad: per-block partial dW [blocks][COUT*9*C]
  int COUT, tiles_x, tiles_y;
};

__device__ __forceinline__ void head_stage_patch(const HeadParams& P, float* patch, int img, int y0, int x0, int c0, int t) {
  for (int idx = t; idx < HP * HP * 8; idx += 256) {
    const int pix = idx >> 3, q = idx & 7;
    const int py = pix / HP, px = pix - py * HP;
    const int ly = y0 - 1 + py, lx = x0 - 1 + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)ly < (unsigned)P.H && (unsigned)lx < (unsigned)P.W)
      v = *reinterpret_cast<const float4*>(P.x + (((long long)img * P.H + ly) * P.W + lx) * P.C + c0 + q * 4);
    *reinterpret_cast<float4*>(&patch[pix * HPITCH + q * 4]) = v;
  }
}

template <int COUT>
__global__ void __launch_bounds__(256) convhead_fwd_kernel(const HeadParams P) {
  __shared__ __attribute__((aligned(16))) float patch[HP * HP * HPITCH];
  const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
  int b = blockIdx.x;
  const int bx = b % P.tiles_x; b /= P.tiles_x;
  const int by = b % P.tiles_y; const int img = b / P.tiles_y;
  const int y0 = by * HT, x0 = bx * HT;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = P.bias ? P.bias[co] : 0.f;
  for (int c0 = 0; c0 < P.C; c0 += 32) {
    __syncthreads();
    head_stage_patch(P, patch, img, y0, x0, c0, t);
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* src = &patch[((ty + tap / 3) * HP + tx + tap % 3) * HPITCH];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(src + q * 4);
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          const float* wr = P.w + ((size_t)co * 9 + tap) * P.C + c0 + q * 4;      // uniform address: scalar loads
          acc[co] = fmaf(v.x, wr[0], acc[co]); acc[co] = fmaf(v.y, wr[1], acc[co]);
          acc[co] = fmaf(v.z, wr[2], acc[co]); acc[co] = fmaf(v.w, wr[3], acc[co]);
        }
      }
    }
  }
  const int oy = y0 + ty, ox = x0 + tx;
  if (oy < P.H && ox < P.W) {
    float* dst = P.y + (((long long)img * P.H + oy) * P.W + ox) * COUT;
#pragma unroll
    for (int co = 0; co < COUT; ++co) dst[co] = acc[co];
  }
}

// wgrad: thread (item = tap * 8 + quad, pg = pixel third); per 32-channel chunk 72 items x 3 pixel groups = 216 active threads
template <int COUT>
__global__ void __launch_bounds__(256) convhead_wgrad_kernel(const HeadParams P) {
  __shared__ __attribute__((aligned(16))) float patch[HP * HP * HPITCH];
  __shared__ float sdy[HT * HT * 4];
  __shared__ float red[216 * COUT * 4];
  const int t = threadIdx.x;
  int b = blockIdx.x;
  const int bx = b % P.tiles_x; b /= P.tiles_x;
  const int by = b % P.tiles_y; const int img = b / P.tiles_y;
  const int y0 = by * HT, x0 = bx * HT;
  {                                               // dY tile (zero outside the image)
    const int py = t >> 4, px = t & 15, oy = y0 + py, ox = x0 + px;
    const bool ok = oy < P.H && ox < P.W;
    const float* s = P.dy + (((long long)img * P.H + oy) * P.W + ox) * COUT;
#pragma unroll
    for (int co = 0; co < 4; ++co) sdy[t * 4 + co] = (ok && co < COUT) ? s[co] : 0.f;
  }
  const int item = t % 72, pg = t / 72;           // pg = 3: idle
  const int tap = item >> 3, quad = item & 7;
  const int dyo = tap / 3, dxo = tap % 3;
  const int p_begin = pg * 86, p_end = pg == 2 ? 256 : p_begin + 86;
  float* outp = P.part + (size_t)blockIdx.x * COUT * 9 * P.C;
  for (int c0 = 0; c0 < P.C; c0 += 32) {
    __syncthreads();
    head_stage_patch(P, patch, img, y0, x0, c0, t);
    __syncthreads();
    float acc[COUT][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co][0] = acc[co][1] = acc[co][2] = acc[co][3] = 0.f;
    if (pg < 3) {
      for (int p = p_begin; p < p_end; ++p) {
        const int py = p >> 4, px = p & 15;
        const float4 v = *reinterpret_cast<const float4*>(&patch[((py + dyo) * HP + px + dxo) * HPITCH + quad * 4]);
        const float4 g = *reinterpret_cast<const float4*>(&sdy[p * 4]);
        const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          acc[co][0] = fmaf(gg[co], v.x, acc[co][0]); acc[co][1] = fmaf(gg[co], v.y, acc[co][1]);
          acc[co][2] = fmaf(gg[co], v.z, acc[co][2]); acc[co][3] = fmaf(gg[co], v.w, acc[co][3]);
        }
      }
#pragma unroll
      for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(t * COUT + co) * 4 + j] = acc[co][j];
    }
    __syncthreads();
    if (t < 72) {
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        float4 s;
        s.x = red[(t * COUT + co) * 4 + 0] + red[((t + 72) * COUT + co) * 4 + 0] + red[((t + 144) * COUT + co) * 4 + 0];
        s.y = red[(t * COUT + co) * 4 + 1] + red[((t + 72) * COUT + co) * 4 + 1] + red[((t + 144) * COUT + co) * 4 + 1];
        s.z = red[(t * COUT + co) * 4 + 2] + red[((t + 72) * COUT + co) * 4 + 2] + red[((t + 144) * COUT + co) * 4 + 2];
        s.w = red[(t * COUT + co) * 4 + 3] + red[((t + 72) * COUT + co) * 4 + 3] + red[((t + 144) * COUT + co) * 4 + 3];
        *reinterpret_cast<float4*>(outp + ((size_t)co * 9 + tap) * P.C + c0 + quad * 4) = s;
      }
    }
  }
}

bool convhead_ok(int KH, int KW, int stride, int pad, int up, int C1, int C, int Cout) {
  return KH == 3 && KW == 3 && stride == 1 && pad == 1 && !up && C1 == 0 && (C & 31) == 0 && Cout >= 1 && Cout <= 4;
}

static void head_params(HeadParams& P, const float* x, int N, int H, int W, int C, int Cout) {
  P.x = x; P.N = N; P.H = H; P.W = W; P.C = C; P.COUT = Cout;
  P.tiles_x = (W + HT - 1) / HT; P.tiles_y = (H + HT - 1) / HT;
}

int convhead_fwd(const float* x, int N, int H, int W, int C, const float* w, int Cout, const float* bias, float* y, hipStream_t s) {
  HeadParams P; head_params(P, x, N, H, W, C, Cout);
  P.w = w; P.bias = bias; P.y = y; P.dy = nullptr; P.part = nullptr;
  dim3 grid(N * P.tiles_x * P.tiles_y);
  switch (Cout) {
    case 1: hipLaunchKernelGGL(convhead_fwd_kernel<1>, grid, dim3(256), 0, s, P); break;
    case 2: hipLaunchKernelGGL(convhead_fwd_kernel<2>, grid, dim3(256), 0, s, P); break;
    case 3: hipLaunchKernelGGL(convhead_fwd_kernel<3>, grid, dim3(256), 0, s, P); break;
    default: hipLaunchKernelGGL(convhead_fwd_kernel<4>, grid, dim3(256), 0, s, P); break;
  }
  return pdae_launch_status("convhead_fwd");
}

// workspace: per-block partials [blocks][Cout*9*C] + the column-sum scratch
size_t convhead_wgrad_workspace_bytes(int N, int H, int W, int C, int Cout) {
  long long blocks = (long long)N * ((W + HT - 1) / HT) * ((H + HT - 1) / HT);
  if (blocks < 4) blocks = 4;                      // convedge.hip: up to four partial rows (one per wave) from a single tile
  return ((size_t)blocks * Cout * 9 * C + k_colsum_workspace_floats(blocks, Cout * 9 * C)) * sizeof(float);
}

int convhead_wgrad(const float* x, int N, int H, int W, int C, const float* dy, int Cout, float* dw, int accumulate, float* ws, size_t ws_bytes,
                   hipStream_t s) {
  HeadParams P; head_params(P, x, N, H, W, C, Cout);
  const long long blocks = (long long)N * P.tiles_x * P.tiles_y;
  if (!ws || ws_bytes < convhead_wgrad_workspace_bytes(N, H, W, C, Cout)) { pdae_set_error("convhead_wgrad: workspace too small"); return PDAE_EINVAL; }
  P.w = nullptr; P.bias = nullptr; P.y = nullptr; P.dy = dy; P.part = ws;
  dim3 grid((int)blocks);
  switch (Cout) {
    case 1: hipLaunchKernelGGL(convhead_wgrad_kernel<1>, grid, dim3(256), 0, s, P); break;
    case 2: hipLaunchKernelGGL(convhead_wgrad_kernel<2>, grid, dim3(256), 0, s, P); break;
    case 3: hipLaunchKernelGGL(convhead_wgrad_kernel<3>, grid, dim3(256), 0, s, P); break;
    default: hipLaunchKernelGGL(convhead_wgrad_kernel<4>, grid, dim3(256), 0, s, P); break;
  }
  if (int e = pdae_launch_status("convhead_wgrad")) return e;
  return k_colsum(ws, blocks, Cout * 9 * C, dw, accumulate, ws + (size_t)blocks * Cout * 9 * C, s);
}
