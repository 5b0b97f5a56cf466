// Skinny linear layer  C[m][n] (+)= sum_k A[m][k] * B[n][k] + bias[n]   for M <= 32 rows (the batch):
// timestep / semantic embedding MLPs, per-ResBlock emb_layers (model/module.py:258,337,341, unet.py:52-54, shift_unet.py:63).
//
// These are ~100 launches per step with M = batch = 32: a 64x64 MFMA tile walks K alone in one block (20 us of pure latency).
// Here ONE WAVE owns one output feature n: lane l holds B[n][8l..8l+7] (+512 per pass) and accumulates the 32 partial dot products
// against the L1/L2-resident A rows; the 32 sums are then reduced over the 64 lanes with a butterfly that HALVES the number of
// live values per stage (16+8+4+2+1+1 = 32 DPP/permute adds instead of 32 x 6): exact fp32 FMA arithmetic, HBM/latency bound.
#include <stdlib.h>

#include "common.h"
#include "igemm.h"

#define SK_M 32

// one wave: feature n of C = A B^T (+ bias); Brow = B + n * ldb
__device__ __forceinline__ void skinny_feature(const float* __restrict__ A, long long lda, const float* __restrict__ Brow, float* __restrict__ C,
                                               long long ldc, const float* __restrict__ bias, int n, int M, int K, int accumulate, int lane) {
  float acc[SK_M];
#pragma unroll
  for (int m = 0; m < SK_M; ++m) acc[m] = 0.f;
  for (int k0 = lane * 8; k0 < K; k0 += 512) {
    const float4 b0 = *reinterpret_cast<const float4*>(Brow + k0);
    const float4 b1 = *reinterpret_cast<const float4*>(Brow + k0 + 4);
#pragma unroll
    for (int m = 0; m < SK_M; ++m) {
      if (m < M) {
        const float4 a0 = *reinterpret_cast<const float4*>(A + m * lda + k0);
        const float4 a1 = *reinterpret_cast<const float4*>(A + m * lda + k0 + 4);
        float s = acc[m];
        s = fmaf(a0.x, b0.x, s); s = fmaf(a0.y, b0.y, s); s = fmaf(a0.z, b0.z, s); s = fmaf(a0.w, b0.w, s);
        s = fmaf(a1.x, b1.x, s); s = fmaf(a1.y, b1.y, s); s = fmaf(a1.z, b1.z, s); s = fmaf(a1.w, b1.w, s);
        acc[m] = s;
      }
    }
  }
  // butterfly: after the stage with partner distance d (32, 16, 8, 4, 2) a lane keeps the half of its values selected by bit d
#define SK_STAGE(D, CNT)                                                                       \
  {                                                                                            \
    const bool up = (lane & D) != 0;                                                           \
    _Pragma("unroll") for (int j = 0; j < CNT; ++j) {                                          \
      const float keep = up ? acc[j + CNT] : acc[j];                                           \
      const float give = up ? acc[j] : acc[j + CNT];                                           \
      acc[j] = keep + __shfl_xor(give, D);                                                     \
    }                                                                                          \
  }
  SK_STAGE(32, 16) SK_STAGE(16, 8) SK_STAGE(8, 4) SK_STAGE(4, 2) SK_STAGE(2, 1)
#undef SK_STAGE
  float v = acc[0] + __shfl_xor(acc[0], 1);
  // value index held by this lane: bit4 <- lane bit 5 (distance 32 chose the upper 16), bit3 <- lane bit 4, ... bit0 <- lane bit 1
  const int m = ((lane >> 5) & 1) << 4 | ((lane >> 4) & 1) << 3 | ((lane >> 3) & 1) << 2 | ((lane >> 2) & 1) << 1 | ((lane >> 1) & 1);
  if ((lane & 1) == 0 && m < M) {
    if (bias) v += bias[n];
    float* dst = C + m * ldc + n;
    *dst = accumulate ? *dst + v : v;
  }
}

__global__ void __launch_bounds__(256) skinny_nt_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ B, long long ldb,
                                                        float* __restrict__ C, long long ldc, const float* __restrict__ bias, int M, int N, int K,
                                                        int accumulate) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  skinny_feature(A, lda, B + (long long)n * ldb, C, ldc, bias, n, M, K, accumulate, threadIdx.x & 63);
}

// A whole family of linear layers in ONE launch: y_i[M][n_i] = x_i[M][K] W_i[n_i][K]^T + b_i for i < n_items (every ResBlock's
// emb_layers / emb_z_layers Linear of a network pass: they all read SiLU(emb) / SiLU(shift_emb), which exist before the first block runs --
// the reference issues them one by one inside each block, model/module.py:287-293,371-380).  items / first are DEVICE arrays; first[i] = index
// of item i's first feature in the concatenated feature list (first[n_items] = total).  One wave per feature, located by binary search.
struct LinearItem { const float* x; const float* w; const float* bias; float* y; int n_out; int pad; };
__global__ void __launch_bounds__(256) skinny_group_kernel(const LinearItem* __restrict__ items, const int* __restrict__ first, int n_items, int total,
                                                           int M, int K) {
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= total) return;
  int lo = 0, hi = n_items - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (first[mid] <= f) lo = mid; else hi = mid - 1; }
  const LinearItem it = items[lo];
  const int n = f - first[lo];
  skinny_feature(it.x, K, it.w + (long long)n * K, it.y, it.n_out, it.bias, n, it.pad > 0 ? it.pad : M, K, 0, threadIdx.x & 63);      // pad = rows of this item (0: M)
}

int skinny_group_launch(const void* items, const int* first, int n_items, int total, int M, int K, hipStream_t s) {
  hipLaunchKernelGGL(skinny_group_kernel, dim3((total + 3) / 4), dim3(256), 0, s, (const LinearItem*)items, first, n_items, total, M, K);
  return pdae_launch_status("linear_group");
}

bool skinny_ok(int transA, int transB, int M, int N, int K, float alpha, long long lda, long long ldb, const float* A, const float* B, int batch) {
  const bool off = pdae_knob(KNOB_NO_SKINNY) != 0;        // tuning aid: force the MFMA tile kernel
  if (off || transA || !transB || batch != 1 || alpha != 1.0f || M > SK_M || N < 64) return false;
  return (K & 7) == 0 && (lda & 3) == 0 && (ldb & 3) == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0;
}

int skinny_launch(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, const float* bias, int M, int N, int K,
                  int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(skinny_nt_kernel, dim3((N + 3) / 4), dim3(256), 0, s, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate);
  return pdae_launch_status("skinny_nt");
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Backward of a family of M <= 32 linear layers y_i = x_i W_i^T + b_i in ONE launch (the emb_layers / emb_z_layers pair of a ResBlock:
// module.py:287-293, 371-380 under autograd):   dW_i (+)= dy_i^T x_i,   db_i (+)= column sums of dy_i,   dx_i (+)= dy_i W_i.
// Exact fp32 FMA, fixed summation order (deterministic).  Block ranges: per item first n_out/8 blocks of weight-gradient rows (thread = 4 rows x
// one k quad: the x tile, 32 x K, is read once per thread-quad of rows), then (dx != NULL) M * ceil(K/128) blocks, one per (batch row, k tile).
// ---------------------------------------------------------------------------------------------------------------------------------
struct LinearBwdItem { const float* x; const float* dy; const float* w; float* dw; float* db; float* dx; int n_out; int acc_w; int acc_x; int pad; };
__global__ void __launch_bounds__(256) linear_bwd_group_kernel(const LinearBwdItem* __restrict__ items, const int* __restrict__ first, int n_items, int M, int K) {
  __shared__ float4 red[256];
  int lo = 0, hi = n_items - 1;
  const int b = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (first[mid] <= b) lo = mid; else hi = mid - 1; }
  const LinearBwdItem it = items[lo];
  const int lb = b - first[lo], wblocks = (it.n_out + 7) >> 3;
  const int t = threadIdx.x, KQ = K >> 2;
  if (lb < wblocks) {                                    // ---- dW rows lb*8 .. +7 (+ db)
    const int rg = t / 128, kq0 = t % 128;               // two groups of four rows, 128 k-quad lanes
    const int n0 = lb * 8 + rg * 4;
    for (int kq = kq0; kq < KQ; kq += 128) {
      float4 a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8                          // eight rows of loads in flight: the serial form was a chain of M memory round trips
      for (int m = 0; m < M; ++m) {
        const float4 xv = *reinterpret_cast<const float4*>(it.x + (size_t)m * K + kq * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = (n0 + j < it.n_out) ? it.dy[(size_t)m * it.n_out + n0 + j] : 0.f;
          a[j].x = fmaf(d, xv.x, a[j].x); a[j].y = fmaf(d, xv.y, a[j].y); a[j].z = fmaf(d, xv.z, a[j].z); a[j].w = fmaf(d, xv.w, a[j].w);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + j < it.n_out) {
          float4* dst = reinterpret_cast<float4*>(it.dw + (size_t)(n0 + j) * K + kq * 4);
          if (it.acc_w) { const float4 u = *dst; a[j].x += u.x; a[j].y += u.y; a[j].z += u.z; a[j].w += u.w; }
          *dst = a[j];
        }
    }
    if (it.db && kq0 == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + j < it.n_out) {
          float sdb = 0.f;
#pragma unroll 8
          for (int m = 0; m < M; ++m) sdb += it.dy[(size_t)m * it.n_out + n0 + j];
          it.db[n0 + j] = it.acc_w ? it.db[n0 + j] + sdb : sdb;
        }
    }
    return;
  }
  // ---- dx: block (m, k tile of 128 floats) = lb - wblocks;  dx[m][k] (+)= sum_n dy[m][n] W[n][k]; 8 slices of the thread block split n
  // (each W row segment is a 512-byte coalesced read), combined through LDS in fixed order
  const int kt_n = (KQ + 31) >> 5, xb = lb - wblocks, m = xb / kt_n, kq = (xb - m * kt_n) * 32 + (t & 31), sl = t >> 5;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (kq < KQ) {
    const int per = (it.n_out + 7) >> 3, nb = sl * per, ne = min(it.n_out, nb + per);
#pragma unroll 8
    for (int n = nb; n < ne; ++n) {
      const float d = it.dy[(size_t)m * it.n_out + n];
      const float4 wv = *reinterpret_cast<const float4*>(it.w + (size_t)n * K + kq * 4);
      a.x = fmaf(d, wv.x, a.x); a.y = fmaf(d, wv.y, a.y); a.z = fmaf(d, wv.z, a.z); a.w = fmaf(d, wv.w, a.w);
    }
  }
  red[t] = a;
  __syncthreads();
  if (sl == 0 && kq < KQ) {
#pragma unroll
    for (int j = 1; j < 8; ++j) { const float4 u = red[t + 32 * j]; a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
    float4* dst = reinterpret_cast<float4*>(it.dx + (size_t)m * K + kq * 4);
    if (it.acc_x) { const float4 e = *dst; a.x += e.x; a.y += e.y; a.z += e.z; a.w += e.w; }
    *dst = a;
  }
}

int linear_bwd_group_launch(const void* items, const int* first, int n_items, int total_blocks, int M, int K, hipStream_t s) {
  hipLaunchKernelGGL(linear_bwd_group_kernel, dim3(total_blocks), dim3(256), 0, s, (const LinearBwdItem*)items, first, n_items, M, K);
  return pdae_launch_status("linear_bwd_group");
}
