// MLPSkipNet layer body (latent DPM, config #5): per row r of a [R][C] activation
//     m = u * (1 + e)            condition scale           (model/mlp_skip_net.py:126-131)
//     v = LayerNorm_C(m) * gamma + beta   (norm = 1)       (:133, nn.LayerNorm eps 1e-5, biased variance)
//     y = silu(v)                                           (:137)
// and its backward.  HBM-bound: one block per row, the row lives in registers (C <= 8192), two-pass mean / variance.
#include "common.h"
#include "kernels.h"

#define MLP_T 256
#define MLP_MAXE 32        // elements per thread: C <= 8192

__device__ __forceinline__ float mlp_silu(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float mlp_dsilu(float v) {
  float sg = 1.0f / (1.0f + expf(-v));
  return sg * (1.0f + v * (1.0f - sg));
}

// block-wide sum of one float per thread (all threads get the result)
__device__ __forceinline__ float mlp_block_sum(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(MLP_T) mlp_modln_fwd_kernel(const float* __restrict__ u, const float* __restrict__ e, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int C, int norm, int act, float eps,
                                                              float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd) {
  __shared__ float red[4];
  const int r = blockIdx.x, t = threadIdx.x;
  const size_t base = (size_t)r * C;
  float m[MLP_MAXE];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MLP_MAXE; ++j) {
    const int c = t + j * MLP_T;
    m[j] = 0.f;
    if (c < C) { float v = u[base + c]; if (e) v *= 1.0f + e[base + c]; m[j] = v; s += v; }
  }
  float mu = 0.f, rs = 1.f;
  if (norm) {
    mu = mlp_block_sum(s, red) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MLP_MAXE; ++j) { const int c = t + j * MLP_T; if (c < C) { float d = m[j] - mu; q += d * d; } }
    rs = rsqrtf(mlp_block_sum(q, red) / (float)C + eps);
    if (t == 0) { mean[r] = mu; rstd[r] = rs; }
  }
#pragma unroll
  for (int j = 0; j < MLP_MAXE; ++j) {
    const int c = t + j * MLP_T;
    if (c < C) {
      float v = norm ? (m[j] - mu) * rs * gamma[c] + beta[c] : m[j];
      y[base + c] = act ? mlp_silu(v) : v;
    }
  }
}

// backward of the above.  Writes du, de (when e != null) and the per-element parameter-gradient terms tg = dv * xhat, tb = dv
// (summed over rows by the caller with pdae_colsum).
__global__ void __launch_bounds__(MLP_T) mlp_modln_bwd_kernel(const float* __restrict__ u, const float* __restrict__ e, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ dy, int C, int norm, int act,
                                                              float* __restrict__ du, float* __restrict__ de, float* __restrict__ tg,
                                                              float* __restrict__ tb) {
  __shared__ float red[4];
  const int r = blockIdx.x, t = threadIdx.x;
  const size_t base = (size_t)r * C;
  const float mu = norm ? mean[r] : 0.f, rs = norm ? rstd[r] : 1.f;
  float xh[MLP_MAXE], dx[MLP_MAXE];       // xhat, d xhat
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < MLP_MAXE; ++j) {
    const int c = t + j * MLP_T;
    xh[j] = 0.f; dx[j] = 0.f;
    if (c < C) {
      float m = u[base + c]; if (e) m *= 1.0f + e[base + c];
      const float x = (m - mu) * rs;
      const float g = norm ? gamma[c] : 1.f;
      const float v = norm ? x * g + beta[c] : m;
      const float dv = act ? dy[base + c] * mlp_dsilu(v) : dy[base + c];
      if (norm) { tg[base + c] = dv * x; tb[base + c] = dv; }
      xh[j] = x; dx[j] = dv * g;
      s1 += dx[j]; s2 += dx[j] * x;
    }
  }
  float a1 = 0.f, a2 = 0.f;
  if (norm) { a1 = mlp_block_sum(s1, red) / (float)C; a2 = mlp_block_sum(s2, red) / (float)C; }
#pragma unroll
  for (int j = 0; j < MLP_MAXE; ++j) {
    const int c = t + j * MLP_T;
    if (c < C) {
      const float dm = norm ? rs * (dx[j] - a1 - xh[j] * a2) : dx[j];
      const float uu = u[base + c];
      if (e) { du[base + c] = dm * (1.0f + e[base + c]); de[base + c] = dm * uu; }
      else du[base + c] = dm;
    }
  }
}

int k_mlp_modln_fwd(const float* u, const float* e, const float* gamma, const float* beta, int R, int C, int norm, int act, float eps, float* y,
                    float* mean, float* rstd, hipStream_t st) {
  PDAE_CHECK_ARG(R > 0 && C > 0 && C <= MLP_T * MLP_MAXE, "mlp_modln: need 0 < C <= %d (got %d)", MLP_T * MLP_MAXE, C);
  PDAE_CHECK_ARG(u && y && (!norm || (gamma && beta && mean && rstd)), "mlp_modln_fwd: null pointer");
  hipLaunchKernelGGL(mlp_modln_fwd_kernel, dim3(R), dim3(MLP_T), 0, st, u, e, gamma, beta, C, norm, act, eps, y, mean, rstd);
  return pdae_launch_status("mlp_modln_fwd");
}

int k_mlp_modln_bwd(const float* u, const float* e, const float* gamma, const float* beta, const float* mean, const float* rstd, const float* dy,
                    int R, int C, int norm, int act, float* du, float* de, float* tg, float* tb, hipStream_t st) {
  PDAE_CHECK_ARG(R > 0 && C > 0 && C <= MLP_T * MLP_MAXE, "mlp_modln: need 0 < C <= %d (got %d)", MLP_T * MLP_MAXE, C);
  PDAE_CHECK_ARG(u && dy && du && (!e || de) && (!norm || (gamma && beta && mean && rstd && tg && tb)), "mlp_modln_bwd: null pointer");
  hipLaunchKernelGGL(mlp_modln_bwd_kernel, dim3(R), dim3(MLP_T), 0, st, u, e, gamma, beta, mean, rstd, dy, C, norm, act, du, de, tg, tb);
  return pdae_launch_status("mlp_modln_bwd");
}
