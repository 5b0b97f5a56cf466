// GroupNorm(32,C) + AdaGN + SiLU family on NHWC fp32 (HBM-bound kernels).
//
// Replaces nn.GroupNorm / SiLU / Dropout / AvgPool2d / the AdaGN arithmetic of
// model/module.py:56-63, 241, 257-263, 279-284, 293-294, 379-381 and their autograd backward.
//
// Forward is split so that the only full-tensor passes are (1) one statistics read and (2) one
// fused "apply" pass  v = a[n,c]*(x - mu[n,c]) + b[n,c];  y = silu(v) * dropmask  (optionally
// 2x2-average-pooled, optionally also emitting the pooled raw x for the skip path, and reading a
// virtual channel-concat of two tensors so torch.cat is never materialised).  The per-(n,c)
// coefficients fold gamma/beta, the timestep (scale,shift) and the semantic (z_scale,z_shift) pairs.
//
// Backward needs only two reductions per (n,c):  S0 = sum dv,  S1 = sum dv*(x-mu);  every parameter
// gradient (gamma, beta, scale/shift, z_scale/z_shift) and the GroupNorm input gradient derive from
// them (see gn_bwd_finalize).
#include "common.h"
#include "kernels.h"

struct Src2 { const float* x0; const float* x1; int C0, C1; };
#define UNR 4        // memory-level parallelism of the streaming loops: independent float4 loads per thread

__device__ __forceinline__ float4 ld4(const Src2& s, size_t pix, int c) {
  return c < s.C0 ? *reinterpret_cast<const float4*>(s.x0 + pix * s.C0 + c)
                  : *reinterpret_cast<const float4*>(s.x1 + pix * s.C1 + (c - s.C0));
}
__device__ __forceinline__ float ld1(const Src2& s, size_t pix, int c) {
  return c < s.C0 ? s.x0[pix * s.C0 + c] : s.x1[pix * s.C1 + (c - s.C0)];
}
// SiLU and its derivative through v_exp_f32 / v_rcp_f32 (~2 ulp): the GroupNorm apply / backward kernels evaluate them once per element and
// were VALU-bound, not HBM-bound, with expf + IEEE divisions (two per derivative)
__device__ __forceinline__ float sigmoidf_fast(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float siluf(float v) { return v * sigmoidf_fast(v); }
__device__ __forceinline__ float dsiluf(float v) {
  const float sg = sigmoidf_fast(v);
  return sg * (1.0f + v * (1.0f - sg));
}

// Philox2x32-10 counter RNG: the dropout keep-mask is a pure function of (seed, offset, element quad), so backward regenerates it instead of
// storing a mask tensor.  One call yields 64 bits = four 16-bit uniforms, one per element of the quad (keep-probability resolution 2^-16);
// the 4x32 variant with a 32-bit uniform per element spent 40 quarter-rate integer multiplies per quad -- three times per dropout layer and
// step (forward, backward reduce, backward apply) -- and made those kernels compute-bound at 4 TB/s.
__device__ __forceinline__ uint2 philox2x32(uint2 c, unsigned k) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned hi = __umulhi(0xD256D193u, c.x), lo = 0xD256D193u * c.x;
    c = make_uint2(hi ^ k ^ c.y, lo);
    k += 0x9E3779B9u;
  }
  return c;
}
__device__ __forceinline__ float4 drop_mask(unsigned long long seed, unsigned long long offset, unsigned long long quad, float p, float scale) {
  const uint2 r = philox2x32(make_uint2((unsigned)quad, (unsigned)offset ^ ((unsigned)(quad >> 32) * 0x85EBCA6Bu)),
                             (unsigned)seed ^ ((unsigned)(seed >> 32) * 0xC2B2AE35u) ^ ((unsigned)(offset >> 32) * 0x27D4EB2Fu));
  const unsigned thr = (unsigned)(p * 65536.0f + 0.5f);      // drop when the 16-bit uniform is below p
  return make_float4((r.x & 0xffffu) >= thr ? scale : 0.f, (r.x >> 16) >= thr ? scale : 0.f, (r.y & 0xffffu) >= thr ? scale : 0.f, (r.y >> 16) >= thr ? scale : 0.f);
}

// ----------------------------------------------------------------------------------------------
// "last block finishes the job": kernels that write per-block partials take a ticket per sample; the block that draws the last ticket of its
// sample runs the finalize stage itself instead of a second launch (a dependent launch costs 1.5-2 us of boundary plus a ~10 us tail for a
// few hundred flops).  Hand-off protocol of the CDNA programming guide (G16): plain stores -> __syncthreads() -> one lane: agent-scope
// release fence, explicit vmcnt(0), relaxed agent atomic; the last arriver: agent-scope acquire fence -> __syncthreads() -> plain loads.
// The ticket word is reset by the last arriver, so a zero-initialised array stays valid across launches (launches on one stream are serial).
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ bool last_block_of(unsigned* ticket, unsigned count) {
  __shared__ int is_last;
  __syncthreads();                                       // every partial of this block has been stored (barrier waits for vmcnt)
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = prev == count - 1;
    if (is_last) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
  return is_last != 0;
}

// ----------------------------------------------------------------------------------------------
// statistics: per (n, chunk) block, threads = channel quads x pixel lanes; shifted sums (shift =
// first element of the group) keep E[x^2]-E[x]^2 benign; finalize combines chunks in double.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void gn_stats_partial_body(const Src2& s, int HW, int C, int G, int chunk, float* __restrict__ part) {
  __shared__ float red[2 * 1024];            // [PL][C] x {s1,s2} with PL*C <= 1024
  const int n = blockIdx.y, sidx = blockIdx.x, S = gridDim.x;
  const int NQ = C >> 2, PL = 256 / NQ, cg = C / G;
  const int t = threadIdx.x, q = t % NQ, pl = t / NQ, c = q * 4;
  const size_t base = (size_t)n * HW;
  const int p0 = sidx * chunk, p1 = min(HW, p0 + chunk);
  if (pl < PL) {
    float K[4], s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) K[j] = ld1(s, base, ((c + j) / cg) * cg);
    for (int p = p0 + pl; p < p1; p += UNR * PL) {            // UNR independent 16-byte loads in flight per thread
      float4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) if (p + u * PL < p1) v[u] = ld4(s, base + p + u * PL, c);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (p + u * PL >= p1) break;
        float d;
        d = v[u].x - K[0]; s1[0] += d; s2[0] += d * d;
        d = v[u].y - K[1]; s1[1] += d; s2[1] += d * d;
        d = v[u].z - K[2]; s1[2] += d; s2[2] += d * d;
        d = v[u].w - K[3]; s1[3] += d; s2[3] += d * d;
      }
    }
    if ((cg & 3) == 0) {                       // a thread's four channels lie in one group: one (s1, s2) pair per thread, summed in fixed order below
      red[(pl * NQ + q) * 2] = (s1[0] + s1[1]) + (s1[2] + s1[3]); red[(pl * NQ + q) * 2 + 1] = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { red[(pl * C + c + j) * 2] = s1[j]; red[(pl * C + c + j) * 2 + 1] = s2[j]; }
    }
  }
  __syncthreads();
  if (t < G) {
    float a = 0.f, b = 0.f;
    if ((cg & 3) == 0) {                         // PL x cg/4 = 256 / G entries per group (was PL x cg single-thread LDS reads: the tail of every block)
      const int qg = cg >> 2;
      for (int l = 0; l < PL; ++l)
        for (int j = 0; j < qg; ++j) { const float2 v = *reinterpret_cast<const float2*>(&red[(l * NQ + t * qg + j) * 2]); a += v.x; b += v.y; }
    } else {
      for (int l = 0; l < PL; ++l)
        for (int j = 0; j < cg; ++j) { a += red[(l * C + t * cg + j) * 2]; b += red[(l * C + t * cg + j) * 2 + 1]; }
    }
    float* o = part + (((size_t)n * S + sidx) * G + t) * 2;
    o[0] = a; o[1] = b;
  }
}
__global__ void __launch_bounds__(256) gn_stats_partial_kernel(Src2 s, int HW, int C, int G, int chunk, float* __restrict__ part) {
  gn_stats_partial_body(s, HW, C, G, chunk, part);
}

// one wave per (n, group): lanes over the S <= 64 partials, butterfly-reduced in double
__global__ void __launch_bounds__(256) gn_stats_finalize_kernel(Src2 s, int N, int HW, int C, int G, int S, float eps, const float* __restrict__ part,
                                                                float* __restrict__ mean, float* __restrict__ rstd) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= N * G) return;
  const int n = i / G, g = i - n * G, cg = C / G;
  double a = 0.0, b = 0.0;
  for (int k = lane; k < S; k += 64) { const float* o = part + (((size_t)n * S + k) * G + g) * 2; a += o[0]; b += o[1]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
  if (lane) return;
  double cnt = (double)cg * HW;
  double K = ld1(s, (size_t)n * HW, g * cg);
  double m = a / cnt;
  double var = b / cnt - m * m;
  if (var < 0.0) var = 0.0;
  mean[i] = (float)(K + m);
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

// coef[0]=mu, coef[1]=a, coef[2]=b, each [N][C]
__global__ void gn_coef_kernel(int N, int C, int G, const float* __restrict__ mean, const float* __restrict__ rstd,
                               const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ ss, const float* __restrict__ zss, float* __restrict__ coef) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  int n = i / C, c = i - n * C, g = c / (C / G);
  float r = rstd[n * G + g];
  float k = gamma[c] * r, b = beta[c];
  if (ss) { float sc = 1.0f + ss[(size_t)n * 2 * C + c]; k *= sc; b = b * sc + ss[(size_t)n * 2 * C + C + c]; }
  if (zss) { float sc = 1.0f + zss[(size_t)n * 2 * C + c]; k *= sc; b = b * sc + zss[(size_t)n * 2 * C + C + c]; }
  coef[i] = mean[n * G + g];
  coef[(size_t)N * C + i] = k;
  coef[(size_t)2 * N * C + i] = b;
}

// statistics finalize + coefficient fold in one launch: block = sample n; waves reduce the per-chunk partials of their groups
// (double butterfly), then all threads write the per-(n,c) coefficients
// coef = [mu | a | b] of sample n from the group statistics in LDS:  y = a * (x - mu) + b  with the GroupNorm affine and the AdaGN scale / shift
// pairs (ss: time embedding, zss: latent shift embedding) folded in
__device__ __forceinline__ void gn_fold_coef(int n, int N, int C, int cg, const float* smean, const float* srstd, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, const float* __restrict__ ss, const float* __restrict__ zss,
                                             float* __restrict__ coef) {
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cg;
    float k = gamma[c] * srstd[g], b = beta[c];
    if (ss) { float sc = 1.0f + ss[(size_t)n * 2 * C + c]; k *= sc; b = b * sc + ss[(size_t)n * 2 * C + C + c]; }
    if (zss) { float sc = 1.0f + zss[(size_t)n * 2 * C + c]; k *= sc; b = b * sc + zss[(size_t)n * 2 * C + C + c]; }
    const size_t i = (size_t)n * C + c;
    coef[i] = smean[g];
    coef[(size_t)N * C + i] = k;
    coef[(size_t)2 * N * C + i] = b;
  }
}

__device__ __forceinline__ void gn_finalize_coef_body(const Src2& s, int n, int N, int HW, int C, int G, int S, float eps, const float* __restrict__ part,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ ss,
                                                      const float* __restrict__ zss, float* __restrict__ mean, float* __restrict__ rstd,
                                                      float* __restrict__ coef) {
  __shared__ float smean[64], srstd[64];
  const int t = threadIdx.x, lane = t & 63, cg = C / G;
  // eight groups per wave AT ONCE (lane = group-in-octet * 8 + slice of the partials): one round trip for the partial sums, one for the shift
  // element and one fp64 divide / sqrt sequence per wave.  (One group per wave iteration -- eight dependent global-load round trips and eight
  // fp64 sequences in a row -- made this 300-flop kernel take 11 us, 152 times per training step.)
  for (int g0 = (t >> 6) * 8; g0 < G; g0 += 32) {
    const int g = g0 + (lane >> 3), j = lane & 7;
    double a = 0.0, b = 0.0;
    if (g < G) for (int k = j; k < S; k += 8) { const float* o = part + (((size_t)n * S + k) * G + g) * 2; a += o[0]; b += o[1]; }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    if (j == 0 && g < G) {
      const double cnt = (double)cg * HW, K = ld1(s, (size_t)n * HW, g * cg), m = a / cnt;
      double var = b / cnt - m * m;
      if (var < 0.0) var = 0.0;
      smean[g] = (float)(K + m); srstd[g] = (float)(1.0 / sqrt(var + (double)eps));
      mean[n * G + g] = smean[g]; rstd[n * G + g] = srstd[g];
    }
  }
  __syncthreads();
  gn_fold_coef(n, N, C, cg, smean, srstd, gamma, beta, ss, zss, coef);
}

__global__ void __launch_bounds__(256) gn_finalize_coef_kernel(Src2 s, int N, int HW, int C, int G, int S, float eps, const float* __restrict__ part,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ ss, const float* __restrict__ zss,
                                                               float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ coef) {
  gn_finalize_coef_body(s, blockIdx.x, N, HW, C, G, S, eps, part, gamma, beta, ss, zss, mean, rstd, coef);
}

// Group statistics + coefficients from the partial sums the 3x3 convolution kernel left behind while it stored the tensor (conv3x3p.hip,
// PatchParams::stat_part): per source [N][tpi wave-tiles][channels / 4] x (sum, sum of squares), unshifted fp32 over 128 pixels x 4 channels
// each, combined here in fp64.  A group is a run of cg / 4 channel quads of the virtual concat [x0 | x1]; each quad lies in one source.
__global__ void __launch_bounds__(256) gn_coef_from_conv_stats_kernel(int N, int HW, int C0, int C1, int G, float eps, const float2* __restrict__ part0,
                                                                      int tpi0, const float2* __restrict__ part1, int tpi1,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      const float* __restrict__ ss, const float* __restrict__ zss,
                                                                      float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ coef) {
  // block = (sample, eight groups): 32 lanes per group (two groups per wave), so the chain of dependent loads over the wave-tile partials is
  // tpi / 32 long (it was tpi / 8 with one block per sample: 5.8 us per launch, ~110 launches per training step, all latency)
  __shared__ float smean[8], srstd[8];
  const int gpb = G < 8 ? G : 8, bps = (G + gpb - 1) / gpb;                      // groups per block, blocks per sample
  const int n = blockIdx.x / bps, gb = (blockIdx.x - n * bps) * gpb;
  const int t = threadIdx.x, lane = t & 63, C = C0 + C1, cg = C / G, qg = cg >> 2, nq0 = C0 >> 2, nq1 = C1 >> 2;
  const int gl = (t >> 6) * 2 + (lane >> 5), g = gb + gl, j = lane & 31;
  double a = 0.0, b = 0.0;
  if (gl < gpb && g < G)
    for (int q = 0; q < qg; ++q) {
      const int cq = g * qg + q;
      const bool first = cq < nq0;
      const float2* src = first ? part0 + (size_t)n * tpi0 * nq0 + cq : part1 + (size_t)n * tpi1 * nq1 + (cq - nq0);
      const int tpi = first ? tpi0 : tpi1, nq = first ? nq0 : nq1;
#pragma unroll 4
      for (int k = j; k < tpi; k += 32) { const float2 v = src[(size_t)k * nq]; a += v.x; b += v.y; }
    }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
  if (j == 0 && gl < gpb && g < G) {
    const double cnt = (double)cg * HW, m = a / cnt;
    double var = b / cnt - m * m;
    if (var < 0.0) var = 0.0;
    smean[gl] = (float)m; srstd[gl] = (float)(1.0 / sqrt(var + (double)eps));
    mean[n * G + g] = smean[gl]; rstd[n * G + g] = srstd[gl];
  }
  __syncthreads();
  const int c_lo = gb * cg, c_hi = min(C, (gb + gpb) * cg);                       // this block's channels
  for (int c = c_lo + t; c < c_hi; c += 256) {
    const int gi = c / cg - gb;
    float k = gamma[c] * srstd[gi], bb = beta[c];
    if (ss) { float sc = 1.0f + ss[(size_t)n * 2 * C + c]; k *= sc; bb = bb * sc + ss[(size_t)n * 2 * C + C + c]; }
    if (zss) { float sc = 1.0f + zss[(size_t)n * 2 * C + c]; k *= sc; bb = bb * sc + zss[(size_t)n * 2 * C + C + c]; }
    const size_t i = (size_t)n * C + c;
    coef[i] = smean[gi];
    coef[(size_t)N * C + i] = k;
    coef[(size_t)2 * N * C + i] = bb;
  }
}

// Partial statistics of a tensor whose producer could not leave them (the 3 -> 128 stem on the edge kernel), in the FORMAT the convolution epilogues
// write -- part[image][tpi runs of `run` pixels][C / 4] x (sum, sum of squares), unshifted fp32 over run x 4 values, combined in fp64 by
// gn_coef_from_conv_stats_kernel.  One pass serves every GroupNorm that reads the tensor: the stem output feeds input_blocks.1 and, as the last skip,
// the final output block of BOTH branches of ShiftUNet -- three statistics passes over 128 x 128 x 128 x N floats (two of them over a 256-channel
// concat) became one.  block = (run, image), threads = channel quads x pixel lanes.
__global__ void __launch_bounds__(256) gn_stats_quads_kernel(const float* __restrict__ x, int HW, int C, int run, float2* __restrict__ part) {
  __shared__ float2 red[256];
  const int NQ = C >> 2, PL = 256 / NQ, t = threadIdx.x, q = t % NQ, pl = t / NQ;
  const int n = blockIdx.y, k = blockIdx.x, tpi = gridDim.x;
  const int p0 = k * run, p1 = min(HW, p0 + run);
  float s1 = 0.f, s2 = 0.f;
  if (pl < PL) {
    const float* b = x + ((size_t)n * HW) * C + q * 4;
    for (int p = p0 + pl; p < p1; p += PL) {
      const float4 v = *reinterpret_cast<const float4*>(b + (size_t)p * C);
      s1 += (v.x + v.y) + (v.z + v.w);
      s2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, s2))));
    }
    red[pl * NQ + q] = make_float2(s1, s2);
  }
  __syncthreads();
  if (t < NQ) {
    float a = 0.f, b2 = 0.f;
    for (int l = 0; l < PL; ++l) { const float2 v = red[l * NQ + t]; a += v.x; b2 += v.y; }
    part[((size_t)n * tpi + k) * NQ + t] = make_float2(a, b2);
  }
}

// statistics + finalize + coefficient fold in ONE launch: the partial-sum kernel, whose last block per sample finishes that sample
__global__ void __launch_bounds__(256) gn_stats_coef_fused_kernel(Src2 s, int N, int HW, int C, int G, int chunk, float eps, float* __restrict__ part,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const float* __restrict__ ss, const float* __restrict__ zss,
                                                                  float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ coef,
                                                                  unsigned* __restrict__ ticket) {
  gn_stats_partial_body(s, HW, C, G, chunk, part);
  if (last_block_of(ticket + blockIdx.y, gridDim.x))
    gn_finalize_coef_body(s, blockIdx.y, N, HW, C, G, gridDim.x, eps, part, gamma, beta, ss, zss, mean, rstd, coef);
}

// ----------------------------------------------------------------------------------------------
// apply: y = act(a*(x-mu)+b) [* dropmask]; mode 0 = same resolution, 1 = 2x2 average pool of y
// (and of raw x into xpool when given).  One thread per output float4.
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gn_apply_kernel(Src2 s, int N, int H, int W, int C, const float* __restrict__ coef, int act, int mode,
                                                       float* __restrict__ y, float* __restrict__ xpool,
                                                       float drop_p, unsigned long long seed, unsigned long long offset) {
  const int NQ = C >> 2;
  const int Ho = mode == 1 ? H >> 1 : H, Wo = mode == 1 ? W >> 1 : W;
  const size_t total = (size_t)N * Ho * Wo * NQ;
  const size_t NC = (size_t)N * C;
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int q = (int)(i % NQ); size_t pix = i / NQ; int c = q * 4;
    int n = (int)(pix / ((size_t)Ho * Wo));
    const float4 mu = *reinterpret_cast<const float4*>(coef + (size_t)n * C + c);
    const float4 a = *reinterpret_cast<const float4*>(coef + NC + (size_t)n * C + c);
    const float4 b = *reinterpret_cast<const float4*>(coef + 2 * NC + (size_t)n * C + c);
    float4 o;
    if (mode == 0) {
      float4 x = ld4(s, pix, c);
      o.x = a.x * (x.x - mu.x) + b.x; o.y = a.y * (x.y - mu.y) + b.y; o.z = a.z * (x.z - mu.z) + b.z; o.w = a.w * (x.w - mu.w) + b.w;
      if (act) { o.x = siluf(o.x); o.y = siluf(o.y); o.z = siluf(o.z); o.w = siluf(o.w); }
      if (drop_p > 0.f) { float4 m = drop_mask(seed, offset, i, drop_p, dscale); o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w; }
    } else {
      int rem = (int)(pix - (size_t)n * Ho * Wo); int oy = rem / Wo, ox = rem - oy * Wo;
      o = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 xs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        size_t ip = ((size_t)n * H + (2 * oy + (d >> 1))) * W + 2 * ox + (d & 1);
        float4 x = ld4(s, ip, c);
        float4 v;
        v.x = a.x * (x.x - mu.x) + b.x; v.y = a.y * (x.y - mu.y) + b.y; v.z = a.z * (x.z - mu.z) + b.z; v.w = a.w * (x.w - mu.w) + b.w;
        if (act) { v.x = siluf(v.x); v.y = siluf(v.y); v.z = siluf(v.z); v.w = siluf(v.w); }
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
        xs.x += x.x; xs.y += x.y; xs.z += x.z; xs.w += x.w;
      }
      o.x *= 0.25f; o.y *= 0.25f; o.z *= 0.25f; o.w *= 0.25f;
      if (xpool) *reinterpret_cast<float4*>(xpool + pix * C + c) = make_float4(xs.x * 0.25f, xs.y * 0.25f, xs.z * 0.25f, xs.w * 0.25f);
    }
    *reinterpret_cast<float4*>(y + pix * C + c) = o;
  }
}

// mode-0 apply, streaming form: block = (pixel chunk, sample), thread = (channel quad, pixel lane) so the per-(n,c)
// coefficients live in registers and UNR independent loads are in flight per thread
__global__ void __launch_bounds__(256) gn_apply_stream_kernel(Src2 s, int HW, int C, int N, int chunk, const float* __restrict__ coef, int act,
                                                              float* __restrict__ y, float drop_p, unsigned long long seed,
                                                              unsigned long long offset) {
  const int n = blockIdx.y, NQ = C >> 2, PL = 256 / NQ;
  const int t = threadIdx.x, q = t % NQ, pl = t / NQ, c = q * 4;
  if (pl >= PL) return;
  const size_t NC = (size_t)N * C, base = (size_t)n * HW;
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const float4 mu = *reinterpret_cast<const float4*>(coef + (size_t)n * C + c);
  const float4 a = *reinterpret_cast<const float4*>(coef + NC + (size_t)n * C + c);
  const float4 b = *reinterpret_cast<const float4*>(coef + 2 * NC + (size_t)n * C + c);
  const int p0 = blockIdx.x * chunk, p1 = min(HW, p0 + chunk);
  for (int p = p0 + pl; p < p1; p += UNR * PL) {
    float4 x[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) if (p + u * PL < p1) x[u] = ld4(s, base + p + u * PL, c);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (p + u * PL >= p1) break;
      const size_t pix = base + p + u * PL;
      float4 o;
      o.x = a.x * (x[u].x - mu.x) + b.x; o.y = a.y * (x[u].y - mu.y) + b.y; o.z = a.z * (x[u].z - mu.z) + b.z; o.w = a.w * (x[u].w - mu.w) + b.w;
      if (act) { o.x = siluf(o.x); o.y = siluf(o.y); o.z = siluf(o.z); o.w = siluf(o.w); }
      if (drop_p > 0.f) { float4 m = drop_mask(seed, offset, pix * NQ + q, drop_p, dscale); o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w; }
      *reinterpret_cast<float4*>(y + pix * C + c) = o;
    }
  }
}

// gradient wrt the activated tensor at input-resolution pixel (n,py,px):
//   mode 0: dA[pix]; mode 1 (y was pooled): dA[pool pix]/4; mode 2 (consumer read y nearest-upsampled): sum of the 4 children
__device__ __forceinline__ float4 fetch_da(const float* __restrict__ dA, int mode, int n, int py, int px, int H, int W, int C, int c) {
  if (mode == 0) return *reinterpret_cast<const float4*>(dA + (((size_t)n * H + py) * W + px) * C + c);
  if (mode == 1) {
    float4 v = *reinterpret_cast<const float4*>(dA + (((size_t)n * (H >> 1) + (py >> 1)) * (W >> 1) + (px >> 1)) * C + c);
    return make_float4(v.x * 0.25f, v.y * 0.25f, v.z * 0.25f, v.w * 0.25f);
  }
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    float4 v = *reinterpret_cast<const float4*>(dA + (((size_t)n * 2 * H + 2 * py + (d >> 1)) * 2 * W + 2 * px + (d & 1)) * C + c);
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
  }
  return o;
}

// dv = dA * act'(v) * dropmask, with v recomputed from x and the forward coefficients
__device__ __forceinline__ float4 compute_dv(float4 da, float4 xm, float4 a, float4 b, int act, float drop_p, float dscale,
                                             unsigned long long seed, unsigned long long offset, size_t quad) {
  if (drop_p > 0.f) { float4 m = drop_mask(seed, offset, quad, drop_p, dscale); da.x *= m.x; da.y *= m.y; da.z *= m.z; da.w *= m.w; }
  if (act) {
    da.x *= dsiluf(a.x * xm.x + b.x); da.y *= dsiluf(a.y * xm.y + b.y); da.z *= dsiluf(a.z * xm.z + b.z); da.w *= dsiluf(a.w * xm.w + b.w);
  }
  return da;
}

__device__ __forceinline__ void gn_bwd_reduce_body(const Src2& s, int H, int W, int C, int chunk, const float* __restrict__ coef, int N,
                                                   const float* __restrict__ dA, int act, int mode, float drop_p,
                                                   unsigned long long seed, unsigned long long offset, float* __restrict__ part) {
  __shared__ float red[2 * 1024];
  const int n = blockIdx.y, sidx = blockIdx.x, S = gridDim.x, HW = H * W;
  const int NQ = C >> 2, PL = 256 / NQ;
  const int t = threadIdx.x, q = t % NQ, pl = t / NQ, c = q * 4;
  const size_t NC = (size_t)N * C;
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const int p0 = sidx * chunk, p1 = min(HW, p0 + chunk);
  if (pl < PL) {
    const float4 mu = *reinterpret_cast<const float4*>(coef + (size_t)n * C + c);
    const float4 a = *reinterpret_cast<const float4*>(coef + NC + (size_t)n * C + c);
    const float4 b = *reinterpret_cast<const float4*>(coef + 2 * NC + (size_t)n * C + c);
    float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    for (int p = p0 + pl; p < p1; p += UNR * PL) {
      float4 x[UNR], da[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int pp = p + u * PL;
        if (pp < p1) { const int py = pp / W, px = pp - py * W; x[u] = ld4(s, (size_t)n * HW + pp, c); da[u] = fetch_da(dA, mode, n, py, px, H, W, C, c); }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int pp = p + u * PL;
        if (pp >= p1) break;
        const size_t pix = (size_t)n * HW + pp;
        float4 xm = make_float4(x[u].x - mu.x, x[u].y - mu.y, x[u].z - mu.z, x[u].w - mu.w);
        float4 dv = compute_dv(da[u], xm, a, b, act, drop_p, dscale, seed, offset, pix * NQ + q);
        s0[0] += dv.x; s0[1] += dv.y; s0[2] += dv.z; s0[3] += dv.w;
        s1[0] += dv.x * xm.x; s1[1] += dv.y * xm.y; s1[2] += dv.z * xm.z; s1[3] += dv.w * xm.w;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[(pl * C + c + j) * 2] = s0[j]; red[(pl * C + c + j) * 2 + 1] = s1[j]; }
  }
  __syncthreads();
  for (int cc = t; cc < C; cc += 256) {
    float u = 0.f, v = 0.f;
    for (int l = 0; l < PL; ++l) { u += red[(l * C + cc) * 2]; v += red[(l * C + cc) * 2 + 1]; }
    float* o = part + (((size_t)n * S + sidx) * C + cc) * 2;
    o[0] = u; o[1] = v;
  }
}
__global__ void __launch_bounds__(256) gn_bwd_reduce_kernel(Src2 s, int H, int W, int C, int chunk, const float* __restrict__ coef, int N,
                                                            const float* __restrict__ dA, int act, int mode, float drop_p,
                                                            unsigned long long seed, unsigned long long offset, float* __restrict__ part) {
  gn_bwd_reduce_body(s, H, W, C, chunk, coef, N, dA, act, mode, drop_p, seed, offset, part);
}

// one block per sample n.  Writes d(scale,shift) pairs, the per-(n,c) [c1,c2] apply coefficients and
// the per-(n,c) gamma/beta contributions (summed over n by gn_bwd_param_kernel).
__device__ __forceinline__ void gn_bwd_finalize_body(int n, int N, int HW, int C, int G, int S, const float* __restrict__ part,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ ss,
                                                     const float* __restrict__ zss, float* __restrict__ dss,
                                                     float* __restrict__ dzss, float* __restrict__ c12, float* __restrict__ pgb,
                                                     unsigned* __restrict__ amax0) {
  __shared__ float g1[1024], g2[1024], m1[64], m2[64];
  const int cg = C / G, t = threadIdx.x;
  if (amax0 && n == 0 && t == 0) *amax0 = 0u;      // the apply kernel (next launch on this stream) accumulates max|dx0| into it
  for (int c = t; c < C; c += 256) {
    float S0 = 0.f, S1 = 0.f;
    const float2* o2 = reinterpret_cast<const float2*>(part) + (size_t)n * S * C + c;
#pragma unroll 8
    for (int k = 0; k < S; ++k) { const float2 o = o2[(size_t)k * C]; S0 += o.x; S1 += o.y; }      // independent 8-byte loads, eight in flight
    const float r = rstd[n * G + c / cg], gm = gamma[c], bt = beta[c];
    float sc = 1.f, sh = 0.f, zsc = 1.f;
    if (ss) { sc = 1.0f + ss[(size_t)n * 2 * C + c]; sh = ss[(size_t)n * 2 * C + C + c]; }
    if (zss) zsc = 1.0f + zss[(size_t)n * 2 * C + c];
    const float rS1 = r * S1;
    if (dzss) { dzss[(size_t)n * 2 * C + c] = sc * gm * rS1 + (sc * bt + sh) * S0; dzss[(size_t)n * 2 * C + C + c] = S0; }
    if (dss) { dss[(size_t)n * 2 * C + c] = zsc * (gm * rS1 + bt * S0); dss[(size_t)n * 2 * C + C + c] = zsc * S0; }
    const float kp = sc * zsc;
    pgb[((size_t)n * C + c) * 2] = kp * rS1;       // d gamma contribution
    pgb[((size_t)n * C + c) * 2 + 1] = kp * S0;    // d beta contribution
    g1[c] = gm * kp * S0; g2[c] = gm * kp * rS1;
  }
  __syncthreads();
  if (t < G) {
    float a = 0.f, b = 0.f;
    for (int j = 0; j < cg; ++j) { a += g1[t * cg + j]; b += g2[t * cg + j]; }
    const float inv = 1.0f / ((float)cg * (float)HW);
    m1[t] = a * inv; m2[t] = b * inv;
  }
  __syncthreads();
  for (int c = t; c < C; c += 256) {
    const int g = c / cg; const float r = rstd[n * G + g];
    c12[((size_t)n * C + c) * 2] = r * m1[g];
    c12[((size_t)n * C + c) * 2 + 1] = r * r * m2[g];
  }
}
__global__ void __launch_bounds__(256) gn_bwd_finalize_kernel(int N, int HW, int C, int G, int S, const float* __restrict__ part,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ ss,
                                                              const float* __restrict__ zss, float* __restrict__ dss,
                                                              float* __restrict__ dzss, float* __restrict__ c12, float* __restrict__ pgb,
                                                              unsigned* __restrict__ amax0) {
  gn_bwd_finalize_body(blockIdx.x, N, HW, C, G, S, part, rstd, gamma, beta, ss, zss, dss, dzss, c12, pgb, amax0);
}

__device__ __forceinline__ void gn_bwd_param_one(int c, int N, int C, const float* __restrict__ pgb, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                 int accumulate) {
  float a = 0.f, b = 0.f;
#pragma unroll 8
  for (int n = 0; n < N; ++n) { const float2 v = *reinterpret_cast<const float2*>(pgb + ((size_t)n * C + c) * 2); a += v.x; b += v.y; }    // eight loads in flight
  if (accumulate) { a += dgamma[c]; b += dbeta[c]; }
  dgamma[c] = a; dbeta[c] = b;
}
__global__ void gn_bwd_param_kernel(int N, int C, const float* __restrict__ pgb, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) gn_bwd_param_one(c, N, C, pgb, dgamma, dbeta, accumulate);
}

// reduce + finalize + parameter gradients in ONE launch: the last block of each sample finalizes it, the last finalizer sums the
// per-sample gamma / beta contributions (ticket[0..N) per sample, ticket[N] across samples; see last_block_of)
__global__ void __launch_bounds__(256) gn_bwd_reduce_fused_kernel(Src2 s, int H, int W, int C, int G, int chunk, const float* __restrict__ coef, int N,
                                                                  const float* __restrict__ dA, int act, int mode, float drop_p,
                                                                  unsigned long long seed, unsigned long long offset, float* __restrict__ part,
                                                                  const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, const float* __restrict__ ss,
                                                                  const float* __restrict__ zss, float* __restrict__ dss, float* __restrict__ dzss,
                                                                  float* __restrict__ c12, float* __restrict__ pgb, unsigned* __restrict__ amax0,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, int acc_param,
                                                                  unsigned* __restrict__ ticket) {
  gn_bwd_reduce_body(s, H, W, C, chunk, coef, N, dA, act, mode, drop_p, seed, offset, part);
  const int n = blockIdx.y;
  if (!last_block_of(ticket + n, gridDim.x)) return;
  gn_bwd_finalize_body(n, N, H * W, C, G, gridDim.x, part, rstd, gamma, beta, ss, zss, dss, dzss, c12, pgb, amax0);
  if (dgamma == nullptr) return;
  if (!last_block_of(ticket + N, (unsigned)N)) return;
  for (int c = threadIdx.x; c < C; c += 256) gn_bwd_param_one(c, N, C, pgb, dgamma, dbeta, acc_param);
}

// dx = a*dv - c1 - (x-mu)*c2 (+ add, resampled like dA); split into the two concat sources.  Streaming form as gn_apply_stream.
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(Src2 s, int N, int H, int W, int C, int chunk, const float* __restrict__ coef,
                                                           const float* __restrict__ c12, const float* __restrict__ dA, int act, int mode,
                                                           float drop_p, unsigned long long seed, unsigned long long offset,
                                                           const float* __restrict__ add, float* __restrict__ dx0, int acc0,
                                                           float* __restrict__ dx1, int acc1, unsigned* __restrict__ amax0,
                                                           const float* __restrict__ pgb, float* __restrict__ dgamma, float* __restrict__ dbeta, int acc_param) {
  __shared__ float wmax[4];
  // the gamma / beta gradients (sum over samples of the finalize kernel's per-sample contributions) ride in this launch: the blocks of
  // sample 0 take 256 channels each first -- one launch less per GroupNorm backward (53 per FFHQ-128 step)
  if (dgamma && blockIdx.y == 0 && (int)(blockIdx.x * 256) < C) {
    const int cpar = blockIdx.x * 256 + threadIdx.x;
    if (cpar < C) gn_bwd_param_one(cpar, N, C, pgb, dgamma, dbeta, acc_param);
  }
  const int n = blockIdx.y, NQ = C >> 2, PL = 256 / NQ, HW = H * W;
  const int t = threadIdx.x, q = t % NQ, pl = t / NQ, c = q * 4;
  float* dbase = nullptr; int accf = 0, Cd = 0, cd = 0;
  if (pl < PL) {
    if (c < s.C0) { dbase = dx0; accf = acc0; Cd = s.C0; cd = c; }
    else { dbase = dx1; accf = acc1; Cd = s.C1; cd = c - s.C0; }
  }
  float vmax = 0.f;                               // max |value written to dx0| by this thread (the dY scale of the next gradient convs)
  const bool track = amax0 != nullptr && dbase == dx0;
  if (dbase) {
  const size_t NC = (size_t)N * C;
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const float4 mu = *reinterpret_cast<const float4*>(coef + (size_t)n * C + c);
  const float4 a = *reinterpret_cast<const float4*>(coef + NC + (size_t)n * C + c);
  const float4 b = *reinterpret_cast<const float4*>(coef + 2 * NC + (size_t)n * C + c);
  const float* cp = c12 + ((size_t)n * C + c) * 2;
  const float4 ca = *reinterpret_cast<const float4*>(cp), cb = *reinterpret_cast<const float4*>(cp + 4);   // c1,c2 interleaved
  const int p0 = blockIdx.x * chunk, p1 = min(HW, p0 + chunk);
  for (int p = p0 + pl; p < p1; p += UNR * PL) {
    float4 x[UNR], da[UNR], ad[UNR], ex[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int pp = p + u * PL;
      if (pp < p1) {
        const int py = pp / W, px = pp - py * W; const size_t pix = (size_t)n * HW + pp;
        x[u] = ld4(s, pix, c); da[u] = fetch_da(dA, mode, n, py, px, H, W, C, c);
        if (add) ad[u] = fetch_da(add, mode, n, py, px, H, W, C, c);
        if (accf) ex[u] = *reinterpret_cast<const float4*>(dbase + pix * Cd + cd);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int pp = p + u * PL;
      if (pp >= p1) break;
      const size_t pix = (size_t)n * HW + pp;
      float4 xm = make_float4(x[u].x - mu.x, x[u].y - mu.y, x[u].z - mu.z, x[u].w - mu.w);
      float4 dv = compute_dv(da[u], xm, a, b, act, drop_p, dscale, seed, offset, pix * NQ + q);
      float4 o;
      o.x = a.x * dv.x - ca.x - xm.x * ca.y;
      o.y = a.y * dv.y - ca.z - xm.y * ca.w;
      o.z = a.z * dv.z - cb.x - xm.z * cb.y;
      o.w = a.w * dv.w - cb.z - xm.w * cb.w;
      if (add) { o.x += ad[u].x; o.y += ad[u].y; o.z += ad[u].z; o.w += ad[u].w; }
      if (accf) { o.x += ex[u].x; o.y += ex[u].y; o.z += ex[u].z; o.w += ex[u].w; }
      *reinterpret_cast<float4*>(dbase + pix * Cd + cd) = o;
      if (track) vmax = fmaxf(fmaxf(vmax, fabsf(o.x)), fmaxf(fmaxf(fabsf(o.y), fabsf(o.z)), fabsf(o.w)));
    }
  }
  }
  if (amax0) {                                    // block-wide max -> one atomic (non-negative floats order like their bit patterns)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    if ((t & 63) == 0) wmax[t >> 6] = vmax;
    __syncthreads();
    if (t == 0) atomicMax(amax0, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
  }
}

// ----------------------------------------------------------------------------------------------
// host launchers
// ----------------------------------------------------------------------------------------------
static int stats_chunks(int HW, int C) {
  long long el = (long long)HW * C;
  int S = (int)(el / 16384); if (S < 1) S = 1; if (S > 64) S = 64; if (S > HW) S = HW;
  return S;
}

// pixel chunks per sample of the streaming apply kernels: ~64 KB of the tensor per block, enough blocks to fill the chip
static int stream_chunks(int HW, int C) {
  long long el = (long long)HW * C;
  int S = (int)(el / 16384); if (S < 1) S = 1; if (S > HW) S = HW;
  return S;
}

static int check_c(int C0, int C1, int G) {
  int C = C0 + C1;
  PDAE_CHECK_ARG(C > 0 && C <= 1024 && (C % G) == 0 && (C0 % 4) == 0 && (C1 % 4) == 0 && G <= 64,
                 "groupnorm: need C0%%4==0, C1%%4==0, C<=1024, C%%G==0 (got C0=%d C1=%d G=%d)", C0, C1, G);
  return PDAE_OK;
}

size_t k_gn_workspace_floats(int N, int C) { return (size_t)N * 64 * C * 2 + (size_t)N * C * 4; }

int k_gn_stats(const float* x0, int C0, const float* x1, int C1, int N, int HW, int G, float eps, float* mean, float* rstd,
               float* ws, hipStream_t st) {
  if (int e = check_c(C0, C1, G)) return e;
  const int C = C0 + C1;
  Src2 s{x0, x1, C0, C1};
  int S = stats_chunks(HW, C), chunk = cdiv(HW, S);
  S = cdiv(HW, chunk);
  hipLaunchKernelGGL(gn_stats_partial_kernel, dim3(S, N), dim3(256), 0, st, s, HW, C, G, chunk, ws);
  hipLaunchKernelGGL(gn_stats_finalize_kernel, dim3(cdiv(N * G, 4)), dim3(256), 0, st, s, N, HW, C, G, S, eps, ws, mean, rstd);
  return pdae_launch_status("gn_stats");
}

int k_gn_coef_from_conv_stats(int N, int HW, int C0, int C1, int G, float eps, const float* part0, int tpi0, const float* part1, int tpi1,
                              const float* gamma, const float* beta, const float* ss, const float* zss, float* mean, float* rstd, float* coef,
                              hipStream_t st) {
  hipLaunchKernelGGL(gn_coef_from_conv_stats_kernel, dim3(N * ((G + 7) / 8)), dim3(256), 0, st, N, HW, C0, C1, G, eps, reinterpret_cast<const float2*>(part0), tpi0,
                     reinterpret_cast<const float2*>(part1), tpi1, gamma, beta, ss, zss, mean, rstd, coef);
  return pdae_launch_status("gn_coef_from_conv_stats");
}

int k_gn_stats_quads(const float* x, int N, int HW, int C, int tpi, float* part, hipStream_t st) {
  PDAE_CHECK_ARG(x && part && N > 0 && HW > 0 && C >= 4 && C <= 1024 && (C & 3) == 0 && tpi > 0 && tpi <= HW, "gn_stats_quads: need C %% 4 == 0, C <= 1024, 0 < tpi <= HW");
  hipLaunchKernelGGL(gn_stats_quads_kernel, dim3(tpi, N), dim3(256), 0, st, x, HW, C, cdiv(HW, tpi), reinterpret_cast<float2*>(part));
  return pdae_launch_status("gn_stats_quads");
}

int k_gn_stats_coef(const float* x0, int C0, const float* x1, int C1, int N, int HW, int G, float eps, const float* gamma, const float* beta,
                    const float* ss, const float* zss, float* mean, float* rstd, float* coef, float* ws, hipStream_t st, unsigned* ticket) {
  if (int e = check_c(C0, C1, G)) return e;
  const int C = C0 + C1;
  Src2 s{x0, x1, C0, C1};
  int S = stats_chunks(HW, C), chunk = cdiv(HW, S);
  S = cdiv(HW, chunk);
  if (ticket) {
    hipLaunchKernelGGL(gn_stats_coef_fused_kernel, dim3(S, N), dim3(256), 0, st, s, N, HW, C, G, chunk, eps, ws, gamma, beta, ss, zss, mean, rstd, coef, ticket);
    return pdae_launch_status("gn_stats_coef");
  }
  hipLaunchKernelGGL(gn_stats_partial_kernel, dim3(S, N), dim3(256), 0, st, s, HW, C, G, chunk, ws);
  hipLaunchKernelGGL(gn_finalize_coef_kernel, dim3(N), dim3(256), 0, st, s, N, HW, C, G, S, eps, ws, gamma, beta, ss, zss, mean, rstd, coef);
  return pdae_launch_status("gn_stats_coef");
}

int k_gn_coef(int N, int C, int G, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* ss,
              const float* zss, float* coef, hipStream_t st) {
  hipLaunchKernelGGL(gn_coef_kernel, dim3(cdiv((long long)N * C, 256)), dim3(256), 0, st, N, C, G, mean, rstd, gamma, beta, ss, zss, coef);
  return pdae_launch_status("gn_coef");
}

static int ew_grid(size_t total) { size_t b = (total + 255) / 256; if (b > 8192) b = 8192; if (b < 1) b = 1; return (int)b; }

int k_gn_apply(const float* x0, int C0, const float* x1, int C1, int N, int H, int W, const float* coef, int act, int mode, float* y,
               float* xpool, float drop_p, unsigned long long seed, unsigned long long offset, hipStream_t st) {
  if (int e = check_c(C0, C1, 1)) return e;
  PDAE_CHECK_ARG(mode == 0 || (mode == 1 && (H % 2) == 0 && (W % 2) == 0 && drop_p == 0.f), "gn_apply: bad mode/shape");
  const int C = C0 + C1;
  Src2 s{x0, x1, C0, C1};
  if (mode == 0 && xpool == nullptr) {
    const int HW = H * W;
    int S = stream_chunks(HW, C), chunk = cdiv(HW, S);
    S = cdiv(HW, chunk);
    hipLaunchKernelGGL(gn_apply_stream_kernel, dim3(S, N), dim3(256), 0, st, s, HW, C, N, chunk, coef, act, y, drop_p, seed, offset);
    return pdae_launch_status("gn_apply");
  }
  size_t total = (size_t)N * (mode ? (H / 2) * (W / 2) : H * W) * (C / 4);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(ew_grid(total)), dim3(256), 0, st, s, N, H, W, C, coef, act, mode, y, xpool, drop_p, seed, offset);
  return pdae_launch_status("gn_apply");
}

int k_gn_bwd(const float* x0, int C0, const float* x1, int C1, int N, int H, int W, int G, const float* coef, const float* rstd,
             const float* gamma, const float* beta, const float* ss, const float* zss, const float* dA, int act, int mode, float drop_p,
             unsigned long long seed, unsigned long long offset, const float* add, float* dx0, int acc0, float* dx1, int acc1,
             float* dgamma, float* dbeta, int acc_param, float* dss, float* dzss, float* ws, hipStream_t st, float* dx0_amax, unsigned* ticket,
             const float* parts, int parts_tiles) {
  if (int e = check_c(C0, C1, G)) return e;
  const int C = C0 + C1, HW = H * W;
  unsigned* am = (dx0 && dx0_amax) ? reinterpret_cast<unsigned*>(dx0_amax) : nullptr;
  Src2 s{x0, x1, C0, C1};
  int S = stats_chunks(HW, C), chunk = cdiv(HW, S);
  S = cdiv(HW, chunk);
  // parts: the two per-(n, c) sums were left by the data gradient that wrote dA (conv3x3y GB epilogue), [N][parts_tiles][C][2] in the layout of the
  // reduce kernel: the reduction pass over (x, dA) does not run
  if (parts) S = parts_tiles;
  float* part = parts ? const_cast<float*>(parts) : ws;   // [N][S][C][2]
  float* c12 = ws + (size_t)N * 64 * C * 2;           // [N][C][2]
  float* pgb = c12 + (size_t)N * C * 2;               // [N][C][2]
  if (ticket) {
    hipLaunchKernelGGL(gn_bwd_reduce_fused_kernel, dim3(S, N), dim3(256), 0, st, s, H, W, C, G, chunk, coef, N, dA, act, mode, drop_p, seed, offset, part,
                       rstd, gamma, beta, ss, zss, dss, dzss, c12, pgb, am, dgamma, dbeta, acc_param, ticket);
  } else {
    if (!parts) hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(S, N), dim3(256), 0, st, s, H, W, C, chunk, coef, N, dA, act, mode, drop_p, seed, offset, part);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(N), dim3(256), 0, st, N, HW, C, G, S, part, rstd, gamma, beta, ss, zss, dss, dzss, c12, pgb, am);
  }
  int Sa = stream_chunks(HW, C), chunk_a = cdiv(HW, Sa);
  Sa = cdiv(HW, chunk_a);
  // parameter gradients: inside the apply launch when it has enough blocks per sample to cover C in slices of 256, else their own launch
  const bool ride = !ticket && dgamma && (dx0 || dx1) && Sa * 256 >= C;
  if (!ticket && dgamma && !ride)
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, N, C, pgb, dgamma, dbeta, acc_param);
  if (dx0 || dx1) {
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(Sa, N), dim3(256), 0, st, s, N, H, W, C, chunk_a, coef, c12, dA, act, mode, drop_p, seed, offset,
                       add, dx0, acc0, dx1, acc1, am, pgb, ride ? dgamma : nullptr, dbeta, acc_param);
  }
  return pdae_launch_status("gn_bwd");
}
