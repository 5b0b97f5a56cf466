// Fused self-attention core of AttentionBlock (model/module.py:422-428) for gfx950: QK^T -> softmax -> PV in ONE kernel, and its backward in
// two, with the T x T probability matrix living only in LDS / registers (the unfused form writes P, and dP in the backward, to HBM and
// needs 3 launches forward + 6 backward).
//
// Reference ops replaced (ckczzj/PDAE):
//   QKVAttentionLegacy.forward   model/module.py:431-457   per head [q|k|v] channel blocks, q and k each scaled by ch^-1/4, softmax over keys
//   QKVAttention.forward         model/module.py:460-488   [q(all heads)|k|v] channel order (use_new_attention_order)
//   and their autograd.
//
// Shapes on this path: T = H*W in {64, 256}, head width ch in {32 .. 512} (F128: 1 head x 384 @16x16 and x 512 @8x8; encoders: 4 heads x 64 @16x16).
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact fp32 (the attention GEMMs are 0.3 % of the step's FLOPs, SURVEY 8d: no reason to split operands).
//
// Block = 256 threads (4 waves) = one 64-row tile of one (image, head):
//   forward       rows = queries:  S[64][T] = Q_t K^T (channels streamed through LDS in chunks of 32)  ->  row softmax in LDS (+ log-sum-exp saved)
//                                  ->  O[64][ch] = P V (V streamed in chunks of 64 channels)
//   backward dq   rows = queries:  P = exp(s*QK^T - lse), dP = dO_t V^T, dS = P o (dP - D)  ->  dQ = s * dS K
//   backward dkv  rows = keys:     P^T = exp(s*K_t Q^T - lse), dP^T = V_t dO^T  ->  dV = P^T dO ;  dS^T = P^T o (dP^T - D)  ->  dK = s * dS^T Q
//   with D[q] = sum_c dO[q][c] O[q][c] (one small kernel).  Everything is recomputed from q, k, v, lse: nothing of size T x T is ever stored.
// MFMA operand mapping (32x32x2): lane l supplies A[row = l%32][k = l/32] and B[k = l/32][col = l%32]; accumulator register r of lane l is
// D[(r&3) + 8*(r>>2) + 4*(l/32)][l%32].
#include "common.h"
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define AT_ROWS 64          // rows of the block tile
#define AT_KC 32            // channels per staging chunk of the "NT" products (A B^T)
#define AT_LDK 34           // LDS row stride of those chunks (floats): 34/2 = 17 is odd -> the 32 lanes of a ds_read_b64 hit 32 distinct bank pairs
#define AT_VC 64            // channels per staging chunk of the "NN" products (M B)
#define AT_LDV 65           // LDS row stride of that chunk

struct AttnParams {
  const float* qkv; const float* o; const float* d_o; const float* lse_in; const float* dvec;     // inputs (backward ones may be NULL in forward)
  float* out; float* lse_out; float* dqkv;                                                        // outputs
  int N, T, C, heads, ch;          // C = heads * ch
  int oq, ok, ov, hs;              // channel offsets of q / k / v inside a 3C row and the per-head stride
  float scale2;                    // 1 / sqrt(ch) = (ch^-1/4)^2
};

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// acc[j] += A_tile[64][ch] * B[T][ch]^T for this wave's 32 rows and its NB 32-column blocks.  A, B: global, row strides lda / ldb (floats).
// sm: >= (64 + T) * AT_LDK floats.  All 256 threads must call it.
template <int NB>
__device__ __forceinline__ void at_gemm_nt(const float* __restrict__ A, long long lda, const float* __restrict__ B, long long ldb, int ch, int T,
                                           f32x16 (&acc)[NB], float* sm) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, h = lane >> 5;
  float* sa = sm; float* sb = sm + AT_ROWS * AT_LDK;
  const int row0 = (wv & 1) * 32, colb = (wv >> 1) * NB;
  // the next chunk's rows travel global -> registers under this chunk's MFMAs (the load -> LDS -> barrier sequence used to sit exposed in
  // front of every chunk: these kernels run one block of four waves per CU, nothing else hides it)
  constexpr int NL = 2 + 2 * NB;                       // float4 per thread and chunk: (64 + T) rows x 8 quads / 256 threads
  float4 pre[NL];
  auto gl = [&](int c0) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int i = t + 256 * l, r = i >> 3, q = (i & 7) * 4;
      pre[l] = r < AT_ROWS ? *reinterpret_cast<const float4*>(A + (long long)r * lda + c0 + q)
                           : *reinterpret_cast<const float4*>(B + (long long)(r - AT_ROWS) * ldb + c0 + q);
    }
  };
  gl(0);
  for (int c0 = 0; c0 < ch; c0 += AT_KC) {
    __syncthreads();                                   // the previous chunk has been consumed
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int i = t + 256 * l, r = i >> 3, q = (i & 7) * 4;
      float* d = sm + r * AT_LDK + q;                  // 8-byte aligned (AT_LDK even): two ds_write_b64
      *reinterpret_cast<float2*>(d) = make_float2(pre[l].x, pre[l].y);
      *reinterpret_cast<float2*>(d + 2) = make_float2(pre[l].z, pre[l].w);
    }
    __syncthreads();
    if (c0 + AT_KC < ch) gl(c0 + AT_KC);
#pragma unroll
    for (int m = 0; m < AT_KC / 4; ++m) {              // 4 channels per iteration = 2 MFMAs: k set {4m+2h, 4m+2h+1}
      const float2 a = *reinterpret_cast<const float2*>(sa + (row0 + li) * AT_LDK + 4 * m + 2 * h);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const float2 b = *reinterpret_cast<const float2*>(sb + ((colb + j) * 32 + li) * AT_LDK + 4 * m + 2 * h);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[j], 0, 0, 0);
      }
    }
  }
}

// dst[64 rows][ch] (global, row stride ldd) = alpha * M[64][T] (LDS, row stride T + 1) * B[T][ch] (global, row stride ldb).
// sv: >= T * AT_LDV floats.  All 256 threads must call it; M must be complete (caller syncs).
template <int NB>
__device__ __forceinline__ void at_gemm_nn(const float* sM, const float* __restrict__ B, long long ldb, int ch, int T, float alpha,
                                           float* __restrict__ dst, long long ldd, float* sv) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, h = lane >> 5;
  const int row0 = (wv & 1) * 32, colw = (wv >> 1) * 32, ldm = T + 1;
  constexpr int NL = 4 * NB;                           // float4 per thread and chunk: T rows x 16 quads / 256 threads
  float4 pre[NL];
  const int pq = (t & 15) * 4;                         // this thread's channel quad inside a chunk; rows (t >> 4) + 16 l
  auto gl = [&](int c0) {                              // next chunk's rows: global -> registers under this chunk's MFMAs (see at_gemm_nt)
    const int cq = c0 + pq < ch ? c0 + pq : 0;         // quads beyond the last channel load a valid address and are never stored to LDS
#pragma unroll
    for (int l = 0; l < NL; ++l) pre[l] = *reinterpret_cast<const float4*>(B + (long long)((t >> 4) + 16 * l) * ldb + cq);
  };
  gl(0);
  for (int c0 = 0; c0 < ch; c0 += AT_VC) {
    const int cw = min(AT_VC, ch - c0);                // 32 or 64 valid channels
    __syncthreads();
    if (pq < cw) {
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        float* d = sv + ((t >> 4) + 16 * l) * AT_LDV + pq;
        d[0] = pre[l].x; d[1] = pre[l].y; d[2] = pre[l].z; d[3] = pre[l].w;
      }
    }
    __syncthreads();
    if (c0 + AT_VC < ch) gl(c0 + AT_VC);
    if (colw < cw) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      for (int kk = 0; kk < T / 2; ++kk) {
        const float a = sM[(row0 + li) * ldm + 2 * kk + h];
        const float b = sv[(2 * kk + h) * AT_LDV + colw + li];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(long long)(row0 + acc_row(r, h)) * ldd + c0 + colw + li] = alpha * acc[r];
    }
  }
}

template <int NB> __device__ __forceinline__ void at_zero(f32x16 (&acc)[NB]) {
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
}

// LDS: [ M: 64 x (T+1) | staging: max((64+T) * AT_LDK, T * AT_LDV) | vec: 2 x max(64, T) ]
__host__ __device__ inline size_t at_smem_floats(int T) {
  const size_t st = (size_t)(AT_ROWS + T) * AT_LDK > (size_t)T * AT_LDV ? (size_t)(AT_ROWS + T) * AT_LDK : (size_t)T * AT_LDV;
  return (size_t)AT_ROWS * (T + 1) + st + 2 * (size_t)(T > AT_ROWS ? T : AT_ROWS);
}

// ------------------------------------------------------------------------------------------------ forward
template <int NB>      // NB = T / 64
__global__ void __launch_bounds__(256) attn_fwd_kernel(const AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  const int T = NB * 64, ldm = T + 1;
  float* sM = smf; float* st = smf + AT_ROWS * ldm;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * AT_ROWS, hd = blockIdx.y, n = blockIdx.z;
  const long long ld3 = 3ll * P.C;
  const float* base = P.qkv + (long long)n * T * ld3 + hd * P.hs;
  f32x16 acc[NB];
  at_zero<NB>(acc);
  at_gemm_nt<NB>(base + P.oq + (long long)q0 * ld3, ld3, base + P.ok, ld3, P.ch, T, acc, st);
  const int row0 = (wv & 1) * 32, colb = (wv >> 1) * NB;
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) sM[(row0 + acc_row(r, h)) * ldm + (colb + j) * 32 + li] = acc[j][r] * P.scale2;
  __syncthreads();
  for (int row = wv * 16; row < wv * 16 + 16; ++row) {           // softmax over keys, one wave per row (module.py:455)
    float mx = -3.0e38f;
    for (int c = lane; c < T; c += 64) mx = fmaxf(mx, sM[row * ldm + c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int c = lane; c < T; c += 64) { const float e = expf(sM[row * ldm + c] - mx); sM[row * ldm + c] = e; sum += e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int c = lane; c < T; c += 64) sM[row * ldm + c] *= inv;
    if (lane == 0 && P.lse_out) P.lse_out[((long long)n * P.heads + hd) * T + q0 + row] = mx + logf(sum);
  }
  __syncthreads();
  at_gemm_nn<NB>(sM, base + P.ov, ld3, P.ch, T, 1.0f, P.out + ((long long)n * T + q0) * P.C + hd * P.ch, P.C, st);
}

// ------------------------------------------------------------------------------------------------ backward
// D[n][head][q] = sum_c dO[q][c] * O[q][c]: one wave per (n, q), heads looped
__global__ void __launch_bounds__(256) attn_dvec_kernel(const AttnParams P) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);       // n * T + q
  if (row >= (long long)P.N * P.T) return;
  const long long n = row / P.T; const int q = (int)(row - n * P.T);
  for (int hd = 0; hd < P.heads; ++hd) {
    float s = 0.f;
    for (int c = lane; c < P.ch; c += 64) s += P.d_o[row * P.C + hd * P.ch + c] * P.o[row * P.C + hd * P.ch + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) const_cast<float*>(P.dvec)[(n * P.heads + hd) * P.T + q] = s;
  }
}

// P (or P^T) and dS (or dS^T) of this wave's accumulator blocks, in place:  p = exp(s * scale2 - lse), ds = p * (dp - D).
// by_row: lse / D are indexed by the accumulator ROW (query-row tiles) else by its COLUMN (key-row tiles: columns are queries).
template <int NB> __device__ __forceinline__ void at_p_ds(f32x16 (&s)[NB], f32x16 (&dp)[NB], const float* lse, const float* dv, bool by_row, int row0,
                                                          int colb, float scale2, int li, int h) {
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int idx = by_row ? row0 + acc_row(r, h) : (colb + j) * 32 + li;
      const float p = expf(s[j][r] * scale2 - lse[idx]);
      s[j][r] = p;
      dp[j][r] = p * (dp[j][r] - dv[idx]);
    }
}

template <int NB> __device__ __forceinline__ void at_store_tile(float* sM, int ldm, const f32x16 (&a)[NB], int row0, int colb, int li, int h) {
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) sM[(row0 + acc_row(r, h)) * ldm + (colb + j) * 32 + li] = a[j][r];
}

template <int NB>
__global__ void __launch_bounds__(256) attn_bwd_q_kernel(const AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  const int T = NB * 64, ldm = T + 1;
  float* sM = smf; float* st = smf + AT_ROWS * ldm;
  float* svec = st + ((size_t)(AT_ROWS + T) * AT_LDK > (size_t)T * AT_LDV ? (size_t)(AT_ROWS + T) * AT_LDK : (size_t)T * AT_LDV);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * AT_ROWS, hd = blockIdx.y, n = blockIdx.z;
  const long long ld3 = 3ll * P.C, nh = (long long)n * P.heads + hd;
  const float* base = P.qkv + (long long)n * T * ld3 + hd * P.hs;
  const float* dO = P.d_o + ((long long)n * T + q0) * P.C + hd * P.ch;
  if (t < AT_ROWS) { svec[t] = P.lse_in[nh * T + q0 + t]; svec[AT_ROWS + t] = P.dvec[nh * T + q0 + t]; }
  f32x16 s[NB], dp[NB];
  at_zero<NB>(s); at_zero<NB>(dp);
  at_gemm_nt<NB>(base + P.oq + (long long)q0 * ld3, ld3, base + P.ok, ld3, P.ch, T, s, st);       // S = Q_t K^T
  at_gemm_nt<NB>(dO, P.C, base + P.ov, ld3, P.ch, T, dp, st);                                       // dP = dO_t V^T
  const int row0 = (wv & 1) * 32, colb = (wv >> 1) * NB;
  at_p_ds<NB>(s, dp, svec, svec + AT_ROWS, true, row0, colb, P.scale2, li, h);
  at_store_tile<NB>(sM, ldm, dp, row0, colb, li, h);                                                // dS
  __syncthreads();
  at_gemm_nn<NB>(sM, base + P.ok, ld3, P.ch, T, P.scale2, P.dqkv + ((long long)n * T + q0) * ld3 + hd * P.hs + P.oq, ld3, st);   // dQ = s dS K
}

template <int NB>
__global__ void __launch_bounds__(256) attn_bwd_kv_kernel(const AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  const int T = NB * 64, ldm = T + 1;
  float* sM = smf; float* st = smf + AT_ROWS * ldm;
  float* svec = st + ((size_t)(AT_ROWS + T) * AT_LDK > (size_t)T * AT_LDV ? (size_t)(AT_ROWS + T) * AT_LDK : (size_t)T * AT_LDV);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, h = lane >> 5;
  const int k0 = blockIdx.x * AT_ROWS, hd = blockIdx.y, n = blockIdx.z;
  const long long ld3 = 3ll * P.C, nh = (long long)n * P.heads + hd;
  const float* base = P.qkv + (long long)n * T * ld3 + hd * P.hs;
  const float* dO = P.d_o + (long long)n * T * P.C + hd * P.ch;
  for (int i = t; i < T; i += 256) { svec[i] = P.lse_in[nh * T + i]; svec[T + i] = P.dvec[nh * T + i]; }
  f32x16 s[NB], dp[NB];
  at_zero<NB>(s); at_zero<NB>(dp);
  at_gemm_nt<NB>(base + P.ok + (long long)k0 * ld3, ld3, base + P.oq, ld3, P.ch, T, s, st);        // S^T = K_t Q^T
  at_gemm_nt<NB>(base + P.ov + (long long)k0 * ld3, ld3, dO, P.C, P.ch, T, dp, st);                 // dP^T = V_t dO^T
  const int row0 = (wv & 1) * 32, colb = (wv >> 1) * NB;
  at_p_ds<NB>(s, dp, svec, svec + T, false, row0, colb, P.scale2, li, h);
  float* dq = P.dqkv + ((long long)n * T + k0) * ld3 + hd * P.hs;
  at_store_tile<NB>(sM, ldm, s, row0, colb, li, h);                                                 // P^T
  __syncthreads();
  at_gemm_nn<NB>(sM, dO, P.C, P.ch, T, 1.0f, dq + P.ov, ld3, st);                                       // dV = P^T dO
  __syncthreads();
  at_store_tile<NB>(sM, ldm, dp, row0, colb, li, h);                                                // dS^T
  __syncthreads();
  at_gemm_nn<NB>(sM, base + P.oq, ld3, P.ch, T, P.scale2, dq + P.ok, ld3, st);                          // dK = s dS^T Q
}

// ------------------------------------------------------------------------------------------------ host
bool attn_fused_ok(int T, int ch, int C, int heads) {
  return (T == 64 || T == 128 || T == 192 || T == 256) && ch >= 32 && (ch % 32) == 0 && C == ch * heads && (C % 4) == 0;
}

static void at_fill(AttnParams& P, const float* qkv, int N, int T, int C, int heads, int new_order) {
  P.qkv = qkv; P.N = N; P.T = T; P.C = C; P.heads = heads; P.ch = C / heads;
  if (new_order) { P.oq = 0; P.ok = C; P.ov = 2 * C; P.hs = P.ch; }             // [q(all heads) | k | v]          (module.py:470-476)
  else { P.oq = 0; P.ok = P.ch; P.ov = 2 * P.ch; P.hs = 3 * P.ch; }             // per head [q | k | v]           (module.py:447-449)
  P.scale2 = 1.0f / sqrtf((float)P.ch);
  P.o = P.d_o = P.lse_in = P.dvec = nullptr; P.out = P.lse_out = P.dqkv = nullptr;
}

template <typename K> static int at_attr(K kern, size_t smem, const char* what) {
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) { pdae_set_error("%s: cannot raise dynamic LDS to %zu: %s", what, smem, hipGetErrorString(e)); return (int)e; }
  return PDAE_OK;
}

#define AT_DISPATCH(KERN, GRID, SMEM, ST, P)                                                   \
  switch ((P).T / 64) {                                                                        \
    case 1: if (int e = at_attr(KERN<1>, SMEM, #KERN)) return e; hipLaunchKernelGGL(KERN<1>, GRID, dim3(256), SMEM, ST, P); break; \
    case 2: if (int e = at_attr(KERN<2>, SMEM, #KERN)) return e; hipLaunchKernelGGL(KERN<2>, GRID, dim3(256), SMEM, ST, P); break; \
    case 3: if (int e = at_attr(KERN<3>, SMEM, #KERN)) return e; hipLaunchKernelGGL(KERN<3>, GRID, dim3(256), SMEM, ST, P); break; \
    default: if (int e = at_attr(KERN<4>, SMEM, #KERN)) return e; hipLaunchKernelGGL(KERN<4>, GRID, dim3(256), SMEM, ST, P); break; \
  }

int k_attn_fwd(const float* qkv, int N, int T, int C, int heads, int new_order, float* out, float* lse, hipStream_t st) {
  AttnParams P; at_fill(P, qkv, N, T, C, heads, new_order);
  P.out = out; P.lse_out = lse;
  const size_t smem = at_smem_floats(T) * sizeof(float);
  const dim3 grid(T / AT_ROWS, heads, N);
  AT_DISPATCH(attn_fwd_kernel, grid, smem, st, P)
  return pdae_launch_status("attn_fwd");
}

int k_attn_bwd(const float* qkv, const float* o, const float* lse, const float* d_o, int N, int T, int C, int heads, int new_order, float* dqkv,
               float* dvec, hipStream_t st) {
  AttnParams P; at_fill(P, qkv, N, T, C, heads, new_order);
  P.o = o; P.d_o = d_o; P.lse_in = lse; P.dvec = dvec; P.dqkv = dqkv;
  hipLaunchKernelGGL(attn_dvec_kernel, dim3(cdiv((long long)N * T, 4)), dim3(256), 0, st, P);
  const size_t smem = at_smem_floats(T) * sizeof(float);
  const dim3 grid(T / AT_ROWS, heads, N);
  AT_DISPATCH(attn_bwd_q_kernel, grid, smem, st, P)
  AT_DISPATCH(attn_bwd_kv_kernel, grid, smem, st, P)
  return pdae_launch_status("attn_bwd");
}
