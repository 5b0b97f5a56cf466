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
// Arithmetic (round 3): 3 bf16 planes x 6 products on v_mfma_f32_32x32x16_bf16, fp32 accumulate -- the range-free fp32-grade format of the
// 1x1 kernels, the same in forward and backward (gradients need no scale).  The exact-fp32 v_mfma_f32_32x32x2_f32 form this replaces was
// MFMA-bound at the fp32 matrix rate on HALF the chip: T / 64 x images = 128 workgroups at 16x16, 49 us of MFMA each (forward 82 us,
// backward 200 us per layer; 1.9 ms of a 58 ms step for 0.3 % of its FLOPs).
//
// Block = 256 threads (4 waves) = one 64-row tile of one (image, head); wave (a, b): rows a*32..+31, half b of the columns.
//   forward       rows = queries:  S[64][T] = Q_t K^T in registers -> row softmax in registers (row max / sum across the lanes of a row and,
//                                  through 1 KB of LDS, the two column halves; log-sum-exp saved) -> P as bf16 planes in LDS -> O[64][ch] = P V
//   backward dq   rows = queries:  P = exp(s*QK^T - lse), dP = dO_t V^T, dS = P o (dP - D)  ->  dQ = s * dS K
//   backward dkv  rows = keys:     P^T = exp(s*K_t Q^T - lse), dP^T = V_t dO^T  ->  dV = P^T dO ;  dS^T = P^T o (dP^T - D)  ->  dK = s * dS^T Q
//   with D[q] = sum_c dO[q][c] O[q][c] (one small kernel).  Everything is recomputed from q, k, v, lse: nothing of size T x T is ever stored.
//   "NT" products (A B^T, both operands channel-contiguous): 32-channel chunks of both operands through LDS as bf16 planes (at_gemm_nt);
//   "NN" products (M B, B token-major): M = the 64 x T probability / dS tile, three bf16 planes in LDS ([plane][64][T + 8]: 528-byte rows,
//        conflict-free ds_read_b128), written once from the accumulators; B fragments straight from global memory -- lane (channel, 8
//        consecutive tokens) = eight 4-byte loads, the 32 lanes of a half wave covering one 128-byte line each -- split in registers and
//        prefetched one whole 64-channel chunk ahead: no barrier in the whole product.
// MFMA operand mapping (32x32x16): lane l supplies A[row = l%32][k = (l/32)*8 + 0..7] and B[k = (l/32)*8 + 0..7][col = l%32]; accumulator
// register r of lane l is D[(r&3) + 8*(r>>2) + 4*(l/32)][l%32].
#include "common.h"
#include "kernels.h"
#include "conv3x3p.h"        // f32x16, bf16x8, p_split2

#define AT_ROWS 64          // rows of the block tile
#define AT_PAD 8            // bf16 of padding per M row

struct AttnParams {
  const float* qkv; const float* o; const float* d_o; const float* lse_in; const float* dvec;     // inputs (backward ones may be NULL in forward)
  float* out; float* lse_out; float* dqkv;                                                        // outputs
  int N, T, C, heads, ch;          // C = heads * ch
  int oq, ok, ov, hs;              // channel offsets of q / k / v inside a 3C row and the per-head stride
  float scale2;                    // 1 / sqrt(ch) = (ch^-1/4)^2
};

typedef unsigned at_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// eight fp32 values (k-contiguous) -> the three bf16 planes of one MFMA fragment
__device__ __forceinline__ void at_split8(float e0, float e1, float e2, float e3, float e4, float e5, float e6, float e7, bf16x8 (&f)[3]) {
  unsigned w0[3], w1[3], w2[3], w3[3];
  p_split2<3>(e0, e1, w0); p_split2<3>(e2, e3, w1); p_split2<3>(e4, e5, w2); p_split2<3>(e6, e7, w3);
#pragma unroll
  for (int p = 0; p < 3; ++p) { const at_u32x4 v = {w0[p], w1[p], w2[p], w3[p]}; f[p] = __builtin_bit_cast(bf16x8, v); }
}
__device__ __forceinline__ void at_mma6(const bf16x8 (&A)[3], const bf16x8 (&B)[3], f32x16& acc) {       // small terms first
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], acc, 0, 0, 0);
}

// acc[j] += A[64 rows of the block][ch] * B[T][ch]^T for this wave's 32 rows (row0) and its NB 32-column blocks (colb ..).  A, B: global, row
// strides lda / ldb (floats).  Chunks of 32 channels: 256 threads load (64 + T) rows x 128 bytes (8 lanes per row: whole cache lines -- loading
// MFMA fragments straight from global memory, 16 bytes per lane from 64 different lines per instruction, ran into the L1 tag rate), split each
// value ONCE into the three bf16 planes and store them with 80-byte rows (conflict-free ds_read_b128 fragments).  Two LDS buffers: the planes of
// chunk c+1 are written and chunk c+2 is loaded in the same barrier interval as the MFMAs of chunk c (one barrier per chunk).
// st: 2 x 3 x (64 + T) x AT_SROW bf16.  All 256 threads must call it; ends with a barrier.
#define AT_SROW 40
template <int NB, int T, int NTHR>      // NB: 32-column blocks of this wave; T: all columns; NTHR: threads of the block
__device__ __forceinline__ void at_gemm_nt(const float* __restrict__ A, long long lda, const float* __restrict__ B, long long ldb, int ch,
                                           f32x16 (&acc)[NB], unsigned short* st, int row0, int colb, int li, int h) {
  constexpr int R = AT_ROWS + T, PL = R * AT_SROW, BUF = 3 * PL, NL = R * 8 / NTHR;            // NL float4 per thread and chunk
  const int t = threadIdx.x;
#ifdef PDAE_AT_PROBE_NONT
  return;
#endif
  float4 pre[2][NL];                                   // chunks c+1 (being split / stored) and c+2 (in flight); c+3 is loaded into the slot c+1 frees
  auto gl = [&](int c0, float4 (&q)[NL]) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int i = t + NTHR * l, r = i >> 3, qd = (i & 7) * 4;
      q[l] = r < AT_ROWS ? *reinterpret_cast<const float4*>(A + (long long)r * lda + c0 + qd)
                         : *reinterpret_cast<const float4*>(B + (long long)(r - AT_ROWS) * ldb + c0 + qd);
    }
  };
  auto sw = [&](unsigned short* buf, const float4 (&q)[NL]) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int i = t + NTHR * l, r = i >> 3, qd = (i & 7) * 4;
      unsigned w0[3], w1[3];
      p_split2<3>(q[l].x, q[l].y, w0); p_split2<3>(q[l].z, q[l].w, w1);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(buf + p * PL + r * AT_SROW + qd) = make_uint2(w0[p], w1[p]);
    }
  };
  const int nch = ch >> 5, lastc = nch - 1;
  gl(0, pre[0]); sw(st, pre[0]);
  gl(min(1, lastc) * 32, pre[1]); gl(min(2, lastc) * 32, pre[0]);
  __syncthreads();
  // One chunk = one scheduling region: the 48 (NB = 4) MFMAs of chunk c carry the operand split + LDS stores of chunk c+1 and the global loads
  // of chunk c+3 between them; loads are consumed TWO chunk times after their issue (one chunk time, ~1 us, left every chunk waiting on L2 /
  // HBM latency: 2.3 us per chunk for 0.8 us of MFMA).  Branch-free: the last chunks reload / restage a valid chunk that nobody reads.
  auto chunk = [&](int c, float4 (&qa)[NL]) {          // qa: holds chunk c+1 on entry, chunk c+3 (in flight) on exit
    const unsigned short* cur = st + (c & 1) * BUF;
    __builtin_amdgcn_sched_barrier(0);
    sw(st + ((c + 1) & 1) * BUF, qa);
    gl(min(c + 3, lastc) * 32, qa);
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      bf16x8 Af[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) Af[p] = *reinterpret_cast<const bf16x8*>(cur + p * PL + (row0 + li) * AT_SROW + (kc * 2 + h) * 8);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        bf16x8 Bf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) Bf[p] = *reinterpret_cast<const bf16x8*>(cur + p * PL + (AT_ROWS + (colb + j) * 32 + li) * AT_SROW + (kc * 2 + h) * 8);
        at_mma6(Af, Bf, acc[j]);
      }
    }
#ifndef PDAE_AT_PROBE_NOSCHED
    __builtin_amdgcn_sched_group_barrier(0x100, 3 + 3 * NB, 0);                  // fragments of the first k-step
#pragma unroll
    for (int i = 0; i < 12 * NB; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x006, 7, 0);
      if (i < 3 + 3 * NB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // second k-step's fragments under the first one's MFMAs
      if (i < NL) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
#endif
    __syncthreads();
  };
  for (int c = 0; c < nch; c += 2) { chunk(c, pre[1]); if (c + 1 < nch) chunk(c + 1, pre[0]); }
}

// dst[this wave's 32 rows][its 32 of every 64 channels] = alpha * M[64][T] (bf16 planes in LDS) * B[T][ch] (global, token-major, row stride ldb).
// sM must be complete (caller syncs); no barrier inside.  dst already offset to the wave's first row, B / dst to its first channel (colw).
template <int NB, int CG, int DR = NB * 4>   // the block's waves cover CG * 32 channels per chunk; DR: k-steps of B in flight (divides NB * 4)
__device__ __forceinline__ void at_gemm_nn(const unsigned short* sM, const float* __restrict__ B, long long ldb, int ch, int colw, float alpha,
                                           float* __restrict__ dst, long long ldd, int row0, int li, int h) {
  constexpr int CW = 32 * CG;                          // channels per chunk
#ifdef PDAE_AT_PROBE_NONN
  return;
#endif
  // ring of D k-steps in flight; default one whole chunk (2 us of work at T = 256): D = 4 left the product waiting on L2 latency.  The dV product of
  // attn_bwd_kv<4, 4> runs with D = 8: dS^T (32 registers) stays live across it and the 16-deep ring spilled 45 VGPRs to scratch (184 bytes per lane)
  constexpr int T = NB * 64, KS = NB * 4, LDM = T + AT_PAD, D = DR;
  static_assert(KS % D == 0, "the ring slot of a k-step must be a compile-time constant");
  const int myn = (ch - colw + CW - 1) / CW;            // chunks in which this wave's 32 channels exist
  // buffer loads: one lane offset in a VGPR, everything else ((token, chunk) -> bytes) in the scalar offset -- 64-bit per-load addresses in
  // VGPRs were hoisted out of the chunk loop by the compiler, 256 registers of them
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, 0x7fffffff, 0x00020000);
  const int lane_off = (int)(((long long)(h * 8) * ldb + li) * 4), ldb4 = (int)(ldb * 4);
  const __amdgpu_buffer_rsrc_t drd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0x7fffffff, 0x00020000);
  const int dlane_off = (int)(((long long)(h * 4) * ldd + li) * 4), ldd4 = (int)(ldd * 4);
  const unsigned short* pm = sM + (row0 + li) * LDM + h * 8;
  float rb[D][8];
  auto load = [&](int c, int ks, float (&q)[8]) {
    const int so = ks * 16 * ldb4 + c * CW * 4;
#pragma unroll
#ifdef PDAE_AT_PROBE_NNNOLOAD
    for (int j = 0; j < 8; ++j) q[j] = (float)(so + j);
#else
    for (int j = 0; j < 8; ++j) q[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, lane_off, so + j * ldb4, 0));
#endif
  };
  if (myn <= 0) return;
#pragma unroll
  for (int i = 0; i < D; ++i) load(0, i, rb[i]);
  // fragments are double buffered: step k's six MFMAs run with the LDS reads, the operand split and the refill loads of step k+1 between them
  bf16x8 Af[2][3], Bf[2][3];
  const int last = myn - 1;
  // fragments of k-step ks of chunk cb; its ring slot is refilled with the step D further on (same chunk, or the next one -- beyond the last chunk
  // the last one is reloaded and never consumed)
  auto prep = [&](int cb, int ks, bf16x8 (&A)[3], bf16x8 (&Bv)[3]) {             // branch-free (one basic block with the MFMAs around it)
#pragma unroll
    for (int p = 0; p < 3; ++p) A[p] = *reinterpret_cast<const bf16x8*>(pm + p * AT_ROWS * LDM + ks * 16);
    float (&q)[8] = rb[ks % D];
    at_split8(q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], Bv);
    if (ks + D < KS) load(cb, ks + D, q);
    else load(min(cb + 1, last), ks + D - KS, q);
  };
  prep(0, 0, Af[0], Bf[0]);
#pragma unroll 1
  for (int c = 0; c < myn; ++c) {
    const int c1 = min(c + 1, last);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) prep(c, ks + 1, Af[(ks + 1) & 1], Bf[(ks + 1) & 1]);
      else prep(c1, 0, Af[0], Bf[0]);
#ifdef PDAE_AT_PROBE_NNNOMMA
      acc[0] += (float)Af[ks & 1][0][0] + (float)Bf[ks & 1][0][0] + (float)Af[ks & 1][1][1] + (float)Bf[ks & 1][1][1] + (float)Af[ks & 1][2][2] + (float)Bf[ks & 1][2][2];
#else
      at_mma6(Af[ks & 1], Bf[ks & 1], acc);
#endif
#ifndef PDAE_AT_PROBE_NOSCHED
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 11, 0);
        if (i < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (i < 4) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    // buffer stores: one lane offset in a VGPR, (row of the register, chunk) in the scalar offset -- sixteen 64-bit per-lane addresses were hoisted
    // out of the chunk loop otherwise
#pragma unroll
    for (int r = 0; r < 16; ++r)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, alpha * acc[r]), drd, dlane_off, (((r & 3) + 8 * (r >> 2)) * ldd4) + c * CW * 4, 0);
  }
}

template <int NB> __device__ __forceinline__ void at_zero(f32x16 (&acc)[NB]) {
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
}

// this wave's accumulator blocks -> the three bf16 planes of M (2-byte stores: a lane holds one column of 16 rows)
template <int NB, int T> __device__ __forceinline__ void at_store_planes(unsigned short* sM, const f32x16 (&a)[NB], int row0, int colb, int li, int h) {
  constexpr int LDM = T + AT_PAD;
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = a[j][r], hi = p_trunc(v), r1 = v - hi, mid = p_trunc(r1), lo = r1 - mid;
      unsigned short* d = sM + (row0 + acc_row(r, h)) * LDM + (colb + j) * 32 + li;
      d[0] = (unsigned short)(__float_as_uint(hi) >> 16);
      d[AT_ROWS * LDM] = (unsigned short)(__float_as_uint(mid) >> 16);
      d[2 * AT_ROWS * LDM] = (unsigned short)(__float_as_uint(lo) >> 16);
    }
}

// LDS: [ max(M planes: 3 x 64 x (T + 8) bf16, NT staging: 2 x 3 x (64 + T) x 40 bf16) | row exchange: 2 x 4 x 64 floats | vec: 2 x max(64, T) floats ]
__host__ __device__ inline size_t at_main_bytes(int T) {
  const size_t m = (size_t)3 * AT_ROWS * (T + AT_PAD) * 2, st = (size_t)2 * 3 * (AT_ROWS + T) * AT_SROW * 2;
  return m > st ? m : st;
}
__host__ __device__ inline size_t at_smem_bytes(int T) { return at_main_bytes(T) + 8 * AT_ROWS * 4 + 2 * (size_t)(T > AT_ROWS ? T : AT_ROWS) * 4; }

// ------------------------------------------------------------------------------------------------ forward
template <int NB, int CG>      // NB = T / 64; CG column groups: waves = 2 row halves x CG (8 waves when T % 128 == 0: two per SIMD overlap each other's
                               // split / LDS / MFMA phases; left to one wave per SIMD those phases ran strictly one after the other)
__global__ void __launch_bounds__(128 * CG) attn_fwd_kernel(const AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned short sm16[];
  constexpr int T = NB * 64, LDM = T + AT_PAD, NBW = 2 * NB / CG, NTHR = 128 * CG;
  unsigned short* sM = sm16;
  float* sx = reinterpret_cast<float*>(reinterpret_cast<char*>(sm16) + at_main_bytes(T));          // [CG column groups][64 rows] x {max, sum}
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * AT_ROWS, hd = blockIdx.y, n = blockIdx.z;
  const long long ld3 = 3ll * P.C;
  const float* base = P.qkv + (long long)n * T * ld3 + hd * P.hs;
  const int row0 = (wv & 1) * 32, cg = wv >> 1, colb = cg * NBW;
  f32x16 acc[NBW];
  at_zero<NBW>(acc);
  at_gemm_nt<NBW, T, NTHR>(base + P.oq + (long long)q0 * ld3, ld3, base + P.ok, ld3, P.ch, acc, sM, row0, colb, li, h);
  // softmax over keys (module.py:455): a row lives in the 32 lanes of one half wave x NBW registers x the CG column-group waves
  float mx[16], sum[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float m = -3.0e38f;
#pragma unroll
    for (int j = 0; j < NBW; ++j) { acc[j][r] *= P.scale2; m = fmaxf(m, acc[j][r]); }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    mx[r] = m;
    if (li == 0) sx[cg * AT_ROWS + row0 + acc_row(r, h)] = m;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float m = sx[row0 + acc_row(r, h)];
#pragma unroll
    for (int g = 1; g < CG; ++g) m = fmaxf(m, sx[g * AT_ROWS + row0 + acc_row(r, h)]);
    mx[r] = m;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NBW; ++j) { const float e = expf(acc[j][r] - m); acc[j][r] = e; s += e; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
    sum[r] = s;
    if (li == 0) sx[(4 + cg) * AT_ROWS + row0 + acc_row(r, h)] = s;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float tot = sx[4 * AT_ROWS + row0 + acc_row(r, h)];                       // group 0 + group 1 + ...: the same order in every wave
#pragma unroll
    for (int g = 1; g < CG; ++g) tot += sx[(4 + g) * AT_ROWS + row0 + acc_row(r, h)];
    const float inv = 1.0f / tot;
#pragma unroll
    for (int j = 0; j < NBW; ++j) acc[j][r] *= inv;
    if (cg == 0 && li == 0 && P.lse_out) P.lse_out[((long long)n * P.heads + hd) * T + q0 + row0 + acc_row(r, h)] = mx[r] + logf(tot);
  }
  at_store_planes<NBW, T>(sM, acc, row0, colb, li, h);
  __syncthreads();
  at_gemm_nn<NB, CG>(sM, base + P.ov + cg * 32, ld3, P.ch, cg * 32, 1.0f, P.out + ((long long)n * T + q0 + row0) * P.C + hd * P.ch + cg * 32, P.C, row0, li, h);
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

template <int NB, int CG>
__global__ void __launch_bounds__(128 * CG) attn_bwd_q_kernel(const AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned short sm16[];
  constexpr int T = NB * 64, NBW = 2 * NB / CG, NTHR = 128 * CG;
  unsigned short* sM = sm16;
  float* svec = reinterpret_cast<float*>(reinterpret_cast<char*>(sm16) + at_main_bytes(T)) + 8 * AT_ROWS;
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * AT_ROWS, hd = blockIdx.y, n = blockIdx.z;
  const long long ld3 = 3ll * P.C, nh = (long long)n * P.heads + hd;
  const float* base = P.qkv + (long long)n * T * ld3 + hd * P.hs;
  const int row0 = (wv & 1) * 32, cg = wv >> 1, colb = cg * NBW;
  const float* dO = P.d_o + ((long long)n * T + q0) * P.C + hd * P.ch;
  if (t < AT_ROWS) { svec[t] = P.lse_in[nh * T + q0 + t]; svec[AT_ROWS + t] = P.dvec[nh * T + q0 + t]; }
  f32x16 s[NBW], dp[NBW];
  at_zero<NBW>(s); at_zero<NBW>(dp);
  at_gemm_nt<NBW, T, NTHR>(base + P.oq + (long long)q0 * ld3, ld3, base + P.ok, ld3, P.ch, s, sM, row0, colb, li, h);     // S = Q_t K^T
  at_gemm_nt<NBW, T, NTHR>(dO, P.C, base + P.ov, ld3, P.ch, dp, sM, row0, colb, li, h);                                       // dP = dO_t V^T
  at_p_ds<NBW>(s, dp, svec, svec + AT_ROWS, true, row0, colb, P.scale2, li, h);
  at_store_planes<NBW, T>(sM, dp, row0, colb, li, h);                                               // dS
  __syncthreads();
  at_gemm_nn<NB, CG>(sM, base + P.ok + cg * 32, ld3, P.ch, cg * 32, P.scale2,
                     P.dqkv + ((long long)n * T + q0 + row0) * ld3 + hd * P.hs + P.oq + cg * 32, ld3, row0, li, h);                    // dQ = s dS K
}

template <int NB, int CG>
__global__ void __launch_bounds__(128 * CG) attn_bwd_kv_kernel(const AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned short sm16[];
  constexpr int T = NB * 64, NBW = 2 * NB / CG, NTHR = 128 * CG;
  unsigned short* sM = sm16;
  float* svec = reinterpret_cast<float*>(reinterpret_cast<char*>(sm16) + at_main_bytes(T)) + 8 * AT_ROWS;
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, h = lane >> 5;
  const int k0 = blockIdx.x * AT_ROWS, hd = blockIdx.y, n = blockIdx.z;
  const long long ld3 = 3ll * P.C, nh = (long long)n * P.heads + hd;
  const float* base = P.qkv + (long long)n * T * ld3 + hd * P.hs;
  const float* dO = P.d_o + (long long)n * T * P.C + hd * P.ch;
  const int row0 = (wv & 1) * 32, cg = wv >> 1, colb = cg * NBW;
  for (int i = t; i < T; i += NTHR) { svec[i] = P.lse_in[nh * T + i]; svec[T + i] = P.dvec[nh * T + i]; }
  f32x16 s[NBW], dp[NBW];
  at_zero<NBW>(s); at_zero<NBW>(dp);
  at_gemm_nt<NBW, T, NTHR>(base + P.ok + (long long)k0 * ld3, ld3, base + P.oq, ld3, P.ch, s, sM, row0, colb, li, h);     // S^T = K_t Q^T
  at_gemm_nt<NBW, T, NTHR>(base + P.ov + (long long)k0 * ld3, ld3, dO, P.C, P.ch, dp, sM, row0, colb, li, h);              // dP^T = V_t dO^T
  at_p_ds<NBW>(s, dp, svec, svec + T, false, row0, colb, P.scale2, li, h);
  float* dq = P.dqkv + ((long long)n * T + k0 + row0) * ld3 + hd * P.hs + cg * 32;
  at_store_planes<NBW, T>(sM, s, row0, colb, li, h);                                                // P^T
  __syncthreads();
  at_gemm_nn<NB, CG, (NB == 4 ? 8 : NB * 4)>(sM, dO + cg * 32, P.C, P.ch, cg * 32, 1.0f, dq + P.ov, ld3, row0, li, h);       // dV = P^T dO
  __syncthreads();
  at_store_planes<NBW, T>(sM, dp, row0, colb, li, h);                                               // dS^T
  __syncthreads();
  at_gemm_nn<NB, CG>(sM, base + P.oq + cg * 32, ld3, P.ch, cg * 32, P.scale2, dq + P.ok, ld3, row0, li, h);   // dK = s dS^T Q
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

#define AT_LAUNCH(KERN, NB_, CG_, GRID, SMEM, ST, P) \
  { if (int e = at_attr(KERN<NB_, CG_>, SMEM, #KERN)) return e; hipLaunchKernelGGL((KERN<NB_, CG_>), GRID, dim3(128 * CG_), SMEM, ST, P); }
#define AT_DISPATCH(KERN, GRID, SMEM, ST, P)                                                   \
  switch ((P).T / 64) {                                                                        \
    case 1: AT_LAUNCH(KERN, 1, 2, GRID, SMEM, ST, P) break;                                    \
    case 2: AT_LAUNCH(KERN, 2, 4, GRID, SMEM, ST, P) break;                                    \
    case 3: AT_LAUNCH(KERN, 3, 2, GRID, SMEM, ST, P) break;                                    \
    default: AT_LAUNCH(KERN, 4, 4, GRID, SMEM, ST, P) break;                                   \
  }

int k_attn_fwd(const float* qkv, int N, int T, int C, int heads, int new_order, float* out, float* lse, hipStream_t st) {
  AttnParams P; at_fill(P, qkv, N, T, C, heads, new_order);
  P.out = out; P.lse_out = lse;
  const size_t smem = at_smem_bytes(T);
  const dim3 grid(T / AT_ROWS, heads, N);
  AT_DISPATCH(attn_fwd_kernel, grid, smem, st, P)
  return pdae_launch_status("attn_fwd");
}

int k_attn_bwd(const float* qkv, const float* o, const float* lse, const float* d_o, int N, int T, int C, int heads, int new_order, float* dqkv,
               float* dvec, hipStream_t st) {
  AttnParams P; at_fill(P, qkv, N, T, C, heads, new_order);
  P.o = o; P.d_o = d_o; P.lse_in = lse; P.dvec = dvec; P.dqkv = dqkv;
  hipLaunchKernelGGL(attn_dvec_kernel, dim3(cdiv((long long)N * T, 4)), dim3(256), 0, st, P);
  const size_t smem = at_smem_bytes(T);
  const dim3 grid(T / AT_ROWS, heads, N);
  AT_DISPATCH(attn_bwd_q_kernel, grid, smem, st, P)
  AT_DISPATCH(attn_bwd_kv_kernel, grid, smem, st, P)
  return pdae_launch_status("attn_bwd");
}
