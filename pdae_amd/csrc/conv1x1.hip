// 1x1 convolution (forward and data gradient) for gfx950 on the bf16 MFMA pipe with split fp32 operands.
//
// In NHWC a 1x1 convolution is the GEMM  Y[M = pixels][Nout] = X[M][C] * W[Nout][C]^T.  Block = 256 threads (2x2 waves),
// tile = 128 pixels x 128 output channels.  Per stage of 64 input channels the 128 x 64 activation tile is loaded with fully
// coalesced float4 loads (16 lanes per pixel row), split into bf16 planes and staged in LDS (144-byte rows: every
// ds_read_b128 of a 16-lane group is conflict-free); the next stage's loads are in flight under the MFMAs.  B fragments come
// straight from L2 out of the fragment-ordered prepared weights (conv1x1_wprep, same idea as conv3x3p_wprep), so the only
// barriers are the two per stage that hand the activation tile over.  The torch.cat of the decoder skip (unet.py:200) is two
// base pointers.  (A row-per-lane "register-direct" A load was measured first: 32 different cache lines per load
// instruction make it TA-bound, 45 TFLOP/s; the LDS transpose costs less.)
//
// Replaces F.conv2d(k=1) / conv1d(k=1) of model/module.py:276,412,420 (ResBlock skip, attention qkv / proj) and their dX.
#include <stdlib.h>

#include "common.h"
#include "igemm.h"
#include "conv3x3p.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// NS = 4: two fp16 planes, 3 products (see conv3x3p.hip).  Activations enter unscaled (1x1 inputs may be the raw residual stream:
// window up to 6e4) or, as gradients, with the per-tensor power-of-two scale from pdae_amax; the weights carry 2^4 * 2^ceil(log2(sqrt(C)))
#define QNPL(NS_) ((NS_) == 4 ? 2 : (NS_))

__device__ __forceinline__ float q_trunc(float a) { return __uint_as_float(__float_as_uint(a) & 0xffff0000u); }
__device__ __forceinline__ unsigned q_hi16(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }
__device__ __forceinline__ unsigned q_rn(float a, float b) {
  unsigned short x = __builtin_bit_cast(unsigned short, (__bf16)a), y = __builtin_bit_cast(unsigned short, (__bf16)b);
  return (unsigned)x | ((unsigned)y << 16);
}
__device__ __forceinline__ float q_pow2_scale(float amax) {            // 2^(10 - floor(log2(amax))); 1 for 0 / non-finite
  const int ex = (__float_as_int(amax) >> 23) & 0xff;
  if (ex == 0 || ex == 255) return 1.0f;
  int sb = 127 + 10 - (ex - 127);
  sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
  return __int_as_float(sb << 23);
}
template <int NS> __device__ __forceinline__ void q_split2(float e0, float e1, unsigned (&w)[QNPL(NS)], float sc = 1.0f) {
  if constexpr (NS == 4) {          // fp16 planes of e * sc (the other formats take no scale)
    pdae_f16_split2s(e0, e1, sc, w[0], w[1]);
  } else if constexpr (NS == 1) { w[0] = q_rn(e0, e1); }
  else {
    float h0 = q_trunc(e0), h1 = q_trunc(e1);
    float r0 = e0 - h0, r1 = e1 - h1;
    w[0] = q_hi16(h0, h1);
    if constexpr (NS == 2) { w[1] = q_rn(r0, r1); }
    else {
      float m0 = q_trunc(r0), m1 = q_trunc(r1);
      w[1] = q_hi16(m0, m1);
      w[2] = q_hi16(r0 - m0, r1 - m1);
    }
  }
}

struct PointParams {
  const float* x0; const float* x1; int C0, C1;     // A = [x0 | x1] rows of M pixels
  long long M; int C;                                // C = C0 + C1 (multiple of 16)
  const unsigned short* wp; int NT;                  // prepared weights [NS][C/16][NT][64][8], NT 32-channel tiles in it
  int nt_off, Nout;                                  // output channels = prepared rows nt_off*32 .. + Nout
  float* y; const float* bias; const float* res; int accumulate;
  float woscale; const float* amax;                  // fp16 format: 1 / weight scale; device scalar max|input| (gradients) or NULL
  unsigned int* sat;                                 // fp16 format: saturation counter (common.h) or NULL
  int res_mode, H, W;                                // res_mode 2: residual stored at half resolution (needs the image geometry)
  int tiles_m, tiles_n, splits, sps;                 // split-K: `splits` ranges of `sps` 16-channel steps
  float* slab;
  int rot;                                           // 1: workgroup (m-tile i) walks its 64-channel stages starting from stage i mod (stages), see the kernel
};

// pixel row of the half-resolution residual (x_upd(x) skip of an up-ResBlock, module.py:279-284,297)
__device__ __forceinline__ long long half_row(long long row, int H, int W) {
  const int ox = (int)(row % W); const long long t2 = row / W; const int oy = (int)(t2 % H); const long long im = t2 / H;
  return (im * (H >> 1) + (oy >> 1)) * (W >> 1) + (ox >> 1);
}

#define QLDH 72                                          // bf16 per LDS row: 64 channels + 8 pad = 144 bytes
template <int NS, bool GEN>
__global__ void __launch_bounds__(256) conv1x1_kernel(const PointParams P) {
  constexpr int SA_PLANES = QNPL(NS) * 128 * QLDH, SA_EPI = 4 * 32 * 36 * 2;        // operand planes | epilogue transpose tiles (4 waves x 32 x 36 fp32)
  __shared__ __attribute__((aligned(16))) unsigned short sA[SA_PLANES > SA_EPI ? SA_PLANES : SA_EPI];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, h = lane >> 5;   // wave id in an SGPR: everything derived from it is scalar
  // Wave tile = ALL 128 pixels x 32 output channels (round 5; it was 64 px x 64 ch): every weight fragment is fetched by ONE wave and used for four
  // products.  The 64 x 64 tile had pairs of waves fetch identical fragments: 64 + 32 one-KB vector-memory instructions per 64-channel stage and
  // workgroup at ~35 cycles of the CU's vector-memory path each (tools/micro/unit_pipe.hip) -- more than half of the ~6900 cycles a stage took on
  // the 128^2 skip convolutions (MFMA busy 0.17, 3.7 TB/s: neither roof).  Now 32 + 32; the activation fragments (LDS) double instead.
  const int wn = wv;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  int tid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int tn_i = tid % P.tiles_n; tid /= P.tiles_n;
  const int tm_i = tid % P.tiles_m; const int sp = tid / P.tiles_m;
  const long long m0 = (long long)tm_i * 128;
  const int n0 = tn_i * 128 + wn * 32;
  const int nsteps = P.C >> 4;
  const int s_begin = sp * P.sps, s_end = min(nsteps, s_begin + P.sps);      // 16-channel steps; sps is a multiple of 2

  const float ascale = NS == 4 ? (P.amax ? q_pow2_scale(*P.amax) : 1.0f) : 1.0f;
  // ---- A staging: thread -> (pixel row = idx >> 4, channel quad = idx & 15), 8 float4 per thread and stage
  float4 apre0[8];
  float sat_hit = 0.f;                            // fp16 format: lanes with a value clamped into the window (common.h)
  // Buffer loads (round 5, GEN = false): the lane's offset inside a source (row t >> 4 of the tile, channel quad t & 15) is computed ONCE; the
  // m-tile, the stage's channel offset and the 16-row step between a lane's eight loads are scalar offsets; rows beyond M and channel quads
  // beyond the range get an out-of-range lane offset (the scalar offset is not relied on for the range check) and return zeros; the source
  // is a scalar select of the descriptor.  ONE straight-line sequence of eight loads whatever the stage: the compiler's wait counts stay exact
  // (see the weight ring below).  Needs every 64-channel stage inside one source (C0 % 64 == 0 or C1 == 0) and M * C_src * 4 < 4 GiB;
  // GEN = true is the pointer form for everything else (64-bit address arithmetic and a predicated branch per load: 346 of the 830
  // instructions of a stage).
  constexpr unsigned G_OOB = 0xfffffff0u, G_ALL = 0xffffffefu;
  const unsigned g_row = (unsigned)(t >> 4), g_qd = (unsigned)(t & 15);
  const unsigned g_v0 = g_row * (unsigned)P.C0 * 4u + g_qd * 16u, g_v1 = g_row * (unsigned)P.C1 * 4u + g_qd * 16u;
  const unsigned g_t0 = (unsigned)m0 * (unsigned)P.C0 * 4u, g_t1 = (unsigned)m0 * (unsigned)P.C1 * 4u;      // (m0 < M: below 4 GiB)
  const int g_lim = (int)min((long long)128, P.M - m0) - (int)g_row;         // the lane's load l (row g_row + 16 l) exists iff 16 l < g_lim
  auto a_gload = [&](float4 (&apre)[8], int s0, bool dead) {       // stage starting at step s0 (4 steps = 64 channels, fewer at the tail); dead: all zeros, no traffic
    if constexpr (!GEN) {
      const int c0 = s0 << 4, cend = min(s_end << 4, c0 + 64);
      const unsigned live = dead ? 0u : (unsigned)(cend - c0) >> 2;           // channel quads of this stage that exist
      const bool second = c0 >= P.C0;
      const float* base = second ? P.x1 : P.x0;
      const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)G_ALL, 0x00020000);
      const unsigned vo = g_qd < live ? (second ? g_v1 : g_v0) : G_OOB;
      const unsigned so = second ? g_t1 + (unsigned)(c0 - P.C0) * 4u : g_t0 + (unsigned)c0 * 4u;
      const unsigned rs = (unsigned)(second ? P.C1 : P.C0) * 64u;            // 16 rows
#pragma unroll
      for (int l = 0; l < 8; ++l) apre[l] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)(16 * l < g_lim ? vo : G_OOB), (int)(so + l * rs), 0));
    } else {
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        const int idx = t + 256 * l, row = idx >> 4, qd = idx & 15;
        const int c = (s0 << 4) + qd * 4;
        const long long p = m0 + row;
        const bool ok = p < P.M && c < (s_end << 4) && !dead;
        const long long pp = ok ? p : 0;               // unconditional load from a clamped address, zeroed afterwards
        const int cq = ok ? c : 0;
        const bool first = cq < P.C0;
        const float4 v = *reinterpret_cast<const float4*>(first ? P.x0 + pp * P.C0 + cq : P.x1 + pp * P.C1 + (cq - P.C0));
        apre[l] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto a_lstore = [&](float4 (&apre)[8]) {
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      const int idx = t + 256 * l, row = idx >> 4, qd = idx & 15;
      unsigned a[QNPL(NS)], b[QNPL(NS)];
#ifdef PDAE_C1_PROBE_NOCONV
      for (int p = 0; p < QNPL(NS); ++p) { a[p] = __builtin_bit_cast(unsigned, apre[l].x) + p; b[p] = __builtin_bit_cast(unsigned, apre[l].z) + p; }
#else
      if constexpr (NS == 4) pdae_f16_amax4(apre[l], ascale, sat_hit);
      q_split2<NS>(apre[l].x, apre[l].y, a, ascale);
      q_split2<NS>(apre[l].z, apre[l].w, b, ascale);
#endif
#pragma unroll
      for (int p = 0; p < QNPL(NS); ++p) *reinterpret_cast<uint2*>(&sA[(p * 128 + row) * QLDH + qd * 4]) = make_uint2(a[p], b[p]);
    }
  };

  const int nt_end = P.nt_off + ((P.Nout + 31) >> 5);
  const int nt0 = min(P.nt_off + (n0 >> 5), nt_end - 1);          // tiles beyond the last are clamped onto it: those columns are never stored
  const size_t plane_stride = (size_t)nsteps * P.NT * 512;
  auto ldb = [&](uint4 (&bq)[QNPL(NS)], int s) {
    const unsigned short* base = P.wp + ((size_t)s * P.NT + nt0) * 512 + lane * 8;
#ifdef PDAE_C1_PROBE_NOB
    if (s != s_begin) return;
#endif
#pragma unroll
    for (int p = 0; p < QNPL(NS); ++p) bq[p] = *reinterpret_cast<const uint4*>(base + p * plane_stride);       // branch-free (clamped tile): counted vmcnt waits
  };

  f32x16 acc[4];                                  // m-tiles a = pixel rows 32 a .. 32 a + 31
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  // Stage order.  All workgroups of a launch run in near lock step, and stage j of every one of them reads the SAME 256-byte slice of its pixel
  // rows (channels 64 j .. 64 j + 63 of C0- / C1-channel rows): at any moment the chip reads every second (C = 128) or every fourth 256-byte chunk
  // of the tensor, i.e. a fraction of the HBM channels -- the 128^2 skip convolutions sat at 3.6 - 3.8 TB/s whatever the prefetch depth (a persistent
  // form with two stages in flight across tiles measured 0.85 - 1.03x, tools/c1_ab.py).  With P.rot the workgroup of m-tile i starts at stage
  // i mod (stages) and wraps: neighbouring workgroups read different slices at the same time.  The accumulation order of a tile then depends on
  // its index (deterministic; fp32 sums of the same products in another order).
  const int nst = (s_end - s_begin + 3) >> 2;
  const int rot = (P.rot && nst > 1) ? (int)(tm_i % nst) : 0;
  auto stage_first = [&](int j) { int st = j + rot; if (st >= nst) st -= nst; return s_begin + 4 * st; };      // first step of the j-th stage in walk order
  // Weight fragments: a ring of four steps (one whole stage).  Slot k is refilled with step k of the NEXT stage right after the products that
  // read it, so every fragment a stage waits for was requested before that stage's activation prefetch: vector-memory results return in
  // order, and the two-slot ring this replaces (fragment of step k+1 requested in step k, i.e. AFTER the prefetch of the next stage) made the
  // `s_waitcnt vmcnt(2)` of every step wait for the whole HBM prefetch in front of it -- the prefetch distance was one step, not one stage,
  // and the kernel streamed at 4 TB/s whatever else was changed (tools/c1_probe.py).
  auto step = [&](int ks, uint4 (&bq)[QNPL(NS)], int s_refill) {
#ifdef PDAE_C1_PROBE_NOMMA
    if (P.C >= 0) { acc[0][0] += __builtin_bit_cast(float, bq[0].x); ldb(bq, s_refill); return; }
#endif
    uint4 af[4][QNPL(NS)];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int p = 0; p < QNPL(NS); ++p)
        af[a][p] = *reinterpret_cast<const uint4*>(&sA[(p * 128 + a * 32 + li) * QLDH + ks * 16 + h * 8]);
    __builtin_amdgcn_sched_barrier(0);              // loads first, then the MFMA cluster (the scheduler would sink them behind it)
#define PDAE_A(P_) __builtin_bit_cast(bf16x8, af[a][P_])
#define PDAE_B(P_) __builtin_bit_cast(bf16x8, bq[P_])
#define PDAE_AH(P_) __builtin_bit_cast(f16x8, af[a][P_])
#define PDAE_BH(P_) __builtin_bit_cast(f16x8, bq[P_])
#define PDAE_EACH(STMT) _Pragma("unroll") for (int a = 0; a < 4; ++a) { STMT; }
    // product-major over the four accumulators: a dependent pair is four issues apart
    if constexpr (NS == 4) {
      PDAE_EACH(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_AH(0), PDAE_BH(1), acc[a], 0, 0, 0))
      PDAE_EACH(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_AH(1), PDAE_BH(0), acc[a], 0, 0, 0))
      PDAE_EACH(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(PDAE_AH(0), PDAE_BH(0), acc[a], 0, 0, 0))
    } else {
      if constexpr (NS == 3) {
        PDAE_EACH(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_A(1), PDAE_B(1), acc[a], 0, 0, 0))
        PDAE_EACH(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_A(0), PDAE_B(2), acc[a], 0, 0, 0))
        PDAE_EACH(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_A(2), PDAE_B(0), acc[a], 0, 0, 0))
      }
      if constexpr (NS >= 2) {
        PDAE_EACH(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_A(0), PDAE_B(1), acc[a], 0, 0, 0))
        PDAE_EACH(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_A(1), PDAE_B(0), acc[a], 0, 0, 0))
      }
      PDAE_EACH(acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PDAE_A(0), PDAE_B(0), acc[a], 0, 0, 0))
    }
#undef PDAE_EACH
#undef PDAE_A
#undef PDAE_AH
#undef PDAE_BH
#undef PDAE_B
    ldb(bq, s_refill);                              // always issued (straight-line loads: counted waits); beyond the range: a harmless reload
  };

  uint4 q0[QNPL(NS)], q1[QNPL(NS)], q2[QNPL(NS)], q3[QNPL(NS)];
  if (s_begin < s_end) {
    const int sl = s_end - 1;
    { const int f = stage_first(0); ldb(q0, f); ldb(q1, f + 1); ldb(q2, min(f + 2, sl)); ldb(q3, min(f + 3, sl)); }
    a_gload(apre0, stage_first(0), false);
    a_lstore(apre0);
    __syncthreads();
    a_gload(apre0, stage_first(nst > 1 ? 1 : 0), nst <= 1);
    // A stage = 4 steps.  At the tail of the channel range (2 steps: C % 32 == 0) the last two run on zero activations (a_gload) and a clamped
    // weight step.  The loop body is branch-free on purpose (the prefetch beyond the last stage is a dead load, the last stage's refills
    // harmless reloads): with a short stage's refills or the prefetch under a branch, the compiler's wait insertion takes the minimum over the
    // paths and the fragment waits of the common path drain the prefetch again (and once, with refill targets shared with fragment
    // temporaries of the other path, put an `s_waitcnt vmcnt(0)` into every stage).
    for (int j = 0; j + 1 < nst; ++j) {
      const int sn = stage_first(j + 1);
      step(0, q0, sn);
      step(1, q1, sn + 1);
      step(2, q2, min(sn + 2, sl));
      step(3, q3, min(sn + 3, sl));
      __syncthreads();                                 // tile hand-over
      a_lstore(apre0);
      __syncthreads();
      a_gload(apre0, stage_first(j + 2 < nst ? j + 2 : 0), j + 2 >= nst);
    }
    step(0, q0, sl); step(1, q1, sl); step(2, q2, sl); step(3, q3, sl);
  }
  // ---- epilogue: each wave transposes its four 32-pixel x 32-channel accumulators through a private LDS tile (the activation tile is dead by now)
  // so that global traffic is float4 per lane in 128-byte runs; residual / accumulate operands are loaded up front
  const float oscale = NS == 4 ? P.woscale / ascale : 1.0f;      // exact: powers of two
  if constexpr (NS == 4) pdae_sat_report(P.sat, sat_hit);
  __syncthreads();
  float* tw = reinterpret_cast<float*>(sA) + wv * (32 * 36);
  const int er = lane >> 3, ec = (lane & 7) * 4;
  const int colb = n0 + ec;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (P.bias && P.splits == 1 && colb < P.Nout) bias4 = *reinterpret_cast<const float4*>(P.bias + colb);      // Nout % 4 == 0
  const bool col_ok = colb < P.Nout;
  const unsigned lane_d = (unsigned)(er * P.Nout + (col_ok ? colb : 0));     // the lane's offset inside an 8-row group: the only per-lane address term
  const size_t row8 = (size_t)8 * P.Nout;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const long long rbase = m0 + a * 32;                                     // wave-uniform first row of this 32-row group
    float4 rv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) rv[it] = bias4;
#pragma unroll
    for (int r = 0; r < 16; ++r) tw[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + li] = acc[a][r];
    if (rbase + 32 <= P.M && P.res_mode != 2) {
      // whole group, same-resolution operands: scalar 64-bit bases + lane_d, no per-row index arithmetic (see conv3x3p.hip)
      const size_t eb = (size_t)rbase * P.Nout;
      if (P.splits == 1 && P.res) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const float4 u = *reinterpret_cast<const float4*>(P.res + eb + it * row8 + lane_d);
          rv[it].x += u.x; rv[it].y += u.y; rv[it].z += u.z; rv[it].w += u.w;
        }
      }
      if (P.splits == 1 && P.accumulate) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const float4 u = *reinterpret_cast<const float4*>(P.y + eb + it * row8 + lane_d);
          rv[it].x += u.x; rv[it].y += u.y; rv[it].z += u.z; rv[it].w += u.w;
        }
      }
      float* dst = P.splits > 1 ? P.slab + (size_t)sp * P.M * P.Nout + eb + lane_d : P.y + eb + lane_d;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        float4 v = *reinterpret_cast<const float4*>(&tw[(it * 8 + er) * 36 + ec]);
        // the operand scales are powers of two: scaling after the transpose, fused with the bias / residual add, is exact
        if (P.splits > 1) { v.x *= oscale; v.y *= oscale; v.z *= oscale; v.w *= oscale; }
        else { v.x = fmaf(v.x, oscale, rv[it].x); v.y = fmaf(v.y, oscale, rv[it].y); v.z = fmaf(v.z, oscale, rv[it].z); v.w = fmaf(v.w, oscale, rv[it].w); }
#ifdef PDAE_C1_PROBE_NOSTORE
        if (col_ok && v.x == 1.2345e37f) *reinterpret_cast<float4*>(dst + it * row8) = v;
#else
        if (col_ok) *reinterpret_cast<float4*>(dst + it * row8) = v;
#endif
      }
      continue;
    }
    // partial last group or half-resolution residual: per-row indices
    long long rowv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long long row = rbase + it * 8 + er;
      rowv[it] = (row >= P.M || !col_ok) ? -1 : row;
    }
    if (P.splits == 1 && P.res) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const long long row = rowv[it] < 0 ? 0 : rowv[it];
        const float4 u = *reinterpret_cast<const float4*>(P.res + (P.res_mode == 2 ? half_row(row, P.H, P.W) : row) * P.Nout + (col_ok ? colb : 0));
        rv[it].x += u.x; rv[it].y += u.y; rv[it].z += u.z; rv[it].w += u.w;
      }
    }
    if (P.splits == 1 && P.accumulate) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float4 u = *reinterpret_cast<const float4*>(P.y + (rowv[it] < 0 ? 0 : rowv[it]) * P.Nout + (col_ok ? colb : 0));
        rv[it].x += u.x; rv[it].y += u.y; rv[it].z += u.z; rv[it].w += u.w;
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float4 v = *reinterpret_cast<const float4*>(&tw[(it * 8 + er) * 36 + ec]);
      if (rowv[it] < 0) continue;
      if (P.splits > 1) { v.x *= oscale; v.y *= oscale; v.z *= oscale; v.w *= oscale; *reinterpret_cast<float4*>(P.slab + ((long long)sp * P.M + rowv[it]) * P.Nout + colb) = v; continue; }
      v.x = fmaf(v.x, oscale, rv[it].x); v.y = fmaf(v.y, oscale, rv[it].y); v.z = fmaf(v.z, oscale, rv[it].z); v.w = fmaf(v.w, oscale, rv[it].w);
      *reinterpret_cast<float4*>(P.y + rowv[it] * P.Nout + colb) = v;
    }
  }
}

// y = sum of the split slabs (fixed order) + bias + residual (+ y)
__global__ void __launch_bounds__(256) conv1x1_reduce_kernel(const PointParams P) {
  const int n4 = P.Nout >> 2;
  const long long total = P.M * n4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long row = i / n4; const int col = (int)(i - row * n4) * 4;
    float4 v = *reinterpret_cast<const float4*>(P.slab + row * P.Nout + col);
    const float* sl = P.slab + row * P.Nout + col; const long long ss = P.M * P.Nout;
    int k = 1;                                  // four slabs in flight, added in slab order (see splitk_reduce_kernel)
    for (; k + 4 <= P.splits; k += 4) {
      const float4 u0 = *reinterpret_cast<const float4*>(sl + k * ss), u1 = *reinterpret_cast<const float4*>(sl + (k + 1) * ss);
      const float4 u2 = *reinterpret_cast<const float4*>(sl + (k + 2) * ss), u3 = *reinterpret_cast<const float4*>(sl + (k + 3) * ss);
      v.x += u0.x; v.y += u0.y; v.z += u0.z; v.w += u0.w;
      v.x += u1.x; v.y += u1.y; v.z += u1.z; v.w += u1.w;
      v.x += u2.x; v.y += u2.y; v.z += u2.z; v.w += u2.w;
      v.x += u3.x; v.y += u3.y; v.z += u3.z; v.w += u3.w;
    }
    for (; k < P.splits; ++k) {
      const float4 u = *reinterpret_cast<const float4*>(sl + k * ss);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (P.bias) { const float4 u = *reinterpret_cast<const float4*>(P.bias + col); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    if (P.res) { const float4 u = *reinterpret_cast<const float4*>(P.res + (P.res_mode == 2 ? half_row(row, P.H, P.W) : row) * P.Nout + col); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    float4* dst = reinterpret_cast<float4*>(P.y + row * P.Nout + col);
    if (P.accumulate) { const float4 u = *dst; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    *dst = v;
  }
}

// weight preparation: GEMM weight w'[n][c] -> [NS][C/16][NT][64][8] bf16 planes in B-fragment order.
//   transposed = 0: w' = w [Nout][C];   transposed = 1 (data gradient): w'[n][c] = w[c][n], w stored [C][Nout]
template <int NS>
__global__ void __launch_bounds__(256) conv1x1_wprep_kernel(const float* __restrict__ w, int Nout, int C, int NT, int transposed, float wscale,
                                                            unsigned short* __restrict__ wp) {
  const size_t nslot = (size_t)(C >> 4) * NT * 64;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nslot; i += (size_t)gridDim.x * 256) wprep1_slot<NS>(w, Nout, C, NT, transposed, wscale, wp, i);
}

// ---------------------------------------------------------------------------------------------
struct PointPlan { int tiles_m, tiles_n, splits, sps; long long blocks; };
static PointPlan point_plan(long long M, int C, int Nout) {
  PointPlan q;
  q.tiles_m = (int)((M + 127) / 128); q.tiles_n = (Nout + 127) / 128;
  const long long base = (long long)q.tiles_m * q.tiles_n;
  const int nstage = (C + 63) >> 6;                      // 64-channel stages
  const double slab_w = (double)pdae_knob(KNOB_C1_SLAB);      // 0: the round-2 plan (A/B aid)
  long long best = -1; int best_s = 1;
  for (int sN = 1; sN <= nstage && sN <= 16; ++sN) {     // minimise rounds(grid) x stages-per-block (+1 for prologue / epilogue)
    const int per = (nstage + sN - 1) / sN, sp = (nstage + per - 1) / per;
    if (sp != sN) continue;
    const long long rounds = (base * sp + 511) / 512;
    // + the slabs: sp x M x Nout floats written and read again by the reduce launch, in stage times (~1.2 us) at ~5 TB/s.  Without this term
    // the 16x16 attention projections (M = 8192, Nout = 1152) were split in two: 150 MB of slab traffic around a 12 us GEMM
    const long long slab = sp > 1 ? (long long)(slab_w * 2.0 * sp * (double)M * Nout * 4.0 / 6.0e6) : 0;
    const long long cost = rounds * (per + 1) + slab;
    if (best < 0 || cost < best) { best = cost; best_s = sN; }
  }
  const int per = (nstage + best_s - 1) / best_s;
  q.sps = per * 4;                                       // 16-channel steps per split: whole stages
  q.splits = (nstage + per - 1) / per;
  q.blocks = base * q.splits;
  return q;
}

// eligibility: 1x1, stride 1, no pad / upsample, both sources multiples of 32 channels, Nout % 4
bool conv1x1_ok(int math, int KH, int KW, int stride, int pad, int up, int C0, int C1, int Nout) {
  if (math < 1 || KH != 1 || KW != 1 || stride != 1 || pad != 0 || up) return false;
  return (C0 & 31) == 0 && (C1 & 31) == 0 && (Nout & 3) == 0 && Nout >= 32;
}

static size_t point_prep_bytes(int math, int Nrows, int C) {
  const int NS = math < 1 ? 1 : (math == 4 ? 2 : (math > 3 ? 3 : math));
  const size_t b = (size_t)NS * (C >> 4) * ((Nrows + 31) / 32) * 512 * sizeof(unsigned short);
  return (b + 255) & ~(size_t)255;
}

// PDAE_C1_BF16=1 (tuning aid): the 1x1 kernels run the three-plane bf16 split also in mode 4
static int c1_math(int math) {
  return (math == 4 && pdae_knob(KNOB_C1_BF16)) ? 3 : math;
}

// prepared weights of all Nrows GEMM-N rows + split-K slabs sized for ANY launch on a 32-aligned sub-range of the rows
// (the data gradient of one concat source computes only that source's rows)
size_t conv1x1_wprep_bytes(int math, int Nrows, int C, long long M) {
  math = c1_math(math);
  size_t slab = 0;
  for (int nsub = 32; nsub <= ((Nrows + 31) & ~31); nsub += 32) {
    const int n = nsub < Nrows ? nsub : Nrows;
    const PointPlan q = point_plan(M, C, n);
    if (q.splits > 1) { const size_t b = (size_t)q.splits * M * n * sizeof(float); if (b > slab) slab = b; }
  }
  return point_prep_bytes(math, Nrows, C) + slab;
}

// power-of-two scale of the fp16-format prepared weights (x 2^4: the activations enter unscaled)
static float conv1x1_wscale(int C) { int k = 0; while ((1 << (2 * k)) < C) ++k; return (float)(16 << k); }

int conv1x1_wprep(int math, const float* w, int Nrows, int C, int transposed, unsigned short* wp, hipStream_t s) {
  math = c1_math(math);
  const float wscale = conv1x1_wscale(C);
  const int NT = (Nrows + 31) / 32;
  const size_t nslot = (size_t)(C >> 4) * NT * 64;
  int grid = (int)((nslot + 255) / 256); if (grid > 4096) grid = 4096;
  if (math == 1) hipLaunchKernelGGL(conv1x1_wprep_kernel<1>, dim3(grid), dim3(256), 0, s, w, Nrows, C, NT, transposed, wscale, wp);
  else if (math == 2) hipLaunchKernelGGL(conv1x1_wprep_kernel<2>, dim3(grid), dim3(256), 0, s, w, Nrows, C, NT, transposed, wscale, wp);
  else if (math == 4) hipLaunchKernelGGL(conv1x1_wprep_kernel<4>, dim3(grid), dim3(256), 0, s, w, Nrows, C, NT, transposed, wscale, wp);
  else hipLaunchKernelGGL(conv1x1_wprep_kernel<3>, dim3(grid), dim3(256), 0, s, w, Nrows, C, NT, transposed, wscale, wp);
  return pdae_launch_status("conv1x1_wprep");
}

void conv1x1_wprep_job(int math, const float* w, int Nrows, int C, int transposed, unsigned short* wp, WprepJob* j) {
  math = c1_math(math);
  j->w = w; j->wp = wp; j->Nout = Nrows; j->C = C; j->NT = (Nrows + 31) / 32; j->transposed = transposed; j->T = 0;      // T = 0: the 1x1 layout
  j->ns = math; j->wscale = conv1x1_wscale(C);
  j->nblocks = (int)(((size_t)(C >> 4) * j->NT * 64 + 255) / 256);
}

int conv1x1_launch(int math, const float* x0, int C0, const float* x1, int C1, long long M, const unsigned short* wp, int Nrows, int row_off,
                   int Nout, float* y, const float* bias, const float* res, int res_mode, int H, int W, int accumulate, hipStream_t s,
                   const float* amax) {
  math = c1_math(math);
  PointParams P;
  P.woscale = 1.0f / conv1x1_wscale(C0 + C1); P.amax = amax; P.sat = pdae_sat_counter();
  P.res_mode = res_mode; P.H = H; P.W = W;
  P.x0 = x0; P.x1 = x1; P.C0 = C0; P.C1 = C1; P.M = M; P.C = C0 + C1; P.wp = wp; P.NT = (Nrows + 31) / 32;
  P.nt_off = row_off >> 5; P.Nout = Nout; P.y = y; P.bias = bias; P.res = res; P.accumulate = accumulate;
  const PointPlan q = point_plan(M, P.C, Nout);
  P.tiles_m = q.tiles_m; P.tiles_n = q.tiles_n; P.splits = q.splits; P.sps = q.sps;
  P.slab = (float*)((char*)wp + point_prep_bytes(math, Nrows, P.C));
  P.rot = pdae_knob(KNOB_C1_ROT) != 0;
  // general (pointer) form: a 64-channel stage would straddle the two sources, or a source reaches 4 GiB
  const bool gen = (P.C1 > 0 && (P.C0 & 63)) || (unsigned long long)M * (unsigned)(P.C0 > P.C1 ? P.C0 : P.C1) * 4ull >= 0xfff00000ull;
  dim3 grid(q.tiles_m * q.tiles_n * q.splits);
#define PDAE_C1(NS_)                                                                     \
  if (gen) hipLaunchKernelGGL((conv1x1_kernel<NS_, true>), grid, dim3(256), 0, s, P);    \
  else hipLaunchKernelGGL((conv1x1_kernel<NS_, false>), grid, dim3(256), 0, s, P);                    \
  if (P.splits > 1) {                                                                    \
    long long nb = (M * (Nout >> 2) + 255) / 256; if (nb > 4096) nb = 4096;              \
    hipLaunchKernelGGL(conv1x1_reduce_kernel, dim3((int)nb), dim3(256), 0, s, P);        \
  }
  if (math == 1) { PDAE_C1(1) } else if (math == 2) { PDAE_C1(2) } else if (math == 4) { PDAE_C1(4) } else { PDAE_C1(3) }
#undef PDAE_C1
  return pdae_launch_status("conv1x1");
}
