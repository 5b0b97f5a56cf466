// f32-MFMA implicit-GEMM family for gfx950 (CDNA4).
//
// One kernel template serves every GEMM-shaped op on the PDAE hot path:
//   conv fwd   : A = im2col gather of NHWC activations (K-contiguous), B = weights [Cout][KH*KW*Cin]
//   conv dgrad : A = gather of dY (flipped taps, zero-dilated for stride 2), B = weights read "N-contiguous"
//   conv wgrad : A = dY^T (M-contiguous), B = gather of activations (N-contiguous), split-K over pixels
//   dense GEMM : NT / NN / TN strided-batched (attention QK^T, PV, their grads; Linear fwd/bwd)
//
// Arithmetic is exact fp32: v_mfma_f32_32x32x2_f32 (one rounding per product, fmaf chain,
// MI355X_MICROARCH.md "Matrix cores").  That instruction issues once per 64 cycles per SIMD, which
// leaves ample slack for the gather/predication VALU work, so the tiles are staged global->reg->LDS
// with the next tile's global loads in flight under the current tile's MFMAs (guide T14).
//
// Reference ops replaced: F.conv2d / F.conv1d(k=1) / F.linear / torch.einsum in
// model/module.py:242,265,276,412,420,450-457 and their autograd backward.
#include "common.h"
#include "igemm.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 32
#define LDK 36          // K-contiguous LDS row stride (floats): 16 rows hit 16 distinct 4-bank slots for ds_read_b128

__device__ __forceinline__ bool map_pix(const ConvGeom& g, int ly, int lx, int& sy, int& sx) {
  if ((unsigned)ly >= (unsigned)g.Hl || (unsigned)lx >= (unsigned)g.Wl) return false;
  if (g.up) { sy = ly >> 1; sx = lx >> 1; return true; }
  if (g.dil) { if ((ly | lx) & 1) return false; sy = ly >> 1; sx = lx >> 1; return true; }
  sy = ly; sx = lx; return true;
}

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }


// ---------------------------------------------------------------------------------------------
// bf16 "split operand" support.  An fp32 value is the exact sum of three bf16 planes (hi + mid + lo, 8 significand bits
// each; truncation keeps every remainder exact), so with NS planes per operand the products are formed on the bf16 MFMA
// pipe (16x the f32-MFMA rate) and accumulated in fp32:
//   NS = 1 : plain bf16 operands (round-to-nearest)                      1 MFMA  per 32x32x16 block
//   NS = 2 : hi + rn(lo)          products hi.hi + hi.lo + lo.hi         3 MFMAs, ~2^-17 relative per product
//   NS = 3 : hi + mid + lo exact  6 leading products                     6 MFMAs, ~2^-23 relative per product (fp32 grade)
// LDS holds NS planes per operand, each [rows][LDH] bf16, K-contiguous (80-byte rows: conflict-free ds_read_b128).
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define LDH 40
// the transposing stores of the outer-contiguous loaders write rows 4 apart (float4 along the outer index): with plain 80-byte
// rows that is a 16-way LDS bank conflict, hence the slot rotation below
#define BPLANE(BO) ((BO) * LDH)
// An 80-byte row holds five 16-byte slots (four of data + one of padding); the slot order is rotated by (row >> 2) so that
// rows 4 apart land on different banks (only for the outer-contiguous operands: plain rows are conflict-free for ds_read_b128).
// BSLOT(row, slot, rot) = bf16 offset of 16-byte k-slot `slot` (8 bf16) of `row`.
#define BSLOT(row, slot, rot) ((row) * LDH + (((rot) ? ((slot) + ((row) >> 2)) % 5 : (slot)) << 3))

__device__ __forceinline__ float trunc_bf(float a) { return __uint_as_float(__float_as_uint(a) & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_hi16(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_rn(float a, float b) {
  unsigned short x = __builtin_bit_cast(unsigned short, (__bf16)a), y = __builtin_bit_cast(unsigned short, (__bf16)b);
  return (unsigned)x | ((unsigned)y << 16);
}
// two consecutive-k values -> one packed word per plane
template <int NS> __device__ __forceinline__ void split2(float e0, float e1, unsigned (&w)[NS]) {
  if constexpr (NS == 1) { w[0] = pack_rn(e0, e1); }
  else {
    float h0 = trunc_bf(e0), h1 = trunc_bf(e1);
    float r0 = e0 - h0, r1 = e1 - h1;
    w[0] = pack_hi16(h0, h1);
    if constexpr (NS == 2) { w[1] = pack_rn(r0, r1); }
    else {
      float m0 = trunc_bf(r0), m1 = trunc_bf(r1);
      w[1] = pack_hi16(m0, m1);
      w[2] = pack_hi16(r0 - m0, r1 - m1);         // exact: <= 8 significant bits remain
    }
  }
}
// K-contiguous loaders: v[r] = 4 consecutive k of one row
template <int NS, int BO, int R> __device__ __forceinline__ void store_bf_kc(unsigned short* s, const float4 (&v)[R], int row0, int kq) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    unsigned a[NS], b[NS];
    split2<NS>(v[r].x, v[r].y, a);
    split2<NS>(v[r].z, v[r].w, b);
#pragma unroll
    for (int p = 0; p < NS; ++p)
      *reinterpret_cast<uint2*>(&s[p * BPLANE(BO) + BSLOT(row0 + 32 * r, kq >> 1, 0) + (kq & 1) * 4]) = make_uint2(a[p], b[p]);
  }
}
// outer-contiguous loaders with k-adjacent rows: v[r] = 4 consecutive outer indices at k = krow0*R + r  -> transpose while storing
template <int NS, int BO, int R> __device__ __forceinline__ void store_bf_oc(unsigned short* s, const float4 (&v)[R], int krow0, int oq) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float e[R];
#pragma unroll
    for (int r = 0; r < R; ++r) e[r] = j == 0 ? v[r].x : (j == 1 ? v[r].y : (j == 2 ? v[r].z : v[r].w));
    const int row = oq * 4 + j;
    if constexpr (R == 4) {
      unsigned a[NS], b[NS];
      split2<NS>(e[0], e[1], a);
      split2<NS>(e[2], e[3], b);
#pragma unroll
      for (int p = 0; p < NS; ++p) *reinterpret_cast<uint2*>(&s[p * BPLANE(BO) + BSLOT(row, krow0 >> 1, 1) + (krow0 & 1) * 4]) = make_uint2(a[p], b[p]);
    } else {
      unsigned a[NS];
      split2<NS>(e[0], e[1], a);
#pragma unroll
      for (int p = 0; p < NS; ++p) *reinterpret_cast<unsigned*>(&s[p * BPLANE(BO) + BSLOT(row, krow0 >> 2, 1) + (krow0 & 3) * 2]) = a[p];
    }
  }
}

template <int MODE, bool VEC, int BO> struct Op;

// ---------------------------------------------------------------------------------------------
// A = im2col gather, K-contiguous tile [BO rows][32 k]; k = tap*Cin + ci (ci fastest, NHWC)
// ---------------------------------------------------------------------------------------------
template <bool VEC, int BO> struct Op<OP_CONV_KC, VEC, BO> {
  static constexpr int R = BO / 32;
  static constexpr bool KC = true;
  int iy0[R], ix0[R], nb[R];
  float4 v[R];
  int kq, row0;
  __device__ __forceinline__ void init(const OpParams& p, long long, int o0, int O, int t) {
    const ConvGeom& g = p.g;
    kq = t & 7; row0 = t >> 3;
    const int hw = g.Ho * g.Wo;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int m = o0 + row0 + 32 * r;
      if (m < O) {
        int n = m / hw; int rem = m - n * hw; int oy = rem / g.Wo; int ox = rem - oy * g.Wo;
        iy0[r] = oy * g.stride - g.pad; ix0[r] = ox * g.stride - g.pad; nb[r] = n * g.Hs * g.Ws;
      } else { nb[r] = -1; iy0[r] = 0; ix0[r] = 0; }
    }
  }
  __device__ __forceinline__ void load(const OpParams& p, int k0, int kend) {
    const ConvGeom& g = p.g;
    if constexpr (VEC) {
      // K order of the vector path: (32-channel chunk, tap, channel-in-chunk) -- the 9 taps of one chunk run back to back,
      // so a block re-reads a (3 rows x 32 ch) patch from L1/L2 instead of sweeping whole rows once per tap
      const int T = g.KH * g.KW;
      int step = k0 >> 5; int chunk = step / T; int tap = step - chunk * T; int ci0 = chunk << 5;
      int dy = tap / g.KW; int dx = tap - dy * g.KW;
      const float* src; int Cs; int ci = ci0 + kq * 4;
      if (ci0 < g.C0) { src = g.src0; Cs = g.C0; } else { src = g.src1; Cs = g.C1; ci -= g.C0; }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        v[r] = f4zero();
        int sy, sx;
        if (nb[r] >= 0 && map_pix(g, iy0[r] + dy, ix0[r] + dx, sy, sx))
          v[r] = *reinterpret_cast<const float4*>(src + ((size_t)(nb[r] + sy * g.Ws + sx) * Cs + ci));
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          e[j] = 0.f;
          int k = k0 + kq * 4 + j;
          if (k < kend && nb[r] >= 0) {
            int tap = k / g.Cin; int ci = k - tap * g.Cin; int dy = tap / g.KW; int dx = tap - dy * g.KW;
            int sy, sx;
            if (map_pix(g, iy0[r] + dy, ix0[r] + dx, sy, sx)) {
              size_t pix = (size_t)(nb[r] + sy * g.Ws + sx);
              e[j] = (ci < g.C0) ? g.src0[pix * g.C0 + ci] : g.src1[pix * g.C1 + (ci - g.C0)];
            }
          }
        }
        v[r] = make_float4(e[0], e[1], e[2], e[3]);
      }
    }
  }
  __device__ __forceinline__ void store(float* s) const {
#pragma unroll
    for (int r = 0; r < R; ++r) *reinterpret_cast<float4*>(&s[(row0 + 32 * r) * LDK + kq * 4]) = v[r];
  }
  template <int NS> __device__ __forceinline__ void store_bf(unsigned short* s) const { store_bf_kc<NS, BO, R>(s, v, row0, kq); }
};

// ---------------------------------------------------------------------------------------------
// dense, K-contiguous: elem(o,k) = p[o*ld + k]
// ---------------------------------------------------------------------------------------------
template <bool VEC, int BO> struct Op<OP_DENSE_KC, VEC, BO> {
  static constexpr int R = BO / 32;
  static constexpr bool KC = true;
  const float* ptr[R];
  float4 v[R];
  int kq, row0;
  __device__ __forceinline__ void init(const OpParams& p, long long boff, int o0, int O, int t) {
    kq = t & 7; row0 = t >> 3;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int o = o0 + row0 + 32 * r;
      ptr[r] = (o < O) ? p.p + boff + (long long)o * p.ld : nullptr;
    }
  }
  __device__ __forceinline__ void load(const OpParams& p, int k0, int kend) {
    int k = k0 + kq * 4;
    int km = k;                                  // memory offset of logical k
    if (p.kpT) { int step = k0 >> 5; int chunk = step / p.kpT; int tap = step - chunk * p.kpT; km = tap * p.kpC + (chunk << 5) + kq * 4; }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if constexpr (VEC) {
        v[r] = (ptr[r] && k < kend) ? *reinterpret_cast<const float4*>(ptr[r] + km) : f4zero();
      } else {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = (ptr[r] && (k + j) < kend) ? ptr[r][k + j] : 0.f;
        v[r] = make_float4(e[0], e[1], e[2], e[3]);
      }
    }
  }
  __device__ __forceinline__ void store(float* s) const {
#pragma unroll
    for (int r = 0; r < R; ++r) *reinterpret_cast<float4*>(&s[(row0 + 32 * r) * LDK + kq * 4]) = v[r];
  }
  template <int NS> __device__ __forceinline__ void store_bf(unsigned short* s) const { store_bf_kc<NS, BO, R>(s, v, row0, kq); }
};

// ---------------------------------------------------------------------------------------------
// outer-contiguous family: LDS tile [32 k][BO + 4]
//   DENSE_OC : elem(o,k) = p[k*ld + o]
//   DGRAD_OC : k = tap'*Cout + co -> W[co][T-1-tap'][ciOff + o]           (conv dgrad weights)
//   GATHER_OC: k = output pixel, o = tap*Cin + ci -> activation gather     (conv wgrad)
// ---------------------------------------------------------------------------------------------
template <int MODE, bool VEC, int BO, bool KADJ = false> struct OpOC {
  static constexpr int R = BO / 32;
  static constexpr int CPR = BO / 4;           // float4 chunks per k-row
  static constexpr int KSTEP = 256 / CPR;      // k rows covered per pass
  static constexpr bool KC = false;
  float4 v[R];
  int krow0, oq, o, O_;
  const float* base;
  // GATHER_OC per-column state
  int dy[4], dx[4], cc[4], sel[4];
  __device__ __forceinline__ void init(const OpParams& p, long long boff, int o0, int O, int t) {
    krow0 = t / CPR; oq = t - krow0 * CPR; o = o0 + oq * 4; O_ = O; base = p.p + boff;
    if constexpr (MODE == OP_GATHER_OC) {
      const ConvGeom& g = p.g;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int n = o + j; if (n >= O) n = O - 1;
        int tap = n / g.Cin; int ci = n - tap * g.Cin;
        dy[j] = tap / g.KW; dx[j] = tap - dy[j] * g.KW;
        sel[j] = ci >= g.C0; cc[j] = sel[j] ? ci - g.C0 : ci;
      }
    }
  }
  __device__ __forceinline__ void load(const OpParams& p, int k0, int kend) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int k = KADJ ? k0 + krow0 * R + r : k0 + krow0 + KSTEP * r;
      v[r] = f4zero();
      if (k >= kend || o >= O_) continue;
      if constexpr (MODE == OP_GATHER_OC) {
        const ConvGeom& g = p.g;
        const int hw = g.Ho * g.Wo;
        int n = k / hw; int rem = k - n * hw; int oy = rem / g.Wo; int ox = rem - oy * g.Wo;
        int by = oy * g.stride - g.pad, bx = ox * g.stride - g.pad; int nb = n * g.Hs * g.Ws;
        if constexpr (VEC) {
          int sy, sx;
          if (map_pix(g, by + dy[0], bx + dx[0], sy, sx)) {
            size_t pix = (size_t)(nb + sy * g.Ws + sx);
            v[r] = sel[0] ? *reinterpret_cast<const float4*>(g.src1 + pix * g.C1 + cc[0])
                          : *reinterpret_cast<const float4*>(g.src0 + pix * g.C0 + cc[0]);
          }
        } else {
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            e[j] = 0.f; int sy, sx;
            if (o + j < O_ && map_pix(g, by + dy[j], bx + dx[j], sy, sx)) {
              size_t pix = (size_t)(nb + sy * g.Ws + sx);
              e[j] = sel[j] ? g.src1[pix * g.C1 + cc[j]] : g.src0[pix * g.C0 + cc[j]];
            }
          }
          v[r] = make_float4(e[0], e[1], e[2], e[3]);
        }
      } else {
        const float* row;
        if constexpr (MODE == OP_DGRAD_OC) {
          int tap, co;
          if (p.kpT) { int step = k0 >> 5; int chunk = step / p.kpT; tap = step - chunk * p.kpT; co = (chunk << 5) + (k - k0); }
          else { tap = k / p.dgCout; co = k - tap * p.dgCout; }
          row = base + ((size_t)co * p.dgT + (p.dgT - 1 - tap)) * p.dgWCin + p.dgCiOff;
        } else {
          row = base + (long long)k * p.ld;
        }
        if constexpr (VEC) {
          v[r] = *reinterpret_cast<const float4*>(row + o);
        } else {
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) e[j] = (o + j < O_) ? row[o + j] : 0.f;
          v[r] = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
    }
  }
  __device__ __forceinline__ void store(float* s) const {
#pragma unroll
    for (int r = 0; r < R; ++r) *reinterpret_cast<float4*>(&s[(krow0 + KSTEP * r) * (BO + 4) + oq * 4]) = v[r];
  }
  template <int NS> __device__ __forceinline__ void store_bf(unsigned short* s) const {
    static_assert(KADJ, "bf16 planes need the k-adjacent row mapping");
    store_bf_oc<NS, BO, R>(s, v, krow0, oq);
  }
};
template <bool VEC, int BO> struct Op<OP_DENSE_OC, VEC, BO> : OpOC<OP_DENSE_OC, VEC, BO> {};
template <bool VEC, int BO> struct Op<OP_DGRAD_OC, VEC, BO> : OpOC<OP_DGRAD_OC, VEC, BO> {};
template <bool VEC, int BO> struct Op<OP_GATHER_OC, VEC, BO> : OpOC<OP_GATHER_OC, VEC, BO> {};
template <int MODE, bool VEC, int BO> struct OpBF : Op<MODE, VEC, BO> {};                        // K-contiguous: unchanged
template <bool VEC, int BO> struct OpBF<OP_DENSE_OC, VEC, BO> : OpOC<OP_DENSE_OC, VEC, BO, true> {};
template <bool VEC, int BO> struct OpBF<OP_DGRAD_OC, VEC, BO> : OpOC<OP_DGRAD_OC, VEC, BO, true> {};
template <bool VEC, int BO> struct OpBF<OP_GATHER_OC, VEC, BO> : OpOC<OP_GATHER_OC, VEC, BO, true> {};

// ---------------------------------------------------------------------------------------------
// the kernel: 256 threads = 4 waves (WM x WN), each wave owns TM x TN tiles of 32x32
// ---------------------------------------------------------------------------------------------
template <int AM, bool AV, int BMODE, bool BV, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256) igemm_kernel(const GemmParams P) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  using LA = Op<AM, AV, BM>;
  using LB = Op<BMODE, BV, BN>;
  constexpr int SA = LA::KC ? BM * LDK : BK * (BM + 4);
  constexpr int SB = LB::KC ? BN * LDK : BK * (BN + 4);
  __shared__ __attribute__((aligned(16))) float smem[SA + SB];
  float* sA = smem;
  float* sB = smem + SA;

  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, h = lane >> 5;
  const int wm = w / WN, wn = w - wm * WN;

  // XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous run of tiles so
  // neighbouring output rows (shared halo) and the weight panel stay in one L2 (guide T1, bijective form)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int tid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int tiles_n = (P.N + BN - 1) / BN;
  const int tmi = tid / tiles_n, tni = tid - tmi * tiles_n;
  const int m0 = tmi * BM, n0 = tni * BN;

  int kbeg = 0, kend = P.K;
  long long aoff = 0, boff = 0, coff = 0;
  const int z = blockIdx.z;
  if (P.splitk > 1) {
    kbeg = z * P.kchunk; kend = min(P.K, kbeg + P.kchunk); coff = (long long)z * P.split_stride;
  } else {
    int bo = z / P.Bi, bi = z - bo * P.Bi;
    aoff = bo * P.a.so + bi * P.a.si; boff = bo * P.b.so + bi * P.b.si; coff = bo * P.sCo + bi * P.sCi;
  }

  LA la; LB lb;
  la.init(P.a, aoff, m0, P.M, t);
  lb.init(P.b, boff, n0, P.N, t);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) { la.load(P.a, kbeg, kend); lb.load(P.b, kbeg, kend); }
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    la.store(sA); lb.store(sB);
    __syncthreads();
    if (kt + 1 < nk) { la.load(P.a, kbeg + (kt + 1) * BK, kend); lb.load(P.b, kbeg + (kt + 1) * BK, kend); }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      // MFMA k-slot convention: step s of group kk consumes k = kk*8 + h*4 + s on lane half h
      float af[TM][4], bf[TN][4];
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const int row = (wm * TM + a) * 32 + li;
        if constexpr (LA::KC) {
          float4 x = *reinterpret_cast<const float4*>(&sA[row * LDK + kk * 8 + h * 4]);
          af[a][0] = x.x; af[a][1] = x.y; af[a][2] = x.z; af[a][3] = x.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) af[a][s] = sA[(kk * 8 + h * 4 + s) * (BM + 4) + row];
        }
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = (wn * TN + b) * 32 + li;
        if constexpr (LB::KC) {
          float4 x = *reinterpret_cast<const float4*>(&sB[col * LDK + kk * 8 + h * 4]);
          bf[b][0] = x.x; bf[b][1] = x.y; bf[b][2] = x.z; bf[b][3] = x.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) bf[b][s] = sB[(kk * 8 + h * 4 + s) * (BN + 4) + col];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][s], bf[b][s], acc[a][b], 0, 0, 0);
    }
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Cb = P.C + coff;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row >= P.M) continue;
      long long rrow = row;                     // residual row
      if (P.res_mode == 2) {
        int hw = P.rHo * P.rWo; int n = row / hw; int rem = row - n * hw; int oy = rem / P.rWo; int ox = rem - oy * P.rWo;
        rrow = ((long long)n * (P.rHo >> 1) + (oy >> 1)) * (P.rWo >> 1) + (ox >> 1);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = n0 + (wn * TN + b) * 32 + li;
        if (col >= P.N) continue;
        float val = P.alpha * acc[a][b][r];
        if (P.bias) val += P.bias[col];
        if (P.res_mode) val += P.res[rrow * P.ldr + col];
        float* dst = Cb + (long long)row * P.ldc + col;
        if (P.accumulate) val += *dst;
        *dst = val;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// bf16-MFMA variant of the kernel above (v_mfma_f32_32x32x16_bf16), NS planes per operand
// ---------------------------------------------------------------------------------------------
template <int AM, bool AV, int BMODE, bool BV, int BM, int BN, int WM, int WN, int NS>
__global__ void __launch_bounds__(256) igemm_bf_kernel(const GemmParams P) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  using LA = OpBF<AM, AV, BM>;
  using LB = OpBF<BMODE, BV, BN>;
  constexpr int SA = NS * BPLANE(BM), SB = NS * BPLANE(BN);
  __shared__ __attribute__((aligned(16))) unsigned short smem[SA + SB];
  unsigned short* sA = smem;
  unsigned short* sB = smem + SA;

  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, h = lane >> 5;
  const int wm = w / WN, wn = w - wm * WN;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int tid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int tiles_n = (P.N + BN - 1) / BN;
  const int tmi = tid / tiles_n, tni = tid - tmi * tiles_n;
  const int m0 = tmi * BM, n0 = tni * BN;

  int kbeg = 0, kend = P.K;
  long long aoff = 0, boff = 0, coff = 0;
  const int z = blockIdx.z;
  if (P.splitk > 1) {
    kbeg = z * P.kchunk; kend = min(P.K, kbeg + P.kchunk); coff = (long long)z * P.split_stride;
  } else {
    int bo = z / P.Bi, bi = z - bo * P.Bi;
    aoff = bo * P.a.so + bi * P.a.si; boff = bo * P.b.so + bi * P.b.si; coff = bo * P.sCo + bi * P.sCi;
  }
  LA la; LB lb;
  la.init(P.a, aoff, m0, P.M, t);
  lb.init(P.b, boff, n0, P.N, t);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) { la.load(P.a, kbeg, kend); lb.load(P.b, kbeg, kend); }
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    la.template store_bf<NS>(sA); lb.template store_bf<NS>(sB);
    __syncthreads();
    if (kt + 1 < nk) { la.load(P.a, kbeg + (kt + 1) * BK, kend); lb.load(P.b, kbeg + (kt + 1) * BK, kend); }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      bf16x8 af[TM][NS], bfr[TN][NS];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int p = 0; p < NS; ++p)
          af[a][p] = *reinterpret_cast<const bf16x8*>(&sA[p * BPLANE(BM) + BSLOT((wm * TM + a) * 32 + li, kc * 2 + h, !LA::KC)]);
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int p = 0; p < NS; ++p)
          bfr[b][p] = *reinterpret_cast<const bf16x8*>(&sB[p * BPLANE(BN) + BSLOT((wn * TN + b) * 32 + li, kc * 2 + h, !LB::KC)]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          if constexpr (NS == 3) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bfr[b][1], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bfr[b][2], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bfr[b][0], acc[a][b], 0, 0, 0);
          }
          if constexpr (NS >= 2) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bfr[b][1], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bfr[b][0], acc[a][b], 0, 0, 0);
          }
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bfr[b][0], acc[a][b], 0, 0, 0);
        }
    }
  }
  float* Cb = P.C + coff;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row >= P.M) continue;
      long long rrow = row;
      if (P.res_mode == 2) {
        int hw = P.rHo * P.rWo; int n = row / hw; int rem = row - n * hw; int oy = rem / P.rWo; int ox = rem - oy * P.rWo;
        rrow = ((long long)n * (P.rHo >> 1) + (oy >> 1)) * (P.rWo >> 1) + (ox >> 1);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = n0 + (wn * TN + b) * 32 + li;
        if (col >= P.N) continue;
        float val = P.alpha * acc[a][b][r];
        if (P.bias) val += P.bias[col];
        if (P.res_mode) val += P.res[rrow * P.ldr + col];
        float* dst = Cb + (long long)row * P.ldc + col;
        if (P.accumulate) val += *dst;
        *dst = val;
      }
    }
  }
}

// sums split-K slabs: out[i] = (acc ? out[i] : 0) + sum_s ws[s*n + i]   (fixed order => deterministic)
// Blocks behind the first `nmain` finish the bias gradient that rode in the weight-gradient launch: db[c] (+)= sum over `rows` partial rows
// (16 columns x 16 lanes per block, fixed order) -- it used to be a launch of its own behind every weight gradient (76 per FFHQ-128 step)
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long long n,
                                                             int splits, int accumulate, int nmain, const float* __restrict__ dbp, int rows, int C,
                                                             float* __restrict__ db) {
  if ((int)blockIdx.x >= nmain) {
    __shared__ float red[256];
    const int t = threadIdx.x, cl = t & 15, lane = t >> 4;
    const int c = ((int)blockIdx.x - nmain) * 16 + cl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
      int b = lane;
      for (; b + 48 < rows; b += 64) {
        a0 += dbp[(size_t)b * C + c]; a1 += dbp[(size_t)(b + 16) * C + c]; a2 += dbp[(size_t)(b + 32) * C + c]; a3 += dbp[(size_t)(b + 48) * C + c];
      }
      for (; b < rows; b += 16) a0 += dbp[(size_t)b * C + c];
    }
    red[t] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (lane == 0 && c < C) {
      float s = 0.f;
      for (int l = 0; l < 16; ++l) s += red[l * 16 + cl];
      db[c] = accumulate ? db[c] + s : s;
    }
    return;
  }
  long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  if (i4 + 3 < n) {
    float4 s = accumulate ? *reinterpret_cast<const float4*>(out + i4) : f4zero();
    // four slabs in flight per thread, added in slab order (the sum is bit-identical to the serial loop): one load -> wait -> add round trip
    // per slab made these few-microsecond kernels latency chains of `splits` memory round trips
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
      const float4 v0 = *reinterpret_cast<const float4*>(ws + (long long)k * n + i4), v1 = *reinterpret_cast<const float4*>(ws + (long long)(k + 1) * n + i4);
      const float4 v2 = *reinterpret_cast<const float4*>(ws + (long long)(k + 2) * n + i4), v3 = *reinterpret_cast<const float4*>(ws + (long long)(k + 3) * n + i4);
      s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
      s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
      s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
      s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
    }
    for (; k < splits; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(ws + (long long)k * n + i4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i4) = s;
  } else {
    for (long long i = i4; i < n; ++i) {
      float s = accumulate ? out[i] : 0.f;
      for (int k = 0; k < splits; ++k) s += ws[(long long)k * n + i];
      out[i] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------
template <int AM, bool AV, int BMODE, bool BV>
static int launch_tiles(int tile, const GemmParams& P, int zdim, hipStream_t s) {
  if (tile == 128) {
    dim3 grid(cdiv(P.M, 128) * cdiv(P.N, 128), 1, zdim);
    hipLaunchKernelGGL((igemm_kernel<AM, AV, BMODE, BV, 128, 128, 2, 2>), grid, dim3(256), 0, s, P);
  } else {
    dim3 grid(cdiv(P.M, 64) * cdiv(P.N, 64), 1, zdim);
    hipLaunchKernelGGL((igemm_kernel<AM, AV, BMODE, BV, 64, 64, 2, 2>), grid, dim3(256), 0, s, P);
  }
  return pdae_launch_status("igemm");
}

template <int AM, int BMODE>
static int launch_tiles_bf(int ns, int tile, const GemmParams& P, int zdim, hipStream_t s) {
#define PDAE_BF_LAUNCH(NS_)                                                                                              \
  if (tile == 128) {                                                                                                     \
    dim3 grid(cdiv(P.M, 128) * cdiv(P.N, 128), 1, zdim);                                                                 \
    hipLaunchKernelGGL((igemm_bf_kernel<AM, true, BMODE, true, 128, 128, 2, 2, NS_>), grid, dim3(256), 0, s, P);        \
  } else {                                                                                                               \
    dim3 grid(cdiv(P.M, 64) * cdiv(P.N, 64), 1, zdim);                                                                   \
    hipLaunchKernelGGL((igemm_bf_kernel<AM, true, BMODE, true, 64, 64, 2, 2, NS_>), grid, dim3(256), 0, s, P);          \
  }
  if (ns == 1) { PDAE_BF_LAUNCH(1) } else if (ns == 2) { PDAE_BF_LAUNCH(2) } else { PDAE_BF_LAUNCH(3) }
#undef PDAE_BF_LAUNCH
  return pdae_launch_status("igemm_bf");
}

static int pick_tile(long long M, long long N) {
  // 128x128 tiles once they fill the 256 CUs at least twice; otherwise 64x64 for occupancy
  long long t128 = (long long)cdiv(M, 128) * cdiv(N, 128);
  return (t128 >= 512 && N >= 96) ? 128 : 64;
}

static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

int igemm_conv_fwd(const GemmParams& P, int tile, int math, hipStream_t s) {
  const ConvGeom& g = P.a.g;
  bool vec = (g.Cin % 32 == 0) && (g.C0 % 32 == 0) && al16(g.src0) && (g.C1 == 0 || al16(g.src1)) && al16(P.b.p);
  if (tile == 0) tile = pick_tile(P.M, P.N);
  if (vec) { GemmParams Q = P; Q.b.kpT = g.KH * g.KW; Q.b.kpC = g.Cin;
    if (math > 0) return launch_tiles_bf<OP_CONV_KC, OP_DENSE_KC>(math, tile, Q, 1, s);
    return launch_tiles<OP_CONV_KC, true, OP_DENSE_KC, true>(tile, Q, 1, s); }
  if (vec && math > 0) return launch_tiles_bf<OP_CONV_KC, OP_DENSE_KC>(math, tile, P, 1, s);
  if (vec) return launch_tiles<OP_CONV_KC, true, OP_DENSE_KC, true>(tile, P, 1, s);
  return launch_tiles<OP_CONV_KC, false, OP_DENSE_KC, false>(tile, P, 1, s);
}

int igemm_conv_dgrad(const GemmParams& P, int tile, int math, hipStream_t s) {
  const ConvGeom& g = P.a.g;
  bool avec = (g.Cin % 32 == 0) && al16(g.src0);
  bool bvec = (P.b.dgWCin % 4 == 0) && (P.b.dgCiOff % 4 == 0) && (P.N % 4 == 0) && al16(P.b.p);
  if (tile == 0) tile = pick_tile(P.M, P.N);
  if (avec && bvec) { GemmParams Q = P; Q.b.kpT = g.KH * g.KW; Q.b.kpC = g.Cin;
    if (math > 0) return launch_tiles_bf<OP_CONV_KC, OP_DGRAD_OC>(math, tile, Q, 1, s);
    return launch_tiles<OP_CONV_KC, true, OP_DGRAD_OC, true>(tile, Q, 1, s); }
  if (!avec && bvec) return launch_tiles<OP_CONV_KC, false, OP_DGRAD_OC, true>(tile, P, 1, s);
  return launch_tiles<OP_CONV_KC, false, OP_DGRAD_OC, false>(tile, P, 1, s);
}

int igemm_conv_wgrad(const GemmParams& P, int tile, int splits, int math, hipStream_t s) {
  const ConvGeom& g = P.b.g;
  bool avec = (P.M % 4 == 0) && (P.a.ld % 4 == 0) && al16(P.a.p);
  bool bvec = (g.Cin % 4 == 0) && (g.C0 % 4 == 0) && al16(g.src0) && (g.C1 == 0 || al16(g.src1));
  if (avec && bvec && math > 0) return launch_tiles_bf<OP_DENSE_OC, OP_GATHER_OC>(math, tile, P, splits, s);
  if (avec && bvec) return launch_tiles<OP_DENSE_OC, true, OP_GATHER_OC, true>(tile, P, splits, s);
  if (avec && !bvec) return launch_tiles<OP_DENSE_OC, true, OP_GATHER_OC, false>(tile, P, splits, s);
  if (!avec && bvec) return launch_tiles<OP_DENSE_OC, false, OP_GATHER_OC, true>(tile, P, splits, s);
  return launch_tiles<OP_DENSE_OC, false, OP_GATHER_OC, false>(tile, P, splits, s);
}

int igemm_splitk_reduce(const float* ws, float* out, long long n, int splits, int accumulate, hipStream_t s, const float* db_part, int db_rows,
                        int C, float* db) {
  const int nmain = cdiv(n, 1024), nextra = (db_part && db) ? cdiv(C, 16) : 0;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nmain + nextra), dim3(256), 0, s, ws, out, n, splits, accumulate, nmain, db_part, db_rows, C, db);
  return pdae_launch_status("splitk_reduce");
}

int igemm_dense(int transA, int transB, const GemmParams& P, int zdim, hipStream_t s) {
  auto m4 = [](long long v) { return (v & 3) == 0; };
  const int tile = 64;
  if (!transA && transB) {          // NT: A[m][k], B[n][k]
    bool vec = m4(P.K) && m4(P.a.ld) && m4(P.b.ld) && m4(P.a.so) && m4(P.a.si) && m4(P.b.so) && m4(P.b.si) && al16(P.a.p) && al16(P.b.p);
    if (vec) return launch_tiles<OP_DENSE_KC, true, OP_DENSE_KC, true>(tile, P, zdim, s);
    return launch_tiles<OP_DENSE_KC, false, OP_DENSE_KC, false>(tile, P, zdim, s);
  }
  if (!transA && !transB) {         // NN: A[m][k], B[k][n]
    bool vec = m4(P.K) && m4(P.N) && m4(P.a.ld) && m4(P.b.ld) && m4(P.a.so) && m4(P.a.si) && m4(P.b.so) && m4(P.b.si) && al16(P.a.p) && al16(P.b.p);
    if (vec) return launch_tiles<OP_DENSE_KC, true, OP_DENSE_OC, true>(tile, P, zdim, s);
    return launch_tiles<OP_DENSE_KC, false, OP_DENSE_OC, false>(tile, P, zdim, s);
  }
  if (transA && !transB) {          // TN: A[k][m], B[k][n]
    bool vec = m4(P.M) && m4(P.N) && m4(P.a.ld) && m4(P.b.ld) && m4(P.a.so) && m4(P.a.si) && m4(P.b.so) && m4(P.b.si) && al16(P.a.p) && al16(P.b.p);
    if (vec) return launch_tiles<OP_DENSE_OC, true, OP_DENSE_OC, true>(tile, P, zdim, s);
    return launch_tiles<OP_DENSE_OC, false, OP_DENSE_OC, false>(tile, P, zdim, s);
  }
  pdae_set_error("pdae_gemm: transA=1,transB=1 is not used on this path and not built");
  return PDAE_EINVAL;
}
