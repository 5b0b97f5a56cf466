// C ABI of libpdae_hip.so (see include/pdae_hip.h): argument validation, GEMM parameter blocks for the
// convolution entry points, and the planned-graph executor.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/pdae_hip.h"
#include "common.h"
#include "igemm.h"
#include "kernels.h"
#include "conv3x3p.h"

static thread_local char g_err[512] = "";

void pdae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* pdae_last_error(void) { return g_err; }
extern "C" int pdae_abi_version(void) { return 11; }

// fp16-window saturation counter (common.h): one device word per process (one process drives one GPU)
static unsigned int* g_sat = nullptr;
unsigned int* pdae_sat_counter() { return g_sat; }
extern "C" int pdae_set_saturation_counter(unsigned int* counter) { g_sat = counter; return PDAE_OK; }

static inline hipStream_t S(pdae_stream_t s) { return (hipStream_t)s; }

// ---- knob registry (common.h): the only getenv of the library
#include <stdlib.h>
#include <mutex>
static const struct { const char* name; int def; } g_knob_def[KNOB_COUNT] = {
    {"PDAE_W1", 1}, {"PDAE_W1_EFF", 85}, {"PDAE_P3R", 1}, {"PDAE_P3R_MIN", 512}, {"PDAE_P3R_EFF", 85}, {"PDAE_EDGE", 1}, {"PDAE_P3_TH", 0},
    {"PDAE_SPLIT_STATS", 1}, {"PDAE_W3_STAGGER", 0}, {"PDAE_Y_STAGGER", 0}, {"PDAE_C1_SLAB", 1}, {"PDAE_C1_BF16", 0}, {"PDAE_NO_SKINNY", 0},
    {"PDAE_C1_ROT", 1}, {"PDAE_W1_ROWS8", 1}, {"PDAE_W1_EFF8", 70}, {"PDAE_W1_MIN8", 160}, {"PDAE_SIDE_STREAM", 1}, {"PDAE_W3V", 1}, {"PDAE_Y_XCD", 0}, {"PDAE_Y_GRID_TRIM", 0}};
#include <atomic>
static std::atomic<int> g_knob_val[KNOB_COUNT];
static std::atomic<bool> g_knob_set[KNOB_COUNT];
static std::mutex g_knob_mu;
int pdae_knob(int id) {
  if (g_knob_set[id].load(std::memory_order_acquire)) return g_knob_val[id].load(std::memory_order_relaxed);      // the per-launch path: no lock
  std::lock_guard<std::mutex> lk(g_knob_mu);
  if (!g_knob_set[id].load(std::memory_order_relaxed)) {
    const char* e = getenv(g_knob_def[id].name);
    g_knob_val[id].store((e && *e) ? atoi(e) : g_knob_def[id].def, std::memory_order_relaxed);
    g_knob_set[id].store(true, std::memory_order_release);
  }
  return g_knob_val[id].load(std::memory_order_relaxed);
}
static int knob_id(const char* name) {
  if (name)
    for (int k = 0; k < KNOB_COUNT; ++k)
      if (!strcmp(name, g_knob_def[k].name)) return k;
  return -1;
}
extern "C" int pdae_set_knob(const char* name, int value) {
  const int id = knob_id(name);
  PDAE_CHECK_ARG(id >= 0, "set_knob: unknown knob '%s'", name ? name : "(null)");
  std::lock_guard<std::mutex> lk(g_knob_mu);
  g_knob_val[id].store(value, std::memory_order_relaxed);
  g_knob_set[id].store(true, std::memory_order_release);
  return PDAE_OK;
}
extern "C" int pdae_get_knob(const char* name, int* value) {
  const int id = knob_id(name);
  PDAE_CHECK_ARG(id >= 0 && value, "get_knob: unknown knob '%s'", name ? name : "(null)");
  *value = pdae_knob(id);
  return PDAE_OK;
}

static int check_desc(const pdae_conv_desc* d) {
  PDAE_CHECK_ARG(d && d->N > 0 && d->Hi > 0 && d->Wi > 0 && d->C0 > 0 && d->C1 >= 0 && d->Cout > 0, "conv: bad dims");
  PDAE_CHECK_ARG(d->stride == 1 || d->stride == 2, "conv: stride must be 1 or 2");
  PDAE_CHECK_ARG(!(d->up && d->stride != 1), "conv: up with stride!=1");
  PDAE_CHECK_ARG(d->math >= 0 && d->math <= 4, "conv: math mode must be 0..4");
  int Hl = d->up ? 2 * d->Hi : d->Hi, Wl = d->up ? 2 * d->Wi : d->Wi;
  PDAE_CHECK_ARG(d->Ho == (Hl + 2 * d->pad - d->KH) / d->stride + 1 && d->Wo == (Wl + 2 * d->pad - d->KW) / d->stride + 1,
                 "conv: output size %dx%d inconsistent with input %dx%d k%d s%d p%d", d->Ho, d->Wo, Hl, Wl, d->KH, d->stride, d->pad);
  PDAE_CHECK_ARG((long long)d->N * Hl * Wl < (1ll << 31) && (long long)d->N * d->Ho * d->Wo < (1ll << 31), "conv: too many pixels");
  return PDAE_OK;
}

// pdae_conv_desc.math may carry PDAE_MATH_DIRECT: every entry point works on a copy without it (d is re-pointed) and keeps the bit in d_dflag;
// only the FORWARD weight preparation / launch of a 3x3 convolution look at it (conv3x3p_form).
static int skip_ok_impl(const pdae_conv_desc* d, const pdae_conv_desc* ds, int dflag);
#define PDAE_DESC_NORM(d)                                                                                                  \
  pdae_conv_desc d##_norm; int d##_dflag = 0;                                                                              \
  if (d) { d##_norm = *d; d##_dflag = d##_norm.math & PDAE_MATH_DIRECT; d##_norm.math &= ~PDAE_MATH_DIRECT; d = &d##_norm; }

static void fwd_geom(ConvGeom& g, const pdae_conv_desc* d, const float* x0, const float* x1) {
  g.src0 = x0; g.src1 = x1; g.C0 = d->C0; g.C1 = d->C1; g.Cin = d->C0 + d->C1;
  g.Hs = d->Hi; g.Ws = d->Wi; g.Hl = d->up ? 2 * d->Hi : d->Hi; g.Wl = d->up ? 2 * d->Wi : d->Wi;
  g.Ho = d->Ho; g.Wo = d->Wo; g.KH = d->KH; g.KW = d->KW; g.stride = d->stride; g.pad = d->pad; g.up = d->up; g.dil = 0;
}

// mode 4 (two fp16 planes, 3 products) is a FORWARD format: activations and weights live inside the fp16 range, gradients do not
// (dY of a converged network underflows); every gradient kernel runs the exact three-plane bf16 split instead
static int bwd_math(const pdae_conv_desc* d, bool f16_grad = false) { return d->math == 4 ? (f16_grad ? 4 : 3) : d->math; }

// fast-path kind of the forward (transposed = 0) / data-gradient (transposed = 1) convolution of d:
//   3 = LDS-patch 3x3 kernel (conv3x3p.hip), 1 = 1x1 kernel (conv1x1.hip), 0 = generic implicit GEMM only
static int fast_kind(const pdae_conv_desc* d, int transposed, bool fill) {
  const int Hl = d->up ? 2 * d->Hi : d->Hi, Wl = d->up ? 2 * d->Wi : d->Wi, Cin = d->C0 + d->C1;
  if (!transposed) {
    if (conv3x3p_ok(d->math, d->KH, d->KW, d->stride, d->pad, d->C1, d->C0, d->Ho, d->Wo, d->N, d->Cout, fill)) return 3;
    if (conv1x1_ok(d->math, d->KH, d->KW, d->stride, d->pad, d->up, d->C0, d->C1, d->Cout)) return 1;
    return 0;
  }
  if (conv3x3p_ok(d->math, d->KH, d->KW, d->stride, d->pad, 0, d->Cout, Hl, Wl, d->N, Cin, fill)) return 3;
  if (conv1x1_ok(d->math, d->KH, d->KW, d->stride, d->pad, d->up, d->Cout, 0, Cin)) return 1;
  return 0;
}

// forward 3x3 convolution that applies GroupNorm/AdaGN(+SiLU) to its (possibly two-source) input inside the patch staging
static bool gn_patch_ok(const pdae_conv_desc* d, bool fill) {
  if ((d->C0 & 31) || (d->C1 & 31) || (d->Wo % 16)) return false;
  return conv3x3p_ok(d->math, d->KH, d->KW, d->stride, d->pad, 0, d->C0 + d->C1, d->Ho, d->Wo, d->N, d->Cout, fill);
}

extern "C" size_t pdae_conv_wprep_bytes(const pdae_conv_desc* d, int flags) {
  PDAE_DESC_NORM(d)
  const int transposed = flags & PDAE_WPREP_TRANSPOSED;
  if (!d || check_desc(d)) return 0;
  if (flags & PDAE_WPREP_GN)
    return (!transposed && gn_patch_ok(d, !(flags & PDAE_WPREP_FORCE))) ? conv3x3p_wprep_bytes(d->math | d_dflag, d->Cout, d->C0 + d->C1, d->Ho, d->Wo, d->N) : 0;
  const int kind = fast_kind(d, transposed, !(flags & PDAE_WPREP_FORCE));
  const int Hl = d->up ? 2 * d->Hi : d->Hi, Wl = d->up ? 2 * d->Wi : d->Wi, Cin = d->C0 + d->C1;
  if (kind == 3)
    return transposed ? conv3x3p_wprep_bytes(bwd_math(d, flags & PDAE_WPREP_F16_GRAD), Cin, d->Cout, Hl, Wl, d->N)
                      : conv3x3p_wprep_bytes(d->math | d_dflag, d->Cout, d->C0, d->Ho, d->Wo, d->N);
  if (kind == 1) {
    const long long M = (long long)d->N * d->Ho * d->Wo;
    return transposed ? conv1x1_wprep_bytes(bwd_math(d, flags & PDAE_WPREP_F16_GRAD), Cin, d->Cout, M) : conv1x1_wprep_bytes(d->math, d->Cout, Cin, M);
  }
  return 0;
}

extern "C" int pdae_conv3x3_form(const pdae_conv_desc* d, int flags) {
  PDAE_DESC_NORM(d)
  if (pdae_conv_wprep_bytes(d, flags) == 0 || d->KH != 3) return 0;
  const int transposed = flags & PDAE_WPREP_TRANSPOSED;
  const int Hl = d->up ? 2 * d->Hi : d->Hi, Wl = d->up ? 2 * d->Wi : d->Wi, Cin = d->C0 + d->C1;
  if (transposed) return conv3x3p_form(bwd_math(d, flags & PDAE_WPREP_F16_GRAD), d->Cout, Hl, Wl, d->N, Cin);
  return conv3x3p_form(d->math | d_dflag, Cin, d->Ho, d->Wo, d->N, d->Cout);
}

extern "C" int pdae_conv_wprep(const pdae_conv_desc* d, const float* w, int flags, void* wp, pdae_stream_t stream) {
  PDAE_DESC_NORM(d)
  if (int e = check_desc(d)) return e;
  const int transposed = flags & PDAE_WPREP_TRANSPOSED;
  PDAE_CHECK_ARG(w && wp, "conv_wprep: null pointer");
  if (flags & PDAE_WPREP_GN) {
    PDAE_CHECK_ARG(!transposed && gn_patch_ok(d, false), "conv_wprep: convolution not eligible for the fused-GroupNorm patch kernel");
    return conv3x3p_wprep(d->math | d_dflag, w, d->Cout, d->C0 + d->C1, 0, (unsigned short*)wp, S(stream), d->Ho, d->Wo, d->N);
  }
  const int kind = fast_kind(d, transposed, false), Cin = d->C0 + d->C1;
  const int Hl = d->up ? 2 * d->Hi : d->Hi, Wl = d->up ? 2 * d->Wi : d->Wi;       // the data gradient's output grid
  PDAE_CHECK_ARG(kind != 0, "conv_wprep: convolution shape not eligible for a prepared-weight kernel");
  if (kind == 3) {
    if (transposed) return conv3x3p_wprep(bwd_math(d, flags & PDAE_WPREP_F16_GRAD), w, Cin, d->Cout, 1, (unsigned short*)wp, S(stream), Hl, Wl, d->N);
    return conv3x3p_wprep(d->math | d_dflag, w, d->Cout, d->C0, 0, (unsigned short*)wp, S(stream), d->Ho, d->Wo, d->N);
  }
  if (transposed) return conv1x1_wprep(bwd_math(d, flags & PDAE_WPREP_F16_GRAD), w, Cin, d->Cout, 1, (unsigned short*)wp, S(stream));
  return conv1x1_wprep(d->math, w, d->Cout, Cin, 0, (unsigned short*)wp, S(stream));
}

// ---- grouped weight preparation (wprep.hip): the host fills one job per prepared copy with the SAME decisions pdae_conv_wprep /
// pdae_conv_skip_wprep take, uploads the table once, and every run of the plan prepares all of them in one launch
static_assert(sizeof(pdae_wprep_job) == sizeof(WprepJob), "pdae_wprep_job must mirror WprepJob");
static_assert(PDAE_MATH_DIRECT == PDAE_MATH_DIRECT_BIT, "the direct-form bit of pdae_conv_desc.math");
extern "C" int pdae_conv_wprep_job(const pdae_conv_desc* d, const float* w, int flags, void* wp, pdae_wprep_job* job) {
  PDAE_DESC_NORM(d)
  if (int e = check_desc(d)) return e;
  const int transposed = flags & PDAE_WPREP_TRANSPOSED;
  PDAE_CHECK_ARG(w && wp && job, "conv_wprep_job: null pointer");
  WprepJob* j = reinterpret_cast<WprepJob*>(job);
  if (flags & PDAE_WPREP_GN) {
    PDAE_CHECK_ARG(!transposed && gn_patch_ok(d, false), "conv_wprep_job: convolution not eligible for the fused-GroupNorm patch kernel");
    conv3x3p_wprep_job(d->math | d_dflag, w, d->Cout, d->C0 + d->C1, 0, (unsigned short*)wp, j, d->Ho, d->Wo, d->N);
    return PDAE_OK;
  }
  const int kind = fast_kind(d, transposed, false), Cin = d->C0 + d->C1;
  const int Hl = d->up ? 2 * d->Hi : d->Hi, Wl = d->up ? 2 * d->Wi : d->Wi;
  PDAE_CHECK_ARG(kind != 0, "conv_wprep_job: convolution shape not eligible for a prepared-weight kernel");
  if (kind == 3) {
    if (transposed) conv3x3p_wprep_job(bwd_math(d, flags & PDAE_WPREP_F16_GRAD), w, Cin, d->Cout, 1, (unsigned short*)wp, j, Hl, Wl, d->N);
    else conv3x3p_wprep_job(d->math | d_dflag, w, d->Cout, d->C0, 0, (unsigned short*)wp, j, d->Ho, d->Wo, d->N);
  } else if (transposed) conv1x1_wprep_job(bwd_math(d, flags & PDAE_WPREP_F16_GRAD), w, Cin, d->Cout, 1, (unsigned short*)wp, j);
  else conv1x1_wprep_job(d->math, w, d->Cout, Cin, 0, (unsigned short*)wp, j);
  return PDAE_OK;
}
extern "C" int pdae_conv_skip_wprep_job(const pdae_conv_desc* d, const pdae_conv_desc* ds, const float* w_skip, void* wps, pdae_wprep_job* job) {
  PDAE_DESC_NORM(d)
  PDAE_DESC_NORM(ds)
  PDAE_CHECK_ARG(w_skip && wps && job && skip_ok_impl(d, ds, d_dflag), "conv_skip_wprep_job: not an eligible (conv3x3, skip 1x1) pair");
  conv3x3p_skip_wprep_job(d->math, w_skip, ds->Cout, ds->C0 + ds->C1, d->C0 + d->C1, (unsigned short*)wps, reinterpret_cast<WprepJob*>(job), d->Ho, d->Wo, d->N);
  return PDAE_OK;
}
extern "C" int pdae_conv_wprep_group(const pdae_wprep_job* jobs_dev, const int32_t* first_block_dev, int njobs, int total_blocks, pdae_stream_t stream) {
  PDAE_CHECK_ARG(njobs >= 0 && total_blocks >= 0 && (njobs == 0 || (jobs_dev && first_block_dev)), "conv_wprep_group: bad arguments");
  return wprep_group_launch(reinterpret_cast<const WprepJob*>(jobs_dev), first_block_dev, njobs, total_blocks, S(stream));
}

// PDAE_EDGE=0: the 3-channel edge layers stay on the fp32 FMA head kernels / the generic implicit GEMM (A/B aid)
static bool edge_on() { return pdae_knob(KNOB_EDGE) != 0; }

extern "C" int pdae_conv2d_fwd(const pdae_conv_desc* d, const float* x0, const float* x1, const float* w, const void* wp, const float* bias,
                               const float* res, int res_mode, float* y, int tile, pdae_stream_t stream) {
  PDAE_DESC_NORM(d)
  float* const stat = conv3x3p_take_stats();            // consumed here whatever happens below: a failed call never leaves the request armed
  if (int e = check_desc(d)) return e;
  PDAE_CHECK_ARG(x0 && w && y && (d->C1 == 0 || x1), "conv2d_fwd: null pointer");
  PDAE_CHECK_ARG(res_mode == 0 || res, "conv2d_fwd: res_mode without res");
  PDAE_CHECK_ARG(res_mode != 2 || ((d->Ho % 2) == 0 && (d->Wo % 2) == 0), "conv2d_fwd: res_mode 2 needs even output");
  if (tile == 0 && res_mode == 0 && !stat && edge_on() && edge_head_ok(d->KH, d->KW, d->stride, d->pad, d->up, d->C1, d->C0, d->Cout))
    return edge_head_fwd(x0, d->N, d->Hi, d->Wi, d->C0, w, d->Cout, bias, y, S(stream));
  if (tile == 0 && res_mode == 0 && !stat && convhead_ok(d->KH, d->KW, d->stride, d->pad, d->up, d->C1, d->C0, d->Cout))
    return convhead_fwd(x0, d->N, d->Hi, d->Wi, d->C0, w, d->Cout, bias, y, S(stream));
  if (tile == 0 && res_mode == 0 && !stat && !wp && edge_on() && edge_in_ok(d->KH, d->KW, d->stride, d->pad, d->up, d->C1, d->C0, d->Cout))
    return edge_in_conv(x0, d->N, d->Hi, d->Wi, d->C0, w, 0, d->Cout, bias, y, 0, S(stream));
  const int kind = wp ? fast_kind(d, 0, false) : 0;
  PDAE_CHECK_ARG(!wp || (tile == 0 && kind != 0), "conv2d_fwd: wp given but the convolution is not eligible for a prepared-weight kernel");
  if (kind == 3)
    return conv3x3p_launch(d->math | d_dflag, x0, d->N, d->Hi, d->Wi, d->C0, d->Ho, d->Wo, d->up, (const unsigned short*)wp, d->Cout, y, bias,
                           res_mode ? res : nullptr, res_mode, 0, S(stream), nullptr, 0, nullptr, 0, nullptr, nullptr, stat);
  if (stat) {
    pdae_set_error("conv2d_fwd: output statistics were requested (pdae_conv_stats_arm) but this convolution does not run on the 3x3 patch kernel");
    return PDAE_EINVAL;
  }
  if (kind == 1)
    return conv1x1_launch(d->math, x0, d->C0, x1, d->C1, (long long)d->N * d->Ho * d->Wo, (const unsigned short*)wp, d->Cout, 0, d->Cout, y, bias,
                          res_mode ? res : nullptr, res_mode, d->Ho, d->Wo, 0, S(stream));
  GemmParams P;
  memset(&P, 0, sizeof(P));
  fwd_geom(P.a.g, d, x0, x1);
  P.M = d->N * d->Ho * d->Wo; P.N = d->Cout; P.K = d->KH * d->KW * (d->C0 + d->C1);
  P.splitk = 1; P.kchunk = P.K; P.Bi = 1;
  P.b.p = w; P.b.ld = P.K;
  P.C = y; P.ldc = d->Cout; P.bias = bias; P.res = res_mode ? res : nullptr; P.ldr = d->Cout; P.res_mode = res_mode;
  P.rHo = d->Ho; P.rWo = d->Wo; P.alpha = 1.0f; P.accumulate = 0;
  return igemm_conv_fwd(P, tile, d->math, S(stream));
}

extern "C" int pdae_conv2d_fwd_gn(const pdae_conv_desc* d, const float* x0, const float* x1, const float* coef, int act, const void* wp,
                                  const float* bias, const float* res, int res_mode, float* y, pdae_stream_t stream) {
  PDAE_DESC_NORM(d)
  float* const stat = conv3x3p_take_stats();
  if (int e = check_desc(d)) return e;
  PDAE_CHECK_ARG(x0 && coef && wp && y && (d->C1 == 0 || x1), "conv2d_fwd_gn: null pointer");
  PDAE_CHECK_ARG(res_mode == 0 || res, "conv2d_fwd_gn: res_mode without res");
  PDAE_CHECK_ARG(res_mode != 2 || ((d->Ho % 2) == 0 && (d->Wo % 2) == 0), "conv2d_fwd_gn: res_mode 2 needs even output");
  PDAE_CHECK_ARG(gn_patch_ok(d, false), "conv2d_fwd_gn: convolution not eligible (pdae_conv_wprep_bytes(d, PDAE_WPREP_GN) == 0)");
  return conv3x3p_launch(d->math | d_dflag, x0, d->N, d->Hi, d->Wi, d->C0 + d->C1, d->Ho, d->Wo, d->up, (const unsigned short*)wp, d->Cout, y, bias,
                         res_mode ? res : nullptr, res_mode, 0, S(stream), d->C1 ? x1 : nullptr, d->C0, coef, act, nullptr, nullptr, stat);
}

static bool skip_desc_ok(const pdae_conv_desc* d, const pdae_conv_desc* ds) {
  return ds && ds->KH == 1 && ds->KW == 1 && ds->stride == 1 && ds->pad == 0 && !ds->up && ds->N == d->N && ds->Hi == d->Ho && ds->Wi == d->Wo &&
         ds->Ho == d->Ho && ds->Wo == d->Wo && ds->Cout == d->Cout && ds->math == d->math;
}

static int skip_ok_impl(const pdae_conv_desc* d, const pdae_conv_desc* ds, int dflag) {      // normalised descriptors + the main convolution's PDAE_MATH_DIRECT bit
  if (!d || !ds || check_desc(d) || check_desc(ds) || !skip_desc_ok(d, ds)) return 0;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || (d->C0 & 31) || (d->C1 & 31)) return 0;
  return conv3x3p_skip_ok(d->math | dflag, d->C0 + d->C1, d->Ho, d->Wo, d->N, d->Cout, d->up, ds->C0, ds->C1) ? 1 : 0;
}
extern "C" int pdae_conv2d_fwd_skip_ok(const pdae_conv_desc* d, const pdae_conv_desc* ds) {
  PDAE_DESC_NORM(d)
  PDAE_DESC_NORM(ds)
  (void)ds_dflag;
  return skip_ok_impl(d, ds, d_dflag);
}

extern "C" size_t pdae_conv_skip_wprep_bytes(const pdae_conv_desc* d, const pdae_conv_desc* ds) {
  PDAE_DESC_NORM(d)
  PDAE_DESC_NORM(ds)
  if (!skip_ok_impl(d, ds, d_dflag)) return 0;
  return conv3x3p_skip_wprep_bytes(d->math, ds->Cout, ds->C0 + ds->C1);
}

extern "C" int pdae_conv_skip_wprep(const pdae_conv_desc* d, const pdae_conv_desc* ds, const float* w_skip, void* wps, pdae_stream_t stream) {
  PDAE_DESC_NORM(d)
  PDAE_DESC_NORM(ds)
  PDAE_CHECK_ARG(w_skip && wps && skip_ok_impl(d, ds, d_dflag), "conv_skip_wprep: not an eligible (conv3x3, skip 1x1) pair");
  return conv3x3p_skip_wprep(d->math, w_skip, ds->Cout, ds->C0 + ds->C1, d->C0 + d->C1, (unsigned short*)wps, S(stream), d->Ho, d->Wo, d->N);
}

extern "C" int pdae_conv2d_fwd_skip(const pdae_conv_desc* d, const float* x0, const float* x1, const float* coef, int act, const void* wp,
                                    const float* bias, const pdae_conv_desc* ds, const float* s0, const float* s1, const void* wps,
                                    const float* bias_s, float* y, pdae_stream_t stream) {
  PDAE_DESC_NORM(d)
  PDAE_DESC_NORM(ds)
  float* const stat = conv3x3p_take_stats();
  if (int e = check_desc(d)) return e;
  PDAE_CHECK_ARG(ds && !check_desc(ds) && skip_desc_ok(d, ds), "conv2d_fwd_skip: the 1x1 descriptor must map the conv's output grid (same N, H, W, Cout, math)");
  PDAE_CHECK_ARG(x0 && wp && y && s0 && wps && (ds->C1 == 0 || s1), "conv2d_fwd_skip: null pointer");
  PDAE_CHECK_ARG(coef || d->C1 == 0, "conv2d_fwd_skip: a two-source main input needs the fused GroupNorm form (coef)");
  PDAE_CHECK_ARG(d->C1 == 0 || x1, "conv2d_fwd_skip: null second source");
  PDAE_CHECK_ARG(skip_ok_impl(d, ds, d_dflag), "conv2d_fwd_skip: shape not eligible (pdae_conv2d_fwd_skip_ok == 0)");
  PatchSkip sk{s0, ds->C1 ? s1 : nullptr, ds->C0, ds->C1, (const unsigned short*)wps, bias_s};
  return conv3x3p_launch(d->math | d_dflag, x0, d->N, d->Hi, d->Wi, d->C0 + d->C1, d->Ho, d->Wo, d->up, (const unsigned short*)wp, d->Cout, y, bias, nullptr, 0,
                         0, S(stream), d->C1 ? x1 : nullptr, d->C0, coef, act, &sk, nullptr, stat);
}

// ---- GroupNorm statistics of a convolution's output, produced by the convolution itself
extern "C" size_t pdae_conv_stats_bytes(const pdae_conv_desc* d, const pdae_conv_desc* ds, int32_t* tiles_per_image) {
  PDAE_DESC_NORM(d)
  PDAE_DESC_NORM(ds)
  if (tiles_per_image) *tiles_per_image = 0;
  if (!d || check_desc(d)) return 0;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || ((d->C0 + d->C1) & 31) || (d->Cout & 3)) return 0;
  if (ds ? !skip_ok_impl(d, ds, d_dflag) : (fast_kind(d, 0, false) != 3 && !gn_patch_ok(d, false))) return 0;
  int tpi = 0;
  const size_t b = conv3x3p_stats_bytes(d->math | d_dflag, d->C0 + d->C1, d->Ho, d->Wo, d->N, d->Cout, ds ? (ds->C0 + ds->C1) >> 5 : 0, &tpi);
  if (tiles_per_image) *tiles_per_image = b ? tpi : 0;
  return b;
}
extern "C" int pdae_conv_stats_arm(float* part) {
  conv3x3p_arm_stats(part);
  return PDAE_OK;
}
extern "C" int pdae_gn_coef_from_conv_stats(int N, int HW, int C0, int C1, int G, float eps, const float* part0, int tpi0, const float* part1, int tpi1,
                                            const float* gamma, const float* beta, const float* ss, const float* zss, float* mean, float* rstd,
                                            float* coef, pdae_stream_t stream) {
  PDAE_CHECK_ARG(part0 && tpi0 > 0 && (C1 == 0 || (part1 && tpi1 > 0)) && gamma && beta && mean && rstd && coef, "gn_coef_from_conv_stats: null pointer");
  PDAE_CHECK_ARG(G > 0 && G <= 64 && (C0 + C1) % G == 0 && (((C0 + C1) / G) & 3) == 0 && (C0 & 3) == 0,
                 "gn_coef_from_conv_stats: groups must be whole channel quads (C / G and C0 multiples of 4)");
  return k_gn_coef_from_conv_stats(N, HW, C0, C1, G, eps, part0, tpi0, C1 ? part1 : nullptr, tpi1, gamma, beta, ss, zss, mean, rstd, coef, S(stream));
}

// ---- GroupNorm-backward sums from the data gradient that produces dA (conv3x3y.hip, GB instantiation): the reduction pass of pdae_gn_bwd over
// (x, dA) -- 2 of its 5 tensor passes -- is replaced by an epilogue of the launch that writes dA.
extern "C" size_t pdae_conv_gnbwd_bytes(const pdae_conv_desc* d, int flags, int32_t* tiles_per_image) {
  PDAE_DESC_NORM(d)
  (void)d_dflag;
  if (tiles_per_image) *tiles_per_image = 0;
  if (!d || check_desc(d) || d->up || d->stride != 1 || fast_kind(d, 1, false) != 3) return 0;
  const int t = conv3x3p_gnb_tiles(bwd_math(d, flags & PDAE_WPREP_F16_GRAD), d->Cout, d->Hi, d->Wi, d->N, d->C0 + d->C1, d->C0, d->C1);
  if (tiles_per_image) *tiles_per_image = t;
  return (size_t)d->N * t * (d->C0 + d->C1) * 2 * sizeof(float);
}
static thread_local PatchGnb g_gnb_arm = {nullptr, nullptr, 0, 0, nullptr, nullptr};
extern "C" int pdae_conv_gnbwd_arm(const float* x0, int C0, const float* x1, int C1, const float* coef, int act, float* part) {
  PDAE_CHECK_ARG(x0 && coef && part && C0 > 0 && C1 >= 0 && (C1 == 0 || x1), "conv_gnbwd_arm: null pointer");
  PDAE_CHECK_ARG(act == 1, "conv_gnbwd_arm: only the SiLU form (act = 1, no dropout) is built");
  g_gnb_arm = PatchGnb{x0, C1 ? x1 : nullptr, C0, C1, coef, part};
  return PDAE_OK;
}
static bool take_gnb(PatchGnb* out) { *out = g_gnb_arm; g_gnb_arm.part = nullptr; return out->part != nullptr; }
static thread_local const float* g_gnparts = nullptr;
static thread_local int g_gnparts_tiles = 0;
extern "C" int pdae_gn_bwd_parts_arm(const float* part, int tiles_per_image) {
  PDAE_CHECK_ARG(part && tiles_per_image > 0 && tiles_per_image <= 64, "gn_bwd_parts_arm: bad arguments (1..64 tiles per image)");
  g_gnparts = part; g_gnparts_tiles = tiles_per_image;
  return PDAE_OK;
}

extern "C" int pdae_conv2d_dgrad(const pdae_conv_desc* d, const float* dy, const float* w, const void* wp_t, float* dx, int ci_off, int ci_cnt,
                                 int accumulate, int tile, const float* dy_amax, pdae_stream_t stream) {
  PDAE_DESC_NORM(d)
  conv3x3p_take_stats();                  // output statistics belong to forward convolutions only
  PatchGnb gnb;
  const bool want_gnb = take_gnb(&gnb);   // consumed here whatever happens below
  if (int e = check_desc(d)) return e;
  const int Cin = d->C0 + d->C1;
  PDAE_CHECK_ARG(dy && w && dx && ci_off >= 0 && ci_cnt > 0 && ci_off + ci_cnt <= Cin, "conv2d_dgrad: bad arguments");
  const int Hl = d->up ? 2 * d->Hi : d->Hi, Wl = d->up ? 2 * d->Wi : d->Wi;
  PDAE_CHECK_ARG(d->stride == 1 || (Hl == 2 * d->Ho && Wl == 2 * d->Wo), "conv2d_dgrad: stride 2 needs even input");
  if (tile == 0 && !wp_t && ci_off == 0 && ci_cnt == Cin && edge_on() && edge_in_ok(d->KH, d->KW, d->stride, d->pad, d->up, d->C1, d->Cout, Cin))
    return edge_in_conv(dy, d->N, d->Ho, d->Wo, d->Cout, w, 1, Cin, nullptr, dx, accumulate, S(stream));
  const int kind = wp_t ? fast_kind(d, 1, false) : 0;
  PDAE_CHECK_ARG(!wp_t || (tile == 0 && ((kind == 3 && ci_off == 0 && ci_cnt == Cin) || (kind == 1 && (ci_off & 31) == 0 && (ci_cnt & 3) == 0))),
                 "conv2d_dgrad: wp_t given but the convolution / channel range is not eligible for a prepared-weight kernel");
  PDAE_CHECK_ARG(!want_gnb || (kind == 3 && !d->up && gnb.C0 == d->C0 && gnb.C1 == d->C1),
                 "conv2d_dgrad: GroupNorm-backward sums were armed (pdae_conv_gnbwd_arm) but this data gradient cannot leave them (pdae_conv_gnbwd_bytes == 0, "
                 "or the armed channel split differs from the descriptor's C0 / C1)");
  if (kind == 3)
    return conv3x3p_launch(bwd_math(d, dy_amax != nullptr), dy, d->N, d->Ho, d->Wo, d->Cout, Hl, Wl, 0, (const unsigned short*)wp_t, Cin, dx, nullptr,
                           nullptr, 0, accumulate, S(stream), nullptr, 0, nullptr, 0, nullptr, dy_amax, nullptr, want_gnb ? &gnb : nullptr);
  if (kind == 1)
    return conv1x1_launch(bwd_math(d, dy_amax != nullptr), dy, d->Cout, nullptr, 0, (long long)d->N * d->Ho * d->Wo, (const unsigned short*)wp_t, Cin,
                          ci_off, ci_cnt, dx, nullptr, nullptr, 0, d->Ho, d->Wo, accumulate, S(stream), dy_amax);
  GemmParams P;
  memset(&P, 0, sizeof(P));
  ConvGeom& g = P.a.g;
  g.src0 = dy; g.src1 = nullptr; g.C0 = d->Cout; g.C1 = 0; g.Cin = d->Cout;
  g.Hs = d->Ho; g.Ws = d->Wo; g.dil = d->stride == 2; g.up = 0;
  g.Hl = g.dil ? 2 * d->Ho : d->Ho; g.Wl = g.dil ? 2 * d->Wo : d->Wo;
  g.Ho = Hl; g.Wo = Wl; g.KH = d->KH; g.KW = d->KW; g.stride = 1; g.pad = d->KH - 1 - d->pad;
  PDAE_CHECK_ARG(d->KH == d->KW, "conv2d_dgrad: square kernels only");
  P.M = d->N * Hl * Wl; P.N = ci_cnt; P.K = d->KH * d->KW * d->Cout;
  P.splitk = 1; P.kchunk = P.K; P.Bi = 1;
  P.b.p = w; P.b.dgT = d->KH * d->KW; P.b.dgCout = d->Cout; P.b.dgWCin = Cin; P.b.dgCiOff = ci_off;
  P.C = dx; P.ldc = ci_cnt; P.alpha = 1.0f; P.accumulate = accumulate;
  return igemm_conv_dgrad(P, tile, d->math, S(stream));
}

static void wgrad_plan(const pdae_conv_desc* d, int& tile, int& splits, int& kchunk) {
  const long long M = d->Cout, N = (long long)d->KH * d->KW * (d->C0 + d->C1), K = (long long)d->N * d->Ho * d->Wo;
  tile = (M >= 96 && N >= 96) ? 128 : 64;
  long long tiles = (long long)cdiv(M, tile) * cdiv(N, tile);
  long long want = cdiv(768, tiles);
  long long maxs = K / 256; if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want > 256) want = 256;
  kchunk = (int)(((K + want - 1) / want + 31) / 32 * 32);
  splits = cdiv(K, kchunk);
}

// ---- weight gradient of a convolution whose forward applied GroupNorm (+ SiLU) to its raw input inside the staging (pdae_conv2d_fwd_gn):
// the activated tensor was never written, so the weight gradient recomputes it the same way (conv3x3w.hip, GN instantiation).
static bool wgrad_gn_ok(const pdae_conv_desc* d) {
  return conv3x3w_gn_ok(d->math, d->KH, d->KW, d->stride, d->pad, d->C0, d->C1, d->Ho, d->Wo, d->N, d->Cout);
}
extern "C" int pdae_conv2d_wgrad_gn_ok(const pdae_conv_desc* d) {
  PDAE_DESC_NORM(d)
  (void)d_dflag;
  return (d && !check_desc(d) && wgrad_gn_ok(d)) ? 1 : 0;
}
// One-shot request, as pdae_conv_stats_arm: the next pdae_conv2d_wgrad on this host thread TAKES it on entry (whatever happens afterwards).
static thread_local const float* g_wgn_coef = nullptr;
static thread_local int g_wgn_act = 0;
extern "C" int pdae_conv_gn_input_arm(const float* coef, int act) {
  PDAE_CHECK_ARG(coef && (act == 0 || act == 1), "conv_gn_input_arm: bad arguments");
  g_wgn_coef = coef; g_wgn_act = act;
  return PDAE_OK;
}
static const float* take_wgn(int* act) { const float* c = g_wgn_coef; *act = g_wgn_act; g_wgn_coef = nullptr; return c; }

// workspace of the weight gradient proper (slabs / partials), 256-byte aligned; the bias-gradient column-sum scratch follows it
static size_t wgrad_path_bytes(const pdae_conv_desc* d) {
  size_t b;
  if (convhead_ok(d->KH, d->KW, d->stride, d->pad, d->up, d->C1, d->C0, d->Cout))
    b = convhead_wgrad_workspace_bytes(d->N, d->Hi, d->Wi, d->C0, d->Cout);
  else if (conv3x3w_ok(d->math, d->KH, d->KW, d->stride, d->pad, d->C1, d->C0, d->Ho, d->Wo, d->N, d->Cout))
    b = conv3x3w_workspace_bytes(d->N, d->Ho, d->Wo, d->C0, d->Cout);
  else {
    int tile, splits, kchunk;
    wgrad_plan(d, tile, splits, kchunk);
    b = splits <= 1 ? 16 : (size_t)splits * d->Cout * d->KH * d->KW * (d->C0 + d->C1) * sizeof(float);
    // the dedicated 1x1 kernel runs when the launch also brings what its arithmetic needs (math 4: dy_amax); size for whichever path is larger
    if (conv1x1w_ok(d->math, d->KH, d->KW, d->stride, d->pad, d->up, d->C0, d->C1, (long long)d->N * d->Ho * d->Wo, d->Cout)) {
      const size_t b1 = conv1x1w_workspace_bytes((long long)d->N * d->Ho * d->Wo, d->C0 + d->C1, d->Cout);
      if (b1 > b) b = b1;
    }
    if (d->C1 && wgrad_gn_ok(d)) {                      // two-source 3x3: the launch may arrive with pdae_conv_gn_input_arm
      const size_t b3 = conv3x3w_workspace_bytes(d->N, d->Ho, d->Wo, d->C0 + d->C1, d->Cout);
      if (b3 > b) b = b3;
    }
  }
  return (b + 255) & ~(size_t)255;
}

// (ABI 11, informational) which kernel pdae_conv2d_wgrad(d, ...) runs on: 3 = conv3x3v (producer / consumer form), 2 = conv3x3w, 1 = another
// dedicated kernel (1x1 weight gradient, edge / head layers), 0 = the generic implicit GEMM.  with_dy_amax / with_gn_input: as the launch will be
// made (the fp16 format needs the dY scale; a fused GroupNorm input joins the two sources).  Follows the knob PDAE_W3V like the launch does.
extern "C" int pdae_conv2d_wgrad_form(const pdae_conv_desc* d, int with_dy_amax, int with_gn_input) {
  PDAE_DESC_NORM(d)
  (void)d_dflag;
  if (!d || check_desc(d)) return 0;
  const long long Mpix = (long long)d->N * d->Ho * d->Wo;
  if (!with_gn_input) {
    if (edge_on() && edge_head_wgrad_ok(d->KH, d->KW, d->stride, d->pad, d->up, d->C1, d->C0, d->Cout)) return 1;
    if (convhead_ok(d->KH, d->KW, d->stride, d->pad, d->up, d->C1, d->C0, d->Cout)) return 1;
  }
  const bool three = with_gn_input ? wgrad_gn_ok(d) : conv3x3w_ok(d->math, d->KH, d->KW, d->stride, d->pad, d->C1, d->C0, d->Ho, d->Wo, d->N, d->Cout);
  if (three) {
    int math = d->math;
    if (math == 4 && !with_dy_amax) math = 3;
    return (pdae_knob(KNOB_W3V) && conv3x3v_ok(math, d->C0 + d->C1, d->Ho, d->Wo, d->N, d->Cout)) ? 3 : 2;
  }
  if (conv1x1w_ok(d->math, d->KH, d->KW, d->stride, d->pad, d->up, d->C0, d->C1, Mpix, d->Cout) && d->math != 3 && (d->math != 4 || with_dy_amax)) return 1;
  return 0;
}

extern "C" size_t pdae_conv2d_wgrad_workspace_bytes(const pdae_conv_desc* d) {
  PDAE_DESC_NORM(d)
  return wgrad_path_bytes(d) + k_colsum_workspace_floats((long long)d->N * d->Ho * d->Wo, d->Cout) * sizeof(float);
}

extern "C" int pdae_conv2d_wgrad(const pdae_conv_desc* d, const float* x0, const float* x1, const float* dy, float* dw, float* db, int accumulate,
                                 void* ws, size_t ws_bytes, const float* dy_amax, pdae_stream_t stream) {
  PDAE_DESC_NORM(d)
  int gn_act = 0;
  const float* const gn_coef = take_wgn(&gn_act);        // consumed here: a failed call never leaves the request armed
  if (int e = check_desc(d)) return e;
  PDAE_CHECK_ARG(x0 && dy && dw && (d->C1 == 0 || x1), "conv2d_wgrad: null pointer");
  const size_t pb = wgrad_path_bytes(d);
  const long long Mpix = (long long)d->N * d->Ho * d->Wo;
  if (gn_coef) {
    PDAE_CHECK_ARG(wgrad_gn_ok(d), "conv2d_wgrad: a GroupNorm input was armed (pdae_conv_gn_input_arm) but pdae_conv2d_wgrad_gn_ok(d) == 0");
    PDAE_CHECK_ARG(!db || (ws && ws_bytes >= pb + k_colsum_workspace_floats(Mpix, d->Cout) * sizeof(float)),
                   "conv2d_wgrad: workspace too small for the bias gradient (%zu < pdae_conv2d_wgrad_workspace_bytes)", ws_bytes);
    float* part = nullptr; int rows = 0;
    if (int e = conv3x3w_launch(d->math, x0, d->N, d->Hi, d->Wi, d->C0 + d->C1, d->Ho, d->Wo, d->up, dy, d->Cout, dw, accumulate, (float*)ws,
                                ws_bytes < pb ? ws_bytes : pb, S(stream), db ? &part : nullptr, db ? &rows : nullptr, dy_amax, db, d->C1 ? x1 : nullptr,
                                d->C0, gn_coef, gn_act))
      return e;
    return (db && part) ? k_colsum(part, rows, d->Cout, db, accumulate, ws ? (float*)((char*)ws + pb) : nullptr, S(stream)) : PDAE_OK;
  }
  PDAE_CHECK_ARG(!db || (ws && ws_bytes >= pb + k_colsum_workspace_floats(Mpix, d->Cout) * sizeof(float)),
                 "conv2d_wgrad: workspace too small for the bias gradient (%zu < pdae_conv2d_wgrad_workspace_bytes)", ws_bytes);
  float* cws = ws ? (float*)((char*)ws + pb) : nullptr;                 // column-sum scratch
  const size_t wsb = ws_bytes < pb ? ws_bytes : pb;                     // what the weight-gradient path may use
  if (edge_on() && edge_head_wgrad_ok(d->KH, d->KW, d->stride, d->pad, d->up, d->C1, d->C0, d->Cout)) {
    if (int e = edge_head_wgrad(x0, d->N, d->Hi, d->Wi, d->C0, dy, d->Cout, dw, accumulate, (float*)ws, wsb, S(stream))) return e;
    return db ? k_colsum(dy, Mpix, d->Cout, db, accumulate, cws, S(stream)) : PDAE_OK;
  }
  if (convhead_ok(d->KH, d->KW, d->stride, d->pad, d->up, d->C1, d->C0, d->Cout)) {
    if (int e = convhead_wgrad(x0, d->N, d->Hi, d->Wi, d->C0, dy, d->Cout, dw, accumulate, (float*)ws, wsb, S(stream))) return e;
    return db ? k_colsum(dy, Mpix, d->Cout, db, accumulate, cws, S(stream)) : PDAE_OK;
  }
  if (conv3x3w_ok(d->math, d->KH, d->KW, d->stride, d->pad, d->C1, d->C0, d->Ho, d->Wo, d->N, d->Cout)) {
    float* part = nullptr; int rows = 0;                                  // bias gradient: column sums of dY fall out of the dY staging
    if (int e = conv3x3w_launch(d->math, x0, d->N, d->Hi, d->Wi, d->C0, d->Ho, d->Wo, d->up, dy, d->Cout, dw, accumulate, (float*)ws, wsb, S(stream),
                                db ? &part : nullptr, db ? &rows : nullptr, dy_amax, db))
      return e;
    return (db && part) ? k_colsum(part, rows, d->Cout, db, accumulate, cws, S(stream)) : PDAE_OK;      // part == NULL: summed inside the reduce launch
  }
  // three-plane formats do not fit the two-operand LDS tiles of the 1x1 kernel: math 3, and math 4 without a dY scale, stay on the generic kernel
  if (conv1x1w_ok(d->math, d->KH, d->KW, d->stride, d->pad, d->up, d->C0, d->C1, Mpix, d->Cout) && d->math != 3 && (d->math != 4 || dy_amax)) {
    float* part = nullptr; int rows = 0;
    if (int e = conv1x1w_launch(d->math, x0, d->C0, x1, d->C1, Mpix, dy, d->Cout, dw, accumulate, (float*)ws, wsb, S(stream), db ? &part : nullptr,
                                db ? &rows : nullptr, dy_amax, db))
      return e;
    return (db && part) ? k_colsum(part, rows, d->Cout, db, accumulate, cws, S(stream)) : PDAE_OK;
  }
  int tile, splits, kchunk;
  wgrad_plan(d, tile, splits, kchunk);
  GemmParams P;
  memset(&P, 0, sizeof(P));
  fwd_geom(P.b.g, d, x0, x1);
  P.M = d->Cout; P.N = d->KH * d->KW * (d->C0 + d->C1); P.K = d->N * d->Ho * d->Wo;
  P.a.p = dy; P.a.ld = d->Cout;
  P.Bi = 1; P.alpha = 1.0f;
  const long long MN = (long long)P.M * P.N;
  if (splits <= 1) {
    P.splitk = 1; P.kchunk = P.K; P.C = dw; P.ldc = P.N; P.accumulate = accumulate;
    if (int e = igemm_conv_wgrad(P, tile, 1, d->math, S(stream))) return e;
  } else {
    PDAE_CHECK_ARG(ws && wsb >= (size_t)splits * MN * sizeof(float), "conv2d_wgrad: workspace too small (%zu < %zu)", wsb,
                   (size_t)splits * MN * sizeof(float));
    P.splitk = splits; P.kchunk = kchunk; P.split_stride = MN; P.C = (float*)ws; P.ldc = P.N; P.accumulate = 0;
    if (int e = igemm_conv_wgrad(P, tile, splits, d->math, S(stream))) return e;
    if (int e = igemm_splitk_reduce((const float*)ws, dw, MN, splits, accumulate, S(stream))) return e;
  }
  return db ? k_colsum(dy, Mpix, d->Cout, db, accumulate, cws, S(stream)) : PDAE_OK;
}

extern "C" int pdae_gemm(int transA, int transB, int M, int N, int K, float alpha, const float* A, int64_t lda, int64_t sAo, int64_t sAi,
                         const float* B, int64_t ldb, int64_t sBo, int64_t sBi, float* C, int64_t ldc, int64_t sCo, int64_t sCi, int batch_outer,
                         int batch_inner, const float* bias, int accumulate, pdae_stream_t stream) {
  PDAE_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && batch_outer > 0 && batch_inner > 0, "gemm: bad arguments");
  if (skinny_ok(transA, transB, M, N, K, alpha, lda, ldb, A, B, batch_outer * batch_inner))
    return skinny_launch(A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate, S(stream));
  GemmParams P;
  memset(&P, 0, sizeof(P));
  P.M = M; P.N = N; P.K = K; P.splitk = 1; P.kchunk = K; P.Bi = batch_inner;
  P.a.p = A; P.a.ld = lda; P.a.so = sAo; P.a.si = sAi;
  P.b.p = B; P.b.ld = ldb; P.b.so = sBo; P.b.si = sBi;
  P.C = C; P.ldc = ldc; P.sCo = sCo; P.sCi = sCi; P.bias = bias; P.alpha = alpha; P.accumulate = accumulate;
  return igemm_dense(transA, transB, P, batch_outer * batch_inner, S(stream));
}

extern "C" int pdae_linear_group(const pdae_linear_item* items, const int32_t* first_feature, int n_items, int total_features, int M, int K,
                                 pdae_stream_t stream) {
  PDAE_CHECK_ARG(items && first_feature && n_items > 0 && total_features > 0 && M > 0 && M <= 32 && K > 0 && (K & 7) == 0,
                 "linear_group: bad arguments (M <= 32, K %% 8 == 0)");
  return skinny_group_launch(items, first_feature, n_items, total_features, M, K, S(stream));
}

extern "C" int pdae_linear_bwd_group(const pdae_linear_bwd_item* items, const int32_t* first_block, int n_items, int total_blocks, int M, int K,
                                     pdae_stream_t stream) {
  PDAE_CHECK_ARG(items && first_block && n_items > 0 && total_blocks > 0 && M > 0 && M <= 32 && K > 0 && (K & 3) == 0,
                 "linear_bwd_group: bad arguments (M <= 32, K %% 4 == 0)");
  return linear_bwd_group_launch(items, first_block, n_items, total_blocks, M, K, S(stream));
}

// ---- GroupNorm family
extern "C" size_t pdae_gn_workspace_bytes(int N, int C) { return k_gn_workspace_floats(N, C) * sizeof(float); }
extern "C" int pdae_gn_stats(const float* x0, int C0, const float* x1, int C1, int N, int HW, int G, float eps, float* mean, float* rstd, void* ws,
                             pdae_stream_t stream) {
  PDAE_CHECK_ARG(x0 && mean && rstd && ws && (C1 == 0 || x1), "gn_stats: null pointer");
  return k_gn_stats(x0, C0, x1, C1, N, HW, G, eps, mean, rstd, (float*)ws, S(stream));
}
extern "C" int pdae_gn_stats_quads(const float* x, int N, int HW, int C, int tiles_per_image, float* part, pdae_stream_t stream) {
  return k_gn_stats_quads(x, N, HW, C, tiles_per_image, part, S(stream));
}
extern "C" int pdae_gn_stats_coef(const float* x0, int C0, const float* x1, int C1, int N, int HW, int G, float eps, const float* gamma,
                                  const float* beta, const float* ss, const float* zss, float* mean, float* rstd, float* coef, void* ws,
                                  uint32_t* ticket, pdae_stream_t stream) {
  PDAE_CHECK_ARG(x0 && gamma && beta && mean && rstd && coef && ws && (C1 == 0 || x1) && (C0 + C1) % G == 0, "gn_stats_coef: bad arguments");
  return k_gn_stats_coef(x0, C0, x1, C1, N, HW, G, eps, gamma, beta, ss, zss, mean, rstd, coef, (float*)ws, S(stream), ticket);
}
extern "C" int pdae_gn_coef(int N, int C, int G, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* ss,
                            const float* zss, float* coef, pdae_stream_t stream) {
  PDAE_CHECK_ARG(mean && rstd && gamma && beta && coef && C % G == 0, "gn_coef: bad arguments");
  return k_gn_coef(N, C, G, mean, rstd, gamma, beta, ss, zss, coef, S(stream));
}
extern "C" int pdae_gn_apply(const float* x0, int C0, const float* x1, int C1, int N, int H, int W, const float* coef, int act, int mode, float* y,
                             float* xpool, float drop_p, uint64_t seed, uint64_t offset, pdae_stream_t stream) {
  PDAE_CHECK_ARG(x0 && coef && y && (C1 == 0 || x1), "gn_apply: null pointer");
  return k_gn_apply(x0, C0, x1, C1, N, H, W, coef, act, mode, y, xpool, drop_p, seed, offset, S(stream));
}
extern "C" int pdae_gn_bwd(const float* x0, int C0, const float* x1, int C1, int N, int H, int W, int G, const float* coef, const float* rstd,
                           const float* gamma, const float* beta, const float* ss, const float* zss, const float* dA, int act, int mode,
                           float drop_p, uint64_t seed, uint64_t offset, const float* add, float* dx0, int acc0, float* dx1, int acc1,
                           float* dgamma, float* dbeta, int acc_param, float* dss, float* dzss, void* ws, float* dx0_amax, uint32_t* ticket,
                           pdae_stream_t stream) {
  const float* const parts = g_gnparts; const int parts_tiles = g_gnparts_tiles;      // one-shot (pdae_gn_bwd_parts_arm): consumed here
  g_gnparts = nullptr; g_gnparts_tiles = 0;
  PDAE_CHECK_ARG(x0 && coef && rstd && gamma && beta && dA && ws && (C1 == 0 || x1), "gn_bwd: null pointer");
  PDAE_CHECK_ARG(!parts || (mode == 0 && act == 1 && drop_p == 0.f && !ticket), "gn_bwd: externally computed sums need mode 0, SiLU, no dropout");
  PDAE_CHECK_ARG(mode >= 0 && mode <= 2 && (mode != 1 || ((H % 2) == 0 && (W % 2) == 0)), "gn_bwd: bad mode");
  PDAE_CHECK_ARG((!dss || ss) && (!dzss || zss) && (!dgamma || dbeta), "gn_bwd: gradient requested for an absent input");
  return k_gn_bwd(x0, C0, x1, C1, N, H, W, G, coef, rstd, gamma, beta, ss, zss, dA, act, mode, drop_p, seed, offset, add, dx0, acc0, dx1, acc1,
                  dgamma, dbeta, acc_param, dss, dzss, (float*)ws, S(stream), dx0_amax, ticket, parts, parts_tiles);
}

// ---- elementwise
extern "C" int pdae_timestep_embedding(const int64_t* t, const float* freqs, int N, int dim, float* out, pdae_stream_t stream) {
  PDAE_CHECK_ARG(t && freqs && out && N > 0 && dim > 1, "timestep_embedding: bad arguments");
  return k_timestep_embedding((const long long*)t, freqs, N, dim, out, S(stream));
}
extern "C" int pdae_mlp_modln_fwd(const float* u, const float* e, const float* gamma, const float* beta, int R, int C, int norm, int act, float eps,
                                  float* y, float* mean, float* rstd, pdae_stream_t stream) {
  return k_mlp_modln_fwd(u, e, gamma, beta, R, C, norm, act, eps, y, mean, rstd, S(stream));
}
extern "C" int pdae_mlp_modln_bwd(const float* u, const float* e, const float* gamma, const float* beta, const float* mean, const float* rstd,
                                  const float* dy, int R, int C, int norm, int act, float* du, float* de, float* tg, float* tb,
                                  pdae_stream_t stream) {
  return k_mlp_modln_bwd(u, e, gamma, beta, mean, rstd, dy, R, C, norm, act, du, de, tg, tb, S(stream));
}
extern "C" int pdae_amax(const float* x, size_t n, float* out, pdae_stream_t stream) {
  PDAE_CHECK_ARG(x && out && n > 0 && (((uintptr_t)x) & 15) == 0, "amax: bad arguments (x must be 16-byte aligned)");
  return k_amax(x, n, out, S(stream));
}
extern "C" int pdae_silu(const float* x, float* y, size_t n, pdae_stream_t stream) { return k_silu(x, y, n, S(stream)); }
extern "C" int pdae_subsample2(const float* x, int N, int H, int W, int C, float* y, pdae_stream_t stream) {
  PDAE_CHECK_ARG(x && y && N >= 0 && H > 0 && W > 0 && !(H & 1) && !(W & 1) && C > 0 && !(C & 3), "subsample2: need even H, W and C %% 4 == 0");
  return k_subsample2(x, N, H, W, C, y, S(stream));
}
extern "C" int pdae_zero_insert2(const float* x, int N, int Ho, int Wo, int C, float* y, pdae_stream_t stream) {
  PDAE_CHECK_ARG(x && y && N >= 0 && Ho > 0 && Wo > 0 && C > 0 && !(C & 3), "zero_insert2: need C %% 4 == 0");
  return k_zero_insert2(x, N, Ho, Wo, C, y, S(stream));
}
extern "C" int pdae_silu_bwd(const float* x, const float* dy, float* dx, size_t n, int acc, pdae_stream_t stream) {
  return k_silu_bwd(x, dy, dx, n, acc, S(stream));
}
extern "C" int pdae_axpby(const float* x, float* y, size_t n, float alpha, float beta, pdae_stream_t stream) {
  return k_axpby(x, y, n, alpha, beta, S(stream));
}
extern "C" int pdae_embedding(const float* table, const int64_t* idx, int N, int D, float* out, int acc, pdae_stream_t stream) {
  return k_embedding(table, (const long long*)idx, N, D, out, acc, S(stream));
}
extern "C" int pdae_embedding_bwd(const float* dout, const int64_t* idx, int N, int D, float* dtable, pdae_stream_t stream) {
  return k_embedding_bwd(dout, (const long long*)idx, N, D, dtable, S(stream));
}
extern "C" int pdae_to_nhwc(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int N, int C, int H, int W, float* y,
                            pdae_stream_t stream) {
  return k_to_nhwc(x, sn, sc, sh, sw, N, C, H, W, y, S(stream));
}
extern "C" int pdae_from_nhwc(const float* x, int N, int C, int H, int W, float* y, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                              pdae_stream_t stream) {
  return k_from_nhwc(x, N, C, H, W, y, sn, sc, sh, sw, S(stream));
}
extern "C" int pdae_softmax(float* s, int64_t rows, int T, pdae_stream_t stream) { return k_softmax(s, rows, T, S(stream)); }
extern "C" int pdae_softmax_bwd(const float* p, float* dp, int64_t rows, int T, pdae_stream_t stream) {
  return k_softmax_bwd(p, dp, rows, T, S(stream));
}
extern "C" size_t pdae_colsum_workspace_bytes(int64_t M, int C) { return k_colsum_workspace_floats(M, C) * sizeof(float); }
extern "C" int pdae_colsum(const float* x, int64_t M, int C, float* out, int acc, void* ws, pdae_stream_t stream) {
  PDAE_CHECK_ARG(x && out && ws && M > 0 && C > 0, "colsum: bad arguments");
  return k_colsum(x, M, C, out, acc, (float*)ws, S(stream));
}
extern "C" int pdae_q_sample(const float* x0, const float* noise, const int64_t* t, const float* ta, const float* tb, int N, size_t per, float* xt,
                             pdae_stream_t stream) {
  PDAE_CHECK_ARG(x0 && noise && t && ta && tb && xt, "q_sample: null pointer");
  return k_q_sample(x0, noise, (const long long*)t, ta, tb, N, per, xt, S(stream));
}
extern "C" int pdae_loss(const float* noise, const float* eps, const float* g, const int64_t* t, const float* tc, const float* tw, int N, size_t per,
                         int l1, float scale, float* loss, float* deps, float* dg, void* ws, pdae_stream_t stream) {
  PDAE_CHECK_ARG(noise && eps && loss && ws && (!g || (t && tc)) && (!tw || t), "loss: bad arguments");
  return k_loss(noise, eps, g, (const long long*)t, tc, tw, N, per, l1, scale, loss, deps, dg, (float*)ws, S(stream));
}
extern "C" int pdae_ddim_step(const float* x, const float* eps, const float* g, size_t total, float c_shift, float ra, float rm1, float sab,
                              float s1ab, int clamp, float* out, pdae_stream_t stream) {
  PDAE_CHECK_ARG(x && eps && out, "ddim_step: null pointer");
  return k_ddim_step(x, eps, g, total, c_shift, ra, rm1, sab, s1ab, clamp, out, S(stream));
}
extern "C" int pdae_ddpm_step(const float* x, const float* eps, const float* g, const float* z, size_t total, float cx, float ce, float cs,
                              float sigma, float* out, pdae_stream_t stream) {
  PDAE_CHECK_ARG(x && eps && out, "ddpm_step: null pointer");
  return k_ddpm_step(x, eps, g, z, total, cx, ce, cs, sigma, out, S(stream));
}
extern "C" int pdae_axpby_rows(const float* a, const float* b, const float* ca, const float* cb, int N, size_t per, float* out, pdae_stream_t stream) {
  PDAE_CHECK_ARG(a && b && ca && cb && out && N > 0 && N < 65536, "axpby_rows: bad arguments");
  return k_axpby_rows(a, b, ca, cb, N, per, out, S(stream));
}
extern "C" int pdae_ddim_step_rows(const float* x, const float* eps, const float* g, const float* coef, int N, size_t per, int clamp, float* out,
                                   pdae_stream_t stream) {
  PDAE_CHECK_ARG(x && eps && coef && out && N > 0 && N < 65536, "ddim_step_rows: bad arguments");
  return k_ddim_step_rows(x, eps, g, coef, N, per, clamp, out, S(stream));
}
extern "C" int pdae_ddpm_step_rows(const float* x, const float* eps, const float* g, const float* noise, const float* learned_range, const float* coef,
                                   int N, size_t per, float* out, pdae_stream_t stream) {
  PDAE_CHECK_ARG(x && eps && coef && out && N > 0 && N < 65536, "ddpm_step_rows: bad arguments");
  return k_ddpm_step_rows(x, eps, g, noise, learned_range, coef, N, per, out, S(stream));
}
extern "C" int pdae_adam_ema(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float b1, float b2, float eps, float wd,
                             int decoupled, float step_size, float inv_sqrt_bc2, float grad_scale, float ema_decay, unsigned int* guard, int count_skip,
                             pdae_stream_t stream) {
  PDAE_CHECK_ARG(p && g && m && v, "adam_ema: null pointer");
  return k_adam_ema(p, g, m, v, ema, n, lr, b1, b2, eps, wd, decoupled, step_size, inv_sqrt_bc2, grad_scale, ema_decay, guard, count_skip, S(stream));
}

// ---- data-parallel gradient exchange (RCCL bound at run time, comm.hip)
extern "C" int pdae_comm_unique_id(const char* librccl_path, void* id128) {
  PDAE_CHECK_ARG(id128, "comm_unique_id: null pointer");
  return k_comm_unique_id(librccl_path, id128);
}
extern "C" int pdae_comm_init(const char* librccl_path, const void* id128, int nranks, int rank, void** comm) {
  PDAE_CHECK_ARG(id128 && comm && nranks > 0 && rank >= 0 && rank < nranks, "comm_init: bad arguments");
  return k_comm_init(librccl_path, id128, nranks, rank, comm);
}
extern "C" int pdae_allreduce_bucket(void* comm, void* buf, size_t count, int dtype, int op, pdae_stream_t stream) {
  PDAE_CHECK_ARG(comm && buf && count > 0 && (dtype == 0 || dtype == 1) && (op == 0 || op == 1), "allreduce_bucket: bad arguments");
  return k_allreduce(comm, buf, count, dtype, op, S(stream));
}
extern "C" int pdae_comm_destroy(void* comm) { return k_comm_destroy(comm); }

// ---- fused attention core
extern "C" int pdae_attn_fused_ok(int T, int C, int heads) { return (heads > 0 && C % heads == 0 && attn_fused_ok(T, C / heads, C, heads)) ? 1 : 0; }
extern "C" int pdae_attn_fwd(const float* qkv, int N, int T, int C, int heads, int new_order, float* out, float* lse, pdae_stream_t stream) {
  PDAE_CHECK_ARG(qkv && out && N > 0 && N < 65536 && heads > 0 && heads < 65536, "attn_fwd: bad arguments");
  PDAE_CHECK_ARG(pdae_attn_fused_ok(T, C, heads), "attn_fwd: shape not supported by the fused kernel (T in {64,128,192,256}, head width %% 32 == 0): T=%d C=%d heads=%d", T, C, heads);
  return k_attn_fwd(qkv, N, T, C, heads, new_order, out, lse, S(stream));
}
extern "C" int pdae_attn_bwd(const float* qkv, const float* out, const float* lse, const float* d_out, int N, int T, int C, int heads, int new_order,
                             float* dqkv, void* ws, pdae_stream_t stream) {
  PDAE_CHECK_ARG(qkv && out && lse && d_out && dqkv && ws && N > 0 && N < 65536 && heads > 0 && heads < 65536, "attn_bwd: bad arguments");
  PDAE_CHECK_ARG(pdae_attn_fused_ok(T, C, heads), "attn_bwd: shape not supported by the fused kernel: T=%d C=%d heads=%d", T, C, heads);
  return k_attn_bwd(qkv, out, lse, d_out, N, T, C, heads, new_order, dqkv, (float*)ws, S(stream));
}

// ---- evaluator + input pipeline
extern "C" size_t pdae_ssim_mse_workspace_bytes(int N, int C, int H, int W) { return k_ssim_mse_workspace_floats(N, C, H, W) * sizeof(float); }
extern "C" int pdae_ssim_mse(const float* a, const int64_t* a_strides, const float* b, const int64_t* b_strides, int N, int C, int H, int W, float mul,
                             float add, const float* window11, float* ssim, float* mse, void* ws, pdae_stream_t stream) {
  PDAE_CHECK_ARG(a && b && a_strides && b_strides && window11 && ws && (ssim || mse) && N > 0 && N < 65536 && C > 0 && C < 65536 && H > 0 && W > 0,
                 "ssim_mse: bad arguments");
  return k_ssim_mse(a, (const long long*)a_strides, b, (const long long*)b_strides, N, C, H, W, mul, add, window11, ssim, mse, (float*)ws, S(stream));
}
extern "C" size_t pdae_image_prepare_workspace_bytes(int B, int crop_h, int S_out, int C) { return k_image_workspace_bytes(B, crop_h, S_out, C); }
extern "C" int pdae_image_prepare(const uint8_t* src, int B, int Hs, int Ws, int C, int crop_y, int crop_x, int crop_h, int crop_w, int S_out,
                                  const int32_t* coef_x, const int32_t* bounds_x, int ksize_x, const int32_t* coef_y, const int32_t* bounds_y,
                                  int ksize_y, const uint8_t* flip, float* x0, const int64_t* x0_strides, uint8_t* gts, void* ws,
                                  pdae_stream_t stream) {
  PDAE_CHECK_ARG(src && coef_x && bounds_x && coef_y && bounds_y && x0 && x0_strides && ws && B > 0 && C > 0 && S_out > 0, "image_prepare: null pointer");
  PDAE_CHECK_ARG(crop_y >= 0 && crop_x >= 0 && crop_h > 0 && crop_w > 0 && crop_y + crop_h <= Hs && crop_x + crop_w <= Ws, "image_prepare: crop box outside the image");
  return k_image_prepare(src, B, Hs, Ws, C, crop_y, crop_x, crop_h, crop_w, S_out, coef_x, bounds_x, ksize_x, coef_y, bounds_y, ksize_y, flip, x0,
                         (const long long*)x0_strides, gts, (unsigned char*)ws, S(stream));
}

// ---- planned-graph executor
static void desc_from(const int64_t* i, pdae_conv_desc& d) {
  d.N = (int)i[0]; d.Hi = (int)i[1]; d.Wi = (int)i[2]; d.C0 = (int)i[3]; d.C1 = (int)i[4]; d.Ho = (int)i[5]; d.Wo = (int)i[6]; d.Cout = (int)i[7];
  d.KH = (int)i[8]; d.KW = (int)i[9]; d.stride = (int)i[10]; d.pad = (int)i[11]; d.up = (int)i[12]; d.math = (int)i[13];
}

static int run_one(const pdae_op& o, pdae_stream_t st) {
  void* const* p = o.p; const int64_t* i = o.i; const double* f = o.f;
#define F(k) ((const float*)p[k])
#define FM(k) ((float*)p[k])
  pdae_conv_desc d;
  switch (o.kind) {
    case PDAE_OP_CONV_FWD: desc_from(i, d); if (p[19]) conv3x3p_arm_stats((float*)p[19]); return pdae_conv2d_fwd(&d, F(0), F(1), F(2), p[6], F(3), F(4), (int)i[14], FM(5), (int)i[15], st);
    case PDAE_OP_CONV_DGRAD:
      desc_from(i, d);
      if (p[8]) { if (int e = pdae_conv_gnbwd_arm(F(5), (int)i[18], F(6), (int)i[19], F(7), (int)i[20], FM(8))) return e; }
      return pdae_conv2d_dgrad(&d, F(0), F(1), p[3], FM(2), (int)i[14], (int)i[15], (int)i[16], (int)i[17], F(4), st);
    case PDAE_OP_CONV_WGRAD: desc_from(i, d); if (p[7]) pdae_conv_gn_input_arm(F(7), (int)i[16]); return pdae_conv2d_wgrad(&d, F(0), F(1), F(2), FM(3), FM(5), (int)i[14], p[4], (size_t)i[15], F(6), st);
    case PDAE_OP_GEMM:
      return pdae_gemm((int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], (float)f[0], F(0), i[5], i[6], i[7], F(1), i[8], i[9], i[10], FM(2),
                       i[11], i[12], i[13], (int)i[14], (int)i[15], F(3), (int)i[16], st);
    case PDAE_OP_GN_STATS: return pdae_gn_stats(F(0), (int)i[0], F(1), (int)i[1], (int)i[2], (int)i[3], (int)i[4], (float)f[0], FM(2), FM(3), p[4], st);
    case PDAE_OP_GN_STATS_COEF:
      return pdae_gn_stats_coef(F(0), (int)i[0], F(1), (int)i[1], (int)i[2], (int)i[3], (int)i[4], (float)f[0], F(2), F(3), F(4), F(5), FM(6), FM(7),
                                FM(8), p[9], (uint32_t*)p[10], st);
    case PDAE_OP_GN_COEF_FROM_CONV_STATS:
      return pdae_gn_coef_from_conv_stats((int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], (float)f[0], F(0), (int)i[5], F(1), (int)i[6], F(2), F(3),
                                          F(4), F(5), FM(6), FM(7), FM(8), st);
    case PDAE_OP_GN_COEF: return pdae_gn_coef((int)i[0], (int)i[1], (int)i[2], F(0), F(1), F(2), F(3), F(4), F(5), FM(6), st);
    case PDAE_OP_GN_APPLY:
      return pdae_gn_apply(F(0), (int)i[0], F(1), (int)i[1], (int)i[2], (int)i[3], (int)i[4], F(2), (int)i[5], (int)i[6], FM(3), FM(4), (float)f[0],
                           (uint64_t)i[7], (uint64_t)i[8], st);
    case PDAE_OP_GN_BWD:
      if (p[19]) { if (int e = pdae_gn_bwd_parts_arm(F(19), (int)i[13])) return e; }
      return pdae_gn_bwd(F(0), (int)i[0], F(1), (int)i[1], (int)i[2], (int)i[3], (int)i[4], (int)i[5], F(2), F(3), F(4), F(5), F(6), F(7), F(8),
                         (int)i[6], (int)i[7], (float)f[0], (uint64_t)i[11], (uint64_t)i[12], F(9), FM(10), (int)i[8], FM(11), (int)i[9], FM(12),
                         FM(13), (int)i[10], FM(14), FM(15), p[16], FM(17), (uint32_t*)p[18], st);
    case PDAE_OP_TEMB: return pdae_timestep_embedding((const int64_t*)p[0], F(1), (int)i[0], (int)i[1], FM(2), st);
    case PDAE_OP_MLP_MODLN_FWD:
      return pdae_mlp_modln_fwd(F(0), F(1), F(2), F(3), (int)i[0], (int)i[1], (int)i[2], (int)i[3], (float)f[0], FM(4), FM(5), FM(6), st);
    case PDAE_OP_MLP_MODLN_BWD:
      return pdae_mlp_modln_bwd(F(0), F(1), F(2), F(3), F(4), F(5), F(6), (int)i[0], (int)i[1], (int)i[2], (int)i[3], FM(7), FM(8), FM(9), FM(10), st);
    case PDAE_OP_AMAX: return pdae_amax(F(0), (size_t)i[0], FM(1), st);
    case PDAE_OP_SILU: return pdae_silu(F(0), FM(1), (size_t)i[0], st);
    case PDAE_OP_SILU_BWD: return pdae_silu_bwd(F(0), F(1), FM(2), (size_t)i[0], (int)i[1], st);
    case PDAE_OP_AXPBY: return pdae_axpby(F(0), FM(1), (size_t)i[0], (float)f[0], (float)f[1], st);
    case PDAE_OP_EMBEDDING: return pdae_embedding(F(0), (const int64_t*)p[1], (int)i[0], (int)i[1], FM(2), (int)i[2], st);
    case PDAE_OP_EMBEDDING_BWD: return pdae_embedding_bwd(F(0), (const int64_t*)p[1], (int)i[0], (int)i[1], FM(2), st);
    case PDAE_OP_TO_NHWC: return pdae_to_nhwc(F(0), i[0], i[1], i[2], i[3], (int)i[4], (int)i[5], (int)i[6], (int)i[7], FM(1), st);
    case PDAE_OP_FROM_NHWC: return pdae_from_nhwc(F(0), (int)i[4], (int)i[5], (int)i[6], (int)i[7], FM(1), i[0], i[1], i[2], i[3], st);
    case PDAE_OP_Q_SAMPLE: return pdae_q_sample(F(0), F(1), (const int64_t*)p[2], F(3), F(4), (int)i[0], (size_t)i[1], FM(5), st);
    case PDAE_OP_LOSS:
      return pdae_loss(F(0), F(1), F(2), (const int64_t*)p[3], F(4), F(5), (int)i[0], (size_t)i[1], (int)i[2], (float)f[0], FM(6), FM(7), FM(8), p[9], st);
    case PDAE_OP_DDIM_STEP:
      return pdae_ddim_step(F(0), F(1), F(2), (size_t)i[0], (float)f[0], (float)f[1], (float)f[2], (float)f[3], (float)f[4], (int)i[1], FM(3), st);
    case PDAE_OP_DDPM_STEP:
      return pdae_ddpm_step(F(0), F(1), F(2), F(3), (size_t)i[0], (float)f[0], (float)f[1], (float)f[2], (float)f[3], FM(4), st);
    case PDAE_OP_AXPBY_ROWS: return pdae_axpby_rows(F(0), F(1), F(2), F(3), (int)i[0], (size_t)i[1], FM(4), st);
    case PDAE_OP_DDIM_STEP_ROWS: return pdae_ddim_step_rows(F(0), F(1), F(2), F(3), (int)i[0], (size_t)i[1], (int)i[2], FM(4), st);
    case PDAE_OP_DDPM_STEP_ROWS: return pdae_ddpm_step_rows(F(0), F(1), F(2), F(3), F(4), F(5), (int)i[0], (size_t)i[1], FM(6), st);
    case PDAE_OP_ATTN_FWD: return pdae_attn_fwd(F(0), (int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], FM(1), FM(2), st);
    case PDAE_OP_ATTN_BWD: return pdae_attn_bwd(F(0), F(1), F(2), F(3), (int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], FM(4), p[5], st);
    case PDAE_OP_LINEAR_BWD_GROUP:
      return pdae_linear_bwd_group((const pdae_linear_bwd_item*)p[0], (const int32_t*)p[1], (int)i[0], (int)i[1], (int)i[2], (int)i[3], st);
    case PDAE_OP_LINEAR_GROUP: return pdae_linear_group((const pdae_linear_item*)p[0], (const int32_t*)p[1], (int)i[0], (int)i[1], (int)i[2], (int)i[3], st);
    case PDAE_OP_ADAM_EMA:
      return pdae_adam_ema(FM(0), F(1), FM(2), FM(3), FM(4), (size_t)i[0], (float)f[0], (float)f[1], (float)f[2], (float)f[3], (float)f[4], (int)i[1],
                           (float)f[5], (float)f[6], (float)f[7], (float)f[8], (unsigned int*)p[5], (int)i[2], st);
    case PDAE_OP_SOFTMAX: return pdae_softmax(FM(0), i[0], (int)i[1], st);
    case PDAE_OP_SOFTMAX_BWD: return pdae_softmax_bwd(F(0), FM(1), i[0], (int)i[1], st);
    case PDAE_OP_COLSUM: return pdae_colsum(F(0), i[0], (int)i[1], FM(1), (int)i[2], p[2], st);
    case PDAE_OP_CONV_FWD_GN:
      desc_from(i, d);
      if (p[19]) conv3x3p_arm_stats((float*)p[19]);      // p[19] of the forward-convolution records: pdae_conv_stats_arm
      return pdae_conv2d_fwd_gn(&d, F(0), F(1), F(2), (int)i[15], p[3], F(4), F(5), (int)i[14], FM(6), st);
    case PDAE_OP_CONV_FWD_SKIP: {
      desc_from(i, d);
      pdae_conv_desc ds = d;
      ds.Hi = d.Ho; ds.Wi = d.Wo; ds.C0 = (int)i[15]; ds.C1 = (int)i[16]; ds.KH = ds.KW = 1; ds.pad = 0; ds.up = 0; ds.stride = 1;
      if (p[19]) conv3x3p_arm_stats((float*)p[19]);
      return pdae_conv2d_fwd_skip(&d, F(0), F(1), F(2), (int)i[14], p[3], F(4), &ds, F(5), F(6), p[7], F(8), FM(9), st);
    }
    case PDAE_OP_SUBSAMPLE2: return pdae_subsample2(F(0), (int)i[0], (int)i[1], (int)i[2], (int)i[3], FM(1), st);
    case PDAE_OP_GN_STATS_QUADS: return pdae_gn_stats_quads(F(0), (int)i[0], (int)i[1], (int)i[2], (int)i[3], FM(1), st);
    case PDAE_OP_ZERO_INSERT2: return pdae_zero_insert2(F(0), (int)i[0], (int)i[1], (int)i[2], (int)i[3], FM(1), st);
    case PDAE_OP_CONV_WPREP_GROUP: return pdae_conv_wprep_group((const pdae_wprep_job*)p[0], (const int32_t*)p[1], (int)i[0], (int)i[1], st);
    case PDAE_OP_CONV_SKIP_WPREP: {
      desc_from(i, d);
      pdae_conv_desc ds = d;
      ds.Hi = d.Ho; ds.Wi = d.Wo; ds.C0 = (int)i[14]; ds.C1 = (int)i[15]; ds.KH = ds.KW = 1; ds.pad = 0; ds.up = 0; ds.stride = 1;
      return pdae_conv_skip_wprep(&d, &ds, F(0), p[1], st);
    }
    case PDAE_OP_CONV_WPREP: desc_from(i, d); return pdae_conv_wprep(&d, F(0), (int)i[14], p[1], st);
    case PDAE_OP_MEMSET: {
      hipError_t e = hipMemsetAsync(p[0], 0, (size_t)i[0], S(st));
      if (e != hipSuccess) { pdae_set_error("memset: %s", hipGetErrorString(e)); return (int)e; }
      return PDAE_OK;
    }
    case PDAE_OP_COPY: {
      hipError_t e = hipMemcpyAsync(p[1], p[0], (size_t)i[0], hipMemcpyDeviceToDevice, S(st));
      if (e != hipSuccess) { pdae_set_error("copy: %s", hipGetErrorString(e)); return (int)e; }
      return PDAE_OK;
    }
    default: pdae_set_error("run_ops: unknown op kind %d", o.kind); return PDAE_EINVAL;
  }
#undef F
#undef FM
}

// ---- second stream of the executor.  Ops flagged PDAE_OPF_SIDE (pdae_op.flags) are launched on a low-priority stream owned by the library:
// each of them first waits for the caller's stream at its position in the op array (its inputs are complete), PDAE_OP_JOIN -- and the end of
// every pdae_run_ops call -- makes the caller's stream wait for everything launched on the side stream so far.  What the plan builder has to
// guarantee between a side op and the next join: nothing on the caller's stream overwrites its inputs or reads its outputs (engine.py defers
// the recycling of such buffers to the join).  Used for the weight gradients: they feed nothing in the backward chain, and the chain's
// HBM-bound passes (GroupNorm backward, 1x1 data gradients, reductions) leave the matrix pipes idle.
// Round 6: one side stream PER (device, calling stream), created under a mutex -- two host threads that issue plans on
// their own streams (the training thread and an evaluation / autograd thread: SURVEY section 8(b) "re-entrant") share neither the fork / join
// events nor the pending flag; a partially created entry is torn down and the ops run in order on the caller's stream instead.
struct SideStream { int dev = -1; hipStream_t caller = nullptr; hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool dirty = false; };
#define PDAE_MAX_SIDE 64
static SideStream g_side[PDAE_MAX_SIDE];
static int g_side_n = 0;
static std::mutex g_side_mu;
static SideStream* side_of(pdae_stream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_side_mu);
  for (int k = 0; k < g_side_n; ++k)
    if (g_side[k].dev == dev && g_side[k].caller == S(stream)) return &g_side[k];
  if (g_side_n >= PDAE_MAX_SIDE) return nullptr;                  // (more caller streams than slots: their plans run in order)
  SideStream q;
  int lo = 0, hi = 0;
  hipDeviceGetStreamPriorityRange(&lo, &hi);                      // lo = least priority: the backward chain on the caller's stream goes first
  const int mode = pdae_knob(KNOB_SIDE_STREAM);                   // 1: least priority, 2: the default priority, 3: highest
  if (hipStreamCreateWithPriority(&q.s, hipStreamNonBlocking, mode == 2 ? 0 : (mode == 3 ? hi : lo)) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&q.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&q.join, hipEventDisableTiming) != hipSuccess) {
    if (q.fork) hipEventDestroy(q.fork);
    hipStreamDestroy(q.s);
    return nullptr;
  }
  q.dev = dev; q.caller = S(stream);
  g_side[g_side_n] = q;
  return &g_side[g_side_n++];
}
static int side_join(SideStream* q, pdae_stream_t stream) {
  if (!q || !q->dirty) return PDAE_OK;
  hipError_t e = hipEventRecord(q->join, q->s);
  if (e == hipSuccess) e = hipStreamWaitEvent(S(stream), q->join, 0);
  q->dirty = false;
  if (e != hipSuccess) { pdae_set_error("side stream join: %s", hipGetErrorString(e)); return (int)e; }
  return PDAE_OK;
}

extern "C" int pdae_run_ops(const pdae_op* ops, int n, pdae_stream_t stream) {
  SideStream* side = nullptr;
  const bool use_side = pdae_knob(KNOB_SIDE_STREAM) != 0;
  for (int k = 0; k < n; ++k) {
    int e;
    if (ops[k].kind == PDAE_OP_JOIN) e = side_join(side, stream);
    else if ((ops[k].flags & PDAE_OPF_SIDE) && use_side && (side || (side = side_of(stream)))) {
      hipError_t he = hipEventRecord(side->fork, S(stream));
      if (he == hipSuccess) he = hipStreamWaitEvent(side->s, side->fork, 0);
      if (he != hipSuccess) { pdae_set_error("side stream fork: %s", hipGetErrorString(he)); e = (int)he; }
      else { side->dirty = true; e = run_one(ops[k], (pdae_stream_t)side->s); }
    } else e = run_one(ops[k], stream);
    if (e != PDAE_OK) {
      conv3x3p_take_stats();                // a statistics request armed for an op that failed before its entry point must not reach a later launch
      { int a_; take_wgn(&a_); PatchGnb g_; take_gnb(&g_); g_gnparts = nullptr; }
      char msg[400];
      snprintf(msg, sizeof(msg), "%s", g_err);
      pdae_set_error("op %d (kind %d): %s", k, ops[k].kind, msg);
      side_join(side, stream);
      return e;
    }
  }
  return side_join(side, stream);
}
