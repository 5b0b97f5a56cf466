// Shared declarations for libpdae_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PDAE_OK 0
#define PDAE_EINVAL (-1)

// sets the thread-local error string returned by pdae_last_error()
void pdae_set_error(const char* fmt, ...);

#define PDAE_CHECK_ARG(cond, ...)          \
  do {                                     \
    if (!(cond)) {                         \
      pdae_set_error(__VA_ARGS__);         \
      return PDAE_EINVAL;                  \
    }                                      \
  } while (0)

// launch-error check without synchronising the stream
static inline int pdae_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pdae_set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return PDAE_OK;
}

// Timing-probe macros compile pieces of a kernel out (WRONG RESULTS by design).  They are only legal in the side builds of tools/probe_build.py,
// which defines PDAE_PROBE_BUILD: a product build that picks one up by accident does not compile.
#if (defined(PDAE_C1_PROBE_NOB) || defined(PDAE_C1_PROBE_NOMMA) || defined(PDAE_C1_PROBE_NOCONV) || defined(PDAE_C1_PROBE_NOSTORE) || defined(PDAE_AT_PROBE_NNNOLOAD) || defined(PDAE_AT_PROBE_NNNOMMA) || defined(PDAE_AT_PROBE_NONN) || defined(PDAE_AT_PROBE_NONT) || defined(PDAE_AT_PROBE_NOSCHED) || defined(PDAE_PROBE_NOA) || defined(PDAE_PROBE_NOB) || defined(PDAE_PROBE_NOSTAGE) || defined(PDAE_R_PROBE_24U) || defined(PDAE_R_PROBE_NOA) || defined(PDAE_R_PROBE_NOB) || defined(PDAE_R_PROBE_NOCONV) || defined(PDAE_R_PROBE_NODRAIN) || defined(PDAE_R_PROBE_NOGLOAD) || defined(PDAE_W3_PROBE_6TAPS) || defined(PDAE_W3_PROBE_NOLOAD) || defined(PDAE_W3_PROBE_NOMMA) || defined(PDAE_W3_PROBE_NOSTAGE) || defined(PDAE_V_PROBE_NOSTAGE) || defined(PDAE_V_PROBE_NOLOAD) || defined(PDAE_V_PROBE_NOMMA) || defined(PDAE_Y_PROBE_NOA) || defined(PDAE_Y_PROBE_NOB) || defined(PDAE_Y_PROBE_NOCONV) || defined(PDAE_Y_PROBE_NOEPI) || defined(PDAE_Y_PROBE_NOGLOAD)) && !defined(PDAE_PROBE_BUILD)
#error "PDAE_*_PROBE_* macros give wrong results by design: build probes with tools/probe_build.py (-DPDAE_PROBE_BUILD), never the product library"
#endif

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- tuning / A-B switches of the library (api.hip).  ONE registry: a knob's value is pdae_set_knob() > its environment variable, read ONCE at the
// knob's first use > its default.  Nothing re-reads the environment per call: a process cannot change the routing between preparing a
// convolution's weights and launching it by editing its environment (tests switch through pdae_set_knob; prepared buffers are additionally
// tagged with their form, conv3x3p.hip).  The names are the environment variables of DESIGN.md section 10.
enum PdaeKnob {
  KNOB_W1 = 0,        // PDAE_W1: Winograd F(2,3)-along-x form of the 3x3 convolutions: 0 off, 1 by fill rules (default), 2 / 3 every eligible shape in 16- / 8-row tiles
  KNOB_W1_EFF,        // PDAE_W1_EFF: minimum % of the CUs busy in the last round of tiles for that form (85)
  KNOB_P3R,           // PDAE_P3R: conv3x3r (direct persistent form): 0 never, 1 by fill heuristic (default), 2 every eligible shape
  KNOB_P3R_MIN,       // PDAE_P3R_MIN (512)
  KNOB_P3R_EFF,       // PDAE_P3R_EFF (85)
  KNOB_EDGE,          // PDAE_EDGE: 3-channel edge layers on the MFMA edge kernels (1)
  KNOB_P3_TH,         // PDAE_P3_TH: tile height of conv3x3p (0 = default 8)
  KNOB_SPLIT_STATS,   // PDAE_SPLIT_STATS: split-K 3x3 launches leave GroupNorm partial statistics (1)
  KNOB_W3_STAGGER,    // PDAE_W3_STAGGER (0)
  KNOB_Y_STAGGER,     // PDAE_Y_STAGGER (0)
  KNOB_C1_SLAB,       // PDAE_C1_SLAB: slab-traffic term of the conv1x1 split-K plan (1)
  KNOB_C1_BF16,       // PDAE_C1_BF16: three-plane bf16 format in the 1x1 kernels also in mode 4 (0)
  KNOB_NO_SKINNY,     // PDAE_NO_SKINNY: M <= 32 linears on the generic GEMM (0)
  KNOB_C1_ROT,        // PDAE_C1_ROT: conv1x1 workgroups start their channel stages at different offsets (1)
  KNOB_W1_ROWS8,      // PDAE_W1_ROWS8: 8-row tiles of the Winograd form for layers too small for 16-row tiles (1)
  KNOB_W1_EFF8,       // PDAE_W1_EFF8: minimum % of the CUs busy in the last round of 8-row tiles when there is more than one round (70)
  KNOB_W1_MIN8,       // PDAE_W1_MIN8: minimum number of 8-row tiles of a launch for that form (160)
  KNOB_SIDE_STREAM,   // PDAE_SIDE_STREAM: ops flagged PDAE_OPF_SIDE run on the library's second stream (1); 0: on the caller's stream, in order
  KNOB_W3V,           // PDAE_W3V: 3x3 weight gradients in the producer / consumer form conv3x3v.hip where it applies (1); 0: conv3x3w.hip everywhere
  KNOB_Y_XCD,         // PDAE_Y_XCD: conv3x3y gives the workgroups of an XCD consecutive tile indices: 1 launches with >= 2 output-channel tiles, 2 all, 0 none (default: measured slower)
  KNOB_Y_GRID_TRIM,   // PDAE_Y_GRID_TRIM: conv3x3y launches 256 - k workgroups (0): k CUs stay free for a collective's channels at world size > 1
  KNOB_COUNT
};
int pdae_knob(int id);

// fp16 operand split of (e0 * sc, e1 * sc), sc a power of two: packed head plane (round-to-nearest fp16) and packed residual plane (fp16 of
// e * sc - head, which is exact in fp32).  v_fma_mixlo/mixhi_f16 fuse scale, conversion and packing, and take the fp16 head straight back
// as the addend of the residual: 4 VALU instructions per pair where mul + cvt + cvt back + sub + cvt + shift/or packing took 13.  It matters:
// on a SIMD with two resident waves every VALU instruction costs ~5 issue cycles and the matrix pipe idles meanwhile (tools/probes/valu_rate.hip;
// the weight-gradient kernel spent 6.6 VALU instructions per MFMA on this).  Bit-identical to the scalar conversions (tools/probes/mix_probe.hip).
__device__ __forceinline__ void pdae_f16_split2s(float e0, float e1, float sc, unsigned& head, unsigned& resid) {
  unsigned h, l;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(e0), "v"(sc));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(e1), "v"(sc));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(e0), "v"(sc), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(e1), "v"(sc), "v"(h));
  head = h; resid = l;
}

// ---- fp16-window guard of the two-fp16-plane format (math 4).  Scaled operands must stay inside the fp16 range (|x| <= 65504) for both planes to
// be finite.  Nothing is clamped: an operand beyond the window becomes Inf in the high plane and turns its output rows into NaN exactly like
// an fp16 autocast overflow would, NaN / Inf inputs propagate as they do in fp32 arithmetic -- and the launch COUNTS the event in the device
// counter registered with pdae_set_saturation_counter.  The optimizer kernel refuses to apply a step while that counter is non-zero and the
// host falls back to the range-free bf16x6 split (pdae_amd/hip.py: SaturationGuard).  nullptr = guard not armed.
unsigned int* pdae_sat_counter();
#ifdef __HIPCC__
#define PDAE_F16_LIMIT 60000.f        // margin below 65504 for the round-to-nearest of the high plane
// amax = running per-lane max |v * sc| of the operands this lane has split (one VGPR for the whole kernel; NaN does not raise it, Inf does)
__device__ __forceinline__ void pdae_f16_amax4(const float4& v, float sc, float& amax) {
  amax = fmaxf(amax, sc * fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
}
__device__ __forceinline__ void pdae_sat_report(unsigned int* counter, float amax) {
  const unsigned long long over = __builtin_amdgcn_ballot_w64(!(amax <= PDAE_F16_LIMIT));
  if (counter && over && (threadIdx.x & 63) == 0) atomicAdd(counter, 1u);
}
#endif
