/* libpdae_hip.so -- C ABI of the MI355X-native (gfx950) PDAE hot path.
 *
 * The reference (ckczzj/PDAE) has no FFI / operator plug-in interface: its hot path is the set of
 * torch.nn / ATen calls issued by model/module.py, model/unet.py, model/shift_unet.py,
 * diffusion/gaussian_diffusion.py and diffusion/ddim.py.  Each entry point below
 * replaces one fused group of those calls and cites them (paths relative to the reference repo).
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = invalid argument, >0 = hipError_t; the message of the
 *     last failure on the calling thread is returned by pdae_last_error();
 *   - all buffers are caller-owned DEVICE pointers (fp32 unless noted, int64 for timesteps/labels);
 *     no allocation, no host synchronisation inside; workspaces are sized by the *_workspace_bytes queries;
 *   - the HIP stream is always explicit; functions are re-entrant (usable from the autograd thread);
 *   - activations are NHWC ("channels last"), contiguous: [N][H][W][C];
 *   - 2-D conv weights are [Cout][KH][KW][Cin] == the memory of a torch (Cout,Cin,KH,KW) tensor in
 *     channels_last format, so a reference state-dict loads without repacking;
 *   - "x0/C0, x1/C1" is a virtual channel concat [x0 | x1] (torch.cat(...,dim=1) in unet.py:200,
 *     shift_unet.py:278,281) that is never materialised; pass x1 = NULL, C1 = 0 for a single tensor.
 */
#ifndef PDAE_HIP_H
#define PDAE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pdae_stream_t; /* hipStream_t */

const char* pdae_last_error(void);
/* ABI history.  11 (round 6): + pdae_conv2d_wgrad_form (which kernel a weight gradient runs on); knobs PDAE_W3V, PDAE_Y_XCD, PDAE_Y_GRID_TRIM.
 * 10 (round 5): pdae_op.reserved became pdae_op.flags (PDAE_OPF_SIDE: the executor's second stream) + PDAE_OP_JOIN + knob
 *   PDAE_SIDE_STREAM; arrays written for ABI 9 (flags = 0) run unchanged.
 * 9 (round 5): - pdae_wino_* (the gated 2-D Winograd probe left the library: tools/probes/r04_winograd/); + pdae_set_knob / pdae_get_knob
 *   (the library no longer reads its environment per call); + pdae_conv_gn_input_arm / pdae_conv2d_wgrad_gn_ok (weight gradient that recomputes a fused
 *   GroupNorm input); + pdae_conv_gnbwd_bytes / pdae_conv_gnbwd_arm / pdae_gn_bwd_parts_arm (GroupNorm-backward sums from the data
 *   gradient's epilogue); prepared 3x3 weights are TAGGED with their form and a launch that expects the other form fails with PDAE_EINVAL; the
 *   prepared copy of a direct-form 3x3 convolution shrinks back to 18 k-halves per 32 channels (24 only in the Winograd form) and a fused-skip
 *   copy to 2 -- cached copies written by ABI 7 / 8 must be re-prepared.
 * 8 (round 4): + pdae_conv3x3_form, PDAE_MATH_DIRECT; the Winograd F(2,3)-along-x layout of prepared 3x3 weights (12 transform taps = 24 k-halves per 32
 *   channels; ABI 7 sized EVERY 3x3 copy for it: +33 % over ABI 6) -- the internal job field `transposed` of pdae_wprep_job carries bit 8
 *   (PDAE_WPREP_FORM_X) for that layout.  7: the same layout behind PDAE_W1 without the query.
 * 6: + pdae_wino_* (removed in 9); 5: + pdae_subsample2 / pdae_zero_insert2; 4: + pdae_conv_wprep_job / _group; 3: + pdae_conv_stats_* /
 *   pdae_gn_coef_from_conv_stats; 2: saturation counter, row-coefficient samplers, fused attention, RCCL. */
int pdae_abi_version(void);

/* Tuning / A-B switches (PDAE_W1, PDAE_W1_EFF, PDAE_P3R, PDAE_P3R_MIN, PDAE_P3R_EFF, PDAE_EDGE, PDAE_P3_TH, PDAE_SPLIT_STATS,
 * PDAE_W3_STAGGER, PDAE_Y_STAGGER, PDAE_C1_SLAB, PDAE_C1_BF16, PDAE_NO_SKINNY, PDAE_C1_ROT, PDAE_W1_ROWS8, PDAE_W1_EFF8, PDAE_W1_MIN8).  A knob's value is pdae_set_knob() > the environment variable of
 * the same name, read ONCE at the knob's first use > the default; the library never re-reads its environment.  Unknown name: PDAE_EINVAL.
 * Changing PDAE_W1 between pdae_conv_wprep and the launch that takes the copy makes that launch fail (form tag), it does not corrupt results. */
int pdae_set_knob(const char* name, int value);
int pdae_get_knob(const char* name, int* value);

/* ---- convolution (F.conv2d / conv1d k=1: module.py:242,265,276,412,420; unet.py:62,174; encoder/ffhq.py:12-30) */
typedef struct pdae_conv_desc {
  int32_t N, Hi, Wi;      /* stored input spatial size */
  int32_t C0, C1;         /* input channels: virtual concat of two tensors */
  int32_t Ho, Wo, Cout;   /* output size */
  int32_t KH, KW, stride, pad;
  int32_t up;             /* 1: the conv reads the nearest-x2 upsample of the stored input (F.interpolate, module.py:169) */
  int32_t math;           /* MFMA arithmetic for fp32 tensors: 0 = f32 MFMA (exact fp32 fmaf chain);
                           * 1 = bf16 operands (rn); 2 = 2 bf16 planes, 3 products (~2^-17 per product);
                           * 3 = 3 exact bf16 planes, 6 products (~2^-23 per product, fp32 grade);
                           * 4 = 2 fp16 planes (11+11 mantissa bits), 3 products (~2^-21 per product) in the FORWARD 3x3 patch kernel -- operands must
                           *     lie inside the fp16 range (post-GroupNorm activations, weights): the 3x3 patch kernel (forward and, given dy_amax,
                           *     data gradient), the 3x3 weight-gradient kernel (given dy_amax) and the 1x1 kernel; scaled operands outside
                           *     +-60000 are clamped and COUNTED (pdae_set_saturation_counter).  The generic implicit-GEMM kernels and gradient
                           *     launches without dy_amax run mode 3.  fp32 accumulate in all modes. */
} pdae_conv_desc;
/* OR-ed into pdae_conv_desc.math by the caller: the FORWARD form of this 3x3 convolution (prepared weights and launch) stays on the direct kernels
 * whatever the shape heuristics say (pdae_conv3x3_form -> 0), which keeps it eligible for fused 1x1 skip chunks (pdae_conv2d_fwd_skip).  The data- and
 * weight-gradient entry points ignore the bit.  Use the SAME descriptor for pdae_conv_wprep and the launch. */
#define PDAE_MATH_DIRECT 0x100

/* fp16-window guard of math 4: counter = device word (zero it yourself) that every math-4 convolution launch increments when it had to clamp
 * a scaled operand (|x| > 60000, NaN, Inf) into the fp16 window; NULL disarms the guard.  A non-zero counter means "results of this pass are
 * not fp32-grade: discard and re-run with math 3" -- pdae_adam_ema refuses the update, the Python host (pdae_amd/hip.py SaturationGuard)
 * rebuilds its plans in bf16x6.  One counter per process. */
int pdae_set_saturation_counter(unsigned int* counter);

/* Fast paths in the bf16 modes (math >= 1): 3x3 / stride-1 / pad-1 convolutions run on the LDS-patch kernel (conv3x3p.hip) and 1x1
 * convolutions on the 1x1 kernel (conv1x1.hip).  Both read their weights pre-split into bf16 planes in MFMA-fragment
 * order.  pdae_conv_wprep_bytes returns the size of that copy (+ split-K scratch) for the forward (flags = 0) or data-gradient
 * (PDAE_WPREP_TRANSPOSED) convolution of d, or 0 when the convolution is not eligible (then pass wp = NULL and the generic
 * implicit-GEMM kernel runs).  pdae_conv_wprep writes it from the fp32 weights w [Cout][KH][KW][Cin]; it must be re-run whenever w
 * changes (it is ~1% of the convolution's time). */
#define PDAE_WPREP_TRANSPOSED 1   /* weights for the data gradient */
#define PDAE_WPREP_GN 4           /* forward conv with fused GroupNorm input (pdae_conv2d_fwd_gn): two sources allowed, W % 16 == 0 */
#define PDAE_WPREP_F16_GRAD 16     /* with TRANSPOSED and math 4: data-gradient weights in the fp16 format (the launch then needs dy_amax) */
#define PDAE_WPREP_FORCE 2        /* _bytes: shape eligibility only, ignore the "enough tiles to fill 256 CUs" heuristic */
size_t pdae_conv_wprep_bytes(const pdae_conv_desc* d, int flags);
/* 1 when the prepared copy pdae_conv_wprep(d, w, flags) and the launch that takes it are in the Winograd F(2, 3)-along-x form (conv3x3y.hip: two
 * thirds of the matrix instructions of the direct form, chip-filling layers only); 0 for the direct kernels or when there is no prepared copy.
 * Informational (bench.py prices a launch's roofline by the instructions it issues); flags as for pdae_conv_wprep_bytes. */
int pdae_conv3x3_form(const pdae_conv_desc* d, int flags);
int pdae_conv_wprep(const pdae_conv_desc* d, const float* w, int flags, void* wp, pdae_stream_t stream);
/* y[N,Ho,Wo,Cout] = conv(x) + bias (+ res).  res_mode: 0 none, 1 res[N,Ho,Wo,Cout], 2 res stored at half resolution
 * (the x_upd(x) skip of an up-ResBlock, module.py:279-284,297).  tile: 0 = auto, 64 or 128.  wp: NULL or pdae_conv_wprep(d, w, 0). */
int pdae_conv2d_fwd(const pdae_conv_desc* d, const float* x0, const float* x1, const float* w, const void* wp, const float* bias,
                    const float* res, int res_mode, float* y, int tile, pdae_stream_t stream);
/* Forward 3x3 convolution of act(a[n,c] * (x - mu[n,c]) + b[n,c]) -- GroupNorm / AdaGN (+ SiLU, act = 1) of module.py:241,257-263,
 * 293-294,379-381 applied while the LDS patch is staged, so the normalised activation never exists in HBM (used where nothing is kept
 * for a backward pass: the frozen trunk, sampling).  coef = [mu | a | b], each [N][C0+C1], from pdae_gn_coef; zero padding applies to the
 * activated tensor; wp = pdae_conv_wprep(d, w, PDAE_WPREP_GN).  act = 0 (the affine map alone) exists in the DIRECT form only: where
 * pdae_conv3x3_form(d, PDAE_WPREP_GN) == 1 the call fails with PDAE_EINVAL -- OR PDAE_MATH_DIRECT into d->math (before the weights are prepared) to
 * pin the direct form for such a stage.  Since ABI 9 the same launch also serves TRAINED stages: pair it with pdae_conv_gn_input_arm on the weight
 * gradient, which recomputes the activation instead of reading a saved copy. */
int pdae_conv2d_fwd_gn(const pdae_conv_desc* d, const float* x0, const float* x1, const float* coef, int act, const void* wp, const float* bias,
                       const float* res, int res_mode, float* y, pdae_stream_t stream);
/* ResBlock tail in one launch (module.py:265,276,297):  y = conv3x3_d(in) + bias + conv1x1_ds([s0 | s1]) + bias_s,  in = coef ?
 * act(GN-affine([x0 | x1])) : x0.  The 1x1 skip_connection enters the 3x3 kernel's K loop as extra centre-tap chunks of the raw block
 * input, so its output never exists in HBM.  ds: 1x1 descriptor on d's output grid (C0/C1 = channels of s0/s1);
 * wps = pdae_conv_skip_wprep(d, ds, w_skip) (the skip weights in the plane format / scale of d's main loop); wp as for pdae_conv2d_fwd
 * (coef == NULL) or pdae_conv2d_fwd_gn.  pdae_conv2d_fwd_skip_ok tells whether the pair is eligible (otherwise run the two convolutions
 * separately, the second with res_mode 1). */
int pdae_conv2d_fwd_skip_ok(const pdae_conv_desc* d, const pdae_conv_desc* ds);
size_t pdae_conv_skip_wprep_bytes(const pdae_conv_desc* d, const pdae_conv_desc* ds);
int pdae_conv_skip_wprep(const pdae_conv_desc* d, const pdae_conv_desc* ds, const float* w_skip, void* wps, pdae_stream_t stream);
/* Grouped form of pdae_conv_wprep / pdae_conv_skip_wprep for hosts that run the same set of convolutions every step (a training or
 * sampling plan): describe every prepared copy ONCE with pdae_conv_wprep_job / pdae_conv_skip_wprep_job (same arguments, same decisions;
 * they only fill `job`, nothing is launched), upload the job table and the prefix table first_block[j] = sum of jobs[0..j).nblocks to the
 * device, and call pdae_conv_wprep_group(jobs, first_block, njobs, total_blocks) whenever the weights have changed: one launch instead of
 * one per convolution (134 -> 1 per FFHQ-128 training step).  The pointers inside a job must stay valid; the layout of pdae_wprep_job is
 * part of the ABI (48 bytes). */
typedef struct pdae_wprep_job {
  const float* w; void* wp;          /* fp32 weights, prepared copy */
  int32_t Nout, C, NT, transposed;   /* GEMM N, GEMM K channels, 32-channel tiles, bit 0: data-gradient form, bit 3 (8): Winograd-along-x layout (T = 12 / 2) */
  int32_t T, ns;                     /* taps of the 3x3 layout (9, 1 for fused skip chunks; 0 = the 1x1 layout), operand format (1..4) */
  float wscale; int32_t nblocks;     /* power-of-two weight scale (format 4), 256-thread blocks this job needs */
} pdae_wprep_job;
int pdae_conv_wprep_job(const pdae_conv_desc* d, const float* w, int flags, void* wp, pdae_wprep_job* job);
int pdae_conv_skip_wprep_job(const pdae_conv_desc* d, const pdae_conv_desc* ds, const float* w_skip, void* wps, pdae_wprep_job* job);
int pdae_conv_wprep_group(const pdae_wprep_job* jobs_dev, const int32_t* first_block_dev, int njobs, int total_blocks, pdae_stream_t stream);
int pdae_conv2d_fwd_skip(const pdae_conv_desc* d, const float* x0, const float* x1, const float* coef, int act, const void* wp, const float* bias,
                         const pdae_conv_desc* ds, const float* s0, const float* s1, const void* wps, const float* bias_s, float* y,
                         pdae_stream_t stream);
/* GroupNorm statistics of a convolution's OUTPUT, produced by the convolution while it stores the tensor -- the GroupNorm that follows
 * (module.py:241,257: in_layers / out_layers norm of the next stage) then needs no pass over the tensor at all:
 *   bytes = pdae_conv_stats_bytes(d, ds, &tpi)   size of the partial-sum buffer, 0 when the forward convolution of d (with the fused skip
 *                                                convolution ds, or NULL) does not run on the 3x3 patch kernels (split-K launches leave the
 *                                                sums from their slab reduction: tpi is then pixels / 8 or / 16 per image)
 *   pdae_conv_stats_arm(part)                    one-shot: the NEXT pdae_conv2d_fwd / _fwd_gn / _fwd_skip on this host thread also writes
 *                                                part[N][tpi][Cout/4] x (sum, sum of squares) of its output (in a pdae_op record: p[19])
 *   pdae_gn_coef_from_conv_stats(...)            mean / rstd / coef ([mu | a | b], as pdae_gn_stats_coef) of the virtual concat of one or two such
 *                                                tensors from their partial sums (fp64 combine); C / G and C0 must be multiples of 4. */
size_t pdae_conv_stats_bytes(const pdae_conv_desc* d, const pdae_conv_desc* ds, int32_t* tiles_per_image);
/* The same partial sums for a tensor x[N][HW][C] whose producer cannot leave them (ABI 9; the 3 -> 128 stem runs on the edge kernel): one pass,
 * part[N][tiles_per_image][C / 4] x (sum, sum of squares) over HW / tiles_per_image pixels each; every GroupNorm that reads x (alone or as one source of a
 * concat) then takes pdae_gn_coef_from_conv_stats instead of its own statistics pass.  C % 4 == 0.  pdae_op: p[0] = x, p[1] = part, i = N, HW, C, tiles. */
int pdae_gn_stats_quads(const float* x, int N, int HW, int C, int tiles_per_image, float* part, pdae_stream_t stream);
int pdae_conv_stats_arm(float* part);
int pdae_gn_coef_from_conv_stats(int N, int HW, int C0, int C1, int G, float eps, const float* part0, int tpi0, const float* part1, int tpi1,
                                 const float* gamma, const float* beta, const float* ss, const float* zss, float* mean, float* rstd, float* coef,
                                 pdae_stream_t stream);
/* dx[N,Hl,Wl,ci_cnt] (+)= dL/d(conv input channels ci_off..ci_off+ci_cnt) on the LOGICAL input grid (Hl = 2*Hi when up).
 * wp_t: NULL or pdae_conv_wprep(d, w, PDAE_WPREP_TRANSPOSED): the data gradient then runs as a forward convolution of dy (3x3: the
 * whole channel range only; 1x1: any 32-aligned ci_off). */
int pdae_conv2d_dgrad(const pdae_conv_desc* d, const float* dy, const float* w, const void* wp_t, float* dx, int ci_off, int ci_cnt,
                      int accumulate, int tile, const float* dy_amax, pdae_stream_t stream);
/* dy_amax (dgrad / wgrad, optional, math 4 only): device scalar max|dy| from pdae_amax.  When given, the 3x3 gradient kernels run the
 * two-fp16-plane format too, with dy scaled by the power of two that puts its abs-max into [1024, 2048) (undone exactly in the epilogue);
 * NULL: the exact three-plane bf16 split (range-free). */
int pdae_amax(const float* x, size_t n, float* out, pdae_stream_t stream);
/* GroupNorm-backward sums from the data gradient that PRODUCES dA (ABI 9; conv3x3y.hip, GB instantiation).  The backward of
 * y = silu(a[n,c] (x - mu[n,c]) + b[n,c]) needs two sums per (n, c) over the pixels: S0 = sum dv, S1 = sum dv (x - mu), dv = dA silu'(.) --
 * pdae_gn_bwd takes them with a reduction pass over (x, dA), 2 of its 5 tensor passes.  When dA is the output of a 3x3 data gradient in the
 * Winograd-along-x form, that launch can leave them from its epilogue instead (x enters as its epilogue operand):
 *   bytes = pdae_conv_gnbwd_bytes(d, flags, &tiles)   size of part[N][tiles][C0 + C1][2] for the data gradient of d (d->C0 / C1 = the channel split of
 *                                                     the GroupNorm's two-source input; flags: PDAE_WPREP_F16_GRAD when the launch brings dy_amax);
 *                                                     0 = not available (not the Winograd form, d->up, sources not whole 32-channel runs, > 64 tiles
 *                                                     of 16 x 16 pixels per image)
 *   pdae_conv_gnbwd_arm(x0, C0, x1, C1, coef, 1, part) one-shot: the NEXT pdae_conv2d_dgrad on this host thread (whole input, no accumulate) also
 *                                                     writes part; coef = [mu | a | b] of the GroupNorm's forward; SiLU, no dropout (act must be 1)
 *   pdae_gn_bwd_parts_arm(part, tiles)                one-shot: the NEXT pdae_gn_bwd (mode 0, act 1, drop_p 0) skips its reduction and finalizes from part
 * In a pdae_op record: CONV_DGRAD p[5..8] = x0, x1, coef, part, i[18..20] = C0, C1, act;  GN_BWD p[19] = part, i[13] = tiles. */
size_t pdae_conv_gnbwd_bytes(const pdae_conv_desc* d, int flags, int32_t* tiles_per_image);
int pdae_conv_gnbwd_arm(const float* x0, int C0, const float* x1, int C1, const float* coef, int act, float* part);
int pdae_gn_bwd_parts_arm(const float* part, int tiles_per_image);
/* dw[Cout][KH][KW][Cin] (+)= sum over pixels; split-K over pixels through the workspace, reduced in fixed order.
 * db (optional) [Cout] (+)= column sums of dy = the bias gradient; the 3x3 kernel takes them from its own dY staging (no second read of dy). */
size_t pdae_conv2d_wgrad_workspace_bytes(const pdae_conv_desc* d);
/* (ABI 11, informational) the kernel pdae_conv2d_wgrad(d, ...) runs on: 3 = conv3x3v (3x3, producer / consumer waves: whole 64-channel blocks on both
 * sides, 16-pixel-wide tiles, >= 64 pixel tiles, a two-plane format), 2 = conv3x3w (the other 3x3 / stride-1 / pad-1 layers), 1 = another dedicated
 * kernel (1x1 weight gradient, 3-channel edge / head layers), 0 = the generic implicit GEMM.  with_dy_amax / with_gn_input describe the launch
 * (the fp16 format needs the dY scale; pdae_conv_gn_input_arm joins the two sources).  Follows the knob PDAE_W3V as the launch does. */
int pdae_conv2d_wgrad_form(const pdae_conv_desc* d, int with_dy_amax, int with_gn_input);
int pdae_conv2d_wgrad(const pdae_conv_desc* d, const float* x0, const float* x1, const float* dy, float* dw, float* db, int accumulate, void* ws,
                      size_t ws_bytes, const float* dy_amax, pdae_stream_t stream);
/* Weight gradient of a convolution whose FORWARD applied GroupNorm / AdaGN (+ SiLU) to its raw two-source input inside the staging
 * (pdae_conv2d_fwd_gn; module.py:241-242: in_layers GroupNorm -> SiLU -> conv): that forward never wrote the activated tensor, so the weight
 * gradient recomputes act(a[n,c] * (x - mu[n,c]) + b[n,c]) the same way while it stages X (ABI 9).  pdae_conv_gn_input_arm(coef, act) is a one-shot
 * request like pdae_conv_stats_arm: the NEXT pdae_conv2d_wgrad on this host thread takes it on entry and then reads (x0, x1) as the RAW sources;
 * coef = [mu | a | b], each [N][C0 + C1], as pdae_gn_coef / pdae_gn_stats_coef / pdae_gn_coef_from_conv_stats leave them; act 1 = SiLU, 0 = none.
 * Eligible (pdae_conv2d_wgrad_gn_ok == 1): 3x3 / stride 1 / pad 1, C0 and C1 multiples of 32, Wo % 16 == 0, Ho % 8 == 0, Cout % 4 == 0, math >= 1;
 * anything else fails with PDAE_EINVAL (there is no silent fallback: the materialised form is the caller's to choose). */
int pdae_conv2d_wgrad_gn_ok(const pdae_conv_desc* d);
int pdae_conv_gn_input_arm(const float* coef, int act);

/* ---- strided-batched GEMM (F.linear, torch.einsum of module.py:450-457 / 479-488, and their backward)
 * C[b][m][n] (+)= alpha * sum_k opA[b][m][k] * opB[b][k][n] + bias[n];  opA = A[m*lda+k] (transA=0) | A[k*lda+m] (1);
 * opB = B[k*ldb+n] (transB=0) | B[n*ldb+k] (1).  batch b = bo*batch_inner + bi -> offset bo*s?o + bi*s?i.  (1,1) unsupported. */
int pdae_gemm(int transA, int transB, int M, int N, int K, float alpha, const float* A, int64_t lda, int64_t sAo, int64_t sAi, const float* B,
              int64_t ldb, int64_t sBo, int64_t sBi, float* C, int64_t ldc, int64_t sCo, int64_t sCi, int batch_outer, int batch_inner,
              const float* bias, int accumulate, pdae_stream_t stream);

/* A family of small linear layers in ONE launch: y_i[M][n_out_i] = x_i[M][K] W_i[n_out_i][K]^T + bias_i (M <= 32, exact fp32 FMA).  Every
 * ResBlock's emb_layers / emb_z_layers Linear (model/module.py:287-293, 371-380) reads SiLU(emb) / SiLU(shift_emb), which exist before the
 * first block: one call per network pass instead of one launch per block.  items and first_feature are DEVICE arrays; first_feature[i] is
 * the position of item i's first output feature in the concatenated feature list, first_feature[n_items] == total_features. */
typedef struct pdae_linear_item { const float* x; const float* w; const float* bias; float* y; int32_t n_out; int32_t rows; /* 0: M; else 1..32 rows of THIS item (a batch > 32 is a list of row slices) */ } pdae_linear_item;
int pdae_linear_group(const pdae_linear_item* items, const int32_t* first_feature, int n_items, int total_features, int M, int K, pdae_stream_t stream);

/* Backward of such a family in ONE launch: dw_i (+)= dy_i^T x_i, db_i (+)= column sums of dy_i (acc_w selects "+="), and where dx_i != NULL
 * dx_i (+)= dy_i W_i (acc_x).  At most ONE item of a call may write a given dx (no cross-block reduction).  first_block[i] = index of item i's
 * first thread block: an item takes ceil(n_out/8) blocks, plus M * ceil(K/128) more when it has a dx; first_block[n_items] = total_blocks. */
typedef struct pdae_linear_bwd_item {
  const float* x; const float* dy; const float* w; float* dw; float* db; float* dx; int32_t n_out, acc_w, acc_x, reserved;
} pdae_linear_bwd_item;
int pdae_linear_bwd_group(const pdae_linear_bwd_item* items, const int32_t* first_block, int n_items, int total_blocks, int M, int K,
                          pdae_stream_t stream);

/* ---- GroupNorm(32,C) + AdaGN + SiLU (+Dropout, +AvgPool2d) : module.py:56-63,241,257-263,279-284,293-294,379-381 */
size_t pdae_gn_workspace_bytes(int N, int C);
int pdae_gn_stats(const float* x0, int C0, const float* x1, int C1, int N, int HW, int G, float eps, float* mean, float* rstd, void* ws,
                  pdae_stream_t stream);
/* coef[3][N][C] = (mu, a, b) such that v = a*(x-mu)+b equals (1+zs)*((GN(x))*(1+s)+sh)+zsh; ss/zss = [N][2C] (scale|shift) or NULL */
/* pdae_gn_stats followed by pdae_gn_coef with the finalize and the coefficient fold in ONE launch (two launches instead of three) */
/* ticket (pdae_gn_stats_coef, pdae_gn_bwd; optional): N + 1 uint32 words, ZEROED ONCE by the caller and owned by one stream.  With it the
 * partial-sum kernel's last block per sample runs the finalize stage itself (agent-scope release / acquire hand-off), i.e. statistics +
 * coefficients are ONE launch and the backward's reduce + finalize + parameter-gradient stages are ONE launch; the words return to zero. */
int pdae_gn_stats_coef(const float* x0, int C0, const float* x1, int C1, int N, int HW, int G, float eps, const float* gamma, const float* beta,
                       const float* ss, const float* zss, float* mean, float* rstd, float* coef, void* ws, uint32_t* ticket, pdae_stream_t stream);
int pdae_gn_coef(int N, int C, int G, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* ss,
                 const float* zss, float* coef, pdae_stream_t stream);
/* y = act(v) * dropmask; act: 0 identity, 1 SiLU.  mode 0: same size; mode 1: y (and xpool = raw x) are 2x2 average pooled. */
int pdae_gn_apply(const float* x0, int C0, const float* x1, int C1, int N, int H, int W, const float* coef, int act, int mode, float* y,
                  float* xpool, float drop_p, uint64_t seed, uint64_t offset, pdae_stream_t stream);
/* backward of stats+coef+apply.  dA = grad wrt y (mode 0: [N,H,W,C]; 1: [N,H/2,W/2,C]; 2: [N,2H,2W,C] when the consumer read y upsampled).
 * add (optional, resampled like dA) is added into dx (identity skip path).  dx0/dx1 receive the two concat halves (NULL = skip). */
int pdae_gn_bwd(const float* x0, int C0, const float* x1, int C1, int N, int H, int W, int G, const float* coef, const float* rstd,
                const float* gamma, const float* beta, const float* ss, const float* zss, const float* dA, int act, int mode, float drop_p,
                uint64_t seed, uint64_t offset, const float* add, float* dx0, int acc0, float* dx1, int acc1, float* dgamma, float* dbeta,
                int acc_param, float* dss, float* dzss, void* ws, float* dx0_amax, uint32_t* ticket, pdae_stream_t stream);
/* dx0_amax (optional): device scalar that receives max|dx0| -- the dy_amax of the convolution whose output gradient dx0 is (saves the
 * separate pdae_amax pass over it). */

/* ---- MLPSkipNet layer body (model/mlp_skip_net.py:123-141): per row of a [R][C] activation
 *   y = act( LayerNorm_C( u * (1 + e) ) * gamma + beta );  e may be NULL (no condition), norm = 0 skips the LayerNorm, act: 1 = SiLU.
 * fwd stores the row statistics; bwd writes du, de and the per-element terms tg = dv*xhat, tb = dv whose column sums (pdae_colsum)
 * are d gamma / d beta. */
int pdae_mlp_modln_fwd(const float* u, const float* e, const float* gamma, const float* beta, int R, int C, int norm, int act, float eps, float* y,
                       float* mean, float* rstd, pdae_stream_t stream);
int pdae_mlp_modln_bwd(const float* u, const float* e, const float* gamma, const float* beta, const float* mean, const float* rstd, const float* dy,
                       int R, int C, int norm, int act, float* du, float* de, float* tg, float* tb, pdae_stream_t stream);

/* ---- small elementwise pieces */
int pdae_timestep_embedding(const int64_t* t, const float* freqs, int N, int dim, float* out, pdae_stream_t stream); /* module.py:66-84 */
int pdae_silu(const float* x, float* y, size_t n, pdae_stream_t stream);
int pdae_silu_bwd(const float* x, const float* dy, float* dx, size_t n, int accumulate, pdae_stream_t stream);
/* dense-grid form of a stride-2 3x3 convolution (encoder/ffhq.py:18-31 `nn.Conv2d(.., 3, 2, 1)`): the stride-1 convolution on the patch kernels
 * plus y[n,oy,ox,:] = x[n,2oy,2ox,:] (forward) / dY scattered onto the even grid of a zero tensor (backward).  NHWC, H and W even, C % 4 == 0. */
int pdae_subsample2(const float* x, int N, int H, int W, int C, float* y, pdae_stream_t stream);
int pdae_zero_insert2(const float* x, int N, int Ho, int Wo, int C, float* y, pdae_stream_t stream);
int pdae_axpby(const float* x, float* y, size_t n, float alpha, float beta, pdae_stream_t stream);
int pdae_embedding(const float* table, const int64_t* idx, int N, int D, float* out, int accumulate, pdae_stream_t stream); /* unet.py:190-192 */
int pdae_embedding_bwd(const float* dout, const int64_t* idx, int N, int D, float* dtable, pdae_stream_t stream);
int pdae_to_nhwc(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int N, int C, int H, int W, float* y, pdae_stream_t stream);
int pdae_from_nhwc(const float* x, int N, int C, int H, int W, float* y, int64_t sn, int64_t sc, int64_t sh, int64_t sw, pdae_stream_t stream);
int pdae_softmax(float* s, int64_t rows, int T, pdae_stream_t stream);                         /* module.py:455 */
int pdae_softmax_bwd(const float* p, float* dp, int64_t rows, int T, pdae_stream_t stream);
size_t pdae_colsum_workspace_bytes(int64_t M, int C);
int pdae_colsum(const float* x, int64_t M, int C, float* out, int accumulate, void* ws, pdae_stream_t stream); /* bias gradients */

/* ---- diffusion (gaussian_diffusion.py / ddim.py) */
int pdae_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac, int N, size_t per_sample,
                  float* xt, pdae_stream_t stream);                                             /* gaussian_diffusion.py:98-103 */
/* loss = scale*mean(w[t]*(noise-(eps+c[t]*g))^2) (or |.| when l1); deps/dg = dloss/d(eps|g) or NULL; ws >= 4 KiB */
int pdae_loss(const float* noise, const float* eps, const float* g, const int64_t* t, const float* shift_coef, const float* weight, int N,
              size_t per_sample, int l1, float scale, float* loss, float* deps, float* dg, void* ws, pdae_stream_t stream); /* :166-175,246-251 */
int pdae_ddim_step(const float* x, const float* eps, const float* g, size_t total, float c_shift, float sqrt_recip_ac, float sqrt_recip_ac_m1,
                   float sqrt_ac_to, float sqrt_1m_ac_to, int clamp, float* out, pdae_stream_t stream);    /* ddim.py:46-55,94-107,126-138 */
int pdae_ddpm_step(const float* x, const float* eps, const float* g, const float* z, size_t total, float cx, float ce, float cs, float sigma,
                   float* out, pdae_stream_t stream);                                            /* gaussian_diffusion.py:112-126 */

/* Per-sample-timestep forms of the single-step API (the reference takes t[B]: ddim.py:43-55,66-79,91-107,123-138; gaussian_diffusion.py:105-126,
 * 148-164).  Coefficient rows are device arrays gathered from the schedule tables with t, so no value of t is ever read on the host.
 *   axpby_rows:      out[n,:] = ca[n]*a[n,:] + cb[n]*b[n,:]   (q_posterior_mean :105-108, predicted_noise_to_predicted_x_0 :156-159, _mean :161-164)
 *   ddim_step_rows:  coef[n] = {c_shift, sqrt_recip_ac, sqrt_recip_ac_m1, sqrt(ac_to), sqrt(1-ac_to)}
 *   ddpm_step_rows:  coef[n] = {cx, ce, cs, mask, lv_min, lv_max}; out = cx*x - ce*(eps + cs*g) + mask*exp(0.5*lv)*noise with lv = lv_min, or
 *                    lv_min + (learned_range+1)/2*(lv_max-lv_min) when learned_range != NULL (learn_sigma models, unet.py:49) */
int pdae_axpby_rows(const float* a, const float* b, const float* ca, const float* cb, int N, size_t per_sample, float* out, pdae_stream_t stream);
int pdae_ddim_step_rows(const float* x, const float* eps, const float* g, const float* coef, int N, size_t per_sample, int clamp, float* out,
                        pdae_stream_t stream);
int pdae_ddpm_step_rows(const float* x, const float* eps, const float* g, const float* noise, const float* learned_range, const float* coef, int N,
                        size_t per_sample, float* out, pdae_stream_t stream);

/* ---- fused attention core of AttentionBlock (module.py:422-428, QKVAttentionLegacy :431-457 / QKVAttention :460-488) and its backward.
 * qkv [N][T][3C] (T = H*W pixels, the output of the qkv 1x1 convolution in NHWC), out [N][T][C]; new_order = use_new_attention_order
 * ([q(all heads)|k|v] channel blocks instead of per-head [q|k|v]).  softmax(q k^T / sqrt(ch)) v per (image, head) in ONE launch: the
 * T x T probabilities never leave the CU.  lse [N*heads][T] (log-sum-exp of the scaled scores; NULL when no backward follows) is all the
 * backward needs besides qkv / out: it recomputes the probabilities.  ws of pdae_attn_bwd: N*heads*T floats.  Exact fp32 (f32 MFMA).
 * Supported: T in {64,128,192,256}, head width C/heads a multiple of 32 (pdae_attn_fused_ok); other shapes: the strided-batched pdae_gemm +
 * pdae_softmax composition. */
int pdae_attn_fused_ok(int T, int C, int heads);
int pdae_attn_fwd(const float* qkv, int N, int T, int C, int heads, int new_order, float* out, float* lse, pdae_stream_t stream);
int pdae_attn_bwd(const float* qkv, const float* out, const float* lse, const float* d_out, int N, int T, int C, int heads, int new_order, float* dqkv,
                  void* ws, pdae_stream_t stream);

/* ---- evaluator (sampler/autoencoding_eval.py:83-99): per-image SSIM and MSE of two (N,C,H,W)-shaped batches of ANY strides in one pass.
 * Both inputs are mapped v -> v*mul + add first (the (x+1)/2 of autoencoding_eval.py:83-88: mul = add = 0.5).  SSIM = metric/utils.py:35-57
 * (11-tap Gaussian window, given normalised as window11 so that the caller controls its rounding; zero padding; C1 = 1e-4, C2 = 9e-4; mean over
 * C,H,W), MSE = metric/utils.py:62-63.  ssim / mse: [N] outputs (either may be NULL). */
size_t pdae_ssim_mse_workspace_bytes(int N, int C, int H, int W);
int pdae_ssim_mse(const float* a, const int64_t* a_strides, const float* b, const int64_t* b_strides, int N, int C, int H, int W, float mul, float add,
                  const float* window11, float* ssim, float* mse, void* ws, pdae_stream_t stream);

/* ---- input pipeline (dataset/ffhq.py:19-31,46; dataset/celeba64.py:11-34): a batch of decoded uint8 images [B][Hs][Ws][C] ->
 * crop -> PIL-exact antialiased bilinear resize to S_out x S_out (22-bit fixed point, uint8 rounding after each pass) -> optional per-image
 * horizontal flip -> x0 = (v/255 - 0.5)/0.5 written through x0_strides (element strides of the (B,C,S,S) view: NCHW or the NHWC plan buffer)
 * and gts [B][S][S][C] uint8 (may be NULL).  coef_* [S_out][ksize] int32 and bounds_* [S_out][2] = (first input index, tap count) are
 * built on the host like PIL's precompute_coeffs (pdae_amd/dataset/resample.py).  ws: pdae_image_prepare_workspace_bytes bytes. */
size_t pdae_image_prepare_workspace_bytes(int B, int crop_h, int S_out, int C);
int pdae_image_prepare(const uint8_t* src, int B, int Hs, int Ws, int C, int crop_y, int crop_x, int crop_h, int crop_w, int S_out,
                       const int32_t* coef_x, const int32_t* bounds_x, int ksize_x, const int32_t* coef_y, const int32_t* bounds_y, int ksize_y,
                       const uint8_t* flip, float* x0, const int64_t* x0_strides, uint8_t* gts, void* ws, pdae_stream_t stream);

/* ---- optimizer: torch.optim.Adam / AdamW + EMA (train_representation_learning.py:58-70,192-212) over a flat segment */
/* guard (optional) = device {saturation counter, skipped-step counter} (pdae_set_saturation_counter): while guard[0] != 0 the update is NOT
 * applied (a convolution of this step clamped an operand into the fp16 window, so its gradients are not fp32-grade) and, when count_skip != 0,
 * guard[1] is incremented -- the GradScaler-style "skip the step" of torch.cuda.amp, decided on the device without a host sync. */
int pdae_adam_ema(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int decoupled, float step_size, float inv_sqrt_bc2, float grad_scale, float ema_decay,
                  unsigned int* guard, int count_skip, pdae_stream_t stream);

/* ---- data-parallel gradient exchange: DistributedDataParallel's bucketed all-reduce (train_representation_learning.py:29,39) as plain calls.
 * One process per GPU.  Rank 0 obtains the 128-byte id (pdae_comm_unique_id) and hands it to the other ranks over the host's control channel;
 * every rank then calls pdae_comm_init (collective) with the device it will use current.  pdae_allreduce_bucket(comm, buf, count, dtype, op,
 * stream) reduces `count` elements IN PLACE across the ranks on `stream` (dtype 0 = float32, 1 = int32; op 0 = sum, 1 = max), stream-ordered
 * like any kernel: enqueue it on a side stream behind an event of the compute stream as soon as a gradient bucket is final, and make the
 * optimizer wait on an event recorded after the last bucket.  RCCL is dlopen'ed on first use (librccl_path may be NULL: a copy already
 * mapped into the process -- e.g. PyTorch's -- is preferred, then the system's). */
int pdae_comm_unique_id(const char* librccl_path, void* id128);
int pdae_comm_init(const char* librccl_path, const void* id128, int nranks, int rank, void** comm);
int pdae_allreduce_bucket(void* comm, void* buf, size_t count, int dtype, int op, pdae_stream_t stream);
int pdae_comm_destroy(void* comm);

/* ---- planned-graph executor: a network pass is a static array of ops, issued back-to-back on one stream
 * with a single host call (replaces the per-module Python dispatch of TimestepSequential, module.py:131-140). */
enum {
  PDAE_OP_CONV_FWD = 1, PDAE_OP_CONV_DGRAD, PDAE_OP_CONV_WGRAD, PDAE_OP_GEMM, PDAE_OP_GN_STATS, PDAE_OP_GN_COEF, PDAE_OP_GN_APPLY,
  PDAE_OP_GN_BWD, PDAE_OP_TEMB, PDAE_OP_SILU, PDAE_OP_SILU_BWD, PDAE_OP_AXPBY, PDAE_OP_EMBEDDING, PDAE_OP_EMBEDDING_BWD, PDAE_OP_TO_NHWC,
  PDAE_OP_FROM_NHWC, PDAE_OP_Q_SAMPLE, PDAE_OP_LOSS, PDAE_OP_DDIM_STEP, PDAE_OP_DDPM_STEP, PDAE_OP_ADAM_EMA, PDAE_OP_SOFTMAX,
  PDAE_OP_SOFTMAX_BWD, PDAE_OP_COLSUM, PDAE_OP_MEMSET, PDAE_OP_COPY, PDAE_OP_CONV_WPREP, PDAE_OP_MLP_MODLN_FWD, PDAE_OP_MLP_MODLN_BWD, PDAE_OP_CONV_FWD_GN, PDAE_OP_CONV_FWD_SKIP, PDAE_OP_GN_STATS_COEF, PDAE_OP_CONV_SKIP_WPREP, PDAE_OP_AMAX,
  PDAE_OP_AXPBY_ROWS, PDAE_OP_DDIM_STEP_ROWS, PDAE_OP_DDPM_STEP_ROWS, PDAE_OP_LINEAR_GROUP,
  PDAE_OP_ATTN_FWD, PDAE_OP_ATTN_BWD, PDAE_OP_LINEAR_BWD_GROUP, PDAE_OP_GN_COEF_FROM_CONV_STATS, PDAE_OP_CONV_WPREP_GROUP,
  PDAE_OP_SUBSAMPLE2, PDAE_OP_ZERO_INSERT2, PDAE_OP_GN_STATS_QUADS,
  PDAE_OP_JOIN     /* (ABI 10) the caller's stream waits for every PDAE_OPF_SIDE op issued so far; no arguments */
};
/* pdae_op.flags (ABI 10; the field was `reserved`, always 0).  PDAE_OPF_SIDE: issue this op on the executor's second (low-priority) stream.
 * It starts when the ops in front of it in the array have finished, and runs beside the ops behind it until the next PDAE_OP_JOIN or the end
 * of the pdae_run_ops call (which joins).  The caller guarantees that no op between a side op and the next join overwrites the side op's
 * inputs or touches its outputs.  Knob PDAE_SIDE_STREAM=0: the flag is ignored (everything in order on the caller's stream). */
#define PDAE_OPF_SIDE 1
typedef struct pdae_op {
  int32_t kind;
  int32_t flags;   /* PDAE_OPF_* (0: none) */
  void* p[20];     /* pointer arguments, in the order of the corresponding function's pointer parameters */
  int64_t i[24];   /* integer arguments, in order (a pdae_conv_desc is flattened to its 14 fields) */
  double f[12];    /* floating-point arguments, in order */
} pdae_op;
/* returns 0, or the first failing op's status (its index is in pdae_last_error()). */
int pdae_run_ops(const pdae_op* ops, int n, pdae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
