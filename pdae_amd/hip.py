"""ctypes binding of libpdae_hip.so (include/pdae_hip.h) + op-record builders.

The product path has NO CPU fallback: importing this module on a machine where the
library has not been built raises, and every launch raises on a non-zero status.
PyTorch supplies device memory and streams only (tensor.data_ptr(), current stream).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDAE_HIP_LIB") or os.path.join(_HERE, "lib", "libpdae_hip.so")     # PDAE_HIP_LIB: timing-probe builds (tools/probe_build.py)

(OP_CONV_FWD, OP_CONV_DGRAD, OP_CONV_WGRAD, OP_GEMM, OP_GN_STATS, OP_GN_COEF, OP_GN_APPLY, OP_GN_BWD, OP_TEMB, OP_SILU,
 OP_SILU_BWD, OP_AXPBY, OP_EMBEDDING, OP_EMBEDDING_BWD, OP_TO_NHWC, OP_FROM_NHWC, OP_Q_SAMPLE, OP_LOSS, OP_DDIM_STEP,
 OP_DDPM_STEP, OP_ADAM_EMA, OP_SOFTMAX, OP_SOFTMAX_BWD, OP_COLSUM, OP_MEMSET, OP_COPY, OP_CONV_WPREP, OP_MLP_MODLN_FWD, OP_MLP_MODLN_BWD, OP_CONV_FWD_GN, OP_CONV_FWD_SKIP, OP_GN_STATS_COEF, OP_CONV_SKIP_WPREP, OP_AMAX,
 OP_AXPBY_ROWS, OP_DDIM_STEP_ROWS, OP_DDPM_STEP_ROWS, OP_LINEAR_GROUP, OP_ATTN_FWD, OP_ATTN_BWD, OP_LINEAR_BWD_GROUP, OP_GN_COEF_FROM_CONV_STATS, OP_CONV_WPREP_GROUP, OP_SUBSAMPLE2,
 OP_ZERO_INSERT2, OP_GN_STATS_QUADS, OP_JOIN) = range(1, 48)
OPF_SIDE = 1            # PdaeOp.flags: issue on the executor's second stream (pdae_hip.h)


class PdaeOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("flags", ctypes.c_int32), ("p", ctypes.c_void_p * 20),
                ("i", ctypes.c_int64 * 24), ("f", ctypes.c_double * 12)]


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("N", "Hi", "Wi", "C0", "C1", "Ho", "Wo", "Cout", "KH", "KW", "stride", "pad", "up", "math")]


MATH_F32, MATH_BF16, MATH_BF16X3, MATH_BF16X6 = 0, 1, 2, 3      # pdae_conv_desc.math
MATH_NAMES = {"f32": 0, "bf16": 1, "bf16x3": 2, "bf16x6": 3, "f16x3": 4}
MATH_DIRECT = 0x100                                               # PDAE_MATH_DIRECT (pdae_hip.h)
# default arithmetic of the conv GEMMs on fp32 tensors.  Every fp32 operand is split into low-precision planes whose leading cross
# products are accumulated in fp32 on the MFMA pipe:
#   "f16x3"  (default): the 3x3 kernels (forward, data gradient, weight gradient) use two fp16 planes (11+11 mantissa bits, 3 products);
#            power-of-two operand scales keep both planes normal -- static for weights / normalised activations, per-tensor dynamic
#            (pdae_amax) for gradients.  Measured 2.8e-7 relative error vs fp64 on a 256->128 conv (torch fp32: 2.3e-7).  The 1x1 and the
#            generic kernels run "bf16x6".
#   "bf16x6": three exact bf16 planes, 6 products (3.7e-7 on the same conv) everywhere -- range-safe for any input.
# Both pass the same parity gates as the exact f32-MFMA kernels ("f32"), which remain selectable with PDAE_CONV_MATH.
DEFAULT_MATH = "f16x3"
_math_override = None


def default_math():
    """Name of the arithmetic mode new plans are built with: set_default_math() > $PDAE_CONV_MATH > DEFAULT_MATH."""
    return _math_override or os.environ.get("PDAE_CONV_MATH", DEFAULT_MATH)


def set_default_math(name):
    """Process-wide override (None restores the environment / built-in default).  Existing plans keep the mode they were built with:
    call net.invalidate_plans() / rebuild the fused step to move them."""
    global _math_override
    assert name is None or name in MATH_NAMES, name
    _math_override = name


class PdaeError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads the shared library (once).  Raises if it is missing: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PdaeError(f"{LIB_PATH} is missing -- run `python -m pdae_amd.build` (hipcc, gfx950). "
                            "pdae_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        L.pdae_last_error.restype = ctypes.c_char_p
        L.pdae_run_ops.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.pdae_run_ops.restype = ctypes.c_int
        L.pdae_conv2d_wgrad_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
        L.pdae_conv2d_wgrad_workspace_bytes.restype = ctypes.c_size_t
        L.pdae_conv_wprep_bytes.argtypes = [ctypes.POINTER(ConvDesc), ctypes.c_int]
        L.pdae_conv_wprep_bytes.restype = ctypes.c_size_t
        L.pdae_conv_skip_wprep_bytes.restype = ctypes.c_size_t
        L.pdae_gn_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
        L.pdae_gn_workspace_bytes.restype = ctypes.c_size_t
        L.pdae_colsum_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
        L.pdae_colsum_workspace_bytes.restype = ctypes.c_size_t
        L.pdae_abi_version.restype = ctypes.c_int
        L.pdae_conv_gnbwd_bytes.argtypes = [ctypes.POINTER(ConvDesc), ctypes.c_int, ctypes.c_void_p]
        L.pdae_conv_gnbwd_bytes.restype = ctypes.c_size_t
        L.pdae_conv_stats_bytes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.pdae_conv_stats_bytes.restype = ctypes.c_size_t
        L.pdae_ssim_mse_workspace_bytes.restype = ctypes.c_size_t
        L.pdae_ssim_mse_workspace_bytes.argtypes = [ctypes.c_int] * 4
        L.pdae_ssim_mse.restype = ctypes.c_int
        L.pdae_ssim_mse.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)] + [ctypes.c_int] * 4 + \
            [ctypes.c_float, ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.pdae_image_prepare_workspace_bytes.restype = ctypes.c_size_t
        L.pdae_image_prepare_workspace_bytes.argtypes = [ctypes.c_int] * 4
        L.pdae_image_prepare.restype = ctypes.c_int
        L.pdae_image_prepare.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 9 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                                                 ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.pdae_set_knob.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.pdae_set_knob.restype = ctypes.c_int
        L.pdae_get_knob.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        L.pdae_get_knob.restype = ctypes.c_int
        L.pdae_set_saturation_counter.argtypes = [ctypes.c_void_p]
        L.pdae_set_saturation_counter.restype = ctypes.c_int
        _lib = L
    return _lib


EXPORTS = ["pdae_last_error", "pdae_abi_version", "pdae_set_knob", "pdae_get_knob", "pdae_set_saturation_counter", "pdae_conv2d_fwd", "pdae_conv_gnbwd_bytes", "pdae_conv_gnbwd_arm", "pdae_gn_bwd_parts_arm", "pdae_conv2d_dgrad", "pdae_conv2d_wgrad_workspace_bytes",
           "pdae_conv2d_wgrad", "pdae_conv2d_wgrad_form", "pdae_conv2d_wgrad_gn_ok", "pdae_conv_gn_input_arm", "pdae_conv_wprep_bytes", "pdae_conv3x3_form", "pdae_conv_wprep", "pdae_conv2d_fwd_gn", "pdae_conv2d_fwd_skip_ok", "pdae_conv_skip_wprep_bytes", "pdae_conv_skip_wprep", "pdae_conv_wprep_job", "pdae_conv_skip_wprep_job", "pdae_conv_wprep_group", "pdae_conv2d_fwd_skip", "pdae_conv_stats_bytes", "pdae_conv_stats_arm", "pdae_gn_stats_quads", "pdae_gn_coef_from_conv_stats", "pdae_gemm", "pdae_gn_workspace_bytes", "pdae_gn_stats", "pdae_gn_stats_coef", "pdae_gn_coef", "pdae_gn_apply", "pdae_gn_bwd",
           "pdae_mlp_modln_fwd", "pdae_mlp_modln_bwd", "pdae_timestep_embedding", "pdae_amax", "pdae_silu", "pdae_silu_bwd", "pdae_subsample2", "pdae_zero_insert2", "pdae_axpby", "pdae_embedding", "pdae_embedding_bwd", "pdae_to_nhwc",
           "pdae_from_nhwc", "pdae_softmax", "pdae_softmax_bwd", "pdae_colsum_workspace_bytes", "pdae_colsum", "pdae_linear_bwd_group", "pdae_comm_unique_id", "pdae_comm_init", "pdae_allreduce_bucket", "pdae_comm_destroy", "pdae_linear_group", "pdae_attn_fused_ok", "pdae_attn_fwd", "pdae_attn_bwd", "pdae_q_sample", "pdae_loss",
           "pdae_ddim_step", "pdae_ddpm_step", "pdae_axpby_rows", "pdae_ddim_step_rows", "pdae_ddpm_step_rows", "pdae_adam_ema", "pdae_run_ops",
           "pdae_ssim_mse_workspace_bytes", "pdae_ssim_mse", "pdae_image_prepare_workspace_bytes", "pdae_image_prepare"]


class SaturationGuard:
    """fp16-window guard of the default "f16x3" arithmetic (include/pdae_hip.h: pdae_set_saturation_counter).

    One int32[2] device tensor per process: [0] = number of math-4 convolution launches that met a scaled operand outside the fp16
    window (|x| > 60000 after the power-of-two pre-scale: post-GroupNorm activations beyond ~3750, a raw residual stream beyond 6e4,
    or Inf), [1] = optimizer steps the device refused to apply because [0] was non-zero.  Reading it synchronises: callers poll it at
    their own cadence (trainers: every display interval; samplers: once per loop) and fall back to "bf16x6" when it fires."""
    _inst = None

    def __init__(self, device):
        self.device = torch.device(device)
        self.t = torch.zeros(2, dtype=torch.int32, device=self.device)
        rc = lib().pdae_set_saturation_counter(ctypes.c_void_p(self.t.data_ptr()))
        if rc != 0:
            raise PdaeError("pdae_set_saturation_counter failed")

    @classmethod
    def get(cls, device):
        device = torch.device(device)
        if device.type != "cuda":
            return None
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if cls._inst is None or cls._inst.device != device:
            cls._inst = cls(device)
        return cls._inst

    def read(self):
        """(saturated launches, skipped optimizer steps) -- host sync."""
        a = self.t.tolist()
        return int(a[0]) & 0xffffffff, int(a[1]) & 0xffffffff

    def reset(self):
        self.t.zero_()

    def ptr(self):
        return self.t.data_ptr()


def set_knob(name, value):
    """Sets a tuning / A-B switch of the library (pdae_set_knob; names = the environment variables of DESIGN.md section 10, which are only read once, at
    a knob's first use).  Do not change PDAE_W1 between preparing a convolution's weights and launching it: the launch is refused."""
    rc = lib().pdae_set_knob(name.encode(), int(value))
    if rc != 0:
        raise PdaeError(f"pdae_set_knob failed ({rc}): {lib().pdae_last_error().decode()}")


def get_knob(name):
    v = ctypes.c_int(0)
    rc = lib().pdae_get_knob(name.encode(), ctypes.byref(v))
    if rc != 0:
        raise PdaeError(f"pdae_get_knob failed ({rc}): {lib().pdae_last_error().decode()}")
    return int(v.value)


def saturated(device="cuda"):
    """True when a math-4 convolution has overflowed its fp16 window since the last reset (host sync)."""
    g = SaturationGuard.get(device)
    return g is not None and g.read()[0] != 0


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def make_op(kind, p=(), i=(), f=()):
    o = PdaeOp()
    o.kind = kind
    for k, v in enumerate(p):
        o.p[k] = _ptr(v)
    for k, v in enumerate(i):
        o.i[k] = int(v)
    for k, v in enumerate(f):
        o.f[k] = float(v)
    return o


def op_join():
    """The caller's stream waits for every side-stream op issued so far (PDAE_OP_JOIN)."""
    return make_op(OP_JOIN)


def current_stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def run_ops(arr, n, stream=None):
    """arr: ctypes array of PdaeOp (or a single PdaeOp with n == 1)."""
    st = current_stream_ptr() if stream is None else stream
    rc = lib().pdae_run_ops(ctypes.addressof(arr), n, ctypes.c_void_p(st))
    if rc != 0:
        raise PdaeError(f"pdae_run_ops failed (status {rc}): {lib().pdae_last_error().decode()}")


def run(op, stream=None):
    run_ops(op, 1, stream)


def ops_array(ops):
    arr = (PdaeOp * len(ops))()
    for k, o in enumerate(ops):
        ctypes.memmove(ctypes.addressof(arr) + k * ctypes.sizeof(PdaeOp), ctypes.addressof(o), ctypes.sizeof(PdaeOp))
    return arr


# ------------------------------------------------------------------------------------------
# record builders: one per PDAE_OP_* (field order documented in include/pdae_hip.h / api.hip)
# ------------------------------------------------------------------------------------------
class Conv:
    """Geometry of one convolution over NHWC activations (pdae_conv_desc)."""

    def __init__(self, N, Hi, Wi, C0, C1, Cout, k=3, stride=1, pad=None, up=False, math=0, direct=False):
        pad = k // 2 if pad is None else pad
        Hl, Wl = (2 * Hi, 2 * Wi) if up else (Hi, Wi)
        self.N, self.Hi, self.Wi, self.C0, self.C1, self.Cout = N, Hi, Wi, C0, C1, Cout
        self.KH = self.KW = k
        self.stride, self.pad, self.up = stride, pad, int(up)
        self.math = int(math) & 0xff       # MATH_F32 / MATH_BF16 / MATH_BF16X3 / MATH_BF16X6 / f16x3
        # PDAE_MATH_DIRECT: the FORWARD form of this 3x3 convolution stays on the direct kernels (eligible for fused skip chunks); set before the
        # weights are prepared, travels in the descriptor's math field
        self.direct = bool(direct) or bool(int(math) & MATH_DIRECT)
        self.Ho = (Hl + 2 * pad - k) // stride + 1
        self.Wo = (Wl + 2 * pad - k) // stride + 1
        self.Hl, self.Wl = Hl, Wl

    @property
    def Cin(self):
        return self.C0 + self.C1

    def fields(self):
        return [self.N, self.Hi, self.Wi, self.C0, self.C1, self.Ho, self.Wo, self.Cout, self.KH, self.KW, self.stride, self.pad, self.up,
                self.math | (MATH_DIRECT if self.direct else 0)]

    def cdesc(self):
        d = ConvDesc()
        for n, v in zip([f[0] for f in ConvDesc._fields_], self.fields()):
            setattr(d, n, v)
        return d

    def winograd_form(self, transposed=0, gn=False, f16_grad=False):
        """True when this convolution's prepared copy / launch are in the Winograd F(2, 3)-along-x form (pdae_conv3x3_form)."""
        d = self.cdesc()
        return bool(lib().pdae_conv3x3_form(ctypes.byref(d), int(transposed) | (4 if gn else 0) | (16 if f16_grad else 0)))

    def wprep_bytes(self, transposed=0, force=False, gn=False, f16_grad=False):
        """Size of the prepared-weight copy for the patch kernel; 0 = not eligible (generic kernel runs).
        gn: for the fused-GroupNorm forward (op_conv_fwd_gn); pass flags = 4 to op_conv_wprep as well."""
        d = self.cdesc()
        return int(lib().pdae_conv_wprep_bytes(ctypes.byref(d), int(transposed) | (2 if force else 0) | (4 if gn else 0) | (16 if f16_grad else 0)))

    def wgrad_ws_bytes(self):
        d = self.cdesc()
        return int(lib().pdae_conv2d_wgrad_workspace_bytes(ctypes.byref(d)))


def _arm_stats(op, stats):
    """stats: partial-sum buffer of pdae_conv_stats_bytes(c) bytes -> pointer slot 19 of a forward-convolution record (pdae_conv_stats_arm)."""
    if stats is not None:
        op.p[19] = _ptr(stats)
    return op


def op_conv_fwd(c, x0, x1, w, bias, y, res=None, res_mode=0, tile=0, wp=None, stats=None):
    return _arm_stats(make_op(OP_CONV_FWD, [x0, x1, w, bias, res, y, wp], c.fields() + [res_mode, tile]), stats)


def op_conv_fwd_gn(c, x0, x1, coef, act, wp, bias, y, res=None, res_mode=0, stats=None):
    """conv of act(GN-affine(x)): GroupNorm/AdaGN(+SiLU) applied in the patch staging (pdae_conv2d_fwd_gn)."""
    return _arm_stats(make_op(OP_CONV_FWD_GN, [x0, x1, coef, wp, bias, res, y], c.fields() + [res_mode, act]), stats)


def op_conv_fwd_skip(c, x0, x1, coef, act, wp, bias, cs, s0, s1, wps, bias_s, y, stats=None):
    """y = conv3x3_c(in) + bias + conv1x1_cs([s0 | s1]) + bias_s in one launch (pdae_conv2d_fwd_skip)."""
    return _arm_stats(make_op(OP_CONV_FWD_SKIP, [x0, x1, coef, wp, bias, s0, s1, wps, bias_s, y], c.fields() + [act, cs.C0, cs.C1]), stats)


def conv_stats_bytes(c, cs=None):
    """(bytes, wave-tiles per image) of the GroupNorm partial statistics the forward convolution c (with fused skip cs) can leave behind while
    it stores its output (split-K launches: from their slab reduction); (0, 0) when it does not run on the 3x3 patch kernels (pdae_conv_stats_bytes)."""
    d = c.cdesc()
    tpi = ctypes.c_int32(0)
    if cs is None:
        b = lib().pdae_conv_stats_bytes(ctypes.byref(d), None, ctypes.byref(tpi))
    else:
        ds = cs.cdesc()
        b = lib().pdae_conv_stats_bytes(ctypes.byref(d), ctypes.byref(ds), ctypes.byref(tpi))
    return int(b), int(tpi.value)


def op_conv_skip_wprep(c, cs, w_skip, wps):
    """skip_connection weights in the plane format / scale of conv c's main loop (pdae_conv_skip_wprep)."""
    return make_op(OP_CONV_SKIP_WPREP, [w_skip, wps], c.fields() + [cs.C0, cs.C1])


def conv_skip_wprep_bytes(c, cs):
    d, ds = c.cdesc(), cs.cdesc()
    return int(lib().pdae_conv_skip_wprep_bytes(ctypes.byref(d), ctypes.byref(ds)))


def conv_fwd_skip_ok(c, cs):
    d, ds = c.cdesc(), cs.cdesc()
    return bool(lib().pdae_conv2d_fwd_skip_ok(ctypes.byref(d), ctypes.byref(ds)))


def op_conv_dgrad(c, dy, w, dx, ci_off=0, ci_cnt=None, accumulate=0, tile=0, wp_t=None, dy_amax=None, gnb=None):
    """gnb = (x0, C0, x1, C1, coef, part): the launch also leaves the GroupNorm-backward sums of dx (pdae_conv_gnbwd_arm; conv_gnbwd_bytes(c) != 0)."""
    op = make_op(OP_CONV_DGRAD, [dy, w, dx, wp_t, dy_amax], c.fields() + [ci_off, c.Cin if ci_cnt is None else ci_cnt, accumulate, tile])
    if gnb is not None:
        x0, C0, x1, C1, coef, part = gnb
        op.p[5], op.p[6], op.p[7], op.p[8] = _ptr(x0), _ptr(x1), _ptr(coef), _ptr(part)
        op.i[18], op.i[19], op.i[20] = int(C0), int(C1), 1
    return op


def conv_gnbwd_bytes(c, f16_grad=False):
    """(bytes, tiles per image) of the GroupNorm-backward partial sums the data gradient of c can leave in its epilogue; (0, 0): not available."""
    d = c.cdesc()
    t = ctypes.c_int32(0)
    b = lib().pdae_conv_gnbwd_bytes(ctypes.byref(d), 16 if f16_grad else 0, ctypes.byref(t))
    return int(b), int(t.value)


def op_amax(x, n, out):
    """out[0] = max |x| (device scalar): feeds the power-of-two dY scale of the fp16-format gradient kernels."""
    return make_op(OP_AMAX, [x, out], [n])


def op_conv_wprep(c, w, transposed, wp):
    """Pre-split w into the MFMA-fragment-ordered bf16 planes the patch kernel reads (pdae_conv_wprep)."""
    return make_op(OP_CONV_WPREP, [w, wp], c.fields() + [transposed])


class WprepJob(ctypes.Structure):
    """pdae_wprep_job (include/pdae_hip.h): one prepared-weight copy of the grouped launch."""
    _fields_ = [("w", ctypes.c_void_p), ("wp", ctypes.c_void_p), ("Nout", ctypes.c_int32), ("C", ctypes.c_int32), ("NT", ctypes.c_int32),
                ("transposed", ctypes.c_int32), ("T", ctypes.c_int32), ("ns", ctypes.c_int32), ("wscale", ctypes.c_float), ("nblocks", ctypes.c_int32)]


def wprep_job(c, w, flags, wp):
    """Job record of pdae_conv_wprep(c, w, flags, wp) for the grouped launch (pdae_conv_wprep_job: filled by the library, nothing runs)."""
    d, j = c.cdesc(), WprepJob()
    rc = lib().pdae_conv_wprep_job(ctypes.byref(d), ctypes.c_void_p(_ptr(w)), int(flags), ctypes.c_void_p(_ptr(wp)), ctypes.byref(j))
    if rc != 0:
        raise PdaeError(f"pdae_conv_wprep_job failed ({rc}): {lib().pdae_last_error().decode()}")
    return j


def skip_wprep_job(c, cs, w_skip, wps):
    d, ds, j = c.cdesc(), cs.cdesc(), WprepJob()
    rc = lib().pdae_conv_skip_wprep_job(ctypes.byref(d), ctypes.byref(ds), ctypes.c_void_p(_ptr(w_skip)), ctypes.c_void_p(_ptr(wps)), ctypes.byref(j))
    if rc != 0:
        raise PdaeError(f"pdae_conv_skip_wprep_job failed ({rc}): {lib().pdae_last_error().decode()}")
    return j


def wprep_group_tables(jobs, device):
    """(job table, prefix table, total blocks) on `device` for op_conv_wprep_group."""
    import torch
    raw = b"".join(bytes(j) for j in jobs)
    first, tot = [], 0
    for j in jobs:
        first.append(tot)
        tot += int(j.nblocks)
    jt = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    ft = torch.tensor(first, dtype=torch.int32).to(device)
    return jt, ft, tot


def op_conv_wprep_group(jobs_t, first_t, njobs, total_blocks):
    """Every prepared-weight copy of a plan in one launch (pdae_conv_wprep_group)."""
    return make_op(OP_CONV_WPREP_GROUP, [jobs_t, first_t], [njobs, total_blocks])


def op_conv_wgrad(c, x0, x1, dy, dw, ws, ws_bytes, accumulate=0, db=None, dy_amax=None, gn_coef=None, gn_act=1):
    """dw (+)= weight gradient; db (optional): bias gradient = column sums of dy, same accumulate flag.
    gn_coef: (x0, x1) are the RAW sources of a fused-GroupNorm forward (op_conv_fwd_gn); the kernel recomputes act(a (x - mu) + b) while it
    stages X (pdae_conv_gn_input_arm; conv_wgrad_gn_ok(c) must hold)."""
    return make_op(OP_CONV_WGRAD, [x0, x1, dy, dw, ws, db, dy_amax, gn_coef], c.fields() + [accumulate, ws_bytes, gn_act if gn_coef is not None else 0])


def conv_wgrad_form(c, with_dy_amax=True, with_gn_input=False):
    """3 = conv3x3v, 2 = conv3x3w, 1 = another dedicated kernel, 0 = generic implicit GEMM (pdae_conv2d_wgrad_form)."""
    d = c.cdesc()
    return int(lib().pdae_conv2d_wgrad_form(ctypes.byref(d), int(with_dy_amax), int(with_gn_input)))


def conv_wgrad_gn_ok(c):
    d = c.cdesc()
    return bool(lib().pdae_conv2d_wgrad_gn_ok(ctypes.byref(d)))


def op_gemm(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, alpha=1.0, bias=None, accumulate=0,
            batch_outer=1, batch_inner=1, sA=(0, 0), sB=(0, 0), sC=(0, 0)):
    return make_op(OP_GEMM, [A, B, C, bias],
                   [transA, transB, M, N, K, lda, sA[0], sA[1], ldb, sB[0], sB[1], ldc, sC[0], sC[1], batch_outer, batch_inner, accumulate],
                   [alpha])


def linear_group_tables(items, device):
    """Device tables of pdae_linear_group for items = [(x, w, bias, y[, rows])] (tensors): (items int64 [n,5], first int32 [n+1], total features)."""
    rows, first, tot = [], [0], 0
    for it in items:
        x, w, b, y = it[:4]
        nrows = int(it[4]) if len(it) > 4 else 0                     # rows of this item (0: the launch's M)
        n_out = int(w.shape[0])
        rows.append([x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else 0, y.data_ptr(), n_out | (nrows << 32)])     # n_out | rows<<32 in one int64 (little endian)
        tot += n_out
        first.append(tot)
    return (torch.tensor(rows, dtype=torch.int64).to(device), torch.tensor(first, dtype=torch.int32).to(device), tot)


def linear_bwd_group_tables(items, M, device, K=None):
    """Device tables of pdae_linear_bwd_group for items = [(x, dy, w, dw, db, dx, acc_w, acc_x)]: (items int64 [n,8], first int32 [n+1], total blocks)."""
    rows, first, tot = [], [0], 0
    for x, dy, w, dw, db, dx, acc_w, acc_x in items:
        n_out = int(w.shape[0])
        p = lambda t: t.data_ptr() if t is not None else 0
        rows.append([p(x), p(dy), p(w), p(dw), p(db), p(dx), n_out | (int(acc_w) << 32), int(acc_x)])     # two int32 per int64 word (little endian)
        kk = int(w.shape[1]) if K is None else K
        tot += (n_out + 7) // 8 + (M * ((kk // 4 + 31) // 32) if dx is not None else 0)
        first.append(tot)
    return (torch.tensor(rows, dtype=torch.int64).to(device), torch.tensor(first, dtype=torch.int32).to(device), tot)


def op_linear_bwd_group(items_t, first_t, n_items, total_blocks, M, K):
    return make_op(OP_LINEAR_BWD_GROUP, [items_t, first_t], [n_items, total_blocks, M, K])


def op_linear_group(items_t, first_t, n_items, total, M, K):
    return make_op(OP_LINEAR_GROUP, [items_t, first_t], [n_items, total, M, K])


def attn_fused_ok(T, C, heads):
    return bool(lib().pdae_attn_fused_ok(int(T), int(C), int(heads)))


def op_attn_fwd(qkv, N, T, C, heads, new_order, out, lse=None):
    return make_op(OP_ATTN_FWD, [qkv, out, lse], [N, T, C, heads, int(new_order)])


def op_attn_bwd(qkv, out, lse, d_out, N, T, C, heads, new_order, dqkv, ws):
    return make_op(OP_ATTN_BWD, [qkv, out, lse, d_out, dqkv, ws], [N, T, C, heads, int(new_order)])


def op_gn_stats(x0, C0, x1, C1, N, HW, G, eps, mean, rstd, ws):
    return make_op(OP_GN_STATS, [x0, x1, mean, rstd, ws], [C0, C1, N, HW, G], [eps])


def op_gn_stats_coef(x0, C0, x1, C1, N, HW, G, eps, gamma, beta, ss, zss, mean, rstd, coef, ws, ticket=None):
    return make_op(OP_GN_STATS_COEF, [x0, x1, gamma, beta, ss, zss, mean, rstd, coef, ws, ticket], [C0, C1, N, HW, G], [eps])


def op_gn_coef_from_conv_stats(N, HW, C0, C1, G, eps, part0, tpi0, part1, tpi1, gamma, beta, ss, zss, mean, rstd, coef):
    """mean / rstd / coef of [x0 | x1] from the partial sums their producing convolutions wrote (pdae_gn_coef_from_conv_stats)."""
    return make_op(OP_GN_COEF_FROM_CONV_STATS, [part0, part1, gamma, beta, ss, zss, mean, rstd, coef], [N, HW, C0, C1, G, tpi0, tpi1], [eps])


def op_gn_stats_quads(x, N, HW, C, tiles, part):
    """Partial (sum, sum of squares) of x[N][HW][C] per (image, run of HW / tiles pixels, channel quad) in the format of the convolution epilogues
    (pdae_gn_stats_quads): feeds op_gn_coef_from_conv_stats."""
    return make_op(OP_GN_STATS_QUADS, [x, part], [N, HW, C, tiles])


def op_gn_coef(N, C, G, mean, rstd, gamma, beta, ss, zss, coef):
    return make_op(OP_GN_COEF, [mean, rstd, gamma, beta, ss, zss, coef], [N, C, G])


def op_gn_apply(x0, C0, x1, C1, N, H, W, coef, act, mode, y, xpool=None, drop_p=0.0, seed=0, offset=0):
    return make_op(OP_GN_APPLY, [x0, x1, coef, y, xpool], [C0, C1, N, H, W, act, mode, seed, offset], [drop_p])


def op_gn_bwd(x0, C0, x1, C1, N, H, W, G, coef, rstd, gamma, beta, ss, zss, dA, act, mode, ws, add=None, dx0=None, acc0=0,
              dx1=None, acc1=0, dgamma=None, dbeta=None, acc_param=0, dss=None, dzss=None, drop_p=0.0, seed=0, offset=0, dx0_amax=None, ticket=None,
              parts=None, parts_tiles=0):
    """parts: the per-(n, c) sums were left by the data gradient that wrote dA (op_conv_dgrad(gnb=...)): no reduction pass (pdae_gn_bwd_parts_arm)."""
    return make_op(OP_GN_BWD, [x0, x1, coef, rstd, gamma, beta, ss, zss, dA, add, dx0, dx1, dgamma, dbeta, dss, dzss, ws, dx0_amax, ticket, parts],
                   [C0, C1, N, H, W, G, act, mode, acc0, acc1, acc_param, seed, offset, parts_tiles if parts is not None else 0], [drop_p])


def op_mlp_modln_fwd(u, e, gamma, beta, R, C, norm, act, eps, y, mean, rstd):
    return make_op(OP_MLP_MODLN_FWD, [u, e, gamma, beta, y, mean, rstd], [R, C, norm, act], [eps])


def op_mlp_modln_bwd(u, e, gamma, beta, mean, rstd, dy, R, C, norm, act, du, de, tg, tb):
    return make_op(OP_MLP_MODLN_BWD, [u, e, gamma, beta, mean, rstd, dy, du, de, tg, tb], [R, C, norm, act])


def op_temb(t, freqs, N, dim, out):
    return make_op(OP_TEMB, [t, freqs, out], [N, dim])


def op_silu(x, y, n):
    return make_op(OP_SILU, [x, y], [n])


def op_subsample2(x, N, Hh, W, C, y):
    """y[n, oy, ox, :] = x[n, 2 oy, 2 ox, :] (pdae_subsample2)."""
    return make_op(OP_SUBSAMPLE2, [x, y], [N, Hh, W, C])


def op_zero_insert2(x, N, Ho, Wo, C, y):
    """y [N, 2 Ho, 2 Wo, C] = x scattered onto the even grid, zero elsewhere (pdae_zero_insert2)."""
    return make_op(OP_ZERO_INSERT2, [x, y], [N, Ho, Wo, C])


def op_silu_bwd(x, dy, dx, n, acc=0):
    return make_op(OP_SILU_BWD, [x, dy, dx], [n, acc])


def op_axpby(x, y, n, alpha=1.0, beta=1.0):
    return make_op(OP_AXPBY, [x, y], [n], [alpha, beta])


def op_embedding(table, idx, N, D, out, acc=0):
    return make_op(OP_EMBEDDING, [table, idx, out], [N, D, acc])


def op_embedding_bwd(dout, idx, N, D, dtable):
    return make_op(OP_EMBEDDING_BWD, [dout, idx, dtable], [N, D])


def op_to_nhwc(x, strides, N, C, H, W, y):
    return make_op(OP_TO_NHWC, [x, y], list(strides) + [N, C, H, W])


def op_from_nhwc(x, N, C, H, W, y, strides):
    return make_op(OP_FROM_NHWC, [x, y], list(strides) + [N, C, H, W])


def op_q_sample(x0, noise, t, ta, tb, N, per, xt):
    return make_op(OP_Q_SAMPLE, [x0, noise, t, ta, tb, xt], [N, per])


def op_loss(noise, eps, g, t, tc, tw, N, per, loss, ws, deps=None, dg=None, l1=0, scale=1.0):
    return make_op(OP_LOSS, [noise, eps, g, t, tc, tw, loss, deps, dg, ws], [N, per, l1], [scale])


def op_ddim_step(x, eps, g, total, c_shift, ra, rm1, sab, s1ab, out, clamp=1):
    return make_op(OP_DDIM_STEP, [x, eps, g, out], [total, clamp], [c_shift, ra, rm1, sab, s1ab])


def op_ddpm_step(x, eps, g, z, total, cx, ce, cs, sigma, out):
    return make_op(OP_DDPM_STEP, [x, eps, g, z, out], [total], [cx, ce, cs, sigma])


def op_axpby_rows(a, b, ca, cb, N, per, out):
    return make_op(OP_AXPBY_ROWS, [a, b, ca, cb, out], [N, per])


def op_ddim_step_rows(x, eps, g, coef, N, per, out, clamp=1):
    return make_op(OP_DDIM_STEP_ROWS, [x, eps, g, coef, out], [N, per, clamp])


def op_ddpm_step_rows(x, eps, g, noise, learned_range, coef, N, per, out):
    return make_op(OP_DDPM_STEP_ROWS, [x, eps, g, noise, learned_range, coef, out], [N, per])


def op_adam_ema(p, g, m, v, ema, n, lr, b1, b2, eps, wd, decoupled, step_size, inv_sqrt_bc2, grad_scale, ema_decay, guard=None, count_skip=0):
    """guard: SaturationGuard tensor pointer -- the kernel leaves everything untouched while guard[0] != 0 (and counts the skip in guard[1])."""
    return make_op(OP_ADAM_EMA, [p, g, m, v, ema, guard], [n, decoupled, count_skip], [lr, b1, b2, eps, wd, step_size, inv_sqrt_bc2, grad_scale, ema_decay])


def op_softmax(s, rows, T):
    return make_op(OP_SOFTMAX, [s], [rows, T])


def op_softmax_bwd(p, dp, rows, T):
    return make_op(OP_SOFTMAX_BWD, [p, dp], [rows, T])


def op_colsum(x, M, C, out, ws, acc=0):
    return make_op(OP_COLSUM, [x, out, ws], [M, C, acc])


def op_memset(dst, nbytes):
    return make_op(OP_MEMSET, [dst], [nbytes])


def op_copy(src, dst, nbytes):
    return make_op(OP_COPY, [src, dst], [nbytes])


def gn_ws_bytes(N, C):
    return int(lib().pdae_gn_workspace_bytes(N, C))


def colsum_ws_bytes(M, C):
    return int(lib().pdae_colsum_workspace_bytes(M, C))
