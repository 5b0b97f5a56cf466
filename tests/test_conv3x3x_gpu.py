"""GPU: csrc/conv3x3x.hip -- the 3x3 patch kernel with the Winograd F(2, 3) transform along x (12 instead of 18 products per pixel pair) --
against fp64 references and against the direct patch kernels on the same operands.  PDAE_W1 = 2 routes every eligible shape to it (prepared
weights AND launch: both are decided by conv3x3p_form, so each case prepares its weights under the same setting it launches with), 0 keeps the
direct kernels.  Gates: the direct kernels' own (1e-5 of max |y| in the f16x3 arithmetic, 5e-4 bf16x3, 2e-2 bf16).
Covers what conv3x3r's tests cover: bias / residual (same- and half-resolution) / nearest-upsampled input, fused GroupNorm input (one and two
sources, AdaGN), fused 1x1 skip chunks with output statistics, the data gradient with the dynamic fp16 scale, accumulate, multi-tile images
whose borders put zero padding on every side (and on the ACTIVATED tensor under GroupNorm)."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = {1: 2e-2, 2: 5e-4, 4: 1e-5}


@pytest.fixture
def H():
    from pdae_amd import hip
    return hip


def rn(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).double().cpu()


WMODE = "2"


@pytest.fixture(autouse=True, params=["2", "3"], ids=["rows16", "rows8"])
def _wmode(request):
    """Every test of this file runs twice: the Winograd form in tiles of 16 rows (PDAE_W1 = 2) and in tiles of 8 rows (PDAE_W1 = 3, round 5: the
    variant that takes the 16 x 16-pixel layers of the FFHQ-128 step)."""
    global WMODE
    WMODE = request.param
    yield


def _both(knob, run):
    """run() = weight preparation + launch, under the direct kernels (PDAE_W1=0) and under the Winograd-along-x form (PDAE_W1 = WMODE)."""
    out = []
    for on in ("0", WMODE):
        knob("PDAE_W1", on)
        out.append(run())
        torch.cuda.synchronize()
    return out


def _gn_ref(x, gamma, beta, ss, G=32):
    N, C = x.shape[:2]
    xn = F.group_norm(x, G, None, None, 1e-5) * gamma.view(1, C, 1, 1) + beta.view(1, C, 1, 1)
    if ss is not None:
        sc, sh = ss[:, :C], ss[:, C:]
        xn = xn * (1 + sc.view(N, C, 1, 1)) + sh.view(N, C, 1, 1)
    return F.silu(xn)


@pytest.mark.parametrize("math_mode", [4, 1, 2])
@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout, up, res_mode
    (2, 32, 16, 32, 128, 0, 0),            # one tile column, one chunk
    (1, 64, 48, 96, 256, 0, 1),            # 4 x 3 tiles, 3 chunks, two 128-channel tiles, same-resolution residual
    (3, 32, 32, 64, 128, 1, 2),            # nearest-upsampled input (stored 16 x 16) + half-resolution residual
    (2, 96, 32, 128, 128, 0, 0),           # interior tile rows see no zero padding at top / bottom
    (5, 80, 112, 64, 256, 0, 1),           # 350 tiles x 2 channel tiles
])
def test_forward_vs_fp64_and_direct(H, knob, case, math_mode):
    N, Hh, W, C, Cout, up, res_mode = case
    if N * Hh * W > 30000 and math_mode != 4:
        pytest.skip("large cases in the default arithmetic only")
    Hs, Ws = (Hh // 2, W // 2) if up else (Hh, W)
    x = rn(1, N, C, Hs, Ws) * 1.3 + 0.2
    w = rn(2, Cout, C, 3, 3, scale=1.0 / math.sqrt(9 * C)); b = rn(3, Cout, scale=0.2)
    c = H.Conv(N, Hs, Ws, C, 0, Cout, k=3, up=bool(up), math=math_mode)
    xl = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    y_ref = F.conv2d(xl.double(), w.double(), b.double(), padding=1)
    res = None
    if res_mode == 1:
        res = rn(4, N, Cout, Hh, W); y_ref = y_ref + res.double()
    elif res_mode == 2:
        res = rn(4, N, Cout, Hh // 2, W // 2); y_ref = y_ref + F.interpolate(res, scale_factor=2, mode="nearest").double()
    xd, wd, bd = nhwc(x).cuda(), nhwc(w).cuda(), b.cuda()
    resd = nhwc(res).cuda() if res is not None else None

    def run():
        wp = torch.empty(c.wprep_bytes(0, force=True) // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 0, wp))
        y = torch.full((N, Hh, W, Cout), float("nan"), device="cuda")
        H.run(H.op_conv_fwd(c, xd, None, wd, bd, y, res=resd, res_mode=res_mode, wp=wp))
        return y
    y_p, y_x = _both(knob, run)
    e_p, e_x = rel_err(nchw(y_p), y_ref), rel_err(nchw(y_x), y_ref)
    print(f"[conv3x3x fwd math {math_mode}] {case}: vs fp64 {e_x:.2e} (direct {e_p:.2e})")
    assert e_x < TOL[math_mode], (e_x, e_p)
    # (a shape the direct plan splits over K keeps the direct form -- conv3x3p_form -- and is then bit-identical; the others differ in rounding)
    if N * Hh * W * Cout >= 256 * 256 * 128 * 2:
        assert not torch.equal(y_p, y_x)
    assert rel_err(y_x, y_p) < 3 * TOL[math_mode]


@pytest.mark.parametrize("case", [
    # N, H, W, C0, C1, Cout, up, res_mode, AdaGN
    (2, 32, 32, 64, 0, 128, 0, 1, True),
    (2, 64, 16, 64, 32, 128, 0, 0, False),     # two-source concat
    (1, 32, 32, 32, 0, 256, 1, 2, True),       # upsampled input, half-resolution residual
    (3, 96, 128, 32, 32, 128, 0, 1, True),
])
def test_fused_groupnorm_input_vs_fp64(H, knob, case):
    N, Hh, W, C0, C1, Cout, up, res_mode, ada = case
    C, G = C0 + C1, 32
    Hs, Ws = (Hh // 2, W // 2) if up else (Hh, W)
    x = rn(1, N, C, Hs, Ws) * 1.5 + 0.7
    gamma, beta = 1 + 0.2 * rn(2, C), 0.2 * rn(3, C) + 0.5          # beta offset: a wrong zero padding of the ACTIVATED tensor would show
    ss = 0.3 * rn(4, N, 2 * C) if ada else None
    w = rn(5, Cout, C, 3, 3, scale=1.0 / math.sqrt(C * 9)); b = rn(6, Cout, scale=0.1)
    c = H.Conv(N, Hs, Ws, C0, C1, Cout, k=3, up=bool(up), math=4)
    a_ref = _gn_ref(x.double(), gamma.double(), beta.double(), None if ss is None else ss.double())
    y_ref = F.conv2d(F.interpolate(a_ref, scale_factor=2, mode="nearest") if up else a_ref, w.double(), b.double(), padding=1)
    res = None
    if res_mode == 1:
        res = rn(7, N, Cout, Hh, W); y_ref = y_ref + res.double()
    elif res_mode == 2:
        res = rn(7, N, Cout, Hh // 2, W // 2); y_ref = y_ref + F.interpolate(res, scale_factor=2, mode="nearest").double()
    xh = nhwc(x).cuda()
    x0 = xh[..., :C0].contiguous(); x1 = xh[..., C0:].contiguous() if C1 else None
    mean, rstd, coef = torch.empty(N * G, device="cuda"), torch.empty(N * G, device="cuda"), torch.empty(3, N, C, device="cuda")
    ws = torch.empty(H.gn_ws_bytes(N, C) // 4 + 64, device="cuda")
    H.run(H.op_gn_stats_coef(x0, C0, x1, C1, N, Hs * Ws, G, 1e-5, gamma.cuda(), beta.cuda(), None if ss is None else ss.cuda(), None, mean, rstd, coef, ws))
    wd, bd = nhwc(w).cuda(), b.cuda()
    resd = nhwc(res).cuda() if res is not None else None

    def run():
        wp = torch.empty(c.wprep_bytes(0, force=True, gn=True) // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 4, wp))
        y = torch.full((N, Hh, W, Cout), float("nan"), device="cuda")
        H.run(H.op_conv_fwd_gn(c, x0, x1, coef, 1, wp, bd, y, res=resd, res_mode=res_mode))
        return y
    y_p, y_x = _both(knob, run)
    e_x = rel_err(nchw(y_x), y_ref)
    print(f"[conv3x3x fused GN] {case}: vs fp64 {e_x:.2e} (direct {rel_err(nchw(y_p), y_ref):.2e})")
    assert e_x < 1e-5


@pytest.mark.parametrize("case", [(16, 64, 32, 32, 96, 0, 128), (8, 64, 64, 96, 32, 32, 256)])
def test_fused_skip_chunks_exist_in_the_direct_form_only(H, knob, case):
    """A convolution launched in the Winograd-along-x form takes no fused 1x1 skip chunks: pdae_conv2d_fwd_skip_ok says no (the caller computes the
    skip convolution separately and hands it in as the residual -- engine.resblock), pdae_conv_skip_wprep_bytes is 0 and the launch itself refuses.
    The same pair IS eligible when the form is off."""
    N, Hh, W, C, Cs0, Cs1, Cout = case
    c = H.Conv(N, Hh, W, C, 0, Cout, k=3, math=4)
    cs = H.Conv(N, Hh, W, Cs0, Cs1, Cout, k=1, math=4)
    knob("PDAE_W1", "0")
    assert H.conv_fwd_skip_ok(c, cs) and H.conv_skip_wprep_bytes(c, cs) > 0
    nb_skip = H.conv_skip_wprep_bytes(c, cs)
    knob("PDAE_W1", WMODE)
    assert not H.conv_fwd_skip_ok(c, cs) and H.conv_skip_wprep_bytes(c, cs) == 0
    x = torch.randn(N, Hh, W, C, device="cuda"); s0 = torch.randn(N, Hh, W, Cs0, device="cuda")
    s1 = torch.randn(N, Hh, W, Cs1, device="cuda") if Cs1 else None
    w = torch.randn(Cout, 3, 3, C, device="cuda") * 0.05; b = torch.zeros(Cout, device="cuda")
    wp = torch.empty(c.wprep_bytes(0, force=True) // 4, device="cuda")
    H.run(H.op_conv_wprep(c, w, 0, wp))
    wps = torch.zeros(nb_skip // 4, device="cuda")
    y = torch.empty(N, Hh, W, Cout, device="cuda")
    with pytest.raises(H.PdaeError):
        H.run(H.op_conv_fwd_skip(c, x, None, None, 0, wp, b, cs, s0, s1, wps, b, y))
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", [(16, 64, 32, 32, 96, 0, 128, False), (8, 64, 64, 96, 32, 32, 256, True)])
def test_direct_bit_in_the_descriptor_keeps_the_fused_skip(H, knob, case):
    """pdae_conv_desc.math | PDAE_MATH_DIRECT (hip.Conv(direct=True)): the FORWARD form of that convolution stays direct under PDAE_W1 -- prepared weights
    and launch agree because both read the same descriptor --, so the fused skip launch is offered and correct; the data gradient through the same
    descriptor ignores the bit and stays in the Winograd form.  (engine.Builder._skip_parts pins wide-skip ResBlocks this way.)"""
    knob("PDAE_W1", WMODE)
    N, Hh, W, C, Cs0, Cs1, Cout, use_gn = case
    Cs, G = Cs0 + Cs1, 32
    x = rn(1, N, C, Hh, W) * 1.2 + 0.3
    sx = rn(2, N, Cs, Hh, W) * 2.0
    w = rn(3, Cout, C, 3, 3, scale=1.0 / math.sqrt(C * 9)); b = rn(4, Cout, scale=0.1)
    wsk = rn(5, Cout, Cs, 1, 1, scale=1.0 / math.sqrt(Cs)); bsk = rn(6, Cout, scale=0.1)
    gamma, beta = 1 + 0.2 * rn(7, C), 0.2 * rn(8, C) + 0.4
    c = H.Conv(N, Hh, W, C, 0, Cout, k=3, math=4, direct=True)
    cw = H.Conv(N, Hh, W, C, 0, Cout, k=3, math=4)
    cs = H.Conv(N, Hh, W, Cs0, Cs1, Cout, k=1, math=4)
    assert cw.winograd_form(0) and not c.winograd_form(0)
    assert H.conv_fwd_skip_ok(c, cs) and not H.conv_fwd_skip_ok(cw, cs)
    if Cout % 128 == 0 and C % 128 == 0:
        assert c.winograd_form(1, f16_grad=True) == cw.winograd_form(1, f16_grad=True)      # the data gradient does not see the bit
    a_ref = _gn_ref(x.double(), gamma.double(), beta.double(), None) if use_gn else x.double()
    y_ref = F.conv2d(a_ref, w.double(), b.double(), padding=1) + F.conv2d(sx.double(), wsk.double(), bsk.double())
    xd, sh = nhwc(x).cuda(), nhwc(sx).cuda()
    s0 = sh[..., :Cs0].contiguous(); s1 = sh[..., Cs0:].contiguous() if Cs1 else None
    wd, wsd = nhwc(w).cuda(), nhwc(wsk).cuda()
    coef = None
    if use_gn:
        mean, rstd, coef = torch.empty(N * G, device="cuda"), torch.empty(N * G, device="cuda"), torch.empty(3, N, C, device="cuda")
        ws = torch.empty(H.gn_ws_bytes(N, C) // 4 + 64, device="cuda")
        H.run(H.op_gn_stats_coef(xd, C, None, 0, N, Hh * W, G, 1e-5, gamma.cuda(), beta.cuda(), None, None, mean, rstd, coef, ws))
    wp = torch.empty(c.wprep_bytes(0, force=True, gn=use_gn) // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 4 if use_gn else 0, wp))
    wps = torch.empty(H.conv_skip_wprep_bytes(c, cs) // 4, device="cuda")
    H.run(H.op_conv_skip_wprep(c, cs, wsd, wps))
    y = torch.full((N, Hh, W, Cout), float("nan"), device="cuda")
    H.run(H.op_conv_fwd_skip(c, xd, None, coef, 1, wp, b.cuda(), cs, s0, s1, wps, bsk.cuda(), y))
    assert rel_err(nchw(y), y_ref) < 1e-5
    # the same weights prepared through the plain descriptor are in the other layout
    wp2 = torch.empty_like(wp)
    H.run(H.op_conv_wprep(cw, wd, 4 if use_gn else 0, wp2))
    assert not torch.equal(wp, wp2)


@pytest.mark.parametrize("case", [(16, 64, 32, 64, 128, True, 0), (16, 64, 32, 32, 128, False, 1), (8, 64, 64, 96, 256, True, 1)])
def test_output_statistics_from_the_epilogue(H, knob, case):
    """The GroupNorm partial statistics of the output ((sum, sum of squares) per wave tile and channel quad, written by the epilogue: plain and
    fused-GroupNorm launches, with and without a residual), after the reader's fp64 combine (pdae_gn_coef_from_conv_stats)."""
    N, Hh, W, C, Cout, use_gn, res_mode = case
    G = 32
    x = rn(1, N, C, Hh, W) * 1.2 + 0.3
    w = rn(3, Cout, C, 3, 3, scale=1.0 / math.sqrt(C * 9)); b = rn(4, Cout, scale=0.1)
    gamma, beta = 1 + 0.2 * rn(7, C), 0.2 * rn(8, C) + 0.4
    res = rn(5, N, Cout, Hh, W) if res_mode else None
    c = H.Conv(N, Hh, W, C, 0, Cout, k=3, math=4)
    a_ref = _gn_ref(x.double(), gamma.double(), beta.double(), None) if use_gn else x.double()
    y_ref = F.conv2d(a_ref, w.double(), b.double(), padding=1) + (res.double() if res_mode else 0.0)
    xd, wd = nhwc(x).cuda(), nhwc(w).cuda()
    resd = nhwc(res).cuda() if res_mode else None
    coef = None
    if use_gn:
        mean, rstd, coef = torch.empty(N * G, device="cuda"), torch.empty(N * G, device="cuda"), torch.empty(3, N, C, device="cuda")
        ws = torch.empty(H.gn_ws_bytes(N, C) // 4 + 64, device="cuda")
        H.run(H.op_gn_stats_coef(xd, C, None, 0, N, Hh * W, G, 1e-5, gamma.cuda(), beta.cuda(), None, None, mean, rstd, coef, ws))
    g2, b2 = (1 + 0.1 * rn(9, Cout)).cuda(), (0.1 * rn(10, Cout)).cuda()

    def run():
        nbytes, tpi = H.conv_stats_bytes(c, None)
        assert nbytes > 0
        wp = torch.empty(c.wprep_bytes(0, force=True, gn=use_gn) // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 4 if use_gn else 0, wp))
        y = torch.full((N, Hh, W, Cout), float("nan"), device="cuda")
        part = torch.full((nbytes // 4,), float("nan"), device="cuda")
        if use_gn:
            H.run(H.op_conv_fwd_gn(c, xd, None, coef, 1, wp, b.cuda(), y, res=resd, res_mode=res_mode, stats=part))
        else:
            H.run(H.op_conv_fwd(c, xd, None, wd, b.cuda(), y, res=resd, res_mode=res_mode, wp=wp, stats=part))
        m, r, k = torch.empty(N * G, device="cuda"), torch.empty(N * G, device="cuda"), torch.empty(3, N, Cout, device="cuda")
        H.run(H.op_gn_coef_from_conv_stats(N, Hh * W, Cout, 0, G, 1e-5, part, tpi, None, 0, g2, b2, None, None, m, r, k))
        return y, part, m, r
    (y_p, part_p, m_p, r_p), (y_x, part_x, m_x, r_x) = _both(knob, run)
    e_x = rel_err(nchw(y_x), y_ref)
    print(f"[conv3x3x statistics] {case}: vs fp64 {e_x:.2e} (direct {rel_err(nchw(y_p), y_ref):.2e})")
    assert e_x < 1e-5
    assert torch.isfinite(part_x).all()
    yd = y_x.double()
    mean_ref = yd.view(N, Hh * W, G, Cout // G).mean((1, 3)).flatten()
    assert (m_x.double() - mean_ref).abs().max() < 5e-6 * max(1.0, float(mean_ref.abs().max()))
    var_ref = yd.view(N, Hh * W, G, Cout // G).var((1, 3), unbiased=False).flatten()
    assert rel_err(r_x, (var_ref + 1e-5).rsqrt()) < 2e-5


@pytest.mark.parametrize("gscale", [1.0, 3e-7, 2e4])
@pytest.mark.parametrize("case", [(2, 32, 32, 128, 64), (1, 64, 32, 256, 128), (3, 160, 160, 128, 32)])
def test_data_gradient_with_dynamic_fp16_scale(H, knob, case, gscale):
    """dX of a 3x3 convolution = the same kernel on transposed, tap-flipped prepared weights (wprepx_slot's transposed form) with the
    per-tensor power-of-two dY scale: Cin of the convolution is the GEMM N here, so it must be a multiple of 128."""
    N, Hh, W, Cin, Cout = case
    x = rn(1, N, Cin, Hh, W)
    w = rn(2, Cout, Cin, 3, 3, scale=1.0 / math.sqrt(9 * Cin))
    dy = rn(3, N, Cout, Hh, W) * gscale
    xr = x.double().requires_grad_(True)
    (F.conv2d(xr, w.double(), None, padding=1) * dy.double()).sum().backward()
    c = H.Conv(N, Hh, W, Cin, 0, Cout, k=3, math=4)
    wd, dyd = nhwc(w).cuda(), nhwc(dy).cuda()
    amax = torch.empty(4, device="cuda")
    H.run(H.op_amax(dyd, dyd.numel(), amax))

    def run():
        wp_t = torch.empty(c.wprep_bytes(1, force=True, f16_grad=True) // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 1 | 16, wp_t))
        dx = torch.full((N, Hh, W, Cin), float("nan"), device="cuda")
        H.run(H.op_conv_dgrad(c, dyd, wd, dx, wp_t=wp_t, dy_amax=amax))
        return dx
    dx_p, dx_x = _both(knob, run)
    e_x = rel_err(nchw(dx_x), xr.grad)
    print(f"[conv3x3x dgrad] {case} x{gscale}: vs fp64 {e_x:.2e} (direct {rel_err(nchw(dx_p), xr.grad):.2e})")
    assert e_x < 1e-5


def test_accumulating_data_gradient(H, knob):
    """accumulate = 1 (a second consumer's gradient joins the buffer): read-modify-write epilogue."""
    N, Hh, W, Cin, Cout = 2, 32, 16, 128, 32
    w = rn(2, Cout, Cin, 3, 3, scale=1.0 / math.sqrt(9 * Cin))
    dy = rn(3, N, Cout, Hh, W)
    base = rn(4, N, Hh, W, Cin)
    c = H.Conv(N, Hh, W, Cin, 0, Cout, k=3, math=4)
    wd, dyd = nhwc(w).cuda(), nhwc(dy).cuda()

    def run():
        wp_t = torch.empty(c.wprep_bytes(1, force=True) // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 1, wp_t))
        dx = base.clone().cuda()
        H.run(H.op_conv_dgrad(c, dyd, wd, dx, accumulate=1, wp_t=wp_t))
        return dx
    dx_p, dx_x = _both(knob, run)
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    assert rel_err(nchw(dx_x) - base.permute(0, 3, 1, 2).double(), ref) < 1e-5


def test_grouped_weight_preparation_uses_the_same_form(H, knob):
    """pdae_conv_wprep_job / pdae_conv_wprep_group (one launch for every prepared copy of a plan) must write the same Winograd-form planes as
    pdae_conv_wprep: forward, fused-GroupNorm forward and data-gradient (transposed, fp16 gradient format) jobs."""
    knob("PDAE_W1", WMODE)
    N, Hh, W, C, Cout = 32, 64, 64, 64, 128
    w = nhwc(rn(1, Cout, C, 3, 3, scale=0.05)).cuda()
    c = H.Conv(N, Hh, W, C, 0, Cout, k=3, math=4)
    cd = H.Conv(N, Hh, W, Cout, 0, C, k=3, math=4)       # its data gradient has GEMM N = Cout of this descriptor... use a 128-input-channel conv
    wd2 = nhwc(rn(3, C, Cout, 3, 3, scale=0.05)).cuda()  # weights of cd: (Cout = C, Cin = Cout)
    singles, jobs = [], []
    for (cc, ww, flags) in ((c, w, 0), (c, w, 4), (cd, wd2, 1 | 16)):
        a = torch.zeros(cc.wprep_bytes(flags & 1, force=True, gn=bool(flags & 4), f16_grad=bool(flags & 16)) // 4, device="cuda")
        b = torch.zeros_like(a)
        H.run(H.op_conv_wprep(cc, ww, flags, a))
        singles.append(a); jobs.append((H.wprep_job(cc, ww, flags, b), b))
    jt, ft, tot = H.wprep_group_tables([j for j, _ in jobs], torch.device("cuda"))
    H.run(H.op_conv_wprep_group(jt, ft, len(jobs), tot))
    torch.cuda.synchronize()
    for s_, (_, g_) in zip(singles, jobs):
        assert torch.equal(s_, g_) and float(s_.abs().max()) > 0
    knob("PDAE_W1", "0")
    a0 = torch.zeros_like(singles[0])
    H.run(H.op_conv_wprep(c, w, 0, a0))
    assert not torch.equal(a0, singles[0])               # the direct layout differs


@pytest.mark.parametrize("case", [(2, 32, 32, 128, 0, 32), (1, 64, 32, 128, 128, 32), (3, 48, 80, 64, 64, 32), (2, 128, 128, 128, 0, 32), (5, 32, 16, 256, 128, 32),
                                  (8, 128, 64, 128, 0, 64)])      # (a launch the direct plan would split over K keeps the direct form: one chunk, or 512 tiles)
def test_groupnorm_backward_sums_from_the_data_gradient_epilogue(H, knob, case):
    """pdae_conv_gnbwd_arm (round 5): the Winograd-form data gradient leaves sum dv and sum dv (x - mu), dv = dA silu'(a (x - mu) + b), per (image,
    16 x 16 tile, channel) while it stores dA, and pdae_gn_bwd finalizes from them (pdae_gn_bwd_parts_arm) instead of running its reduction pass
    over (x, dA) (module.py:241,257: autograd of GroupNorm -> SiLU in front of a 3x3 convolution).  Checked: the partial sums against fp64 on the
    SAME dA, and the whole GroupNorm backward (dx, dgamma, dbeta) against the reduction-pass form on the same tensors."""
    N, Hh, W, C0, C1, Cout = case
    C, G = C0 + C1, 32
    knob("PDAE_W1", WMODE)
    x = rn(1, N, C, Hh, W) * 1.5 + 0.7
    gamma, beta = 1 + 0.2 * rn(2, C), 0.2 * rn(3, C) + 0.3
    w = rn(4, Cout, C, 3, 3, scale=1.0 / math.sqrt(9 * C))
    dy = rn(5, N, Cout, Hh, W) * 2e-3
    c = H.Conv(N, Hh, W, C0, C1, Cout, k=3, math=4)
    nb, tiles = H.conv_gnbwd_bytes(c, f16_grad=True)
    th = 16 if WMODE == "2" else 8
    if (Hh // th) * (W // 16) > 64:                      # more tiles per image than the finalize kernel's workspace holds: the reduction pass runs
        assert nb == 0
        return
    assert nb == N * tiles * C * 2 * 4 and tiles == (Hh // th) * (W // 16)
    xh = nhwc(x).cuda()
    x0 = xh[..., :C0].contiguous()
    x1 = xh[..., C0:].contiguous() if C1 else None
    mean = torch.empty(N * G, device="cuda"); rstd = torch.empty(N * G, device="cuda")
    wsg = torch.empty(H.gn_ws_bytes(N, C) // 4 + 64, device="cuda")
    H.run(H.op_gn_stats(x0, C0, x1, C1, N, Hh * W, G, 1e-5, mean, rstd, wsg))
    coef = torch.empty(3, N, C, device="cuda")
    H.run(H.op_gn_coef(N, C, G, mean, rstd, gamma.cuda(), beta.cuda(), None, None, coef))
    wd, dyd = nhwc(w).cuda(), nhwc(dy).cuda()
    amax = torch.empty(4, device="cuda")
    H.run(H.op_amax(dyd, dyd.numel(), amax))
    wp_t = torch.empty(c.wprep_bytes(1, force=True, f16_grad=True) // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 1 | 16, wp_t))
    dA = torch.full((N, Hh, W, C), float("nan"), device="cuda")
    part = torch.full((nb // 4,), float("nan"), device="cuda")
    H.run(H.op_conv_dgrad(c, dyd, wd, dA, wp_t=wp_t, dy_amax=amax, gnb=(x0, C0, x1, C1, coef, part)))
    dA2 = torch.empty_like(dA)
    H.run(H.op_conv_dgrad(c, dyd, wd, dA2, wp_t=wp_t, dy_amax=amax))
    assert torch.equal(dA, dA2)                                           # the extra epilogue work does not touch the values stored
    # the sums in fp64 from the stored dA
    cf = coef.double().cpu()                                               # [3][N][C]
    xd, dAd = xh.double().cpu(), dA.double().cpu()                         # [N][H][W][C]
    xm = xd - cf[0].view(N, 1, 1, C)
    z = cf[1].view(N, 1, 1, C) * xm + cf[2].view(N, 1, 1, C)
    sg = torch.sigmoid(z)
    dv = dAd * sg * (1 + z * (1 - sg))
    s0 = dv.view(N, Hh // th, th, W // 16, 16, C).sum((2, 4)).reshape(N, tiles, C)
    s1 = (dv * xm).view(N, Hh // th, th, W // 16, 16, C).sum((2, 4)).reshape(N, tiles, C)
    p = part.view(N, tiles, C, 2).double().cpu()
    scale0, scale1 = float(dv.abs().sum((1, 2)).max()) / tiles, float((dv * xm).abs().sum((1, 2)).max()) / tiles     # cancellation-free magnitudes
    assert float((p[..., 0] - s0).abs().max()) < 2e-6 * scale0 and float((p[..., 1] - s1).abs().max()) < 2e-6 * scale1
    # whole GroupNorm backward: finalize + apply from the epilogue's sums vs the reduction-pass form
    outs = []
    for use_parts in (False, True):
        dx0 = torch.empty(N, Hh, W, C0, device="cuda"); dx1 = torch.empty(N, Hh, W, C1, device="cuda") if C1 else None
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        H.run(H.op_gn_bwd(x0, C0, x1, C1, N, Hh, W, G, coef, rstd, gamma.cuda(), beta.cuda(), None, None, dA, 1, 0, wsg, dx0=dx0, dx1=dx1, dgamma=dg, dbeta=db,
                          parts=part if use_parts else None, parts_tiles=tiles))
        outs.append((dx0, dx1, dg, db))
    for a, b in zip(outs[0], outs[1]):
        if a is not None:
            assert rel_err(b, a) < 2e-5
    # one-shot: nothing stays armed
    H.run(H.op_conv_dgrad(c, dyd, wd, dA2, wp_t=wp_t, dy_amax=amax))
    assert torch.equal(dA, dA2)


def test_groupnorm_backward_sums_are_refused_where_not_built(H, knob):
    knob("PDAE_W1", WMODE)
    assert H.conv_gnbwd_bytes(H.Conv(2, 32, 32, 128, 0, 32, k=3, math=3))[0] == 0                   # bf16x6: the direct kernels
    assert H.conv_gnbwd_bytes(H.Conv(2, 32, 32, 128, 0, 32, k=3, math=4), f16_grad=False)[0] == 0   # without dy_amax the data gradient runs bf16x6
    assert H.conv_gnbwd_bytes(H.Conv(2, 32, 32, 96, 32, 32, k=3, math=4), f16_grad=True)[0] > 0
    assert H.conv_gnbwd_bytes(H.Conv(2, 32, 32, 112, 16, 32, k=3, math=4), f16_grad=True)[0] == 0   # sources must be whole 32-channel runs
    assert H.conv_gnbwd_bytes(H.Conv(2, 32, 32, 64, 0, 32, k=3, math=4), f16_grad=True)[0] == 0     # GEMM N = 64: not the Winograd form
    assert H.conv_gnbwd_bytes(H.Conv(1, 144, 144, 128, 0, 32, k=3, math=4), f16_grad=True)[0] == 0  # 81 tiles per image > 64
    c = H.Conv(2, 32, 32, 64, 0, 64, k=3, math=4)
    x = torch.zeros(2, 32, 32, 64, device="cuda"); coef = torch.zeros(3, 2, 64, device="cuda"); w = torch.zeros(64, 3, 3, 64, device="cuda")
    wp_t = torch.empty(c.wprep_bytes(1, force=True) // 4, device="cuda")
    H.run(H.op_conv_wprep(c, w, 1, wp_t))
    with pytest.raises(H.PdaeError):
        H.run(H.op_conv_dgrad(c, x, w, x.clone(), wp_t=wp_t, gnb=(x, 64, None, 0, coef, torch.zeros(4096, device="cuda"))))


@pytest.mark.parametrize("form", ["fwd_residual", "fwd_halfres_residual", "dgrad_accumulate", "dgrad_gnb"])
def test_store_hazard_stress_repeated_launches_are_bit_identical(H, knob, form):
    """the epilogue instantiations with an operand (EX / GB) -- where the output-store data hazard corrupted ~1.9 % of the outputs of the
    8-row form, varying from run to run -- launched 24 times back to back on a shape that keeps every CU's vector-memory queue
    full (several tiles per workgroup, operand loads in front of every block's stores), beside a streaming kernel on a second stream that
    adds pressure on the same path: every run bit-identical to the first, the first within the gate of the direct form.  Runs under both tile
    heights (the file's autouse fixture)."""
    knob("PDAE_W1", WMODE)
    N, Hh, W, C, Cout = 12, 64, 64, 128, 128
    x = rn(1, N, C, Hh, W) * 1.3 + 0.2
    w = rn(2, Cout, C, 3, 3, scale=1.0 / math.sqrt(9 * C)); b = rn(3, Cout, scale=0.2)
    c = H.Conv(N, Hh, W, C, 0, Cout, k=3, math=4)
    xd, wd, bd = nhwc(x).cuda(), nhwc(w).cuda(), b.cuda()
    dy = rn(5, N, Cout, Hh, W) * 2e-3
    dyd = nhwc(dy).cuda()
    amax = torch.empty(4, device="cuda")
    H.run(H.op_amax(dyd, dyd.numel(), amax))
    res = nhwc(rn(4, N, Cout, Hh, W)).cuda()
    res_half = nhwc(rn(4, N, Cout, Hh // 2, W // 2)).cuda()
    base = rn(6, N, Hh, W, C).cuda()
    wp = torch.empty(c.wprep_bytes(0, force=True) // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 0, wp))
    wp_t = torch.empty(c.wprep_bytes(1, force=True, f16_grad=True) // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 1 | 16, wp_t))
    coef = part = None
    if form == "dgrad_gnb":
        G = 32
        mean = torch.empty(N * G, device="cuda"); rstd = torch.empty(N * G, device="cuda")
        wsg = torch.empty(H.gn_ws_bytes(N, C) // 4 + 64, device="cuda")
        H.run(H.op_gn_stats(xd, C, None, 0, N, Hh * W, G, 1e-5, mean, rstd, wsg))
        coef = torch.empty(3, N, C, device="cuda")
        H.run(H.op_gn_coef(N, C, G, mean, rstd, (1 + 0.2 * rn(7, C)).cuda(), (0.2 * rn(8, C)).cuda(), None, None, coef))
        nb, _ = H.conv_gnbwd_bytes(c, f16_grad=True)
        assert nb > 0
        part = torch.empty(nb // 4, device="cuda")

    def launch():
        if form == "fwd_residual":
            y = torch.full((N, Hh, W, Cout), float("nan"), device="cuda")
            H.run(H.op_conv_fwd(c, xd, None, wd, bd, y, res=res, res_mode=1, wp=wp))
            return [y]
        if form == "fwd_halfres_residual":
            y = torch.full((N, Hh, W, Cout), float("nan"), device="cuda")
            H.run(H.op_conv_fwd(c, xd, None, wd, bd, y, res=res_half, res_mode=2, wp=wp))
            return [y]
        if form == "dgrad_accumulate":
            dx = base.clone()
            H.run(H.op_conv_dgrad(c, dyd, wd, dx, accumulate=1, wp_t=wp_t, dy_amax=amax))
            return [dx]
        dx = torch.full((N, Hh, W, C), float("nan"), device="cuda")
        part.fill_(float("nan"))
        H.run(H.op_conv_dgrad(c, dyd, wd, dx, wp_t=wp_t, dy_amax=amax, gnb=(xd, C, None, 0, coef, part)))
        return [dx, part.clone()]
    first = launch()
    torch.cuda.synchronize()
    assert all(torch.isfinite(t).all() for t in first)
    # pressure: a copy loop on another stream while the launches repeat
    side = torch.cuda.Stream()
    big_a, big_b = torch.empty(1 << 26, device="cuda"), torch.empty(1 << 26, device="cuda")
    for it in range(24):
        with torch.cuda.stream(side):
            big_b.copy_(big_a)
        got = launch()
        for a, b_ in zip(first, got):
            assert torch.equal(a, b_), (form, it, int((a != b_).sum()))
    torch.cuda.synchronize()
    # ... and the first run agrees with the direct form on the same operands
    knob("PDAE_W1", "0")
    if form.startswith("fwd"):
        wp0 = torch.empty(c.wprep_bytes(0, force=True) // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 0, wp0))
        y0 = torch.empty_like(first[0])
        H.run(H.op_conv_fwd(c, xd, None, wd, bd, y0, res=res if form == "fwd_residual" else res_half, res_mode=1 if form == "fwd_residual" else 2, wp=wp0))
        assert rel_err(first[0], y0) < 3 * TOL[4]
    else:
        wp0 = torch.empty(c.wprep_bytes(1, force=True, f16_grad=True) // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 1 | 16, wp0))
        dx0 = base.clone() if form == "dgrad_accumulate" else torch.empty_like(first[0])
        H.run(H.op_conv_dgrad(c, dyd, wd, dx0, accumulate=int(form == "dgrad_accumulate"), wp_t=wp0, dy_amax=amax))
        # (accumulate: both forms round base + gradient to fp32 -- compared on the sums, where one ulp of the O(1) base is 6e-8)
        assert rel_err(first[0], dx0) < (1e-6 if form == "dgrad_accumulate" else 3e-5)
