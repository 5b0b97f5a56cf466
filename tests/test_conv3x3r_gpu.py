"""GPU: the persistent 3x3 patch kernel csrc/conv3x3r.hip (one wave per SIMD, 128-pixel x 64-channel wave tiles, epilogue of tile i deferred
into the main loop of tile i+1) against fp64 references AND against the two-waves-per-SIMD kernel it replaces on large layers
(csrc/conv3x3p.hip).  Both accumulate every output element in the same order (chunks, taps, k-halves, products; same epilogue expression),
so their results must be BIT-IDENTICAL: PDAE_P3R = 0 keeps a launch on conv3x3p, = 2 routes it to conv3x3r regardless of the fill
heuristic (read per launch).
Covers: plain forward with bias / residual (same- and half-resolution) / nearest-upsampled input, fused GroupNorm input (one and two
sources), fused 1x1 skip chunks (plain and GroupNorm main input, one and two skip sources), output statistics, the data gradient with the
dynamic fp16 scale, bf16 operands (math 1) and the three-product bf16 split (math 2), multi-tile / multi-image / several 128-channel tiles."""
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


@pytest.fixture(autouse=True)
def _direct_form(knob):
    """These tests are about the two DIRECT kernels: the Winograd-along-x form (conv3x3y.hip, the default on chip-filling layers) is switched off."""
    knob("PDAE_W1", "0")


def rn(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).double().cpu()


KERNELS = ["r"]


def _both(knob, run, kernel):
    """run() under conv3x3p (PDAE_P3R=0) and under conv3x3r (PDAE_P3R=2): returns the two results."""
    out = []
    for on in (False, True):
        knob("PDAE_P3R", "2" if on else "0")
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


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("math_mode", [4, 1, 2])
@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout, up, res_mode
    (2, 32, 16, 32, 128, 0, 0),            # one tile per image, one chunk
    (1, 64, 48, 96, 256, 0, 1),            # 2 x 3 tiles, 3 chunks, two 128-channel tiles, same-resolution residual
    (3, 32, 32, 64, 128, 1, 2),            # nearest-upsampled input (stored 16 x 16) + half-resolution residual
    (2, 96, 32, 128, 128, 0, 0),           # three tile rows: interior rows see no zero padding at top / bottom
    (9, 96, 96, 32, 128, 0, 1),            # 324 tiles of 16 x 16: the persistent workgroups walk two tiles each (deferred epilogue, next-tile prefetch)
    (5, 80, 112, 64, 256, 0, 0),           # 350 tiles x 2 channel tiles = 700: three tiles per workgroup, ragged last round
])
def test_forward_bit_identical_to_conv3x3p_and_close_to_fp64(H, knob, case, math_mode, kernel):
    N, Hh, W, C, Cout, up, res_mode = case
    if N * Hh * W > 50000 and math_mode != 4:
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
    wp = torch.empty(c.wprep_bytes(0, force=True) // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 0, wp))

    def run():
        y = torch.full((N, Hh, W, Cout), float("nan"), device="cuda")
        H.run(H.op_conv_fwd(c, xd, None, wd, bd, y, res=resd, res_mode=res_mode, wp=wp))
        return y
    y_p, y_q = _both(knob, run, kernel)
    assert rel_err(nchw(y_q), y_ref) < TOL[math_mode]
    assert torch.equal(y_p, y_q), float((y_p - y_q).abs().max())


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("case", [
    # N, H, W, C0, C1, Cout, up, res_mode, AdaGN
    (2, 32, 32, 64, 0, 128, 0, 1, True),
    (2, 64, 16, 64, 32, 128, 0, 0, False),     # two-source concat
    (1, 32, 32, 32, 0, 256, 1, 2, True),       # upsampled input, half-resolution residual
    (6, 96, 128, 32, 32, 128, 0, 1, True),     # 288 tiles: two per persistent workgroup, coefficients of the NEXT image prefetched across the tile boundary
])
def test_fused_groupnorm_input_bit_identical_and_close_to_fp64(H, knob, case, kernel):
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
    wp = torch.empty(c.wprep_bytes(0, force=True, gn=True) // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 4, wp))
    resd = nhwc(res).cuda() if res is not None else None

    def run():
        y = torch.full((N, Hh, W, Cout), float("nan"), device="cuda")
        H.run(H.op_conv_fwd_gn(c, x0, x1, coef, 1, wp, bd, y, res=resd, res_mode=res_mode))
        return y
    y_p, y_q = _both(knob, run, kernel)
    assert rel_err(nchw(y_q), y_ref) < 1e-5
    assert torch.equal(y_p, y_q), float((y_p - y_q).abs().max())


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("case", [(16, 64, 32, 64, 64, 32, 128, True), (16, 64, 32, 32, 96, 0, 128, False), (8, 64, 64, 96, 32, 32, 256, False)])
def test_fused_skip_chunks_and_output_statistics(H, knob, case, kernel):
    """conv3x3(x) + conv1x1([s0 | s1]) in one K loop (centre-tap chunks behind the main chunks), and the GroupNorm partial statistics of the
    output written by the epilogue: same tensor, and the same statistics after the reader's fp64 combine, as conv3x3p."""
    N, Hh, W, C, Cs0, Cs1, Cout, use_gn = case
    Cs, G = Cs0 + Cs1, 32
    x = rn(1, N, C, Hh, W) * 1.2 + 0.3
    sx = rn(2, N, Cs, Hh, W) * 2.0
    w = rn(3, Cout, C, 3, 3, scale=1.0 / math.sqrt(C * 9)); b = rn(4, Cout, scale=0.1)
    wsk = rn(5, Cout, Cs, 1, 1, scale=1.0 / math.sqrt(Cs)); bsk = rn(6, Cout, scale=0.1)
    gamma, beta = 1 + 0.2 * rn(7, C), 0.2 * rn(8, C) + 0.4
    c = H.Conv(N, Hh, W, C, 0, Cout, k=3, math=4)
    cs = H.Conv(N, Hh, W, Cs0, Cs1, Cout, k=1, math=4)
    if not H.conv_fwd_skip_ok(c, cs):
        pytest.skip("pair not eligible for the fused launch at this size")
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
    nbytes, tpi = H.conv_stats_bytes(c, cs)
    assert nbytes > 0 and tpi == (Hh // 8) * (W // 16)
    g2, b2 = (1 + 0.1 * rn(9, Cout)).cuda(), (0.1 * rn(10, Cout)).cuda()

    def run():
        y = torch.full((N, Hh, W, Cout), float("nan"), device="cuda")
        part = torch.full((nbytes // 4,), float("nan"), device="cuda")
        H.run(H.op_conv_fwd_skip(c, xd, None, coef, 1, wp, b.cuda(), cs, s0, s1, wps, bsk.cuda(), y, stats=part))
        m, r, k = torch.empty(N * G, device="cuda"), torch.empty(N * G, device="cuda"), torch.empty(3, N, Cout, device="cuda")
        H.run(H.op_gn_coef_from_conv_stats(N, Hh * W, Cout, 0, G, 1e-5, part, tpi, None, 0, g2, b2, None, None, m, r, k))
        return y, part, m, r
    (y_p, part_p, m_p, r_p), (y_q, part_q, m_q, r_q) = _both(knob, run, kernel)
    assert rel_err(nchw(y_q), y_ref) < 1e-5
    assert torch.equal(y_p, y_q)
    assert torch.isfinite(part_q).all()
    assert torch.equal(part_p, part_q)              # same 8 x 16 bands, same slots, same in-lane summation order
    assert (m_p - m_q).abs().max() < 2e-6 * max(1.0, float(m_p.abs().max())) and rel_err(r_q, r_p) < 5e-6
    yd = y_q.double()
    mean_ref = yd.view(N, Hh * W, G, Cout // G).mean((1, 3)).flatten()
    assert (m_q.double() - mean_ref).abs().max() < 5e-6 * max(1.0, float(mean_ref.abs().max()))


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("gscale", [1.0, 3e-7, 2e4])
@pytest.mark.parametrize("case", [(2, 32, 32, 128, 64), (1, 64, 32, 256, 128), (3, 160, 160, 128, 32)])
def test_data_gradient_with_dynamic_fp16_scale(H, knob, case, gscale, kernel):
    """dX of a 3x3 convolution = the same kernel on transposed, tap-flipped prepared weights with the per-tensor power-of-two dY scale
    (pdae_amax): Cin of the convolution is the GEMM N here, so it must be a multiple of 128."""
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
    wp_t = torch.empty(c.wprep_bytes(1, force=True, f16_grad=True) // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 1 | 16, wp_t))

    def run():
        dx = torch.full((N, Hh, W, Cin), float("nan"), device="cuda")
        H.run(H.op_conv_dgrad(c, dyd, wd, dx, wp_t=wp_t, dy_amax=amax))
        return dx
    dx_p, dx_q = _both(knob, run, kernel)
    assert rel_err(nchw(dx_q), xr.grad) < 1e-5
    assert torch.equal(dx_p, dx_q)


@pytest.mark.parametrize("kernel", KERNELS)
def test_accumulating_data_gradient(H, knob, kernel):
    """accumulate = 1 (a second consumer's gradient joins the buffer): read-modify-write epilogue."""
    N, Hh, W, Cin, Cout = 2, 32, 16, 128, 32
    w = rn(2, Cout, Cin, 3, 3, scale=1.0 / math.sqrt(9 * Cin))
    dy = rn(3, N, Cout, Hh, W)
    base = rn(4, N, Hh, W, Cin)
    c = H.Conv(N, Hh, W, Cin, 0, Cout, k=3, math=4)
    wd, dyd = nhwc(w).cuda(), nhwc(dy).cuda()
    wp_t = torch.empty(c.wprep_bytes(1, force=True) // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 1, wp_t))

    def run():
        dx = base.clone().cuda()
        H.run(H.op_conv_dgrad(c, dyd, wd, dx, accumulate=1, wp_t=wp_t))
        return dx
    dx_p, dx_q = _both(knob, run, kernel)
    assert torch.equal(dx_p, dx_q)
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    assert rel_err(nchw(dx_q) - base.permute(0, 3, 1, 2).double(), ref) < 1e-5


def test_large_layers_with_default_routing(H, knob):
    """No override: a benchmark-sized layer (B=32, 64 x 64, 128 output channels = 512 tiles of 16 x 16) takes the default route (conv3x3r) and
    gives the tensor and the partial statistics conv3x3p gives."""
    knob("PDAE_P3R", 1)
    N, Hh, W, C, Cout = 32, 64, 64, 32, 128
    x = rn(1, N, C, Hh, W)
    w = rn(2, Cout, C, 3, 3, scale=1.0 / math.sqrt(9 * C)); b = rn(3, Cout, scale=0.2)
    c = H.Conv(N, Hh, W, C, 0, Cout, k=3, math=4)
    xd, wd, bd = nhwc(x).cuda(), nhwc(w).cuda(), b.cuda()
    wp = torch.empty(c.wprep_bytes(0) // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 0, wp))
    nbytes, tpi = H.conv_stats_bytes(c, None)
    outs = []
    for mode in (None, "0"):
        if mode is not None:
            knob("PDAE_P3R", mode)
        y = torch.empty(N, Hh, W, Cout, device="cuda"); part = torch.zeros(nbytes // 4, device="cuda")
        H.run(H.op_conv_fwd(c, xd, None, wd, bd, y, wp=wp, stats=part))
        outs.append((y, part.view(N, tpi, Cout // 4, 2)))
    (y_d, p_d), (y_p, p_p) = outs
    assert torch.equal(y_d, y_p)
    assert rel_err(p_d.sum(1), p_p.sum(1)) < 1e-5
    assert rel_err(nchw(y_d), F.conv2d(x.double(), w.double(), b.double(), padding=1)) < 1e-5
