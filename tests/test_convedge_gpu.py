"""convedge.hip: the 3-channel edge layers (image head forward / weight gradient / data gradient, stem forward) on the matrix cores, against an
fp64 convolution and against the fp32 FMA / generic kernels they replace (PDAE_EDGE=0 is read per call).  Shapes cover ragged tiles, every
supported channel count, more tiles than persistent workgroups (> 512) and the accumulate forms."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pdae_amd import hip
    return hip


def rn(seed, *shape, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / b.double().norm())


def both(fn):
    from pdae_amd import hip
    out = []
    for sw in (1, 0):
        hip.set_knob("PDAE_EDGE", sw)
        try:
            out.append(fn())
        finally:
            hip.set_knob("PDAE_EDGE", 1)
    return out


HEAD_CASES = [(2, 24, 40, 128, 3), (1, 16, 16, 32, 1), (3, 32, 48, 64, 2), (9, 128, 128, 128, 3), (5, 20, 12, 32, 3), (2, 64, 64, 64, 3)]


@pytest.mark.parametrize("case", HEAD_CASES)
def test_head_forward_wgrad_dgrad(H, case):
    N, Hh, W, C, Cout = case
    x = rn(1, N, C, Hh, W); w = rn(2, Cout, C, 3, 3, scale=1 / math.sqrt(9 * C)); b = rn(3, Cout, scale=0.3); dy = rn(4, N, Cout, Hh, W)
    c = H.Conv(N, Hh, W, C, 0, Cout, k=3)
    xr = x.double().cuda().requires_grad_(True); wr = w.double().cuda().requires_grad_(True)
    yr = F.conv2d(xr, wr, b.double().cuda(), padding=1)
    yr.backward(dy.double().cuda())
    xd, wd, bd, dyd = nhwc(x).cuda(), nhwc(w).cuda(), b.cuda(), nhwc(dy).cuda()

    def fwd():
        y = torch.full((N, Hh, W, Cout), float("nan"), device="cuda")
        H.run(H.op_conv_fwd(c, xd, None, wd, bd, y))
        return y.permute(0, 3, 1, 2)
    y1, y0 = both(fwd)
    assert rel(y1, yr.detach()) < 2e-6 and rel(y1, y0) < 2e-6

    def dgrad():
        dx = torch.full((N, Hh, W, C), float("nan"), device="cuda")
        H.run(H.op_conv_dgrad(c, dyd, wd, dx))
        dxa = torch.ones(N, Hh, W, C, device="cuda")
        H.run(H.op_conv_dgrad(c, dyd, wd, dxa, accumulate=1))
        return dx.permute(0, 3, 1, 2), dxa.permute(0, 3, 1, 2) - 1.0
    (d1, a1), (d0, _) = both(dgrad)
    assert rel(d1, xr.grad) < 2e-6 and rel(a1, xr.grad) < 2e-6 and rel(d1, d0) < 2e-6

    def wgrad():
        wsb = c.wgrad_ws_bytes(); wsp = torch.empty(wsb // 4 + 16, device="cuda")
        dw = torch.full_like(wd, float("nan")); db = torch.empty(Cout, device="cuda")
        H.run(H.op_conv_wgrad(c, xd, None, dyd, dw, wsp, wsb, db=db))
        H.run(H.op_conv_wgrad(c, xd, None, dyd, dw, wsp, wsb, accumulate=1, db=db))
        return dw.permute(0, 3, 1, 2) / 2, db / 2
    (w1, b1), (w0, _) = both(wgrad)
    assert rel(w1, wr.grad) < 5e-6 and rel(w1, w0) < 5e-6 and rel(b1, dy.double().sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("case", [(2, 24, 40, 3, 128), (1, 16, 16, 1, 32), (9, 128, 128, 3, 128), (3, 32, 32, 3, 64), (2, 20, 36, 2, 96), (2, 16, 16, 3, 256)])
def test_stem_forward(H, case):
    N, Hh, W, Cin, Cout = case
    x = rn(1, N, Cin, Hh, W); w = rn(2, Cout, Cin, 3, 3, scale=1 / math.sqrt(9 * Cin)); b = rn(3, Cout, scale=0.3)
    c = H.Conv(N, Hh, W, Cin, 0, Cout, k=3)
    yr = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xd, wd, bd = nhwc(x).cuda(), nhwc(w).cuda(), b.cuda()

    def fwd():
        y = torch.full((N, Hh, W, Cout), float("nan"), device="cuda")
        H.run(H.op_conv_fwd(c, xd, None, wd, bd, y))
        return y.permute(0, 3, 1, 2)
    y1, y0 = both(fwd)
    assert rel(y1, yr) < 2e-6 and rel(y1, y0) < 2e-6
