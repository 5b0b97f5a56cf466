"""GPU: randomised (seeded) convolution geometries through EVERY kernel route -- generic implicit GEMM, LDS-patch 3x3 (plain, split-K,
8-pixel image pairs, fused GroupNorm input, fused skip connection), register/LDS 1x1, image head, transposing-read weight gradient --
against fp64 references.  The fixed cases of test_kernels_gpu.py pin known corner cases; this sweep guards the routing logic (eligibility
functions, split planning, tile raggedness) against shapes nobody thought of."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

from tests.conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = {0: 1e-5, 1: 2e-2, 3: 1e-5, 4: 1e-5}


def rn(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).double().cpu()


def _cases():
    rng = random.Random(1234)
    out = []
    for k in range(36):
        ksz = rng.choice([3, 3, 3, 1])
        up = rng.choice([0, 0, 1]) if ksz == 3 else 0
        N = rng.choice([1, 2, 3, 5])
        Hh = rng.choice([4, 8, 12, 16, 24, 32]) // (2 if up else 1) or 4
        W = rng.choice([8, 16, 24, 32, 48]) // (2 if up else 1) or 4
        C0 = rng.choice([32, 64, 96, 128])
        C1 = rng.choice([0, 0, 32, 64]) if ksz == 1 else 0
        Cout = rng.choice([3, 32, 36, 64, 96, 128, 160])
        res_mode = rng.choice([0, 0, 1, 2 if (up or ksz == 1) and Hh % 2 == 0 and W % 2 == 0 else 0])
        mode = rng.choice([4, 4, 3, 1, 0])
        out.append((k, N, Hh, W, C0, C1, Cout, ksz, up, res_mode, mode))
    return out


@pytest.mark.parametrize("case", _cases())
def test_random_conv_geometry_all_routes(case):
    from pdae_amd import hip as H
    seed, N, Hh, W, C0, C1, Cout, k, up, res_mode, mode = case
    Cin, tol = C0 + C1, TOL[mode]
    x = rn(seed * 7 + 1, N, Cin, Hh, W)
    w = rn(seed * 7 + 2, Cout, Cin, k, k, scale=1.0 / math.sqrt(Cin * k * k))
    b = rn(seed * 7 + 3, Cout, scale=0.1)
    c = H.Conv(N, Hh, W, C0, C1, Cout, k=k, up=bool(up), math=mode)
    xl = (F.interpolate(x, scale_factor=2, mode="nearest") if up else x).double().clone().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    y_ref = F.conv2d(xl, wr, b.double(), padding=k // 2)
    res = None
    if res_mode == 1:
        res = rn(seed * 7 + 4, N, Cout, c.Ho, c.Wo)
    elif res_mode == 2:
        res = rn(seed * 7 + 4, N, Cout, c.Ho // 2, c.Wo // 2)
    y_full = y_ref.detach() + (0 if res is None else (res.double() if res_mode == 1 else F.interpolate(res, scale_factor=2, mode="nearest").double()))
    xh = nhwc(x).cuda()
    x0 = xh[..., :C0].contiguous()
    x1 = xh[..., C0:].contiguous() if C1 else None
    wd, bd = nhwc(w).cuda(), b.cuda()
    resd = nhwc(res).cuda() if res is not None else None
    y = torch.empty(N, c.Ho, c.Wo, Cout, device="cuda")
    # forward: generic route, then the prepared-weight route when the shape is eligible (forced past the fill heuristic)
    H.run(H.op_conv_fwd(c, x0, x1, wd, bd, y, res=resd, res_mode=res_mode))
    assert rel_err(nchw(y), y_full) < tol
    nb = c.wprep_bytes(0, force=True)
    if nb:
        wp = torch.empty(nb // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 0, wp))
        y.zero_()
        H.run(H.op_conv_fwd(c, x0, x1, wd, bd, y, res=resd, res_mode=res_mode, wp=wp))
        assert rel_err(nchw(y), y_full) < tol
    # backward
    dy = rn(seed * 7 + 5, N, Cout, c.Ho, c.Wo)
    (y_ref * dy.double()).sum().backward()
    dyd = nhwc(dy).cuda()
    dx = torch.empty(N, c.Hl, c.Wl, Cin, device="cuda")
    H.run(H.op_conv_dgrad(c, dyd, wd, dx))
    assert rel_err(nchw(dx), xl.grad) < tol
    nbt = c.wprep_bytes(1, force=True)
    if nbt:
        wp_t = torch.empty(nbt // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 1, wp_t))
        dx.zero_()
        H.run(H.op_conv_dgrad(c, dyd, wd, dx, wp_t=wp_t))
        assert rel_err(nchw(dx), xl.grad) < tol
    wsb = c.wgrad_ws_bytes()
    wsp = torch.empty(wsb // 4 + 16, device="cuda")
    dw, db = torch.empty_like(wd), torch.empty(Cout, device="cuda")
    H.run(H.op_conv_wgrad(c, x0, x1, dyd, dw, wsp, wsb, db=db))
    assert rel_err(dw.permute(0, 3, 1, 2), wr.grad) < 2 * tol
    assert rel_err(db, dy.double().sum((0, 2, 3))) < 1e-5
