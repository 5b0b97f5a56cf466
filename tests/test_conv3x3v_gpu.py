"""GPU: conv3x3v.hip -- the producer / consumer form of the 3x3 weight gradient (round 6; one workgroup per CU, four matrix waves fed by four
staging waves) through the C ABI (pdae_conv2d_wgrad), against fp64 autograd of F.conv2d (the reference's path: model/module.py:242,265) and
against conv3x3w.hip (knob PDAE_W3V = 0) on the same inputs.  Shapes are chosen so that conv3x3v_ok holds (whole 64-channel blocks, 16-wide
tiles, >= 64 pixel tiles) and cover: one and several tiles per workgroup (odd counts: both LDS buffers end a loop), ragged last split, the
nearest-upsampled input, accumulate, the riding bias gradient, tiny / large gradient magnitudes in the fp16 format, the bf16 formats, and the
fused GroupNorm (+ SiLU) input on a two-source tensor."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.conftest import rel_err
from tests.test_kernels_gpu import _gn_ref, nhwc, ref_conv, rn, ws

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from pdae_amd import hip
    hip.lib()
    return hip


def _wgrad(H, c, x0, x1, dy, dy_amax, accumulate=0, dw0=None, db0=None, **kw):
    wsb = c.wgrad_ws_bytes()
    C = c.C0 + c.C1
    dw = torch.empty(c.Cout, 3, 3, C, device="cuda") if dw0 is None else dw0.clone()
    db = torch.empty(c.Cout, device="cuda") if db0 is None else db0.clone()
    H.run(H.op_conv_wgrad(c, x0, x1, dy, dw, ws(wsb), wsb, accumulate=accumulate, db=db, dy_amax=dy_amax, **kw))
    return dw, db


# N, H, W, Cin, Cout, up
V_CASES = [
    (4, 32, 64, 64, 64, 0),        # 64 tiles, base 1: 16 splits x 4 tiles
    (5, 24, 48, 64, 128, 0),       # 45 tiles < 64: NOT this form (conv3x3w) -- the routing must still be right
    (9, 24, 48, 128, 64, 0),       # 81 tiles: ragged last split, odd tiles per split
    (2, 64, 64, 192, 128, 0),      # three input chunks, two output tiles
    (8, 16, 32, 64, 64, 1),        # nearest-upsampled input (stored 8 x 16)
    (32, 16, 16, 256, 64, 0),      # one tile column per image, four input chunks
]


@pytest.mark.parametrize("math_mode", [4, 2, 1])
@pytest.mark.parametrize("case", V_CASES)
def test_weight_gradient_producer_consumer_form(H, knob, case, math_mode):
    N, Hh, W, Cin, Cout, up = case
    Hs, Wsd = (Hh // 2, W // 2) if up else (Hh, W)
    x = rn(1, N, Cin, Hs, Wsd)
    dy = rn(5, N, Cout, Hh, W) * 2e-3
    w = rn(2, Cout, Cin, 3, 3, scale=1.0 / math.sqrt(Cin * 9))
    wr = w.double().requires_grad_(True)
    (ref_conv(x.double(), wr, None, 1, 1, up) * dy.double()).sum().backward()
    c = H.Conv(N, Hs, Wsd, Cin, 0, Cout, k=3, up=bool(up), math=math_mode)
    xd, dyd = nhwc(x).cuda(), nhwc(dy).cuda()
    am = torch.empty(4, device="cuda")
    H.run(H.op_amax(dyd, dyd.numel(), am))
    amx = am if math_mode == 4 else None
    knob("PDAE_W3V", 1)
    dw, db = _wgrad(H, c, xd, None, dyd, amx)
    tol = {1: 2e-2, 2: 3e-4, 4: 2e-5}[math_mode]
    assert rel_err(dw.permute(0, 3, 1, 2), wr.grad) < tol
    assert rel_err(db, dy.double().sum((0, 2, 3))) < 1e-5
    knob("PDAE_W3V", 0)
    dw_w, db_w = _wgrad(H, c, xd, None, dyd, amx)
    assert rel_err(dw, dw_w) < (2e-6 if math_mode == 4 else tol)
    assert rel_err(db, db_w) < 1e-6
    # accumulate: dw += ..., db += ...
    knob("PDAE_W3V", 1)
    dw0, db0 = rn(7, Cout, 3, 3, Cin).cuda() * 0.1, rn(8, Cout).cuda()
    dw2, db2 = _wgrad(H, c, xd, None, dyd, amx, accumulate=1, dw0=dw0, db0=db0)
    assert rel_err(dw2 - dw0, dw) < 1e-5 and rel_err(db2 - db0, db) < 1e-5
    # the same launch twice: bit-identical (fixed-order slabs, no atomics)
    dw3, db3 = _wgrad(H, c, xd, None, dyd, amx)
    assert torch.equal(dw3, dw) and torch.equal(db3, db)


@pytest.mark.parametrize("gscale", [3e-7, 2e4])
def test_fp16_format_dynamic_scale(H, knob, gscale):
    N, Hh, W, Cin, Cout = 4, 32, 64, 128, 128
    x = rn(1, N, Cin, Hh, W)
    dy = rn(5, N, Cout, Hh, W) * gscale
    wr = rn(2, Cout, Cin, 3, 3, scale=0.03).double().requires_grad_(True)
    (F.conv2d(x.double(), wr, None, padding=1) * dy.double()).sum().backward()
    c = H.Conv(N, Hh, W, Cin, 0, Cout, k=3, math=4)
    xd, dyd = nhwc(x).cuda(), nhwc(dy).cuda()
    am = torch.empty(4, device="cuda")
    H.run(H.op_amax(dyd, dyd.numel(), am))
    knob("PDAE_W3V", 1)
    dw, db = _wgrad(H, c, xd, None, dyd, am)
    assert rel_err(dw.permute(0, 3, 1, 2), wr.grad) < 2e-5
    assert rel_err(db, dy.double().sum((0, 2, 3))) < 1e-5


# N, H, W, C0, C1, Cout, up, act
GN_CASES = [(8, 32, 32, 128, 128, 128, 0, 1), (4, 32, 64, 64, 0, 64, 0, 1), (8, 32, 32, 96, 32, 64, 0, 0), (8, 16, 32, 64, 64, 128, 1, 1)]


@pytest.mark.parametrize("math_mode", [4, 1])
@pytest.mark.parametrize("case", GN_CASES)
def test_weight_gradient_with_fused_groupnorm_input_producer_consumer_form(H, knob, case, math_mode):
    """pdae_conv_gn_input_arm on the producer / consumer form: the staging waves recompute act(GroupNorm(x)) on the RAW two-source input
    (module.py:241-242, 279-284) -- vs fp64 autograd, vs the materialised activation on the same kernel, vs conv3x3w's GN instantiation."""
    N, Hh, W, C0, C1, Cout, up, act = case
    C, G = C0 + C1, 32
    x = rn(1, N, C, Hh, W) * 1.5 + 0.7
    gamma, beta = 1 + 0.2 * rn(2, C), 0.2 * rn(3, C) + 0.5
    c = H.Conv(N, Hh, W, C0, C1, Cout, k=3, up=bool(up), math=math_mode)
    dy = rn(5, N, Cout, c.Ho, c.Wo) * 3e-3
    w = rn(6, Cout, C, 3, 3, scale=1.0 / math.sqrt(C * 9))
    a_ref = _gn_ref(x.double(), gamma.double(), beta.double(), None, None, act)
    wr = w.double().requires_grad_(True)
    (ref_conv(a_ref, wr, None, 1, 1, up) * dy.double()).sum().backward()
    assert H.conv_wgrad_gn_ok(c)
    xh = nhwc(x).cuda()
    x0 = xh[..., :C0].contiguous()
    x1 = xh[..., C0:].contiguous() if C1 else None
    mean = torch.empty(N * G, device="cuda"); rstd = torch.empty(N * G, device="cuda")
    H.run(H.op_gn_stats(x0, C0, x1, C1, N, Hh * W, G, 1e-5, mean, rstd, ws(H.gn_ws_bytes(N, C))))
    coef = torch.empty(3, N, C, device="cuda")
    H.run(H.op_gn_coef(N, C, G, mean, rstd, gamma.cuda(), beta.cuda(), None, None, coef))
    dyd = nhwc(dy).cuda()
    am = torch.empty(4, device="cuda")
    H.run(H.op_amax(dyd, dyd.numel(), am))
    amx = am if math_mode == 4 else None
    knob("PDAE_W3V", 1)
    dw, db = _wgrad(H, c, x0, x1, dyd, amx, gn_coef=coef, gn_act=act)
    tol = {1: 2e-2, 4: 2e-5}[math_mode]
    assert rel_err(dw.permute(0, 3, 1, 2), wr.grad) < tol
    assert rel_err(db, dy.double().sum((0, 2, 3))) < 1e-5
    a = torch.empty(N, Hh, W, C, device="cuda")
    H.run(H.op_gn_apply(x0, C0, x1, C1, N, Hh, W, coef, act, 0, a))
    c1 = H.Conv(N, Hh, W, C, 0, Cout, k=3, up=bool(up), math=math_mode)
    dw2, _ = _wgrad(H, c1, a, None, dyd, amx)
    assert rel_err(dw, dw2) < (1e-6 if math_mode != 1 else 1e-2)
    knob("PDAE_W3V", 0)
    dw_w, _ = _wgrad(H, c, x0, x1, dyd, amx, gn_coef=coef, gn_act=act)
    assert rel_err(dw, dw_w) < (2e-6 if math_mode == 4 else tol)


@pytest.mark.parametrize("shape", [(32, 128, 128, 128, 128), (32, 64, 64, 256, 256), (32, 16, 16, 384, 384)])
def test_full_size_properties_at_the_benchmark_batch(H, knob, shape):
    """BASELINE sizes (B = 32, the step's own layers), where an fp64 reference would take minutes of host time: size-independent properties instead.
    (a) the two kernels agree on the same operands (conv3x3w is pinned against fp64 at small sizes: 2e-6 of max |dW|); (b) linearity in dY --
    dW(dY1 + dY2) = dW(dY1) + dW(dY2), each launch with its own dynamic fp16 scale; (c) a dY that is zero outside one image gives the dW of that image
    alone (the pixel-tile partition over splits cannot leak between images); (d) bit-identical repeat."""
    N, Hh, W, Cin, Cout = shape
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(N, Hh, W, Cin, device="cuda", generator=g)
    dy1 = torch.randn(N, Hh, W, Cout, device="cuda", generator=g) * 1e-3
    dy2 = torch.randn(N, Hh, W, Cout, device="cuda", generator=g) * 3e-4
    c = H.Conv(N, Hh, W, Cin, 0, Cout, k=3, math=4)

    def run(dy, v):
        knob("PDAE_W3V", v)
        am = torch.empty(4, device="cuda")
        H.run(H.op_amax(dy, dy.numel(), am))
        return _wgrad(H, c, x, None, dy, am)
    dw1, db1 = run(dy1, 1)
    dw1w, db1w = run(dy1, 0)
    assert rel_err(dw1, dw1w) < 2e-6 and rel_err(db1, db1w) < 1e-6
    dw2, _ = run(dy2, 1)
    dw12, _ = run(dy1 + dy2, 1)
    assert rel_err(dw12, dw1 + dw2) < 5e-6
    one = torch.zeros_like(dy1)
    one[N // 2] = dy1[N // 2]
    dwo, _ = run(one, 1)
    c1 = H.Conv(1, Hh, W, Cin, 0, Cout, k=3, math=4)
    if Hh * W >= 64 * 128:                       # one image alone is still a launch of >= 64 pixel tiles: the same kernel on the single image
        knob("PDAE_W3V", 1)
        am = torch.empty(4, device="cuda")
        H.run(H.op_amax(one, one.numel(), am))
        dws, _ = _wgrad(H, c1, x[N // 2:N // 2 + 1].contiguous(), None, dy1[N // 2:N // 2 + 1].contiguous(), am)
        assert rel_err(dwo, dws) < 2e-6
    dw1b, db1b = run(dy1, 1)
    assert torch.equal(dw1, dw1b) and torch.equal(db1, db1b)
