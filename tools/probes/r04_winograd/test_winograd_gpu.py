"""GPU: the Winograd F(2x2, 3x3) forward kernel (csrc/winograd.hip, pdae_wino_fwd) against an fp64 reference and against the direct patch kernels
(conv3x3r / conv3x3p) it is a candidate to replace on weight-constant layers (F.conv2d(k=3, padding=1) of model/module.py:242,265).
Gate: 1e-5 of max |y| against fp64 -- the MATH_TOL of the direct f16x3 kernels (they measure 2.8e-7; Winograd's +-1 transforms amplify
rounding a little).  Cases cover: one tile, several chunks, multi-tile walks of the persistent workgroups with the prefetch pipeline
running across tile boundaries, two and four 64-channel halves, images whose borders put zero padding into every side of a tile,
no bias, and large / small activation scales inside the fp16 window."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.conftest import rel_err

pytestmark = pytest.mark.gpu


def rn(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _run(H, x, w, b):
    """x NCHW, w (Cout, Cin, 3, 3) on the CPU -> (winograd y, direct y) NCHW fp64 CPU, or None when the shape is not eligible."""
    N, C, Hh, W = x.shape
    Cout = w.shape[0]
    c = H.Conv(N, Hh, W, C, 0, Cout, k=3, math=4)
    nb = H.wino_wprep_bytes(c)
    if nb == 0:
        return None
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wd = w.permute(0, 2, 3, 1).contiguous().cuda()
    bd = b.cuda() if b is not None else None
    wp = torch.empty(nb // 4, device="cuda")
    H.wino_wprep(c, wd, wp)
    y = torch.full((N, Hh, W, Cout), float("nan"), device="cuda")
    H.wino_fwd(c, xd, wp, bd, y)
    yd = torch.empty(N, Hh, W, Cout, device="cuda")
    wpd = torch.empty(max(c.wprep_bytes(0), 4) // 4, device="cuda") if c.wprep_bytes(0) else None
    if wpd is not None:
        H.run(H.op_conv_wprep(c, wd, 0, wpd))
    H.run(H.op_conv_fwd(c, xd, None, wd, bd, yd, wp=wpd))
    torch.cuda.synchronize()
    return y.permute(0, 3, 1, 2).double().cpu(), yd.permute(0, 3, 1, 2).double().cpu()


@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout, bias, activation scale
    (1, 16, 16, 16, 64, True, 1.0),          # one tile, one chunk, one channel half
    (2, 32, 48, 48, 128, True, 1.0),         # 2 x 3 tiles, three chunks, two halves
    (3, 64, 64, 128, 64, False, 1.0),        # 48 tiles, eight chunks, no bias
    (1, 16, 32, 32, 256, True, 30.0),        # four channel halves; large activations (|V| ~ 4 x 30 x 4 x 16 stays inside the fp16 window)
    (5, 96, 80, 64, 128, True, 1e-3),        # 300 tiles: the 256 persistent workgroups walk one or two tiles; tiny activations
    (9, 128, 128, 32, 128, True, 1.0),       # 1152 tiles: four or five tiles per workgroup, pipeline across tile boundaries
])
@pytest.mark.parametrize("form", ["8", "1", "9", "0", "2"])
def test_winograd_forward_vs_fp64_and_direct(case, form, monkeypatch):
    """form = PDAE_WINO_SCHED: 1 / 0 four waves per workgroup (compiler-scheduled chunk, with / without an issue pattern), 2 four waves with twelve
    hand-placed units per chunk, 8 eight waves (two per SIMD, phase-skewed pairs: the default), 9 eight waves in a common order."""
    from pdae_amd import hip as H
    monkeypatch.setenv("PDAE_WINO_SCHED", form)
    if form in ("0", "2", "9") and case[0] * case[1] * case[2] > 20000:
        pytest.skip("large cases on the two main forms only")
    N, Hh, W, C, Cout, bias, sc = case
    x = (rn(1, N, C, Hh, W) * 1.3 + 0.2) * sc
    w = rn(2, Cout, C, 3, 3, scale=1.0 / math.sqrt(9 * C))
    b = rn(3, Cout, scale=0.2 * sc) if bias else None
    g = H.SaturationGuard.get("cuda")
    g.reset()
    out = _run(H, x, w, b)
    assert out is not None
    y, yd = out
    assert g.read()[0] == 0
    ref = F.conv2d(x.double(), w.double(), b.double() if bias else None, padding=1)
    e, ed = rel_err(y, ref), rel_err(yd, ref)
    print(f"[winograd form {form}] {case}: vs fp64 {e:.2e} (direct kernel {ed:.2e})")
    assert not torch.isnan(y).any()
    assert e < 1e-5, (e, ed)


def test_winograd_ineligible_shapes_are_refused():
    from pdae_amd import hip as H
    assert H.wino_wprep_bytes(H.Conv(1, 16, 16, 16, 0, 64, k=3, math=4)) == 2 * 16 * 2 * 64 * 8 * 2
    assert H.wino_wprep_bytes(H.Conv(1, 16, 16, 16, 0, 64, k=3, math=3)) == 0        # bf16x6: not offered
    assert H.wino_wprep_bytes(H.Conv(1, 24, 16, 16, 0, 64, k=3, math=4)) == 0        # H % 16
    assert H.wino_wprep_bytes(H.Conv(1, 16, 16, 24, 0, 64, k=3, math=4)) == 0        # C % 16
    assert H.wino_wprep_bytes(H.Conv(1, 16, 16, 16, 0, 32, k=3, math=4)) == 0        # Cout % 64
    assert H.wino_wprep_bytes(H.Conv(1, 16, 16, 16, 16, 64, k=3, math=4)) == 0       # two sources
    assert H.wino_wprep_bytes(H.Conv(1, 16, 16, 16, 0, 64, k=3, stride=2, math=4)) == 0
    c = H.Conv(1, 24, 16, 16, 0, 64, k=3, math=4)
    x = torch.zeros(1, 24, 16, 16, device="cuda")
    with pytest.raises(H.PdaeError, match="not eligible"):
        H.wino_fwd(c, x, x, None, x)


def test_winograd_window_overflow_is_counted():
    """Inputs whose 4-term sums leave the fp16 window after the 2^4 pre-scale trip the saturation counter (conservative bound 4 max |x|)."""
    from pdae_amd import hip as H
    x = rn(1, 1, 16, 16, 16) * 400.0
    w = rn(2, 64, 16, 3, 3, scale=0.1)
    g = H.SaturationGuard.get("cuda")
    g.reset()
    _run(H, x, w, None)
    assert g.read()[0] > 0
    g.reset()
