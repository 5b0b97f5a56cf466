"""CPU: the C-ABI library builds, loads and exports every symbol include/pdae_hip.h declares
(no compute calls without a GPU), and argument validation reports through pdae_last_error()."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from pdae_amd.build import build_library
    build_library(verbose=False)
    from pdae_amd import hip
    return hip.lib()


def test_header_symbols_exported(L):
    from pdae_amd import hip
    hdr = open(os.path.join(ROOT, "include", "pdae_hip.h")).read()
    declared = set(re.findall(r"\b(pdae_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.pdae_abi_version() == 11


def test_struct_layout_matches_header():
    from pdae_amd import hip
    assert ctypes.sizeof(hip.PdaeOp) == 8 + 20 * 8 + 24 * 8 + 12 * 8
    assert ctypes.sizeof(hip.ConvDesc) == 14 * 4


def test_invalid_arguments_fail_loudly(L):
    from pdae_amd import hip
    op = hip.make_op(999)
    with pytest.raises(hip.PdaeError, match="unknown op kind"):
        hip.run_ops(op, 1, stream=0)
    c = hip.Conv(1, 8, 8, 32, 0, 32)
    c.Ho = 5          # inconsistent geometry is rejected before any launch
    with pytest.raises(hip.PdaeError, match="inconsistent"):
        hip.run_ops(hip.make_op(hip.OP_CONV_FWD, [1, None, 1, None, None, 1], c.fields() + [0, 0]), 1, stream=0)


def test_workspace_queries(L):
    from pdae_amd import hip
    assert hip.gn_ws_bytes(2, 64) > 0 and hip.colsum_ws_bytes(100, 8) > 0
    assert hip.Conv(4, 32, 32, 32, 0, 64).wgrad_ws_bytes() > 16


def test_product_has_no_oracle_dependency():
    """The shipped package must never import the test oracle."""
    for dp, _, files in os.walk(os.path.join(ROOT, "pdae_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), os.path.join(dp, f)


def test_integration_md_snippet_runs_verbatim_on_gpu():
    """INTEGRATION.md shows a ctypes binding that calls pdae_conv2d_fwd DIRECTLY (descriptor struct, prepared weights, explicit stream) instead of
    going through pdae_run_ops: the code block is extracted from the document and executed as it stands."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU (marked gpu below for the driver's selection)")
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = md[md.index("   import ctypes, torch"):]
    code = code[:code.index("   ```")]
    code = "\n".join(l[3:] if l.startswith("   ") else l for l in code.splitlines())
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        exec(code, ns)
        g = torch.Generator().manual_seed(0)
        x = torch.randn(4, 32, 32, 64, generator=g).cuda()
        w = (torch.randn(128, 3, 3, 64, generator=g) / 24).cuda()
        b = torch.randn(128, generator=g).cuda()
        y = ns["conv3x3_nhwc"](x, w, b)
    finally:
        os.chdir(cwd)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 1e-5


test_integration_md_snippet_runs_verbatim_on_gpu = pytest.mark.gpu(test_integration_md_snippet_runs_verbatim_on_gpu)


def test_winograd_form_routing_is_a_pure_function_of_descriptor_and_switch(knob):
    """pdae_conv3x3_form (host-only): which prepared-weight layout / kernel a 3x3 convolution gets.  Chip-filling layers of the FFHQ-128 step take the
    Winograd F(2,3)-along-x form (conv3x3y), layers with fewer than 256 tiles or a badly filled last round do not, PDAE_W1=0 switches it off,
    PDAE_MATH_DIRECT in the descriptor pins the FORWARD form only, and preparation and launch read the same answer (the knob registry; the
    environment is read once)."""
    from pdae_amd import hip as H
    knob("PDAE_W1", 1)
    big = H.Conv(32, 128, 128, 128, 0, 128, k=3, math=4)                     # 2048 tiles
    wide32 = H.Conv(32, 32, 32, 256, 0, 256, k=3, math=4)                    # 256 tiles: one per CU
    small = H.Conv(32, 16, 16, 384, 0, 384, k=3, math=4)                     # 96 tiles
    ragged = H.Conv(3, 128, 128, 128, 0, 256, k=3, math=4)                   # 384 tiles: second round 50 % filled
    assert big.winograd_form(0) and big.winograd_form(0, gn=True) and big.winograd_form(1, f16_grad=True)
    assert wide32.winograd_form(0)
    # round 5: layers too small for 16-row tiles take the form in 8-row tiles (16^2 x 384 channels: 192 tiles; 3 x 128^2 x 256: 768 = three full rounds) ...
    assert small.winograd_form(0) and ragged.winograd_form(0)
    tiny = H.Conv(32, 16, 16, 128, 0, 128, k=3, math=4)                       # 64 tiles of 8 rows: a quarter of the chip -> direct, split over K
    assert not tiny.winograd_form(0) and not H.Conv(32, 8, 8, 512, 0, 512, k=3, math=4).winograd_form(0)
    knob("PDAE_W1_ROWS8", 0)                                                  # ... unless switched off (the round-4 routing)
    assert not small.winograd_form(0) and not ragged.winograd_form(0) and big.winograd_form(0)
    knob("PDAE_W1_ROWS8", 1)
    assert not H.Conv(32, 128, 128, 128, 0, 128, k=1, pad=0, math=4).winograd_form(0)
    assert not H.Conv(32, 128, 128, 128, 0, 128, k=3, math=3).winograd_form(0)          # bf16x6: the direct kernels
    pinned = H.Conv(32, 128, 128, 128, 0, 128, k=3, math=4, direct=True)
    assert not pinned.winograd_form(0) and pinned.winograd_form(1, f16_grad=True)      # the data gradient ignores the bit
    assert pinned.fields()[13] == 4 | H.MATH_DIRECT and H.Conv(*[32, 128, 128, 128, 0, 128], math=pinned.fields()[13]).direct
    cs = H.Conv(32, 128, 128, 256, 0, 128, k=1, math=4)
    assert not H.conv_fwd_skip_ok(big, cs) and H.conv_fwd_skip_ok(pinned, cs)           # fused skip chunks: direct form only
    knob("PDAE_W1", 0)
    assert not big.winograd_form(0) and H.conv_fwd_skip_ok(big, cs)
    knob("PDAE_W1", 2)
    assert big.winograd_form(0) and not small.winograd_form(0)                          # 2 = every ELIGIBLE shape; a split-K plan stays direct


def test_weight_gradient_form_query(L):
    """pdae_conv2d_wgrad_form (ABI 11): which kernel a weight gradient runs on -- host logic only, so the routing of the round-6 producer / consumer
    kernel (conv3x3v) is observable without a GPU: whole 64-channel blocks on both sides, 16-pixel-wide tiles, >= 64 pixel tiles, two-plane formats;
    everything else of the 3x3 / stride-1 family stays on conv3x3w; the knob PDAE_W3V switches it off."""
    from pdae_amd import hip
    C = hip.Conv
    assert hip.conv_wgrad_form(C(32, 128, 128, 128, 0, 128, math=4)) == 3
    assert hip.conv_wgrad_form(C(32, 16, 16, 384, 0, 384, math=4)) == 3                 # 64 pixel tiles: just enough
    assert hip.conv_wgrad_form(C(32, 64, 64, 128, 128, 128, math=4), with_gn_input=True) == 3
    assert hip.conv_wgrad_form(C(32, 128, 128, 128, 0, 128, math=4), with_dy_amax=False) == 2     # no dY scale -> three-plane format -> conv3x3w
    assert hip.conv_wgrad_form(C(32, 8, 8, 512, 0, 512, math=4)) == 2                   # 8-pixel-wide level: image-pair tiles
    assert hip.conv_wgrad_form(C(32, 32, 32, 96, 0, 128, math=4)) == 2                  # 96 input channels: not whole 64-channel blocks
    assert hip.conv_wgrad_form(C(8, 16, 16, 384, 0, 384, math=4)) == 0                  # 16 pixel tiles: too few for either 3x3 kernel -> generic
    assert hip.conv_wgrad_form(C(32, 16, 16, 384, 0, 1152, k=1, pad=0, math=4)) == 1    # 1x1: its own kernel
    assert hip.conv_wgrad_form(C(32, 128, 128, 128, 0, 3, math=4)) == 1                 # image head
    assert hip.conv_wgrad_form(C(32, 128, 128, 128, 0, 128, math=0)) == 0               # f32 mode: generic implicit GEMM
    old = hip.get_knob("PDAE_W3V")
    try:
        hip.set_knob("PDAE_W3V", 0)
        assert hip.conv_wgrad_form(C(32, 128, 128, 128, 0, 128, math=4)) == 2
    finally:
        hip.set_knob("PDAE_W3V", old)
