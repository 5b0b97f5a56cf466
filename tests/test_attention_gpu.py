"""GPU: the fused attention core (pdae_attn_fwd / pdae_attn_bwd, csrc/attention.hip) against an fp64 torch restatement of
QKVAttentionLegacy / QKVAttention (model/module.py:431-488) and its autograd, at the shapes of the path: (T, head width, heads) =
(64, 512, 1) and (256, 384, 1) of the FFHQ-128 decoder, (256, 64, 4) of the encoders, plus both channel orders at small widths.
Tolerance 1e-5 relative (three bf16 planes x six products with fp32 accumulation: fp32-grade; the reference itself is fp32)."""
import math

import numpy as np
import pytest
import torch

from tests.conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ref_attention(qkv, heads, new_order):
    """qkv: [N, T, 3C] float64 (pixel-major).  Returns out [N, T, C]."""
    N, T, C3 = qkv.shape
    C = C3 // 3
    ch = C // heads
    x = qkv.permute(0, 2, 1)                                    # [N, 3C, T] like the reference's conv1d output
    if new_order:
        q, k, v = x.chunk(3, dim=1)
        q, k, v = (u.reshape(N * heads, ch, T) for u in (q, k, v))
    else:
        q, k, v = x.reshape(N * heads, 3 * ch, T).split(ch, dim=1)
    scale = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * scale, k * scale), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v)
    return a.reshape(N, C, T).permute(0, 2, 1)


CASES = [(2, 64, 512, 1, False), (2, 256, 384, 1, False), (3, 256, 256, 4, False), (2, 256, 64, 2, True), (2, 64, 64, 2, True), (1, 128, 96, 3, False),
         (2, 192, 32, 1, True)]


@pytest.mark.parametrize("case", CASES)
def test_fused_attention_forward_backward_vs_fp64(case):
    from pdae_amd import hip as H
    N, T, C, heads, new_order = case
    assert H.attn_fused_ok(T, C, heads)
    g = torch.Generator().manual_seed(T + C)
    qkv = torch.randn(N, T, 3 * C, generator=g) * 1.5
    d_out = torch.randn(N, T, C, generator=g)
    q64 = qkv.double().requires_grad_(True)
    ref = ref_attention(q64, heads, new_order)
    ref.backward(d_out.double())
    qd, dd = qkv.to(DEV), d_out.to(DEV)
    out = torch.empty(N, T, C, device=DEV)
    lse = torch.empty(N * heads, T, device=DEV)
    H.run(H.op_attn_fwd(qd, N, T, C, heads, new_order, out, lse))
    assert rel_err(out, ref.detach()) < 1e-5
    # inference form: no log-sum-exp requested
    out2 = torch.empty_like(out)
    H.run(H.op_attn_fwd(qd, N, T, C, heads, new_order, out2, None))
    assert torch.equal(out, out2)
    dqkv = torch.full((N, T, 3 * C), float("nan"), device=DEV)          # every element must be written
    ws = torch.empty(N * heads, T, device=DEV)
    H.run(H.op_attn_bwd(qd, out, lse, dd, N, T, C, heads, new_order, dqkv, ws))
    assert torch.isfinite(dqkv).all()
    assert rel_err(dqkv, q64.grad) < 1e-5


def test_unsupported_shapes_report_and_engine_falls_back():
    from pdae_amd import hip as H
    assert not H.attn_fused_ok(144, 64, 1) and not H.attn_fused_ok(64, 48, 1) and not H.attn_fused_ok(512, 64, 1)
    x = torch.zeros(1, 144, 192, device=DEV)
    with pytest.raises(H.PdaeError, match="not supported"):
        H.run(H.op_attn_fwd(x, 1, 144, 64, 1, False, torch.zeros(1, 144, 64, device=DEV), None))


def test_attention_block_fused_equals_composed_path(monkeypatch):
    """The planned AttentionBlock (GroupNorm -> qkv conv -> attention -> proj + residual) with the fused kernels against the same block on the
    strided-batched GEMM + softmax composition (PDAE_FUSE_ATTN=0): forward and all parameter / input gradients."""
    from pdae_amd.engine import Plan, Builder
    from pdae_amd import hip as H
    N, S, C, heads = 2, 16, 64, 2
    g = torch.Generator().manual_seed(0)
    P = {"a.norm.weight": 1 + 0.1 * torch.randn(C, generator=g), "a.norm.bias": 0.1 * torch.randn(C, generator=g),
         "a.qkv.weight": torch.randn(3 * C, 1, 1, C, generator=g) / 8, "a.qkv.bias": 0.1 * torch.randn(3 * C, generator=g),
         "a.proj_out.weight": torch.randn(C, 1, 1, C, generator=g) / 8, "a.proj_out.bias": 0.1 * torch.randn(C, generator=g)}
    P = {k: v.to(DEV) for k, v in P.items()}
    x = torch.randn(N, S, S, C, generator=g).to(DEV)
    dy = torch.randn(N, S, S, C, generator=g).to(DEV)
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("PDAE_FUSE_ATTN", fused)
        Gr = {k: torch.zeros_like(v) for k, v in P.items()}
        pl = Plan(DEV)
        B = Builder(pl, P, Gr, save=True)
        out, ctx = B.attention("a", x, heads, False)
        assert (ctx.lse is not None) == (fused == "1")
        dx, _ = B.attention_bwd(ctx, dy, need_dx=True)
        pl.compile().run()
        torch.cuda.synchronize()
        res[fused] = (out.clone(), dx.clone(), {k: v.clone() for k, v in Gr.items()})
    assert rel_err(res["1"][0], res["0"][0]) < 1e-5 and rel_err(res["1"][1], res["0"][1]) < 1e-5
    for k in P:
        assert rel_err(res["1"][2][k], res["0"][2][k]) < 2e-5, k
