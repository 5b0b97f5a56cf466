"""Dense-grid form of the encoder's stride-2 3x3 convolutions (engine.Builder._dense_grid_desc): pdae_subsample2 / pdae_zero_insert2 exactly,
and the FFHQ encoder forward + backward with the form on (default) against the generic implicit-GEMM path (PDAE_S2_DENSE=0) and fp64 autograd
of the oracle."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pdae_amd import hip
    return hip


def test_subsample2_and_zero_insert2_are_exact(H):
    g = torch.Generator().manual_seed(3)
    for N, Hh, W, C in ((2, 8, 12, 8), (3, 16, 16, 256), (1, 2, 2, 4)):
        x = torch.randn(N, Hh, W, C, generator=g).cuda()
        y = torch.full((N, Hh // 2, W // 2, C), float("nan"), device="cuda")
        H.run(H.op_subsample2(x, N, Hh, W, C, y))
        assert torch.equal(y, x[:, ::2, ::2, :])
        z = torch.full((N, Hh * 2, W * 2, C), float("nan"), device="cuda")
        H.run(H.op_zero_insert2(x, N, Hh, W, C, z))
        ref = torch.zeros_like(z); ref[:, ::2, ::2, :] = x
        assert torch.equal(z, ref)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def _encoder_pass(flag, x0, dz, seed=41):
    from oracle import pdae_oracle as O
    from pdae_amd.model.representation_learning.encoder import FFHQEncoder
    os.environ["PDAE_S2_DENSE"] = flag
    try:
        enc = FFHQEncoder(device=torch.device("cuda"), latent_dim=512)
        sd = O.synth_state_dict(O.encoder_param_shapes("FFHQEncoder", 512), seed)
        enc.load_state_dict({k: v.clone() for k, v in sd.items()})
        enc.train()
        z = enc(x0)
        z.backward(dz)
        return z.detach().clone(), {k: v.detach().clone() for k, v in enc.grads().items()}, sd
    finally:
        os.environ.pop("PDAE_S2_DENSE", None)


def _close(got, ref, tol):
    floor = 1e-6 * max(float(v.double().norm()) for v in ref.values())
    bad = [(k, float((got[k].double().cpu() - r.double().cpu()).norm()), float(r.double().norm())) for k, r in ref.items()]
    bad = [(k, e / max(n, 1e-30)) for k, e, n in bad if e > tol * n + floor]
    assert not bad, sorted(bad, key=lambda b: -b[1])[:6]


def test_ffhq_encoder_forward_backward_dense_vs_generic_vs_oracle(H):
    from oracle import pdae_oracle as O
    g = torch.Generator().manual_seed(5)
    x0 = (torch.rand(4, 3, 128, 128, generator=g) * 2 - 1).cuda()
    dz = torch.randn(4, 512, generator=g).cuda()
    z1, g1, sd = _encoder_pass("1", x0, dz)
    z0, g0, _ = _encoder_pass("0", x0, dz)
    assert rel(z1, z0) < 1e-5
    _close(g1, g0, 5e-5)
    # fp64 autograd through the oracle's encoder
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    zr = O.encoder_forward(sd64, "FFHQEncoder", x0.double().cpu())
    zr.backward(dz.double().cpu())
    assert rel(z1.cpu(), zr.detach()) < 1e-5
    _close(g1, {k: v.grad for k, v in sd64.items()}, 5e-5)
