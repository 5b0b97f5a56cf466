"""GPU: planned networks (UNet / ShiftUNet / encoders) through the HIP path against
  (a) the committed golden vectors produced by the reference itself, and
  (b) the CPU oracle on the same seeded weights.
Tolerance: 1e-4 relative for network outputs and the loss (north_star), 1e-3 on gradient norms
(reduction-order noise through ~50 layers), small gradients compared element-wise at 2e-3."""
import numpy as np
import pytest
import torch

from tests.conftest import load_golden, T, rel_err
from tests.golden import make_fixtures_cfg as C
from oracle import pdae_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def load_into(net, sd):
    net.load_state_dict({k: v for k, v in sd.items()})
    return net


def check_grads(G, g, prefix="", tol_norm=1e-3, tol_elem=2e-3):
    keys = [str(k) for k in g["grad_keys"]]
    # gradients that are analytically zero (a bias feeding a GroupNorm with one channel per group) are pure
    # rounding noise on both sides: floor the comparison at 1e-6 of the largest gradient norm / value
    floor_n = 1e-6 * float(np.max(g["grad_summary"][:, 1]))
    floor_e = 1e-6 * float(np.max(g["grad_summary"][:, 2]))
    bad = []
    for k, (gs, gn, gm) in zip(keys, g["grad_summary"]):
        if not k.startswith(prefix):
            continue
        got = G[k[len(prefix):]]
        n = float(got.double().norm())
        if abs(n - gn) > tol_norm * gn + floor_n:
            bad.append((k, n, gn))
    assert not bad, bad[:8]
    for k in g:
        if k.startswith("g__" + prefix):
            ref = T(g[k]).double()
            err = float((G[k[3 + len(prefix):]].detach().double().cpu() - ref).abs().max())
            assert err < tol_elem * float(ref.abs().max()) + floor_e, (k, err)


@pytest.mark.parametrize("tag,cfg", [("unet_a", C.CFG_UNET_A), ("unet_b", C.CFG_UNET_B)])
def test_unet_forward_backward_vs_golden(tag, cfg):
    from pdae_amd.model.unet import UNet
    g = load_golden(tag)
    net = load_into(UNet(device=DEV, **cfg), O.synth_state_dict(O.unet_param_shapes(cfg), int(g["seed"])))
    x_t, t = T(g["x_t"]).to(DEV), T(g["t"]).to(DEV)
    cond = T(g["cond"]).to(DEV) if g["cond"].size else None
    with torch.no_grad():
        out = net(x_t, t, cond)
    assert out.shape == x_t.shape
    assert rel_err(out, g["out"]) < 1e-4
    # training path through the autograd bridge (regular_train_one_batch, gaussian_diffusion.py:199-211)
    noise = T(g["noise"]).to(DEV)
    net.train()
    out = net(x_t, t, cond)
    loss = torch.mean((noise - out) ** 2)
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    check_grads(net.grads(), g)
    assert net.P["out.2.weight"].grad is net.grads()["out.2.weight"]


def test_shift_unet_forward_vs_golden_and_oracle():
    from pdae_amd.model.shift_unet import ShiftUNet
    g = load_golden("shift_tiny")
    cfg, latent = C.CFG_SHIFT_T, int(g["latent"])
    sd = O.synth_state_dict(O.unet_param_shapes(cfg, shift=True, latent_dim=latent), int(g["seed"]))
    net = load_into(ShiftUNet(device=DEV, latent_dim=latent, **cfg), sd)
    with torch.no_grad():
        eps, shift = net(T(g["x"]).to(DEV), T(g["t"]).to(DEV), T(g["z"]).to(DEV))
    assert rel_err(eps, g["eps"]) < 1e-4 and rel_err(shift, g["shift"]) < 1e-4
    # a different batch size / odd spatial size against the oracle (plan cache keyed by shape)
    x = torch.randn(3, 3, 24, 24, generator=torch.Generator().manual_seed(5))
    t = torch.tensor([0, 999, 312])
    z = torch.randn(3, latent, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        eps, shift = net(x.to(DEV), t.to(DEV), z.to(DEV))
        e_ref, s_ref = O.shift_unet_forward(sd, cfg, x, t, z)
    assert rel_err(eps, e_ref) < 1e-4 and rel_err(shift, s_ref) < 1e-4


def test_encoder_ffhq_vs_golden():
    from pdae_amd.model.representation_learning.encoder import FFHQEncoder
    g = load_golden("misc")
    enc = load_into(FFHQEncoder(device=DEV, latent_dim=512), O.synth_state_dict(O.encoder_param_shapes("FFHQEncoder", 512), 41))
    with torch.no_grad():
        z = enc(T(g["enc_x0"]).to(DEV))
    assert rel_err(z, g["enc_z"]) < 1e-4


def test_rl_train_step_through_autograd_bridge_vs_golden():
    """representation_learning_train_one_batch (gaussian_diffusion.py:234-255) + backward, (t, noise) injected."""
    from pdae_amd.model.shift_unet import ShiftUNet
    from pdae_amd.model.representation_learning.encoder import CELEBA64Encoder
    g = load_golden("rl_step")
    cfg = C.CFG_SHIFT_64
    enc = load_into(CELEBA64Encoder(device=DEV, latent_dim=512), O.synth_state_dict(O.encoder_param_shapes("CELEBA64Encoder", 512), int(g["seed_enc"])))
    dec = load_into(ShiftUNet(device=DEV, latent_dim=512, **cfg), O.synth_state_dict(O.unet_param_shapes(cfg, shift=True, latent_dim=512), int(g["seed_dec"])))
    enc.train(); dec.set_train_mode()
    s = O.Schedules()
    x0, t, noise = T(g["x0"]).to(DEV), T(g["t0"]).to(DEV), T(g["noise0"]).to(DEV)
    z = enc(x0)
    assert rel_err(z, g["z"]) < 1e-4
    x_t = O.q_sample(s, T(g["x0"]), T(g["t0"]), T(g["noise0"])).to(DEV)
    eps, shift = dec(x_t, t, z)
    assert rel_err(eps, g["eps"]) < 1e-4 and rel_err(shift, g["shift"]) < 1e-4
    sc = s.shift_coef.to(DEV)[t].view(-1, 1, 1, 1)
    w = s.weight.to(DEV)[t].view(-1, 1, 1, 1)
    loss = torch.mean(w * (noise - (eps + sc * shift)) ** 2)
    assert abs(loss.item() - float(g["loss0"])) < 1e-4 * abs(float(g["loss0"]))
    loss.backward()
    check_grads(dec.grads(), g, prefix="dec::")
    check_grads(enc.grads(), g, prefix="enc::")
    # frozen half: no gradient buffers exist for it
    assert "out.2.weight" not in dec.grads() and "input_blocks.0.0.weight" not in dec.grads()


def test_fused_rl_step_three_steps_vs_golden():
    """FusedRLStep (encoder fwd .. Adam+EMA in one plan): losses of 3 consecutive steps and the parameters / EMA after
    steps 1 and 3 against torch.optim.Adam driving the reference (tests/golden/make_fixtures.py section 4)."""
    import copy
    from pdae_amd.model.shift_unet import ShiftUNet
    from pdae_amd.model.representation_learning.encoder import CELEBA64Encoder
    from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    from pdae_amd.trainer.fused_step import FusedRLStep
    g = load_golden("rl_step")
    cfg = C.CFG_SHIFT_64
    enc = load_into(CELEBA64Encoder(device=DEV, latent_dim=512), O.synth_state_dict(O.encoder_param_shapes("CELEBA64Encoder", 512), int(g["seed_enc"])))
    dec = load_into(ShiftUNet(device=DEV, latent_dim=512, **cfg), O.synth_state_dict(O.unet_param_shapes(cfg, shift=True, latent_dim=512), int(g["seed_dec"])))
    frozen_before = dec.flat_frozen.clone()
    enc.train(); dec.set_train_mode()
    ema_enc, ema_dec = copy.deepcopy(enc), copy.deepcopy(dec)
    gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, torch.device(DEV))
    st = FusedRLStep(gd, enc, dec, ema_enc, ema_dec, 2, 64, 64, lr=1e-4, ema_decay=0.9999)
    x0 = T(g["x0"]).to(DEV)
    for step in range(3):
        loss = st.step(x0, t=T(g[f"t{step}"]).to(DEV), noise=T(g[f"noise{step}"]).to(DEV))
        ref = float(g[f"loss{step}"])
        assert abs(loss.item() - ref) < 2e-4 * abs(ref), (step, loss.item(), ref)
        if step == 0:
            assert rel_err(st.z, g["z"]) < 1e-4
            check_grads(dec.grads(), g, prefix="dec::")
            check_grads(enc.grads(), g, prefix="enc::")
        if step in (0, 2):
            for k in g:
                if k.startswith(f"p{step + 1}__"):
                    kind, name = k.split("__", 1)[1].split("::")
                    net, ema = (enc, ema_enc) if kind == "enc" else (dec, ema_dec)
                    # Adam's first steps move every weight by ~lr regardless of gradient scale: compare the UPDATE, not the value
                    sd0 = O.synth_state_dict(O.encoder_param_shapes("CELEBA64Encoder", 512) if kind == "enc" else
                                             O.unet_param_shapes(cfg, shift=True, latent_dim=512), int(g["seed_enc" if kind == "enc" else "seed_dec"]))
                    upd_ref = T(g[k]).double() - sd0[name].double()
                    upd = net.P[name].detach().double().cpu() - sd0[name].double()
                    assert float((upd - upd_ref).abs().max()) < 0.02 * float(upd_ref.abs().max()) + 1e-9, k
                    assert rel_err(ema.P[name], g[f"ema{step + 1}__{kind}::{name}"]) < 1e-6, k
    assert torch.equal(dec.flat_frozen, frozen_before)
    assert torch.equal(ema_dec.flat_frozen, frozen_before)


# ---------------------------------------------------------------------------------------------
# MLPSkipNet (latent DPM, config #5; model/mlp_skip_net.py)
# ---------------------------------------------------------------------------------------------
def _mlp_pair(cfg, seed):
    from pdae_amd.model.mlp_skip_net import MLPSkipNet
    sd = O.synth_state_dict(O.mlp_skip_net_param_shapes(cfg), seed)
    net = MLPSkipNet(device=DEV, **cfg)
    net.load_state_dict(sd, strict=False)          # the duplicate cond_layers.1.* keys alias linear_emb.*
    return net, sd


def test_mlp_skip_net_forward_vs_golden():
    g = load_golden("misc")
    net, _ = _mlp_pair(C.CFG_MLP, 42)
    with torch.no_grad():
        out = net(T(g["mlp_z"]).to(DEV), T(g["mlp_t"]).to(DEV))
    assert rel_err(out, g["mlp_out"]) < 1e-4


def test_mlp_skip_net_state_dict_has_reference_alias_keys():
    net, sd = _mlp_pair(C.CFG_MLP, 42)
    keys = set(net.state_dict().keys())
    for i in range(C.CFG_MLP["num_layers"] - 1):
        assert f"layers.{i}.cond_layers.1.weight" in keys and f"layers.{i}.linear_emb.weight" in keys
    assert net.state_dict()["layers.0.cond_layers.1.weight"].data_ptr() == net.P["layers.0.linear_emb.weight"].data_ptr()
    assert sum(p.numel() for p in net.parameters()) == sum(v.numel() for v in sd.values())


@pytest.mark.parametrize("cfg,R", [(C.CFG_MLP, 6), (dict(input_channel=512, model_channel=2048, num_layers=4, time_emb_channel=64, use_norm=True, dropout=0.0), 16),
                                   (dict(input_channel=64, model_channel=96, num_layers=3, time_emb_channel=32, use_norm=False, dropout=0.0), 5)])
def test_mlp_skip_net_backward_vs_oracle_autograd(cfg, R):
    """L1 latent-DPM loss (gaussian_diffusion.py:373-398) through the autograd bridge against torch autograd of the oracle."""
    net, sd = _mlp_pair(cfg, 7)
    gen = torch.Generator().manual_seed(3)
    z = torch.randn(R, cfg["input_channel"], generator=gen)
    noise = torch.randn(R, cfg["input_channel"], generator=gen)
    t = torch.randint(0, 1000, (R,), generator=gen)
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    zr = z.clone().requires_grad_(True)
    ref_out = O.mlp_skip_net_forward(ref_sd, cfg, zr, t)
    ref_loss = (noise - ref_out).abs().mean()
    ref_loss.backward()
    net.train()
    zd = z.to(DEV).requires_grad_(True)
    out = net(zd, t.to(DEV))
    assert rel_err(out, ref_out.detach()) < 1e-4
    loss = (noise.to(DEV) - out).abs().mean()
    assert abs(loss.item() - ref_loss.item()) < 1e-5 * abs(ref_loss.item()) + 1e-7
    loss.backward()
    G = net.grads()
    for k, v in ref_sd.items():
        assert rel_err(G[k], v.grad) < 2e-4, k
    assert rel_err(zd.grad, zr.grad) < 2e-4


def test_latent_diffusion_train_one_batch_and_sample_loop():
    """latent_diffusion_train_one_batch / latent_ddim_sample_loop (gaussian_diffusion.py:373-415, ddim.py:200-207) vs the oracle."""
    from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    cfg = C.CFG_MLP
    net, sd = _mlp_pair(cfg, 11)
    gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, torch.device(DEV))
    R, ic = 8, cfg["input_channel"]
    gen = torch.Generator().manual_seed(5)
    z0, noise, t = torch.randn(R, ic, generator=gen), torch.randn(R, ic, generator=gen), torch.randint(0, 1000, (R,), generator=gen)
    mean, std = torch.zeros(ic), torch.ones(ic)
    enc = lambda x: x                                          # encoder stand-in: x_0 already is the latent
    net.train()
    out = gd.latent_diffusion_train_one_batch(net, enc, z0.to(DEV), mean.to(DEV), std.to(DEV), t=t.to(DEV), noise=noise.to(DEV))
    ac = np.cumprod(1.0 - np.full(1000, 0.008))
    z_t = torch.tensor(np.sqrt(ac), dtype=torch.float32)[t].view(-1, 1) * z0 + torch.tensor(np.sqrt(1 - ac), dtype=torch.float32)[t].view(-1, 1) * noise
    ref = (noise - O.mlp_skip_net_forward(sd, cfg, z_t, t)).abs().mean()
    assert abs(out["prediction_loss"].item() - ref.item()) < 1e-4 * abs(ref.item())
    out["prediction_loss"].backward()
    assert float(net.grads()["layers.0.linear.weight"].abs().sum()) > 0
    # respaced latent DDIM sampling (clamping variant) for 10 steps
    net.eval()
    zT = torch.randn(R, ic, generator=gen).clamp_(-1, 1)
    dd = gd._ddim("ddim10", gd.latent_diffusion_config["_ac_host"])
    with torch.no_grad():
        z = dd.latent_ddim_sample_loop(net, zT.to(DEV))
    from types import SimpleNamespace
    tab = O.DDIMTables(SimpleNamespace(alphas_cumprod=torch.tensor(ac, dtype=torch.float32)), "ddim10")
    zr = zT.clone()
    for i in reversed(range(1, tab.timesteps + 1)):
        tt = torch.full((R,), i, dtype=torch.long)
        zr = O.ddim_update(tab, zr, tt, O.mlp_skip_net_forward(sd, cfg, zr, tab.timestep_map[tt]))
    assert rel_err(z, zr) < 1e-4


def test_frozen_prepared_weights_follow_parameter_reloads():
    """Plans cache the bf16-split copies of the frozen trunk's conv weights; load_state_dict must refresh them."""
    from pdae_amd.model.shift_unet import ShiftUNet
    g = load_golden("shift_tiny")
    cfg, latent = C.CFG_SHIFT_T, int(g["latent"])
    sd = O.synth_state_dict(O.unet_param_shapes(cfg, shift=True, latent_dim=latent), int(g["seed"]))
    net = ShiftUNet(device=DEV, latent_dim=latent, **cfg)          # random initial weights
    x, t, z = T(g["x"]).to(DEV), T(g["t"]).to(DEV), T(g["z"]).to(DEV)
    with torch.no_grad():
        eps0, _ = net(x, t, z)                                     # builds the plan and its prepared-weight cache
        assert rel_err(eps0, g["eps"]) > 1e-2
        net.load_state_dict(sd)                                    # same plan, new frozen weights
        eps, shift = net(x, t, z)
    assert rel_err(eps, g["eps"]) < 1e-4 and rel_err(shift, g["shift"]) < 1e-4


def test_autograd_bridge_accumulates_like_torch_and_guards_stale_activations():
    """loss.backward() called for several micro-batches between zero_grad() and step() (runner_config.num_iterations > 1) must
    SUM the gradients; optimizer.zero_grad() of a torch optimizer (.grad = None) must restart from zero; a backward whose activations were
    overwritten by a later forward must raise instead of returning a wrong gradient."""
    from pdae_amd.model.unet import UNet
    cfg = C.CFG_UNET_B
    net = load_into(UNet(device=DEV, **cfg), O.synth_state_dict(O.unet_param_shapes(cfg), 3))
    net.train()
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn(2, 1, 16, 16, generator=g).to(DEV) for _ in range(2)]
    t = torch.tensor([10, 500], device=DEV)

    def grad_of(batches):
        opt = torch.optim.SGD(net.parameters(), lr=0.0)
        opt.zero_grad()                                  # set_to_none=True: every .grad becomes None
        for x in batches:
            net(x, t).pow(2).mean().backward()
        assert all(p.grad is not None for p in net.parameters())
        return net.flat_grad.clone()

    g0, g1 = grad_of(xs[:1]), grad_of(xs[1:])
    both = grad_of(xs)
    assert float((both - (g0 + g1)).norm() / both.norm()) < 1e-5
    assert float((grad_of(xs[:1]) - g0).norm()) == 0.0   # zero_grad really restarted the sum
    y_old = net(xs[0], t)
    net(xs[1], t)                                        # overwrites the plan's saved activations
    with pytest.raises(RuntimeError, match="overwritten"):
        y_old.sum().backward()
