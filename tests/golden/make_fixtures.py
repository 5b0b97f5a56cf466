"""Generates tests/golden/*.npz by IMPORTING THE REFERENCE (ckczzj/PDAE) on CPU.

Runs only in the build container, where /root/reference exists:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py
Nothing here travels to the GPU box except the emitted .npz vectors (inputs and
expected outputs only -- weights are re-derived on the box from a seed through
`oracle.pdae_oracle.synth_state_dict`, which needs no reference code).

What the vectors pin: module forward values, the ShiftUNet
(eps, shift) pair, the representation-learning train step (loss, every trainable
gradient, Adam + EMA after 1 and 3 steps with torch.optim.Adam), DDIM encode/sample
trajectories, schedule tables / respacing maps, MLPSkipNet, SSIM / MSE.
"""
import os
import sys
import copy

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np
import torch

torch.manual_seed(0)
torch.set_num_threads(8)

from model.unet import UNet                                     # reference
from model.shift_unet import ShiftUNet                          # reference
from model.mlp_skip_net import MLPSkipNet                       # reference
from model.module import timestep_embedding as ref_temb        # reference
import model.representation_learning.encoder as ref_enc        # reference
from diffusion.gaussian_diffusion import GaussianDiffusion      # reference
from diffusion.ddim import DDIM                                 # reference
from metric.utils import calculate_ssim, calculate_mse          # reference

from oracle import pdae_oracle as O                             # only for shapes + synthetic weights

from tests.golden.make_fixtures_cfg import CFG_UNET_A, CFG_UNET_B, CFG_SHIFT_T, CFG_SHIFT_64, CFG_MLP


def load_synth(module, shapes, seed):
    sd = O.synth_state_dict(shapes, seed)
    ref_keys = {k: tuple(v.shape) for k, v in module.state_dict().items() if "cond_layers" not in k}
    assert ref_keys == {k: tuple(v) for k, v in shapes.items()}, "oracle key/shape table differs from the reference"
    assert list(ref_keys.keys()) == list(shapes.keys()), "key ORDER differs from the reference"
    module.load_state_dict(sd, strict=False)
    return sd


def rnd(seed, *shape):
    return torch.tensor(np.random.default_rng(seed).standard_normal(shape), dtype=torch.float32)


def uni(seed, *shape):
    return torch.tensor(np.random.default_rng(seed).uniform(-1, 1, shape), dtype=torch.float32)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items() if v.size > 1})


def grad_summary(named):
    """Per-tensor [sum, l2, abs-max] plus full arrays for small tensors."""
    keys = sorted(named.keys())
    summ = np.array([[float(named[k].double().sum()), float(named[k].double().norm()), float(named[k].abs().max())] for k in keys])
    return keys, summ


def main():
    dev = torch.device("cpu")
    gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, dev)

    # ---- 1. schedules, respacing, timestep embedding ---------------------------------
    gd_cos = GaussianDiffusion({"timesteps": 1000, "betas_type": "cosine"}, dev)
    names = ["alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
             "sqrt_recip_alphas_cumprod_m1", "posterior_variance", "posterior_log_variance_clipped",
             "x_0_posterior_mean_x_0_coef", "x_0_posterior_mean_x_t_coef", "noise_posterior_mean_x_t_coef",
             "noise_posterior_mean_noise_coef", "shift_coef", "weight"]
    arrs = {"lin_" + n: getattr(gd, n) for n in names}
    arrs.update({"cos_" + n: getattr(gd_cos, n) for n in ["alphas_cumprod", "shift_coef", "weight"]})
    for style in ["ddim10", "ddim20", "ddim100", "ddim1000"]:
        nb, tmap = gd.get_ddim_betas_and_timestep_map(style, gd.alphas_cumprod.cpu().numpy())
        d = DDIM(nb, tmap, dev)
        arrs[style + "_map"] = tmap
        arrs[style + "_betas"] = nb
        for n in ["alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_one_minus_alphas_cumprod",
                  "sqrt_recip_alphas_cumprod", "sqrt_recip_alphas_cumprod_m1"]:
            arrs[style + "_" + n] = getattr(d, n)
    tt = torch.tensor([0, 1, 7, 500, 999], dtype=torch.long)
    arrs["temb_t"] = tt
    arrs["temb_32"] = ref_temb(tt, 32)
    arrs["temb_128"] = ref_temb(tt, 128)
    save("schedules", **arrs)

    # ---- 2. UNet forward + regular loss grads ----------------------------------------
    for tag, cfg, hw, seed in [("unet_a", CFG_UNET_A, 16, 11), ("unet_b", CFG_UNET_B, 32, 12)]:
        net = UNet(**cfg).eval()
        sd = load_synth(net, O.unet_param_shapes(cfg), seed)
        B = 2
        x0 = uni(seed + 100, B, cfg["input_channel"], hw, hw)
        noise = rnd(seed + 200, B, cfg["input_channel"], hw, hw)
        t = torch.tensor([17, 803], dtype=torch.long)
        cond = torch.tensor([3, 8], dtype=torch.long) if cfg.get("num_class") else None
        x_t = gd.q_sample(x0, t, noise)
        for p in net.parameters():
            p.requires_grad_(True)
        out = net(x_t, t, cond)
        loss = gd.p_loss(noise, out)
        loss.backward()
        keys, summ = grad_summary({k: p.grad for k, p in net.named_parameters()})
        small = {("g__" + k): p.grad for k, p in net.named_parameters() if p.numel() <= 2048}
        save(tag, x0=x0, noise=noise, t=t, cond=(cond if cond is not None else np.zeros(0)), x_t=x_t, out=out,
             loss=loss, grad_keys=np.array(keys), grad_summary=summ, seed=seed, **small)

    # ---- 3. ShiftUNet forward + DDIM trajectories (16x16) ----------------------------
    latent = 64
    dec = ShiftUNet(latent_dim=latent, **CFG_SHIFT_T)
    dec_sd = load_synth(dec, O.unet_param_shapes(CFG_SHIFT_T, shift=True, latent_dim=latent), 21)
    dec.eval()
    B = 2
    x = rnd(301, B, 3, 16, 16)
    t = torch.tensor([5, 640], dtype=torch.long)
    z = rnd(302, B, latent)
    with torch.no_grad():
        eps, shift = dec(x, t, z)
    x0 = uni(303, B, 3, 16, 16)

    class _Enc(torch.nn.Module):        # the loops only call encoder when z is None; we pass z
        pass

    with torch.no_grad():
        nb, tmap = gd.get_ddim_betas_and_timestep_map("ddim20", gd.alphas_cumprod.cpu().numpy())
        d20 = DDIM(nb, tmap, dev)
        traj_e = []
        xt = x0
        for i in range(0, d20.timesteps):
            tt_ = torch.full((B,), i, dtype=torch.long)
            xt = d20.shift_ddim_encode(dec, z, xt, tt_)
            traj_e.append(xt.clone())
        x_T = gd.representation_learning_ddim_encode("ddim20", None, dec, x0, z)
        assert torch.equal(x_T, traj_e[-1])
        x_rec = gd.representation_learning_ddim_sample("ddim10", None, dec, None, x_T, z)
        x_rec_stop = gd.representation_learning_ddim_sample("ddim10", None, dec, None, x_T, z, stop_percent=0.3)
        # the README protocol, ddim1000 -> ddim100, on the tiny net
        x_T_1000 = gd.representation_learning_ddim_encode("ddim1000", None, dec, x0, z)
        x_rec_100 = gd.representation_learning_ddim_sample("ddim100", None, dec, None, x_T_1000, z)
        n0, n1 = (x0 + 1.0) / 2.0, (x_rec_100 + 1.0) / 2.0
        ssim_v, mse_v = calculate_ssim(n0, n1), calculate_mse(n0, n1)
        n2 = (x_rec + 1.0) / 2.0
        ssim_s, mse_s = calculate_ssim(n0, n2), calculate_mse(n0, n2)
    save("shift_tiny", x=x, t=t, z=z, eps=eps, shift=shift, x0=x0, enc_traj=torch.stack(traj_e[:3] + traj_e[-1:]),
         x_T=x_T, x_rec=x_rec, x_rec_stop=x_rec_stop, x_T_1000=x_T_1000, x_rec_100=x_rec_100,
         ssim_100=ssim_v, mse_100=mse_v, ssim_10=ssim_s, mse_10=mse_s, seed=21, latent=latent)

    # ---- 4. representation-learning train step (64x64) -------------------------------
    latent = 512
    enc = ref_enc.CELEBA64Encoder(latent_dim=latent)
    enc_sd = load_synth(enc, O.encoder_param_shapes("CELEBA64Encoder", latent), 31)
    dec = ShiftUNet(latent_dim=latent, **CFG_SHIFT_64)
    dec_sd = load_synth(dec, O.unet_param_shapes(CFG_SHIFT_64, shift=True, latent_dim=latent), 32)
    enc.train(); dec.set_train_mode()
    ema_enc, ema_dec = copy.deepcopy(enc), copy.deepcopy(dec)
    B = 2
    x0 = uni(401, B, 3, 64, 64)
    ts = [torch.tensor([123, 877]), torch.tensor([3, 500]), torch.tensor([999, 42])]
    noises = [rnd(410 + i, B, 3, 64, 64) for i in range(3)]
    params = [p for p in enc.parameters()] + [p for p in dec.parameters() if p.requires_grad]
    opt = torch.optim.Adam([{"params": list(enc.parameters())},
                            {"params": list(dec.label_emb.parameters())},
                            {"params": list(dec.shift_middle_block.parameters())},
                            {"params": list(dec.shift_output_blocks.parameters())},
                            {"params": list(dec.shift_out.parameters())}],
                           lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    watch = ["enc::encoder.0.bias", "enc::encoder.14.bias", "dec::label_emb.bias", "dec::shift_out.2.weight",
             "dec::shift_middle_block.0.emb_z_layers.1.bias", "dec::shift_output_blocks.0.0.out_layers.0.weight"]
    out = dict(x0=x0, seed_enc=31, seed_dec=32)
    decay = 0.9999
    for step in range(3):
        t, noise = ts[step], noises[step]
        # reproduce representation_learning_train_one_batch (gaussian_diffusion.py:234-255) with injected t/noise
        opt.zero_grad()
        zz = enc(x0)
        x_t = gd.q_sample(x0, t, noise)
        e, g = dec(x_t, t, zz)
        sc = gd.extract_coef_at_t(gd.shift_coef, t, x0.shape)
        w = gd.extract_coef_at_t(gd.weight, t, x0.shape)
        loss = gd.p_loss(noise, e + sc * g, weight=w)
        loss.backward()
        named = {"enc::" + k: p.grad for k, p in enc.named_parameters()}
        named.update({"dec::" + k: p.grad for k, p in dec.named_parameters() if p.requires_grad})
        if step == 0:
            # check the reference really draws (t, noise) this way: seed and call the real method
            keys, summ = grad_summary(named)
            out.update(z=zz, eps=e, shift=g, loss0=loss, grad_keys=np.array(keys), grad_summary=summ)
            for k in named:
                if named[k].numel() <= 4096:
                    out["g__" + k] = named[k].clone()
            frozen_grad = [k for k, p in dec.named_parameters() if (not p.requires_grad) and p.grad is not None]
            assert not frozen_grad
        out[f"t{step}"] = t
        out[f"noise{step}"] = noise
        out[f"loss{step}"] = loss.detach().clone()
        opt.step()
        # accumulate (train_representation_learning.py:192-212)
        for (k, pe), (_, p) in zip(ema_enc.named_parameters(), enc.named_parameters()):
            pe.data.mul_(decay).add_(p.data, alpha=1.0 - decay)
        for (k, pe), (_, p) in zip(ema_dec.named_parameters(), dec.named_parameters()):
            if p.requires_grad:
                pe.data.mul_(decay).add_(p.data, alpha=1.0 - decay)
        if step in (0, 2):
            cur = {"enc::" + k: p for k, p in enc.named_parameters()}
            cur.update({"dec::" + k: p for k, p in dec.named_parameters()})
            cure = {"enc::" + k: p for k, p in ema_enc.named_parameters()}
            cure.update({"dec::" + k: p for k, p in ema_dec.named_parameters()})
            for k in watch:
                out[f"p{step + 1}__" + k] = cur[k].detach().clone()
                out[f"ema{step + 1}__" + k] = cure[k].detach().clone()
    # frozen params must be untouched
    for k, p in dec.named_parameters():
        if not p.requires_grad:
            assert torch.equal(p.data, dec_sd[k]), k
    save("rl_step", **out)

    # the reference's own RNG draw order (gaussian_diffusion.py:240-241): randint then randn_like
    torch.manual_seed(666666666 + 2)
    t_ref = torch.randint(0, 1000, (B,), dtype=torch.long)
    n_ref = torch.randn_like(x0)
    torch.manual_seed(666666666 + 2)
    enc.eval()
    res = gd.representation_learning_train_one_batch(enc, dec, x0)["prediction_loss"]
    save("rl_rng", t=t_ref, noise_head=n_ref.flatten()[:64], loss=res)

    # ---- 5. FFHQ encoder (128x128) + MLPSkipNet + metrics ----------------------------
    enc = ref_enc.FFHQEncoder(latent_dim=512).eval()
    load_synth(enc, O.encoder_param_shapes("FFHQEncoder", 512), 41)
    x0 = uni(501, 1, 3, 128, 128)
    with torch.no_grad():
        zf = enc(x0)
    mlp = MLPSkipNet(**CFG_MLP).eval()
    load_synth(mlp, O.mlp_skip_net_param_shapes(CFG_MLP), 42)
    zt = rnd(502, 3, 64)
    tm = torch.tensor([0, 250, 999], dtype=torch.long)
    with torch.no_grad():
        mo = mlp(zt, tm)
    a, b = uni(503, 2, 3, 32, 32) * 0.5 + 0.5, uni(504, 2, 3, 32, 32) * 0.5 + 0.5
    save("misc", enc_x0=x0, enc_z=zf, mlp_z=zt, mlp_t=tm, mlp_out=mo, m_a=a, m_b=b,
         ssim=calculate_ssim(a, b), mse=calculate_mse(a, b))


if __name__ == "__main__":
    main()
