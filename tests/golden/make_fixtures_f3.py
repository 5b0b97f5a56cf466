"""Generates tests/golden/f3.npz by IMPORTING THE REFERENCE (ckczzj/PDAE) on CPU -- the remaining GaussianDiffusion / DDIM front-ends:

    noise_p_sample (fixed and LEARNED variance, per-sample t), q_posterior_mean, predicted_noise_to_predicted_x_0 / _mean,
    learned_range_to_log_variance, regular_ddpm_sample with a learn_sigma UNet, representation_learning_ddpm_sample,
    representation_learning_gap_measure (uniform noise!), representation_learning_denoise_one_step,
    shift_ddim_trajectory_interpolation, manipulation_sample.

Runs only in the build container:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures_f3.py
The reference draws its noise internally (torch.randn / randn_like / rand_like); the draws are replaced, for the duration of each call, by
`make_fixtures_cfg.f3_noise(stream, call index)` so that the GPU tests can inject the same values.  Only inputs / expected outputs are saved.
"""
import os
import sys
from contextlib import contextmanager

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np
import torch

torch.set_num_threads(8)

from model.unet import UNet                                     # reference
from model.shift_unet import ShiftUNet                          # reference
from diffusion.gaussian_diffusion import GaussianDiffusion      # reference

from oracle import pdae_oracle as O                             # shapes + synthetic weights only
from tests.golden.make_fixtures_cfg import CFG_SHIFT_T, CFG_UNET_SIGMA, F3_T, f3_noise
from tests.golden.make_fixtures import load_synth, rnd, uni, save


@contextmanager
def injected(stream, T, descending=True):
    """torch.randn / randn_like / rand_like return f3_noise(stream, i): i counts T-1, T-2, ... (the samplers' loop index) or 0, 1, ..."""
    state = {"k": 0}

    def idx():
        k = state["k"]
        state["k"] += 1
        return (T - 1 - k) if descending else k

    saved = (torch.randn, torch.randn_like, torch.rand_like)
    torch.randn = lambda *shape, **kw: torch.from_numpy(f3_noise(stream, idx(), tuple(shape[0]) if isinstance(shape[0], (tuple, list, torch.Size)) else shape))
    torch.randn_like = lambda x, **kw: torch.from_numpy(f3_noise(stream, idx(), tuple(x.shape)))
    torch.rand_like = lambda x, **kw: torch.from_numpy(f3_noise(stream, idx(), tuple(x.shape), uniform=True))
    try:
        yield
    finally:
        torch.randn, torch.randn_like, torch.rand_like = saved


def main():
    dev = torch.device("cpu")
    gd = GaussianDiffusion({"timesteps": F3_T, "betas_type": "linear"}, dev)
    out = {}
    B, S, latent = 2, 16, 64

    # ---- posterior algebra with a different timestep per sample
    x_t, x_0, eps = rnd(601, 3, 3, S, S), uni(602, 3, 3, S, S), rnd(603, 3, 3, S, S)
    vr = uni(604, 3, 3, S, S)
    t3 = torch.tensor([0, 37, F3_T - 1])
    out.update(a_x_t=x_t, a_x_0=x_0, a_eps=eps, a_vr=vr, a_t=t3,
               a_post_mean=gd.q_posterior_mean(x_0, x_t, t3), a_pred_x0=gd.predicted_noise_to_predicted_x_0(x_t, t3, eps),
               a_pred_mean=gd.predicted_noise_to_predicted_mean(x_t, t3, eps), a_logvar=gd.learned_range_to_log_variance(vr, t3))
    with injected(0, 1, descending=False):
        out["a_step_fixed"] = gd.noise_p_sample(x_t, t3, eps)
    with injected(0, 1, descending=False):
        out["a_step_learned"] = gd.noise_p_sample(x_t, t3, eps, vr)

    # ---- regular_ddpm_sample with a learn_sigma UNet (2C output channels)
    net = UNet(**CFG_UNET_SIGMA).eval()
    load_synth(net, O.unet_param_shapes(CFG_UNET_SIGMA), 61)
    xT = rnd(611, B, 3, S, S)
    with torch.no_grad(), injected(1, F3_T):
        out["b_x_T"], out["b_sample"] = xT, gd.regular_ddpm_sample(net, xT)

    # ---- representation-learning front-ends on the tiny ShiftUNet; the encoder is a stand-in returning a fixed code
    dec = ShiftUNet(latent_dim=latent, **CFG_SHIFT_T).eval()
    load_synth(dec, O.unet_param_shapes(CFG_SHIFT_T, shift=True, latent_dim=latent), 62)
    z, z2 = rnd(621, B, latent), rnd(622, B, latent)
    x0 = uni(623, B, 3, S, S)
    enc = lambda x: z
    with torch.no_grad():
        with injected(2, F3_T):
            out["c_ddpm"] = gd.representation_learning_ddpm_sample(enc, dec, x0, xT)
        with injected(3, F3_T):
            gp, ga = gd.representation_learning_gap_measure(enc, dec, x0)
        tl = [5, F3_T - 3]
        with injected(4, 1, descending=False):
            p0, a0 = gd.representation_learning_denoise_one_step(enc, dec, x0, tl)
        traj = gd.representation_learning_ddim_trajectory_interpolation("ddim10", dec, z, z2, xT, 0.3)
        cw = rnd(631, 5, latent)
        mean, std = rnd(632, latent) * 0.1, 1.0 + 0.1 * rnd(633, latent).abs()
        # the reference hard-codes sqrt(512) and torch F.normalize; latent size here is 64 -- the formula does not depend on it
        man = gd.manipulation_sample("ddim10", cw, enc, dec, x0, xT, mean, std, 3, 0.25)
    out.update(c_z=z, c_z2=z2, c_x0=x0, c_gap_p=np.array(gp), c_gap_a=np.array(ga), c_tl=np.array(tl), c_one_p=p0, c_one_a=a0,
               c_traj=traj, c_cw=cw, c_mean=mean, c_std=std, c_man=man, seed_unet=61, seed_dec=62, latent=latent)
    save("f3", **out)


if __name__ == "__main__":
    main()
