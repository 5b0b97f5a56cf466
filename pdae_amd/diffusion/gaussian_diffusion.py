"""GaussianDiffusion: the reference's constructor, public attributes and method names (diffusion/gaussian_diffusion.py) over the
fused HIP kernels.

Structure (this file is written around three things, none of which the reference has):
  * `schedule_tables(betas)` -- ONE function that derives every per-timestep table from the beta sequence in float64; the
    constructor casts each to an fp32 device tensor under the attribute name the reference's callers read;
  * `respace(style, alphas_cumprod)` + a per-style cache of `DDIM` objects (the reference rebuilds and re-uploads the respaced
    tables on every sampling call, :187-188, 276-277);
  * per-sample coefficient gathers that stay on the device (`_rows`), feeding `pdae_axpby_rows` / `pdae_ddpm_step_rows`, so none of the
    single-step helpers reads `t` on the host and every one of them accepts a different timestep per sample like the reference does.
RNG draws (t, noise) happen in the reference's order and can be injected for parity tests.
"""
import math

import numpy as np
import torch

from . import ops
from .ddim import DDIM


def make_betas(betas_type, timesteps):
    """float64 beta sequence: 'linear' 1e-4 .. 0.02, or the cosine schedule of Nichol & Dhariwal capped at 0.999 (:17-29)."""
    if betas_type == "linear":
        return np.linspace(1e-4, 2e-2, timesteps, dtype=np.float64)
    if betas_type == "cosine":
        grid = np.arange(timesteps + 1, dtype=np.float64) / timesteps
        bar = [math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2 for u in grid]
        return np.array([min(1.0 - bar[k + 1] / bar[k], 0.999) for k in range(timesteps)], dtype=np.float64)
    raise NotImplementedError(betas_type)


def schedule_tables(betas):
    """name -> float64 table of length T for every schedule the diffusion math reads (:31-70)."""
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.concatenate(([1.0], ac[:-1]))
    ac_next = np.concatenate((ac[1:], [0.0]))
    rest = 1.0 - ac                                   # 1 - alpha_bar_t
    post_var = betas * (1.0 - ac_prev) / rest         # variance of q(x_{t-1} | x_t, x_0)
    snr = ac / rest
    return {
        "alphas": alphas, "betas": betas, "alphas_cumprod": ac, "alphas_cumprod_prev": ac_prev, "alphas_cumprod_next": ac_next,
        "sqrt_alphas_cumprod": np.sqrt(ac), "sqrt_one_minus_alphas_cumprod": np.sqrt(rest), "log_one_minus_alphas_cumprod": np.log(rest),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac), "sqrt_recip_alphas_cumprod_m1": np.sqrt(1.0 / ac - 1.0),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": np.log(np.concatenate((post_var[1:2], post_var[1:]))),     # entry 0 is 0: borrow entry 1
        "x_0_posterior_mean_x_0_coef": betas * np.sqrt(ac_prev) / rest,
        "x_0_posterior_mean_x_t_coef": (1.0 - ac_prev) * np.sqrt(alphas) / rest,
        "noise_posterior_mean_x_t_coef": np.sqrt(1.0 / alphas),
        "noise_posterior_mean_noise_coef": betas / (np.sqrt(alphas) * np.sqrt(rest)),
        "shift_coef": -np.sqrt(alphas) * (1.0 - ac_prev) / np.sqrt(rest),          # PDAE eq. for the mean shift (:65)
        "weight": snr ** 0.1 / (1.0 + snr),                                        # loss weight gamma = 0.1 (:68-70)
    }


def respace(ddim_style, alphas_cumprod):
    """'ddimN' -> (betas of the respaced chain, int64 map respaced step -> original step)  (:76-94).

    Kept steps = the distinct integer parts of linspace(0, T-1, N+1); beta'_k = 1 - ac[k] / ac[previous kept].  The arithmetic runs
    in the dtype of `alphas_cumprod` -- the reference hands over the fp32 table (`.cpu().numpy()`), and with NumPy >= 2 scalar
    promotion its Python loop stays in float32; the vector form below is element-for-element the same."""
    T = alphas_cumprod.shape[0]
    n = int(ddim_style[len("ddim"):])
    kept = np.unique(np.linspace(0, T - 1, n + 1).astype(np.int64))
    ac = np.asarray(alphas_cumprod)[kept]
    before = np.concatenate((np.ones(1, dtype=ac.dtype), ac[:-1]))
    return 1 - ac / before, torch.from_numpy(kept)


class GaussianDiffusion:
    def __init__(self, config, device):
        self.device = device
        self.timesteps = config["timesteps"]
        tables = schedule_tables(make_betas(config["betas_type"], self.timesteps))
        for name, tab in tables.items():
            setattr(self, name, torch.tensor(tab, dtype=torch.float32, device=device))
        self.to_torch = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
        self._ac_host = tables["alphas_cumprod"].astype(np.float32)      # what .cpu().numpy() of the fp32 table yields (:187)
        self._log_betas = torch.log(self.betas)                          # upper variance bound of learn_sigma models (:151)
        self._ddim_cache = {}
        self._row_tabs = {}
        self._latent_cfg = None

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def extract_coef_at_t(schedule, t, x_shape):
        return schedule.gather(-1, t).view(x_shape[0], *([1] * (len(x_shape) - 1)))

    @staticmethod
    def get_ddim_betas_and_timestep_map(ddim_style, original_alphas_cumprod):
        return respace(ddim_style, original_alphas_cumprod)

    def _ddim(self, ddim_style, alphas_cumprod_host=None):
        key = (ddim_style, id(alphas_cumprod_host) if alphas_cumprod_host is not None else 0)
        d = self._ddim_cache.get(key)
        if d is None:
            d = DDIM(*respace(ddim_style, self._ac_host if alphas_cumprod_host is None else alphas_cumprod_host), self.device)
            self._ddim_cache[key] = d
        return d

    def _rows(self, t, *names, neg=()):
        """One float32 device vector per named table, gathered at the per-sample t (negated for names in `neg`)."""
        out = []
        for nme in names:
            v = getattr(self, nme).gather(0, t)
            out.append(-v if nme in neg else v)
        return out

    # ------------------------------------------------------------------ forward process / posterior algebra (per-sample t)
    def q_sample(self, x_0, t, noise):                                                   # :98-103
        return ops.q_sample(x_0, noise, t, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod)

    def q_posterior_mean(self, x_0, x_t, t):                                             # :105-108
        c0, ct = self._rows(t, "x_0_posterior_mean_x_0_coef", "x_0_posterior_mean_x_t_coef")
        return ops.axpby_rows(x_0, x_t, c0, ct)

    def predicted_noise_to_predicted_x_0(self, x_t, t, predicted_noise):                 # :156-159
        ra, rm1 = self._rows(t, "sqrt_recip_alphas_cumprod", "sqrt_recip_alphas_cumprod_m1", neg=("sqrt_recip_alphas_cumprod_m1",))
        return ops.axpby_rows(x_t, predicted_noise, ra, rm1)

    def predicted_noise_to_predicted_mean(self, x_t, t, predicted_noise):                # :161-164
        cx, ce = self._rows(t, "noise_posterior_mean_x_t_coef", "noise_posterior_mean_noise_coef", neg=("noise_posterior_mean_noise_coef",))
        return ops.axpby_rows(x_t, predicted_noise, cx, ce)

    def learned_range_to_log_variance(self, learned_range, t):                           # :148-154
        """log-variance interpolated between the posterior (v = -1) and beta (v = +1) bounds: lo + (v+1)/2 * (hi - lo)."""
        lo = self.posterior_log_variance_clipped.gather(0, t)
        hi = self._log_betas.gather(0, t)
        half = 0.5 * (hi - lo)
        return ops.axpby_rows(learned_range, torch.ones_like(learned_range), half, lo + half)

    def noise_p_sample(self, x_t, t, predicted_noise, learned_range=None, gradient=None, noise=None):   # :112-126 (+ :268-269 with `gradient`)
        """x_{t-1} ~ p(. | x_t): mean from the predicted noise (shifted by shift_coef[t] * gradient when given), fixed-small or learned
        variance, no noise where t == 0.  One kernel; `noise` may be injected."""
        noise = torch.randn(x_t.shape, device=x_t.device) if noise is None else noise
        cx, ce, cs, lv0 = self._rows(t, "noise_posterior_mean_x_t_coef", "noise_posterior_mean_noise_coef", "shift_coef",
                                     "posterior_log_variance_clipped")
        coef = torch.stack([cx, ce, cs, (t != 0).to(torch.float32), lv0, self._log_betas.gather(0, t)], 1)
        return ops.ddpm_step_rows(x_t, predicted_noise, gradient, noise, learned_range, coef)

    def p_loss(self, noise, predicted_noise, weight=None, loss_type="l2"):               # :166-175
        if loss_type not in ("l1", "l2"):
            raise NotImplementedError(loss_type)
        if weight is None or loss_type == "l1":       # the reference ignores `weight` for l1 as well
            return ops.loss(noise, predicted_noise, l1=(loss_type == "l1"))
        n = noise.shape[0]
        if weight.numel() != n:
            raise NotImplementedError("p_loss: only per-sample weights (shape [N,1,...,1]) are supported by the fused loss kernel")
        idx = torch.arange(n, device=noise.device)
        return ops.loss(noise, predicted_noise, None, idx, None, weight.reshape(n).contiguous())

    # ------------------------------------------------------------------ DDIM front-ends (:181-195)
    def test_pretrained_dpms(self, ddim_style, denoise_fn, x_T, condition=None):
        return self.ddim_sample(ddim_style, denoise_fn, x_T, condition)

    def ddim_sample(self, ddim_style, denoise_fn, x_T, condition=None):
        return self._ddim(ddim_style).ddim_sample_loop(denoise_fn, x_T, condition)

    def ddim_encode(self, ddim_style, denoise_fn, x_0, condition=None):
        return self._ddim(ddim_style).ddim_encode_loop(denoise_fn, x_0, condition)

    # ------------------------------------------------------------------ plain DDPM (:199-229)
    def _draw(self, like, t, noise, timesteps=None):
        """(t, noise) in the reference's draw order: randint first, randn_like second -- unless injected."""
        if t is None:
            t = torch.randint(0, self.timesteps if timesteps is None else timesteps, (like.shape[0],), device=self.device, dtype=torch.long)
        if noise is None:
            noise = torch.randn_like(like)
        return t, noise

    def regular_train_one_batch(self, denoise_fn, x_0, condition=None, t=None, noise=None):
        t, noise = self._draw(x_0, t, noise)
        predicted = denoise_fn(self.q_sample(x_0, t, noise), t, condition)
        return {"prediction_loss": self.p_loss(noise, predicted)}

    def regular_ddim_sample(self, ddim_style, denoise_fn, x_T, condition=None):
        return self.ddim_sample(ddim_style, denoise_fn, x_T, condition)

    def regular_ddpm_sample(self, denoise_fn, x_T, condition=None, noises=None):
        """1000-step ancestral sampler; a learn_sigma model returns 2C channels = (eps, variance range).  `noises`: optional callable
        i -> noise tensor (parity tests)."""
        channels = x_T.shape[1]
        img = x_T
        for i in range(self.timesteps - 1, -1, -1):
            t = torch.full((x_T.shape[0],), i, device=self.device, dtype=torch.long)
            out = denoise_fn(img, t, condition)
            eps, vrange = (out[:, :channels], out[:, channels:]) if out.shape[1] == 2 * channels else (out, None)
            img = self.noise_p_sample(img, t, eps, vrange, noise=None if noises is None else noises(i))
        return img

    # ------------------------------------------------------------------ representation learning (:234-339)
    def representation_learning_train_one_batch(self, encoder, decoder, x_0, t=None, noise=None):
        z = encoder(x_0)
        t, noise = self._draw(x_0, t, noise)
        eps, grad = decoder(self.q_sample(x_0, t, noise), t, z)
        # mean(weight[t] * (noise - (eps + shift_coef[t] * grad))^2) and both gradients in one launch
        return {"prediction_loss": ops.loss(noise, eps, grad, t, self.shift_coef, self.weight)}

    def representation_learning_ddpm_sample(self, encoder, decoder, x_0, x_T, z=None, noises=None):
        z = encoder(x_0) if z is None else z
        img = x_T
        for i in range(self.timesteps - 1, -1, -1):
            t = torch.full((x_T.shape[0],), i, device=self.device, dtype=torch.long)
            eps, grad = decoder(img, t, z)
            img = self.noise_p_sample(img, t, eps, gradient=grad, noise=None if noises is None else noises(i))
        return img

    def representation_learning_ddim_sample(self, ddim_style, encoder, decoder, x_0, x_T, z=None, stop_percent=0.0):
        z = encoder(x_0) if z is None else z
        return self._ddim(ddim_style).shift_ddim_sample_loop(decoder, z, x_T, stop_percent=stop_percent)

    def representation_learning_ddim_encode(self, ddim_style, encoder, decoder, x_0, z=None):
        z = encoder(x_0) if z is None else z
        return self._ddim(ddim_style).shift_ddim_encode_loop(decoder, z, x_0)

    def representation_learning_autoencoding(self, encoder_ddim_style, decoder_ddim_style, encoder, decoder, x_0):
        z = encoder(x_0)
        x_T = self.representation_learning_ddim_encode(encoder_ddim_style, None, decoder, x_0, z)
        return self.representation_learning_ddim_sample(decoder_ddim_style, None, decoder, None, x_T, z)

    def _both_x_0(self, decoder, x_t, t, z):
        """(x_0 implied by the pre-trained eps, x_0 implied by the shifted eps + shift_coef * grad) for one decoder pass."""
        eps, grad = decoder(x_t, t, z)
        ones = torch.ones_like(t, dtype=torch.float32)
        shifted = ops.axpby_rows(eps, grad, ones, self.shift_coef.gather(0, t))
        return self.predicted_noise_to_predicted_x_0(x_t, t, eps), self.predicted_noise_to_predicted_x_0(x_t, t, shifted)

    def representation_learning_gap_measure(self, encoder, decoder, x_0, noises=None):
        """Per timestep (T-1 .. 0): MSE between the true posterior mean and the one implied by (a) the pre-trained eps, (b) the shifted
        eps (:292-318).  The perturbation is UNIFORM noise (`rand_like`, :302), as in the reference.  Values stay on the device until the end
        (one host sync instead of 2000)."""
        z = encoder(x_0)
        gaps = []
        for i in range(self.timesteps - 1, -1, -1):
            t = torch.full((x_0.shape[0],), i, device=self.device, dtype=torch.long)
            x_t = self.q_sample(x_0, t, torch.rand_like(x_0) if noises is None else noises(i))
            x0_dpm, x0_ae = self._both_x_0(decoder, x_t, t, z)
            truth = self.q_posterior_mean(x_0, x_t, t)
            gaps.append(torch.stack([ops.loss(truth, self.q_posterior_mean(x0_dpm, x_t, t)),
                                     ops.loss(truth, self.q_posterior_mean(x0_ae, x_t, t))]))
        g = torch.stack(gaps).tolist()
        return [a for a, _ in g], [b for _, b in g]

    def representation_learning_denoise_one_step(self, encoder, decoder, x_0, timestep_list, noise=None):
        """x_0 estimates after ONE denoising step from per-sample timesteps (:320-334)."""
        t = torch.as_tensor(timestep_list, device=self.device, dtype=torch.long)
        x_t = self.q_sample(x_0, t, torch.randn_like(x_0) if noise is None else noise)
        return self._both_x_0(decoder, x_t, t, encoder(x_0))

    def representation_learning_ddim_trajectory_interpolation(self, ddim_style, decoder, z_1, z_2, x_T, alpha):
        return self._ddim(ddim_style).shift_ddim_trajectory_interpolation(decoder, z_1, z_2, x_T, alpha)

    # ------------------------------------------------------------------ latent DPM (:344-415)
    @property
    def latent_diffusion_config(self):
        if self._latent_cfg is None:
            T = 1000
            betas = np.full(T, 0.008, dtype=np.float64)
            tabs = schedule_tables(betas)
            dev = lambda a: torch.tensor(a, dtype=torch.float32, device=self.device)
            self._latent_cfg = {"timesteps": T, "betas": betas, "alphas_cumprod": dev(tabs["alphas_cumprod"]),
                                "sqrt_alphas_cumprod": dev(tabs["sqrt_alphas_cumprod"]),
                                "sqrt_one_minus_alphas_cumprod": dev(tabs["sqrt_one_minus_alphas_cumprod"]), "loss_type": "l1",
                                "_ac_host": tabs["alphas_cumprod"].astype(np.float32)}
        return self._latent_cfg

    @staticmethod
    def normalize(z, mean, std):
        return (z - mean) / std

    @staticmethod
    def denormalize(z, mean, std):
        return z * std + mean

    def latent_diffusion_train_one_batch(self, latent_denoise_fn, encoder, x_0, latents_mean, latents_std, t=None, noise=None):
        cfg = self.latent_diffusion_config
        with torch.no_grad():
            z_0 = self.normalize(encoder(x_0).detach(), latents_mean, latents_std)
        t, noise = self._draw(z_0, t, noise, cfg["timesteps"])
        z_t = ops.q_sample(z_0, noise, t, cfg["sqrt_alphas_cumprod"], cfg["sqrt_one_minus_alphas_cumprod"])
        return {"prediction_loss": self.p_loss(noise, latent_denoise_fn(z_t, t), loss_type=cfg["loss_type"])}

    def latent_diffusion_sample(self, latent_ddim_style, decoder_ddim_style, latent_denoise_fn, decoder, x_T, latents_mean, latents_std, z_T=None):
        """z_T ~ N(0,I) clamped to [-1,1] (:407) -> latent DDIM -> de-normalise -> image DDIM with the shift switched off for the last 30 %
        of the steps (:415).  `z_T` may be injected (already clamped or not)."""
        cfg = self.latent_diffusion_config
        if z_T is None:
            z_T = torch.randn((x_T.shape[0], latent_denoise_fn.input_channel), device=self.device)
        z = self._ddim(latent_ddim_style, cfg["_ac_host"]).latent_ddim_sample_loop(latent_denoise_fn, z_T.clamp(-1.0, 1.0))
        return self.representation_learning_ddim_sample(decoder_ddim_style, None, decoder, None, x_T, self.denormalize(z, latents_mean, latents_std),
                                                        stop_percent=0.3)

    # ------------------------------------------------------------------ manipulation (:435-443; the classifier trainer is out of scope)
    def manipulation_sample(self, ddim_style, classifier_weight, encoder, decoder, x_0, inferred_x_T, latents_mean, latents_std, class_id, scale):
        """Moves the normalised latent along the unit direction of one linear-classifier row by scale * sqrt(latent_dim=512), then decodes."""
        direction = classifier_weight[class_id]
        direction = direction / direction.norm().clamp_min(1e-12)
        z = self.normalize(encoder(x_0), latents_mean, latents_std) + scale * math.sqrt(512) * direction[None, :]
        return self.representation_learning_ddim_sample(ddim_style, None, decoder, None, inferred_x_T, self.denormalize(z, latents_mean, latents_std),
                                                        stop_percent=0.0)
