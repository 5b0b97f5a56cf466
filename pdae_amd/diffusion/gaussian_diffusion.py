"""GaussianDiffusion with the reference's constructor, attributes and method names
(diffusion/gaussian_diffusion.py), running its elementwise math through the fused HIP kernels.

Schedules are built in float64 numpy and cast to fp32 device tensors exactly as the reference does (:17-70)."""
import math
from functools import partial

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .ddim import DDIM


class GaussianDiffusion:
    def __init__(self, config, device):
        self.device = device
        self.timesteps = config["timesteps"]
        betas_type = config["betas_type"]
        if betas_type == "linear":
            betas = np.linspace(0.0001, 0.02, self.timesteps)
        elif betas_type == "cosine":
            alpha_bar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
            betas = np.array([min(1 - alpha_bar((i + 1) / self.timesteps) / alpha_bar(i / self.timesteps), 0.999)
                              for i in range(self.timesteps)])
        else:
            raise NotImplementedError

        alphas = 1. - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1., alphas_cumprod[:-1])
        alphas_cumprod_next = np.append(alphas_cumprod[1:], 0.)
        to_torch = partial(torch.tensor, dtype=torch.float32, device=self.device)
        self.to_torch = to_torch
        self.alphas, self.betas = to_torch(alphas), to_torch(betas)
        self.alphas_cumprod = to_torch(alphas_cumprod)
        self.alphas_cumprod_prev = to_torch(alphas_cumprod_prev)
        self.alphas_cumprod_next = to_torch(alphas_cumprod_next)
        self.sqrt_alphas_cumprod = to_torch(np.sqrt(alphas_cumprod))
        self.sqrt_one_minus_alphas_cumprod = to_torch(np.sqrt(1. - alphas_cumprod))
        self.log_one_minus_alphas_cumprod = to_torch(np.log(1. - alphas_cumprod))
        self.sqrt_recip_alphas_cumprod = to_torch(np.sqrt(1. / alphas_cumprod))
        self.sqrt_recip_alphas_cumprod_m1 = to_torch(np.sqrt(1. / alphas_cumprod - 1.))
        posterior_variance = betas * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod)
        self.posterior_variance = to_torch(posterior_variance)
        self.posterior_log_variance_clipped = to_torch(np.log(np.append(posterior_variance[1], posterior_variance[1:])))
        self.x_0_posterior_mean_x_0_coef = to_torch(betas * np.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod))
        self.x_0_posterior_mean_x_t_coef = to_torch((1. - alphas_cumprod_prev) * np.sqrt(alphas) / (1. - alphas_cumprod))
        self.noise_posterior_mean_x_t_coef = to_torch(np.sqrt(1. / alphas))
        self.noise_posterior_mean_noise_coef = to_torch(betas / (np.sqrt(alphas) * np.sqrt(1. - alphas_cumprod)))
        self.shift_coef = to_torch(-np.sqrt(alphas) * (1. - alphas_cumprod_prev) / np.sqrt(1. - alphas_cumprod))
        snr = alphas_cumprod / (1. - alphas_cumprod)
        self.weight = to_torch(snr ** 0.1 / (1. + snr))
        self._ac_host = np.asarray(alphas_cumprod, dtype=np.float32)      # what .cpu().numpy() of the fp32 table yields (:187)
        self._ddim_cache = {}

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def extract_coef_at_t(schedule, t, x_shape):
        return torch.gather(schedule, -1, t).reshape([x_shape[0]] + [1] * (len(x_shape) - 1))

    @staticmethod
    def get_ddim_betas_and_timestep_map(ddim_style, original_alphas_cumprod):           # :76-94
        original_timesteps = original_alphas_cumprod.shape[0]
        ddim_step = int(ddim_style[len("ddim"):])
        use_timesteps = set([int(s) for s in list(np.linspace(0, original_timesteps - 1, ddim_step + 1))])
        timestep_map, new_betas, last = [], [], 1.0
        for i, alpha_cumprod in enumerate(original_alphas_cumprod):
            if i in use_timesteps:
                new_betas.append(1 - alpha_cumprod / last)
                last = alpha_cumprod
                timestep_map.append(i)
        return np.array(new_betas), torch.tensor(timestep_map, dtype=torch.long)

    def _ddim(self, ddim_style, alphas_cumprod_host=None):
        """The reference rebuilds the respaced tables (D2H + H2D) on every sampling call (:187-188,276-277);
        they only depend on the style, so they are built once and cached."""
        key = (ddim_style, id(alphas_cumprod_host) if alphas_cumprod_host is not None else 0)
        d = self._ddim_cache.get(key)
        if d is None:
            ac = self._ac_host if alphas_cumprod_host is None else alphas_cumprod_host
            new_betas, timestep_map = self.get_ddim_betas_and_timestep_map(ddim_style, ac)
            d = DDIM(new_betas, timestep_map, self.device)
            self._ddim_cache[key] = d
        return d

    def q_sample(self, x_0, t, noise):                                                   # :98-103
        return ops.q_sample(x_0, noise, t, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod)

    def q_posterior_mean(self, x_0, x_t, t):                                             # :105-108
        shape = x_t.shape
        return self.extract_coef_at_t(self.x_0_posterior_mean_x_0_coef, t, shape) * x_0 \
            + self.extract_coef_at_t(self.x_0_posterior_mean_x_t_coef, t, shape) * x_t

    def learned_range_to_log_variance(self, learned_range, t):                           # :148-154
        shape = learned_range.shape
        min_lv = self.extract_coef_at_t(self.posterior_log_variance_clipped, t, shape)
        max_lv = self.extract_coef_at_t(torch.log(self.betas), t, shape)
        return min_lv + (learned_range + 1) / 2 * (max_lv - min_lv)

    def noise_p_sample(self, x_t, t, predicted_noise, learned_range=None, gradient=None, noise=None):   # :112-126
        """All samples of a DDPM loop share t, so the coefficients are scalars of one fused kernel."""
        i = int(t.reshape(-1)[0].item())
        shape = x_t.shape
        noise = torch.randn(shape, device=self.device) if noise is None else noise
        if learned_range is not None:
            lv = self.learned_range_to_log_variance(learned_range, t)
            mean = ops.ddpm_step(x_t, predicted_noise, gradient, None, float(self.noise_posterior_mean_x_t_coef[i]),
                                 float(self.noise_posterior_mean_noise_coef[i]), float(self.shift_coef[i]), 0.0)
            return mean + (0.0 if i == 0 else 1.0) * (0.5 * lv).exp() * noise
        sigma = 0.0 if i == 0 else math.exp(0.5 * float(self.posterior_log_variance_clipped[i]))
        return ops.ddpm_step(x_t, predicted_noise, gradient, noise, float(self.noise_posterior_mean_x_t_coef[i]),
                             float(self.noise_posterior_mean_noise_coef[i]), float(self.shift_coef[i]), sigma)

    def predicted_noise_to_predicted_x_0(self, x_t, t, predicted_noise):                 # :156-159
        shape = x_t.shape
        return self.extract_coef_at_t(self.sqrt_recip_alphas_cumprod, t, shape) * x_t \
            - self.extract_coef_at_t(self.sqrt_recip_alphas_cumprod_m1, t, shape) * predicted_noise

    def predicted_noise_to_predicted_mean(self, x_t, t, predicted_noise):                # :161-164
        shape = x_t.shape
        return self.extract_coef_at_t(self.noise_posterior_mean_x_t_coef, t, shape) * x_t - \
            self.extract_coef_at_t(self.noise_posterior_mean_noise_coef, t, shape) * predicted_noise

    def p_loss(self, noise, predicted_noise, weight=None, loss_type="l2"):               # :166-175
        if loss_type not in ("l1", "l2"):
            raise NotImplementedError
        if weight is not None:       # generic broadcast weight (the fused RL loss below never takes this path)
            return torch.mean(weight * (noise - predicted_noise) ** 2)
        return ops.loss(noise, predicted_noise, l1=(loss_type == "l1"))

    # ------------------------------------------------------------------ ddim front-ends
    def test_pretrained_dpms(self, ddim_style, denoise_fn, x_T, condition=None):
        return self.ddim_sample(ddim_style, denoise_fn, x_T, condition)

    def ddim_sample(self, ddim_style, denoise_fn, x_T, condition=None):
        return self._ddim(ddim_style).ddim_sample_loop(denoise_fn, x_T, condition)

    def ddim_encode(self, ddim_style, denoise_fn, x_0, condition=None):
        return self._ddim(ddim_style).ddim_encode_loop(denoise_fn, x_0, condition)

    # ------------------------------------------------------------------ regular
    def regular_train_one_batch(self, denoise_fn, x_0, condition=None, t=None, noise=None):   # :199-211
        batch_size = x_0.shape[0]
        t = torch.randint(0, self.timesteps, (batch_size,), device=self.device, dtype=torch.long) if t is None else t
        noise = torch.randn_like(x_0) if noise is None else noise
        x_t = self.q_sample(x_0=x_0, t=t, noise=noise)
        predicted_noise = denoise_fn(x_t, t, condition)
        return {'prediction_loss': self.p_loss(noise, predicted_noise)}

    def regular_ddim_sample(self, ddim_style, denoise_fn, x_T, condition=None):
        return self.ddim_sample(ddim_style, denoise_fn, x_T, condition)

    def regular_ddpm_sample(self, denoise_fn, x_T, condition=None):                      # :216-229
        shape = x_T.shape
        img = x_T
        for i in reversed(range(0, self.timesteps)):
            t = torch.full((shape[0],), i, device=self.device, dtype=torch.long)
            output = denoise_fn(img, t, condition)
            if output.shape[1] == 2 * shape[1]:
                predicted_noise, learned_range = torch.split(output, shape[1], dim=1)
            else:
                predicted_noise, learned_range = output, None
            img = self.noise_p_sample(img, t, predicted_noise, learned_range)
        return img

    # ------------------------------------------------------------------ representation learning
    def representation_learning_train_one_batch(self, encoder, decoder, x_0, t=None, noise=None):   # :234-255
        """(t, noise) are drawn like the reference (randint, then randn_like) unless injected."""
        batch_size = x_0.shape[0]
        z = encoder(x_0)
        t = torch.randint(0, self.timesteps, (batch_size,), device=self.device, dtype=torch.long) if t is None else t
        noise = torch.randn_like(x_0) if noise is None else noise
        x_t = self.q_sample(x_0=x_0, t=t, noise=noise)
        predicted_noise, gradient = decoder(x_t, t, z)
        # fused: mean(weight[t] * (noise - (eps + shift_coef[t] * gradient))^2) and both gradients in one launch
        prediction_loss = ops.loss(noise, predicted_noise, gradient, t, self.shift_coef, self.weight)
        return {'prediction_loss': prediction_loss}

    def representation_learning_ddpm_sample(self, encoder, decoder, x_0, x_T, z=None):   # :257-270
        if z is None:
            z = encoder(x_0)
        img = x_T
        for i in reversed(range(0, self.timesteps)):
            t = torch.full((x_T.shape[0],), i, device=self.device, dtype=torch.long)
            predicted_noise, gradient = decoder(img, t, z)
            img = self.noise_p_sample(img, t, predicted_noise, gradient=gradient)
        return img

    def representation_learning_ddim_sample(self, ddim_style, encoder, decoder, x_0, x_T, z=None, stop_percent=0.0):
        if z is None:
            z = encoder(x_0)
        return self._ddim(ddim_style).shift_ddim_sample_loop(decoder, z, x_T, stop_percent=stop_percent)

    def representation_learning_ddim_encode(self, ddim_style, encoder, decoder, x_0, z=None):
        if z is None:
            z = encoder(x_0)
        return self._ddim(ddim_style).shift_ddim_encode_loop(decoder, z, x_0)

    def representation_learning_autoencoding(self, encoder_ddim_style, decoder_ddim_style, encoder, decoder, x_0):
        z = encoder(x_0)
        inferred_x_T = self.representation_learning_ddim_encode(encoder_ddim_style, encoder, decoder, x_0, z)
        return self.representation_learning_ddim_sample(decoder_ddim_style, None, decoder, None, inferred_x_T, z)

    def representation_learning_gap_measure(self, encoder, decoder, x_0):                # :292-318
        shape = x_0.shape
        z = encoder(x_0)
        gap_p, gap_a = [], []
        for i in reversed(range(0, self.timesteps)):
            t = torch.full((shape[0],), i, device=self.device, dtype=torch.long)
            x_t = self.q_sample(x_0, t, torch.rand_like(x_0))
            predicted_noise, gradient = decoder(x_t, t, z)
            pm = self.q_posterior_mean(self.predicted_noise_to_predicted_x_0(x_t, t, predicted_noise), x_t, t)
            ae_noise = predicted_noise + self.extract_coef_at_t(self.shift_coef, t, shape) * gradient
            am = self.q_posterior_mean(self.predicted_noise_to_predicted_x_0(x_t, t, ae_noise), x_t, t)
            tm = self.q_posterior_mean(x_0, x_t, t)
            gap_p.append(torch.mean((tm - pm) ** 2).cpu().item())
            gap_a.append(torch.mean((tm - am) ** 2).cpu().item())
        return gap_p, gap_a

    def representation_learning_denoise_one_step(self, encoder, decoder, x_0, timestep_list):   # :320-334
        shape = x_0.shape
        t = torch.tensor(timestep_list, device=self.device, dtype=torch.long)
        x_t = self.q_sample(x_0, t, noise=torch.randn_like(x_0))
        z = encoder(x_0)
        predicted_noise, gradient = decoder(x_t, t, z)
        predicted_x_0 = self.predicted_noise_to_predicted_x_0(x_t, t, predicted_noise)
        ae_noise = predicted_noise + self.extract_coef_at_t(self.shift_coef, t, shape) * gradient
        return predicted_x_0, self.predicted_noise_to_predicted_x_0(x_t, t, ae_noise)

    def representation_learning_ddim_trajectory_interpolation(self, ddim_style, decoder, z_1, z_2, x_T, alpha):
        return self._ddim(ddim_style).shift_ddim_trajectory_interpolation(decoder, z_1, z_2, x_T, alpha)

    # ------------------------------------------------------------------ latent DPM (:344-415)
    @property
    def latent_diffusion_config(self):
        if not hasattr(self, "_latent_cfg"):
            timesteps = 1000
            betas = np.array([0.008] * timesteps)
            alphas_cumprod = np.cumprod(1. - betas, axis=0)
            to_torch = partial(torch.tensor, dtype=torch.float32, device=self.device)
            self._latent_cfg = {
                "timesteps": timesteps, "betas": betas, "alphas_cumprod": to_torch(alphas_cumprod),
                "sqrt_alphas_cumprod": to_torch(np.sqrt(alphas_cumprod)),
                "sqrt_one_minus_alphas_cumprod": to_torch(np.sqrt(1. - alphas_cumprod)), "loss_type": "l1",
                "_ac_host": np.asarray(alphas_cumprod, dtype=np.float32),
            }
        return self._latent_cfg

    def normalize(self, z, mean, std):
        return (z - mean) / std

    def denormalize(self, z, mean, std):
        return z * std + mean

    def latent_diffusion_train_one_batch(self, latent_denoise_fn, encoder, x_0, latents_mean, latents_std, t=None, noise=None):
        cfg = self.latent_diffusion_config
        with torch.no_grad():
            z_0 = encoder(x_0)
        z_0 = self.normalize(z_0.detach(), latents_mean, latents_std)
        batch_size = z_0.shape[0]
        t = torch.randint(0, cfg["timesteps"], (batch_size,), device=self.device, dtype=torch.long) if t is None else t
        noise = torch.randn_like(z_0) if noise is None else noise
        z_t = ops.q_sample(z_0, noise, t, cfg["sqrt_alphas_cumprod"], cfg["sqrt_one_minus_alphas_cumprod"])
        predicted_noise = latent_denoise_fn(z_t, t)
        return {'prediction_loss': self.p_loss(noise, predicted_noise, loss_type=cfg["loss_type"])}

    def latent_diffusion_sample(self, latent_ddim_style, decoder_ddim_style, latent_denoise_fn, decoder, x_T, latents_mean, latents_std):
        cfg = self.latent_diffusion_config
        z_T = torch.randn((x_T.shape[0], latent_denoise_fn.input_channel), device=self.device)
        z_T.clamp_(-1.0, 1.0)
        z = self._ddim(latent_ddim_style, cfg["_ac_host"]).latent_ddim_sample_loop(latent_denoise_fn, z_T)
        z = self.denormalize(z, latents_mean, latents_std)
        return self.representation_learning_ddim_sample(decoder_ddim_style, None, decoder, None, x_T, z, stop_percent=0.3)

    # ------------------------------------------------------------------ manipulation (:422-443)
    def manipulation_train_one_batch(self, classifier, encoder, x_0, label, latents_mean, latents_std):
        with torch.no_grad():
            z = encoder(x_0)
        prediction = classifier(self.normalize(z.detach(), latents_mean, latents_std))
        gt = torch.where(label > 0, torch.ones_like(label).float(), torch.zeros_like(label).float())
        return {'bce_loss': F.binary_cross_entropy_with_logits(prediction, gt)}

    def manipulation_sample(self, ddim_style, classifier_weight, encoder, decoder, x_0, inferred_x_T, latents_mean, latents_std, class_id, scale):
        z_norm = self.normalize(encoder(x_0), latents_mean, latents_std)
        z_norm = z_norm + scale * math.sqrt(512) * F.normalize(classifier_weight[class_id][None, :], dim=1)
        z = self.denormalize(z_norm, latents_mean, latents_std)
        return self.representation_learning_ddim_sample(ddim_style, None, decoder, None, inferred_x_T, z, stop_percent=0.0)
