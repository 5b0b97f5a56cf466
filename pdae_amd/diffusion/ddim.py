"""Respaced DDIM (eta = 0) sample / encode loops -- same class and method names as diffusion/ddim.py.

Each step is ONE planned decoder forward followed by ONE fused update kernel (pdae_ddim_step) instead of the
reference's ~15 elementwise launches and 8 gathers per step (ddim.py:94-107).  When the model is a planned
network of this package the loop runs on the plan's static NHWC buffers with no per-step allocation; any other
callable `fn(x, t, cond)` goes through the generic path with identical arithmetic.
"""
from functools import partial

import numpy as np
import torch

from . import ops


class DDIM:
    def __init__(self, betas, timestep_map, device):                       # ddim.py:8-33
        self.device = device
        self.timestep_map = timestep_map.to(self.device)
        self._map_host = [int(v) for v in timestep_map.tolist()]
        self.timesteps = betas.shape[0] - 1
        alphas = 1. - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1., alphas_cumprod[:-1])
        alphas_cumprod_next = np.append(alphas_cumprod[1:], 0.)
        f32 = lambda a: np.asarray(a, dtype=np.float32)
        to_torch = partial(torch.tensor, dtype=torch.float32, device=self.device)
        self._h = dict(prev=f32(alphas_cumprod_prev), next=f32(alphas_cumprod_next), s1m=f32(np.sqrt(1. - alphas_cumprod)),
                       ra=f32(np.sqrt(1. / alphas_cumprod)), rm1=f32(np.sqrt(1. / alphas_cumprod - 1.)))
        self.alphas_cumprod_prev = to_torch(alphas_cumprod_prev)
        self.alphas_cumprod_next = to_torch(alphas_cumprod_next)
        self.sqrt_one_minus_alphas_cumprod = to_torch(np.sqrt(1. - alphas_cumprod))
        self.sqrt_recip_alphas_cumprod = to_torch(np.sqrt(1. / alphas_cumprod))
        self.sqrt_recip_alphas_cumprod_m1 = to_torch(np.sqrt(1. / alphas_cumprod - 1.))

    @staticmethod
    def extract_coef_at_t(schedule, t, x_shape):
        return torch.gather(schedule, -1, t).reshape([x_shape[0]] + [1] * (len(x_shape) - 1))

    def t_transform(self, t):
        return self.timestep_map[t]

    # ---- scalar coefficients of step i (all samples of a DDIM loop share the step index)
    def _coefs(self, i, encode):
        h = self._h
        ab = h["next"][i] if encode else h["prev"][i]
        sab = np.sqrt(ab, dtype=np.float32)
        s1ab = np.sqrt(np.float32(1.0) - ab, dtype=np.float32)
        return float(h["s1m"][i]), float(h["ra"][i]), float(h["rm1"][i]), float(sab), float(s1ab)

    def _update(self, x_t, i, eps, grad, encode, out=None):
        c_shift, ra, rm1, sab, s1ab = self._coefs(i, encode)
        return ops.ddim_step(x_t, eps, grad, c_shift, ra, rm1, sab, s1ab, out=out)

    @staticmethod
    def _step_index(t):
        return int(t.reshape(-1)[0].item())

    # ---- single steps (reference signatures)
    def ddim_sample(self, denoise_fn, x_t, t, condition=None):                         # ddim.py:43-55
        return self._update(x_t, self._step_index(t), denoise_fn(x_t, self.t_transform(t), condition), None, False)

    def ddim_encode(self, denoise_fn, x_t, t, condition=None):                         # ddim.py:66-79
        return self._update(x_t, self._step_index(t), denoise_fn(x_t, self.t_transform(t), condition), None, True)

    def shift_ddim_sample(self, decoder, z, x_t, t, use_shift=True):                   # ddim.py:91-107
        eps, grad = decoder(x_t, self.t_transform(t), z)
        return self._update(x_t, self._step_index(t), eps, grad if use_shift else None, False)

    def shift_ddim_encode(self, decoder, z, x_t, t):                                   # ddim.py:123-138
        eps, grad = decoder(x_t, self.t_transform(t), z)
        return self._update(x_t, self._step_index(t), eps, grad, True)

    # ---- loops
    def _planned_loop(self, net, x_start, steps, encode, z=None, condition=None, use_shift=lambda i: True, trajectory=None):
        """Runs on the network's inference plan: static buffers, two host calls per step."""
        N, _, Hh, W = x_start.shape
        p = net.plan(N, Hh, W, False)
        p.x.copy_(x_start.permute(0, 2, 3, 1))
        if z is not None:
            p.z.copy_(z)
        if condition is not None and getattr(p, "cond", None) is not None:
            p.cond.copy_(condition)
        shift = getattr(p, "shift", None)
        for i in steps:
            p.t.fill_(self._map_host[i])
            p.run(0, p.n_fwd)
            c_shift, ra, rm1, sab, s1ab = self._coefs(i, encode)
            g = shift if (shift is not None and use_shift(i)) else None
            ops.ddim_step(p.x, p.eps, g, c_shift, ra, rm1, sab, s1ab, out=p.x)
            if trajectory is not None:
                trajectory.append(p.x.clone().permute(0, 3, 1, 2))
        return p.x.clone().permute(0, 3, 1, 2)

    @staticmethod
    def _planned(net):
        return hasattr(net, "plan") and hasattr(net, "P") and not torch.is_grad_enabled()

    def ddim_sample_loop(self, denoise_fn, x_T, condition=None):                        # ddim.py:57-64
        steps = list(reversed(range(1, self.timesteps + 1)))
        if self._planned(denoise_fn) and x_T.dim() == 4:
            return self._planned_loop(denoise_fn, x_T, steps, False, condition=condition)
        img = x_T
        for i in steps:
            t = torch.full((x_T.shape[0],), i, device=self.device, dtype=torch.long)
            img = self.ddim_sample(denoise_fn, img, t, condition)
        return img

    def ddim_encode_loop(self, denoise_fn, x_0, condition=None):                        # ddim.py:81-88
        steps = list(range(0, self.timesteps))
        if self._planned(denoise_fn) and x_0.dim() == 4:
            return self._planned_loop(denoise_fn, x_0, steps, True, condition=condition)
        x_t = x_0
        for i in steps:
            t = torch.full((x_0.shape[0],), i, device=self.device, dtype=torch.long)
            x_t = self.ddim_encode(denoise_fn, x_t, t, condition)
        return x_t

    def shift_ddim_sample_loop(self, decoder, z, x_T, stop_percent=0.0, trajectory=None):   # ddim.py:110-120
        stop_step = int(stop_percent * self.timesteps)
        steps = list(reversed(range(1, self.timesteps + 1)))
        use = lambda i: (i - 1) >= stop_step
        if self._planned(decoder):
            return self._planned_loop(decoder, x_T, steps, False, z=z, use_shift=use, trajectory=trajectory)
        img = x_T
        for i in steps:
            t = torch.full((x_T.shape[0],), i, device=self.device, dtype=torch.long)
            img = self.shift_ddim_sample(decoder, z, img, t, use_shift=use(i))
        return img

    def shift_ddim_encode_loop(self, decoder, z, x_0, trajectory=None):                 # ddim.py:140-147
        steps = list(range(0, self.timesteps))
        if self._planned(decoder):
            return self._planned_loop(decoder, x_0, steps, True, z=z, trajectory=trajectory)
        x_t = x_0
        for i in steps:
            t = torch.full((x_0.shape[0],), i, device=self.device, dtype=torch.long)
            x_t = self.shift_ddim_encode(decoder, z, x_t, t)
        return x_t

    def shift_ddim_trajectory_interpolation(self, decoder, z_1, z_2, x_T, alpha):       # ddim.py:149-174
        x_t = x_T
        for i in reversed(range(1, self.timesteps + 1)):
            t = torch.full((x_T.shape[0],), i, device=self.device, dtype=torch.long)
            eps, g1 = decoder(x_t, self.t_transform(t), z_1)
            _, g2 = decoder(x_t, self.t_transform(t), z_2)
            grad = (1.0 - alpha) * g1 + alpha * g2
            x_t = self._update(x_t, i, eps, grad, False)
        return x_t

    def latent_ddim_sample_loop(self, latent_denoise_fn, z_T):                          # ddim.py:200-207 (the clamping variant)
        z = z_T
        for i in reversed(range(1, self.timesteps + 1)):
            t = torch.full((z_T.shape[0],), i, device=self.device, dtype=torch.long)
            z = self._update(z, i, latent_denoise_fn(z, self.t_transform(t), None), None, False)
        return z
