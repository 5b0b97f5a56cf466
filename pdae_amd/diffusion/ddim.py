"""Respaced DDIM (eta = 0): sample / encode steps and loops under the reference's class and method names (diffusion/ddim.py).

What is different from the reference:
  * one step = ONE planned decoder forward + ONE fused update kernel (the reference: ~15 elementwise launches and 8 gathers,
    ddim.py:94-107);
  * the single-step methods take a per-sample `t` exactly like the reference and never read it on the host: the five schedule
    values of every sample are gathered on the device into a [N,5] row block that `pdae_ddim_step_rows` consumes;
  * the loops own their step index, so they use the scalar-coefficient kernel; when the model is a planned network of this package
    they run on the plan's static NHWC buffers with no per-step allocation (any other callable `fn(x, t, cond)` takes the generic
    path with identical arithmetic);
  * loops on planned networks poll the fp16-window guard once at their end and re-run in bf16x6 if it fired (hip.SaturationGuard).
"""
import os
import sys

import numpy as np
import torch

from .. import hip as H
from . import ops


def respaced_tables(betas):
    """Schedule tables over a respaced beta sequence (ddim.py:8-33), evaluated in the dtype the reference's numpy expressions run in:
    `betas` keeps its dtype (float32 when it comes from float32 alphas_cumprod, gaussian_diffusion.py:276), the products / roots of the
    cumulative product stay in it, while the shifted copies are float64 because they are joined with a Python float."""
    betas = np.asarray(betas)
    ac = np.cumprod(1.0 - betas, axis=0)
    shifted_prev = np.concatenate((np.ones(1), ac[:-1]))            # float64 by promotion with the float64 constant
    shifted_next = np.concatenate((ac[1:], np.zeros(1)))
    return {"alphas_cumprod_prev": shifted_prev, "alphas_cumprod_next": shifted_next,
            "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac), "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
            "sqrt_recip_alphas_cumprod_m1": np.sqrt(1.0 / ac - 1.0)}


class DDIM:
    TABLES = ("alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
              "sqrt_recip_alphas_cumprod_m1")

    def __init__(self, betas, timestep_map, device):
        self.device = device
        self.timestep_map = timestep_map.to(device)
        self._map_host = timestep_map.tolist()
        self.timesteps = int(betas.shape[0]) - 1
        self._host = {}
        for name, table in respaced_tables(betas).items():
            f32 = np.asarray(table, dtype=np.float32)
            self._host[name] = f32
            setattr(self, name, torch.from_numpy(f32.copy()).to(device))
        # targets of the update: sqrt(ac_to), sqrt(1 - ac_to) of the fp32 table values (the reference takes torch.sqrt of the gathered fp32 entries)
        for tag in ("prev", "next"):
            a = self._host["alphas_cumprod_" + tag]
            self._host["to_" + tag] = np.stack([np.sqrt(a), np.sqrt(np.float32(1.0) - a)], 1).astype(np.float32)
        self._rows = {}
        self.ops_run = 0          # op records issued by the planned loops of this object (diagnostic: tests count the eps-only steps with it)

    @staticmethod
    def extract_coef_at_t(schedule, t, x_shape):
        return schedule.gather(-1, t).view(x_shape[0], *([1] * (len(x_shape) - 1)))

    def t_transform(self, t):
        return self.timestep_map[t]

    # ------------------------------------------------------------------ coefficients
    def _coefs(self, i, encode):
        """(c_shift, sqrt_recip_ac, sqrt_recip_ac_m1, sqrt(ac_to), sqrt(1-ac_to)) of step i as Python floats."""
        h = self._host
        to = h["to_next" if encode else "to_prev"][i]
        return (float(h["sqrt_one_minus_alphas_cumprod"][i]), float(h["sqrt_recip_alphas_cumprod"][i]), float(h["sqrt_recip_alphas_cumprod_m1"][i]),
                float(to[0]), float(to[1]))

    def _coef_rows(self, t, encode):
        """[N,5] device rows for a per-sample t (gathered from a [T+1,5] table built once per direction)."""
        tab = self._rows.get(encode)
        if tab is None:
            h = self._host
            tab = np.concatenate([h["sqrt_one_minus_alphas_cumprod"][:, None], h["sqrt_recip_alphas_cumprod"][:, None],
                                  h["sqrt_recip_alphas_cumprod_m1"][:, None], h["to_next" if encode else "to_prev"]], 1)
            tab = torch.from_numpy(np.ascontiguousarray(tab, dtype=np.float32)).to(self.device)
            self._rows[encode] = tab
        return tab.index_select(0, t)

    # ------------------------------------------------------------------ single steps (reference signatures, per-sample t)
    def ddim_sample(self, denoise_fn, x_t, t, condition=None):                         # ddim.py:43-55
        return ops.ddim_step_rows(x_t, denoise_fn(x_t, self.t_transform(t), condition), None, self._coef_rows(t, False))

    def ddim_encode(self, denoise_fn, x_t, t, condition=None):                         # ddim.py:66-79
        return ops.ddim_step_rows(x_t, denoise_fn(x_t, self.t_transform(t), condition), None, self._coef_rows(t, True))

    def shift_ddim_sample(self, decoder, z, x_t, t, use_shift=True):                   # ddim.py:91-107
        eps, grad = decoder(x_t, self.t_transform(t), z)
        return ops.ddim_step_rows(x_t, eps, grad if use_shift else None, self._coef_rows(t, False))

    def shift_ddim_encode(self, decoder, z, x_t, t):                                   # ddim.py:123-138
        eps, grad = decoder(x_t, self.t_transform(t), z)
        return ops.ddim_step_rows(x_t, eps, grad, self._coef_rows(t, True))

    def latent_ddim_sample(self, latent_denoise_fn, z_t, t):                           # ddim.py:179-198 (active body: no clamp, eps reused)
        return ops.ddim_step_rows(z_t, latent_denoise_fn(z_t, self.t_transform(t)), None, self._coef_rows(t, False), clamp=False)

    # ------------------------------------------------------------------ loops
    @staticmethod
    def _planned(net):
        return hasattr(net, "plan") and hasattr(net, "P") and not torch.is_grad_enabled()

    def _full(self, n, i):
        return torch.full((n,), i, device=self.device, dtype=torch.long)

    def _guarded(self, net, body):
        """Runs body() on a planned network; if the fp16-window guard fired INSIDE it, switches the process to bf16x6 and runs it again.
        The guard words are process-wide and sticky: word [0] may already be non-zero from a training step the trainer has not polled yet
        (handle_saturation owns that state, and word [1], its discarded-step count), so only the events of this loop are looked at -- the
        difference against a snapshot -- and only they are taken back."""
        g = H.SaturationGuard.get(net.device)
        watch = g is not None and H.default_math() == "f16x3"
        before = g.read()[0] if watch else 0
        out = body()
        if watch and g.read()[0] != before:
            print("[pdae_amd] fp16 window exceeded during a DDIM loop: re-running it in bf16x6 arithmetic", file=sys.stderr, flush=True)
            H.set_default_math("bf16x6")
            net.invalidate_plans()
            g.t[0:1].fill_(before)
            out = body()
        return out

    def _planned_loop(self, net, x_start, steps, encode, z=None, condition=None, use_shift=lambda i: True, trajectory=None, z_mix=None):
        """Static-buffer loop: per step one op-list launch (two for trajectory interpolation) and one update kernel.
        z_mix = (z_2, alpha): the shift term is (1-alpha)*g(z) + alpha*g(z_2), eps from the first pass (ddim.py:157-160)."""
        N, _, Hh, W = x_start.shape

        def body():
            p = net.plan(N, Hh, W, False)
            p.x.copy_(x_start.permute(0, 2, 3, 1))
            if z is not None:
                p.z.copy_(z)
            if condition is not None and getattr(p, "cond", None) is not None:
                p.cond.copy_(condition)
            shift = getattr(p, "shift", None)
            # steps whose shift term is discarded (use_shift(i) False) run the eps half alone, when the network offers that op list
            pe = net.plan_eps(N, Hh, W) if (shift is not None and z_mix is None and hasattr(net, "plan_eps") and os.environ.get("PDAE_DDIM_EPS_ONLY", "1") != "0"
                                            and any(not use_shift(i) for i in steps)) else None
            keep_eps = keep_g = None
            if z_mix is not None:
                keep_eps, keep_g = torch.empty_like(p.eps), torch.empty_like(p.shift)
            if trajectory is not None:
                del trajectory[:]
            fresh = True                               # first step: loop-invariant prefix (z-only ops) and the weight preparation run too
            for i in steps:
                p.t.fill_(self._map_host[i])
                if pe is not None and not use_shift(i):
                    pe.run(0, pe.n_fwd)                # same x / t buffers; frozen weights only: nothing to prepare per run
                    self.ops_run += pe.n_fwd
                    c_shift, ra, rm1, sab, s1ab = self._coefs(i, encode)
                    ops.ddim_step(p.x, pe.eps, None, c_shift, ra, rm1, sab, s1ab, out=p.x)
                    if trajectory is not None:
                        trajectory.append(p.x.clone().permute(0, 3, 1, 2))
                    continue
                first = 0 if fresh else getattr(p, "n_const", 0)
                p.run(first, p.n_fwd, prep=fresh)
                self.ops_run += p.n_fwd - first
                fresh = z_mix is not None              # trajectory interpolation swaps z inside the step: nothing is invariant
                eps, g = p.eps, (shift if (shift is not None and use_shift(i)) else None)
                if z_mix is not None:
                    z_2, alpha = z_mix
                    keep_eps.copy_(p.eps)
                    keep_g.copy_(p.shift)
                    p.z.copy_(z_2)
                    p.run(0, p.n_fwd)
                    p.z.copy_(z)
                    H.run(H.op_axpby(p.shift, keep_g, keep_g.numel(), alpha, 1.0 - alpha))      # keep_g = alpha*g_2 + (1-alpha)*g_1
                    eps, g = keep_eps, keep_g
                c_shift, ra, rm1, sab, s1ab = self._coefs(i, encode)
                ops.ddim_step(p.x, eps, g, c_shift, ra, rm1, sab, s1ab, out=p.x)
                if trajectory is not None:
                    trajectory.append(p.x.clone().permute(0, 3, 1, 2))
            return p.x.clone().permute(0, 3, 1, 2)

        return self._guarded(net, body)

    def _generic_loop(self, step_fn, x, steps):
        for i in steps:
            x = step_fn(x, i)
        return x

    def _step(self, x, i, eps, grad, encode, clamp=True):
        c_shift, ra, rm1, sab, s1ab = self._coefs(i, encode)
        return ops.ddim_step(x, eps, grad, c_shift, ra, rm1, sab, s1ab, clamp=clamp)

    def _down(self):
        return range(self.timesteps, 0, -1)

    def _up(self):
        return range(0, self.timesteps)

    def ddim_sample_loop(self, denoise_fn, x_T, condition=None):                        # ddim.py:57-64
        if self._planned(denoise_fn) and x_T.dim() == 4:
            return self._planned_loop(denoise_fn, x_T, self._down(), False, condition=condition)
        n = x_T.shape[0]
        return self._generic_loop(lambda x, i: self._step(x, i, denoise_fn(x, self.t_transform(self._full(n, i)), condition), None, False),
                                  x_T, self._down())

    def ddim_encode_loop(self, denoise_fn, x_0, condition=None):                        # ddim.py:81-88
        if self._planned(denoise_fn) and x_0.dim() == 4:
            return self._planned_loop(denoise_fn, x_0, self._up(), True, condition=condition)
        n = x_0.shape[0]
        return self._generic_loop(lambda x, i: self._step(x, i, denoise_fn(x, self.t_transform(self._full(n, i)), condition), None, True),
                                  x_0, self._up())

    def shift_ddim_sample_loop(self, decoder, z, x_T, stop_percent=0.0, trajectory=None):   # ddim.py:110-120
        stop_step = int(stop_percent * self.timesteps)
        use = lambda i: (i - 1) >= stop_step
        if self._planned(decoder):
            return self._planned_loop(decoder, x_T, self._down(), False, z=z, use_shift=use, trajectory=trajectory)
        n = x_T.shape[0]

        def one(x, i):
            eps, grad = decoder(x, self.t_transform(self._full(n, i)), z)
            return self._step(x, i, eps, grad if use(i) else None, False)
        return self._generic_loop(one, x_T, self._down())

    def shift_ddim_encode_loop(self, decoder, z, x_0, trajectory=None):                 # ddim.py:140-147
        if self._planned(decoder):
            return self._planned_loop(decoder, x_0, self._up(), True, z=z, trajectory=trajectory)
        n = x_0.shape[0]

        def one(x, i):
            eps, grad = decoder(x, self.t_transform(self._full(n, i)), z)
            return self._step(x, i, eps, grad, True)
        return self._generic_loop(one, x_0, self._up())

    def shift_ddim_trajectory_interpolation(self, decoder, z_1, z_2, x_T, alpha):       # ddim.py:149-174
        """Both latents are decoded at every step; eps comes from the z_1 pass, the shift terms are blended with weight alpha."""
        alpha = float(alpha)
        if self._planned(decoder):
            return self._planned_loop(decoder, x_T, self._down(), False, z=z_1, z_mix=(z_2, alpha))
        n = x_T.shape[0]

        def one(x, i):
            t = self.t_transform(self._full(n, i))
            eps, g_1 = decoder(x, t, z_1)
            g_2 = decoder(x, t, z_2)[1]
            return self._step(x, i, eps, ops.blend(g_1, g_2, alpha), False)
        return self._generic_loop(one, x_T, self._down())

    def latent_ddim_sample_loop(self, latent_denoise_fn, z_T):                          # ddim.py:200-207: the loop calls ddim_sample (clamping form)
        n = z_T.shape[0]
        return self._generic_loop(lambda z, i: self._step(z, i, latent_denoise_fn(z, self.t_transform(self._full(n, i)), None), None, False),
                                  z_T, self._down())
