"""Tensor-level wrappers over the diffusion elementwise kernels (allocate outputs, launch on the current stream)."""
import torch

from .. import hip as H


def _c(x):
    return x if x.is_contiguous() else x.contiguous()


def q_sample(x0, noise, t, sqrt_ac, sqrt_1mac):
    x0, noise = _c(x0), _c(noise)
    out = torch.empty_like(x0)
    N = x0.shape[0]
    H.run(H.op_q_sample(x0, noise, t, sqrt_ac, sqrt_1mac, N, x0.numel() // N, out))
    return out


class _Loss(torch.autograd.Function):
    """mean(w[t] * f(noise - (eps + c[t] * g))) with both gradients produced by the same kernel launch."""

    @staticmethod
    def forward(ctx, eps, g, noise, t, tc, tw, l1):
        eps, noise = _c(eps), _c(noise)
        g = _c(g) if g is not None else None
        N = eps.shape[0]
        loss = torch.empty(1, device=eps.device)
        ws = torch.empty(2048, device=eps.device)
        deps = torch.empty_like(eps) if ctx.needs_input_grad[0] else None
        dg = torch.empty_like(g) if (g is not None and ctx.needs_input_grad[1]) else None
        H.run(H.op_loss(noise, eps, g, t, tc, tw, N, eps.numel() // N, loss, ws, deps=deps, dg=dg, l1=int(l1)))
        ctx.save_for_backward(*[x for x in (deps, dg) if x is not None])
        ctx.has = (deps is not None, dg is not None)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        saved = list(ctx.saved_tensors)
        deps = saved.pop(0) if ctx.has[0] else None
        dg = saved.pop(0) if ctx.has[1] else None
        return (None if deps is None else deps * gout, None if dg is None else dg * gout, None, None, None, None, None)


def loss(noise, eps, g=None, t=None, shift_coef=None, weight=None, l1=False):
    return _Loss.apply(eps, g, noise, t, shift_coef, weight, l1)


def ddim_step(x, eps, g, c_shift, ra, rm1, sab, s1ab, out=None, clamp=True):
    x, eps = _c(x), _c(eps)
    g = _c(g) if g is not None else None
    out = torch.empty_like(x) if out is None else out
    H.run(H.op_ddim_step(x, eps, g, x.numel(), c_shift, ra, rm1, sab, s1ab, out, clamp=int(clamp)))
    return out


def ddpm_step(x, eps, g, z, cx, ce, cs, sigma):
    x, eps = _c(x), _c(eps)
    out = torch.empty_like(x)
    H.run(H.op_ddpm_step(x, eps, None if g is None else _c(g), None if z is None else _c(z), x.numel(), cx, ce, cs, sigma, out))
    return out


def _rows(x):
    """(contiguous tensor, N, elements per sample)"""
    x = _c(x)
    return x, x.shape[0], x.numel() // x.shape[0]


def axpby_rows(a, b, ca, cb):
    """out[n] = ca[n] * a[n] + cb[n] * b[n]; ca / cb: float32 device vectors of length N."""
    a, N, per = _rows(a)
    b = _c(b)
    out = torch.empty_like(a)
    H.run(H.op_axpby_rows(a, b, _c(ca), _c(cb), N, per, out))
    return out


def ddim_step_rows(x, eps, g, coef, clamp=True):
    """DDIM update with one coefficient row per sample: coef [N,5] = (c_shift, sqrt_recip_ac, sqrt_recip_ac_m1, sqrt(ac_to), sqrt(1-ac_to))."""
    x, N, per = _rows(x)
    out = torch.empty_like(x)
    H.run(H.op_ddim_step_rows(x, _c(eps), None if g is None else _c(g), _c(coef), N, per, out, clamp=int(clamp)))
    return out


def ddpm_step_rows(x, eps, g, noise, learned_range, coef):
    """Ancestral step with one coefficient row per sample: coef [N,6] = (cx, ce, cs, mask, lv_min, lv_max)."""
    x, N, per = _rows(x)
    out = torch.empty_like(x)
    H.run(H.op_ddpm_step_rows(x, _c(eps), None if g is None else _c(g), None if noise is None else _c(noise),
                              None if learned_range is None else _c(learned_range), _c(coef), N, per, out))
    return out


def blend(a, b, alpha):
    """(1 - alpha) * a + alpha * b in one kernel (fresh tensor)."""
    out = _c(a).clone()
    H.run(H.op_axpby(_c(b), out, out.numel(), float(alpha), 1.0 - float(alpha)))
    return out
