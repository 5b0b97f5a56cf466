"""One representation-learning optimisation step as a single planned graph.

Replaces the body of RepresentationLearningTrainer.train (trainer/train_representation_learning.py:81-124):
    zero_grad -> encoder fwd -> randint/randn -> q_sample -> ShiftUNet fwd -> weighted L2 -> backward
    (DDP all-reduce of the trainable gradients) -> Adam -> EMA
with one static op list over pre-allocated buffers:
  * the frozen trunk / eps-branch keep no activations and emit no backward;
  * loss and d(loss)/d(shift) come out of one kernel; the loss stays on the device (the reference's two
    `.item()` syncs per micro-batch, :107-108, are gone -- read `last_loss` when you want the value);
  * gradients live in two flat buffers (decoder-trainable, encoder); the data-parallel exchange is an
    all-reduce(sum) over contiguous ranges of them, launched as soon as the backward has finished each range
    (reverse execution order) so RCCL overlaps with the rest of the backward; 1/world is folded into Adam;
  * Adam (torch.optim.Adam semantics) and the EMA update are one kernel per flat buffer.
"""
import math
import random

import torch
import torch.distributed as dist

from .. import hip as H
from ..engine import Plan, Builder
from ..model import graph as G


class FusedRLStep:
    def __init__(self, gaussian_diffusion, encoder, decoder, ema_encoder, ema_decoder, batch, height, width, lr=1e-4, betas=(0.9, 0.999),
                 eps=1e-8, weight_decay=0.0, decoupled=False, ema_decay=0.9999, ema_every=1, num_iterations=1, process_group=None,
                 bucket_mb=48, math=None):
        gd = gaussian_diffusion
        self.gd, self.enc, self.dec, self.ema_enc, self.ema_dec = gd, encoder, decoder, ema_encoder, ema_decoder
        self.N, self.Hh, self.W = batch, height, width
        self.lr, self.b1, self.b2, self.eps, self.wd, self.decoupled = lr, betas[0], betas[1], eps, weight_decay, int(decoupled)
        self.ema_decay, self.ema_every, self.num_iterations = ema_decay, ema_every, num_iterations
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.step_count = 0
        self.micro = 0
        dev = decoder.device
        cfg = decoder.cfg
        N, Hh, W, Cimg = batch, height, width, cfg["input_channel"]
        per = Hh * W * Cimg
        acc = num_iterations > 1
        drop = float(cfg["dropout"]) if decoder._shift_train else 0.0

        p = Plan(dev)
        self.plan = p
        self.x0 = p.buf(N, Hh, W, Cimg)
        self.noise = p.buf(N, Hh, W, Cimg)
        self.t = p.buf(N, dtype=torch.int64)
        self.loss = p.buf(1)
        Be = Builder(p, encoder.P, encoder.grads(), save=True, acc_grads=acc, math=math)
        Bd = Builder(p, decoder.P, decoder.grads(), save=False, drop_p=drop, acc_grads=acc, math=math, frozen_of=decoder)
        # ---- forward
        z, ex = G.encoder_forward(Be, encoder.NAME, self.x0)
        x_t = p.buf(N, Hh, W, Cimg)
        p.emit(H.op_q_sample(self.x0, self.noise, self.t, gd.sqrt_alphas_cumprod, gd.sqrt_one_minus_alphas_cumprod, N, per, x_t))
        fx = G.unet_forward(Bd, cfg, x_t, self.t, decoder.freqs, z=z, shift=True, train_shift=True, dropout=drop > 0)
        d_shift = p.buf(N, Hh, W, Cimg)
        p.emit(H.op_loss(self.noise, fx.eps, fx.shift, self.t, gd.shift_coef, gd.weight, N, per, self.loss, None, dg=d_shift,
                         scale=1.0 / num_iterations), ws_slot=9)
        self.n_fwd = len(p.recs)
        self.z, self.eps, self.shift = z, fx.eps, fx.shift
        # ---- backward, with the op index at which each parameter block's gradients are final
        marks = []
        dz = G.shift_backward(Bd, fx, d_shift, mark=lambda prefix: marks.append((len(p.recs), prefix)))
        G.encoder_backward(Be, ex, dz)
        self.n_bwd = len(p.recs)
        # ---- optimizer (+EMA): one op per flat buffer; scalars are patched every step
        self.m = [torch.zeros_like(decoder.flat_train), torch.zeros_like(encoder.flat_train)]
        self.v = [torch.zeros_like(decoder.flat_train), torch.zeros_like(encoder.flat_train)]
        self.adam_idx = []
        self.flat_nets = [decoder, encoder]
        for k, (net, ema) in enumerate([(decoder, ema_decoder), (encoder, ema_encoder)]):
            idx = p.emit(H.op_adam_ema(net.flat_train, net.flat_grad, self.m[k], self.v[k], ema.flat_train if ema is not None else None,
                                       net.flat_train.numel(), lr, self.b1, self.b2, eps, weight_decay, self.decoupled, lr, 1.0, 1.0, ema_decay))
            self.adam_idx.append((idx, ema.flat_train.data_ptr() if ema is not None else 0))
        p.compile()
        self.buckets = self._make_buckets(marks, bucket_mb)

    # ------------------------------------------------------------------ DDP buckets
    def _make_buckets(self, marks, bucket_mb):
        """[(op_index_after_which_ready, flat_grad_tensor_slice)] in backward order."""
        dec = self.dec
        limit = bucket_mb * (1 << 20) // 4
        ranges = []
        for op_idx, prefix in marks:
            offs = [(o, n) for k, (tr, o, n) in dec._offs.items() if tr and k.startswith(prefix)]
            if offs:
                ranges.append((op_idx, min(o for o, _ in offs), max(o + n for o, n in offs)))
        buckets, cur_hi = [], dec.flat_grad.numel()
        pend_lo = cur_hi
        for op_idx, lo, hi in ranges:                    # backward order: descending offsets
            pend_lo = min(pend_lo, lo)
            if cur_hi - pend_lo >= limit:
                buckets.append((op_idx, dec.flat_grad[pend_lo:cur_hi]))
                cur_hi = pend_lo
        if cur_hi > 0:
            buckets.append((ranges[-1][0] if ranges else self.n_bwd, dec.flat_grad[0:cur_hi]))
        buckets.append((self.n_bwd, self.enc.flat_grad))
        return buckets

    # ------------------------------------------------------------------ one micro-batch / one step
    def load_batch(self, x_0, t=None, noise=None):
        """x_0 / noise: (N,C,H,W) tensors of any strides.  t, noise are drawn like the reference
        (torch.randint then torch.randn_like, gaussian_diffusion.py:240-241) unless injected."""
        self.x0.copy_(x_0.permute(0, 2, 3, 1))
        if t is None:
            t = torch.randint(0, self.gd.timesteps, (self.N,), device=self.x0.device, dtype=torch.long)
        self.t.copy_(t)
        if noise is None:
            self.noise.normal_()
        else:
            self.noise.copy_(noise.permute(0, 2, 3, 1))

    def _patch_adam(self):
        step = self.step_count + 1
        bc1, bc2 = 1.0 - self.b1 ** step, 1.0 - self.b2 ** step
        use_ema = (step % self.ema_every) == 0
        for idx, ema_ptr in self.adam_idx:
            op = self.plan.arr[idx]
            op.f[5] = self.lr / bc1
            op.f[6] = 1.0 / math.sqrt(bc2)
            op.f[7] = 1.0 / self.world
            op.p[4] = ema_ptr if (use_ema and ema_ptr) else None

    def step(self, x_0, t=None, noise=None):
        """Runs one micro-batch; every `num_iterations`-th call also reduces gradients and applies Adam + EMA.
        Returns the (device-resident) loss of this micro-batch."""
        p = self.plan
        if self.micro == 0 and self.num_iterations > 1:
            self.dec.flat_grad.zero_()
            self.enc.flat_grad.zero_()
        self.load_batch(x_0, t, noise)
        if p.drop_ops:
            p.set_dropout(random.getrandbits(31), self.step_count * self.num_iterations + self.micro)   # host RNG: no device sync
        last = self.micro == self.num_iterations - 1
        if last and self.world > 1:
            self.backward_with_allreduce(p.run)
        else:
            p.run(0, self.n_bwd)
        self.micro += 1
        if last:
            self._patch_adam()
            p.run(self.n_bwd, p.n)
            self.step_count += 1
            self.micro = 0
        return self.loss

    def backward_with_allreduce(self, run):
        """Issues the forward+backward op list in segments; as soon as a bucket's gradients are final its all-reduce(sum) is
        launched asynchronously (RCCL stream), overlapping with the rest of the backward.  `run(first, last)` issues plan ops."""
        works, cur = [], 0
        for op_idx, view in self.buckets:
            run(cur, op_idx)
            cur = op_idx
            works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        run(cur, self.n_bwd)
        for w in works:
            w.wait()

    @property
    def last_loss(self):
        return float(self.loss.item())


# ----------------------------------------------------------------------------------------------------------
# torch.optim.Adam-compatible optimizer state (checkpoint key 'optimizer', train_representation_learning.py:214-226)
# ----------------------------------------------------------------------------------------------------------
def _param_order(groups):
    """[(net, name)] in the reference's param-group order."""
    order = []
    for net, prefixes in groups:
        for k, p in net.P.items():
            if p.requires_grad and (prefixes is None or k.startswith(prefixes)):
                order.append((net, k))
    return order


def export_adam_state(step_obj, groups, lr, betas, eps, weight_decay):
    nets = {id(n): i for i, n in enumerate(step_obj.flat_nets)}
    state, idx, param_groups = {}, 0, []
    for net, prefixes in groups:
        ids = []
        for n_, k in _param_order([(net, prefixes)]):
            tr, o, n = net._offs[k]
            fi = nets[id(net)]
            state[idx] = {"step": torch.tensor(float(step_obj.step_count)),
                          "exp_avg": net._view(step_obj.m[fi], o, net._shapes[k]).clone(),
                          "exp_avg_sq": net._view(step_obj.v[fi], o, net._shapes[k]).clone()}
            ids.append(idx)
            idx += 1
        param_groups.append({"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False, "params": ids})
    return {"state": state, "param_groups": param_groups}


def load_adam_state(step_obj, groups, sd):
    nets = {id(n): i for i, n in enumerate(step_obj.flat_nets)}
    idx = 0
    for net, prefixes in groups:
        for n_, k in _param_order([(net, prefixes)]):
            st = sd["state"].get(idx)
            if st is not None:
                tr, o, n = net._offs[k]
                fi = nets[id(net)]
                net._view(step_obj.m[fi], o, net._shapes[k]).copy_(st["exp_avg"])
                net._view(step_obj.v[fi], o, net._shapes[k]).copy_(st["exp_avg_sq"])
                step_obj.step_count = int(float(st["step"]))
            idx += 1


class FusedRegularStep:
    """One optimisation step of a plain DDPM UNet (config #1, trainer/train_regular_diffusion.py:59-141 +
    gaussian_diffusion.py:199-211): q_sample -> UNet fwd -> L2 -> full backward -> all-reduce -> Adam -> EMA, one plan."""

    def __init__(self, gaussian_diffusion, net, ema_net, batch, height, width, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 decoupled=False, ema_decay=0.9999, process_group=None):
        gd = gaussian_diffusion
        self.gd, self.net, self.ema = gd, net, ema_net
        self.N = batch
        self.lr, self.b1, self.b2 = lr, betas[0], betas[1]
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.step_count = 0
        cfg = net.cfg
        N, Hh, W, Cimg = batch, height, width, cfg["input_channel"]
        per = Hh * W * Cimg
        p = Plan(net.device)
        self.plan = p
        self.x0, self.noise = p.buf(N, Hh, W, Cimg), p.buf(N, Hh, W, Cimg)
        self.t = p.buf(N, dtype=torch.int64)
        self.cond = p.buf(N, dtype=torch.int64) if cfg.get("num_class") is not None else None
        self.loss = p.buf(1)
        drop = float(cfg["dropout"])
        B = Builder(p, net.P, net.grads(), save=True, drop_p=drop)
        x_t = p.buf(N, Hh, W, Cimg)
        p.emit(H.op_q_sample(self.x0, self.noise, self.t, gd.sqrt_alphas_cumprod, gd.sqrt_one_minus_alphas_cumprod, N, per, x_t))
        fx = G.unet_forward(B, cfg, x_t, self.t, net.freqs, cond=self.cond, dropout=drop > 0)
        d_eps = p.buf(*fx.eps.shape)
        p.emit(H.op_loss(self.noise, fx.eps, None, None, None, None, N, fx.eps.numel() // N, self.loss, None, deps=d_eps), ws_slot=9)
        G.unet_backward(B, fx, d_eps)
        self.n_bwd = len(p.recs)
        self.m, self.v = [torch.zeros_like(net.flat_train)], [torch.zeros_like(net.flat_train)]
        self.flat_nets = [net]
        self.adam_idx = p.emit(H.op_adam_ema(net.flat_train, net.flat_grad, self.m[0], self.v[0], ema_net.flat_train if ema_net is not None else None,
                                             net.flat_train.numel(), lr, self.b1, self.b2, eps, weight_decay, int(decoupled), lr, 1.0, 1.0, ema_decay))
        p.compile()

    def step(self, x_0, condition=None, t=None, noise=None):
        p = self.plan
        self.x0.copy_(x_0.permute(0, 2, 3, 1))
        self.t.copy_(torch.randint(0, self.gd.timesteps, (self.N,), device=self.x0.device, dtype=torch.long) if t is None else t)
        if noise is None:
            self.noise.normal_()
        else:
            self.noise.copy_(noise.permute(0, 2, 3, 1))
        if self.cond is not None:
            self.cond.copy_(condition)
        if p.drop_ops:
            p.set_dropout(random.getrandbits(31), self.step_count)
        p.run(0, self.n_bwd)
        if self.world > 1:
            dist.all_reduce(self.net.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
        step = self.step_count + 1
        op = p.arr[self.adam_idx]
        op.f[5] = self.lr / (1.0 - self.b1 ** step)
        op.f[6] = 1.0 / math.sqrt(1.0 - self.b2 ** step)
        op.f[7] = 1.0 / self.world
        p.run(self.n_bwd, p.n)
        self.step_count = step
        return self.loss


class FusedLatentStep:
    """One optimisation step of the latent DPM (config #5, trainer/train_latent_diffusion.py:95-178 +
    gaussian_diffusion.py:373-398): q_sample on the latent schedule (constant beta 0.008) -> MLPSkipNet fwd -> L1 ->
    backward -> all-reduce -> Adam / AdamW -> EMA, one plan.  `z_0` is the already normalised latent."""

    def __init__(self, gaussian_diffusion, net, ema_net, batch, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, decoupled=True,
                 ema_decay=0.9999, process_group=None):
        gd = gaussian_diffusion
        lcfg = gd.latent_diffusion_config
        self.gd, self.net, self.ema = gd, net, ema_net
        self.N, self.timesteps = batch, lcfg["timesteps"]
        self.lr, self.b1, self.b2 = lr, betas[0], betas[1]
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.step_count = 0
        ic = net.cfg["input_channel"]
        p = Plan(net.device)
        self.plan = p
        self.z0, self.noise = p.buf(batch, ic), p.buf(batch, ic)
        self.t = p.buf(batch, dtype=torch.int64)
        self.loss = p.buf(1)
        B = Builder(p, net.P, net.grads(), save=True)
        z_t = p.buf(batch, ic)
        p.emit(H.op_q_sample(self.z0, self.noise, self.t, lcfg["sqrt_alphas_cumprod"], lcfg["sqrt_one_minus_alphas_cumprod"], batch, ic, z_t))
        fx = net._emit_forward(B, z_t, self.t)
        d_out = p.buf(batch, ic)
        p.emit(H.op_loss(self.noise, fx.out, None, None, None, None, batch, ic, self.loss, None, deps=d_out, l1=1), ws_slot=9)
        net._emit_backward(B, fx, d_out)
        self.n_bwd = len(p.recs)
        self.m, self.v = [torch.zeros_like(net.flat_train)], [torch.zeros_like(net.flat_train)]
        self.flat_nets = [net]
        self.adam_idx = p.emit(H.op_adam_ema(net.flat_train, net.flat_grad, self.m[0], self.v[0], ema_net.flat_train if ema_net is not None else None,
                                             net.flat_train.numel(), lr, self.b1, self.b2, eps, weight_decay, int(decoupled), lr, 1.0, 1.0, ema_decay))
        p.compile()

    def step(self, z_0, t=None, noise=None):
        p = self.plan
        self.z0.copy_(z_0)
        self.t.copy_(torch.randint(0, self.timesteps, (self.N,), device=self.z0.device, dtype=torch.long) if t is None else t)
        if noise is None:
            self.noise.normal_()
        else:
            self.noise.copy_(noise)
        p.run(0, self.n_bwd)
        if self.world > 1:
            dist.all_reduce(self.net.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
        step = self.step_count + 1
        op = p.arr[self.adam_idx]
        op.f[5] = self.lr / (1.0 - self.b1 ** step)
        op.f[6] = 1.0 / math.sqrt(1.0 - self.b2 ** step)
        op.f[7] = 1.0 / self.world
        p.run(self.n_bwd, p.n)
        self.step_count = step
        return self.loss
