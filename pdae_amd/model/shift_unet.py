"""ShiftUNet front-end (model/shift_unet.py:29-44, 253-310): frozen pre-trained UNet + trainable shift branch."""
import random

import torch

from .. import hip as H
from ..engine import Plan, Builder
from . import graph as G
from .base import PlannedNet, default_device, temb_freqs, to_nhwc_, as_nchw, _Bridge

TRAINABLE_PREFIXES = ("label_emb.", "shift_middle_block.", "shift_output_blocks.", "shift_out.")


def is_trainable(key):
    return key.startswith(TRAINABLE_PREFIXES)


class ShiftUNet(PlannedNet):
    def __init__(self, input_channel, base_channel, channel_multiplier, num_residual_blocks_of_a_block, attention_resolutions,
                 num_heads, head_channel, use_new_attention_order, dropout, latent_dim, dims=2, learn_sigma=False, device=None, **kwargs):
        super().__init__()
        assert dims == 2, "the PDAE path is 2-D"
        cfg = dict(input_channel=input_channel, base_channel=base_channel, channel_multiplier=list(channel_multiplier),
                   num_residual_blocks_of_a_block=num_residual_blocks_of_a_block, attention_resolutions=list(attention_resolutions),
                   num_heads=num_heads, head_channel=head_channel, use_new_attention_order=use_new_attention_order, dropout=dropout,
                   learn_sigma=learn_sigma)
        object.__setattr__(self, "cfg", cfg)
        self.latent_dim = latent_dim
        self.base_channel = base_channel
        dev = default_device(device)
        # freeze(): time_embed / input_blocks / middle_block / output_blocks / out never receive gradients
        self._materialize(G.unet_shapes(cfg, shift=True, latent_dim=latent_dim), is_trainable, dev)
        self.reset_parameters(zero_names=G.ZERO_INIT)
        object.__setattr__(self, "freqs", temb_freqs(base_channel, dev))
        object.__setattr__(self, "_shift_train", False)

    def _clone_empty(self):
        return ShiftUNet(device=self.device, latent_dim=self.latent_dim, **self.cfg)

    # reference API (shift_unet.py:287-310)
    def set_train_mode(self):
        object.__setattr__(self, "_shift_train", True)

    def set_eval_mode(self):
        object.__setattr__(self, "_shift_train", False)

    def freeze(self):
        pass   # the frozen half lives in its own flat buffer and is never given a gradient

    # ------------------------------------------------------------------ plans
    def plan(self, N, Hh, W, train):
        dropout = bool(train) and self._shift_train and float(self.cfg["dropout"]) > 0
        key = (N, Hh, W, bool(train), dropout)
        pl = self._plans.get(key)
        if pl is not None:
            return pl
        cfg = self.cfg
        p = Plan(self.device)
        x = p.buf(N, Hh, W, cfg["input_channel"])
        t = p.buf(N, dtype=torch.int64)
        z = p.buf(N, self.latent_dim)
        B = Builder(p, self.P, self.grads() if train else None, save=False, drop_p=float(cfg["dropout"]) if dropout else 0.0, frozen_of=self,
                    acc_grads=bool(train))
        fx = G.unet_forward(B, cfg, x, t, self.freqs, z=z, shift=True, train_shift=bool(train), dropout=dropout)
        p.n_fwd = len(p.recs)
        p.d_shift = p.dz = None
        if train:
            p.d_shift = p.buf(*fx.shift.shape)
            p.dz = G.shift_backward(B, fx, p.d_shift)
        p.x, p.t, p.z, p.eps, p.shift = x, t, z, fx.eps, fx.shift
        p.compile()
        self._plans[key] = p
        return p

    def plan_eps(self, N, Hh, W):
        """Sampling plan of the frozen half alone: the eps prediction of the pre-trained UNet (time_embed, input_blocks, middle_block,
        output_blocks, out) on the SAME x / t buffers as plan(N, Hh, W, False), and none of the shift branch's ops.  The last
        stop_percent * N steps of shift_ddim_sample_loop discard the shift term (diffusion/ddim.py:94-96,115,119 of the reference: the decoder
        still runs there and `g` is dropped); a DDIM loop on planned networks runs this op list on those steps instead -- 43 % fewer decoder
        FLOPs on 30 % of the steps of latent_diffusion_sample (gaussian_diffusion.py:415, stop_percent 0.3).  Same kernels on the same
        shapes as the eps half of the full plan: the eps tensor is bit-identical (tests/test_diffusion_gpu.py)."""
        key = (N, Hh, W, "eps")
        pl = self._plans.get(key)
        if pl is not None:
            return pl
        full = self.plan(N, Hh, W, False)
        p = Plan(self.device)
        B = Builder(p, self.P, None, save=False, drop_p=0.0, frozen_of=self)
        fx = G.unet_forward(B, self.cfg, full.x, full.t, self.freqs, shift=False)
        p.n_fwd = len(p.recs)
        p.x, p.t, p.z, p.eps, p.shift = full.x, full.t, None, fx.eps, None
        p.compile()
        self._plans[key] = p
        return p

    # ------------------------------------------------------------------ forward
    def forward(self, x, time, condition):
        N, _, Hh, W = x.shape
        train = self._wants_grad()
        p = self.plan(N, Hh, W, train)

        def run_fwd(zz):
            to_nhwc_(p.x, x)
            p.t.copy_(time)
            p.z.copy_(zz)
            if p.drop_ops:
                p.set_dropout(random.getrandbits(31), 0)
            p.run(0, p.n_fwd)
            return as_nchw(p.eps), as_nchw(p.shift)

        if not train:
            return run_fwd(condition)

        def run_bwd(d_eps, d_shift):
            # the eps branch is frozen and x / t carry no gradient: only d_shift matters
            if d_shift is None:
                p.d_shift.zero_()
            else:
                to_nhwc_(p.d_shift, d_shift)
            p.run(p.n_fwd, p.n)
            return (p.dz.clone(),)

        return self._bridge(p, run_fwd, run_bwd, 2, condition)
