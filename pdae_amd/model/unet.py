"""UNet front-end with the reference constructor / forward signature (model/unet.py:30-45, 177-202)."""
import random

import torch

from .. import hip as H
from ..engine import Plan, Builder
from . import graph as G
from .base import PlannedNet, default_device, temb_freqs, to_nhwc_, as_nchw, _Bridge


class UNet(PlannedNet):
    def __init__(self, input_channel, base_channel, channel_multiplier, num_residual_blocks_of_a_block, attention_resolutions,
                 num_heads, head_channel, use_new_attention_order, dropout, num_class=None, dims=2, learn_sigma=False, device=None, **kwargs):
        super().__init__()
        assert dims == 2, "the PDAE path is 2-D"
        cfg = dict(input_channel=input_channel, base_channel=base_channel, channel_multiplier=list(channel_multiplier),
                   num_residual_blocks_of_a_block=num_residual_blocks_of_a_block, attention_resolutions=list(attention_resolutions),
                   num_heads=num_heads, head_channel=head_channel, use_new_attention_order=use_new_attention_order, dropout=dropout,
                   num_class=num_class, learn_sigma=learn_sigma)
        object.__setattr__(self, "cfg", cfg)
        self.num_class = num_class
        self.base_channel = base_channel
        dev = default_device(device)
        self._materialize(G.unet_shapes(cfg), lambda k: True, dev)
        self.reset_parameters(zero_names=G.ZERO_INIT)
        if num_class is not None:
            with torch.no_grad():
                self.P["label_emb.weight"].normal_()            # nn.Embedding default init
        object.__setattr__(self, "freqs", temb_freqs(base_channel, dev))

    def _clone_empty(self):
        return UNet(device=self.device, **self.cfg)

    # ------------------------------------------------------------------ plans
    def plan(self, N, Hh, W, train):
        dropout = bool(train) and self.training and float(self.cfg["dropout"]) > 0       # fixed at plan-build time: part of the key
        key = (N, Hh, W, bool(train), dropout)
        pl = self._plans.get(key)
        if pl is not None:
            return pl
        cfg = self.cfg
        p = Plan(self.device)
        x = p.buf(N, Hh, W, cfg["input_channel"])
        t = p.buf(N, dtype=torch.int64)
        cond = p.buf(N, dtype=torch.int64) if cfg["num_class"] is not None else None
        B = Builder(p, self.P, self.grads() if train else None, save=bool(train), drop_p=float(cfg["dropout"]) if dropout else 0.0, acc_grads=bool(train))
        fx = G.unet_forward(B, cfg, x, t, self.freqs, cond=cond, dropout=dropout)
        p.n_fwd = len(p.recs)
        p.d_eps = None
        if train:
            p.d_eps = p.buf(*fx.eps.shape)
            G.unet_backward(B, fx, p.d_eps)
        p.x, p.t, p.cond, p.eps = x, t, cond, fx.eps
        p.compile()
        self._plans[key] = p
        return p

    # ------------------------------------------------------------------ forward
    def forward(self, x, time, condition=None):
        N, _, Hh, W = x.shape
        if self.num_class is not None:
            assert condition is not None
        train = self._wants_grad()
        p = self.plan(N, Hh, W, train)

        def run_fwd():
            to_nhwc_(p.x, x)
            p.t.copy_(time)
            if p.cond is not None:
                p.cond.copy_(condition)
            if p.drop_ops:
                p.set_dropout(random.getrandbits(31), 0)
            p.run(0, p.n_fwd)
            return (as_nchw(p.eps),)

        if not train:
            return run_fwd()[0]

        def run_bwd(d_eps):
            to_nhwc_(p.d_eps, d_eps)
            p.run(p.n_fwd, p.n)
            return ()

        return self._bridge(p, run_fwd, run_bwd, 1)
