"""Semantic encoders (model/representation_learning/encoder/*.py): name -> class, resolved by
`getattr(module, config["model"])` exactly like the reference trainer does (train_representation_learning.py:28)."""
import torch

from ....engine import Plan, Builder
from ... import graph as G
from ...base import PlannedNet, default_device, to_nhwc_, _Bridge


class _Encoder(PlannedNet):
    NAME = None

    def __init__(self, device=None, **kwargs):
        super().__init__()
        self.latent_dim = kwargs["latent_dim"]
        dev = default_device(device)
        self._materialize(G.encoder_shapes(self.NAME, self.latent_dim), lambda k: True, dev)
        self.reset_parameters(zero_names=(".proj_out.",))

    def _clone_empty(self):
        return type(self)(device=self.device, latent_dim=self.latent_dim)

    def plan(self, N, Hh, W, train):
        key = (N, Hh, W, bool(train))
        pl = self._plans.get(key)
        if pl is not None:
            return pl
        p = Plan(self.device)
        x = p.buf(N, Hh, W, 3)
        B = Builder(p, self.P, self.grads() if train else None, save=bool(train), acc_grads=bool(train))
        z, ex = G.encoder_forward(B, self.NAME, x)
        p.n_fwd = len(p.recs)
        p.dz = None
        if train:
            p.dz = p.buf(N, self.latent_dim)
            G.encoder_backward(B, ex, p.dz)
        p.x, p.z = x, z
        p.compile()
        self._plans[key] = p
        return p

    def forward(self, x):
        N, _, Hh, W = x.shape
        train = self._wants_grad()
        p = self.plan(N, Hh, W, train)

        def run_fwd():
            to_nhwc_(p.x, x)
            p.run(0, p.n_fwd)
            return (p.z.clone(),)

        if not train:
            return run_fwd()[0]

        def run_bwd(dz):
            p.dz.copy_(dz)
            p.run(p.n_fwd, p.n)
            return ()

        return self._bridge(p, run_fwd, run_bwd, 1)


class FFHQEncoder(_Encoder):
    NAME = "FFHQEncoder"


class CELEBAHQEncoder(_Encoder):
    NAME = "CELEBAHQEncoder"


class BEDROOMEncoder(_Encoder):
    NAME = "BEDROOMEncoder"


class HORSEEncoder(_Encoder):
    NAME = "HORSEEncoder"


class CELEBA64Encoder(_Encoder):
    NAME = "CELEBA64Encoder"
