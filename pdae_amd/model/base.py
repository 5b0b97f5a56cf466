"""Shared plumbing of the network front-ends: device choice, plan cache, autograd bridge."""
import math

import torch

from .. import hip as H
from ..engine import Plan, Builder
from ..nn import FlatModule


def default_device(device=None):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise H.PdaeError("no ROCm device visible: pdae_amd runs its networks only through libpdae_hip.so "
                          "(pass device='cpu' explicitly only to inspect parameters / build plans)")
    return torch.device("cuda", torch.cuda.current_device())


def temb_freqs(dim, device, max_period=10000):
    """freqs of timestep_embedding, computed on the host exactly like the reference does
    (model/module.py:77-79 builds them with torch on CPU and moves them to the device)."""
    half = dim // 2
    f = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    return f.to(device)


def to_nhwc_(dst, x):
    """copies an (N,C,H,W)-shaped tensor of any strides into the packed NHWC plan buffer."""
    dst.copy_(x.permute(0, 2, 3, 1))


def as_nchw(y):
    """NHWC plan buffer -> fresh tensor with the reference's (N,C,H,W) shape (channels_last strides)."""
    return y.clone().permute(0, 3, 1, 2)


class _Bridge(torch.autograd.Function):
    """Connects a planned forward/backward pair to torch autograd so that `loss.backward()` of reference-style
    training code works.  Parameter gradients are written straight into the module's flat gradient buffer."""

    @staticmethod
    def forward(ctx, dummy, run_fwd, run_bwd, n_out, *inputs):
        ctx.run_bwd = run_bwd
        ctx.n_in = len(inputs)
        outs = run_fwd(*inputs)
        return outs if n_out > 1 else outs[0]

    @staticmethod
    def backward(ctx, *douts):
        gin = ctx.run_bwd(*douts)
        gin = list(gin) + [None] * (ctx.n_in - len(gin))
        return (None, None, None, None, *gin)


class PlannedNet(FlatModule):
    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_dummy", None)

    def _dummy_leaf(self):
        if self._dummy is None:
            object.__setattr__(self, "_dummy", torch.zeros(1, device=self.device, requires_grad=True))
        return self._dummy

    def _wants_grad(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.P.values())

    def _bridge(self, plan, run_fwd, run_bwd, n_out, *inputs):
        """Hooks one planned forward / backward pair into torch autograd with torch's gradient semantics:
          * backward ACCUMULATES into the flat gradient buffer (the bridge plans are built with acc_grads=True), so reference-style loops
            that run several micro-batches between zero_grad() and optimizer.step() (runner_config.num_iterations) see the sum;
          * `optimizer.zero_grad()` of a torch optimizer sets `.grad = None`: the next backward then starts from zero and re-binds every
            parameter's `.grad` to its view of the flat buffer;
          * a plan keeps ONE set of saved activations: a backward whose forward has been overwritten by a later forward of the same module
            and input shape raises instead of returning the wrong gradient."""
        plan.fwd_serial = getattr(plan, "fwd_serial", 0) + 1
        serial = plan.fwd_serial

        def bwd(*douts):
            if plan.fwd_serial != serial:
                raise RuntimeError("pdae_amd: backward() of a forward pass whose saved activations were overwritten by a later forward of the "
                                   "same module and input shape (a planned network keeps one set of activations per plan) -- call backward "
                                   "before the next forward, or run the extra forward under torch.no_grad()")
            G = self.grads()
            first = next(iter(G), None)
            if first is not None and self.P[first].grad is None:          # zero_grad(set_to_none=True)
                self.flat_grad.zero_()
            out = run_bwd(*douts)
            for k, g in G.items():
                if self.P[k].grad is not g:
                    self.P[k].grad = g
            return out

        return _Bridge.apply(self._dummy_leaf(), run_fwd, bwd, n_out, *inputs)
