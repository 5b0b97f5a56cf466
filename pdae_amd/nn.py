"""Parameter containers for the planned-graph engine.

`FlatModule` is a torch.nn.Module whose parameters are *views* into two flat fp32 device buffers
(trainable | frozen), registered under exactly the reference's state-dict key names, so that
  * `state_dict()/load_state_dict()/named_parameters()/parameters()` are drop-in
    (checkpoint layout of trainer/train_representation_learning.py:214-244),
  * the optimizer + EMA are one fused kernel over a flat segment (instead of per-tensor loops,
    train_representation_learning.py:192-212),
  * DDP-style gradient exchange is an all-reduce over contiguous ranges of one flat gradient buffer.
4-D conv weights live in channels_last memory ([Cout][KH][KW][Cin]) -- what the implicit-GEMM kernels
read -- while presenting the reference (Cout,Cin,KH,KW) shape; loading a reference checkpoint needs no repack.
"""
import copy
import math
from collections import OrderedDict

import torch
import torch.nn as nn

ALIGN = 64  # floats: every parameter starts on a 256-byte boundary


class ParamNode(nn.Module):
    """Anonymous container used to rebuild the reference's module tree (names only)."""


def _round_up(n, a=ALIGN):
    return (n + a - 1) // a * a


class FlatModule(nn.Module):
    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_plans", {})
        object.__setattr__(self, "_wp_cache", {})           # prepared copies of frozen convolution weights, shared by all plans of this module (engine.Builder)
        object.__setattr__(self, "_frozen_version", 0)      # bumped whenever the frozen parameters may have changed (see Plan.watch)

    # ------------------------------------------------------------------ construction
    def _materialize(self, shapes, trainable_fn, device):
        device = torch.device(device)
        names = list(shapes.keys())
        offs, n_tr, n_fz = {}, 0, 0
        for k in names:
            n = 1
            for s in shapes[k]:
                n *= s
            if trainable_fn(k):
                offs[k] = (True, n_tr, n); n_tr += _round_up(n)
            else:
                offs[k] = (False, n_fz, n); n_fz += _round_up(n)
        self.flat_train = torch.zeros(max(n_tr, ALIGN), dtype=torch.float32, device=device)
        self.flat_frozen = torch.zeros(max(n_fz, ALIGN), dtype=torch.float32, device=device)
        object.__setattr__(self, "_offs", offs)
        object.__setattr__(self, "_shapes", OrderedDict(shapes))
        object.__setattr__(self, "P", OrderedDict())
        object.__setattr__(self, "G", None)
        object.__setattr__(self, "flat_grad", None)
        for k in names:
            tr, o, n = offs[k]
            p = nn.Parameter(self._view(self.flat_train if tr else self.flat_frozen, o, shapes[k]), requires_grad=tr)
            self.P[k] = p
            node = self
            parts = k.split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, ParamNode())
                node = node._modules[part]
            node.register_parameter(parts[-1], p)

    @staticmethod
    def _view(flat, off, shape):
        n = 1
        for s in shape:
            n *= s
        seg = flat[off:off + n]
        if len(shape) == 4 and (shape[2] > 1 or shape[3] > 1):
            co, ci, kh, kw = shape
            return seg.view(co, kh, kw, ci).permute(0, 3, 1, 2)      # channels_last memory, reference shape
        return seg.view(*shape)

    def grads(self):
        """name -> gradient view (allocates the flat gradient buffer on first use and binds .grad)."""
        if self.G is None:
            fg = torch.zeros_like(self.flat_train)
            G = OrderedDict()
            for k, (tr, o, n) in self._offs.items():
                if tr:
                    G[k] = self._view(fg, o, self._shapes[k])
                    self.P[k].grad = G[k]
            object.__setattr__(self, "flat_grad", fg)
            object.__setattr__(self, "G", G)
        return self.G

    def zero_grad(self, set_to_none=False):
        if self.flat_grad is not None:
            self.flat_grad.zero_()

    @property
    def device(self):
        return self.flat_train.device

    # ------------------------------------------------------------------ nn.Module protocol
    def _apply(self, fn, recurse=True):
        before = self.flat_train.device
        probe = fn(torch.empty(0, dtype=torch.float32, device=before))
        if probe.device != before or probe.dtype != torch.float32:
            raise RuntimeError("FlatModule parameters are views of flat device buffers: construct the module with "
                               "device=... instead of moving or casting it")
        return self

    def __deepcopy__(self, memo):
        new = self._clone_empty()
        new.flat_train.copy_(self.flat_train)
        new.flat_frozen.copy_(self.flat_frozen)
        new.train(self.training)
        return new

    def _clone_empty(self):
        raise NotImplementedError

    def invalidate_plans(self):
        self._plans.clear()
        self._wp_cache.clear()

    def frozen_changed(self):
        """Call after writing frozen parameters by any route other than load_state_dict / reset_parameters: plans keep prepared
        (bf16-split, fragment-ordered) copies of the frozen convolution weights and refresh them when this counter moves."""
        object.__setattr__(self, "_frozen_version", self._frozen_version + 1)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.frozen_changed()
        return out

    def is_frozen_storage(self, t):
        """True when tensor t is a view of the frozen flat buffer (never written by the optimizer / EMA kernels)."""
        lo = self.flat_frozen.data_ptr()
        return lo <= t.data_ptr() < lo + self.flat_frozen.numel() * 4

    # ------------------------------------------------------------------ init (torch default init of the reference layers)
    @torch.no_grad()
    def reset_parameters(self, zero_names=()):
        for k, p in self.P.items():
            shape = self._shapes[k]
            if any(k.endswith(z) or (z in k) for z in zero_names):
                p.zero_()
            elif k.endswith(".bias") and len(shape) == 1:
                wk = k[:-4] + "weight"
                ws = self._shapes.get(wk)
                if ws is not None and len(ws) > 1:
                    fan_in = 1
                    for s in ws[1:]:
                        fan_in *= s
                    bound = 1.0 / math.sqrt(fan_in)
                    p.uniform_(-bound, bound)
                else:
                    p.zero_()                                    # GroupNorm / LayerNorm beta
            elif len(shape) == 1:
                p.fill_(1.0)                                     # GroupNorm / LayerNorm gamma
            else:
                fan_in = 1
                for s in shape[1:]:
                    fan_in *= s
                bound = 1.0 / math.sqrt(fan_in)                  # kaiming_uniform_(a=sqrt(5))
                tmp = torch.empty(shape, dtype=torch.float32, device=p.device).uniform_(-bound, bound)
                p.copy_(tmp)
        self.frozen_changed()
