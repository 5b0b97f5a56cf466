"""MLPSkipNet -- the latent DPM denoiser (model/mlp_skip_net.py:6-141), reference constructor / forward signature.

Planned graph over [B][C] row-major fp32 buffers:
    cond = time_embed(timestep_embedding(t))                                  (:68-69)
    layer i:  u = [h | x] W_i^T + b_i          two GEMMs on the column blocks of W_i, torch.cat (:73-75) never materialised
              e = silu(cond) We_i^T + be_i                                     (:100-101, 125-127)
              h = silu(LayerNorm(u * (1 + e)))   one fused row kernel           (:130-137, pdae_mlp_modln_fwd)
    last layer: plain Linear (:42-47).
Backward is hand-derived (pdae_mlp_modln_bwd + GEMMs) and bridged to torch autograd like the other front-ends.
"""
from collections import OrderedDict
from types import SimpleNamespace as NS

import torch

from .. import hip as H
from ..engine import Plan, Builder
from ..nn import ParamNode
from .base import PlannedNet, default_device, temb_freqs, _Bridge

LN_EPS = 1e-5          # nn.LayerNorm default


def mlp_shapes(cfg):
    """state-dict names / shapes in reference order (mlp_skip_net.py:27-66, 86-105)."""
    sh = OrderedDict()
    ic, mc, n = cfg["input_channel"], cfg["model_channel"], cfg["num_layers"]
    sh["time_embed.0.weight"] = (ic, cfg["time_emb_channel"]); sh["time_embed.0.bias"] = (ic,)
    sh["time_embed.2.weight"] = (ic, ic); sh["time_embed.2.bias"] = (ic,)
    for i in range(n):
        a, b = (ic, mc) if i == 0 else ((mc, ic) if i == n - 1 else (mc, mc))
        if i >= 1:
            a += ic
        p = f"layers.{i}"
        sh[p + ".linear.weight"] = (b, a); sh[p + ".linear.bias"] = (b,)
        if i != n - 1:
            sh[p + ".linear_emb.weight"] = (b, ic); sh[p + ".linear_emb.bias"] = (b,)
            if cfg["use_norm"]:
                sh[p + ".norm.weight"] = (b,); sh[p + ".norm.bias"] = (b,)
    return sh


class MLPSkipNet(PlannedNet):
    def __init__(self, input_channel, model_channel, num_layers, time_emb_channel, use_norm, dropout, device=None, **kwargs):
        super().__init__()
        if dropout and dropout > 0:
            raise NotImplementedError("MLPSkipNet dropout > 0 is not built (every shipped *_latent.yml uses dropout 0.0)")
        assert num_layers >= 2
        cfg = dict(input_channel=input_channel, model_channel=model_channel, num_layers=num_layers, time_emb_channel=time_emb_channel,
                   use_norm=bool(use_norm), dropout=0.0)
        object.__setattr__(self, "cfg", cfg)
        self.input_channel = input_channel                      # read by latent_diffusion_sample (gaussian_diffusion.py:404)
        self.time_emb_channel = time_emb_channel
        self.skip_layers = list(range(1, num_layers))
        dev = default_device(device)
        self._materialize(mlp_shapes(cfg), lambda k: True, dev)
        # the reference registers linear_emb a second time inside cond_layers (mlp_skip_net.py:100-101): same Parameter, two keys
        for i in range(num_layers - 1):
            node = self._modules["layers"]._modules[str(i)]
            cl = ParamNode(); one = ParamNode()
            one.register_parameter("weight", self.P[f"layers.{i}.linear_emb.weight"])
            one.register_parameter("bias", self.P[f"layers.{i}.linear_emb.bias"])
            cl.add_module("1", one)
            node.add_module("cond_layers", cl)
        self.reset_parameters()
        object.__setattr__(self, "freqs", temb_freqs(time_emb_channel, dev))

    def _clone_empty(self):
        return MLPSkipNet(device=self.device, **self.cfg)

    @torch.no_grad()
    def reset_parameters(self, zero_names=()):
        """torch defaults, then kaiming_normal_(relu) on every Linear of the SiLU layers (mlp_skip_net.py:114-121)."""
        super().reset_parameters(zero_names)
        n = self.cfg["num_layers"]
        for i in range(n - 1):
            for nm in ("linear", "linear_emb"):
                w = self.P[f"layers.{i}.{nm}.weight"]
                w.copy_(torch.randn(w.shape, device=w.device) * (2.0 / w.shape[1]) ** 0.5)

    # ------------------------------------------------------------------ graph
    def _emit_forward(self, B, x, t):
        cfg, pl, P = self.cfg, B.p, self.P
        n, ic = cfg["num_layers"], cfg["input_channel"]
        R = x.shape[0]
        te = pl.buf(R, cfg["time_emb_channel"])
        pl.emit(H.op_temb(t, self.freqs, R, cfg["time_emb_channel"], te))
        c0, l0 = B.linear(te, "time_embed.0")
        c0s = B.silu(c0)
        cond, l2 = B.linear(c0s, "time_embed.2")
        ca = B.silu(cond)                                        # cond_layers[0] = SiLU, shared by every layer
        fx = NS(x=x, R=R, te=te, c0=c0, c0s=c0s, cond=cond, ca=ca, l0=l0, l2=l2, layers=[])
        h = x
        for i in range(n):
            p = f"layers.{i}"
            w, b = P[p + ".linear.weight"], P[p + ".linear.bias"]
            out, K = w.shape
            Kh = K - ic if i >= 1 else K
            u = pl.buf(R, out)
            pl.emit(H.op_gemm(0, 1, R, out, Kh, h, Kh, w, K, u, out, bias=b))
            if i >= 1:
                pl.emit(H.op_gemm(0, 1, R, out, ic, x, ic, w[:, Kh:], K, u, out, accumulate=1))
            L = NS(i=i, p=p, h_in=h, u=u, out=out, K=K, Kh=Kh, last=(i == n - 1))
            if not L.last:
                e, le = B.linear(ca, p + ".linear_emb")
                norm = 1 if cfg["use_norm"] else 0
                mean = pl.buf(R) if norm else None
                rstd = pl.buf(R) if norm else None
                y = pl.buf(R, out)
                g = P.get(p + ".norm.weight"); bt = P.get(p + ".norm.bias")
                pl.emit(H.op_mlp_modln_fwd(u, e, g, bt, R, out, norm, 1, LN_EPS, y, mean, rstd))
                L.e, L.le, L.mean, L.rstd, L.y, L.norm = e, le, mean, rstd, y, norm
                if not B.save:
                    pl.free(u, e, mean, rstd)
                    if i >= 1:
                        pl.free(h)
                h = y
            else:
                if not B.save and i >= 1:
                    pl.free(h)
                h = u
            fx.layers.append(L)
        fx.out = h
        return fx

    def _emit_backward(self, B, fx, d_out):
        """d_out: gradient of the network output [R][ic]; accumulates every parameter gradient, returns dL/dx."""
        cfg, pl, P, Gr = self.cfg, B.p, self.P, B.Gr
        ic, R = cfg["input_channel"], fx.R
        dx = pl.buf(R, ic, zero=False)
        pl.emit(H.op_memset(dx, dx.numel() * 4))
        dca = pl.buf(*fx.ca.shape)
        pl.emit(H.op_memset(dca, dca.numel() * 4))
        dy = d_out
        for L in reversed(fx.layers):
            w = P[L.p + ".linear.weight"]
            if L.last:
                du = dy
            else:
                du, de = pl.buf(R, L.out), pl.buf(R, L.out)
                tg = pl.buf(R, L.out) if L.norm else None
                tb = pl.buf(R, L.out) if L.norm else None
                g = P.get(L.p + ".norm.weight"); bt = P.get(L.p + ".norm.bias")
                pl.emit(H.op_mlp_modln_bwd(L.u, L.e, g, bt, L.mean, L.rstd, dy, R, L.out, L.norm, 1, du, de, tg, tb))
                if L.norm:
                    pl.need_ws(H.colsum_ws_bytes(R, L.out))
                    pl.emit(H.op_colsum(tg, R, L.out, Gr[L.p + ".norm.weight"], None, acc=B.acc), ws_slot=2)
                    pl.emit(H.op_colsum(tb, R, L.out, Gr[L.p + ".norm.bias"], None, acc=B.acc), ws_slot=2)
                    pl.free(tg, tb)
                B.linear_bwd(L.le, de, dx=dca, dx_acc=1)        # e = Linear(ca): dWe, dbe, d ca +=
                pl.free(de)
                if dy is not d_out:
                    pl.free(dy)
            # u = [h | x] W^T + b
            gw, gb = Gr[L.p + ".linear.weight"], Gr[L.p + ".linear.bias"]
            pl.emit(H.op_gemm(1, 0, L.out, L.Kh, R, du, L.out, L.h_in, L.Kh, gw, L.K, accumulate=B.acc))
            if L.i >= 1:
                pl.emit(H.op_gemm(1, 0, L.out, ic, R, du, L.out, fx.x, ic, gw[:, L.Kh:], L.K, accumulate=B.acc))
                pl.emit(H.op_gemm(0, 0, R, ic, L.out, du, L.out, w[:, L.Kh:], L.K, dx, ic, accumulate=1))
            pl.need_ws(H.colsum_ws_bytes(R, L.out))
            pl.emit(H.op_colsum(du, R, L.out, gb, None, acc=B.acc), ws_slot=2)
            if L.i >= 1:
                dh = pl.buf(R, L.Kh)
                pl.emit(H.op_gemm(0, 0, R, L.Kh, L.out, du, L.out, w, L.K, dh, L.Kh))
            else:
                pl.emit(H.op_gemm(0, 0, R, L.Kh, L.out, du, L.out, w, L.K, dx, ic, accumulate=1))
                dh = None
            if du is not d_out:
                pl.free(du)
            dy = dh
        # cond path: ca = silu(cond); cond = Linear(silu(c0)); c0 = Linear(te)
        dcond = pl.buf(*fx.cond.shape)
        pl.emit(H.op_silu_bwd(fx.cond, dca, dcond, dcond.numel()))
        dc0s = pl.buf(*fx.c0s.shape)
        B.linear_bwd(fx.l2, dcond, dx=dc0s)
        dc0 = pl.buf(*fx.c0.shape)
        pl.emit(H.op_silu_bwd(fx.c0, dc0s, dc0, dc0.numel()))
        B.linear_bwd(fx.l0, dc0)
        pl.free(dca, dcond, dc0s, dc0)
        return dx

    def plan(self, R, train):
        key = (R, bool(train))
        pl = self._plans.get(key)
        if pl is not None:
            return pl
        p = Plan(self.device)
        x = p.buf(R, self.cfg["input_channel"])
        t = p.buf(R, dtype=torch.int64)
        B = Builder(p, self.P, self.grads() if train else None, save=bool(train), acc_grads=bool(train))
        fx = self._emit_forward(B, x, t)
        p.n_fwd = len(p.recs)
        p.d_out = p.dx = None
        if train:
            p.d_out = p.buf(R, self.cfg["input_channel"])
            p.dx = self._emit_backward(B, fx, p.d_out)
        p.x, p.t, p.out = x, t, fx.out
        p.compile()
        self._plans[key] = p
        return p

    # ------------------------------------------------------------------ forward (mlp_skip_net.py:67-78)
    def forward(self, x, t, condition=None):
        R = x.shape[0]
        train = self._wants_grad()
        p = self.plan(R, train)

        def run_fwd(*_):
            p.x.copy_(x)
            p.t.copy_(t)
            p.run(0, p.n_fwd)
            return (p.out.clone(),)

        if not train:
            return run_fwd()[0]

        def run_bwd(d_out):
            p.d_out.copy_(d_out)
            p.run(p.n_fwd, p.n)
            return (p.dx.clone(),) if x.requires_grad else ()

        if x.requires_grad:
            return self._bridge(p, run_fwd, run_bwd, 1, x)
        return self._bridge(p, run_fwd, run_bwd, 1)
