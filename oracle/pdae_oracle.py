"""CPU oracle for the PDAE hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional restatement (torch fp32/fp64 on CPU) of the
reference algorithm for the path named in BASELINE.json.  It is imported ONLY by
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`; the
product package `pdae_amd/` never imports it and fails loudly when the HIP library
is missing.

Parity status: the reference ships no tests / golden vectors of its own (SURVEY.md
section 8c), so with respect to the reference's *own* tests parity is unpinned; the
oracle is instead pinned against outputs of the reference itself, imported in the
build container (`tests/golden/make_fixtures.py` -> `tests/golden/*.npz`, checked by
`tests/test_oracle_golden.py`).

Every function cites the reference file:line it restates (paths relative to the
reference repo root).  Networks are evaluated *functionally* from a state-dict that
uses the reference's key names, so a reference checkpoint can be fed in unchanged.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------
# building blocks (model/module.py)
# ----------------------------------------------------------------------------------
def timestep_embedding(t, dim, max_period=10000):
    """model/module.py:66-84 -- [cos(t f), sin(t f)], f_i = exp(-ln(P) i / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(x, sd, pre):
    """model/module.py:56-63 -- GroupNorm(32, C), eps 1e-5, affine."""
    return F.group_norm(x, 32, sd[pre + ".weight"], sd[pre + ".bias"], 1e-5)


def _conv(x, sd, pre, stride=1):
    w = sd[pre + ".weight"]
    return F.conv2d(x, w, sd[pre + ".bias"], stride=stride, padding=w.shape[-1] // 2)


# Dropout (module.py:263, nn.Dropout between SiLU and the second conv of out_layers): RNG streams cannot match across devices, so parity of
# the dropout-ON path is defined for INJECTED masks: DROP_MASKS maps a ResBlock prefix to (keep mask [N,C,H,W] of 0/1, p) -- the masks the
# device path actually drew -- and the block computes silu(h) * mask / (1 - p) like nn.Dropout in training mode.  Empty: identity (p = 0 / eval).
DROP_MASKS = {}


def resblock(sd, pre, x, emb, emb_z=None, up=False, down=False):
    """model/module.py:278-297 (ResBlock.forward) and :361-384 (ResBlockShift.forward).

    Dropout is identity (parity is defined at p=0 / eval) unless a keep mask was injected for this block (DROP_MASKS)."""
    h = F.silu(_gn(x, sd, pre + ".in_layers.0"))
    if up:          # module.py:279-284: resample between GN-SiLU and the conv, and the skip too
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif down:
        h = F.avg_pool2d(h, 2)
        x = F.avg_pool2d(x, 2)
    h = _conv(h, sd, pre + ".in_layers.2")
    e = F.linear(F.silu(emb), sd[pre + ".emb_layers.1.weight"], sd[pre + ".emb_layers.1.bias"])
    scale, shift = torch.chunk(e[:, :, None, None], 2, dim=1)
    h = _gn(h, sd, pre + ".out_layers.0") * (1.0 + scale) + shift            # module.py:293-294
    if emb_z is not None:                                                    # module.py:371-381
        ez = F.linear(F.silu(emb_z), sd[pre + ".emb_z_layers.1.weight"], sd[pre + ".emb_z_layers.1.bias"])
        z_scale, z_shift = torch.chunk(ez[:, :, None, None], 2, dim=1)
        h = (1.0 + z_scale) * h + z_shift
    h = F.silu(h)
    if pre in DROP_MASKS:
        keep, p_drop = DROP_MASKS[pre]
        h = h * keep / (1.0 - p_drop)
    h = _conv(h, sd, pre + ".out_layers.3")
    if pre + ".skip_connection.weight" in sd:
        x = _conv(x, sd, pre + ".skip_connection")
    return x + h


def qkv_attention(qkv, n_heads, new_order):
    """model/module.py:431-457 (legacy: heads split first) / :460-488 (new order)."""
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    scale = 1.0 / math.sqrt(math.sqrt(ch))
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q = q.reshape(bs * n_heads, ch, length)
        k = k.reshape(bs * n_heads, ch, length)
        v = v.reshape(bs * n_heads, ch, length)
    else:
        q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w, dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v)
    return a.reshape(bs, -1, length)


def attention_block(sd, pre, x, n_heads, new_order=False):
    """model/module.py:422-428 -- x + proj(attn(qkv(GN(x))))."""
    b, c = x.shape[:2]
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(xf, sd, pre + ".norm"), sd[pre + ".qkv.weight"], sd[pre + ".qkv.bias"])
    h = qkv_attention(qkv, n_heads, new_order)
    h = F.conv1d(h, sd[pre + ".proj_out.weight"], sd[pre + ".proj_out.bias"])
    return (xf + h).reshape(x.shape)


def _heads(cfg, ch):
    """model/module.py:402-409."""
    hc = cfg.get("head_channel", -1)
    return cfg.get("num_heads", 1) if hc == -1 else ch // hc


# ----------------------------------------------------------------------------------
# topology (model/unet.py:60-175, model/shift_unet.py:65-249)
# ----------------------------------------------------------------------------------
def unet_topology(cfg):
    """Walks the config exactly like UNet.__init__ and returns
    (input_blocks, middle, output_blocks); each block is a list of layer descriptors
    ("conv"|"res"|"attn", dict)."""
    base = cfg["base_channel"]
    mult = cfg["channel_multiplier"]
    nres = cfg["num_residual_blocks_of_a_block"]
    attn_res = list(cfg["attention_resolutions"])
    ch = int(mult[0] * base)
    inputs = [[("conv", dict(cin=cfg["input_channel"], cout=ch))]]
    chans = [ch]
    ds = 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            layers = [("res", dict(cin=ch, cout=int(m * base)))]
            ch = int(m * base)
            if ds in attn_res:
                layers.append(("attn", dict(ch=ch)))
            inputs.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            inputs.append([("res", dict(cin=ch, cout=ch, down=True))])
            chans.append(ch)
            ds *= 2
    middle = [("res", dict(cin=ch, cout=ch)), ("attn", dict(ch=ch)), ("res", dict(cin=ch, cout=ch))]
    outputs = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nres + 1):
            ich = chans.pop()
            layers = [("res", dict(cin=ch + ich, cout=int(base * m)))]
            ch = int(base * m)
            if ds in attn_res:
                layers.append(("attn", dict(ch=ch)))
            if level and i == nres:
                layers.append(("res", dict(cin=ch, cout=ch, up=True)))
                ds //= 2
            outputs.append(layers)
    return inputs, middle, outputs


def _run_block(sd, pre, layers, h, emb, emb_z, cfg):
    for j, (kind, d) in enumerate(layers):
        p = f"{pre}.{j}"
        if kind == "conv":
            h = _conv(h, sd, p)
        elif kind == "res":
            h = resblock(sd, p, h, emb, emb_z, up=d.get("up", False), down=d.get("down", False))
        else:
            h = attention_block(sd, p, h, _heads(cfg, d["ch"]), cfg.get("use_new_attention_order", False))
    return h


def _time_embed(sd, cfg, t):
    e = timestep_embedding(t, cfg["base_channel"])
    e = F.linear(e, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    return F.linear(F.silu(e), sd["time_embed.2.weight"], sd["time_embed.2.bias"])


def _out_head(sd, pre, h):
    return _conv(F.silu(_gn(h, sd, pre + ".0")), sd, pre + ".2")


def unet_forward(sd, cfg, x, t, condition=None):
    """model/unet.py:177-202."""
    inputs, middle, outputs = unet_topology(cfg)
    emb = _time_embed(sd, cfg, t)
    if cfg.get("num_class") is not None:
        emb = emb + sd["label_emb.weight"][condition]
    hs = []
    h = x
    for i, layers in enumerate(inputs):
        h = _run_block(sd, f"input_blocks.{i}", layers, h, emb, None, cfg)
        hs.append(h)
    h = _run_block(sd, "middle_block", middle, h, emb, None, cfg)
    for i, layers in enumerate(outputs):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(sd, f"output_blocks.{i}", layers, h, emb, None, cfg)
    return _out_head(sd, "out", h)


def shift_unet_forward(sd, cfg, x, t, z):
    """model/shift_unet.py:253-284 -> (eps, shift)."""
    inputs, middle, outputs = unet_topology(cfg)
    emb = _time_embed(sd, cfg, t)
    shift_emb = F.linear(z, sd["label_emb.weight"], sd["label_emb.bias"])
    hs = []
    h = x
    for i, layers in enumerate(inputs):
        h = _run_block(sd, f"input_blocks.{i}", layers, h, emb, None, cfg)
        hs.append(h)
    eps_h = _run_block(sd, "middle_block", middle, h, emb, None, cfg)
    shift_h = _run_block(sd, "shift_middle_block", middle, h, emb, shift_emb, cfg)
    for i, layers in enumerate(outputs):
        prev = hs.pop()
        eps_h = _run_block(sd, f"output_blocks.{i}", layers, torch.cat([eps_h, prev], 1), emb, None, cfg)
        shift_h = _run_block(sd, f"shift_output_blocks.{i}", layers, torch.cat([shift_h, prev], 1), emb, shift_emb, cfg)
    return _out_head(sd, "out", eps_h), _out_head(sd, "shift_out", shift_h)


def shift_unet_trainable(key):
    """model/shift_unet.py:299-310 + trainer/train_representation_learning.py:58-66."""
    return key.startswith(("label_emb.", "shift_middle_block.", "shift_output_blocks.", "shift_out."))


# encoders: model/representation_learning/encoder/{ffhq,celeba64,...}.py
ENCODER_SPECS = {
    # name: (conv channels, index of the conv after which attention sits, flat dim)
    "FFHQEncoder": ([3, 64, 128, 256, 256, 256], 2),
    "CELEBAHQEncoder": ([3, 64, 128, 256, 256, 256], 2),
    "BEDROOMEncoder": ([3, 64, 128, 256, 256, 256], 2),
    "HORSEEncoder": ([3, 64, 128, 256, 256, 256], 2),
    "CELEBA64Encoder": ([3, 64, 128, 128, 128], 1),
}


def encoder_layout(name):
    """Returns the nn.Sequential index of every layer, as in encoder/ffhq.py:10-36."""
    chans, attn_after = ENCODER_SPECS[name]
    layers = []
    idx = 0
    nconv = len(chans) - 1
    for i in range(nconv):
        layers.append(("conv", idx, chans[i], chans[i + 1])); idx += 1
        if i == attn_after:
            layers.append(("attn", idx, chans[i + 1])); idx += 1
        layers.append(("gn", idx, chans[i + 1])); idx += 2        # GN + SiLU
    # after last GN/SiLU: View, Linear
    layers.append(("linear", idx + 1, chans[-1] * 16))
    return layers


def encoder_forward(sd, name, x):
    """encoder/ffhq.py:39-41 (and celeba64.py:35-37)."""
    h = x
    for l in encoder_layout(name):
        p = f"encoder.{l[1]}"
        if l[0] == "conv":
            h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], stride=2, padding=1)
        elif l[0] == "attn":
            h = attention_block(sd, p, h, 4, False)
        elif l[0] == "gn":
            h = F.silu(_gn(h, sd, p))
        else:
            h = F.linear(h.reshape(h.shape[0], -1), sd[p + ".weight"], sd[p + ".bias"])
    return h


def mlp_skip_net_forward(sd, cfg, x, t):
    """model/mlp_skip_net.py:67-78, 123-141."""
    n = cfg["num_layers"]
    te = timestep_embedding(t, cfg["time_emb_channel"])
    cond = F.linear(te, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    cond = F.linear(F.silu(cond), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    h = x
    for i in range(n):
        if i >= 1:
            h = torch.cat([h, x], dim=1)
        p = f"layers.{i}"
        h = F.linear(h, sd[p + ".linear.weight"], sd[p + ".linear.bias"])
        last = i == n - 1
        if not last:
            c = F.linear(F.silu(cond), sd[p + ".linear_emb.weight"], sd[p + ".linear_emb.bias"])
            h = h * (1.0 + c)
            if cfg.get("use_norm", True):
                h = F.layer_norm(h, (h.shape[1],), sd[p + ".norm.weight"], sd[p + ".norm.bias"])
            h = F.silu(h)
    return h


# ----------------------------------------------------------------------------------
# diffusion (diffusion/gaussian_diffusion.py, diffusion/ddim.py)
# ----------------------------------------------------------------------------------
class Schedules:
    """diffusion/gaussian_diffusion.py:12-70; float64 numpy, cast to fp32 tensors."""

    def __init__(self, timesteps=1000, betas_type="linear"):
        self.timesteps = timesteps
        if betas_type == "linear":
            betas = np.linspace(0.0001, 0.02, timesteps)
        elif betas_type == "cosine":
            ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
            betas = np.array([min(1 - ab((i + 1) / timesteps) / ab(i / timesteps), 0.999) for i in range(timesteps)])
        else:
            raise NotImplementedError
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        ac_next = np.append(ac[1:], 0.0)
        f = lambda a: torch.tensor(a, dtype=torch.float32)
        self.np_alphas_cumprod = ac
        self.alphas, self.betas = f(alphas), f(betas)
        self.alphas_cumprod, self.alphas_cumprod_prev, self.alphas_cumprod_next = f(ac), f(ac_prev), f(ac_next)
        self.sqrt_alphas_cumprod = f(np.sqrt(ac))
        self.sqrt_one_minus_alphas_cumprod = f(np.sqrt(1.0 - ac))
        self.log_one_minus_alphas_cumprod = f(np.log(1.0 - ac))
        self.sqrt_recip_alphas_cumprod = f(np.sqrt(1.0 / ac))
        self.sqrt_recip_alphas_cumprod_m1 = f(np.sqrt(1.0 / ac - 1.0))
        pv = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.posterior_variance = f(pv)
        self.posterior_log_variance_clipped = f(np.log(np.append(pv[1], pv[1:])))
        self.x_0_posterior_mean_x_0_coef = f(betas * np.sqrt(ac_prev) / (1.0 - ac))
        self.x_0_posterior_mean_x_t_coef = f((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac))
        self.noise_posterior_mean_x_t_coef = f(np.sqrt(1.0 / alphas))
        self.noise_posterior_mean_noise_coef = f(betas / (np.sqrt(alphas) * np.sqrt(1.0 - ac)))
        self.shift_coef = f(-np.sqrt(alphas) * (1.0 - ac_prev) / np.sqrt(1.0 - ac))       # :65
        snr = ac / (1.0 - ac)
        self.weight = f(snr ** 0.1 / (1.0 + snr))                                         # :68-70


def _at(table, t, x):
    """gaussian_diffusion.py:72-74."""
    return torch.gather(table, -1, t).reshape([x.shape[0]] + [1] * (x.dim() - 1))


def q_sample(s, x0, t, noise):
    """gaussian_diffusion.py:98-103."""
    return _at(s.sqrt_alphas_cumprod, t, x0) * x0 + _at(s.sqrt_one_minus_alphas_cumprod, t, x0) * noise


def p_loss(noise, pred, weight=None, loss_type="l2"):
    """gaussian_diffusion.py:166-175."""
    if loss_type == "l1":
        return (noise - pred).abs().mean()
    if weight is not None:
        return torch.mean(weight * (noise - pred) ** 2)
    return torch.mean((noise - pred) ** 2)


def regular_loss(s, sd, cfg, x0, t, noise, condition=None):
    """gaussian_diffusion.py:199-211 with t/noise injected."""
    return p_loss(noise, unet_forward(sd, cfg, q_sample(s, x0, t, noise), t, condition))


def rl_loss(s, enc_sd, enc_name, dec_sd, cfg, x0, t, noise):
    """gaussian_diffusion.py:234-255 with t/noise injected."""
    z = encoder_forward(enc_sd, enc_name, x0)
    x_t = q_sample(s, x0, t, noise)
    eps, grad = shift_unet_forward(dec_sd, cfg, x_t, t, z)
    return p_loss(noise, eps + _at(s.shift_coef, t, x0) * grad, weight=_at(s.weight, t, x0))


def ddim_betas_and_map(style, alphas_cumprod_np):
    """gaussian_diffusion.py:76-94."""
    T = alphas_cumprod_np.shape[0]
    n = int(style[len("ddim"):])
    use = set(int(v) for v in list(np.linspace(0, T - 1, n + 1)))
    last, betas, tmap = 1.0, [], []
    for i, a in enumerate(alphas_cumprod_np):
        if i in use:
            betas.append(1 - a / last)
            last = a
            tmap.append(i)
    return np.array(betas), torch.tensor(tmap, dtype=torch.long)


class DDIMTables:
    """diffusion/ddim.py:8-33.  NB the reference feeds float32 alphas_cumprod
    (self.alphas_cumprod.cpu().numpy(), gaussian_diffusion.py:276) into float64 math."""

    def __init__(self, s, style):
        betas, self.timestep_map = ddim_betas_and_map(style, s.alphas_cumprod.numpy())
        self.timesteps = betas.shape[0] - 1
        ac = np.cumprod(1.0 - betas, axis=0)
        f = lambda a: torch.tensor(a, dtype=torch.float32)
        self.alphas_cumprod_prev = f(np.append(1.0, ac[:-1]))
        self.alphas_cumprod_next = f(np.append(ac[1:], 0.0))
        self.sqrt_one_minus_alphas_cumprod = f(np.sqrt(1.0 - ac))
        self.sqrt_recip_alphas_cumprod = f(np.sqrt(1.0 / ac))
        self.sqrt_recip_alphas_cumprod_m1 = f(np.sqrt(1.0 / ac - 1.0))


def ddim_update(d, x_t, t, eps, grad=None, encode=False, use_shift=True):
    """diffusion/ddim.py:94-107 (sample) / :126-138 (encode); grad=None gives :46-55 / :69-79."""
    if grad is not None and use_shift:
        eps = eps - _at(d.sqrt_one_minus_alphas_cumprod, t, x_t) * grad
    ra, rm1 = _at(d.sqrt_recip_alphas_cumprod, t, x_t), _at(d.sqrt_recip_alphas_cumprod_m1, t, x_t)
    x0 = (ra * x_t - rm1 * eps).clamp(-1, 1)
    new_eps = (ra * x_t - x0) / rm1
    ab = _at(d.alphas_cumprod_next if encode else d.alphas_cumprod_prev, t, x_t)
    return x0 * torch.sqrt(ab) + torch.sqrt(1.0 - ab) * new_eps


def shift_ddim_sample_loop(s, style, dec_sd, cfg, z, x_T, stop_percent=0.0, trajectory=None):
    """diffusion/ddim.py:110-120."""
    d = DDIMTables(s, style)
    stop = int(stop_percent * d.timesteps)
    x = x_T
    for i in reversed(range(1, d.timesteps + 1)):
        t = torch.full((x.shape[0],), i, dtype=torch.long)
        eps, g = shift_unet_forward(dec_sd, cfg, x, d.timestep_map[t], z)
        x = ddim_update(d, x, t, eps, g, encode=False, use_shift=(i - 1) >= stop)
        if trajectory is not None:
            trajectory.append(x.clone())
    return x


def shift_ddim_encode_loop(s, style, dec_sd, cfg, z, x_0, trajectory=None):
    """diffusion/ddim.py:140-147."""
    d = DDIMTables(s, style)
    x = x_0
    for i in range(0, d.timesteps):
        t = torch.full((x.shape[0],), i, dtype=torch.long)
        eps, g = shift_unet_forward(dec_sd, cfg, x, d.timestep_map[t], z)
        x = ddim_update(d, x, t, eps, g, encode=True)
        if trajectory is not None:
            trajectory.append(x.clone())
    return x


def ddim_sample_loop(s, style, fn, x_T):
    """diffusion/ddim.py:57-64; fn(x, t_mapped) -> eps."""
    d = DDIMTables(s, style)
    x = x_T
    for i in reversed(range(1, d.timesteps + 1)):
        t = torch.full((x.shape[0],), i, dtype=torch.long)
        x = ddim_update(d, x, t, fn(x, d.timestep_map[t]))
    return x


def ddim_encode_loop(s, style, fn, x_0):
    """diffusion/ddim.py:81-88."""
    d = DDIMTables(s, style)
    x = x_0
    for i in range(0, d.timesteps):
        t = torch.full((x.shape[0],), i, dtype=torch.long)
        x = ddim_update(d, x, t, fn(x, d.timestep_map[t]), encode=True)
    return x


def rl_autoencoding(s, enc_style, dec_style, enc_sd, enc_name, dec_sd, cfg, x_0):
    """gaussian_diffusion.py:287-290."""
    z = encoder_forward(enc_sd, enc_name, x_0)
    x_T = shift_ddim_encode_loop(s, enc_style, dec_sd, cfg, z, x_0)
    return shift_ddim_sample_loop(s, dec_style, dec_sd, cfg, z, x_T), x_T, z


def noise_p_sample_mean(s, x_t, t, eps):
    """gaussian_diffusion.py:112-116 (mean part; the added noise is an injected input)."""
    return _at(s.noise_posterior_mean_x_t_coef, t, x_t) * x_t - _at(s.noise_posterior_mean_noise_coef, t, x_t) * eps


def q_posterior_mean(s, x_0, x_t, t):
    """gaussian_diffusion.py:105-108."""
    return _at(s.x_0_posterior_mean_x_0_coef, t, x_t) * x_0 + _at(s.x_0_posterior_mean_x_t_coef, t, x_t) * x_t


def predicted_x_0(s, x_t, t, eps):
    """gaussian_diffusion.py:156-159."""
    return _at(s.sqrt_recip_alphas_cumprod, t, x_t) * x_t - _at(s.sqrt_recip_alphas_cumprod_m1, t, x_t) * eps


def learned_range_to_log_variance(s, v, t):
    """gaussian_diffusion.py:148-154: v in [-1, 1] interpolates between the clipped posterior log-variance and log(beta)."""
    lo, hi = _at(s.posterior_log_variance_clipped, t, v), _at(torch.log(s.betas), t, v)
    return lo + (v + 1) / 2 * (hi - lo)


def noise_p_sample(s, x_t, t, eps, noise, learned_range=None):
    """gaussian_diffusion.py:112-126 with the internally drawn noise injected."""
    lv = learned_range_to_log_variance(s, learned_range, t) if learned_range is not None else _at(s.posterior_log_variance_clipped, t, x_t)
    mask = (t != 0).float().reshape([-1] + [1] * (x_t.dim() - 1))
    return noise_p_sample_mean(s, x_t, t, eps) + mask * (0.5 * lv).exp() * noise


def regular_ddpm_sample(s, fn, x_T, noise_at):
    """gaussian_diffusion.py:216-229; fn(x, t) -> eps or [eps | range]; noise_at(i) = the draw of step i."""
    c = x_T.shape[1]
    img = x_T
    for i in reversed(range(s.timesteps)):
        t = torch.full((x_T.shape[0],), i, dtype=torch.long)
        o = fn(img, t)
        eps, vr = (o[:, :c], o[:, c:]) if o.shape[1] == 2 * c else (o, None)
        img = noise_p_sample(s, img, t, eps, noise_at(i), vr)
    return img


def rl_ddpm_sample(s, dec_sd, cfg, z, x_T, noise_at):
    """gaussian_diffusion.py:257-270: the ancestral sampler on eps + shift_coef[t] * gradient."""
    img = x_T
    for i in reversed(range(s.timesteps)):
        t = torch.full((x_T.shape[0],), i, dtype=torch.long)
        eps, g = shift_unet_forward(dec_sd, cfg, img, t, z)
        img = noise_p_sample(s, img, t, eps + _at(s.shift_coef, t, img) * g, noise_at(i))
    return img


def rl_two_x_0(s, dec_sd, cfg, z, x_t, t):
    """The pair (x_0 from eps, x_0 from eps + shift_coef * gradient) of gaussian_diffusion.py:305-311 / 326-332."""
    eps, g = shift_unet_forward(dec_sd, cfg, x_t, t, z)
    return predicted_x_0(s, x_t, t, eps), predicted_x_0(s, x_t, t, eps + _at(s.shift_coef, t, x_t) * g)


def rl_gap_measure(s, dec_sd, cfg, z, x_0, noise_at):
    """gaussian_diffusion.py:292-318 (noise_at(i) replaces the UNIFORM rand_like draw of :302)."""
    gp, ga = [], []
    for i in reversed(range(s.timesteps)):
        t = torch.full((x_0.shape[0],), i, dtype=torch.long)
        x_t = q_sample(s, x_0, t, noise_at(i))
        a, b = rl_two_x_0(s, dec_sd, cfg, z, x_t, t)
        truth = q_posterior_mean(s, x_0, x_t, t)
        gp.append(float(torch.mean((truth - q_posterior_mean(s, a, x_t, t)) ** 2)))
        ga.append(float(torch.mean((truth - q_posterior_mean(s, b, x_t, t)) ** 2)))
    return gp, ga


def shift_ddim_trajectory_interpolation(s, style, dec_sd, cfg, z_1, z_2, x_T, alpha):
    """diffusion/ddim.py:149-174."""
    d = DDIMTables(s, style)
    x = x_T
    for i in reversed(range(1, d.timesteps + 1)):
        t = torch.full((x.shape[0],), i, dtype=torch.long)
        eps, g1 = shift_unet_forward(dec_sd, cfg, x, d.timestep_map[t], z_1)
        _, g2 = shift_unet_forward(dec_sd, cfg, x, d.timestep_map[t], z_2)
        x = ddim_update(d, x, t, eps, (1.0 - alpha) * g1 + alpha * g2)
    return x


def manipulated_latent(z, classifier_weight, class_id, scale, mean, std):
    """gaussian_diffusion.py:435-441."""
    zn = (z - mean) / std
    zn = zn + scale * math.sqrt(512) * F.normalize(classifier_weight[class_id][None, :], dim=1)
    return zn * std + mean


# ----------------------------------------------------------------------------------
# optimizer / EMA (trainer/train_representation_learning.py:58-70, 192-212)
# ----------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, weight_decay=0.0, decoupled=False):
    """torch.optim.Adam / AdamW single-tensor math (step is 1-based)."""
    if decoupled:
        p = p * (1.0 - lr * weight_decay)
    elif weight_decay != 0.0:
        g = g + weight_decay * p
    m = m * b1 + (1.0 - b1) * g
    v = v * b2 + (1.0 - b2) * g * g
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def ema_update(ema, p, decay):
    """train_representation_learning.py:201,212."""
    return ema * decay + p * (1.0 - decay)


# ----------------------------------------------------------------------------------
# metrics (metric/utils.py:35-63)
# ----------------------------------------------------------------------------------
def ssim(img1, img2, window_size=11):
    c = img1.shape[1]
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    w = g.mm(g.t()).float()[None, None].expand(c, 1, window_size, window_size).contiguous()
    pad = window_size // 2
    mu1 = F.conv2d(img1, w, padding=pad, groups=c)
    mu2 = F.conv2d(img2, w, padding=pad, groups=c)
    s1 = F.conv2d(img1 * img1, w, padding=pad, groups=c) - mu1 * mu1
    s2 = F.conv2d(img2 * img2, w, padding=pad, groups=c) - mu2 * mu2
    s12 = F.conv2d(img1 * img2, w, padding=pad, groups=c) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    return m.mean(1).mean(1).mean(1)


def mse(img1, img2):
    return (img1 - img2).pow(2).mean(dim=[1, 2, 3])


# ----------------------------------------------------------------------------------
# deterministic synthetic weights (no reference needed): used by fixtures, tests, bench
# ----------------------------------------------------------------------------------
def unet_param_shapes(cfg, shift=False, latent_dim=None):
    """Key -> shape for UNet / ShiftUNet, in the reference's registration order
    (unet.py:51-175 / shift_unet.py:52-249)."""
    inputs, middle, outputs = unet_topology(cfg)
    ted = cfg["base_channel"] * 4
    out_ch = cfg["input_channel"] * (2 if cfg.get("learn_sigma", False) else 1)
    sh = OrderedDict()

    def lin(p, i, o):
        sh[p + ".weight"] = (o, i); sh[p + ".bias"] = (o,)

    def conv(p, i, o, k=3):
        sh[p + ".weight"] = (o, i, k, k); sh[p + ".bias"] = (o,)

    def gn(p, c):
        sh[p + ".weight"] = (c,); sh[p + ".bias"] = (c,)

    def res(p, d, zed):
        gn(p + ".in_layers.0", d["cin"]); conv(p + ".in_layers.2", d["cin"], d["cout"])
        lin(p + ".emb_layers.1", ted, 2 * d["cout"])
        if zed:
            lin(p + ".emb_z_layers.1", ted, 2 * d["cout"])
        gn(p + ".out_layers.0", d["cout"]); conv(p + ".out_layers.3", d["cout"], d["cout"])
        if d["cin"] != d["cout"]:
            conv(p + ".skip_connection", d["cin"], d["cout"], 1)

    def attn(p, c):
        gn(p + ".norm", c)
        sh[p + ".qkv.weight"] = (3 * c, c, 1); sh[p + ".qkv.bias"] = (3 * c,)
        sh[p + ".proj_out.weight"] = (c, c, 1); sh[p + ".proj_out.bias"] = (c,)

    def block(pre, layers, zed):
        for j, (kind, d) in enumerate(layers):
            p = f"{pre}.{j}"
            if kind == "conv":
                conv(p, d["cin"], d["cout"])
            elif kind == "res":
                res(p, d, zed)
            else:
                attn(p, d["ch"])

    lin("time_embed.0", cfg["base_channel"], ted); lin("time_embed.2", ted, ted)
    if shift:
        lin("label_emb", latent_dim, ted)
    elif cfg.get("num_class") is not None:
        sh["label_emb.weight"] = (cfg["num_class"], ted)
    for i, l in enumerate(inputs):
        block(f"input_blocks.{i}", l, False)
    block("middle_block", middle, False)
    if shift:
        block("shift_middle_block", middle, True)
    for i, l in enumerate(outputs):
        block(f"output_blocks.{i}", l, False)
    if shift:
        for i, l in enumerate(outputs):
            block(f"shift_output_blocks.{i}", l, True)
    c0 = int(cfg["channel_multiplier"][0] * cfg["base_channel"])
    gn("out.0", c0); conv("out.2", c0, out_ch)
    if shift:
        gn("shift_out.0", c0); conv("shift_out.2", c0, cfg["input_channel"])
    return sh


def encoder_param_shapes(name, latent_dim):
    sh = OrderedDict()
    for l in encoder_layout(name):
        p = f"encoder.{l[1]}"
        if l[0] == "conv":
            sh[p + ".weight"] = (l[3], l[2], 3, 3); sh[p + ".bias"] = (l[3],)
        elif l[0] == "attn":
            c = l[2]
            sh[p + ".norm.weight"] = (c,); sh[p + ".norm.bias"] = (c,)
            sh[p + ".qkv.weight"] = (3 * c, c, 1); sh[p + ".qkv.bias"] = (3 * c,)
            sh[p + ".proj_out.weight"] = (c, c, 1); sh[p + ".proj_out.bias"] = (c,)
        elif l[0] == "gn":
            sh[p + ".weight"] = (l[2],); sh[p + ".bias"] = (l[2],)
        else:
            sh[p + ".weight"] = (latent_dim, l[2]); sh[p + ".bias"] = (latent_dim,)
    return sh


def mlp_skip_net_param_shapes(cfg):
    """model/mlp_skip_net.py:27-66 (state-dict also aliases cond_layers.1 == linear_emb)."""
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
            if cfg.get("use_norm", True):
                sh[p + ".norm.weight"] = (b,); sh[p + ".norm.bias"] = (b,)
    return sh


def synth_state_dict(shapes, seed):
    """Deterministic non-degenerate weights: every zero-initialised layer of the
    reference (module.py:48-54) is randomised too, otherwise outputs are vacuous
    (SURVEY 0.2).  numpy PCG64 streams are stable across machines."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for k, shp in shapes.items():
        if k.endswith(".bias"):
            a = 0.1 * rng.standard_normal(shp)
        elif len(shp) == 1:                       # GroupNorm / LayerNorm gamma
            a = 1.0 + 0.1 * rng.standard_normal(shp)
        else:
            fan_in = int(np.prod(shp[1:]))
            a = rng.standard_normal(shp) / math.sqrt(fan_in)
        sd[k] = torch.tensor(a, dtype=torch.float32)
    return sd


# ----------------------------------------------------------------------------------
# input pipeline (dataset/ffhq.py:19-31,46; dataset/celeba64.py:11-34).  The resize lives in a third-party dependency that is not vendored
# in the reference tree: Pillow (requirements.txt: unpinned; here 10.x), called through torchvision.transforms.Resize on PIL images =
# Image.resize(size, BILINEAR).  Restated from Pillow's published algorithm (src/libImaging/Resample.c: precompute_coeffs,
# normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc) and pinned in tests/test_oracle_golden.py against Pillow itself.
# ----------------------------------------------------------------------------------
def pil_bilinear_coeffs(in_size, out_size, bits=22):
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = fs
    rows = []
    for o in range(out_size):
        c = (o + 0.5) * scale
        lo, hi = max(int(c - support + 0.5), 0), min(int(c + support + 0.5), in_size)
        w = np.array([max(0.0, 1.0 - abs((k + lo - c + 0.5) / fs)) for k in range(hi - lo)])
        w = w / w.sum() if w.sum() != 0 else w
        rows.append((lo, np.array([int(v * (1 << bits) + 0.5) for v in w], dtype=np.int64)))
    return rows


def resize_u8(img, size, crop=None, bits=22):
    """uint8 [H,W,C] -> uint8 [size,size,C]: optional crop (top, left, h, w), horizontal pass rounded to uint8, vertical pass rounded to uint8."""
    if crop is not None:
        img = img[crop[0]:crop[0] + crop[2], crop[1]:crop[1] + crop[3]]
    half = 1 << (bits - 1)
    tmp = np.stack([np.clip((np.tensordot(img[:, lo:lo + len(k)].astype(np.int64), k, axes=([1], [0])) + half) >> bits, 0, 255)
                    for lo, k in pil_bilinear_coeffs(img.shape[1], size, bits)], 1).astype(np.uint8)
    out = np.stack([np.clip((np.tensordot(tmp[lo:lo + len(k)].astype(np.int64), k, axes=([0], [0])) + half) >> bits, 0, 255)
                    for lo, k in pil_bilinear_coeffs(img.shape[0], size, bits)], 0).astype(np.uint8)
    return out


def image_batch(images, size, crop=None, flips=None):
    """The collate of dataset/ffhq.py:55-74 for decoded uint8 images [B,H,W,C]: x_0 float32 [B,C,S,S] = (v/255 - 0.5)/0.5, gts uint8 [B,S,S,C]."""
    out = []
    for b, im in enumerate(images):
        r = resize_u8(im, size, crop)
        out.append(r[:, ::-1] if (flips is not None and flips[b]) else r)
    gts = np.stack(out)
    x = torch.from_numpy(gts.copy()).float().div(255).permute(0, 3, 1, 2)
    return (x - 0.5) / 0.5, gts
