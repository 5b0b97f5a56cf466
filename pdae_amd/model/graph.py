"""Graph builders: walk a network config once and emit the static stage list (forward and the
hand-derived backward) into a Plan.

Topology follows the reference constructors (model/unet.py:60-175, model/shift_unet.py:65-249,
model/representation_learning/encoder/*.py) and the forward passes (unet.py:177-202,
shift_unet.py:253-284, encoder/ffhq.py:39-41); nothing here is hard-coded to one resolution or width.
"""
import os
from collections import OrderedDict
from types import SimpleNamespace as NS

from .. import hip as H


# ---------------------------------------------------------------------------------- topology / shapes
def topology(cfg):
    """(input_blocks, middle_block, output_blocks): lists of layer lists [(kind, attrs)], kind in conv|res|attn."""
    base, mult = cfg["base_channel"], list(cfg["channel_multiplier"])
    nres, attn_res = cfg["num_residual_blocks_of_a_block"], list(cfg["attention_resolutions"])
    ch = int(mult[0] * base)
    inputs = [[("conv", dict(cin=cfg["input_channel"], cout=ch))]]
    skip_ch = [ch]
    ds = 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            co = int(m * base)
            layers = [("res", dict(cin=ch, cout=co))]
            ch = co
            if ds in attn_res:
                layers.append(("attn", dict(ch=ch)))
            inputs.append(layers)
            skip_ch.append(ch)
        if level != len(mult) - 1:
            inputs.append([("res", dict(cin=ch, cout=ch, down=True))])
            skip_ch.append(ch)
            ds *= 2
    middle = [("res", dict(cin=ch, cout=ch)), ("attn", dict(ch=ch)), ("res", dict(cin=ch, cout=ch))]
    outputs = []
    for level, m in reversed(list(enumerate(mult))):
        for i in range(nres + 1):
            co = int(base * m)
            layers = [("res", dict(cin=ch + skip_ch.pop(), cout=co))]
            ch = co
            if ds in attn_res:
                layers.append(("attn", dict(ch=ch)))
            if level and i == nres:
                layers.append(("res", dict(cin=ch, cout=ch, up=True)))
                ds //= 2
            outputs.append(layers)
    return inputs, middle, outputs


def unet_shapes(cfg, shift=False, latent_dim=None):
    """state-dict key -> shape, in the reference's parameter registration order."""
    inputs, middle, outputs = topology(cfg)
    E = cfg["base_channel"] * 4
    img = cfg["input_channel"]
    out_ch = img * 2 if cfg.get("learn_sigma", False) else img
    S = OrderedDict()

    def lin(p, i, o):
        S[p + ".weight"] = (o, i)
        S[p + ".bias"] = (o,)

    def conv(p, i, o, k):
        S[p + ".weight"] = (o, i, k, k)
        S[p + ".bias"] = (o,)

    def vec(p, c):
        S[p + ".weight"] = (c,)
        S[p + ".bias"] = (c,)

    def block(pre, layers, zed):
        for j, (kind, d) in enumerate(layers):
            p = f"{pre}.{j}"
            if kind == "conv":
                conv(p, d["cin"], d["cout"], 3)
            elif kind == "res":
                vec(p + ".in_layers.0", d["cin"])
                conv(p + ".in_layers.2", d["cin"], d["cout"], 3)
                lin(p + ".emb_layers.1", E, 2 * d["cout"])
                if zed:
                    lin(p + ".emb_z_layers.1", E, 2 * d["cout"])
                vec(p + ".out_layers.0", d["cout"])
                conv(p + ".out_layers.3", d["cout"], d["cout"], 3)
                if d["cin"] != d["cout"]:
                    conv(p + ".skip_connection", d["cin"], d["cout"], 1)
            else:
                c = d["ch"]
                vec(p + ".norm", c)
                S[p + ".qkv.weight"] = (3 * c, c, 1)
                S[p + ".qkv.bias"] = (3 * c,)
                S[p + ".proj_out.weight"] = (c, c, 1)
                S[p + ".proj_out.bias"] = (c,)

    lin("time_embed.0", cfg["base_channel"], E)
    lin("time_embed.2", E, E)
    if shift:
        lin("label_emb", latent_dim, E)
    elif cfg.get("num_class") is not None:
        S["label_emb.weight"] = (cfg["num_class"], E)
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
    vec("out.0", c0)
    conv("out.2", c0, out_ch, 3)
    if shift:
        vec("shift_out.0", c0)
        conv("shift_out.2", c0, img, 3)
    return S


ZERO_INIT = (".out_layers.3.", ".proj_out.", "out.2.", "shift_out.2.")        # zero_module(...) in the reference

ENCODERS = {  # conv channel chain, index of the conv followed by the 4-head attention block
    "FFHQEncoder": ([3, 64, 128, 256, 256, 256], 2),
    "CELEBAHQEncoder": ([3, 64, 128, 256, 256, 256], 2),
    "BEDROOMEncoder": ([3, 64, 128, 256, 256, 256], 2),
    "HORSEEncoder": ([3, 64, 128, 256, 256, 256], 2),
    "CELEBA64Encoder": ([3, 64, 128, 128, 128], 1),
}


def encoder_layers(name):
    """[(kind, sequential_index, ...)] mirroring the nn.Sequential indices of encoder/ffhq.py:10-36."""
    chans, attn_after = ENCODERS[name]
    L, idx = [], 0
    for i in range(len(chans) - 1):
        L.append(("conv", idx, chans[i], chans[i + 1]))
        idx += 1
        if i == attn_after:
            L.append(("attn", idx, chans[i + 1]))
            idx += 1
        L.append(("gn", idx, chans[i + 1]))
        idx += 2
    L.append(("linear", idx + 1, chans[-1] * 16))
    return L


def encoder_shapes(name, latent_dim):
    S = OrderedDict()
    for l in encoder_layers(name):
        p = f"encoder.{l[1]}"
        if l[0] == "conv":
            S[p + ".weight"] = (l[3], l[2], 3, 3)
            S[p + ".bias"] = (l[3],)
        elif l[0] == "attn":
            c = l[2]
            S[p + ".norm.weight"] = (c,)
            S[p + ".norm.bias"] = (c,)
            S[p + ".qkv.weight"] = (3 * c, c, 1)
            S[p + ".qkv.bias"] = (3 * c,)
            S[p + ".proj_out.weight"] = (c, c, 1)
            S[p + ".proj_out.bias"] = (c,)
        elif l[0] == "gn":
            S[p + ".weight"] = (l[2],)
            S[p + ".bias"] = (l[2],)
        else:
            S[p + ".weight"] = (latent_dim, l[2])
            S[p + ".bias"] = (latent_dim,)
    return S


def _heads(cfg, ch):
    hc = cfg.get("head_channel", -1)
    return cfg.get("num_heads", 1) if hc == -1 else ch // hc


# ---------------------------------------------------------------------------------- forward graphs
def _run_block(B, cfg, pre, layers, x0, x1, ea, eza, dropout):
    pl = B.p
    ctxs = []
    h0, h1 = x0, x1
    for j, (kind, d) in enumerate(layers):
        p = f"{pre}.{j}"
        if kind == "conv":
            out, c = B.conv(h0, None, p, 3)
            if pre == "input_blocks.0":
                B.stats_pass(out)                     # the stem (edge kernel: no statistics epilogue) feeds three GroupNorms
        elif kind == "res":
            out, c = B.resblock(p, h0, h1, ea, eza, up=d.get("up", False), down=d.get("down", False), dropout=dropout)
        else:
            out, c = B.attention(p, h0, _heads(cfg, d["ch"]), cfg.get("use_new_attention_order", False))
        ctxs.append((kind, c))
        if j > 0 and not B.save:
            pl.free(h0)
        h0, h1 = out, None
    return h0, ctxs


def _time_embed(B, cfg, t, freqs, cond):
    pl = B.p
    N = t.shape[0]
    E0 = cfg["base_channel"]
    te = pl.buf(N, E0)
    pl.emit(H.op_temb(t, freqs, N, E0, te))
    h, l0 = B.linear(te, "time_embed.0")
    hs = B.silu(h)
    emb, l2 = B.linear(hs, "time_embed.2")
    if cond is not None:                      # class-conditional UNet (unet.py:190-192)
        table = B.P["label_emb.weight"]
        pl.emit(H.op_embedding(table, cond, N, emb.shape[1], emb, acc=1))
    ea = B.silu(emb)
    return NS(te=te, h=h, hs=hs, emb=emb, l0=l0, l2=l2, ea=ea, cond=cond, N=N)


def _head(B, pre, h):
    g = B.gn(h, None, pre + ".0", act=1)
    y, c = B.conv(g.y, None, pre + ".2", 3)
    if not B.save:
        B.p.free(g.y)
    return y, NS(g=g, c=c)


def unet_forward(B, cfg, x, t, freqs, cond=None, z=None, shift=False, train_shift=False, dropout=False, z_side=False):
    """Emits UNet (shift=False) or ShiftUNet (shift=True) forward.  x: NHWC [N,H,W,Cimg]; t: int64 [N].
    train_shift: keep the shift-branch activations (and every skip tensor) for `shift_backward`.
    When B.save is set by the caller (regular UNet training) everything is kept for `unet_backward`."""
    pl = B.p
    inputs, middle, outputs = topology(cfg)
    keep_all = B.save
    def res_prefixes(pre, blocks):
        return [f"{pre}.{i}.{j}" if isinstance(blocks[0], list) else f"{pre}.{j}" for i, layers in enumerate(blocks if isinstance(blocks[0], list) else [blocks])
                for j, (kind, d) in enumerate(layers) if kind == "res"]
    plain = res_prefixes("input_blocks", inputs) + res_prefixes("middle_block", middle) + res_prefixes("output_blocks", outputs)
    shifted = (res_prefixes("shift_middle_block", middle) + res_prefixes("shift_output_blocks", outputs)) if shift else []
    semb = eza = l_lab = None
    # sampling plans (nothing kept for a backward): everything that depends only on the latent z -- label_emb, its SiLU and the emb_z_layers Linear
    # of every shift ResBlock -- goes to the head of the op list and its results are pinned, so a denoising loop runs that prefix once, not on
    # each of its 100 / 1000 steps (ops [0, pl.n_const); diffusion/ddim.py)
    hoist = shift and not keep_all and not train_shift and len(pl.recs) == 0 and os.environ.get("PDAE_DDIM_HOIST", "1") != "0"
    if hoist:
        semb, l_lab = B.linear(z, "label_emb")
        eza = B.silu(semb)
        B.prefetch_emb(None, [], eza, shifted)
        pl.pin(semb, eza, *[y for (y, _) in B._emb.values()])
        pl.n_const = len(pl.recs)
    tctx = _time_embed(B, cfg, t, freqs, cond)
    ea = tctx.ea
    # z_side (round 6, training): z arrives from an encoder pass that runs on the SECOND stream beside the input blocks (trainer/fused_step.py);
    # everything that reads z in the forward pass -- label_emb, its SiLU, the emb_z_layers Linears -- belongs to the shift branch and goes there too,
    # so the main stream's first reader of anything z-dependent is behind the join at the end of this function
    z_side = bool(z_side and shift and not hoist and os.environ.get("PDAE_SIDE_SHIFT", "1") != "0")
    if shift and not hoist:
        with pl.side(z_side):
            semb, l_lab = B.linear(z, "label_emb")
            eza = B.silu(semb)
    # every ResBlock's emb_layers / emb_z_layers Linear in one launch (they only depend on the embeddings computed above)
    if hoist:
        B.prefetch_emb(ea, plain + shifted)
    elif z_side:
        B.prefetch_emb(ea, plain + shifted)
        with pl.side(True):
            B.prefetch_emb(None, [], eza, shifted)
    else:
        B.prefetch_emb(ea, plain + shifted, eza, shifted)
    hs, in_ctx = [], []
    h = x
    for i, layers in enumerate(inputs):
        h, cs = _run_block(B, cfg, f"input_blocks.{i}", layers, h, None, ea, None, dropout and keep_all)
        hs.append(h)
        in_ctx.append(cs)
    skips = list(hs)
    eps_h, mid_ctx = _run_block(B, cfg, "middle_block", middle, h, None, ea, None, dropout and keep_all)
    shift_h = smid_ctx = None
    # The shift branch (shift_middle_block, shift_output_blocks, shift_out) and the trunk's middle / output blocks both start from the input blocks'
    # results and meet only in the caller (loss / sampler update): the shift branch is emitted for the executor's SECOND stream (Plan.side), so the
    # HBM-bound ops of either branch (1x1 skip convolutions, GroupNorm coefficient / apply launches) run beside the other's MFMA kernels.
    side_shift = shift and os.environ.get("PDAE_SIDE_SHIFT", "1") != "0"
    if shift:
        B.save = keep_all or train_shift
        with pl.side(side_shift):
            shift_h, smid_ctx = _run_block(B, cfg, "shift_middle_block", middle, h, None, ea, eza, dropout)
        B.save = keep_all
    out_ctx, sout_ctx = [], []
    recycle_skips = not (keep_all or train_shift)
    for i, layers in enumerate(outputs):
        prev = hs.pop()
        new, cs = _run_block(B, cfg, f"output_blocks.{i}", layers, eps_h, prev, ea, None, dropout and keep_all)
        out_ctx.append(cs)
        if not keep_all and eps_h is not h:
            pl.free(eps_h)
        eps_h = new
        if shift:
            B.save = keep_all or train_shift
            with pl.side(side_shift):
                new, cs = _run_block(B, cfg, f"shift_output_blocks.{i}", layers, shift_h, prev, ea, eza, dropout)
            B.save = keep_all
            sout_ctx.append(cs)
            if recycle_skips and shift_h is not h:
                pl.free(shift_h)
            shift_h = new
        if recycle_skips:
            pl.free(prev)
    eps, head = _head(B, "out", eps_h)
    if not keep_all:
        pl.free(eps_h)
    g_out = shead = None
    if shift:
        B.save = keep_all or train_shift
        with pl.side(side_shift):
            g_out, shead = _head(B, "shift_out", shift_h)
        B.save = keep_all
        if recycle_skips:
            pl.free(shift_h)
        pl.join()                                        # both outputs are read next
    return NS(cfg=cfg, tctx=tctx, semb=semb, eza=eza, l_lab=l_lab, in_ctx=in_ctx, mid_ctx=mid_ctx, smid_ctx=smid_ctx, out_ctx=out_ctx,
              sout_ctx=sout_ctx, head=head, shead=shead, eps=eps, shift=g_out, skips=skips, z=z, x=x)


# ---------------------------------------------------------------------------------- backward graphs
def _block_backward(B, ctxs, dout, d_ea, d_eza, need_dx0_first=True, need_dx1_first=False, dout_amax=None, out_amax=False):
    """Backward through one TimestepSequential.  Returns (dx0, dx1, amax of dx0 or None) of the block's first layer.
    dout_amax: device scalar max|dout| from the producer of dout; out_amax: the caller hands dx0 UNMODIFIED to the next backward stage and
    wants its abs-max (it falls out of the last GroupNorm-backward pass: no separate pdae_amax launch)."""
    pl = B.p
    dx1 = None
    am = dout_amax
    for j in range(len(ctxs) - 1, -1, -1):
        kind, c = ctxs[j]
        first = j == 0
        need0 = (not first) or need_dx0_first
        want = (not first) or out_amax                   # inside the block dx0 always flows straight into the next layer
        if kind == "res":
            dx0, d1, am_next = B.resblock_bwd(c, dout, need_dx0=need0, need_dx1=first and need_dx1_first, d_ea=d_ea, d_eza=d_eza, dout_amax=am, out_amax=want)
            if first:
                dx1 = d1
        elif kind == "attn":
            dx0, am_next = B.attention_bwd(c, dout, need_dx=need0, dout_amax=am, out_amax=want)
        else:
            B.conv_bwd_params(c, dout)
            dx0 = B.conv_dgrad(c, dout) if need0 else None
            am_next = None
        pl.free(dout, am)
        dout, am = dx0, am_next
    return dout, dx1, am


def _head_backward(B, hd, dy):
    """Returns (dh, max|dh| device scalar or None)."""
    pl = B.p
    B.conv_bwd_params(hd.c, dy)
    d_a = B.conv_dgrad(hd.c, dy)
    dh = pl.buf(hd.g.N, hd.g.H, hd.g.W, hd.g.C0)
    am = pl.buf(4) if B.f16_grads else None
    B.gn_bwd(hd.g, d_a, 0, dx0=dh, dx0_amax=am)
    pl.free(d_a)
    return dh, am


def shift_backward(B, fx, d_shift, mark=None):
    """Backward of the trainable half of ShiftUNet (label_emb, shift_middle_block, shift_output_blocks, shift_out;
    shift_unet.py:299-310).  d_shift: NHWC gradient of the `shift` output.  Returns dz [N, latent].
    mark(prefix): called when every parameter gradient under `prefix` has been emitted (DDP bucket boundaries)."""
    pl = B.p
    mark = mark or (lambda prefix: None)
    N, E = fx.semb.shape
    d_eza = pl.buf(N, E, zero=True)
    dh, am = _head_backward(B, fx.shead, d_shift)
    mark("shift_out.")
    for i in range(len(fx.sout_ctx) - 1, -1, -1):
        dh, _, am = _block_backward(B, fx.sout_ctx[i], dh, None, d_eza, dout_amax=am, out_amax=True)
        mark(f"shift_output_blocks.{i}.")
    _, _, am = _block_backward(B, fx.smid_ctx, dh, None, d_eza, need_dx0_first=False, dout_amax=am)
    pl.free(am)
    mark("shift_middle_block.")
    d_semb = pl.buf(N, E)
    pl.emit(H.op_silu_bwd(fx.semb, d_eza, d_semb, N * E))
    dz = pl.buf(N, fx.z.shape[1])
    B.linear_bwd(fx.l_lab, d_semb, dx=dz, dx_acc=0)
    mark("label_emb.")
    pl.free(d_eza, d_semb)
    return dz


def unet_backward(B, fx, d_eps):
    """Full backward of a regular UNet (config #1: trainer/train_regular_diffusion.py), every parameter trained."""
    pl = B.p
    t = fx.tctx
    N, E = t.emb.shape
    d_ea = pl.buf(N, E, zero=True)
    dh, am = _head_backward(B, fx.head, d_eps)
    n_in = len(fx.in_ctx)
    d_skip = [None] * n_in
    for i in range(len(fx.out_ctx) - 1, -1, -1):
        dh, dx1, am = _block_backward(B, fx.out_ctx[i], dh, d_ea, None, need_dx1_first=True, dout_amax=am, out_amax=True)
        d_skip[n_in - 1 - i] = dx1
    dh, _, am = _block_backward(B, fx.mid_ctx, dh, d_ea, None, dout_amax=am)
    pl.free(am)
    for k in range(n_in - 1, -1, -1):
        g = d_skip[k]
        pl.emit(H.op_axpby(dh, g, dh.numel(), 1.0, 1.0))         # the skip gradient joins: the sum's abs-max is not known, pdae_amax runs
        pl.free(dh)
        dh, _, am = _block_backward(B, fx.in_ctx[k], g, d_ea, None, need_dx0_first=(k > 0))
        pl.free(am)
    # time embedding MLP (+ class embedding table)
    d_emb = pl.buf(N, E)
    pl.emit(H.op_silu_bwd(t.emb, d_ea, d_emb, N * E))
    gt = B.Gr.get("label_emb.weight") if t.cond is not None else None
    if gt is not None:
        if not B.acc:                                # accumulating plans add into the (pre-zeroed) table gradient
            pl.emit(H.op_memset(gt, gt.numel() * 4))
        pl.emit(H.op_embedding_bwd(d_emb, t.cond, N, E, gt))
    d_hs = pl.buf(N, E)
    B.linear_bwd(t.l2, d_emb, dx=d_hs, dx_acc=0)
    d_h = pl.buf(N, E)
    pl.emit(H.op_silu_bwd(t.h, d_hs, d_h, N * E))
    B.linear_bwd(t.l0, d_h)
    pl.free(d_ea, d_emb, d_hs, d_h)


# ---------------------------------------------------------------------------------- encoder
def encoder_forward(B, name, x):
    """x: NHWC image.  Returns (z [N, latent], ctx)."""
    pl = B.p
    ctxs = []
    h = x
    L = encoder_layers(name)
    for l in L[:-1]:
        p = f"encoder.{l[1]}"
        if l[0] == "conv":
            y, c = B.conv(h, None, p, 3, stride=2)
            ctxs.append(("conv", c))
        elif l[0] == "attn":
            y, c = B.attention(p, h, 4, False)
            ctxs.append(("attn", c))
        else:
            c = B.gn(h, None, p, act=1)
            y = c.y
            ctxs.append(("gn", c))
        if h is not x and not B.save:
            pl.free(h)
        h = y
    N, Hh, W, C = h.shape
    flat = pl.buf(N, C * Hh * W)                  # View((-1, C*4*4)) flattens in NCHW order (encoder/ffhq.py:34)
    pl.emit(H.op_from_nhwc(h, N, C, Hh, W, flat, (C * Hh * W, Hh * W, W, 1)))
    z, lz = B.linear(flat, f"encoder.{L[-1][1]}")
    if not B.save:
        pl.free(h, flat)
    return z, NS(ctxs=ctxs, lz=lz, last=h, flat=flat)


def encoder_backward(B, ex, dz, mark=None):
    """mark(prefix): called when every parameter gradient under `prefix` has been emitted (DDP bucket boundaries): the final Linear holds more
    than half of the encoder's gradient bytes and is final first, so its all-reduce starts under the rest of the encoder backward instead of
    behind the last backward op."""
    pl = B.p
    mark = mark or (lambda prefix: None)
    N, Hh, W, C = ex.last.shape
    d_flat = pl.buf(N, C * Hh * W)
    B.linear_bwd(ex.lz, dz, dx=d_flat, dx_acc=0)
    mark(ex.lz.wname + ".")
    dh = pl.buf(N, Hh, W, C)
    pl.emit(H.op_to_nhwc(d_flat, (C * Hh * W, Hh * W, W, 1), N, C, Hh, W, dh))
    pl.free(d_flat)
    for j in range(len(ex.ctxs) - 1, -1, -1):
        kind, c = ex.ctxs[j]
        if kind == "gn":
            dprev = pl.buf(c.N, c.H, c.W, c.C0)
            B.gn_bwd(c, dh, 0, dx0=dprev)
            mark(c.gname + ".")
        elif kind == "attn":
            dprev, _ = B.attention_bwd(c, dh, need_dx=True)
            mark(c.pre + ".")
        else:
            B.conv_bwd_params(c, dh)
            dprev = B.conv_dgrad(c, dh) if j > 0 else None
            mark(c.wname + ".")
        pl.free(dh)
        dh = dprev
