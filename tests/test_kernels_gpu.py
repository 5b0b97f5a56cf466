"""GPU: every HIP kernel of libpdae_hip.so, called through the C ABI (pdae_run_ops), against a plain
torch CPU reference of the same op (fp64 where cheap).  Tolerance is the north-star's 1e-4 relative
(max-abs error / max-abs reference) unless tightened below."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from pdae_amd import hip
    hip.lib()
    return hip


def rn(seed, *shape, scale=1.0):
    return torch.tensor(np.random.default_rng(seed).standard_normal(shape) * scale, dtype=torch.float32)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def ws(nbytes):
    return torch.empty(max(int(nbytes), 16) // 4 + 4, dtype=torch.float32, device="cuda")


def ref_conv(x, w, b, stride, pad, up):
    if up:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv2d(x.double(), w.double(), None if b is None else b.double(), stride=stride, padding=pad)


CONV_CASES = [
    # N, H, W, C0, C1, Cout, k, stride, up, res_mode, tile
    (2, 12, 12, 64, 0, 96, 3, 1, 0, 0, 0),
    (2, 12, 12, 64, 0, 96, 3, 1, 0, 1, 128),
    (1, 10, 14, 32, 64, 64, 3, 1, 0, 0, 64),
    (2, 8, 8, 32, 0, 32, 3, 1, 1, 2, 0),
    (2, 16, 16, 3, 0, 64, 3, 2, 0, 0, 0),
    (2, 16, 16, 64, 0, 128, 3, 2, 0, 0, 0),
    (2, 9, 9, 64, 32, 32, 1, 1, 0, 1, 0),
    (1, 16, 16, 32, 0, 3, 3, 1, 0, 0, 0),
    (1, 16, 16, 1, 0, 32, 3, 1, 0, 0, 0),
    (3, 32, 32, 128, 0, 128, 3, 1, 0, 0, 128),
    (2, 24, 40, 128, 0, 3, 3, 1, 0, 0, 0),        # image head (convhead.hip): ragged tiles, 4 channel chunks
    (2, 20, 12, 64, 0, 4, 3, 1, 0, 0, 0),
    (1, 16, 16, 32, 0, 1, 3, 1, 0, 0, 0),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(H, case):
    N, Hh, W, C0, C1, Cout, k, stride, up, res_mode, tile = case
    Cin = C0 + C1
    x = rn(1, N, Cin, Hh, W)
    w = rn(2, Cout, Cin, k, k, scale=1.0 / math.sqrt(Cin * k * k))
    b = rn(3, Cout, scale=0.1)
    c = H.Conv(N, Hh, W, C0, C1, Cout, k=k, stride=stride, up=bool(up))
    res = None
    y_ref = ref_conv(x, w, b, stride, k // 2, up)
    if res_mode == 1:
        res = rn(4, N, Cout, c.Ho, c.Wo)
        y_ref = y_ref + res.double()
    elif res_mode == 2:
        res = rn(4, N, Cout, c.Ho // 2, c.Wo // 2)
        y_ref = y_ref + F.interpolate(res, scale_factor=2, mode="nearest").double()
    xh = nhwc(x).cuda()
    x0 = xh[..., :C0].contiguous()
    x1 = xh[..., C0:].contiguous() if C1 else None
    wd = nhwc(w).cuda()                      # [Cout][KH][KW][Cin]
    bd = b.cuda()
    resd = nhwc(res).cuda() if res is not None else None
    y = torch.empty(N, c.Ho, c.Wo, Cout, device="cuda")
    H.run(H.op_conv_fwd(c, x0, x1, wd, bd, y, res=resd, res_mode=res_mode, tile=tile))
    assert rel_err(nchw(y), y_ref) < 1e-5

    # backward references via autograd in fp64; the input gradient is taken wrt the LOGICAL (upsampled) conv input
    wr = w.double().requires_grad_(True)
    dy = rn(5, N, Cout, c.Ho, c.Wo)
    xl = (F.interpolate(x, scale_factor=2, mode="nearest") if up else x).double().clone().requires_grad_(True)
    (F.conv2d(xl, wr, None, stride=stride, padding=k // 2) * dy.double()).sum().backward()
    dx_ref, dw_ref = xl.grad, wr.grad
    dyd = nhwc(dy).cuda()
    if Cin >= 4:        # image-channel convs never need an input gradient
        dx = torch.empty(N, c.Hl, c.Wl, Cin, device="cuda")
        H.run(H.op_conv_dgrad(c, dyd, wd, dx, tile=tile))
        assert rel_err(nchw(dx), dx_ref) < 1e-5
        if C1:          # only the first concat half, accumulated onto existing content
            dx0 = torch.ones(N, c.Hl, c.Wl, C0, device="cuda")
            H.run(H.op_conv_dgrad(c, dyd, wd, dx0, ci_off=0, ci_cnt=C0, accumulate=1))
            assert rel_err(nchw(dx0) - 1.0, dx_ref[:, :C0]) < 1e-5
            dx1 = torch.empty(N, c.Hl, c.Wl, C1, device="cuda")
            H.run(H.op_conv_dgrad(c, dyd, wd, dx1, ci_off=C0, ci_cnt=C1))
            assert rel_err(nchw(dx1), dx_ref[:, C0:]) < 1e-5
    wsb = c.wgrad_ws_bytes()
    wsp = ws(wsb)
    dw = torch.empty_like(wd)
    db = torch.empty(Cout, device="cuda")            # bias gradient rides along (column sums of dy)
    db_ref = dy.double().sum((0, 2, 3))
    H.run(H.op_conv_wgrad(c, x0, x1, dyd, dw, wsp, wsb, db=db))
    assert rel_err(dw.permute(0, 3, 1, 2), dw_ref) < 2e-5 and rel_err(db, db_ref) < 1e-5
    H.run(H.op_conv_wgrad(c, x0, x1, dyd, dw, wsp, wsb, accumulate=1, db=db))
    assert rel_err(dw.permute(0, 3, 1, 2), 2 * dw_ref) < 2e-5 and rel_err(db, 2 * db_ref) < 1e-5


def test_conv_wgrad_splitk_large(H):
    N, Hh, C, Cout = 4, 32, 32, 64
    x, dy = rn(1, N, C, Hh, Hh), rn(2, N, Cout, Hh, Hh)
    c = H.Conv(N, Hh, Hh, C, 0, Cout)
    wr = torch.zeros(Cout, C, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), wr, padding=1) * dy.double()).sum().backward()
    wsb = c.wgrad_ws_bytes()
    assert wsb > 16            # really exercises the split-K path
    dw = torch.empty(Cout, 3, 3, C, device="cuda")
    H.run(H.op_conv_wgrad(c, nhwc(x).cuda(), None, nhwc(dy).cuda(), dw, ws(wsb), wsb))
    assert rel_err(dw.permute(0, 3, 1, 2), wr.grad) < 2e-5


GEMM_CASES = [  # transA, transB, M, N, K, bo, bi
    (0, 1, 70, 50, 36, 2, 3), (0, 0, 64, 48, 100, 1, 2), (1, 0, 33, 65, 130, 2, 1),
    (0, 1, 7, 5, 3, 1, 1), (0, 0, 5, 3, 9, 1, 1), (1, 0, 6, 10, 7, 1, 1), (0, 1, 256, 256, 64, 2, 4),
]


@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm(H, case):
    tA, tB, M, N, K, bo, bi = case
    nb = bo * bi
    A = rn(1, nb, *((K, M) if tA else (M, K)))
    B = rn(2, nb, *((N, K) if tB else (K, N)))
    bias = rn(3, N)
    C0 = rn(4, nb, M, N)
    opA = A.transpose(1, 2) if tA else A
    opB = B.transpose(1, 2) if tB else B
    ref = 0.5 * torch.bmm(opA.double(), opB.double()) + bias.double() + C0.double()
    Ad, Bd, Cd = A.cuda(), B.cuda(), C0.clone().cuda()
    sa, sb, sc = A.shape[1] * A.shape[2], B.shape[1] * B.shape[2], M * N
    H.run(H.op_gemm(tA, tB, M, N, K, Ad, A.shape[2], Bd, B.shape[2], Cd, N, alpha=0.5, bias=bias.cuda(), accumulate=1,
                    batch_outer=bo, batch_inner=bi, sA=(bi * sa, sa), sB=(bi * sb, sb), sC=(bi * sc, sc)))
    assert rel_err(Cd, ref) < 1e-5


def _gn_ref(x, gamma, beta, ss, zss, act, G=32):
    y = F.group_norm(x, G, gamma, beta, 1e-5)
    if ss is not None:
        s, sh = ss[:, :, None, None].chunk(2, 1)
        y = y * (1 + s) + sh
    if zss is not None:
        s, sh = zss[:, :, None, None].chunk(2, 1)
        y = (1 + s) * y + sh
    return F.silu(y) if act else y


GN_CASES = [  # N, H, W, C0, C1, use_ss, use_zss, act, mode(0 same,1 down,2 up-consumer)
    (2, 8, 8, 64, 0, 0, 0, 1, 0), (2, 8, 8, 64, 32, 1, 1, 1, 0), (2, 6, 10, 32, 0, 1, 0, 0, 0),
    (2, 8, 8, 64, 0, 0, 0, 1, 1), (2, 4, 4, 96, 0, 0, 0, 1, 2), (1, 32, 32, 128, 128, 1, 1, 1, 0),
    (2, 16, 16, 512, 384, 0, 0, 1, 0),
]


@pytest.mark.parametrize("case", GN_CASES)
def test_groupnorm_family(H, case):
    N, Hh, W, C0, C1, use_ss, use_zss, act, mode = case
    C, G = C0 + C1, 32
    x = (rn(1, N, C, Hh, W) * 1.5 + 0.7).double().requires_grad_(True)
    gamma = (1 + 0.2 * rn(2, C)).double().requires_grad_(True)
    beta = (0.2 * rn(3, C)).double().requires_grad_(True)
    ss = (0.3 * rn(4, N, 2 * C)).double().requires_grad_(True) if use_ss else None
    zss = (0.3 * rn(5, N, 2 * C)).double().requires_grad_(True) if use_zss else None
    y_ref = _gn_ref(x, gamma, beta, ss, zss, act)
    xp_ref = None
    if mode == 1:
        y_ref, xp_ref = F.avg_pool2d(y_ref, 2), F.avg_pool2d(x, 2)
    y_used = F.interpolate(y_ref, scale_factor=2, mode="nearest") if mode == 2 else y_ref
    dA = rn(6, *y_used.shape)
    add = rn(7, *y_used.shape) if C1 == 0 else None
    loss = (y_used * dA.double()).sum()
    if add is not None:   # identity-skip path resampled the same way
        xs = F.avg_pool2d(x, 2) if mode == 1 else (F.interpolate(x, scale_factor=2, mode="nearest") if mode == 2 else x)
        loss = loss + (xs * add.double()).sum()
    loss.backward()

    f32 = lambda t: None if t is None else t.detach().float().cuda()
    xh = nhwc(x.detach().float()).cuda()
    x0 = xh[..., :C0].contiguous()
    x1 = xh[..., C0:].contiguous() if C1 else None
    mean = torch.empty(N * G, device="cuda"); rstd = torch.empty(N * G, device="cuda")
    wsp = ws(H.gn_ws_bytes(N, C))
    H.run(H.op_gn_stats(x0, C0, x1, C1, N, Hh * W, G, 1e-5, mean, rstd, wsp))
    xg = x.detach().reshape(N, G, -1)
    assert rel_err(mean, xg.mean(-1).flatten()) < 1e-6
    assert rel_err(rstd, 1.0 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-5).flatten()) < 1e-5
    coef = torch.empty(3, N, C, device="cuda")
    H.run(H.op_gn_coef(N, C, G, mean, rstd, f32(gamma), f32(beta), f32(ss), f32(zss), coef))
    # the two-launch form (statistics partials + fused finalize / coefficient fold) must give the same numbers
    mean2, rstd2, coef2 = torch.empty_like(mean), torch.empty_like(rstd), torch.empty_like(coef)
    H.run(H.op_gn_stats_coef(x0, C0, x1, C1, N, Hh * W, G, 1e-5, f32(gamma), f32(beta), f32(ss), f32(zss), mean2, rstd2, coef2, wsp))
    assert torch.equal(mean, mean2) and torch.equal(rstd, rstd2) and torch.equal(coef, coef2)
    Ho, Wo = (Hh // 2, W // 2) if mode == 1 else (Hh, W)
    y = torch.empty(N, Ho, Wo, C, device="cuda")
    xp = torch.empty(N, Ho, Wo, C, device="cuda") if mode == 1 else None
    H.run(H.op_gn_apply(x0, C0, x1, C1, N, Hh, W, coef, act, 1 if mode == 1 else 0, y, xpool=xp))
    assert rel_err(nchw(y), y_ref) < 1e-5
    if xp is not None:
        assert rel_err(nchw(xp), xp_ref) < 1e-6

    dx0 = torch.empty(N, Hh, W, C0, device="cuda")
    dx1 = torch.empty(N, Hh, W, C1, device="cuda") if C1 else None
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dss = torch.empty(N, 2 * C, device="cuda") if use_ss else None
    dzss = torch.empty(N, 2 * C, device="cuda") if use_zss else None
    am = torch.full((4,), 123.0, device="cuda")          # receives max|dx0| (stale content must be overwritten)
    H.run(H.op_gn_bwd(x0, C0, x1, C1, N, Hh, W, G, coef, rstd, f32(gamma), f32(beta), f32(ss), f32(zss), nhwc(dA).cuda(), act, mode, wsp,
                      add=None if add is None else nhwc(add).cuda(), dx0=dx0, dx1=dx1, dgamma=dg, dbeta=db, dss=dss, dzss=dzss, dx0_amax=am))
    assert float(am[0]) == float(dx0.abs().max())
    # single-launch forms ("the last block of a sample finalizes it", ticket words): same arithmetic in the same order -> bit-identical, and the
    # tickets return to zero so that the same words serve every later launch (three rounds)
    tick = torch.zeros(N + 1, dtype=torch.int32, device="cuda")
    for _ in range(3):
        mean3, rstd3, coef3 = torch.zeros_like(mean), torch.zeros_like(rstd), torch.zeros_like(coef)
        H.run(H.op_gn_stats_coef(x0, C0, x1, C1, N, Hh * W, G, 1e-5, f32(gamma), f32(beta), f32(ss), f32(zss), mean3, rstd3, coef3, wsp, ticket=tick))
        assert torch.equal(mean, mean3) and torch.equal(rstd, rstd3) and torch.equal(coef, coef3)
        t_dx0, t_dg, t_db = torch.zeros_like(dx0), torch.zeros_like(dg), torch.zeros_like(db)
        t_dx1 = torch.zeros_like(dx1) if C1 else None
        t_dss = torch.zeros_like(dss) if use_ss else None
        t_dzss = torch.zeros_like(dzss) if use_zss else None
        am2 = torch.full((4,), 7.0, device="cuda")
        H.run(H.op_gn_bwd(x0, C0, x1, C1, N, Hh, W, G, coef, rstd, f32(gamma), f32(beta), f32(ss), f32(zss), nhwc(dA).cuda(), act, mode, wsp,
                          add=None if add is None else nhwc(add).cuda(), dx0=t_dx0, dx1=t_dx1, dgamma=t_dg, dbeta=t_db, dss=t_dss, dzss=t_dzss,
                          dx0_amax=am2, ticket=tick))
        assert torch.equal(t_dx0, dx0) and torch.equal(t_dg, dg) and torch.equal(t_db, db) and float(am2[0]) == float(am[0])
        assert (not C1 or torch.equal(t_dx1, dx1)) and (not use_ss or torch.equal(t_dss, dss)) and (not use_zss or torch.equal(t_dzss, dzss))
        assert int(tick.abs().sum()) == 0
    dx = torch.cat([dx0, dx1], -1) if C1 else dx0
    assert rel_err(nchw(dx), x.grad) < 2e-5
    assert rel_err(dg, gamma.grad) < 2e-5 and rel_err(db, beta.grad) < 2e-5
    if use_ss:
        assert rel_err(dss, ss.grad) < 2e-5
    if use_zss:
        assert rel_err(dzss, zss.grad) < 2e-5


def test_dropout_mask_consistency(H):
    N, Hh, W, C, G, p = 2, 16, 16, 64, 32, 0.25
    x = nhwc(rn(1, N, C, Hh, W)).cuda()
    mean = torch.empty(N * G, device="cuda"); rstd = torch.empty(N * G, device="cuda")
    wsp = ws(H.gn_ws_bytes(N, C))
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    H.run(H.op_gn_stats(x, C, None, 0, N, Hh * W, G, 1e-5, mean, rstd, wsp))
    coef = torch.empty(3, N, C, device="cuda")
    H.run(H.op_gn_coef(N, C, G, mean, rstd, gamma, beta, None, None, coef))
    y0 = torch.empty_like(x); y1 = torch.empty_like(x); y2 = torch.empty_like(x)
    H.run(H.op_gn_apply(x, C, None, 0, N, Hh, W, coef, 0, 0, y0))
    H.run(H.op_gn_apply(x, C, None, 0, N, Hh, W, coef, 0, 0, y1, drop_p=p, seed=1234, offset=7))
    H.run(H.op_gn_apply(x, C, None, 0, N, Hh, W, coef, 0, 0, y2, drop_p=p, seed=1234, offset=7))
    assert torch.equal(y1, y2)
    keep = y1 != 0
    frac = keep.float().mean().item()
    assert abs(frac - (1 - p)) < 0.02
    assert rel_err(y1[keep], y0[keep] / (1 - p)) < 1e-6
    # backward regenerates the same mask: with act=0, gamma=1 the input gradient of a masked element is pure GN-coupling
    dA = torch.ones_like(x)
    dx = torch.empty_like(x)
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    H.run(H.op_gn_bwd(x, C, None, 0, N, Hh, W, G, coef, rstd, gamma, beta, None, None, dA, 0, 0, wsp, dx0=dx, dgamma=dg, dbeta=db,
                      drop_p=p, seed=1234, offset=7))
    assert rel_err(db, (keep.float() / (1 - p)).sum((0, 1, 2))) < 1e-5
    # the mask stream: keep rate at the configured probability (16-bit uniforms: resolution 1.5e-5), a new mask per step offset and per layer
    # seed, no correlation between the four lanes of a Philox2x32 call or between neighbouring quads
    for pp in (0.1, 0.5):
        ya = torch.empty_like(x); yb = torch.empty_like(x); yc = torch.empty_like(x)
        H.run(H.op_gn_apply(x, C, None, 0, N, Hh, W, coef, 0, 0, ya, drop_p=pp, seed=3, offset=11))
        H.run(H.op_gn_apply(x, C, None, 0, N, Hh, W, coef, 0, 0, yb, drop_p=pp, seed=3, offset=12))
        H.run(H.op_gn_apply(x, C, None, 0, N, Hh, W, coef, 0, 0, yc, drop_p=pp, seed=4, offset=11))
        ka, kb, kc = (ya != 0).float(), (yb != 0).float(), (yc != 0).float()
        n = ka.numel()
        sigma = math.sqrt(pp * (1 - pp) / n)
        for k in (ka, kb, kc):
            assert abs(k.mean().item() - (1 - pp)) < 5 * sigma + 2e-5
        for u, v in ((ka, kb), (ka, kc)):                                  # independent masks: agreement rate = p^2 + (1-p)^2
            assert abs((u == v).float().mean().item() - (pp * pp + (1 - pp) * (1 - pp))) < 6 * math.sqrt(0.25 / n)
        flat = ka.reshape(-1, 4)                                           # the four elements of a quad come from one call
        for i in range(1, 4):
            cov = ((flat[:, 0] - (1 - pp)) * (flat[:, i] - (1 - pp))).mean().item()
            assert abs(cov) < 6 * pp * (1 - pp) / math.sqrt(flat.shape[0])


def test_elementwise_diffusion(H):
    from oracle import pdae_oracle as O
    s = O.Schedules()
    N, per = 3, 3 * 8 * 8
    t = torch.tensor([0, 500, 999])
    x0, noise, eps, g = rn(1, N, 3, 8, 8), rn(2, N, 3, 8, 8), rn(3, N, 3, 8, 8), rn(4, N, 3, 8, 8)
    td = t.cuda()
    # timestep embedding
    for dim in (32, 128):
        half = dim // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).cuda()
        out = torch.empty(N, dim, device="cuda")
        H.run(H.op_temb(td, freqs, N, dim, out))
        assert (out.cpu() - O.timestep_embedding(t, dim)).abs().max() < 2e-6
    # q_sample
    xt = torch.empty(N, 3, 8, 8, device="cuda")
    H.run(H.op_q_sample(x0.cuda(), noise.cuda(), td, s.sqrt_alphas_cumprod.cuda(), s.sqrt_one_minus_alphas_cumprod.cuda(), N, per, xt))
    assert rel_err(xt, O.q_sample(s, x0, t, noise)) < 1e-6
    # weighted L2 loss + grads, plain L2, L1
    er, gr = eps.clone().requires_grad_(True), g.clone().requires_grad_(True)
    ref = O.p_loss(noise, er + O._at(s.shift_coef, t, x0) * gr, weight=O._at(s.weight, t, x0))
    ref.backward()
    loss = torch.empty(1, device="cuda"); deps = torch.empty(N, 3, 8, 8, device="cuda"); dg = torch.empty_like(deps)
    wsp = ws(8192)
    H.run(H.op_loss(noise.cuda(), eps.cuda(), g.cuda(), td, s.shift_coef.cuda(), s.weight.cuda(), N, per, loss, wsp, deps=deps, dg=dg))
    assert abs(loss.item() - ref.item()) < 1e-6 * abs(ref.item()) + 1e-9
    assert rel_err(deps, er.grad) < 1e-5 and rel_err(dg, gr.grad) < 1e-5
    er.grad = None
    ref = O.p_loss(noise, er); ref.backward()
    H.run(H.op_loss(noise.cuda(), eps.cuda(), None, None, None, None, N, per, loss, wsp, deps=deps))
    assert abs(loss.item() - ref.item()) < 1e-6 * abs(ref.item()) and rel_err(deps, er.grad) < 1e-5
    er.grad = None
    ref = O.p_loss(noise, er, loss_type="l1"); ref.backward()
    H.run(H.op_loss(noise.cuda(), eps.cuda(), None, None, None, None, N, per, loss, wsp, deps=deps, l1=1))
    assert abs(loss.item() - ref.item()) < 1e-6 * abs(ref.item()) and rel_err(deps, er.grad) < 1e-5
    # DDIM update, sample and encode direction, with and without shift
    d = O.DDIMTables(s, "ddim20")
    for i, encode, use_shift in [(20, False, True), (7, False, False), (0, True, True), (13, True, True)]:
        tt = torch.full((N,), i, dtype=torch.long)
        ref = O.ddim_update(d, x0 * 1.5, tt, eps, g, encode=encode, use_shift=use_shift)
        ab = float((d.alphas_cumprod_next if encode else d.alphas_cumprod_prev)[i])
        out = torch.empty(N, 3, 8, 8, device="cuda")
        H.run(H.op_ddim_step((x0 * 1.5).cuda(), eps.cuda(), g.cuda() if use_shift else None, N * per,
                             float(d.sqrt_one_minus_alphas_cumprod[i]), float(d.sqrt_recip_alphas_cumprod[i]),
                             float(d.sqrt_recip_alphas_cumprod_m1[i]), float(np.sqrt(np.float32(ab))),
                             float(np.sqrt(np.float32(1.0) - np.float32(ab))), out))
        assert rel_err(out, ref) < 1e-5
    # DDPM mean
    tt = torch.full((N,), 400, dtype=torch.long)
    ref = O.noise_p_sample_mean(s, x0, tt, eps + O._at(s.shift_coef, tt, x0) * g) + 0.3 * noise
    out = torch.empty(N, 3, 8, 8, device="cuda")
    H.run(H.op_ddpm_step(x0.cuda(), eps.cuda(), g.cuda(), noise.cuda(), N * per, float(s.noise_posterior_mean_x_t_coef[400]),
                         float(s.noise_posterior_mean_noise_coef[400]), float(s.shift_coef[400]), 0.3, out))
    assert rel_err(out, ref) < 1e-5


def test_adam_ema_matches_torch_optim(H):
    n = 5000
    for cls, wd, dec in [(torch.optim.Adam, 0.0, 0), (torch.optim.AdamW, 0.01, 1), (torch.optim.Adam, 0.05, 0)]:
        p = torch.nn.Parameter(rn(1, n))
        opt = cls([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
        pd, m, v = p.detach().clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        ema, ema_ref = pd.clone(), p.detach().clone()
        for step in range(1, 4):
            g = rn(10 + step, n)
            p.grad = g.clone()
            opt.step()
            ema_ref.mul_(0.99).add_(p.detach(), alpha=0.01)
            bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
            H.run(H.op_adam_ema(pd, (g * 2).cuda(), m, v, ema, n, 1e-3, 0.9, 0.999, 1e-8, wd, dec, 1e-3 / bc1, 1.0 / math.sqrt(bc2), 0.5, 0.99))
        assert rel_err(pd, p.detach()) < 2e-6 and rel_err(ema, ema_ref) < 2e-6


def test_softmax_colsum_layout_misc(H):
    rows, T = 37, 256
    s = rn(1, rows, T, scale=3.0)
    sd = s.clone().cuda()
    H.run(H.op_softmax(sd, rows, T))
    assert rel_err(sd, torch.softmax(s.double(), -1)) < 1e-6
    sr = s.double().requires_grad_(True)
    dp = rn(2, rows, T)
    (torch.softmax(sr, -1) * dp.double()).sum().backward()
    dpd = dp.clone().cuda()
    H.run(H.op_softmax_bwd(sd, dpd, rows, T))
    assert rel_err(dpd, sr.grad) < 1e-5
    for M, C in [(1000, 96), (5000, 3), (7, 512)]:
        x = rn(3, M, C)
        out = torch.ones(C, device="cuda")
        H.run(H.op_colsum(x.cuda(), M, C, out, ws(H.colsum_ws_bytes(M, C)), acc=1))
        assert rel_err(out - 1, x.double().sum(0)) < 1e-5
    x = rn(4, 2, 3, 5, 7)
    y = torch.empty(2, 5, 7, 3, device="cuda")
    xs = x.cuda()
    H.run(H.op_to_nhwc(xs, xs.stride(), 2, 3, 5, 7, y))
    assert torch.equal(y.cpu(), nhwc(x))
    z = torch.empty(2, 3, 5, 7, device="cuda")
    H.run(H.op_from_nhwc(y, 2, 3, 5, 7, z, z.stride()))
    assert torch.equal(z.cpu(), x)
    a = rn(5, 300)
    ad = a.cuda(); o = torch.empty(300, device="cuda")
    H.run(H.op_silu(ad, o, 300))
    assert rel_err(o, F.silu(a.double())) < 1e-6
    ar = a.double().requires_grad_(True); F.silu(ar).sum().backward()
    dx = torch.ones(300, device="cuda")
    H.run(H.op_silu_bwd(ad, torch.ones(300, device="cuda"), dx, 300, acc=1))
    assert rel_err(dx - 1, ar.grad) < 1e-5
    yb = torch.ones(300, device="cuda")
    H.run(H.op_axpby(ad, yb, 300, alpha=2.0, beta=3.0))
    assert rel_err(yb, 2 * a + 3) < 1e-6
    table, idx = rn(6, 10, 16), torch.tensor([3, 3, 9])
    out = torch.zeros(3, 16, device="cuda")
    H.run(H.op_embedding(table.cuda(), idx.cuda(), 3, 16, out))
    assert torch.equal(out.cpu(), table[idx])
    dt = torch.zeros(10, 16, device="cuda")
    H.run(H.op_embedding_bwd(out, idx.cuda(), 3, 16, dt))
    ref = torch.zeros(10, 16); ref.index_add_(0, idx, table[idx])
    assert rel_err(dt, ref) < 1e-6
    buf = torch.ones(64, device="cuda"); dst = torch.zeros(64, device="cuda")
    H.run(H.op_copy(buf, dst, 256)); H.run(H.op_memset(buf, 256))
    assert dst.sum().item() == 64 and buf.sum().item() == 0


MATH_TOL = {1: 2e-2, 2: 1e-4, 3: 1e-5, 4: 1e-5}     # bf16 | 2 bf16 planes (3 products) | 3 exact bf16 planes (6 products): fp32 grade | 2 fp16 planes (3 products, forward patch kernel): fp32 grade


@pytest.mark.parametrize("math_mode", [1, 2, 3, 4])
@pytest.mark.parametrize("case", [(2, 12, 12, 64, 0, 96, 3, 1, 0, 0), (1, 10, 14, 32, 64, 64, 3, 1, 0, 64), (2, 8, 8, 32, 0, 32, 3, 1, 1, 0),
                                  (2, 16, 16, 64, 0, 128, 3, 2, 0, 0), (2, 9, 9, 64, 32, 32, 1, 1, 0, 0), (3, 32, 32, 128, 0, 128, 3, 1, 0, 128),
                                  (4, 64, 48, 64, 0, 160, 3, 1, 0, 0), (3, 32, 32, 96, 0, 64, 3, 1, 1, 0), (6, 32, 32, 32, 0, 256, 3, 1, 0, 0), (2, 24, 32, 64, 0, 64, 3, 1, 0, 0),
                                  (4, 8, 8, 128, 0, 64, 3, 1, 0, 0), (3, 8, 8, 64, 0, 32, 3, 1, 0, 0), (2, 4, 4, 64, 0, 64, 3, 1, 1, 0),
                                  (2, 16, 16, 96, 0, 64, 3, 1, 0, 0), (8, 64, 64, 32, 0, 32, 3, 1, 0, 0), (17, 8, 8, 64, 0, 64, 3, 1, 0, 0)])
def test_conv_bf16_split_modes(H, case, math_mode):
    """The bf16-MFMA split-operand variants of conv fwd / dgrad / wgrad against fp64."""
    N, Hh, W, C0, C1, Cout, k, stride, up, tile = case
    Cin = C0 + C1
    tol = MATH_TOL[math_mode]
    x = rn(1, N, Cin, Hh, W)
    w = rn(2, Cout, Cin, k, k, scale=1.0 / math.sqrt(Cin * k * k))
    b = rn(3, Cout, scale=0.1)
    c = H.Conv(N, Hh, W, C0, C1, Cout, k=k, stride=stride, up=bool(up), math=math_mode)
    xh = nhwc(x).cuda()
    x0 = xh[..., :C0].contiguous()
    x1 = xh[..., C0:].contiguous() if C1 else None
    wd, bd = nhwc(w).cuda(), b.cuda()
    y = torch.empty(N, c.Ho, c.Wo, Cout, device="cuda")
    H.run(H.op_conv_fwd(c, x0, x1, wd, bd, y, tile=tile))
    assert rel_err(nchw(y), ref_conv(x, w, b, stride, k // 2, up)) < tol
    wr = w.double().requires_grad_(True)
    dy = rn(5, N, Cout, c.Ho, c.Wo)
    xl = (F.interpolate(x, scale_factor=2, mode="nearest") if up else x).double().clone().requires_grad_(True)
    (F.conv2d(xl, wr, None, stride=stride, padding=k // 2) * dy.double()).sum().backward()
    dyd = nhwc(dy).cuda()
    dx = torch.empty(N, c.Hl, c.Wl, Cin, device="cuda")
    H.run(H.op_conv_dgrad(c, dyd, wd, dx, tile=tile))
    assert rel_err(nchw(dx), xl.grad) < tol
    # the fast path: LDS-patch kernel on fragment-ordered pre-split weights (forced: these shapes are too small to be chosen by the
    # fill heuristic), forward and -- on the transposed, tap-flipped weights -- data gradient
    nb = c.wprep_bytes(0, force=True)
    assert (nb > 0) == ((k == 3 and stride == 1 and C1 == 0 and C0 % 32 == 0 and c.Ho % 8 == 0 and (c.Wo % 16 == 0 or c.Wo == 8)) or
                        (k == 1 and stride == 1 and not up and C0 % 32 == 0 and C1 % 32 == 0 and Cout % 4 == 0 and Cout >= 32))
    if nb:
        wp = torch.empty(nb // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 0, wp))
        y2 = torch.empty_like(y)
        H.run(H.op_conv_fwd(c, x0, x1, wd, bd, y2, wp=wp))
        assert rel_err(nchw(y2), ref_conv(x, w, b, stride, k // 2, up)) < tol
    nb = c.wprep_bytes(1, force=True)
    if nb:
        wp_t = torch.empty(nb // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 1, wp_t))
        dx2 = torch.empty(N, c.Hl, c.Wl, Cin, device="cuda")
        H.run(H.op_conv_dgrad(c, dyd, wd, dx2, wp_t=wp_t))
        assert rel_err(nchw(dx2), xl.grad) < tol
    wsb = c.wgrad_ws_bytes()
    dw = torch.empty_like(wd)
    db = torch.empty(Cout, device="cuda")
    H.run(H.op_conv_wgrad(c, x0, x1, dyd, dw, ws(wsb), wsb, db=db))
    assert rel_err(dw.permute(0, 3, 1, 2), wr.grad) < 2 * tol
    assert rel_err(db, dy.double().sum((0, 2, 3))) < 1e-5          # exact fp32 sums in every math mode


@pytest.mark.parametrize("math_mode", [1, 3, 4])
@pytest.mark.parametrize("case", [(3, 16, 16, 128, 64, 96), (1, 8, 8, 256, 0, 64), (2, 12, 20, 64, 32, 36), (4, 32, 32, 64, 64, 160), (2, 16, 16, 96, 0, 64),
                                  (2, 12, 20, 96, 32, 64), (1, 16, 16, 32, 96, 96)])
def test_conv1x1_kernel(H, case, math_mode):
    """conv1x1.hip: dual-source 1x1 conv with bias / residual (full and half resolution), split-K on small layers, and the data
    gradient on the whole channel range and on one 32-aligned concat source, against fp64.  The last two cases have a 64-channel stage
    straddling the two sources (C0 % 64 != 0): the general (pointer) instantiation; the others run the buffer-load one, with a short
    last stage (C % 64 == 32) and a partial last pixel tile among them."""
    N, Hh, W, C0, C1, Cout = case
    Cin = C0 + C1
    tol = MATH_TOL[math_mode]
    x = rn(1, N, Cin, Hh, W)
    w = rn(2, Cout, Cin, 1, 1, scale=1.0 / math.sqrt(Cin))
    b = rn(3, Cout, scale=0.1)
    res = rn(4, N, Cout, Hh, W)
    res2 = rn(6, N, Cout, Hh // 2, W // 2)
    c = H.Conv(N, Hh, W, C0, C1, Cout, k=1, math=math_mode)
    xh = nhwc(x).cuda()
    x0 = xh[..., :C0].contiguous()
    x1 = xh[..., C0:].contiguous() if C1 else None
    wd, bd = nhwc(w).cuda(), b.cuda()
    nb = c.wprep_bytes(0)
    assert nb > 0
    wp = torch.empty(nb // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 0, wp))
    ref = F.conv2d(x.double(), w.double(), b.double())
    y = torch.empty(N, Hh, W, Cout, device="cuda")
    H.run(H.op_conv_fwd(c, x0, x1, wd, bd, y, wp=wp))
    assert rel_err(nchw(y), ref) < tol
    H.run(H.op_conv_fwd(c, x0, x1, wd, bd, y, res=nhwc(res).cuda(), res_mode=1, wp=wp))
    assert rel_err(nchw(y), ref + res.double()) < tol
    H.run(H.op_conv_fwd(c, x0, x1, wd, None, y, res=nhwc(res2).cuda(), res_mode=2, wp=wp))
    assert rel_err(nchw(y), ref - b.double().view(1, -1, 1, 1) + F.interpolate(res2.double(), scale_factor=2, mode="nearest")) < tol
    # data gradient
    dy = rn(5, N, Cout, Hh, W)
    dxr = F.conv_transpose2d(dy.double(), w.double())
    dyd = nhwc(dy).cuda()
    nbt = c.wprep_bytes(1)
    assert (nbt > 0) == (Cout % 32 == 0)            # dy is the A operand of the data gradient: 32-channel chunks
    if not nbt:
        return
    wp_t = torch.empty(nbt // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 1, wp_t))
    dx = torch.empty(N, Hh, W, Cin, device="cuda")
    H.run(H.op_conv_dgrad(c, dyd, wd, dx, wp_t=wp_t))
    assert rel_err(nchw(dx), dxr) < tol
    if C1 and C0 % 32 == 0:
        dx1 = torch.full((N, Hh, W, C1), 1.0, device="cuda")
        H.run(H.op_conv_dgrad(c, dyd, wd, dx1, ci_off=C0, ci_cnt=C1, accumulate=1, wp_t=wp_t))
        assert rel_err(nchw(dx1), dxr[:, C0:] + 1.0) < tol
    if math_mode == 4:          # fp16-format data gradient with the dynamic dY scale, for a tiny-magnitude gradient
        dys = dyd * 1e-6
        am = torch.empty(4, device="cuda")
        H.run(H.op_amax(dys, dys.numel(), am))
        wp_h = torch.empty(c.wprep_bytes(1, f16_grad=True) // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 1 | 16, wp_h))
        H.run(H.op_conv_dgrad(c, dys, wd, dx, wp_t=wp_h, dy_amax=am))
        assert rel_err(nchw(dx), dxr * 1e-6) < tol


@pytest.mark.parametrize("case", [(32, 256, 512, 0), (32, 1024, 512, 1), (7, 64, 128, 0), (32, 512, 4096, 0), (16, 768, 1032, 1), (1, 96, 72, 0)])
def test_skinny_linear(H, case):
    """skinny.hip (M <= 32 linear layers, one wave per output feature) through pdae_gemm, incl. strided operands and accumulate."""
    M, N, K, acc = case
    lda, ldb, ldc = K + 8, K + 4, N + 4
    A, B = rn(1, M, lda), rn(2, N, ldb)
    bias, C0 = rn(3, N), rn(4, M, ldc)
    ref = A[:, :K].double() @ B[:, :K].double().t() + bias.double() + (C0[:, :N].double() if acc else 0)
    Cd = C0.clone().cuda()
    H.run(H.op_gemm(0, 1, M, N, K, A.cuda(), lda, B.cuda(), ldb, Cd, ldc, bias=bias.cuda(), accumulate=acc))
    assert rel_err(Cd[:, :N], ref) < 1e-5
    assert torch.equal(Cd[:, N:].cpu(), C0[:, N:])          # padding columns untouched


@pytest.mark.parametrize("math_mode", [1, 3, 4])
@pytest.mark.parametrize("case", [(2, 16, 32, 64, 32, 64, 0, 1), (2, 8, 8, 32, 0, 32, 1, 2), (1, 24, 16, 96, 0, 160, 0, 0), (2, 32, 32, 128, 128, 128, 0, 1)])
def test_conv_with_fused_groupnorm_input(H, case, math_mode):
    """pdae_conv2d_fwd_gn: GroupNorm + AdaGN + SiLU applied inside the conv's patch staging (two-source concat, upsample, residuals,
    zero padding of the ACTIVATED tensor) against fp64 and against the unfused gn_apply -> conv pair."""
    N, Hh, W, C0, C1, Cout, up, res_mode = case
    C, G = C0 + C1, 32
    tol = MATH_TOL[math_mode]
    x = rn(1, N, C, Hh, W) * 1.5 + 0.7
    gamma, beta = 1 + 0.2 * rn(2, C), 0.2 * rn(3, C) + 0.5          # beta offset: silu(b - a mu) != 0 would expose wrong padding
    ss = 0.3 * rn(4, N, 2 * C)
    w = rn(5, Cout, C, 3, 3, scale=1.0 / math.sqrt(C * 9))
    b = rn(6, Cout, scale=0.1)
    c = H.Conv(N, Hh, W, C0, C1, Cout, k=3, up=bool(up), math=math_mode)
    a_ref = _gn_ref(x.double(), gamma.double(), beta.double(), ss.double(), None, 1)
    y_ref = ref_conv(a_ref, w, b, 1, 1, up)
    res = None
    if res_mode == 1:
        res = rn(7, N, Cout, c.Ho, c.Wo); y_ref = y_ref + res.double()
    elif res_mode == 2:
        res = rn(7, N, Cout, c.Ho // 2, c.Wo // 2); y_ref = y_ref + F.interpolate(res, scale_factor=2, mode="nearest").double()
    xh = nhwc(x).cuda()
    x0 = xh[..., :C0].contiguous()
    x1 = xh[..., C0:].contiguous() if C1 else None
    mean = torch.empty(N * G, device="cuda"); rstd = torch.empty(N * G, device="cuda")
    H.run(H.op_gn_stats(x0, C0, x1, C1, N, Hh * W, G, 1e-5, mean, rstd, ws(H.gn_ws_bytes(N, C))))
    coef = torch.empty(3, N, C, device="cuda")
    H.run(H.op_gn_coef(N, C, G, mean, rstd, gamma.cuda(), beta.cuda(), ss.cuda(), None, coef))
    nb = c.wprep_bytes(0, force=True, gn=True)
    assert nb > 0
    wd, bd = nhwc(w).cuda(), b.cuda()
    wp = torch.empty(nb // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 4, wp))
    y = torch.empty(N, c.Ho, c.Wo, Cout, device="cuda")
    resd = nhwc(res).cuda() if res is not None else None
    H.run(H.op_conv_fwd_gn(c, x0, x1, coef, 1, wp, bd, y, res=resd, res_mode=res_mode))
    assert rel_err(nchw(y), y_ref) < tol
    # unfused pair on the same kernels
    a = torch.empty(N, Hh, W, C, device="cuda")
    H.run(H.op_gn_apply(x0, C0, x1, C1, N, Hh, W, coef, 1, 0, a))
    c1 = H.Conv(N, Hh, W, C, 0, Cout, k=3, up=bool(up), math=math_mode)
    y2 = torch.empty_like(y)
    H.run(H.op_conv_fwd(c1, a, None, wd, bd, y2, res=resd, res_mode=res_mode))
    assert rel_err(y, y2) < tol


@pytest.mark.parametrize("math_mode", [1, 3, 4])
@pytest.mark.parametrize("case", [(16, 64, 32, 64, 64, 32, 64, True), (16, 64, 32, 64, 96, 0, 128, False), (32, 32, 32, 32, 32, 32, 32, True)])
def test_conv_with_fused_skip_connection(H, knob, case, math_mode):
    """pdae_conv2d_fwd_skip: conv3x3(in) + conv1x1([s0 | s1]) + both biases in one launch, with plain and fused-GroupNorm main input.
    (Fused skip chunks exist in the direct form only: the Winograd-along-x form of chip-filling layers is switched off here.)"""
    knob("PDAE_W1", "0")
    N, Hh, W, C, Cs0, Cs1, Cout, use_gn = case
    Cs, G = Cs0 + Cs1, 32
    tol = MATH_TOL[math_mode]
    x = rn(1, N, C, Hh, W) * 1.2 + 0.3
    sx = rn(2, N, Cs, Hh, W)
    w = rn(3, Cout, C, 3, 3, scale=1.0 / math.sqrt(C * 9)); b = rn(4, Cout, scale=0.1)
    wsk = rn(5, Cout, Cs, 1, 1, scale=1.0 / math.sqrt(Cs)); bsk = rn(6, Cout, scale=0.1)
    gamma, beta = 1 + 0.2 * rn(7, C), 0.2 * rn(8, C) + 0.4
    c = H.Conv(N, Hh, W, C, 0, Cout, k=3, math=math_mode)
    cs = H.Conv(N, Hh, W, Cs0, Cs1, Cout, k=1, math=math_mode)
    assert H.conv_fwd_skip_ok(c, cs)
    a_ref = _gn_ref(x.double(), gamma.double(), beta.double(), None, None, 1) if use_gn else x.double()
    y_ref = ref_conv(a_ref, w, b, 1, 1, 0) + F.conv2d(sx.double(), wsk.double(), bsk.double())
    xd = nhwc(x).cuda()
    sh = nhwc(sx).cuda()
    s0 = sh[..., :Cs0].contiguous()
    s1 = sh[..., Cs0:].contiguous() if Cs1 else None
    wd, wsd = nhwc(w).cuda(), nhwc(wsk).cuda()
    coef = None
    if use_gn:
        mean = torch.empty(N * G, device="cuda"); rstd = torch.empty(N * G, device="cuda")
        H.run(H.op_gn_stats(xd, C, None, 0, N, Hh * W, G, 1e-5, mean, rstd, ws(H.gn_ws_bytes(N, C))))
        coef = torch.empty(3, N, C, device="cuda")
        H.run(H.op_gn_coef(N, C, G, mean, rstd, gamma.cuda(), beta.cuda(), None, None, coef))
    nb = c.wprep_bytes(0, force=True, gn=use_gn)
    wp = torch.empty(nb // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 4 if use_gn else 0, wp))
    wps = torch.empty(H.conv_skip_wprep_bytes(c, cs) // 4, device="cuda")
    H.run(H.op_conv_skip_wprep(c, cs, wsd, wps))
    y = torch.empty(N, Hh, W, Cout, device="cuda")
    H.run(H.op_conv_fwd_skip(c, xd, None, coef, 1, wp, b.cuda(), cs, s0, s1, wps, bsk.cuda(), y))
    assert rel_err(nchw(y), y_ref) < tol


@pytest.mark.parametrize("case", [(8, 64, 64, 32, 0, 128, "plain"), (3, 16, 48, 32, 0, 256, "plain"), (5, 8, 8, 32, 0, 128, "plain"), (32, 32, 32, 32, 0, 128, "skip"),
                                  (32, 64, 64, 32, 32, 128, "gn"), (16, 64, 32, 64, 0, 128, "skip"), (32, 128, 128, 128, 0, 128, "plain"),
                                  # split-K launches (small layers): the statistics come from the slab reduction
                                  (32, 8, 8, 512, 0, 512, "plain"), (32, 16, 16, 384, 0, 384, "plain"), (4, 16, 16, 128, 0, 128, "gn"),
                                  (8, 32, 32, 256, 0, 256, "plain"), (3, 16, 16, 96, 32, 128, "gn")])
def test_groupnorm_statistics_from_the_producing_convolution(H, knob, case):
    """pdae_conv_stats_arm + pdae_gn_coef_from_conv_stats: the 3x3 forward kernels (plain, fused-GroupNorm input, fused skip, image-pair tiles,
    odd batch) leave per-wave (sum, sum of squares) of their OUTPUT behind, and the next GroupNorm's mean / rstd / coefficients computed from
    them match the statistics pass over the stored tensor -- alone and as the second source of a two-tensor concat."""
    N, Hh, W, C0, C1, Cout, form = case
    if form == "skip":
        knob("PDAE_W1", "0")          # fused skip chunks exist in the direct form only
    C, G = C0 + C1, 32
    x = rn(1, N, C, Hh, W) * 1.3 + 0.4
    w = rn(2, Cout, C, 3, 3, scale=1.0 / math.sqrt(C * 9)); b = rn(3, Cout, scale=0.5) + 0.8       # a mean well away from zero
    c = H.Conv(N, Hh, W, C0, C1, Cout, k=3, math=4)
    xh = nhwc(x).cuda()
    x0 = xh[..., :C0].contiguous(); x1 = xh[..., C0:].contiguous() if C1 else None
    wd, bd = nhwc(w).cuda(), b.cuda()
    y = torch.empty(N, Hh, W, Cout, device="cuda")
    cs = None
    if form == "skip":
        cs = H.Conv(N, Hh, W, C0, 0, Cout, k=1, math=4)
        assert H.conv_fwd_skip_ok(c, cs)
    nbytes, tpi = H.conv_stats_bytes(c, cs)
    assert nbytes > 0 and tpi > 0
    part = torch.full((nbytes // 4,), float("nan"), device="cuda")                              # every entry must be written
    if form == "plain":
        wp = torch.empty(c.wprep_bytes(0, force=True) // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 0, wp))
        H.run(H.op_conv_fwd(c, x0, None, wd, bd, y, wp=wp, stats=part))
    else:
        gamma, beta = (1 + 0.2 * rn(7, C)).cuda(), (0.2 * rn(8, C) + 0.4).cuda()
        mean = torch.empty(N * G, device="cuda"); rstd = torch.empty(N * G, device="cuda"); coef = torch.empty(3, N, C, device="cuda")
        H.run(H.op_gn_stats_coef(x0, C0, x1, C1, N, Hh * W, G, 1e-5, gamma, beta, None, None, mean, rstd, coef, ws(H.gn_ws_bytes(N, C))))
        wp = torch.empty(c.wprep_bytes(0, force=True, gn=True) // 4, device="cuda")
        H.run(H.op_conv_wprep(c, wd, 4, wp))
        if form == "gn":
            H.run(H.op_conv_fwd_gn(c, x0, x1, coef, 1, wp, bd, y, stats=part))
        else:
            wsk = rn(5, Cout, C0, 1, 1, scale=1.0 / math.sqrt(C0)); bsk = rn(6, Cout, scale=0.1)
            wps = torch.empty(H.conv_skip_wprep_bytes(c, cs) // 4, device="cuda")
            H.run(H.op_conv_skip_wprep(c, cs, nhwc(wsk).cuda(), wps))
            H.run(H.op_conv_fwd_skip(c, x0, None, coef, 1, wp, bd, cs, x0, None, wps, bsk.cuda(), y, stats=part))
    assert torch.isfinite(part).all()
    # the same convolution without the request stores the same tensor
    y_plain = torch.empty_like(y)
    if form == "plain":
        H.run(H.op_conv_fwd(c, x0, None, wd, bd, y_plain, wp=wp))
        assert torch.equal(y, y_plain)
    g2, b2 = (1 + 0.1 * rn(9, Cout)).cuda(), (0.1 * rn(10, Cout)).cuda()
    ss = (0.3 * rn(11, N, 2 * Cout)).cuda()

    def both(C0_, x0_, p0, C1_, x1_, p1, gam, bet, ss_):
        Ct = C0_ + C1_
        m_a, r_a, k_a = torch.empty(N * G, device="cuda"), torch.empty(N * G, device="cuda"), torch.empty(3, N, Ct, device="cuda")
        m_b, r_b, k_b = torch.empty_like(m_a), torch.empty_like(r_a), torch.empty_like(k_a)
        H.run(H.op_gn_stats_coef(x0_, C0_, x1_, C1_, N, Hh * W, G, 1e-5, gam, bet, ss_, None, m_a, r_a, k_a, ws(H.gn_ws_bytes(N, Ct))))
        H.run(H.op_gn_coef_from_conv_stats(N, Hh * W, C0_, C1_, G, 1e-5, p0, tpi, p1, tpi if p1 is not None else 0, gam, bet, ss_, None, m_b, r_b, k_b))
        assert (m_a - m_b).abs().max() < 2e-6 * max(1.0, float(m_a.abs().max()))
        assert rel_err(r_b, r_a) < 5e-6 and rel_err(k_b, k_a) < 5e-6
    both(Cout, y, part, 0, None, None, g2, b2, ss)
    if Cout % 8 == 0:                                           # groups of the concat [y | y] are still whole channel quads
        g3, b3 = torch.cat([g2, g2 * 0.9]), torch.cat([b2, b2 + 0.1])
        both(Cout, y, part, Cout, y, part, g3, b3, None)


@pytest.mark.parametrize("gscale", [1.0, 3e-7, 2e4])
@pytest.mark.parametrize("case", [(4, 32, 32, 64, 128), (2, 16, 48, 128, 96), (3, 8, 8, 64, 64), (8, 64, 64, 32, 32),
                                  (16, 8, 8, 64, 64), (17, 8, 8, 32, 96), (8, 16, 8, 64, 32)])      # 8-wide: image-pair tiles in conv3x3w
def test_f16_format_gradient_kernels_with_dynamic_scale(H, case, gscale):
    """math 4 with dy_amax: data gradient (patch kernel on transposed fp16-format weights) and weight gradient (conv3x3w, fp16 planes) with
    the per-tensor power-of-two dY scale, for gradients of ordinary, tiny (3e-7) and large magnitude -- fp32-grade at every scale."""
    N, Hh, W, Cin, Cout = case
    x = rn(1, N, Cin, Hh, W)
    w = rn(2, Cout, Cin, 3, 3, scale=1.0 / math.sqrt(Cin * 9))
    dy = rn(5, N, Cout, Hh, W) * gscale
    c = H.Conv(N, Hh, W, Cin, 0, Cout, k=3, math=4)
    xl = x.double().clone().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    (F.conv2d(xl, wr, None, padding=1) * dy.double()).sum().backward()
    xd, wd, dyd = nhwc(x).cuda(), nhwc(w).cuda(), nhwc(dy).cuda()
    am = torch.empty(4, device="cuda")
    H.run(H.op_amax(dyd, dyd.numel(), am))
    assert abs(float(am[0]) - float(dy.abs().max())) <= 1e-6 * float(dy.abs().max())
    nbt = c.wprep_bytes(1, force=True, f16_grad=True)
    assert nbt > 0
    wp_t = torch.empty(nbt // 4, device="cuda")
    H.run(H.op_conv_wprep(c, wd, 1 | 16, wp_t))
    dx = torch.empty(N, Hh, W, Cin, device="cuda")
    H.run(H.op_conv_dgrad(c, dyd, wd, dx, wp_t=wp_t, dy_amax=am))
    assert rel_err(nchw(dx), xl.grad) < 1e-5
    wsb = c.wgrad_ws_bytes()
    dw, db = torch.empty_like(wd), torch.empty(Cout, device="cuda")
    H.run(H.op_conv_wgrad(c, xd, None, dyd, dw, ws(wsb), wsb, db=db, dy_amax=am))
    assert rel_err(dw.permute(0, 3, 1, 2), wr.grad) < 2e-5
    assert rel_err(db, dy.double().sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("math_mode", [1, 2, 4])
@pytest.mark.parametrize("case", [(8, 16, 16, 128, 128, 128), (4, 32, 32, 64, 0, 96), (2, 48, 24, 256, 128, 160), (16, 16, 16, 384, 0, 1152), (3, 40, 28, 32, 32, 36)])
def test_conv1x1_weight_gradient_kernel(H, case, math_mode):
    """conv3x3w.hip: conv1x1w_kernel -- dW and the riding bias gradient of a (dual-source) 1x1 convolution, ragged pixel tiles, channel counts
    that do not fill a 128-channel block, accumulate, and a tiny-magnitude dY in the fp16 format (dynamic scale); fp64 reference."""
    N, Hh, W, C0, C1, Cout = case
    Cin = C0 + C1
    tol = MATH_TOL[math_mode]
    x, dy = rn(1, N, Cin, Hh, W), rn(2, N, Cout, Hh, W)
    dw_ref = torch.einsum("nohw,nihw->oi", dy.double(), x.double())
    db_ref = dy.double().sum((0, 2, 3))
    c = H.Conv(N, Hh, W, C0, C1, Cout, k=1, math=math_mode)
    xh = nhwc(x).cuda()
    x0 = xh[..., :C0].contiguous()
    x1 = xh[..., C0:].contiguous() if C1 else None
    for scale in (1.0, 1e-6):
        dyd = nhwc(dy).cuda() * scale
        am = torch.empty(4, device="cuda")
        H.run(H.op_amax(dyd, dyd.numel(), am))
        wsb = c.wgrad_ws_bytes()
        dw = torch.full((Cout, 1, 1, Cin), float("nan"), device="cuda")
        db = torch.full((Cout,), float("nan"), device="cuda")
        H.run(H.op_conv_wgrad(c, x0, x1, dyd, dw, ws(wsb), wsb, db=db, dy_amax=am if math_mode == 4 else None))
        assert rel_err(dw.reshape(Cout, Cin), dw_ref * scale) < 2 * tol and rel_err(db, db_ref * scale) < 1e-5
        H.run(H.op_conv_wgrad(c, x0, x1, dyd, dw, ws(wsb), wsb, accumulate=1, db=db, dy_amax=am if math_mode == 4 else None))
        assert rel_err(dw.reshape(Cout, Cin), 2 * dw_ref * scale) < 2 * tol and rel_err(db, 2 * db_ref * scale) < 1e-5


def test_grouped_linear_forward_and_backward(H):
    """pdae_linear_group / pdae_linear_bwd_group: several M <= 32 linear layers (different widths, two different inputs) in one launch each,
    against fp64; accumulate flags; an item without dx; every output element written."""
    M, K = 32, 512
    g = torch.Generator().manual_seed(1)
    xa, xb = torch.randn(M, K, generator=g), torch.randn(M, K, generator=g)
    specs = [(xa, 256), (xa, 1024), (xb, 768), (xb, 36)]
    Ws = [torch.randn(n, K, generator=g) / 22 for _, n in specs]
    bs = [torch.randn(n, generator=g) for _, n in specs]
    xd = {id(xa): xa.cuda(), id(xb): xb.cuda()}
    ys = [torch.full((M, n), float("nan"), device="cuda") for _, n in specs]
    Wd, bd = [w.cuda() for w in Ws], [b.cuda() for b in bs]
    items = [(xd[id(x)], Wd[i], bd[i], ys[i]) for i, (x, n) in enumerate(specs)]
    it, first, total = H.linear_group_tables(items, "cuda")
    H.run(H.op_linear_group(it, first, len(items), total, M, K))
    for i, (x, n) in enumerate(specs):
        assert rel_err(ys[i], x.double() @ Ws[i].double().T + bs[i].double()) < 1e-6
    dys = [torch.randn(M, n, generator=g) for _, n in specs]
    dyd = [d.cuda() for d in dys]
    dW = [torch.ones(n, K, device="cuda") for _, n in specs]
    db = [torch.ones(n, device="cuda") for _, n in specs]
    dxa, dxb = torch.full((M, K), 2.0, device="cuda"), torch.full((M, K), float("nan"), device="cuda")
    # item 0: accumulate into dW/db and into dxa; item 1: no dx; item 2: overwrite dxb; item 3: ragged width, no dx
    bw = [(xd[id(xa)], dyd[0], Wd[0], dW[0], db[0], dxa, 1, 1), (xd[id(xa)], dyd[1], Wd[1], dW[1], db[1], None, 0, 0),
          (xd[id(xb)], dyd[2], Wd[2], dW[2], db[2], dxb, 0, 0), (xd[id(xb)], dyd[3], Wd[3], dW[3], db[3], None, 0, 0)]
    it, first, total = H.linear_bwd_group_tables(bw, M, "cuda")
    H.run(H.op_linear_bwd_group(it, first, len(bw), total, M, K))
    for i, (x, n) in enumerate(specs):
        ref_w = dys[i].double().T @ x.double() + (1.0 if i == 0 else 0.0)
        ref_b = dys[i].double().sum(0) + (1.0 if i == 0 else 0.0)
        assert rel_err(dW[i], ref_w) < 1e-6 and rel_err(db[i], ref_b) < 1e-6, i
    assert rel_err(dxa, dys[0].double() @ Ws[0].double() + 2.0) < 1e-6
    assert rel_err(dxb, dys[2].double() @ Ws[2].double()) < 1e-6


def test_grouped_linear_with_a_sampling_batch_over_32_rows(H):
    """Builder.linear_group at the evaluator's batch (100 rows): every layer enters the one launch as row slices of <= 32 rows (per-item `rows`
    of pdae_linear_item); results equal the fp64 product and every output element is written."""
    from types import SimpleNamespace
    from pdae_amd.engine import Builder, Plan
    Nb, K = 100, 512
    g = torch.Generator().manual_seed(3)
    P = {}
    for i, n in enumerate((256, 1024, 36)):
        P[f"l{i}.weight"] = (torch.randn(n, K, generator=g) / 22).cuda()
        P[f"l{i}.bias"] = torch.randn(n, generator=g).cuda()
    x = torch.randn(Nb, K, generator=g).cuda()
    pl = Plan("cuda")
    B = Builder(pl, P, None)
    res = B.linear_group([(x, f"l{i}") for i in range(3)])
    assert [r.kind for r in pl.recs] == [H.OP_LINEAR_GROUP]
    for y, _ in res:
        y.fill_(float("nan"))
    pl.finalize() if hasattr(pl, "finalize") else None
    H.run(pl.recs[0])
    for i, (y, ctx) in enumerate(res):
        assert ctx.Nb == Nb and rel_err(y, x.double() @ P[f"l{i}.weight"].double().T + P[f"l{i}.bias"].double()) < 1e-6


def test_grouped_weight_preparation_equals_the_single_launches(H, knob):
    """pdae_conv_wprep_group: every prepared-weight form the engine uses (3x3 forward, fused-GroupNorm two-source, data-gradient in the exact
    and the fp16-gradient format, 1x1 forward / data gradient, fused skip chunks; bf16 and split formats) written by ONE launch from a job
    table equals, bit for bit, what pdae_conv_wprep / pdae_conv_skip_wprep write one launch at a time."""
    knob("PDAE_W1", "0")              # the fused-skip job below exists in the direct form only (Winograd-form jobs: tests/test_conv3x3x_gpu.py)
    jobs, singles, keep = [], [], []                # a job holds raw pointers: the weights must outlive the grouped launch

    def add(c, w, flags, nbytes):
        keep.append(w)
        a = torch.full((nbytes // 4,), float("nan"), device="cuda"); b = torch.full((nbytes // 4,), float("nan"), device="cuda")
        H.run(H.op_conv_wprep(c, w, flags, a))
        jobs.append(H.wprep_job(c, w, flags, b)); singles.append((a, b))

    for math_mode in (4, 1, 3):
        c3 = H.Conv(2, 16, 16, 64, 0, 96, k=3, math=math_mode)
        w3 = (rn(1, 96, 3, 3, 64) * 0.05).cuda()
        add(c3, w3, 0, c3.wprep_bytes(0, force=True))
        add(c3, w3, 1, c3.wprep_bytes(1, force=True))
        if math_mode == 4:
            add(c3, w3, 1 | 16, c3.wprep_bytes(1, force=True, f16_grad=True))
        cg = H.Conv(2, 16, 16, 32, 32, 96, k=3, math=math_mode)
        add(cg, w3, 4, cg.wprep_bytes(0, force=True, gn=True))
        c1 = H.Conv(3, 16, 16, 128, 64, 96, k=1, math=math_mode)
        w1 = (rn(2, 96, 1, 1, 192) * 0.05).cuda()
        add(c1, w1, 0, c1.wprep_bytes(0, force=True))
        add(c1, w1, 1, c1.wprep_bytes(1, force=True))
    c = H.Conv(16, 64, 32, 64, 0, 128, k=3, math=4); cs = H.Conv(16, 64, 32, 64, 32, 128, k=1, math=4)
    assert H.conv_fwd_skip_ok(c, cs)
    wsk = (rn(3, 128, 1, 1, 96) * 0.05).cuda()
    nb = H.conv_skip_wprep_bytes(c, cs)
    a = torch.full((nb // 4,), float("nan"), device="cuda"); b = torch.full((nb // 4,), float("nan"), device="cuda")
    H.run(H.op_conv_skip_wprep(c, cs, wsk, a))
    jobs.append(H.skip_wprep_job(c, cs, wsk, b)); singles.append((a, b))
    jt, ft, tot = H.wprep_group_tables(jobs, "cuda")
    H.run(H.op_conv_wprep_group(jt, ft, len(jobs), tot))
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(singles):
        fin = torch.isfinite(a)                     # split-K slab space behind the planes is not written by a preparation
        assert torch.equal(a[fin].view(torch.int32), b[fin].view(torch.int32)) and torch.equal(fin, torch.isfinite(b)), k


@pytest.mark.parametrize("math_mode", [1, 3, 4])
@pytest.mark.parametrize("case", [(8, 32, 32, 64, 32, 64, 0, 1), (16, 16, 32, 32, 0, 96, 0, 1), (8, 16, 16, 96, 64, 160, 1, 1), (8, 32, 32, 128, 128, 128, 0, 1),
                                  (22, 24, 16, 32, 32, 36, 0, 0)])      # (the weight-gradient kernel takes launches of >= 64 pixel tiles)
def test_weight_gradient_with_fused_groupnorm_input(H, case, math_mode):
    """pdae_conv_gn_input_arm + pdae_conv2d_wgrad: dW / db of conv3x3(act(GroupNorm(x))) with the activation recomputed on the RAW two-source input
    inside the weight-gradient kernel's X staging (round 5: the in_layers stage of a trained ResBlock never writes its activated tensor,
    module.py:241-242) -- against fp64 autograd and against the same kernel fed the materialised activation."""
    N, Hh, W, C0, C1, Cout, up, act = case
    C, G = C0 + C1, 32
    x = rn(1, N, C, Hh, W) * 1.5 + 0.7
    gamma, beta = 1 + 0.2 * rn(2, C), 0.2 * rn(3, C) + 0.5          # beta offset: act(b - a mu) != 0 would expose a wrong padding
    c = H.Conv(N, Hh, W, C0, C1, Cout, k=3, up=bool(up), math=math_mode)
    dy = rn(5, N, Cout, c.Ho, c.Wo) * 3e-3
    w = rn(6, Cout, C, 3, 3, scale=1.0 / math.sqrt(C * 9))
    a_ref = _gn_ref(x.double(), gamma.double(), beta.double(), None, None, act)
    wr = w.double().requires_grad_(True)
    (ref_conv(a_ref, wr, None, 1, 1, up) * dy.double()).sum().backward()
    assert H.conv_wgrad_gn_ok(c)
    xh = nhwc(x).cuda()
    x0 = xh[..., :C0].contiguous()
    x1 = xh[..., C0:].contiguous() if C1 else None
    mean = torch.empty(N * G, device="cuda"); rstd = torch.empty(N * G, device="cuda")
    H.run(H.op_gn_stats(x0, C0, x1, C1, N, Hh * W, G, 1e-5, mean, rstd, ws(H.gn_ws_bytes(N, C))))
    coef = torch.empty(3, N, C, device="cuda")
    H.run(H.op_gn_coef(N, C, G, mean, rstd, gamma.cuda(), beta.cuda(), None, None, coef))
    dyd = nhwc(dy).cuda()
    am = torch.empty(4, device="cuda")
    H.run(H.op_amax(dyd, dyd.numel(), am))
    wsb = c.wgrad_ws_bytes()
    dw, db = torch.empty(Cout, 3, 3, C, device="cuda"), torch.empty(Cout, device="cuda")
    H.run(H.op_conv_wgrad(c, x0, x1, dyd, dw, ws(wsb), wsb, db=db, dy_amax=am if math_mode == 4 else None, gn_coef=coef, gn_act=act))
    tol = {1: 2e-2, 3: 2e-5, 4: 2e-5}[math_mode]
    assert rel_err(dw.permute(0, 3, 1, 2), wr.grad) < tol
    assert rel_err(db, dy.double().sum((0, 2, 3))) < 1e-5
    # the materialised form on the same kernel: gn_apply -> single-source weight gradient
    a = torch.empty(N, Hh, W, C, device="cuda")
    H.run(H.op_gn_apply(x0, C0, x1, C1, N, Hh, W, coef, act, 0, a))
    c1 = H.Conv(N, Hh, W, C, 0, Cout, k=3, up=bool(up), math=math_mode)
    wsb1 = c1.wgrad_ws_bytes()
    dw2 = torch.empty_like(dw)
    H.run(H.op_conv_wgrad(c1, a, None, dyd, dw2, ws(wsb1), wsb1, dy_amax=am if math_mode == 4 else None))
    assert rel_err(dw, dw2) < (1e-6 if math_mode != 1 else 1e-2)
    # the request is one-shot: the next plain launch on the raw tensor must NOT see it
    if C1 == 0:
        dw3 = torch.empty_like(dw)
        H.run(H.op_conv_wgrad(c1, x0, None, dyd, dw3, ws(wsb1), wsb1, dy_amax=am if math_mode == 4 else None))
        assert rel_err(dw3, dw2) > 1e-2


def test_weight_gradient_gn_input_refuses_ineligible_shapes(H):
    assert not H.conv_wgrad_gn_ok(H.Conv(2, 8, 8, 32, 0, 32, k=3, math=4))          # 8-wide: image-pair tiles
    assert not H.conv_wgrad_gn_ok(H.Conv(2, 16, 16, 48, 16, 32, k=3, math=4))       # sources are not whole 32-channel chunks
    assert not H.conv_wgrad_gn_ok(H.Conv(2, 16, 16, 32, 0, 32, k=1, pad=0, math=4))
    c = H.Conv(2, 8, 8, 32, 0, 32, k=3, math=3)
    x = torch.zeros(2, 8, 8, 32, device="cuda"); coef = torch.zeros(3, 2, 32, device="cuda"); dw = torch.zeros(32, 3, 3, 32, device="cuda")
    wsb = c.wgrad_ws_bytes()
    with pytest.raises(H.PdaeError):
        H.run(H.op_conv_wgrad(c, x, None, x, dw, ws(wsb), wsb, gn_coef=coef, gn_act=1))


@pytest.mark.parametrize("case", [(3, 32, 32, 128, 0), (2, 16, 64, 64, 192), (2, 64, 64, 128, 128)])
def test_statistics_pass_in_the_producers_format(H, case):
    """pdae_gn_stats_quads (round 5): one pass leaves (sum, sum of squares) per (image, run of pixels, channel quad) for a tensor whose producer has no
    statistics epilogue (the stem); pdae_gn_coef_from_conv_stats on them (alone, or as the SECOND source of a concat whose first source has its own
    partials from another run length) equals the statistics pass pdae_gn_stats_coef on the same tensors."""
    N, Hh, W, C, C0 = case                       # C0 > 0: the tensor is the second source behind a C0-channel first one
    G, HW = 32, Hh * W
    x = (rn(1, N, Hh, W, C) * 1.3 + 0.4).cuda()
    x0 = (rn(2, N, Hh, W, C0) * 0.7 - 0.2).cuda() if C0 else None
    Ct = C + C0
    gamma, beta = (1 + 0.1 * rn(3, Ct)).cuda(), (0.1 * rn(4, Ct)).cuda()
    tiles = HW // 128
    part = torch.empty(N * tiles * (C // 4) * 2, device="cuda")
    H.run(H.op_gn_stats_quads(x, N, HW, C, tiles, part))
    ref = x.double().view(N, tiles, 128, C // 4, 4)
    p = part.view(N, tiles, C // 4, 2).double()
    assert rel_err(p[..., 0], ref.sum((2, 4))) < 1e-5 and rel_err(p[..., 1], (ref * ref).sum((2, 4))) < 1e-5
    mean, rstd, coef = torch.empty(N * G, device="cuda"), torch.empty(N * G, device="cuda"), torch.empty(3, N, Ct, device="cuda")
    mean2, rstd2, coef2 = torch.empty_like(mean), torch.empty_like(rstd), torch.empty_like(coef)
    if C0:
        t0 = HW // 64                                # the first source's partials come in runs of 64 pixels
        part0 = torch.empty(N * t0 * (C0 // 4) * 2, device="cuda")
        H.run(H.op_gn_stats_quads(x0, N, HW, C0, t0, part0))
        H.run(H.op_gn_coef_from_conv_stats(N, HW, C0, C, G, 1e-5, part0, t0, part, tiles, gamma, beta, None, None, mean, rstd, coef))
        H.run(H.op_gn_stats_coef(x0, C0, x, C, N, HW, G, 1e-5, gamma, beta, None, None, mean2, rstd2, coef2, ws(H.gn_ws_bytes(N, Ct))))
    else:
        H.run(H.op_gn_coef_from_conv_stats(N, HW, C, 0, G, 1e-5, part, tiles, None, 0, gamma, beta, None, None, mean, rstd, coef))
        H.run(H.op_gn_stats_coef(x, C, None, 0, N, HW, G, 1e-5, gamma, beta, None, None, mean2, rstd2, coef2, ws(H.gn_ws_bytes(N, Ct))))
    assert rel_err(mean, mean2) < 5e-6 and rel_err(rstd, rstd2) < 2e-5 and rel_err(coef, coef2) < 2e-5
