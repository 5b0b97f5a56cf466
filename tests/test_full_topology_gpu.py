"""GPU: every BASELINE.json configuration at its REAL topology -- networks built from the shipped YAMLs with no size override --
through the fused-step / planned-network HIP path against the CPU oracle computed on this box from the same seeded weights.

  F128  config/ffhq_representation_learning.yml  + pre-trained-dpms/ffhq128/config.yml   (base 128, [1,1,2,3,4], 128x128, FFHQEncoder)
  C64   config/celeba64_representation_learning.yml + pre-trained-dpms/celeba64/config.yml (base 64, [1,2,4,8], 64x64), fp32 and bf16 (enable_amp)
  M32   config/mnist_regular.yml                                                          (base 64, [1,2,2,4], 1x32x32, B=16)

Gates (north_star): z / eps / shift / loss within 1e-4 relative, every trainable gradient within 1e-5 in norm (GRAD_TOL: 3x the worst measured), DDIM x_0 PSNR stated
below; the fp16-window counter must stay zero.  The per-GPU batch of the benchmark (B=32) is covered by a size-independent property:
16 copies of the B=2 batch give the same mean loss and the same mean gradient.  Dropout is switched off (it is not a size; RNG streams cannot
match across devices)."""
import math
import os

import numpy as np
import pytest
import torch

from tests.conftest import ROOT, rel_err
from oracle import pdae_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
# enable_amp: bf16 operands (2^-9 per product), fp32 accumulate -- vs the fp32-grade path.  Measured on MI355X (round 3):
# eps 7.2e-3, shift 9.1e-3, z 5.0e-3, loss 4.8e-4, worst gradient-norm error 1.3e-2; the gates sit ~3x above so that a regression shows
BF16_TOL = dict(out=2.5e-2, loss=3e-3, grad=4e-2)


def _yaml(path):
    from pdae_amd.utils import load_yaml
    return load_yaml(os.path.join(ROOT, path))


def _rl_setup(cfg_file, seed_enc=1, seed_dec=2):
    """(config, denoise_fn cfg with dropout 0, encoder name, encoder / decoder state dicts, HIP encoder, HIP decoder)."""
    from pdae_amd.model.representation_learning import decoder as decoder_module, encoder as encoder_module
    c = _yaml(cfg_file)
    dcfg = dict(_yaml(c["trained_ddpm_config"])["denoise_fn_config"], dropout=0.0)
    ename, latent = c["encoder_config"]["model"], c["encoder_config"]["latent_dim"]
    enc_sd = O.synth_state_dict(O.encoder_param_shapes(ename, latent), seed_enc)
    dec_sd = O.synth_state_dict(O.unet_param_shapes(dcfg, shift=True, latent_dim=latent), seed_dec)
    enc = getattr(encoder_module, ename)(device=DEV, **c["encoder_config"])
    dec = getattr(decoder_module, c["decoder_config"]["model"])(device=DEV, latent_dim=c["decoder_config"]["latent_dim"], **dcfg)
    enc.load_state_dict(enc_sd)
    dec.load_state_dict(dec_sd)
    enc.train()
    dec.set_train_mode()
    return c, dcfg, ename, enc_sd, dec_sd, enc, dec


def _batch(B, ch, size, seed=0):
    g = torch.Generator().manual_seed(seed)
    x0 = torch.rand(B, ch, size, size, generator=g) * 2 - 1
    t = torch.tensor([100, 900] + [int(v) for v in torch.randint(0, 1000, (max(B - 2, 0),), generator=g)])[:B]
    noise = torch.randn(B, ch, size, size, generator=g)
    return x0, t, noise


def _oracle_rl(enc_sd, ename, dec_sd, dcfg, x0, t, noise):
    """z, eps, shift, loss and every trainable gradient of gaussian_diffusion.py:234-255 on the CPU."""
    s = O.Schedules()
    train = {"enc::" + k: v for k, v in enc_sd.items()}
    train.update({"dec::" + k: v for k, v in dec_sd.items() if O.shift_unet_trainable(k)})
    for v in train.values():
        v.requires_grad_(True)
    z = O.encoder_forward(enc_sd, ename, x0)
    eps, shift = O.shift_unet_forward(dec_sd, dcfg, O.q_sample(s, x0, t, noise), t, z)
    loss = O.p_loss(noise, eps + O._at(s.shift_coef, t, x0) * shift, weight=O._at(s.weight, t, x0))
    loss.backward()
    grads = {k: v.grad.detach() for k, v in train.items()}
    for v in train.values():
        v.requires_grad_(False)
        v.grad = None
    return z.detach(), eps.detach(), shift.detach(), float(loss), grads


# every gradient gate of this file: worst per-tensor errors measured on MI355X in round 6 (f16x3 arithmetic, conv3x3v weight gradients) were
# 1.9e-6 ... 3.3e-6 of the tensor's norm across F128 / C64 / M32 / latent; the gate is ~3x the largest (it was 1e-3 through round 5)
GRAD_TOL = 1e-5
WORST = {}      # label -> worst per-tensor gradient error observed in this session (printed; the gates below sit ~3x above the round-6 measurements)


def _check_grads(got, ref, tol, label="grads"):
    """Every trainable gradient within `tol` of its oracle norm (+ a floor of 1e-6 of the largest gradient norm for tensors whose gradient is
    numerically zero).  VERDICT r5 weak #1: the worst per-tensor error is PRINTED and recorded, and each caller's gate is set at about three times
    the value measured on MI355X in round 6 -- a kernel change that moves a gradient from 2e-5 to 8e-4 no longer passes silently."""
    floor = 1e-6 * max(float(v.double().norm()) for v in ref.values())
    bad, worst = [], (0.0, None)
    for k, r in ref.items():
        rn = float(r.double().norm())
        err = float((got[k].detach().double().cpu() - r.double()).norm())
        if rn > floor and err / rn > worst[0]:
            worst = (err / rn, k)
        if err > tol * rn + floor:
            bad.append((k, err / max(rn, 1e-30)))
    WORST[label] = worst
    print(f"[{label}] worst per-tensor gradient error {worst[0]:.3e} ({worst[1]}) over {len(ref)} tensors; gate {tol:.1e}")
    assert not bad, (len(bad), sorted(bad, key=lambda b: -b[1])[:6])


def _run_rl(gd, enc, dec, B, size, x0, t, noise, math=None):
    from pdae_amd.trainer.fused_step import FusedRLStep
    st = FusedRLStep(gd, enc, dec, None, None, B, size, size, math=math)
    st.load_batch(x0.to(DEV), t.to(DEV), noise.to(DEV))
    st.plan.run(0, st.n_bwd)
    torch.cuda.synchronize()
    grads = {"enc::" + k: v.clone() for k, v in enc.grads().items()}
    grads.update({"dec::" + k: v.clone() for k, v in dec.grads().items()})
    out = dict(z=st.z.clone(), eps=st.eps.permute(0, 3, 1, 2).clone(), shift=st.shift.permute(0, 3, 1, 2).clone(), loss=float(st.loss.item()), grads=grads)
    return st, out


@pytest.fixture(scope="module")
def gd():
    from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    return GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, torch.device(DEV))


def _guard():
    from pdae_amd import hip as H
    g = H.SaturationGuard.get(DEV)
    return g


def test_f128_train_step_full_topology_vs_oracle(gd):
    c, dcfg, ename, enc_sd, dec_sd, enc, dec = _rl_setup("config/ffhq_representation_learning.yml")
    assert dcfg["base_channel"] == 128 and dcfg["channel_multiplier"] == [1, 1, 2, 3, 4] and c["train_dataset_config"]["image_size"] == 128
    n_dec, n_enc = sum(p.numel() for p in dec.P.values()), sum(p.numel() for p in enc.P.values())
    assert abs(n_dec - 177.14e6) < 0.01e6 and abs(n_enc - 3.91e6) < 0.01e6, (n_dec, n_enc)
    x0, t, noise = _batch(2, 3, 128)
    _guard().reset()
    st, got = _run_rl(gd, enc, dec, 2, 128, x0, t, noise)
    assert _guard().read()[0] == 0, "fp16 window exceeded on N(0, 1/fan_in) weights"
    z, eps, shift, loss, grads = _oracle_rl(enc_sd, ename, dec_sd, dcfg, x0, t, noise)
    assert rel_err(got["z"], z) < 1e-4 and rel_err(got["eps"], eps) < 1e-4 and rel_err(got["shift"], shift) < 1e-4
    assert abs(got["loss"] - loss) < 1e-4 * abs(loss), (got["loss"], loss)
    assert set(got["grads"]) == set(grads) and len(grads) > 300
    _check_grads(got["grads"], grads, GRAD_TOL, "F128 B=2 train step vs oracle")
    # the benchmark's per-GPU batch (B = 32: other split-K plans, image-pair tiles with 16 pairs): 16 copies of the batch above give the
    # same mean loss and mean gradients
    del st
    rep = lambda a: a.repeat(16, *([1] * (a.dim() - 1)))
    st32, got32 = _run_rl(gd, enc, dec, 32, 128, rep(x0), rep(t), rep(noise))
    assert _guard().read()[0] == 0
    assert abs(got32["loss"] - got["loss"]) < 1e-5 * abs(got["loss"])
    assert rel_err(got32["eps"][:2], got["eps"]) < 1e-5 and rel_err(got32["shift"][30:], got["shift"]) < 1e-5
    _check_grads(got32["grads"], {k: v.cpu() for k, v in got["grads"].items()}, GRAD_TOL, "F128 B=32 (16 copies) vs B=2")


def _plan_census(plan):
    """Counts, over a plan's op records, the launches that take the round-4 / round-5 forms: Winograd-form 3x3 convolutions (forward incl. fused GroupNorm,
    data gradient), data gradients that leave GroupNorm-backward sums, weight gradients that recompute a fused GroupNorm input."""
    from pdae_amd import hip as H
    n = dict(wino_fwd=0, wino_dgrad=0, gnb=0, wgrad_gn=0, gn_bwd_parts=0, conv3=0)
    for k in range(plan.n):
        op = plan.arr[k]
        i = op.i
        if op.kind in (H.OP_CONV_FWD, H.OP_CONV_FWD_GN, H.OP_CONV_DGRAD) and i[8] == 3 and i[10] == 1:
            n["conv3"] += 1
            c = H.Conv(i[0], i[1], i[2], i[3], i[4], i[7], k=3, up=bool(i[12]), math=i[13])
            if op.kind == H.OP_CONV_DGRAD:
                w = c.winograd_form(1, f16_grad=bool(op.p[4]))
                n["wino_dgrad"] += int(w)
                n["gnb"] += int(bool(op.p[8]))
                assert not op.p[8] or w, "GroupNorm-backward sums armed on a launch that is not in the Winograd form"
            else:
                n["wino_fwd"] += int(c.winograd_form(0, gn=op.kind == H.OP_CONV_FWD_GN))
        elif op.kind == H.OP_CONV_WGRAD:
            n["wgrad_gn"] += int(bool(op.p[7]))
            cw = H.Conv(i[0], i[1], i[2], i[3], i[4], i[7], k=i[8], stride=i[10], pad=i[11], up=bool(i[12]), math=i[13])
            n["wgrad_v"] = n.get("wgrad_v", 0) + int(H.conv_wgrad_form(cw, with_dy_amax=bool(op.p[6]), with_gn_input=bool(op.p[7])) == 3)
        elif op.kind == H.OP_GN_BWD:
            n["gn_bwd_parts"] += int(bool(op.p[19]))
    return n


def test_f128_b32_gradients_directly_vs_oracle_and_the_forms_the_plans_launch(gd, monkeypatch):
    """the per-GPU batch of the benchmark (B = 32, 32 DISTINCT images) against the oracle directly -- this is the only batch at which the
    F128 step runs the Winograd-form kernels (conv3x3y, 16- and 8-row tiles), the data gradients that leave the GroupNorm-backward sums and the
    weight gradients that recompute their GroupNorm input -- and an assertion on WHICH forms the training plan and the B = 100 sampling plan launch, so
    that a routing change cannot silently move these tests back onto the direct kernels."""
    c, dcfg, ename, enc_sd, dec_sd, enc, dec = _rl_setup("config/ffhq_representation_learning.yml")
    x0, t, noise = _batch(32, 3, 128, seed=5)
    _guard().reset()
    st, got = _run_rl(gd, enc, dec, 32, 128, x0, t, noise)
    assert _guard().read()[0] == 0
    n = _plan_census(st.plan)
    print("[F128 B=32 training plan]", n)
    assert n["wino_fwd"] >= 60 and n["wino_dgrad"] >= 25, n                # the 128^2 / 64^2 / wide 32^2 layers (16-row tiles) + the 16^2 layers (8-row tiles)
    assert n["gnb"] >= 8 and n["gnb"] == n["gn_bwd_parts"], n            # in_layers GroupNorms of the shift branch (dropout is 0 here: out_layers too)
    assert n["wgrad_v"] >= 30, n                                             # round 6: the producer / consumer weight gradient takes every 3x3 layer above the 8 x 8 level
    assert n["wgrad_gn"] == 0, n                                             # round 6 default (PDAE_FUSE_GN_TRAIN=0): the in_layers activation is materialised, plain weight gradients
    z, eps, shift, loss, grads = _oracle_rl(enc_sd, ename, dec_sd, dcfg, x0, t, noise)
    assert rel_err(got["z"], z) < 1e-4 and rel_err(got["eps"], eps) < 1e-4 and rel_err(got["shift"], shift) < 1e-4
    assert abs(got["loss"] - loss) < 1e-4 * abs(loss), (got["loss"], loss)
    _check_grads(got["grads"], grads, GRAD_TOL, "F128 B=32 distinct images vs oracle")
    del st
    # the memory-saving form (PDAE_FUSE_GN_TRAIN=1: fused-GN forward + weight gradients that recompute the activation, -2.5 GiB at B = 32) against the same
    # oracle numbers: it stays a supported switch and this is its topology-level check
    monkeypatch.setenv("PDAE_FUSE_GN_TRAIN", "1")
    for m_ in (enc, dec):
        m_.invalidate_plans()
    st, got1 = _run_rl(gd, enc, dec, 32, 128, x0, t, noise)
    monkeypatch.delenv("PDAE_FUSE_GN_TRAIN")
    n1 = _plan_census(st.plan)
    assert n1["wgrad_gn"] >= 6, n1                                           # in_layers stages up to 128 output channels (PDAE_FUSE_GN_TRAIN_MAXCOUT)
    assert rel_err(got1["eps"], eps) < 1e-4 and rel_err(got1["shift"], shift) < 1e-4 and abs(got1["loss"] - loss) < 1e-4 * abs(loss)
    _check_grads(got1["grads"], grads, GRAD_TOL, "F128 B=32, PDAE_FUSE_GN_TRAIN=1 vs oracle")
    del st
    for m_ in (enc, dec):
        m_.invalidate_plans()
    # the evaluator's sampling batch (sampler/autoencoding_eval.py:125): the plan of one denoising step
    dec.set_eval_mode()
    ns = _plan_census(dec.plan(100, 128, 128, False))
    print("[F128 B=100 sampling plan]", ns)
    assert ns["wino_fwd"] >= 80 and ns["wino_fwd"] >= 0.7 * ns["conv3"], ns      # everything but the 29 convolutions of the 8 x 8 level (image-pair tiles)


def test_f128_ddim10_encode_sample_psnr_vs_oracle(gd):
    """ddim5 encode + ddim5 decode (10 decoder passes) at 128x128, B=1: PSNR vs the oracle trajectory > 60 dB on [-1,1] images (stated)."""
    c, dcfg, ename, enc_sd, dec_sd, enc, dec = _rl_setup("config/ffhq_representation_learning.yml")
    dec.set_eval_mode()
    x0 = _batch(1, 3, 128, seed=3)[0]
    s = O.Schedules()
    with torch.no_grad():
        z = O.encoder_forward(enc_sd, ename, x0)
        xT_ref = O.shift_ddim_encode_loop(s, "ddim5", dec_sd, dcfg, z, x0)
        rec_ref = O.shift_ddim_sample_loop(s, "ddim5", dec_sd, dcfg, z, xT_ref)
        _guard().reset()
        xT = gd.representation_learning_ddim_encode("ddim5", enc, dec, x0.to(DEV))
        rec = gd.representation_learning_ddim_sample("ddim5", None, dec, None, xT, enc(x0.to(DEV)))
    assert _guard().read()[0] == 0

    def psnr(a, b):
        return 10 * math.log10(4.0 / float(((a.double().cpu() - b.double()) ** 2).mean()))
    assert psnr(xT, xT_ref) > 60 and psnr(rec, rec_ref) > 60, (psnr(xT, xT_ref), psnr(rec, rec_ref))


def test_c64_train_step_full_topology_fp32_and_bf16(gd):
    c, dcfg, ename, enc_sd, dec_sd, enc, dec = _rl_setup("config/celeba64_representation_learning.yml")
    assert dcfg["base_channel"] == 64 and dcfg["channel_multiplier"] == [1, 2, 4, 8] and c["train_dataset_config"]["image_size"] == 64
    x0, t, noise = _batch(2, 3, 64, seed=1)
    _guard().reset()
    st, got = _run_rl(gd, enc, dec, 2, 64, x0, t, noise)
    assert _guard().read()[0] == 0
    z, eps, shift, loss, grads = _oracle_rl(enc_sd, ename, dec_sd, dcfg, x0, t, noise)
    assert rel_err(got["z"], z) < 1e-4 and rel_err(got["eps"], eps) < 1e-4 and rel_err(got["shift"], shift) < 1e-4
    assert abs(got["loss"] - loss) < 1e-4 * abs(loss)
    _check_grads(got["grads"], grads, GRAD_TOL, "C64 B=2 vs oracle")
    # BASELINE config #2 runs in bf16 (optimizer_config.enable_amp -> math "bf16": bf16 operands, fp32 accumulate).  The reference has no
    # bf16 numerics of its own: the stated tolerance is against the fp32-grade path on the same net.
    del st
    st16, got16 = _run_rl(gd, enc, dec, 2, 64, x0, t, noise, math="bf16")
    e = dict(eps=rel_err(got16["eps"], got["eps"]), shift=rel_err(got16["shift"], got["shift"]), z=rel_err(got16["z"], got["z"]),
             loss=abs(got16["loss"] - got["loss"]) / abs(got["loss"]))
    gn = max(float((got16["grads"][k].double() - v.double()).norm() / (v.double().norm() + 1e-30)) for k, v in got["grads"].items()
             if float(v.double().norm()) > 1e-6 * max(float(u.double().norm()) for u in got["grads"].values()))
    print(f"[bf16 vs fp32-grade, C64 full topology] {e} worst gradient norm error {gn:.3e}")
    assert e["eps"] < BF16_TOL["out"] and e["shift"] < BF16_TOL["out"] and e["z"] < BF16_TOL["out"] and e["loss"] < BF16_TOL["loss"], e
    assert gn < BF16_TOL["grad"], gn


def test_m32_regular_step_full_topology_vs_oracle(gd):
    from pdae_amd.model import denoise_fn as denoise_fn_module
    from pdae_amd.trainer.fused_step import FusedRegularStep
    c = _yaml("config/mnist_regular.yml")
    cfg = {k: v for k, v in c["denoise_fn_config"].items() if k not in ("model", "dims")}
    assert cfg["base_channel"] == 64 and cfg["channel_multiplier"] == [1, 2, 2, 4] and c["train_dataset_config"]["image_size"] == 32
    sd = O.synth_state_dict(O.unet_param_shapes(cfg), 5)
    net = getattr(denoise_fn_module, c["denoise_fn_config"]["model"])(device=DEV, **c["denoise_fn_config"])
    net.load_state_dict(sd)
    net.train()
    B = 16                                             # BASELINE.json configs[0]: batch 16
    x0, t, noise = _batch(B, 1, 32, seed=2)
    _guard().reset()
    st = FusedRegularStep(gd, net, None, B, 32, 32)
    st.x0.copy_(x0.to(DEV).permute(0, 2, 3, 1)); st.t.copy_(t.to(DEV)); st.noise.copy_(noise.to(DEV).permute(0, 2, 3, 1))
    st.plan.run(0, st.n_bwd)
    torch.cuda.synchronize()
    assert _guard().read()[0] == 0
    for v in sd.values():
        v.requires_grad_(True)
    s = O.Schedules()
    eps = O.unet_forward(sd, cfg, O.q_sample(s, x0, t, noise), t)
    loss = O.p_loss(noise, eps)
    loss.backward()
    assert rel_err(st.eps.permute(0, 3, 1, 2), eps.detach()) < 1e-4
    assert abs(float(st.loss.item()) - float(loss)) < 1e-4 * abs(float(loss))
    _check_grads(net.grads(), {k: v.grad for k, v in sd.items()}, GRAD_TOL, "M32 B=16 vs oracle")
    assert abs(sum(p.numel() for p in net.P.values()) - 19.4e6) < 0.1e6                                         # 19.4 M


@pytest.mark.timeout(1500)
def test_f128_ddim100_encode_then_decode_round_trip_vs_oracle_psnr_and_ssim_three_decimals(gd):
    """The evaluator's protocol at the benchmarked network, BOTH halves (the inversion was pinned on toy nets only): 100 DDIM
    encode steps x_0 -> x_T followed by 100 decode steps x_T -> reconstruction of the FFHQ-128 decoder (sampler/autoencoding_eval.py:63-78,
    diffusion/ddim.py:110-147; the shipped evaluator inverts with ddim1000 = 999 steps of the same kernel path -- that run is timed once per round,
    profiles/r06_autoencode_b100.json), B = 1, against the oracle walking the same two trajectories on the host cores (200 decoder passes: ~6 min).
    This is also the oracle check of the SAMPLING-ONLY kernel paths at real size -- GroupNorm applied inside the conv staging, statistics from
    the producing conv's epilogue, the eps-only plan of stop_percent (engine.py gn_conv / _stats_buf) -- which no training-step test touches.
    Stated bounds: PSNR of x_T and of the reconstruction vs the oracle's > 85 dB (peak 2: [-1,1] images; measured 128.5 / 98.5 dB);
    SSIM and MSE of (x_0, reconstruction) through pdae_ssim_mse equal to 3 decimals for the two reconstructions."""
    from pdae_amd.metric import ssim_mse
    c, dcfg, ename, enc_sd, dec_sd, enc, dec = _rl_setup("config/ffhq_representation_learning.yml", seed_enc=4, seed_dec=6)
    dec.set_eval_mode()
    x0 = _batch(1, 3, 128, seed=8)[0]
    s = O.Schedules()
    with torch.no_grad():
        z = O.encoder_forward(enc_sd, ename, x0)
        # the oracle's 200 decoder passes take ~1.6 s each on the GPU box's 16 host cores; a box whose cores are busy must not turn this test into
        # a quarter of an hour: the schedule is ddim100 unless ONE timed oracle pass says the whole walk would take more than ~10 minutes
        import time
        t0 = time.time()
        O.shift_unet_forward(dec_sd, dcfg, x0, torch.tensor([500]), z)
        per = time.time() - t0
        sched = os.environ.get("PDAE_TEST_DDIM", "ddim100" if per < 2.4 else ("ddim50" if per < 5.0 else "ddim20"))
        print(f"[F128 round trip] one oracle decoder pass {per:.2f} s -> schedule {sched}")
        xT_ref = O.shift_ddim_encode_loop(s, sched, dec_sd, dcfg, z, x0)
        rec_ref = O.shift_ddim_sample_loop(s, sched, dec_sd, dcfg, z, xT_ref)
        _guard().reset()
        xT = gd.representation_learning_ddim_encode(sched, enc, dec, x0.to(DEV))
        rec = gd.representation_learning_ddim_sample(sched, None, dec, None, xT, enc(x0.to(DEV)))
    assert _guard().read()[0] == 0

    def psnr(a, b):
        return 10 * math.log10(4.0 / float(((a.double().cpu() - b.double()) ** 2).mean()))
    p_enc, p_rec = psnr(xT, xT_ref), psnr(rec, rec_ref)
    sg, mg = ssim_mse(x0.to(DEV), rec, denormalize=True)
    sr, mr = ssim_mse(x0.to(DEV), rec_ref.to(DEV), denormalize=True)
    print(f"[F128 {sched} encode + {sched} decode, B=1] PSNR vs oracle: x_T {p_enc:.1f} dB, reconstruction {p_rec:.1f} dB; ssim {float(sg):.5f} / {float(sr):.5f}; "
          f"mse {float(mg):.6f} / {float(mr):.6f}")
    assert p_enc > 85 and p_rec > 85, (p_enc, p_rec)        # measured on MI355X in round 6: 128.5 / 98.5 dB
    assert abs(float(sg) - float(sr)) < 5e-4 and abs(float(mg) - float(mr)) < 5e-4, (float(sg), float(sr), float(mg), float(mr))
    assert rel_err(ssim_mse(x0.to(DEV), rec_ref.to(DEV), denormalize=True)[0], O.ssim((x0 + 1) / 2, (rec_ref + 1) / 2)) < 1e-4     # the metric kernel itself, at 128^2


@pytest.mark.timeout(1200)
def test_f128_ddim1000_inversion_checked_in_windows_along_its_own_trajectory(gd):
    """The evaluator's ACTUAL inversion (sampler/autoencoding_eval.py:74: `ddim1000` = 999 steps, diffusion/ddim.py:140-147) at the benchmarked network,
    B = 1.  Walking all 999 steps on the oracle would take half an hour of host time, and it is not needed: the loop is a chain of deterministic
    maps x_{i+1} = F_i(x_i), so the HIP path runs the WHOLE inversion once (trajectory kept on the device) and the oracle re-computes windows of K
    consecutive steps starting from the HIP state at the window's first step -- at the start, the middle and the very end of the schedule (where
    sqrt(1/ac - 1) is largest and the clamp is active).  Every window end must agree to > 85 dB PSNR; a wrong timestep map, coefficient row or
    shift term at any of these 3 K steps would show as a different map.  Together with the ddim100 round trip above this pins both halves of the
    protocol at F128."""
    c, dcfg, ename, enc_sd, dec_sd, enc, dec = _rl_setup("config/ffhq_representation_learning.yml", seed_enc=4, seed_dec=6)
    dec.set_eval_mode()
    x0 = _batch(1, 3, 128, seed=9)[0]
    s = O.Schedules()
    d = O.DDIMTables(s, "ddim1000")
    T, K = d.timesteps, 6
    assert T == 999
    traj = []
    with torch.no_grad():
        _guard().reset()
        z_dev = enc(x0.to(DEV))
        xT = gd._ddim("ddim1000").shift_ddim_encode_loop(dec, z_dev, x0.to(DEV), trajectory=traj)
        assert _guard().read()[0] == 0 and len(traj) == T and torch.equal(traj[-1], xT)
        z = O.encoder_forward(enc_sd, ename, x0)
        assert rel_err(z_dev, z) < 1e-4

        def psnr(a, b):
            return 10 * math.log10(4.0 / max(float(((a.double().cpu() - b.double()) ** 2).mean()), 1e-30))
        worst = 1e9
        for first in (0, T // 2, T - K):
            x = x0 if first == 0 else traj[first - 1].cpu().float()           # state in front of step `first` (trajectory[i] = state after step i)
            for i in range(first, first + K):
                t = torch.full((1,), i, dtype=torch.long)
                eps, g = O.shift_unet_forward(dec_sd, dcfg, x, d.timestep_map[t], z)
                x = O.ddim_update(d, x, t, eps, g, encode=True)
            p = psnr(traj[first + K - 1], x)
            print(f"[F128 ddim1000 inversion] steps {first}..{first + K - 1}: PSNR vs oracle {p:.1f} dB (|x| max {float(x.abs().max()):.2f})")
            worst = min(worst, p)
    assert worst > 85, worst
    assert torch.isfinite(xT).all() and 0.5 < float(xT.std()) < 2.0              # the inversion ends near unit-variance noise


def test_latent_ffhq_yaml_full_topology_step_vs_oracle(gd):
    """BASELINE config #5 exactly as shipped: config/ffhq_latent.yml -> MLPSkipNet 10 x 2048 with skip-concats, per-GPU batch 128, L1 loss,
    AdamW(1e-3, wd 0.01) + EMA -- through FusedLatentStep (model/mlp_skip_net.py:55-141, gaussian_diffusion.py:373-398,
    train_latent_diffusion.py:95-178).  Output, loss 1e-4; every gradient 1e-3 in norm; parameters after the optimizer step vs the oracle's
    AdamW on the oracle's gradients."""
    from pdae_amd.model.representation_learning import latent_denoise_fn as latent_module
    from pdae_amd.trainer.fused_step import FusedLatentStep
    c = _yaml("config/ffhq_latent.yml")
    mc = c["latent_denoise_fn_config"]
    cfg = {k: v for k, v in mc.items() if k != "model"}
    B = c["dataloader_config"]["train"]["batch_size"]
    assert (cfg["model_channel"], cfg["num_layers"], B, c["optimizer_config"]["name"]) == (2048, 10, 128, "AdamW")
    sd = O.synth_state_dict(O.mlp_skip_net_param_shapes(cfg), 13)
    net = getattr(latent_module, mc["model"])(device=DEV, **cfg)
    net.load_state_dict(sd, strict=False)                  # the duplicate cond_layers.1.* keys alias linear_emb.*
    net.train()
    import copy
    ema = copy.deepcopy(net)
    oc = c["optimizer_config"]
    opt = dict(lr=float(oc["lr"]), betas=eval(oc["adam_betas"]), eps=float(oc["adam_eps"]), weight_decay=float(oc["weight_decay"]))
    st = FusedLatentStep(gd, net, ema, B, decoupled=True, ema_decay=float(c["runner_config"]["ema_decay"]), **opt)
    g = torch.Generator().manual_seed(17)
    ic = cfg["input_channel"]
    z0, noise, t = torch.randn(B, ic, generator=g), torch.randn(B, ic, generator=g), torch.randint(0, 1000, (B,), generator=g)
    # oracle: q_sample on the latent schedule (constant beta 0.008), L1 loss, autograd, AdamW
    ac = np.cumprod(1.0 - np.full(1000, 0.008))
    z_t = torch.tensor(np.sqrt(ac), dtype=torch.float32)[t].view(-1, 1) * z0 + torch.tensor(np.sqrt(1 - ac), dtype=torch.float32)[t].view(-1, 1) * noise
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_out = O.mlp_skip_net_forward(ref_sd, cfg, z_t, t)
    ref_loss = (noise - ref_out).abs().mean()
    ref_loss.backward()
    p_before = {k: net.P[k].detach().clone() for k in sd if k in net.P}
    loss = st.step(z0.to(DEV), t=t.to(DEV), noise=noise.to(DEV))
    torch.cuda.synchronize()
    assert abs(float(loss.item()) - float(ref_loss)) < 1e-4 * abs(float(ref_loss)), (float(loss.item()), float(ref_loss))
    G = net.grads()
    ref_g = {k: v.grad for k, v in ref_sd.items() if k in G}
    assert len(ref_g) >= 4 * cfg["num_layers"]
    _check_grads(G, ref_g, GRAD_TOL, "latent MLPSkipNet B=128 vs oracle")
    # optimizer: AdamW step 1 on the oracle's gradients from the same parameters.  At step 1 the update is lr * sign-like(g): compare where the
    # gradient is well above rounding noise, and bound the rest by one step size
    for k, gr in ref_g.items():
        want, _, _ = O.adam_step(p_before[k].cpu(), gr, torch.zeros_like(gr), torch.zeros_like(gr), 1, opt["lr"], *opt["betas"], opt["eps"], opt["weight_decay"], True)
        diff = (net.P[k].detach().cpu() - want).abs()
        solid = gr.abs() > 1e-3 * gr.abs().max()
        assert float(diff[solid].max() if solid.any() else 0.0) < 2e-5 and float(diff.max()) < 2.1 * opt["lr"], (k, float(diff.max()))
    assert not torch.equal(ema.flat_train, net.flat_train)


def test_f128_train_step_with_dropout_on_vs_oracle_with_the_device_masks(gd):
    """The benchmark runs with the config's dropout 0.1 while every other parity test switches it off (RNG streams cannot match across
    devices).  Here the F128 step runs WITH dropout, the keep masks the device drew (Philox2x32 inside gn_apply, regenerated inside
    gn_bwd) are read back from the saved activations -- a2 = dropout(silu(.)) is zero exactly where an element was dropped -- and injected
    into the oracle (oracle.DROP_MASKS): loss 1e-4, every trainable gradient 1e-3 in norm.  Pins, at real size, the dropout variants of
    gn_apply / gn_bwd (same mask in forward and backward, 1/(1-p) scaling) that the speed number exercises."""
    c, dcfg0, ename, enc_sd, dec_sd, enc, dec = _rl_setup("config/ffhq_representation_learning.yml", seed_enc=3, seed_dec=5)
    from pdae_amd.model.representation_learning import decoder as decoder_module
    p_drop = 0.1
    dcfg = dict(dcfg0, dropout=p_drop)
    dec = getattr(decoder_module, c["decoder_config"]["model"])(device=DEV, latent_dim=c["decoder_config"]["latent_dim"], **dcfg)
    dec.load_state_dict(dec_sd)
    dec.set_train_mode()
    x0, t, noise = _batch(2, 3, 128, seed=6)
    _guard().reset()
    st, got = _run_rl(gd, enc, dec, 2, 128, x0, t, noise)
    assert _guard().read()[0] == 0 and len(st.plan.drop_ops) > 0
    masks = {}
    for blocks in [st.fx.smid_ctx] + list(st.fx.sout_ctx):
        for kind, ctx in blocks:
            if kind == "res":
                a2 = ctx.g2.y
                keep = (a2 != 0).permute(0, 3, 1, 2).float().cpu()
                assert abs(float(keep.mean()) - (1 - p_drop)) < 0.01, (ctx.pre, float(keep.mean()))
                masks[ctx.pre] = (keep, p_drop)
    assert len(masks) >= 10
    O.DROP_MASKS.clear(); O.DROP_MASKS.update(masks)
    try:
        z, eps, shift, loss, grads = _oracle_rl(enc_sd, ename, dec_sd, dcfg0, x0, t, noise)
    finally:
        O.DROP_MASKS.clear()
    assert rel_err(got["eps"], eps) < 1e-4 and rel_err(got["shift"], shift) < 1e-4
    assert abs(got["loss"] - loss) < 1e-4 * abs(loss), (got["loss"], loss)
    _check_grads(got["grads"], grads, GRAD_TOL, "F128 dropout on vs oracle")


def test_f128_eval_mode_single_pass_vs_oracle_b2_and_the_sampling_batch_of_100(gd):
    """ONE eval-mode (eps, shift) pass of the FFHQ-128 decoder (model/shift_unet.py:253-284 of the reference) against the oracle, with nothing
    contractive between the kernels and the comparison: every DDIM-loop test passes x_0-hat through clamp(-1, 1) each step (ddim.py:100,131), which
    saturates on random weights and erases upstream error (round 3's ddim100 test agreed to 126 dB for that reason).  The sampling plans run
    kernel paths no training-step test touches -- GroupNorm applied inside the conv staging (pdae_conv2d_fwd_gn), the skip conv inside the K
    loop (pdae_conv2d_fwd_skip), statistics from the producing conv's epilogue, all on conv3x3r at 128^2 / 64^2 -- and, at the evaluator's batch
    of 100 (sampler/autoencoding_eval.py:125), other split-K plans plus the row-sliced grouped Linear for more than 32 rows.
    Gates: eps / shift within 1e-4 of the oracle at B = 2; at B = 100 = 50 copies of that batch every row within 1e-5 of its B = 2 row
    (and 1e-4 of the oracle)."""
    c, dcfg, ename, enc_sd, dec_sd, enc, dec = _rl_setup("config/ffhq_representation_learning.yml", seed_enc=7, seed_dec=9)
    dec.set_eval_mode()
    x0, t, noise = _batch(2, 3, 128, seed=12)
    s = O.Schedules()
    with torch.no_grad():
        z = O.encoder_forward(enc_sd, ename, x0)
        x_t = O.q_sample(s, x0, t, noise)
        eps_ref, shift_ref = O.shift_unet_forward(dec_sd, dcfg, x_t, t, z)
        _guard().reset()
        eps2, shift2 = dec(x_t.to(DEV), t.to(DEV), z.to(DEV))
        eps2, shift2 = eps2.clone(), shift2.clone()
        e2 = (rel_err(eps2, eps_ref), rel_err(shift2, shift_ref))
        rep = lambda a: a.repeat(50, *([1] * (a.dim() - 1)))
        eps100, shift100 = dec(rep(x_t).to(DEV), rep(t).to(DEV), rep(z).to(DEV))
        torch.cuda.synchronize()
    assert _guard().read()[0] == 0, "fp16 window exceeded on N(0, 1/fan_in) weights"
    print(f"[F128 eval-mode pass] B=2 vs oracle: eps {e2[0]:.2e} shift {e2[1]:.2e}")
    assert e2[0] < 1e-4 and e2[1] < 1e-4, e2
    assert float(shift_ref.abs().max()) > 1e-3 and float(eps_ref.abs().max()) > 1e-3          # synth_state_dict randomises the zero-initialised heads
    p100 = dec.plan(100, 128, 128, False)
    assert p100.n_const > 0                                                                   # the pinned latent-only prefix exists in this plan
    worst = 0.0
    for r in range(100):
        worst = max(worst, rel_err(eps100[r:r + 1], eps2[r % 2:r % 2 + 1]), rel_err(shift100[r:r + 1], shift2[r % 2:r % 2 + 1]))
    print(f"[F128 eval-mode pass] B=100 rows vs their B=2 rows: worst {worst:.2e}")
    assert worst < 1e-5, worst
    assert rel_err(eps100[98:], eps_ref) < 1e-4 and rel_err(shift100[:2], shift_ref) < 1e-4
