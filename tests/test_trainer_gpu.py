"""GPU: config-driven trainer (CLI schema / run dir / checkpoint layout of the reference) and the autoencoding sampler,
end to end on a small synthetic config."""
import json
import os
from types import SimpleNamespace

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_cfg(tmp_path):
    dp = {"denoise_fn_config": dict(model="X", dims=2, input_channel=3, base_channel=32, channel_multiplier=[1, 2, 2],
                                    num_residual_blocks_of_a_block=1, dropout=0.1, attention_resolutions=[4], use_new_attention_order=False,
                                    num_heads=1, head_channel=-1)}
    (tmp_path / "dpm.yml").write_text(yaml.dump(dp))
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "celeba64_representation_learning.yml")))
    cfg["trained_ddpm_config"] = str(tmp_path / "dpm.yml")
    cfg["trained_ddpm_checkpoint"] = str(tmp_path / "none.pt")
    cfg["dataloader_config"]["train"]["batch_size"] = 4
    cfg["runner_config"].update(display_steps=2, save_latest_every_steps=4, evaluate_every_steps=1000)
    (tmp_path / "cfg.yml").write_text(yaml.dump(cfg))
    return str(tmp_path / "cfg.yml")


def test_trainer_runs_saves_and_resumes(tmp_path):
    from pdae_amd.trainer.train_representation_learning import RepresentationLearningTrainer
    cfg_path = _write_cfg(tmp_path)
    run = str(tmp_path / "run")
    tr = RepresentationLearningTrainer(SimpleNamespace(config_path=cfg_path, run_path=run, resume="", allow_random_init=True, max_steps=4))
    frozen = tr.decoder.flat_frozen.clone()
    w0 = tr.decoder.flat_train.clone()
    tr.train()
    assert tr.step == 4 and not torch.equal(w0, tr.decoder.flat_train) and torch.equal(frozen, tr.decoder.flat_frozen)
    logs = [json.loads(l) for l in open(os.path.join(run, "log.jsonl"))]
    assert len(logs) == 2 and all(0 < l["prediction_loss"] < 10 for l in logs)
    ck = torch.load(os.path.join(run, "checkpoints", "latest.pt"), map_location="cpu")
    assert set(ck) == {"step", "encoder", "ema_encoder", "decoder", "ema_decoder", "optimizer", "scaler"} and ck["step"] == 4
    assert len(ck["optimizer"]["param_groups"]) == 5
    n_params = sum(len(g["params"]) for g in ck["optimizer"]["param_groups"])
    assert n_params == len(ck["optimizer"]["state"]) == sum(1 for p in tr.decoder.parameters() if p.requires_grad) + len(list(tr.encoder.parameters()))
    # the exported optimizer state is what torch.optim.Adam itself would accept
    opt = torch.optim.Adam([{"params": list(tr.encoder.parameters())}, {"params": list(tr.decoder.label_emb.parameters())},
                            {"params": list(tr.decoder.shift_middle_block.parameters())}, {"params": list(tr.decoder.shift_output_blocks.parameters())},
                            {"params": list(tr.decoder.shift_out.parameters())}], lr=1e-4)
    opt.load_state_dict(ck["optimizer"])
    # resume continues from the same weights / moments
    tr2 = RepresentationLearningTrainer(SimpleNamespace(config_path=cfg_path, run_path=run, resume=os.path.join(run, "checkpoints", "latest.pt"), allow_random_init=True, max_steps=5))
    assert tr2.step == 4 and torch.equal(tr2.decoder.flat_train, tr.decoder.flat_train) and torch.equal(tr2.fused.m[0], tr.fused.m[0])
    assert torch.equal(tr2.ema_encoder.flat_train, tr.ema_encoder.flat_train)
    tr2.train()
    assert tr2.step == 5


def test_autoencoding_sampler_small(tmp_path):
    from pdae_amd.sampler.autoencoding_eval import Sampler
    from pdae_amd.model.shift_unet import ShiftUNet
    from pdae_amd.model.representation_learning.encoder import CELEBA64Encoder
    enc = CELEBA64Encoder(device="cuda", latent_dim=64)
    dec = ShiftUNet(device="cuda", latent_dim=64, input_channel=3, base_channel=32, channel_multiplier=[1, 2], num_residual_blocks_of_a_block=1,
                    attention_resolutions=[2], num_heads=1, head_channel=-1, use_new_attention_order=False, dropout=0.0)
    cfg = {"diffusion_config": {"timesteps": 1000, "betas_type": "linear"}, "batch_size": 3,
           "dataset_config": {"dataset_name": "SYNTHETIC", "image_channel": 3, "image_size": 64, "length": 5}}
    out = Sampler(cfg, encoder=enc, decoder=dec).start(encoder_style="ddim20", decoder_style="ddim10")
    assert out["n"] == 5 and 0.0 < out["ssim"] <= 1.0 and out["mse"] >= 0.0


def test_regular_diffusion_trainer_runs_and_resumes(tmp_path):
    """config #1 (mnist_regular.yml) through trainer/train_regular_diffusion.py: reference checkpoint keys, resume."""
    from pdae_amd.trainer.train_regular_diffusion import RegularDiffusionTrainer
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "mnist_regular.yml")))
    cfg["denoise_fn_config"].update(base_channel=32, channel_multiplier=[1, 2])
    cfg["dataloader_config"]["train"]["batch_size"] = 4
    cfg["runner_config"].update(display_steps=2, save_latest_every_steps=3)
    (tmp_path / "cfg.yml").write_text(yaml.dump(cfg))
    run = str(tmp_path / "run")
    tr = RegularDiffusionTrainer(SimpleNamespace(config_path=str(tmp_path / "cfg.yml"), run_path=run, resume="", allow_random_init=True, max_steps=3))
    w0 = tr.denoise_fn.flat_train.clone()
    tr.train()
    assert tr.step == 3 and not torch.equal(w0, tr.denoise_fn.flat_train)
    ck = torch.load(os.path.join(run, "checkpoints", "latest.pt"), map_location="cpu")
    assert set(ck) == {"step", "denoise_fn", "ema_denoise_fn", "optimizer", "scaler"}
    tr2 = RegularDiffusionTrainer(SimpleNamespace(config_path=str(tmp_path / "cfg.yml"), run_path=run,
                                                  resume=os.path.join(run, "checkpoints", "latest.pt"), max_steps=4))
    assert tr2.step == 3 and torch.equal(tr2.denoise_fn.flat_train, tr.denoise_fn.flat_train) and torch.equal(tr2.fused.v[0], tr.fused.v[0])
    tr2.train()
    assert tr2.step == 4


def test_latent_diffusion_trainer_runs_and_matches_adamw(tmp_path):
    """config #5 (ffhq_latent.yml) through trainer/train_latent_diffusion.py; the fused AdamW + EMA step against torch.optim.AdamW."""
    from pdae_amd.trainer.train_latent_diffusion import LatentDiffusionTrainer
    cfg_path = _write_cfg(tmp_path)                                      # small autoencoder config (64x64, latent 512)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ffhq_latent.yml")))
    cfg["train_dataset_config"]["image_size"] = 64
    cfg["trained_ddpm_config"] = str(tmp_path / "dpm.yml")
    cfg["trained_representation_learning_config"] = cfg_path
    cfg["trained_representation_learning_checkpoint"] = str(tmp_path / "none.pt")
    cfg["inferred_latents"] = str(tmp_path / "none_latents.pt")
    cfg["latent_denoise_fn_config"].update(model_channel=256, num_layers=4)
    cfg["dataloader_config"]["train"]["batch_size"] = 8
    cfg["runner_config"].update(display_steps=1, save_latest_every_steps=2)
    (tmp_path / "lat.yml").write_text(yaml.dump(cfg))
    run = str(tmp_path / "run_latent")
    tr = LatentDiffusionTrainer(SimpleNamespace(config_path=str(tmp_path / "lat.yml"), run_path=run, resume="", allow_random_init=True, max_steps=2))
    net = tr.latent_denoise_fn
    # reference optimiser on a torch copy of the same parameters, fed the gradients of the fused step
    ref_p = [p.detach().clone().requires_grad_(True) for p in net.parameters()]
    ref_opt = torch.optim.AdamW(ref_p, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    ema0 = tr.ema_latent_denoise_fn.flat_train.clone()
    z0 = torch.randn(8, 512, device="cuda")
    tr.fused.step(z0)
    for rp, p in zip(ref_p, net.parameters()):
        rp.grad = p.grad.detach().clone()
    ref_opt.step()
    for rp, p in zip(ref_p, net.parameters()):
        assert float((rp - p).abs().max()) < 1e-6 + 1e-5 * float(rp.abs().max())
    assert not torch.equal(ema0, tr.ema_latent_denoise_fn.flat_train)
    tr.train()
    assert tr.step == 2
    ck = torch.load(os.path.join(run, "checkpoints", "latest.pt"), map_location="cpu")
    assert set(ck) == {"step", "encoder", "decoder", "latent_denoise_fn", "ema_latent_denoise_fn", "optimizer", "scaler"}
    assert "layers.0.cond_layers.1.weight" in ck["latent_denoise_fn"]          # reference duplicate key present
    logs = [json.loads(l) for l in open(os.path.join(run, "log.jsonl"))]
    assert len(logs) == 2 and all(0 < l["prediction_loss"] < 10 for l in logs)


def test_trainer_enable_amp_runs_in_bf16_operand_mode(tmp_path):
    """optimizer_config.enable_amp (train_representation_learning.py:48-49,94) selects the bf16-operand / fp32-accumulate MFMA mode."""
    from pdae_amd import hip as H
    from pdae_amd.trainer.train_representation_learning import RepresentationLearningTrainer
    cfg_path = _write_cfg(tmp_path)
    cfg = yaml.safe_load(open(cfg_path))
    cfg["optimizer_config"]["enable_amp"] = True
    open(cfg_path, "w").write(yaml.dump(cfg))
    tr = RepresentationLearningTrainer(SimpleNamespace(config_path=cfg_path, run_path=str(tmp_path / "run_amp"), resume="", allow_random_init=True, max_steps=2))
    convs = [op for op in tr.fused.plan.arr if op.kind == H.OP_CONV_FWD]
    assert convs and all(op.i[13] == H.MATH_NAMES["bf16"] for op in convs)
    tr.train()
    logs = [json.loads(l) for l in open(os.path.join(str(tmp_path / "run_amp"), "log.jsonl"))]
    assert len(logs) == 1 and 0 < logs[0]["prediction_loss"] < 10


def test_load_trained_ddpm_imports_the_frozen_half_and_refreshes_prepared_weights(tmp_path):
    """train_representation_learning.py:241-244: a regular-trainer checkpoint ({'ema_denoise_fn': UNet state dict}) is loaded with strict=False
    into decoder and ema_decoder -- the frozen half (time_embed / input / middle / output blocks / out) takes the pre-trained weights, the
    shift_* half and label_emb keep their initialisation, and plans built BEFORE the load must see the new frozen weights (their prepared
    fp16-split copies are refreshed): the eps output then equals the plain UNet's on the same weights."""
    from pdae_amd.model.unet import UNet
    from pdae_amd.trainer.train_regular_diffusion import RegularDiffusionTrainer
    from pdae_amd.trainer.train_representation_learning import RepresentationLearningTrainer
    cfg_path = _write_cfg(tmp_path)
    dpm = yaml.safe_load(open(tmp_path / "dpm.yml"))["denoise_fn_config"]
    # a checkpoint written by the regular-diffusion trainer itself (reference keys: train_regular_diffusion.py:180-190)
    rcfg = yaml.safe_load(open(os.path.join(ROOT, "config", "mnist_regular.yml")))
    rcfg["denoise_fn_config"] = dict(dpm, model="UNet", dropout=0.0)
    rcfg["train_dataset_config"].update(image_size=64, image_channel=3)
    rcfg["dataloader_config"]["train"]["batch_size"] = 2
    rcfg["runner_config"].update(display_steps=1, save_latest_every_steps=2)
    (tmp_path / "reg.yml").write_text(yaml.dump(rcfg))
    reg = RegularDiffusionTrainer(SimpleNamespace(config_path=str(tmp_path / "reg.yml"), run_path=str(tmp_path / "run_reg"), resume="", max_steps=2))
    reg.train()
    ck_path = os.path.join(str(tmp_path / "run_reg"), "checkpoints", "latest.pt")
    ema_sd = torch.load(ck_path, map_location="cpu")["ema_denoise_fn"]
    # RL trainer without the checkpoint (random trunk), a plan already built and run on it
    tr = RepresentationLearningTrainer(SimpleNamespace(config_path=cfg_path, run_path=str(tmp_path / "run_a"), resume="", allow_random_init=True, max_steps=1))
    train_before, frozen_before = tr.decoder.flat_train.clone(), tr.decoder.flat_frozen.clone()
    g = torch.Generator().manual_seed(0)
    x, t, z = torch.randn(2, 3, 64, 64, generator=g).cuda(), torch.tensor([10, 700]).cuda(), torch.randn(2, 512, generator=g).cuda()
    tr.decoder.set_eval_mode()
    with torch.no_grad():
        eps_random, _ = tr.decoder(x, t, z)                      # builds the sampling plan + its prepared frozen weights
    # import: the reference call, on the live trainer
    tr.load_trained_ddpm(ck_path)
    assert torch.equal(tr.decoder.flat_train, train_before), "shift_* / label_emb must be untouched by the strict=False load"
    assert not torch.equal(tr.decoder.flat_frozen, frozen_before)
    sd = tr.decoder.state_dict()
    for k, v in ema_sd.items():
        assert torch.equal(sd[k].cpu(), v), k                    # every UNet key landed in the frozen half
    assert torch.equal(tr.ema_decoder.flat_frozen, tr.decoder.flat_frozen)
    unet = UNet(device="cuda", **{k: v for k, v in dpm.items() if k not in ("model",)})
    unet.load_state_dict(ema_sd)
    unet.eval()
    with torch.no_grad():
        eps_unet = unet(x, t)
        eps_loaded, _ = tr.decoder(x, t, z)                      # SAME plan object as above: prepared weights must have been refreshed
    from tests.conftest import rel_err
    assert rel_err(eps_loaded, eps_unet) < 1e-5 and rel_err(eps_random, eps_unet) > 1e-2
    # and the training plan of the fused step (built in the constructor, before the load) trains against the imported trunk
    tr.decoder.set_train_mode()
    tr.train()
    assert tr.step == 1 and torch.equal(tr.decoder.flat_frozen, tr.ema_decoder.flat_frozen)
    # a constructor-time import (checkpoint present at the configured path) gives the same frozen half
    cfg = yaml.safe_load(open(cfg_path))
    cfg["trained_ddpm_checkpoint"] = ck_path
    (tmp_path / "cfg_b.yml").write_text(yaml.dump(cfg))
    tr_b = RepresentationLearningTrainer(SimpleNamespace(config_path=str(tmp_path / "cfg_b.yml"), run_path=str(tmp_path / "run_b"), resume="", max_steps=1))
    assert torch.equal(tr_b.decoder.flat_frozen, tr.decoder.flat_frozen) and torch.equal(tr_b.decoder.flat_train, train_before)
