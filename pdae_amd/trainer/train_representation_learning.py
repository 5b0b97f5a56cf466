"""Representation-learning trainer with the reference's CLI, config schema, run directory and checkpoint layout
(trainer/train_representation_learning.py + trainer/base_trainer.py), built around FusedRLStep.

    torchrun --nproc_per_node N -m pdae_amd.trainer.train_representation_learning --config_path config/ffhq_representation_learning.yml \
             --run_path runs/ffhq [--resume runs/ffhq/checkpoints/latest.pt] [--max_steps K]
"""
import argparse
import copy
import json
import os
import time

import torch

from .. import dataset as dataset_module
from .. import hip as H
from ..diffusion.gaussian_diffusion import GaussianDiffusion
from ..model.representation_learning import decoder as decoder_module
from ..model.representation_learning import encoder as encoder_module
from ..utils import init_distributed_mode, load_yaml, save_yaml, set_seed
from .fused_step import FusedRLStep, export_adam_state, load_adam_state

DATA_SEED = 666666666          # one permutation stream for all ranks (utils/utils.py:30 base seed)


class RepresentationLearningTrainer:
    def __init__(self, args):
        self.global_rank, self.global_world_size, self.local_rank = init_distributed_mode()
        self.device = torch.device("cuda", self.local_rank)
        torch.cuda.set_device(self.device)
        set_seed(0)                                      # identical parameter initialisation on every rank (base_trainer.py:27-28)
        self.config = load_yaml(args.config_path)
        self.run_path = args.run_path
        self.max_steps = args.max_steps
        self.allow_random_init = bool(getattr(args, "allow_random_init", False))
        self.step = 0
        if self.global_rank == 0:
            os.makedirs(os.path.join(self.run_path, "checkpoints"), exist_ok=True)
            os.makedirs(os.path.join(self.run_path, "samples"), exist_ok=True)
            save_yaml(os.path.join(self.run_path, "config.yml"), self.config)
        self._build_dataloader()
        self._build_model()
        self._build_optimizer()
        if args.resume:
            self.load(args.resume)
        set_seed(self.global_rank)                       # per-rank noise streams afterwards (base_trainer.py:50-52)
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        print(f"rank{self.global_rank}: trainer initialized.", flush=True)

    def _build_dataloader(self):
        cfg = self.config["train_dataset_config"]
        # every rank walks its own share of one common per-epoch permutation (DistributedSampler, base_trainer.py:73-78): the seed is shared,
        # the rank selects the share.  Evaluation reads from its own loader (train config overridden by eval_dataset_config, :62-64).
        share = dict(device=self.device, rank=self.global_rank, world_size=self.global_world_size, seed=DATA_SEED)
        self.dataset = dataset_module.build(cfg, **share)
        ecfg = copy.deepcopy(cfg)
        ecfg.update(self.config.get("eval_dataset_config") or {})
        self.eval_dataset = dataset_module.build(ecfg, **share)
        self.batch_size = self.config["dataloader_config"]["train"]["batch_size"]     # per process (base_trainer.py:71)

    def _build_model(self):
        c = self.config
        self.gaussian_diffusion = GaussianDiffusion(c["diffusion_config"], device=self.device)
        self.encoder = getattr(encoder_module, c["encoder_config"]["model"])(device=self.device, **c["encoder_config"])
        self.ema_encoder = copy.deepcopy(self.encoder)
        ddpm_cfg = load_yaml(c["trained_ddpm_config"])
        self.decoder = getattr(decoder_module, c["decoder_config"]["model"])(device=self.device, latent_dim=c["decoder_config"]["latent_dim"],
                                                                            **ddpm_cfg["denoise_fn_config"])
        self.ema_decoder = copy.deepcopy(self.decoder)
        ck = c.get("trained_ddpm_checkpoint")
        if ck and os.path.exists(ck):
            self.load_trained_ddpm(ck)
        elif not self.allow_random_init:
            # the reference fails in torch.load (train_representation_learning.py:241): training the shift branch against a random trunk
            # produces checkpoints that look valid and mean nothing
            raise FileNotFoundError(f"pre-trained DPM checkpoint {ck!r} not found (pass --allow_random_init for synthetic benchmarking runs)")
        elif self.global_rank == 0:
            print(f"rank0: pre-trained DPM checkpoint {ck!r} not found -- --allow_random_init: the frozen half keeps its random initialisation", flush=True)
        for m in (self.ema_encoder, self.ema_decoder):
            m.eval()
            m.requires_grad_(False)
        self.encoder.train()
        self.decoder.set_train_mode()
        # aliases used by reference-style code
        self.encoder_without_ddp, self.decoder_without_ddp = self.encoder, self.decoder

    def _groups(self):
        d = self.decoder
        return [(self.encoder, None), (d, "label_emb."), (d, "shift_middle_block."), (d, "shift_output_blocks."), (d, "shift_out.")]

    def _build_optimizer(self):
        oc, rc = self.config["optimizer_config"], self.config["runner_config"]
        size = self.config["train_dataset_config"]["image_size"]
        self.opt = dict(lr=float(oc["lr"]), betas=eval(oc["adam_betas"]), eps=float(oc["adam_eps"]), weight_decay=float(oc["weight_decay"]))
        # enable_amp (torch.cuda.amp autocast + GradScaler in the reference, train_representation_learning.py:48-49,94) maps to the
        # bf16-operand / fp32-accumulate MFMA mode of the convolutions: bf16 keeps the fp32 exponent range, so no loss scaling is
        # needed and the "scaler" checkpoint entry stays empty.  Default (False): fp32-grade split-bf16x6 arithmetic.
        math = H.MATH_NAMES["bf16"] if oc.get("enable_amp", False) else None
        self.fused = FusedRLStep(self.gaussian_diffusion, self.encoder, self.decoder, self.ema_encoder, self.ema_decoder, self.batch_size, size, size,
                                 ema_decay=float(rc["ema_decay"]), ema_every=int(rc["ema_every"]), num_iterations=int(rc["num_iterations"]), math=math, **self.opt)

    # ------------------------------------------------------------------ loop (train_representation_learning.py:72-156)
    def train(self):
        rc = self.config["runner_config"]
        display, n_it = int(rc["display_steps"]), int(rc["num_iterations"])
        acc, acc_n = torch.zeros(1, device=self.device), 0
        t_top = time.time()
        gen = torch.Generator(device=self.device)
        gen.manual_seed(int(time.time()) + self.global_rank)
        while self.max_steps is None or self.step < self.max_steps:
            for _ in range(n_it):
                batch = self.dataset.batch(self.batch_size, self.device, gen)
                acc += self.fused.step(batch["x_0"])                    # device-side accumulation: no per-step host sync
            self.step += 1
            acc_n += 1
            rc_save = (self.step % int(rc["save_latest_every_steps"]) == 0 or self.step % int(rc["save_checkpoint_every_steps"]) == 0
                       or self.step % int(rc["evaluate_every_steps"]) == 0)
            if self.step % display == 0 or rc_save:
                # fp16-window guard: discarded steps are re-counted, plan -> bf16x6.  Polled at the logging cadence AND in front of every
                # checkpoint / evaluation, so that no file records a step count or optimizer state that includes a discarded update
                rewound = self.fused.handle_saturation()
                if rewound:
                    # the discarded steps' losses (possibly inf / nan) are not reported, and the counter was rewound: the running mean restarts
                    self.step -= rewound
                    acc.zero_()
                    acc_n = 0
            if self.step % display == 0 and acc_n > 0:
                loss = float(acc.item()) / acc_n
                if torch.distributed.is_initialized():
                    t = torch.tensor([loss], device=self.device)
                    torch.distributed.all_reduce(t)
                    loss = float(t.item()) / self.global_world_size
                dt = time.time() - t_top
                if self.global_rank == 0:
                    rec = {"step": self.step, "prediction_loss": loss, "lr": self.opt["lr"], "secs": round(dt, 2),
                           "images_per_sec": round(display * n_it * self.batch_size * self.global_world_size / dt, 2)}
                    print(json.dumps(rec), flush=True)
                    with open(os.path.join(self.run_path, "log.jsonl"), "a") as f:
                        f.write(json.dumps(rec) + "\n")
                acc.zero_()
                acc_n = 0
                t_top = time.time()
            if self.global_rank == 0 and self.step % int(rc["save_latest_every_steps"]) == 0:
                self.save(os.path.join(self.run_path, "checkpoints", "latest.pt"))
            if self.global_rank == 0 and self.step % int(rc["save_checkpoint_every_steps"]) == 0:
                self.save(os.path.join(self.run_path, "checkpoints", f"save-{self.step // 1000}k.pt"))
            if self.step % int(rc["evaluate_every_steps"]) == 0:
                self.eval()

    def eval(self):
        """DDIM-100 samples from the EMA networks (train_representation_learning.py:158-190), saved as a tensor file."""
        n = min(int(self.config["dataloader_config"]["eval"]["num_generations"]), self.batch_size)
        with torch.no_grad():
            batch = self.eval_dataset.batch(n, self.device)
            images = self.gaussian_diffusion.representation_learning_ddim_sample("ddim100", self.ema_encoder, self.ema_decoder, batch["x_0"],
                                                                                 torch.randn_like(batch["x_0"]))
            images = images.mul(0.5).add(0.5).mul(255).add(0.5).clamp(0, 255).permute(0, 2, 3, 1).to("cpu", torch.uint8)
        if self.global_rank == 0:
            torch.save({"images": images, "gts": batch["gts"].cpu()}, os.path.join(self.run_path, "samples", f"sample{self.step // 1000}k.pt"))

    # ------------------------------------------------------------------ checkpoints (:214-244)
    def save(self, path):
        data = {"step": self.step, "encoder": self.encoder.state_dict(), "ema_encoder": self.ema_encoder.state_dict(),
                "decoder": self.decoder.state_dict(), "ema_decoder": self.ema_decoder.state_dict(),
                "optimizer": export_adam_state(self.fused, self._groups(), **self.opt), "scaler": {}}
        torch.save(data, path)
        print(f"rank{self.global_rank}: step, model, optimizer and scaler saved to {path}(step {self.step // 1000}k).", flush=True)

    def load(self, path):
        data = torch.load(path, map_location=torch.device("cpu"))
        self.step = data["step"]
        self.encoder.load_state_dict(data["encoder"])
        self.ema_encoder.load_state_dict(data["ema_encoder"])
        self.decoder.load_state_dict(data["decoder"])
        self.ema_decoder.load_state_dict(data["ema_decoder"])
        load_adam_state(self.fused, self._groups(), data["optimizer"])
        self.fused.step_count = self.step
        print(f"rank{self.global_rank}: step, model, optimizer and scaler restored from {path}(step {self.step // 1000}k).", flush=True)

    def load_trained_ddpm(self, path):
        data = torch.load(path, map_location=torch.device("cpu"))
        self.decoder.load_state_dict(data["ema_denoise_fn"], strict=False)
        self.ema_decoder.load_state_dict(data["ema_denoise_fn"], strict=False)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config_path", type=str, required=True)
    parser.add_argument("--run_path", type=str, required=True)
    parser.add_argument("--resume", type=str, default="", help="resume from checkpoint")
    parser.add_argument("--max_steps", type=int, default=None, help="stop after this many optimizer steps (the reference loops forever)")
    parser.add_argument("--allow_random_init", action="store_true", help="train against a randomly initialised frozen trunk when the pre-trained "
                        "DPM checkpoint is absent (synthetic benchmarking only; default: fail like the reference's torch.load)")
    runner = RepresentationLearningTrainer(parser.parse_args())
    runner.train()
