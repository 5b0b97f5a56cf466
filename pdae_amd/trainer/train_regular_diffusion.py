"""Plain-DDPM trainer (config #1) with the reference's CLI, config schema and checkpoint keys
(trainer/train_regular_diffusion.py:19-211), built around FusedRegularStep.

    python -m pdae_amd.trainer.train_regular_diffusion --config_path config/mnist_regular.yml --run_path runs/mnist [--max_steps K]
"""
import argparse
import copy
import json
import os
import time

import torch

from .. import dataset as dataset_module
from ..diffusion.gaussian_diffusion import GaussianDiffusion
from ..model import denoise_fn as denoise_fn_module
from ..utils import init_distributed_mode, load_yaml, save_yaml, set_seed
from .fused_step import FusedRegularStep, export_adam_state, load_adam_state


DATA_SEED = 666666666          # one permutation stream for all ranks (utils/utils.py:30 base seed)


class _LoopMixin:
    """Logging / checkpoint cadence shared by the secondary trainers (base_trainer.py + train_*.py main loops)."""

    def _init_common(self, args):
        self.global_rank, self.global_world_size, self.local_rank = init_distributed_mode()
        self.device = torch.device("cuda", self.local_rank)
        torch.cuda.set_device(self.device)
        set_seed(0)                                      # identical initialisation on every rank (base_trainer.py:27-28)
        self.config = load_yaml(args.config_path)
        self.run_path, self.max_steps, self.step = args.run_path, args.max_steps, 0
        self.allow_random_init = bool(getattr(args, "allow_random_init", False))
        if self.global_rank == 0:
            os.makedirs(os.path.join(self.run_path, "checkpoints"), exist_ok=True)
            save_yaml(os.path.join(self.run_path, "config.yml"), self.config)

    def _opt_kwargs(self):
        oc = self.config["optimizer_config"]
        if oc.get("enable_amp", False):
            raise NotImplementedError("enable_amp is wired for the representation-learning trainer only")
        return dict(lr=float(oc["lr"]), betas=eval(oc["adam_betas"]), eps=float(oc["adam_eps"]), weight_decay=float(oc["weight_decay"]))

    def _run(self, one_step, samples_per_step):
        rc = self.config["runner_config"]
        display = int(rc["display_steps"])
        acc, acc_n = torch.zeros(1, device=self.device), 0
        t_top = time.time()
        n_it = int(rc.get("num_iterations", 1))
        while self.max_steps is None or self.step < self.max_steps:
            for _ in range(n_it):                        # micro-batches of one optimizer step (train_regular_diffusion.py:82-110)
                acc += one_step()
            self.step += 1
            acc_n += 1
            saving = self.step % int(rc["save_latest_every_steps"]) == 0 or self.step % int(rc["save_checkpoint_every_steps"]) == 0
            if self.step % display == 0 or saving:
                # fp16-window guard: discarded steps are re-counted, plan -> bf16x6; polled in front of every checkpoint as well
                rewound = self.fused.handle_saturation()
                if rewound:                               # discarded steps: their (possibly non-finite) losses leave the running mean with them
                    self.step -= rewound
                    acc.zero_()
                    acc_n = 0
            if self.step % display == 0 and acc_n > 0:
                loss = float(acc.item()) / acc_n
                if torch.distributed.is_initialized():
                    t = torch.tensor([loss], device=self.device)
                    torch.distributed.all_reduce(t)
                    loss = float(t.item()) / self.global_world_size
                if self.global_rank == 0:
                    dt = time.time() - t_top
                    rec = {"step": self.step, "prediction_loss": loss, "secs": round(dt, 2),
                           "samples_per_sec": round(display * n_it * samples_per_step * self.global_world_size / dt, 2)}
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


class RegularDiffusionTrainer(_LoopMixin):
    def __init__(self, args):
        self._init_common(args)
        c = self.config
        self.dataset = dataset_module.build(c["train_dataset_config"], device=self.device, rank=self.global_rank, world_size=self.global_world_size,
                                            seed=DATA_SEED)          # shared seed, rank-strided shares (DistributedSampler, base_trainer.py:73-78)
        self.batch_size = c["dataloader_config"]["train"]["batch_size"]
        self.gaussian_diffusion = GaussianDiffusion(c["diffusion_config"], device=self.device)
        self.denoise_fn = getattr(denoise_fn_module, c["denoise_fn_config"]["model"])(device=self.device, **c["denoise_fn_config"])
        self.ema_denoise_fn = copy.deepcopy(self.denoise_fn)
        self.ema_denoise_fn.eval(); self.ema_denoise_fn.requires_grad_(False)
        self.denoise_fn.train()
        self.denoise_fn_without_ddp = self.denoise_fn
        self.opt = self._opt_kwargs()
        size = c["train_dataset_config"]["image_size"]
        self.fused = FusedRegularStep(self.gaussian_diffusion, self.denoise_fn, self.ema_denoise_fn, self.batch_size, size, size,
                                      ema_decay=float(c["runner_config"]["ema_decay"]), ema_every=int(c["runner_config"].get("ema_every", 1)),
                                      num_iterations=int(c["runner_config"].get("num_iterations", 1)), **self.opt)
        if args.resume:
            self.load(args.resume)
        set_seed(self.global_rank)
        print(f"rank{self.global_rank}: trainer initialized.", flush=True)

    def train(self):
        gen = torch.Generator(device=self.device)
        gen.manual_seed(int(time.time()) + self.global_rank)
        self._run(lambda: self.fused.step(self.dataset.batch(self.batch_size, self.device, gen)["x_0"]), self.batch_size)

    def save(self, path):                                # train_regular_diffusion.py:180-190
        torch.save({"step": self.step, "denoise_fn": self.denoise_fn.state_dict(), "ema_denoise_fn": self.ema_denoise_fn.state_dict(),
                    "optimizer": export_adam_state(self.fused, [(self.denoise_fn, None)], **self.opt), "scaler": {}}, path)
        print(f"rank{self.global_rank}: step, model, optimizer and scaler saved to {path}(step {self.step // 1000}k).", flush=True)

    def load(self, path):                                # :192-202
        data = torch.load(path, map_location=torch.device("cpu"))
        self.step = data["step"]
        self.denoise_fn.load_state_dict(data["denoise_fn"])
        self.ema_denoise_fn.load_state_dict(data["ema_denoise_fn"])
        load_adam_state(self.fused, [(self.denoise_fn, None)], data["optimizer"])
        self.fused.step_count = self.step
        print(f"rank{self.global_rank}: step, model, optimizer and scaler restored from {path}(step {self.step // 1000}k).", flush=True)


def _parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--config_path", type=str, required=True)
    parser.add_argument("--run_path", type=str, required=True)
    parser.add_argument("--resume", type=str, default="", help="resume from checkpoint")
    parser.add_argument("--max_steps", type=int, default=None, help="stop after this many optimizer steps (the reference loops forever)")
    parser.add_argument("--allow_random_init", action="store_true", help="continue with randomly initialised frozen networks / N(0,1) latent "
                        "statistics when a pre-trained file is absent (synthetic benchmarking only; default: fail like the reference's torch.load)")
    return parser


if __name__ == "__main__":
    RegularDiffusionTrainer(_parser().parse_args()).train()
