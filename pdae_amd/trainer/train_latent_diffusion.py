"""Latent-DPM trainer (config #5) with the reference's CLI, config schema and checkpoint keys
(trainer/train_latent_diffusion.py:20-271), built around FusedLatentStep.

    torchrun --nproc_per_node N -m pdae_amd.trainer.train_latent_diffusion --config_path config/ffhq_latent.yml --run_path runs/latent
"""
import copy
import os
import time

import torch

from .. import dataset as dataset_module
from ..diffusion.gaussian_diffusion import GaussianDiffusion
from ..model.representation_learning import decoder as decoder_module
from ..model.representation_learning import encoder as encoder_module
from ..model.representation_learning import latent_denoise_fn as latent_denoise_fn_module
from ..utils import load_yaml, set_seed
from .fused_step import FusedLatentStep, export_adam_state, load_adam_state
from .train_regular_diffusion import DATA_SEED, _LoopMixin, _parser


class LatentDiffusionTrainer(_LoopMixin):
    def __init__(self, args):
        self._init_common(args)
        c = self.config
        rl = load_yaml(c["trained_representation_learning_config"])
        self.dataset = dataset_module.build(c["train_dataset_config"], device=self.device, rank=self.global_rank, world_size=self.global_world_size,
                                            seed=DATA_SEED)
        self.batch_size = c["dataloader_config"]["train"]["batch_size"]
        self.gaussian_diffusion = GaussianDiffusion(rl["diffusion_config"], device=self.device)
        self.latent_denoise_fn = getattr(latent_denoise_fn_module, c["latent_denoise_fn_config"]["model"])(device=self.device,
                                                                                                         **c["latent_denoise_fn_config"])
        self.ema_latent_denoise_fn = copy.deepcopy(self.latent_denoise_fn)
        self.ema_latent_denoise_fn.eval(); self.ema_latent_denoise_fn.requires_grad_(False)
        self.latent_denoise_fn.train()
        self.latent_denoise_fn_without_ddp = self.latent_denoise_fn
        # frozen autoencoder (train_latent_diffusion.py:38-55)
        self.encoder = getattr(encoder_module, rl["encoder_config"]["model"])(device=self.device, **rl["encoder_config"])
        ddpm = load_yaml(c["trained_ddpm_config"])
        self.decoder = getattr(decoder_module, rl["decoder_config"]["model"])(device=self.device, latent_dim=rl["decoder_config"]["latent_dim"],
                                                                              **ddpm["denoise_fn_config"])
        ck = c.get("trained_representation_learning_checkpoint")
        if ck and os.path.exists(ck):
            data = torch.load(ck, map_location=torch.device("cpu"))
            self.encoder.load_state_dict(data["ema_encoder"])
            self.decoder.load_state_dict(data["ema_decoder"])
        elif not self.allow_random_init:                  # the reference fails in torch.load (train_latent_diffusion.py:44)
            raise FileNotFoundError(f"autoencoder checkpoint {ck!r} not found (pass --allow_random_init for synthetic benchmarking runs)")
        elif self.global_rank == 0:
            print(f"rank0: autoencoder checkpoint {ck!r} not found -- --allow_random_init: encoder / decoder keep their random initialisation", flush=True)
        for m in (self.encoder, self.decoder):
            m.eval(); m.requires_grad_(False)
        # latent statistics (:57-61); N(0,1) stand-in when the inferred-latents file is absent
        lat = c.get("inferred_latents")
        d = self.latent_denoise_fn.input_channel
        if lat and os.path.exists(lat):
            st = torch.load(lat, map_location=torch.device("cpu"))
            self.latents_mean, self.latents_std = st["mean"].to(self.device), st["std"].to(self.device)
        elif not self.allow_random_init:
            raise FileNotFoundError(f"inferred latents {lat!r} not found (pass --allow_random_init to train on N(0,1) stand-in statistics)")
        else:
            self.latents_mean, self.latents_std = torch.zeros(d, device=self.device), torch.ones(d, device=self.device)
        oc = c["optimizer_config"]
        name = oc.get("name", "Adam")
        if name not in ("Adam", "AdamW"):
            raise NotImplementedError(name)                                      # train_latent_diffusion.py:93
        self.opt = self._opt_kwargs()
        self.fused = FusedLatentStep(self.gaussian_diffusion, self.latent_denoise_fn, self.ema_latent_denoise_fn, self.batch_size,
                                     decoupled=(name == "AdamW"), ema_decay=float(c["runner_config"]["ema_decay"]),
                                     ema_every=int(c["runner_config"].get("ema_every", 1)), num_iterations=int(c["runner_config"].get("num_iterations", 1)),
                                     **self.opt)
        if args.resume:
            self.load(args.resume)
        set_seed(self.global_rank)
        print(f"rank{self.global_rank}: trainer initialized.", flush=True)

    def train(self):
        gen = torch.Generator(device=self.device)
        gen.manual_seed(int(time.time()) + self.global_rank)
        gd = self.gaussian_diffusion

        def one_step():
            x_0 = self.dataset.batch(self.batch_size, self.device, gen)["x_0"]
            with torch.no_grad():
                z_0 = self.encoder(x_0)
            return self.fused.step(gd.normalize(z_0, self.latents_mean, self.latents_std))

        self._run(one_step, self.batch_size)

    def save(self, path):                                # train_latent_diffusion.py:225-237
        torch.save({"step": self.step, "encoder": self.encoder.state_dict(), "decoder": self.decoder.state_dict(),
                    "latent_denoise_fn": self.latent_denoise_fn.state_dict(), "ema_latent_denoise_fn": self.ema_latent_denoise_fn.state_dict(),
                    "optimizer": export_adam_state(self.fused, [(self.latent_denoise_fn, None)], **self.opt), "scaler": {}}, path)
        print(f"rank{self.global_rank}: step, model, optimizer and scaler saved to {path}(step {self.step // 1000}k).", flush=True)

    def load(self, path):                                # :239-251
        data = torch.load(path, map_location=torch.device("cpu"))
        self.step = data["step"]
        self.encoder.load_state_dict(data["encoder"])
        self.decoder.load_state_dict(data["decoder"])
        self.latent_denoise_fn.load_state_dict(data["latent_denoise_fn"])
        self.ema_latent_denoise_fn.load_state_dict(data["ema_latent_denoise_fn"])
        load_adam_state(self.fused, [(self.latent_denoise_fn, None)], data["optimizer"])
        self.fused.step_count = self.step
        print(f"rank{self.global_rank}: step, model, optimizer and scaler restored from {path}(step {self.step // 1000}k).", flush=True)


if __name__ == "__main__":
    LatentDiffusionTrainer(_parser().parse_args()).train()
