"""Autoencoding evaluation (sampler/autoencoding_eval.py:72-99): encode with ddim1000, decode with ddim100, per-image SSIM
and MSE on (x+1)/2, averaged over all ranks.  Each rank handles its own shard of the dataset (replicas only)."""
import copy

import torch

from .. import dataset as dataset_module
from ..diffusion.gaussian_diffusion import GaussianDiffusion
from ..metric import MSEMetric, SSIMMetric, ssim_mse
from ..model.representation_learning import decoder as decoder_module
from ..model.representation_learning import encoder as encoder_module
from ..utils import dispatch_num_samples_for_process, init_distributed_mode, load_yaml, set_seed


class Sampler:
    def __init__(self, config, encoder=None, decoder=None):
        self.config = config
        self.global_rank, self.global_world_size, self.local_rank = init_distributed_mode()
        self.device = torch.device("cuda", self.local_rank)
        torch.cuda.set_device(self.device)
        # in order, to the last image, each rank its own stride of the dataset (sampler/autoencoding_eval.py:26-43: DistributedSampler
        # shuffle=False, drop_last=False; DataLoader drop_last=False)
        self.dataset = dataset_module.build(config["dataset_config"], device=self.device, rank=self.global_rank, world_size=self.global_world_size,
                                            shuffle=False, drop_last=False)
        self.gaussian_diffusion = GaussianDiffusion(config["diffusion_config"], device=self.device)
        if encoder is None:
            mc = load_yaml(config["config_path"])
            dc = load_yaml(config["trained_ddpm_config_path"])
            encoder = getattr(encoder_module, mc["encoder_config"]["model"])(device=self.device, **mc["encoder_config"])
            decoder = getattr(decoder_module, mc["decoder_config"]["model"])(device=self.device, latent_dim=mc["decoder_config"]["latent_dim"],
                                                                            **dc["denoise_fn_config"])
            ck = torch.load(config["checkpoint_path"], map_location=torch.device("cpu"))
            encoder.load_state_dict(ck["ema_encoder"])
            decoder.load_state_dict(ck["ema_decoder"])
        self.encoder, self.decoder = encoder.eval(), decoder.eval()
        self.ssim_metric, self.mse_metric = SSIMMetric(), MSEMetric()
        set_seed(self.global_rank)

    def start(self, num_images=None, encoder_style="ddim1000", decoder_style="ddim100"):
        total = len(self.dataset) if num_images is None else num_images
        order = getattr(self.dataset, "order", None)
        if order is not None and num_images is None:
            mine = order.per_rank                  # the sampler's padded share: every rank the same count, like the reference
        else:
            mine = dispatch_num_samples_for_process(total, self.global_world_size, self.global_rank)
        bs = self.config["batch_size"]
        with torch.inference_mode():
            done = 0
            while done < mine:
                n = min(bs, mine - done)
                x_0 = self.dataset.batch(bs if order is not None else n, self.device)["x_0"]
                # a device pipeline serves its epoch share in batches of bs and a ragged last one; with an explicit num_images the rank stops at
                # its dispatched share (the last batch is cut, not scored whole)
                x_0 = x_0[:min(x_0.shape[0], mine - done)]
                n = x_0.shape[0]
                rec = self.gaussian_diffusion.representation_learning_autoencoding(encoder_style, decoder_style, self.encoder, self.decoder, x_0)
                # one fused kernel: (x+1)/2 of both batches, SSIM and MSE per image (autoencoding_eval.py:83-88 + metric/utils.py:35-63)
                s, m = ssim_mse(x_0, rec, denormalize=True)
                self.ssim_metric.results.extend(s.tolist())
                self.mse_metric.results.extend(m.tolist())
                done += n
        ssim = self.ssim_metric.all_gather_results(self.global_world_size)
        mse = self.mse_metric.all_gather_results(self.global_world_size)
        out = {"ssim": self.ssim_metric.compute_metrics(ssim), "mse": self.mse_metric.compute_metrics(mse), "n": len(ssim)}
        if self.global_rank == 0:
            print("ssim: ", out["ssim"], "mse: ", out["mse"])
        return out


if __name__ == "__main__":
    cfg = {
        "diffusion_config": {"timesteps": 1000, "betas_type": "linear"},
        "config_path": "./trained-models/autoencoder/ffhq128/config.yml",
        "checkpoint_path": "./trained-models/autoencoder/ffhq128/checkpoint.pt",
        "trained_ddpm_config_path": "./pre-trained-dpms/ffhq128/config.yml",
        "dataset_config": {"dataset_name": "SYNTHETIC", "image_channel": 3, "image_size": 128, "length": 100},
        "batch_size": 100,
    }
    Sampler(cfg).start()
