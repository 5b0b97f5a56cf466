"""Input pipeline stand-in.  The reference's datasets are LMDB / torchvision readers (dataset/*.py) whose data, lmdb and
torchvision are absent here; BASELINE.json defines the metric on synthetic images.  SYNTHETIC honours the collate contract
{"idx", "x_0", "gts"} (dataset/ffhq.py:55-74): x_0 ~ U(-1,1) float32 [B,C,H,W], gts uint8 [B,H,W,C]."""
import torch


class SYNTHETIC:
    def __init__(self, config):
        self.size = int(config.get("image_size", 128))
        self.channel = int(config.get("image_channel", 3))
        self.length = int(config.get("length", 70000))

    def __len__(self):
        return self.length

    def batch(self, batch_size, device, generator=None):
        x = torch.rand(batch_size, self.channel, self.size, self.size, device=device, generator=generator) * 2 - 1
        gts = ((x + 1) * 127.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
        return {"idx": torch.arange(batch_size), "x_0": x, "gts": gts}


def build(config):
    name = config.get("name", config.get("dataset_name", "SYNTHETIC"))
    if name != "SYNTHETIC":
        raise NotImplementedError(f"dataset {name!r}: only the SYNTHETIC stand-in ships with pdae_amd (no lmdb/torchvision/data in this "
                                  "environment); plug a loader that yields {'idx','x_0','gts'} batches")
    return SYNTHETIC(config)
