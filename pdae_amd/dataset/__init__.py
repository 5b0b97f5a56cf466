"""Input pipeline: LMDB-free, with the per-image work of the reference's DataLoader workers moved onto the GPU.

The reference reads JPEG/PNG bytes from LMDB, decodes with PIL and runs torchvision transforms per image on worker processes
(dataset/ffhq.py:19-53), then collates to {"idx", "x_0", "gts"} (:55-74).  Here a dataset is a set of `.npy` shards of DECODED uint8 images
([n, H, W, C]; memory-mapped, so a 70 000-image FFHQ-256 set is a 13.8 GB file set that never enters the Python heap); a batch is
  1. gathered from the memory map into one of two pinned host buffers by a background thread,
  2. copied to the device on a side stream (double buffered: the copy of batch k+1 overlaps the training step of batch k),
  3. cropped / resized (PIL-exact antialiased bilinear) / flipped / normalised by ONE pdae_image_prepare call (csrc/image.hip),
and handed over as {"idx": LongTensor, "x_0": float32 [B,C,S,S] in [-1,1], "gts": uint8 [B,S,S,C]} -- the collate contract.
`SYNTHETIC` (BASELINE.json: synthetic images of the named resolution) draws x_0 on the device and keeps the same contract.
"""
import ctypes
import glob
import os
import queue
import threading

import numpy as np
import torch

from .. import hip as H
from .resample import bilinear_coefficients

CROPS = {"CELEBA64": (57, 25, 128, 128)}          # dataset/celeba64.py:11-14: F.crop(img, top, left, height, width) in front of the resize


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


class ShardedImages:
    """Decoded uint8 images in `<data_path>/*.npy` shards ([n,H,W,C] or [n,H,W]); index = position in the sorted shard list."""

    def __init__(self, data_path):
        files = sorted(glob.glob(os.path.join(data_path, "*.npy")))
        if not files:
            raise FileNotFoundError(f"no .npy image shards under {data_path!r}")
        self.shards = [np.load(f, mmap_mode="r") for f in files]
        for s in self.shards:
            if s.dtype != np.uint8 or s.ndim not in (3, 4) or s.shape[1:] != self.shards[0].shape[1:]:
                raise ValueError("image shards must be uint8 [n,H,W,C] arrays of one common image shape")
        self.starts = np.cumsum([0] + [s.shape[0] for s in self.shards])
        self.shape = tuple(self.shards[0].shape[1:]) + (() if self.shards[0].ndim == 4 else (1,))

    def __len__(self):
        return int(self.starts[-1])

    def gather(self, indices, out):
        """out[k] = image indices[k]  (out: uint8 [B,H,W,C] numpy view of a pinned buffer)."""
        for k, i in enumerate(indices):
            s = int(np.searchsorted(self.starts, i, side="right")) - 1
            img = self.shards[s][i - self.starts[s]]
            out[k] = img if img.ndim == 3 else img[..., None]


class EpochOrder:
    """Which images a rank visits, in which order, and which of them are mirrored: DistributedSampler semantics
    (base_trainer.py:73-78: shuffle=True for training, a fresh permutation per epoch shared by all ranks, rank-strided shares padded to equal
    length by wrapping around) plus the per-image coin of RandomHorizontalFlip (dataset/ffhq.py:22-23).  Pure numpy: the multi-rank behaviour is
    tested on CPU (tests/test_host_cpu.py)."""

    def __init__(self, length, rank=0, world_size=1, seed=0, shuffle=True, flip=True, drop_tail=False):
        if not (0 <= rank < world_size):
            raise ValueError(f"rank {rank} outside world of {world_size}")
        self.length, self.rank, self.world, self.seed = int(length), int(rank), int(world_size), int(seed)
        self.shuffle, self.flip = bool(shuffle), bool(flip)
        # drop_tail: the TRAINING sampler of the reference is DistributedSampler(drop_last=True) (base_trainer.py:73-78): the len % world tail of the
        # permutation is dropped, floor(len / world) images per rank, no image twice in an epoch.  The evaluator's sampler
        # (sampler/autoencoding_eval.py:26-43, drop_last=False) pads by wrapping around: ceil(len / world) per rank, every image visited.
        self.drop_tail = bool(drop_tail)
        self.per_rank = self.length // self.world if self.drop_tail else (self.length + self.world - 1) // self.world

    def indices(self, epoch):
        """This rank's image indices of `epoch` (every rank the same count; the union over ranks covers the dataset, minus the dropped tail)."""
        order = np.random.default_rng([self.seed, epoch]).permutation(self.length) if self.shuffle else np.arange(self.length)
        if self.drop_tail:
            order = order[:self.per_rank * self.world]
        else:
            pad = self.per_rank * self.world - self.length
            if pad:
                order = np.concatenate([order, order[:pad]])
        return order[self.rank::self.world]

    def flips(self, epoch, start, count):
        """Mirror flags of the `count` images at position `start` of this rank's epoch share (independent streams per rank and epoch)."""
        if not self.flip:
            return np.zeros(count, dtype=np.uint8)
        return (np.random.default_rng([self.seed, epoch, self.rank, start, 0x666c6970]).random(count) < 0.5).astype(np.uint8)


class DeviceImagePipeline:
    """Double-buffered host -> device image batches with on-GPU crop / resize / flip / normalise (module docstring)."""

    def __init__(self, config, device, rank=0, world_size=1, seed=0, shuffle=True, drop_last=True):
        self.device = torch.device(device)
        self.drop_last = bool(drop_last)       # training loaders drop the ragged last batch of an epoch (base_trainer.py:86), the evaluator serves it
        if self.device.type != "cuda":
            raise H.PdaeError("the device image pipeline needs a ROCm device (pdae_image_prepare has no CPU fallback)")
        self.size = int(config["image_size"])
        self.augmentation = bool(config.get("augmentation", True))
        self.images = ShardedImages(config["data_path"])
        Hs, Ws, C = self.images.shape
        if C != int(config.get("image_channel", C)):
            raise ValueError(f"image_channel {config.get('image_channel')} but the shards hold {C}-channel images")
        self.crop = tuple(config["crop"]) if config.get("crop") else CROPS.get(str(config.get("name", "")).upper(), (0, 0, Hs, Ws))
        cy, cx, ch, cw = self.crop
        kx, bx = bilinear_coefficients(cw, self.size)
        ky, by = bilinear_coefficients(ch, self.size)
        dev = self.device
        self._tabs = [torch.from_numpy(a).to(dev) for a in (kx, bx, ky, by)]
        self._ks = (kx.shape[1], ky.shape[1])
        # shuffling is a property of the split (train: yes, eval / inference: no), the mirror flip of `augmentation` (dataset/ffhq.py:21-23)
        # drop_last pipelines are the training ones: their sampler also drops the len % world tail (EpochOrder.drop_tail)
        self.order = EpochOrder(len(self.images), rank, world_size, seed, shuffle=bool(config.get("shuffle", shuffle)), flip=self.augmentation,
                                drop_tail=self.drop_last)
        self.rank, self.world, self.seed, self.epoch = rank, world_size, seed, 0
        self._batch = None
        self._stream = torch.cuda.Stream(device=dev)
        self._slots = None
        self._q = None
        self._thread = None

    def __len__(self):
        return len(self.images)

    def _alloc(self, B):
        Hs, Ws, C = self.images.shape
        L = H.lib()
        slots = []
        for _ in range(2):
            host = torch.empty(B, Hs, Ws, C, dtype=torch.uint8).pin_memory()
            slots.append(dict(host=host, np=host.numpy(), flip_host=torch.zeros(B, dtype=torch.uint8).pin_memory(),
                              dev=torch.empty(B, Hs, Ws, C, dtype=torch.uint8, device=self.device),
                              flip=torch.empty(B, dtype=torch.uint8, device=self.device),
                              ws=torch.empty(int(L.pdae_image_prepare_workspace_bytes(B, self.crop[2], self.size, C)) + 16, dtype=torch.uint8, device=self.device),
                              ready=torch.cuda.Event(), done=None, free=threading.Event()))
            slots[-1]["free"].set()
        return slots

    def _producer(self, B):
        k, epoch = 0, self.epoch
        while True:
            idx = self.order.indices(epoch)
            for s in range(0, len(idx), B):
                ids = idx[s:s + B]
                n = len(ids)
                if n < B and self.drop_last:                       # the reference's training DataLoader drops it (base_trainer.py:86)
                    break
                slot = self._slots[k % 2]
                slot["free"].wait()                                # the consumer has issued the kernel that reads this slot's device buffer
                slot["free"].clear()
                if slot["done"] is not None:
                    slot["ready"].synchronize()                    # the previous H2D out of this pinned buffer has finished
                    self._stream.wait_event(slot["done"])          # the next H2D into the device buffer queues behind that kernel (no host wait)
                self.images.gather(ids, slot["np"][:n])
                slot["flip_host"][:n].copy_(torch.from_numpy(self.order.flips(epoch, s, n)))
                with torch.cuda.stream(self._stream):
                    slot["dev"].copy_(slot["host"], non_blocking=True)
                    slot["flip"].copy_(slot["flip_host"], non_blocking=True)
                    slot["ready"].record(self._stream)
                self._q.put((k % 2, torch.from_numpy(np.asarray(ids, dtype=np.int64))))
                k += 1
            epoch += 1

    def batch(self, batch_size, device=None, generator=None, out=None):
        """Next batch of this rank's epoch share: `batch_size` images, or the ragged remainder of the epoch when the pipeline was built with
        drop_last=False (read the size off x_0).  `out`: optional float32 destination (any strides, e.g. the NHWC plan buffer viewed as (B,C,S,S))."""
        if self._thread is None:
            if batch_size > len(self.order.indices(0)) and self.drop_last:      # without drop_last a short share is served as one ragged batch
                raise ValueError(f"batch of {batch_size} images but this rank's share of the dataset holds {len(self.order.indices(0))}")
            self._batch = int(batch_size)
            self._slots = self._alloc(batch_size)
            self._q = queue.Queue(maxsize=2)
            self._thread = threading.Thread(target=self._producer, args=(batch_size,), daemon=True)
            self._thread.start()
        if batch_size != self._batch:
            # the producer thread, the pinned / device slots and the index bookkeeping are sized by the first request; a different size would
            # hand back an `idx` that does not describe x_0.  Callers that need another size (an evaluation batch) build their own pipeline.
            raise ValueError(f"this pipeline serves batches of {self._batch} images (got a request for {batch_size}): build a second pipeline")
        k, ids = self._q.get()
        slot = self._slots[k]
        batch_size = int(ids.numel())
        Hs, Ws, C = self.images.shape
        S = self.size
        x0 = torch.empty(batch_size, C, S, S, device=self.device) if out is None else out[:batch_size]
        gts = torch.empty(batch_size, S, S, C, dtype=torch.uint8, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(slot["ready"])                              # the H2D copy of this slot ran on the side stream
        kx, bx, ky, by = self._tabs
        cy, cx, ch, cw = self.crop
        strides = (ctypes.c_int64 * 4)(*x0.stride())
        L = H.lib()
        rc = L.pdae_image_prepare(slot["dev"].data_ptr(), batch_size, Hs, Ws, C, cy, cx, ch, cw, S, kx.data_ptr(), bx.data_ptr(), self._ks[0],
                                  ky.data_ptr(), by.data_ptr(), self._ks[1], slot["flip"].data_ptr(), x0.data_ptr(), strides, gts.data_ptr(),
                                  slot["ws"].data_ptr(), ctypes.c_void_p(cur.cuda_stream))
        if rc != 0:
            raise H.PdaeError(f"pdae_image_prepare failed ({rc}): {L.pdae_last_error().decode()}")
        slot["done"] = torch.cuda.Event()
        slot["done"].record(cur)
        slot["free"].set()
        return {"idx": ids, "x_0": x0, "gts": gts}


def build(config, device=None, rank=0, world_size=1, seed=0, shuffle=True, drop_last=True):
    """dataset_module.build(train_dataset_config): 'SYNTHETIC', or any other name with a `data_path` of .npy shards.
    rank / world_size / seed: this process's share of every epoch (the reference wraps its datasets in a DistributedSampler,
    base_trainer.py:73-78); seed must be the same on all ranks.  shuffle / drop_last: True for training splits; the autoencoding evaluator reads its
    dataset in order and to the last image (sampler/autoencoding_eval.py:26-43)."""
    name = config.get("name", config.get("dataset_name", "SYNTHETIC"))
    if name == "SYNTHETIC":
        return SYNTHETIC(config)
    if not config.get("data_path"):
        raise NotImplementedError(f"dataset {name!r}: give train_dataset_config.data_path = a directory of decoded uint8 .npy shards "
                                  "(the reference's LMDB / torchvision readers are not part of pdae_amd)")
    return DeviceImagePipeline(config, device if device is not None else torch.device("cuda", torch.cuda.current_device()), rank, world_size, seed, shuffle,
                               drop_last)
