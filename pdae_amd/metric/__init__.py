"""Evaluators of sampler/autoencoding_eval.py: per-image SSIM (11x11 Gaussian, sigma 1.5, zero padded) and MSE,
metric/utils.py:35-63.  Evaluation-only code (not on the training hot path): plain device tensor ops."""
import math

import torch
import torch.nn.functional as F


def _window(channel, window_size, device):
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    return g.mm(g.t()).float()[None, None].expand(channel, 1, window_size, window_size).contiguous().to(device)


def calculate_ssim(img1, img2, window_size=11):
    c = img1.shape[1]
    w = _window(c, window_size, img1.device)
    pad = window_size // 2
    mu1, mu2 = F.conv2d(img1, w, padding=pad, groups=c), F.conv2d(img2, w, padding=pad, groups=c)
    s1 = F.conv2d(img1 * img1, w, padding=pad, groups=c) - mu1 * mu1
    s2 = F.conv2d(img2 * img2, w, padding=pad, groups=c) - mu2 * mu2
    s12 = F.conv2d(img1 * img2, w, padding=pad, groups=c) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    return m.mean(1).mean(1).mean(1)


def calculate_mse(img1, img2):
    return (img1 - img2).pow(2).mean(dim=[1, 2, 3])


class _Accumulating:
    """metric/base_metric.py: process() per batch, all_gather_results(), compute_metrics()."""

    def __init__(self):
        self.results = []

    def all_gather_results(self, world_size):
        if world_size <= 1 or not torch.distributed.is_initialized():
            return list(self.results)
        gathered = [None for _ in range(world_size)]
        torch.distributed.all_gather_object(gathered, self.results)
        return [v for part in gathered for v in part]

    @staticmethod
    def compute_metrics(results):
        return float(sum(results) / max(len(results), 1))


class SSIMMetric(_Accumulating):
    def process(self, a, b):
        self.results.extend(calculate_ssim(a, b).tolist())


class MSEMetric(_Accumulating):
    def process(self, a, b):
        self.results.extend(calculate_mse(a, b).tolist())
