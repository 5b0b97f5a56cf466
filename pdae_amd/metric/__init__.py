"""Evaluators of sampler/autoencoding_eval.py on the device: per-image SSIM and MSE from ONE fused kernel pass
(pdae_ssim_mse, csrc/metric.hip) instead of five depthwise convolutions and a dozen elementwise passes (metric/utils.py:35-63).

`ssim_mse(a, b, denormalize=True)` also folds the (x+1)/2 map of autoencoding_eval.py:83-88 into the kernel, so the evaluator reads the
two image batches exactly once, in whatever memory format they arrive (the DDIM loop returns NHWC-strided tensors)."""
import ctypes
import math

import numpy as np
import torch

from .. import hip as H

WINDOW_SIZE, SIGMA = 11, 1.5


def gaussian_window():
    """Normalised 1-D Gaussian in float32, rounded the way the reference builds it (float32 tensor of the samples divided by its float32 sum)."""
    g = np.array([math.exp(-(i - WINDOW_SIZE // 2) ** 2 / (2.0 * SIGMA ** 2)) for i in range(WINDOW_SIZE)], dtype=np.float32)
    return g / g.sum(dtype=np.float32)


_WINDOW = (ctypes.c_float * WINDOW_SIZE)(*gaussian_window().tolist())


def ssim_mse(img1, img2, denormalize=False, want_ssim=True, want_mse=True):
    """(ssim [N], mse [N]) of two (N,C,H,W) float32 device tensors of any strides.  denormalize: inputs are in [-1,1] and are mapped to [0,1] first."""
    if img1.shape != img2.shape or img1.dim() != 4:
        raise ValueError(f"ssim_mse: shapes {tuple(img1.shape)} / {tuple(img2.shape)}")
    if img1.device.type != "cuda":
        raise H.PdaeError("pdae_amd.metric runs on a ROCm device only (no CPU fallback)")
    img1, img2 = img1.float(), img2.float()
    N, C, Hh, W = img1.shape
    L = H.lib()
    ws = torch.empty(max(L.pdae_ssim_mse_workspace_bytes(N, C, Hh, W) // 4, 1), device=img1.device)
    s = torch.empty(N, device=img1.device) if want_ssim else None
    m = torch.empty(N, device=img1.device) if want_mse else None
    mul, add = (0.5, 0.5) if denormalize else (1.0, 0.0)
    sa, sb = (ctypes.c_int64 * 4)(*img1.stride()), (ctypes.c_int64 * 4)(*img2.stride())
    rc = L.pdae_ssim_mse(img1.data_ptr(), sa, img2.data_ptr(), sb, N, C, Hh, W, mul, add, _WINDOW, s.data_ptr() if want_ssim else None,
                         m.data_ptr() if want_mse else None, ws.data_ptr(), ctypes.c_void_p(H.current_stream_ptr()))
    if rc != 0:
        raise H.PdaeError(f"pdae_ssim_mse failed ({rc}): {L.pdae_last_error().decode()}")
    return s, m


def calculate_ssim(img1, img2, window_size=WINDOW_SIZE):                     # metric/utils.py:35-57
    if window_size != WINDOW_SIZE:
        raise NotImplementedError("the fused evaluator implements the reference's fixed 11-tap window")
    return ssim_mse(img1, img2, want_mse=False)[0]


def calculate_mse(img1, img2):                                               # metric/utils.py:62-63
    return ssim_mse(img1, img2, want_ssim=False)[1]


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
