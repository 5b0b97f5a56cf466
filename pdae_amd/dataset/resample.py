"""Host side of the device resize: the coefficient tables of PIL's antialiased BILINEAR resampler (what torchvision's
transforms.Resize((S,S)) applies to the PIL images of dataset/ffhq.py:21) in its 8-bit fixed-point form.

For output index o over an input interval of length `in_size` (the crop):  scale = in_size / out_size, the triangle filter is stretched by
max(scale, 1) (antialiasing), taps cover [center - support, center + support) clipped to the interval, weights are normalised to sum 1 and
rounded to 22-bit integers.  The device kernels (csrc/image.hip) accumulate `pixel * coef` from 1 << 21 and shift by 22, once per axis with a
uint8 result after each axis -- the arithmetic of PIL's ImagingResampleHorizontal_8bpc / Vertical_8bpc, so results are bit-identical to PIL."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bilinear_coefficients(in_size, out_size):
    """(coef int32 [out_size, ksize], bounds int32 [out_size, 2] = (first input index, tap count))."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                      # bilinear (triangle) filter: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    inv = 1.0 / filterscale
    for o in range(out_size):
        center = (o + 0.5) * scale
        first = max(int(center - support + 0.5), 0)
        last = min(int(center + support + 0.5), in_size)
        n = last - first
        w = np.array([max(0.0, 1.0 - abs((k + first - center + 0.5) * inv)) for k in range(n)], dtype=np.float64)
        total = w.sum()
        if total != 0.0:
            w = w / total
        coef[o, :n] = [int(v * (1 << PRECISION_BITS) + 0.5) for v in w]
        bounds[o] = (first, n)
    return coef, bounds
