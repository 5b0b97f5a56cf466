"""CPU: numerical model of the split-operand MFMA formats (pdae_conv_desc.math 3 / 4), independent of the kernels.

The GPU tests measure the kernels against fp64; this file pins the ARITHMETIC the kernels implement -- plane splits, product sets, power-of-two
scales -- with numpy, so the error levels quoted in DESIGN.md section 4 follow from the format itself:
  bf16x6: x = hi + mid + lo (three truncated bf16 planes, exact), products a0b0+a0b1+a1b0+a1b1+a0b2+a2b0
  f16x3 : x*s = hi + lo (two fp16 planes, rn), products a0b1+a1b0+a0b0, s a power of two undone exactly afterwards."""
import numpy as np


def bf16_trunc(x):
    return (x.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split_bf16x3(x):
    hi = bf16_trunc(x); r = x - hi
    mid = bf16_trunc(r); lo = bf16_trunc(r - mid)
    return hi, mid, lo


def split_f16x2(x):
    hi = x.astype(np.float16).astype(np.float32)
    lo = (x - hi).astype(np.float16).astype(np.float32)
    return hi, lo


def pow2_scale(amax):
    """Mirror of w_pow2_scale / p_pow2_scale: 2^(10 - floor(log2(amax))) from the exponent bits; 1 for 0 / inf / nan."""
    ex = (np.float32(amax).view(np.uint32) >> 23) & 0xFF
    if ex == 0 or ex == 255:
        return np.float32(1.0)
    sb = min(max(127 + 10 - (int(ex) - 127), 1), 254)
    return np.uint32(sb << 23).view(np.float32)


def mm(a, b):                                    # fp32 accumulate of exact plane products (every plane product is exact in fp32)
    return a.astype(np.float64) @ b.astype(np.float64)


def rel(a, ref):
    return float(np.linalg.norm(a - ref) / np.linalg.norm(ref))


def _operands(scale_a=1.0, scale_b=1.0, K=2304):
    rng = np.random.default_rng(0)
    A = (rng.standard_normal((64, K)) * scale_a).astype(np.float32)
    B = (rng.standard_normal((K, 48)) / np.sqrt(K) * scale_b).astype(np.float32)
    return A, B, A.astype(np.float64) @ B.astype(np.float64)


def test_bf16_three_plane_split_is_exact_and_six_products_are_fp32_grade():
    A, B, ref = _operands()
    a, b = split_bf16x3(A), split_bf16x3(B)
    assert np.array_equal(a[0] + a[1] + a[2], A) and np.array_equal(b[0] + b[1] + b[2], B)       # error-free transformation
    six = mm(a[0], b[0]) + mm(a[0], b[1]) + mm(a[1], b[0]) + mm(a[1], b[1]) + mm(a[0], b[2]) + mm(a[2], b[0])
    three = mm(a[0], b[0]) + mm(a[0], b[1]) + mm(a[1], b[0])
    assert rel(six, ref) < 2e-7                   # dropped terms ~2^-24
    assert 1e-6 < rel(three, ref) < 1e-4          # the 3-product bf16 form is ~2^-16: why it is not the default


def test_f16_two_plane_split_with_pow2_scales_is_fp32_grade():
    for sa, sb in ((1.0, 1.0), (30.0, 1.0), (1.0, 0.02)):
        A, B, ref = _operands(sa, sb)
        ws = np.float32(2.0 ** np.ceil(np.log2(np.sqrt(A.shape[1]))))            # weight scale 2^ceil(log2 sqrt(fan_in))
        a, b = split_f16x2(A * np.float32(16.0)), split_f16x2(B * ws)
        out = (mm(a[0], b[1]) + mm(a[1], b[0]) + mm(a[0], b[0])) / (16.0 * float(ws))
        assert rel(out, ref) < 1e-6, (sa, sb, rel(out, ref))          # 50x smaller weights than 1/sqrt(fan_in): 6.5e-7, still fp32 grade


def test_dynamic_pow2_scale_keeps_tiny_gradients_in_the_fp16_window():
    for g in (3e-7, 1.0, 2e4, 6.6e-5, 1.5e-8):
        A, B, ref = _operands(g, 1.0)
        s = pow2_scale(np.abs(A).max())
        assert 1024.0 <= np.abs(A).max() * s < 2048.0 and float(np.log2(s)) == round(float(np.log2(s)))
        ws = np.float32(2.0 ** np.ceil(np.log2(np.sqrt(A.shape[1]))))
        a, b = split_f16x2(A * s), split_f16x2(B * ws)
        out = (mm(a[0], b[1]) + mm(a[1], b[0]) + mm(a[0], b[0])) / (float(s) * float(ws))
        assert rel(out, ref) < 1e-6, (g, rel(out, ref))
    assert pow2_scale(0.0) == 1.0 and pow2_scale(np.inf) == 1.0
    # without the scale the same tiny gradient loses everything below the fp16 subnormals
    A, B, ref = _operands(3e-7, 1.0)
    a, b = split_f16x2(A), split_f16x2(B)
    assert rel(mm(a[0], b[1]) + mm(a[1], b[0]) + mm(a[0], b[0]), ref) > 1e-2
