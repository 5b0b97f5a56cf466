"""CPU: the algebra behind the Winograd-along-x kernels, in fp64 numpy (no GPU, no library).

conv3x3y.hip computes, per output pixel pair (x, x + 1), row tap and channel,
    s = B^T d = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)          d0..d3 = input pixels 2p - 1 .. 2p + 2
    u = G g   = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2)      (wprepx_slot, conv3x3p.h)
    M = s * u;   Y0 = M0 + M1 + M2,  Y1 = M1 - M2 - M3
and DESIGN.md section 9 names the transposed identity for the weight gradient (F(3, 2)): with a = (e0, e0 + e1, e0 - e1, -e1) of an output-gradient
pair (e0, e1) and N_c = sum over pairs of a_c s_c,  dW = G^T N = (N0 + (N1 + N2) / 2, (N1 - N2) / 2, (N1 + N2) / 2 + N3).  Both are checked here
against plain correlation on random data, so that a kernel built on either has a pinned reference for its transform constants."""
import numpy as np

G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
BT = np.array([[1.0, 0.0, -1.0, 0.0], [0.0, 1.0, 1.0, 0.0], [0.0, -1.0, 1.0, 0.0], [0.0, 1.0, 0.0, -1.0]])
AT = np.array([[1.0, 1.0, 1.0, 0.0], [0.0, 1.0, -1.0, -1.0]])


def test_forward_f23_along_x_equals_the_three_tap_correlation():
    rng = np.random.default_rng(0)
    W, C = 16, 5
    x = rng.standard_normal((W + 2, C))          # one padded row: pixels -1 .. W
    g = rng.standard_normal((3, C))
    ref = np.array([sum((x[i + k] * g[k]).sum() for k in range(3)) for i in range(W)])
    out = np.empty(W)
    for p in range(W // 2):
        d = x[2 * p:2 * p + 4]                   # pixels 2p - 1 .. 2p + 2 (index shifted by the padding pixel)
        s = BT @ d                               # [4][C]
        u = G @ g                                # [4][C]
        m = (s * u).sum(1)
        out[2 * p:2 * p + 2] = AT @ m
    assert np.allclose(out, ref, rtol=0, atol=1e-12)
    # the transform constants as the kernels write them
    d = x[0:4]
    assert np.allclose(BT @ d, np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]))
    assert np.allclose(G @ g, np.stack([g[0], (g[0] + g[1] + g[2]) / 2, (g[0] - g[1] + g[2]) / 2, g[2]]))
    # input magnitude: |s| <= 2 max|d| -- the reason the fp16-window guard of conv3x3y scales the tracked maximum by two
    assert np.abs(BT @ d).max() <= 2 * np.abs(d).max() + 1e-12


def test_a_1x1_convolution_in_the_transform_domain_uses_positions_1_and_2():
    """The centre tap alone (a fused 1x1 skip in the x direction): g = (0, w, 0) => u = (0, w / 2, -w / 2, 0)."""
    w = 0.7
    u = G @ np.array([0.0, w, 0.0])
    assert np.allclose(u, [0.0, w / 2, -w / 2, 0.0])
    d = np.array([0.3, -1.1, 2.0, 0.4])
    assert np.allclose(AT @ ((BT @ d) * u), [w * d[1], w * d[2]])


def test_weight_gradient_f32_identity():
    rng = np.random.default_rng(1)
    W = 32
    x = rng.standard_normal(W + 2)
    e = rng.standard_normal(W)                   # output gradient of one row, one (co, ci) pair
    ref = np.array([sum(e[i] * x[i + k] for i in range(W)) for k in range(3)])
    N = np.zeros(4)
    for p in range(W // 2):
        s = BT @ x[2 * p:2 * p + 4]
        a = np.array([e[2 * p], e[2 * p] + e[2 * p + 1], e[2 * p] - e[2 * p + 1], -e[2 * p + 1]])      # = A e
        assert np.allclose(a, AT.T @ e[2 * p:2 * p + 2])
        N += a * s
    dW = G.T @ N
    assert np.allclose(dW, ref, rtol=0, atol=1e-12)
    assert np.allclose(dW, [N[0] + (N[1] + N[2]) / 2, (N[1] - N[2]) / 2, (N[1] + N[2]) / 2 + N[3]])


def test_2d_f2x2_3x3_identity_of_the_gated_probe():
    """tools/probes/r04_winograd/winograd.hip (the gated 2-D probe of round 4, no longer in the library): V = B^T d B, U = G g G^T, Y = A^T (U * V) A on a 4 x 4 input tile."""
    rng = np.random.default_rng(2)
    d = rng.standard_normal((4, 4)); g = rng.standard_normal((3, 3))
    ref = np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(2)] for i in range(2)])
    Y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
    assert np.allclose(Y, ref, rtol=0, atol=1e-12)
