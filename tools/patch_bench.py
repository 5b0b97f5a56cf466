"""Same-box A/B of the two 3x3 patch kernels (conv3x3p: PDAE_P3R=0, conv3x3r: PDAE_P3R=2) on the large forward / data-gradient shapes of the
FFHQ-128 step: ms and algorithmic TFLOP/s per launch, plain and fused-GroupNorm forms.  Usage: python tools/patch_bench.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdae_amd import hip as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [(B, 128, 128, 0, 128), (B, 128, 128, 128, 128), (B, 64, 128, 0, 128), (B, 64, 256, 0, 256), (B, 64, 256, 128, 128)]


def timeit(op, n=8):
    for _ in range(2):
        H.run(op)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        H.run(op)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (N, S, C0, C1, Cout) in SHAPES:
    Cin = C0 + C1
    x0 = torch.randn(N, S, S, C0, device="cuda"); x1 = torch.randn(N, S, S, C1, device="cuda") if C1 else None
    xa = torch.randn(N, S, S, Cin, device="cuda")
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device="cuda")
    y = torch.empty(N, S, S, Cout, device="cuda"); dy = torch.randn_like(y) * 1e-4
    dx = torch.empty(N, S, S, Cin, device="cuda")
    fl = 2.0 * N * S * S * Cout * 9 * Cin
    c1 = H.Conv(N, S, S, Cin, 0, Cout, k=3, math=4)
    c2 = H.Conv(N, S, S, C0, C1, Cout, k=3, math=4)
    wp = torch.empty(c1.wprep_bytes(0) // 4, device="cuda"); H.run(H.op_conv_wprep(c1, w, 0, wp))
    wpg = torch.empty(c2.wprep_bytes(0, gn=True) // 4, device="cuda"); H.run(H.op_conv_wprep(c2, w, 4, wpg))
    coef = torch.zeros(3, N, Cin, device="cuda"); coef[1] = 1.0
    amax = torch.empty(4, device="cuda"); H.run(H.op_amax(dy, dy.numel(), amax))
    ops = {"fwd": H.op_conv_fwd(c1, xa, None, w, b, y, wp=wp), "fwd_gn": H.op_conv_fwd_gn(c2, x0, x1, coef, 1, wpg, b, y)}
    if Cin % 128 == 0 and c1.wprep_bytes(1, f16_grad=True):
        wpt = torch.empty(c1.wprep_bytes(1, f16_grad=True) // 4, device="cuda"); H.run(H.op_conv_wprep(c1, w, 1 | 16, wpt))
        ops["dgrad"] = H.op_conv_dgrad(c1, dy, w, dx, wp_t=wpt, dy_amax=amax)
    line = f"N{N} {S}x{S} {Cin}->{Cout} ({fl/1e9:6.1f} GF):"
    for name, op in ops.items():
        t = {}
        for rep in range(2):
            for mode, r_ in (("p", "0"), ("r", "2")):
                H.set_knob("PDAE_P3R", int(r_))
                t[mode] = min(t.get(mode, 1e9), timeit(op))
        line += f"  {name}: p {t['p']:.3f} ms {fl/t['p']/1e9:4.0f} TF | r {t['r']:.3f} {fl/t['r']/1e9:4.0f} TF ({t['p']/t['r']:.2f}x) |"
    print(line, flush=True)
