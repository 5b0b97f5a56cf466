"""Same-box timing of the Winograd-along-x patch kernel (conv3x3x: PDAE_W1=2) against the default routing (conv3x3r on these shapes: PDAE_W1=0) on
the large forward / data-gradient shapes of the FFHQ-128 step: plain, fused-GroupNorm and data-gradient launches.  Weights are prepared under the
same setting as the launch.  Usage: python tools/x_bench.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdae_amd import hip as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [(B, 128, 128, 0, 128), (B, 128, 128, 128, 128), (B, 64, 128, 0, 128), (B, 64, 256, 0, 256), (B, 32, 256, 0, 256)]


def timeit(op, n=10):
    for _ in range(3):
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
    coef = torch.zeros(3, N, Cin, device="cuda"); coef[1] = 1.0
    amax = torch.empty(4, device="cuda"); H.run(H.op_amax(dy, dy.numel(), amax))
    t, outs = {}, {}
    for rep in range(2):
        for mode, env in (("r", "0"), ("x", "2")):
            os.environ["PDAE_W1"] = env
            wp = torch.empty(c1.wprep_bytes(0) // 4, device="cuda"); H.run(H.op_conv_wprep(c1, w, 0, wp))
            wpg = torch.empty(c2.wprep_bytes(0, gn=True) // 4, device="cuda"); H.run(H.op_conv_wprep(c2, w, 4, wpg))
            ops = {"fwd": H.op_conv_fwd(c1, xa, None, w, b, y, wp=wp), "fwd_gn": H.op_conv_fwd_gn(c2, x0, x1, coef, 1, wpg, b, y)}
            if Cin % 128 == 0 and c1.wprep_bytes(1, f16_grad=True):
                wpt = torch.empty(c1.wprep_bytes(1, f16_grad=True) // 4, device="cuda"); H.run(H.op_conv_wprep(c1, w, 1 | 16, wpt))
                ops["dgrad"] = H.op_conv_dgrad(c1, dy, w, dx, wp_t=wpt, dy_amax=amax)
            for name, op in ops.items():
                t[(name, mode)] = min(t.get((name, mode), 1e9), timeit(op))
            H.run(ops["fwd"]); torch.cuda.synchronize(); outs[mode] = y.clone()
    err = float((outs["x"] - outs["r"]).abs().max() / outs["r"].abs().max())
    line = f"N{N} {S}x{S} {Cin}->{Cout} ({fl/1e9:6.1f} GF):"
    for name in ("fwd", "fwd_gn", "dgrad"):
        if (name, "r") in t:
            r, xx = t[(name, "r")], t[(name, "x")]
            line += f"  {name}: direct {r:.3f} ms {fl/r/1e9:4.0f} TF | W1 {xx:.3f} ms {fl/xx/1e9:4.0f} TF-equiv ({r/xx:.2f}x) |"
    print(line + f" max diff {err:.1e}", flush=True)
