"""Same-box timing of the Winograd F(2x2, 3x3) forward kernel (pdae_wino_fwd) against the direct patch kernels (conv3x3r: PDAE_P3R=2, conv3x3p:
PDAE_P3R=0) on the large weight-constant forward shapes of the FFHQ-128 step.  The gate of VERDICT r3 item 1: 128x128 128->128, B=32 in <= 0.30 ms
(conv3x3r: 0.376 ms).  Usage: python tools/wino_bench.py [batch]; PDAE_WINO_SCHED=0|1 selects the issue-pattern variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdae_amd import hip as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [(B, 128, 128, 128), (B, 128, 256, 128), (B, 64, 128, 128), (B, 64, 256, 256), (B, 32, 256, 256)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (N, S, Cin, Cout) in SHAPES:
    x = torch.randn(N, S, S, Cin, device="cuda")
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device="cuda")
    y = torch.empty(N, S, S, Cout, device="cuda"); yd = torch.empty_like(y)
    fl = 2.0 * N * S * S * Cout * 9 * Cin
    c = H.Conv(N, S, S, Cin, 0, Cout, k=3, math=4)
    wp = torch.empty(c.wprep_bytes(0) // 4, device="cuda"); H.run(H.op_conv_wprep(c, w, 0, wp))
    wpw = torch.empty(H.wino_wprep_bytes(c) // 4, device="cuda"); H.wino_wprep(c, w, wpw)
    op = H.op_conv_fwd(c, x, None, w, b, yd, wp=wp)
    t = {}
    for rep in range(2):
        for mode, r_ in (("p", "0"), ("r", "2")):
            os.environ["PDAE_P3R"] = r_
            t[mode] = min(t.get(mode, 1e9), timeit(lambda: H.run(op)))
        for sch in ("1", "2", "8", "9"):
            os.environ["PDAE_WINO_SCHED"] = sch
            t["w" + sch] = min(t.get("w" + sch, 1e9), timeit(lambda: H.wino_fwd(c, x, wpw, b, y)))
    H.run(op); H.wino_fwd(c, x, wpw, b, y); torch.cuda.synchronize()
    err = float((y - yd).abs().max() / yd.abs().max())
    print(f"N{N} {S}x{S} {Cin}->{Cout} ({fl/1e9:6.1f} GF): conv3x3p {t['p']:.3f} ms | conv3x3r {t['r']:.3f} ms {fl/t['r']/1e9:4.0f} TF | "
          f"winograd 4w sched1 {t['w1']:.3f} units {t['w2']:.3f} | 8w skew {t['w8']:.3f} plain {t['w9']:.3f} ms {fl/min(t['w8'],t['w9'])/1e9:4.0f} TF-equiv ({t['r']/min(t['w8'],t['w9']):.2f}x vs r) | "
          f"max diff vs direct {err:.1e}", flush=True)
