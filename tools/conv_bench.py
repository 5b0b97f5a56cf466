"""Micro-benchmark of the dominant conv shapes of the FFHQ-128 step in every MFMA math mode (TFLOP/s algorithmic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdae_amd import hip as H

SHAPES = [  # N, H, Cin(C0,C1), Cout, k
    (32, 128, 128, 0, 128, 3), (32, 128, 128, 128, 128, 3), (32, 64, 256, 0, 256, 3), (32, 32, 256, 0, 256, 3),
    (32, 16, 384, 0, 384, 3), (32, 8, 512, 0, 512, 3), (32, 8, 512, 512, 512, 3), (32, 128, 128, 128, 128, 1),
]


def timeit(op, n=5):
    H.run(op); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        H.run(op)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (N, S, C0, C1, Cout, k) in SHAPES:
    Cin = C0 + C1
    x0 = torch.randn(N, S, S, C0, device="cuda")
    x1 = torch.randn(N, S, S, C1, device="cuda") if C1 else None
    w = torch.randn(Cout, k, k, Cin, device="cuda") / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, device="cuda")
    y = torch.empty(N, S, S, Cout, device="cuda"); dy = torch.randn_like(y)
    dx = torch.empty(N, S, S, Cin, device="cuda"); dw = torch.empty_like(w)
    fl = 2.0 * N * S * S * Cout * k * k * Cin
    line = f"N{N} {S}x{S} {Cin}->{Cout} k{k} ({fl/1e9:6.1f} GF):"
    for m, name in [(0, "f32"), (3, "x6"), (2, "x3"), (1, "bf16")]:
        c = H.Conv(N, S, S, C0, C1, Cout, k=k, math=m)
        wsb = c.wgrad_ws_bytes(); wsp = torch.empty(wsb // 4 + 16, device="cuda")
        wp = wp_t = None
        if c.wprep_bytes(0):
            wp = torch.empty(c.wprep_bytes(0) // 4, device="cuda"); H.run(H.op_conv_wprep(c, w, 0, wp))
        if c.wprep_bytes(1):
            wp_t = torch.empty(c.wprep_bytes(1) // 4, device="cuda"); H.run(H.op_conv_wprep(c, w, 1, wp_t))
        tf = [fl / timeit(op) / 1e9 for op in (H.op_conv_fwd(c, x0, x1, w, b, y, wp=wp), H.op_conv_dgrad(c, dy, w, dx, wp_t=wp_t), H.op_conv_wgrad(c, x0, x1, dy, dw, wsp, wsb))]
        line += f"  {name}: fwd {tf[0]:6.1f} dgrad {tf[1]:6.1f} wgrad {tf[2]:6.1f} |"
    print(line, flush=True)
