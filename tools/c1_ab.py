"""conv1x1 forward A/B on the step's 1x1 shapes: rotated stage order (PDAE_C1_ROT=1) against the lock-step order (0), same process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdae_amd import hip as H
SHAPES = [(32, 128, 128, 128, 128), (32, 64, 128, 128, 128), (32, 64, 256, 128, 128), (32, 64, 256, 128, 256), (32, 32, 256, 256, 256), (32, 32, 384, 256, 256), (32, 16, 384, 0, 1152), (32, 16, 384, 0, 384), (32, 16, 512, 384, 384),
          (32, 8, 512, 512, 512), (100, 128, 128, 128, 128), (100, 16, 384, 0, 1152)]
for (N, S, C0, C1, Cout) in SHAPES:
    x0 = torch.randn(N, S, S, C0, device="cuda"); x1 = torch.randn(N, S, S, C1, device="cuda") if C1 else None
    w = torch.randn(Cout, 1, 1, C0 + C1, device="cuda") / (C0 + C1) ** 0.5; b = torch.randn(Cout, device="cuda")
    c = H.Conv(N, S, S, C0, C1, Cout, k=1, math=4)
    wp = torch.empty(max(c.wprep_bytes(0), 4) // 4, device="cuda")
    H.run(H.op_conv_wprep(c, w, 0, wp))
    res, ys = {}, {}
    for pipe in (0, 1):
        H.set_knob("PDAE_C1_ROT", pipe)
        y = torch.empty(N, S, S, Cout, device="cuda")
        op = H.op_conv_fwd(c, x0, x1, w, b, y, wp=wp)
        for _ in range(30): H.run(op)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): H.run(op)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        res[pipe] = best; ys[pipe] = y
    gb = 4.0 * N * S * S * (C0 + C1 + Cout) / 1e9
    print(f"N{N} {S}x{S} {C0}+{C1}->{Cout}: lock-step {res[0]*1e3:7.1f} us {gb/res[0]:5.2f} TB/s | rotated {res[1]*1e3:7.1f} us {gb/res[1]:5.2f} TB/s  ({res[0]/res[1]:.2f}x)  max diff {float((ys[0]-ys[1]).abs().max()):.1e}", flush=True)
