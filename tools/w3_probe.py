"""Times a few 3x3 weight-gradient launches (B=32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pdae_amd import hip as H
def t(op, n=10):
    H.run(op); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): H.run(op)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = []
for (N, S, C, Cout) in [(32, 128, 128, 128), (32, 64, 128, 128), (32, 64, 256, 256), (32, 32, 256, 256), (32, 16, 384, 384), (32, 8, 512, 512)]:
    x = torch.randn(N, S, S, C, device="cuda"); dy = torch.randn(N, S, S, Cout, device="cuda") * 1e-4
    c = H.Conv(N, S, S, C, 0, Cout, math=4)
    wsb = c.wgrad_ws_bytes()
    ws = torch.empty(wsb // 4 + 64, device="cuda"); dw = torch.empty(Cout, 3, 3, C, device="cuda"); db = torch.empty(Cout, device="cuda")
    am = torch.empty(4, device="cuda"); H.run(H.op_amax(dy, dy.numel(), am))
    op = H.op_conv_wgrad(c, x, None, dy, dw, ws, wsb, db=db, dy_amax=am)
    ms = t(op)
    out.append(f"{C}->{Cout}@{S}: {ms:.3f} ms {2.0*N*S*S*Cout*9*C/ms/1e9:6.1f} TF")
print(" | ".join(out), flush=True)
