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
if len(sys.argv) > 1: H.set_knob("PDAE_W3V", int(sys.argv[1]))
out = []
only = os.environ.get("W3_ONLY")
for (N, S, C, Cout) in {"1": [(32, 128, 128, 128)], "gn": [(32, 128, 256, 128)]}[only] if only else [(32, 128, 256, 128), (32, 128, 128, 128), (32, 64, 128, 128), (32, 64, 256, 256), (32, 32, 256, 256), (32, 16, 384, 384), (32, 8, 512, 512)]:
    x = torch.randn(N, S, S, C, device="cuda"); dy = torch.randn(N, S, S, Cout, device="cuda") * 1e-4
    c = H.Conv(N, S, S, C, 0, Cout, math=4)
    wsb = c.wgrad_ws_bytes()
    ws = torch.empty(wsb // 4 + 64, device="cuda"); dw = torch.empty(Cout, 3, 3, C, device="cuda"); db = torch.empty(Cout, device="cuda")
    am = torch.empty(4, device="cuda"); H.run(H.op_amax(dy, dy.numel(), am))
    op = H.op_conv_wgrad(c, x, None, dy, dw, ws, wsb, db=db, dy_amax=am)
    ms = t(op)
    out.append(f"{C}->{Cout}@{S}: {ms:.3f} ms {2.0*N*S*S*Cout*9*C/ms/1e9:6.1f} TF")
if only == "1": print(" | ".join(out), flush=True); sys.exit(0)
# the GroupNorm-recomputing launch of the step: raw 128 + 128 concat -> 128 at 128^2
N, S, C0, C1, Cout = 32, 128, 128, 128, 128
x0 = torch.randn(N, S, S, C0, device="cuda"); x1 = torch.randn(N, S, S, C1, device="cuda"); dy = torch.randn(N, S, S, Cout, device="cuda") * 1e-4
coef = torch.randn(3, N, C0 + C1, device="cuda") * 0.1 + 1.0
c = H.Conv(N, S, S, C0, C1, Cout, math=4)
wsb = c.wgrad_ws_bytes()
ws = torch.empty(wsb // 4 + 64, device="cuda"); dw = torch.empty(Cout, 3, 3, C0 + C1, device="cuda"); db = torch.empty(Cout, device="cuda")
am = torch.empty(4, device="cuda"); H.run(H.op_amax(dy, dy.numel(), am))
op = H.op_conv_wgrad(c, x0, x1, dy, dw, ws, wsb, db=db, dy_amax=am, gn_coef=coef, gn_act=1)
ms = t(op)
out.append(f"GN {C0}+{C1}->{Cout}@{S}: {ms:.3f} ms {2.0*N*S*S*Cout*9*(C0+C1)/ms/1e9:6.1f} TF")
print(" | ".join(out), flush=True)
