"""Runs ONE conv shape in one math mode a few times (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdae_amd import hip as H
m = int(sys.argv[1]) if len(sys.argv) > 1 else 0
kind = sys.argv[2] if len(sys.argv) > 2 else "fwd"
N, S, C0, Cout, k = 32, 64, 256, 256, 3
x0 = torch.randn(N, S, S, C0, device="cuda"); w = torch.randn(Cout, k, k, C0, device="cuda") / (C0 * 9) ** 0.5
b = torch.randn(Cout, device="cuda"); y = torch.empty(N, S, S, Cout, device="cuda"); dy = torch.randn_like(y)
dx = torch.empty_like(x0); dw = torch.empty_like(w)
c = H.Conv(N, S, S, C0, 0, Cout, k=k, math=m)
wsb = c.wgrad_ws_bytes(); wsp = torch.empty(wsb // 4 + 16, device="cuda")
op = {"fwd": H.op_conv_fwd(c, x0, None, w, b, y), "dgrad": H.op_conv_dgrad(c, dy, w, dx), "wgrad": H.op_conv_wgrad(c, x0, None, dy, dw, wsp, wsb)}[kind]
for _ in range(3):
    H.run(op)
torch.cuda.synchronize()
