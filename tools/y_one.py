"""One conv3x3y timing (Winograd along x, PDAE_W1 = 2): python tools/y_one.py [N S Cin Cout [gn]]; prints ms after a clock warm-up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PDAE_W1", "2")
import torch
from pdae_amd import hip as H
a = sys.argv[1:]
N, S, Cin, Cout = (int(v) for v in a[:4]) if len(a) >= 4 else (32, 128, 128, 128)
gn = len(a) > 4 and "gn" in a[4:]
res = len(a) > 4 and "res" in a[4:]              # same-resolution residual: the instantiation with an epilogue operand
x = torch.randn(N, S, S, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5; b = torch.randn(Cout, device="cuda")
y = torch.empty(N, S, S, Cout, device="cuda")
c = H.Conv(N, S, S, Cin, 0, Cout, k=3, math=4)
if gn:
    coef = torch.zeros(3, N, Cin, device="cuda"); coef[1] = 1.0
    wp = torch.empty(c.wprep_bytes(0, gn=True) // 4, device="cuda"); H.run(H.op_conv_wprep(c, w, 4, wp))
    op = H.op_conv_fwd_gn(c, x, None, coef, 1, wp, b, y, res=torch.randn_like(y) if res else None, res_mode=1 if res else 0)
else:
    wp = torch.empty(c.wprep_bytes(0) // 4, device="cuda"); H.run(H.op_conv_wprep(c, w, 0, wp))
    op = H.op_conv_fwd(c, x, None, w, b, y, wp=wp, res=torch.randn_like(y) if res else None, res_mode=1 if res else 0)
for _ in range(60): H.run(op)
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): H.run(op)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
print("ms %.4f" % best)
