import sys, os
sys.path.insert(0, os.getcwd())
import torch
from pdae_amd import hip as H
def t(op, n=20):
    for _ in range(3): H.run(op)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): H.run(op)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (N, S, C, Cout) in [(32, 8, 512, 512), (32, 8, 1024, 512)]:
    x = torch.randn(N, S, S, C, device="cuda"); dy = torch.randn(N, S, S, Cout, device="cuda") * 1e-6
    dw = torch.empty(Cout, 3, 3, C, device="cuda"); db = torch.empty(Cout, device="cuda")
    am = torch.empty(4, device="cuda"); H.run(H.op_amax(dy, dy.numel(), am))
    for m in (4, 3):
        c = H.Conv(N, S, S, C, 0, Cout, math=m)
        wsb = c.wgrad_ws_bytes(); ws = torch.empty(wsb // 4 + 16, device="cuda")
        ms = t(H.op_conv_wgrad(c, x, None, dy, dw, ws, wsb, db=db, dy_amax=am if m == 4 else None))
        print(f"wgrad N{N} {S}x{S} {C}->{Cout} math {m}: {ms*1e3:.0f} us  {2.0*N*S*S*Cout*9*C/ms/1e9:.0f} TF", flush=True)
