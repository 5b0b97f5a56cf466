import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdae_amd import hip as H
def t(op, n=5):
    H.run(op); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): H.run(op)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
N, S = 32, 128
for Cout in (128, 256):
    for C in (32, 64, 128, 256, 512):
        x = torch.randn(N, S, S, C, device="cuda"); w = torch.randn(Cout, 3, 3, C, device="cuda") / (C * 9) ** 0.5
        b = torch.randn(Cout, device="cuda"); y = torch.empty(N, S, S, Cout, device="cuda")
        for m in (4, 3):
            c = H.Conv(N, S, S, C, 0, Cout, math=m)
            wp = torch.empty(c.wprep_bytes(0) // 4, device="cuda")
            t_prep = t(H.op_conv_wprep(c, w, 0, wp))
            ms = t(H.op_conv_fwd(c, x, None, w, b, y, wp=wp))
            ms_nb = t(H.op_conv_fwd(c, x, None, w, None, y, wp=wp))
            print(f"Cout={Cout} Cin={C:4d} math={m}: {ms:.3f} ms (no bias {ms_nb:.3f}, prep {t_prep:.3f})  {2.0*N*S*S*Cout*9*C/ms/1e9:.1f} TF", flush=True)
