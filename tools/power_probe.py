"""Is the 6-product patch kernel clock/power limited?  Same launch on random vs zero-filled operands (MI355X_MICROARCH.md, DVFS give-back:
zero-filled inputs clock higher at identical instruction streams)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
N, S, C, Cout = 32, 128, 256, 128
for m in (3, 1):
    for fill in ("randn", "zeros"):
        x = getattr(torch, fill)(N, S, S, C, device="cuda"); w = getattr(torch, fill)(Cout, 3, 3, C, device="cuda") * 0.02
        y = torch.empty(N, S, S, Cout, device="cuda")
        c = H.Conv(N, S, S, C, 0, Cout, math=m)
        wp = torch.empty(c.wprep_bytes(0) // 4, device="cuda"); H.run(H.op_conv_wprep(c, w, 0, wp))
        ms = t(H.op_conv_fwd(c, x, None, w, None, y, wp=wp))
        print(f"math {m} {fill:6s}: {ms:.3f} ms  {2.0*N*S*S*Cout*9*C/ms/1e9:.1f} TFLOP/s algorithmic  ({(6 if m == 3 else 1)*2.0*N*S*S*Cout*9*C/ms/1e9:.0f} TFLOP/s of bf16 MFMA issued)", flush=True)
