"""1x1 convolution micro-benchmark: prepared-weight kernel vs generic, all math modes, ms and effective GB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdae_amd import hip as H
def t(op, n=10):
    H.run(op); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): H.run(op)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (N, S, C0, C1, Cout) in [(32, 128, 128, 128, 128), (32, 128, 128, 0, 128), (32, 64, 256, 128, 128), (32, 16, 384, 0, 1152), (32, 16, 384, 0, 384), (32, 8, 512, 0, 512)]:
    x0 = torch.randn(N, S, S, C0, device="cuda"); x1 = torch.randn(N, S, S, C1, device="cuda") if C1 else None
    w = torch.randn(Cout, 1, 1, C0 + C1, device="cuda") / (C0 + C1) ** 0.5
    b = torch.randn(Cout, device="cuda"); y = torch.empty(N, S, S, Cout, device="cuda")
    gb = 4.0 * N * S * S * (C0 + C1 + Cout) / 1e9; gf = 2.0 * N * S * S * (C0 + C1) * Cout / 1e9
    line = f"N{N} {S}x{S} {C0}+{C1}->{Cout} ({gf:.1f} GF, {gb*1e3:.0f} MB):"
    for m in (3, 1):
        c = H.Conv(N, S, S, C0, C1, Cout, k=1, math=m)
        wp = torch.empty(c.wprep_bytes(0) // 4, device="cuda"); H.run(H.op_conv_wprep(c, w, 0, wp))
        ms_f = t(H.op_conv_fwd(c, x0, x1, w, b, y, wp=wp)); ms_g = t(H.op_conv_fwd(c, x0, x1, w, b, y))
        line += f"  m{m}: fast {ms_f:.3f} ms ({gf/ms_f:.0f} TF, {gb/ms_f*1e3:.0f} GB/s) generic {ms_g:.3f} ms |"
    print(line, flush=True)
