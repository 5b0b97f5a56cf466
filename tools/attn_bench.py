"""Fused attention at the FFHQ-128 shapes (B=32): microseconds per forward / backward launch group.  PDAE_HIP_LIB selects a probe build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdae_amd import hip as H


def timed(op, n=20):
    for _ in range(3):
        H.run(op)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        H.run(op)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = []
for (N, T, C, heads) in ((32, 256, 384, 1), (32, 64, 512, 1), (32, 256, 256, 4)):
    qkv = torch.randn(N, T, 3 * C, device="cuda"); o = torch.empty(N, T, C, device="cuda"); lse = torch.empty(N * heads, T, device="cuda")
    do = torch.randn(N, T, C, device="cuda"); dq = torch.empty_like(qkv); ws = torch.empty(N * heads, T, device="cuda")
    f = timed(H.op_attn_fwd(qkv, N, T, C, heads, False, o, lse))
    b = timed(H.op_attn_bwd(qkv, o, lse, do, N, T, C, heads, False, dq, ws))
    out.append(f"T{T} C{C}x{heads}: fwd {f:6.1f} bwd {b:6.1f}")
print(os.environ.get("PDAE_HIP_LIB", "product").split("/")[-2] if os.environ.get("PDAE_HIP_LIB") else "product", " | ".join(out), flush=True)
