"""Back-to-back timing of the M=32 linear layers (200 launches in ONE pdae_run_ops call: no host launch gaps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdae_amd import hip as H
for (M, N, K) in [(32, 256, 512), (32, 512, 512), (32, 1024, 512), (32, 512, 1024), (32, 512, 4096)]:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda"); C = torch.empty(M, N, device="cuda")
    arr = H.ops_array([H.op_gemm(0, 1, M, N, K, A, K, B, K, C, N, bias=b)] * 200)
    H.run_ops(arr, 200); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); H.run_ops(arr, 200); e1.record(); torch.cuda.synchronize()
    print(f"M{M} N{N} K{K}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per launch", flush=True)
