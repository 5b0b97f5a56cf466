"""What a few busy CUs cost the persistent conv3x3y launch (the N > 1 risk): k single-wave spin kernels (torch.cuda._sleep) on
k side streams hold k SIMDs while one 128^2 128->128 Winograd-form convolution runs on the main stream.  conv3x3y needs every register of its four
SIMDs, so a CU with a spinner on it cannot take a workgroup; tiles are assigned statically."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PDAE_W1", "2")
import torch
from pdae_amd import hip as H
N, S, Cin, Cout = 32, 128, 128, 128
x = torch.randn(N, S, S, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5; b = torch.randn(Cout, device="cuda")
y = torch.empty(N, S, S, Cout, device="cuda")
c = H.Conv(N, S, S, Cin, 0, Cout, k=3, math=4)
wp = torch.empty(c.wprep_bytes(0) // 4, device="cuda"); H.run(H.op_conv_wprep(c, w, 0, wp))
op = H.op_conv_fwd(c, x, None, w, b, y, wp=wp)
for _ in range(40): H.run(op)
torch.cuda.synchronize()
def timed(k, cycles=20_000_000):
    streams = [torch.cuda.Stream() for _ in range(k)]
    torch.cuda.synchronize()
    for s_ in streams:
        with torch.cuda.stream(s_): torch.cuda._sleep(cycles)          # ~10 ms of spinning, one wave each
    time.sleep(0.002)                                                   # let them become resident
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): H.run(op)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
for k in (0, 1, 4, 16, 32, 0):
    print(f"{k:3d} busy SIMDs: {timed(k):.4f} ms per launch", flush=True)
