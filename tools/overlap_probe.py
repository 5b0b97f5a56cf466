"""Can an HBM-bound kernel run beside the persistent conv3x3y kernel?  Main stream: n Winograd-form 3x3 convolutions (128^2 128->128, B=32);
side stream: m streaming passes (a GroupNorm apply of the library = 1 read + 1 write of a 268 MB tensor, and a torch add = 2 reads + 1 write).
Prints the time of each alone and of both launched together.  (tools/occupy_probe.py: kernels without LDS co-reside with conv3x3y.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PDAE_W1", "2")
import torch
from pdae_amd import hip as H
N, S, C = 32, 128, 128
x = torch.randn(N, S, S, C, device="cuda"); w = torch.randn(C, 3, 3, C, device="cuda") / (C * 9) ** 0.5; b = torch.randn(C, device="cuda")
y = torch.empty(N, S, S, C, device="cuda")
c = H.Conv(N, S, S, C, 0, C, k=3, math=4)
wp = torch.empty(c.wprep_bytes(0) // 4, device="cuda"); H.run(H.op_conv_wprep(c, w, 0, wp))
op = H.op_conv_fwd(c, x, None, w, b, y, wp=wp)
a1 = torch.randn(N, S, S, C, device="cuda"); a2 = torch.randn_like(a1); a3 = torch.empty_like(a1)
side = torch.cuda.Stream()
def convs(n):
    for _ in range(n): H.run(op)
def adds(m):
    for _ in range(m): torch.add(a1, a2, out=a3)
def wall(fn_main, fn_side):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if fn_side:
        with torch.cuda.stream(side): fn_side()
    if fn_main: fn_main()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
convs(20); adds(20); torch.cuda.synchronize()
for n, m in ((20, 20), (20, 40), (20, 10)):
    tc = min(wall(lambda: convs(n), None) for _ in range(3))
    ta = min(wall(None, lambda: adds(m)) for _ in range(3))
    tb = min(wall(lambda: convs(n), lambda: adds(m)) for _ in range(3))
    print(f"{n} convolutions alone {tc:.3f} ms | {m} adds (805 MB each) alone {ta:.3f} ms | together {tb:.3f} ms  (sum {tc + ta:.3f}, max {max(tc, ta):.3f})", flush=True)
