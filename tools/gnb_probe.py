"""Upper bound for folding gn_bwd_reduce into the data gradient's epilogue (VERDICT r4 task 1a): the data gradient of a 3x3 convolution is the same
launch as a forward convolution on transposed weights, so its cost WITH an epilogue operand (the GroupNorm input x) and WITH per-channel-quad sums
(the cheapest conceivable stand-in for  sum dv, sum dv (x - mu)  -- the real thing adds the SiLU derivative: 2 transcendentals + ~10 VALU per element)
is what conv3x3y's existing instantiations measure:  <EX, ST> = (0,0) plain, (1,0) operand, (0,1) sums, (1,1) both.  Against it: the separate
gn_bwd_reduce pass over the same tensor (read x, read dA).   python tools/gnb_probe.py [N S Cin Cout]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdae_amd import hip as H
a = sys.argv[1:]
N, S, Cin, Cout = (int(v) for v in a[:4]) if len(a) >= 4 else (32, 128, 128, 256)
H.set_knob("PDAE_W1", 2)
x = torch.randn(N, S, S, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5
y = torch.empty(N, S, S, Cout, device="cuda"); r = torch.randn_like(y)
c = H.Conv(N, S, S, Cin, 0, Cout, k=3, math=4)
wp = torch.empty(c.wprep_bytes(0) // 4, device="cuda"); H.run(H.op_conv_wprep(c, w, 0, wp))
nb, tpi = H.conv_stats_bytes(c)
part = torch.empty(nb // 4, device="cuda")


def timeit(op):
    for _ in range(40): H.run(op)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): H.run(op)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    return best


t = {}
for ex in (0, 1):
    for st in (0, 1):
        t[(ex, st)] = timeit(H.op_conv_fwd(c, x, None, w, None, y, wp=wp, res=r if ex else None, res_mode=ex, stats=part if st else None))
# the pass it would replace: gn_bwd on (x = r, dA = y) with C = Cout -- reduce + finalize + apply; the reduce is 2 of its 5 tensor passes
G = 32
coef = torch.zeros(3, N, Cout, device="cuda"); coef[1] = 1.0
rstd = torch.ones(N * G, device="cuda"); gamma = torch.ones(Cout, device="cuda"); beta = torch.zeros(Cout, device="cuda")
dx = torch.empty_like(y); ws = torch.empty(H.gn_ws_bytes(N, Cout) // 4 + 64, device="cuda")
tb = timeit(H.op_gn_bwd(r, Cout, None, 0, N, S, S, G, coef, rstd, gamma, beta, None, None, y, 1, 0, ws, dx0=dx))
print(f"N{N} {S}x{S} {Cin}->{Cout}: plain {t[(0,0)]:.4f}  +operand {t[(1,0)]:.4f}  +sums {t[(0,1)]:.4f}  +both {t[(1,1)]:.4f} ms | gn_bwd (reduce+finalize+apply, no dropout) {tb:.4f} ms"
      f" => the reduce pass is ~{tb * 0.4:.4f}; operand+sums alone cost {t[(1,1)] - t[(0,0)]:.4f}")
