"""conv3x3y vs the direct kernels on one small shape (MATH env: arithmetic mode): where the outputs differ -- by pixel parity, m-tile, channel tile, row;\nthe script that located the LDS store source hazard.  python tools/y_dbg.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdae_amd import hip as H
N, S, W, Cin, Cout = 2, 32, 16, 32, 128
x = torch.randn(N, S, W, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5; b = torch.randn(Cout, device="cuda")
outs = {}
for mode in ("0", "2"):
    H.set_knob("PDAE_W1", int(mode))
    c = H.Conv(N, S, W, Cin, 0, Cout, k=3, math=int(os.environ.get("MATH", "4")))
    wp = torch.empty(c.wprep_bytes(0, force=True) // 4, device="cuda"); H.run(H.op_conv_wprep(c, w, 0, wp))
    y = torch.full((N, S, W, Cout), float("nan"), device="cuda")
    H.run(H.op_conv_fwd(c, x, None, w, b, y, wp=wp)); torch.cuda.synchronize()
    outs[mode] = y.clone()
d = (outs["2"] - outs["0"]).abs()
bad = d > 1e-3
print("bad fraction", bad.float().mean().item(), "nan", torch.isnan(outs["2"]).float().mean().item())
print("by x parity", [bad[:, :, j::2].float().mean().item() for j in range(2)])
print("by x half (a2)", [bad[:, :, a * 8:(a + 1) * 8].float().mean().item() for a in range(2)])
print("by channel 32-tile", [bad[..., t * 32:(t + 1) * 32].float().mean().item() for t in range(4)])
print("by row mod 16", [round(bad[:, r::16].float().mean().item(), 2) for r in range(16)])
print("by image", [bad[n].float().mean().item() for n in range(N)])
idx = bad.nonzero()
import collections
print("rows%8:", sorted(set((idx[:, 1] % 8).tolist())), "x:", sorted(set(idx[:, 2].tolist())), "c%32:", sorted(set((idx[:, 3] % 32).tolist())))
print("values W1:", outs["2"][bad][:8].tolist(), "direct:", outs["0"][bad][:8].tolist())
o2, o0 = outs["2"], outs["0"]
for (n, yy, xx, cc) in idx[:6].tolist():
    print((n, yy, xx, cc), "W1 %.4f direct %.4f | direct x+1 %.4f x-1 %.4f | c-1 %.4f c+1 %.4f | y+-2 %.4f %.4f | bias %.4f | W1-direct %.4f" % (
        o2[n, yy, xx, cc], o0[n, yy, xx, cc], o0[n, yy, xx + 1, cc], o0[n, yy, xx - 1, cc], o0[n, yy, xx, cc - 1], o0[n, yy, xx, cc + 1],
        o0[n, yy - 2, xx, cc], o0[n, yy + 2, xx, cc], b[cc], o2[n, yy, xx, cc] - o0[n, yy, xx, cc]))
# is the bad value found anywhere in the direct output of the same image / channel tile?
n, yy, xx, cc = idx[0].tolist()
v = o2[n, yy, xx, cc]
m = ((o0[n] - v).abs() < 1e-4).nonzero()
print("bad value", float(v), "found in direct output at", m[:5].tolist())
m2 = ((o0[n] - b.view(1, 1, -1) - (v - b[cc])).abs() < 1e-4).nonzero()
print("bias-free match", m2[:5].tolist())
