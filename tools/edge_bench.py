"""The 3-channel edge layers at the bench shapes: time per launch with PDAE_EDGE=1 / 0 (same process: pdae_set_knob) and the
error of both against an fp64 convolution.  Usage: python tools/edge_bench.py [N] [size] [C]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from pdae_amd import hip as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
C = int(sys.argv[3]) if len(sys.argv) > 3 else 128
g = torch.Generator().manual_seed(1)


def rn(*s, scale=1.0):
    return (torch.randn(*s, generator=g) * scale)


def timed(op, n=20):
    for _ in range(3):
        H.run(op)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        H.run(op)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def case(name, Cin, Cout, kinds):
    x = rn(N, Cin, S, S); w = rn(Cout, Cin, 3, 3, scale=(Cin * 9) ** -0.5); b = rn(Cout, scale=0.3); dy = rn(N, Cout, S, S)
    c = H.Conv(N, S, S, Cin, 0, Cout, k=3, math=4)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda(); wd = w.permute(0, 2, 3, 1).contiguous().cuda(); bd = b.cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    y = torch.empty(N, S, S, Cout, device="cuda"); dx = torch.empty_like(xd); dw = torch.empty_like(wd)
    wsb = c.wgrad_ws_bytes(); wsp = torch.empty(wsb // 4 + 16, device="cuda")
    nref = min(N, 2)
    xr = x[:nref].double().cuda().requires_grad_(True); wr = w.double().cuda().requires_grad_(True)
    yr = F.conv2d(xr, wr, b.double().cuda(), padding=1)
    yr.backward(dy[:nref].double().cuda())
    ops = {"fwd": (H.op_conv_fwd(c, xd, None, wd, bd, y), lambda: rel(y[:nref].permute(0, 3, 1, 2), yr.detach())),
           "dgrad": (H.op_conv_dgrad(c, dyd, wd, dx), lambda: rel(dx[:nref].permute(0, 3, 1, 2), xr.grad)),
           "wgrad": (H.op_conv_wgrad(c, xd, None, dyd, dw, wsp, wsb), None)}
    for k in kinds:
        op, err = ops[k]
        out = []
        for sw in ("1", "0"):
            H.set_knob("PDAE_EDGE", int(sw))
            us = timed(op)
            out.append(f"EDGE={sw}: {us:7.1f} us" + (f" err {err():.2e}" if err else ""))
        print(f"{name:28s} {k:6s} " + "   ".join(out), flush=True)
    H.set_knob("PDAE_EDGE", 1)


case(f"head {C}->3 @{S} N={N}", C, 3, ["fwd", "dgrad", "wgrad"])
case(f"stem 3->{C} @{S} N={N}", 3, C, ["fwd"])
