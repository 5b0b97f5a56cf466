import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math
import torch.nn.functional as F
from pdae_amd import hip as H
torch.manual_seed(0)
N, S, C, Cout = 4, 64, 256, 128
x = torch.randn(N, C, S, S); w = torch.randn(Cout, C, 3, 3) / math.sqrt(C * 9); b = torch.randn(Cout) * 0.1
ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
ref32 = F.conv2d(x, w, b, padding=1).double()
print("torch fp32 cpu rel err vs fp64:", float((ref32 - ref).norm() / ref.norm()))
xd = x.permute(0, 2, 3, 1).contiguous().cuda(); wd = w.permute(0, 2, 3, 1).contiguous().cuda(); bd = b.cuda()
for scale in (1.0, 1e-3, 30.0):
    for m in (0, 3, 4, 2):
        c = H.Conv(N, S, S, C, 0, Cout, math=m)
        y = torch.empty(N, S, S, Cout, device="cuda")
        xs = xd * scale
        nb = c.wprep_bytes(0, force=True)
        wp = None
        if nb:
            wp = torch.empty(nb // 4, device="cuda"); H.run(H.op_conv_wprep(c, wd, 0, wp))
        H.run(H.op_conv_fwd(c, xs, None, wd, None, y, wp=wp))
        r = F.conv2d(x.double() * scale, w.double(), None, padding=1)
        e = float((y.permute(0, 3, 1, 2).double().cpu() - r).norm() / r.norm())
        print(f"scale {scale:g} math {m}: rel err {e:.3e}")
