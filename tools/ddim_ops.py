"""Per-op timing of one DDIM denoising step of the FFHQ-128 ShiftUNet at the evaluator's batch (100): where forward-only time goes."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pdae_amd import hip as H
from pdae_amd.model.shift_unet import ShiftUNet
dev = torch.device("cuda")
B = int(os.environ.get("B", "100"))
dec = ShiftUNet(device=dev, latent_dim=512, **bench.load_workload()[1])
bench.randomize(dec, 2); dec.eval()
p = dec.plan(B, 128, 128, False)
x = torch.randn(B, 3, 128, 128, device=dev); t = torch.full((B,), 500, device=dev, dtype=torch.long); z = torch.randn(B, 512, device=dev)
dec(x, t, z)                                  # one forward through the plan: binds inputs, prepares weights
durs = bench.profile_plan(p, 0, len(p.recs))
agg = collections.defaultdict(lambda: [0, 0.0])
for k, d in enumerate(durs):
    op = p.arr[k]; i = op.i
    if op.kind in (1, 30, 31):
        key = (op.kind, f"{i[1]}x{i[2]} {i[3]}+{i[4]}->{i[7]} k{i[8]} s{i[10]} up{i[12]}")
    elif op.kind in (5, 7, 32):
        key = (op.kind, f"HW{i[3]}" if op.kind != 7 else f"{i[3]}x{i[4]} C{i[0]}+{i[1]}")
    else:
        key = (op.kind, "")
    agg[key][0] += 1; agg[key][1] += d
tot = sum(durs)
print(f"B={B} total {tot:.1f} ms per denoising step ({len(durs)} ops)")
bykind = collections.defaultdict(lambda: [0, 0.0])
for (kind, shape), (cnt, ms) in agg.items():
    bykind[kind][0] += cnt; bykind[kind][1] += ms
for kind, (cnt, ms) in sorted(bykind.items(), key=lambda kv: -kv[1][1]):
    print(f"kind {kind:3d} x{cnt:4d} {ms:7.2f} ms {100*ms/tot:5.1f}%")
for (kind, shape), (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOPN", "25"))]:
    print(f"  kind {kind:3d} {shape:40s} x{cnt:3d} {ms:7.2f} ms")
