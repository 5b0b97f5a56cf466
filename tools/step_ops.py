"""Per-op timing of one FFHQ-128 training step (B=32): aggregates by (kind, shape) to show where non-patch time goes."""
import sys, os, copy, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pdae_amd import hip as H
from pdae_amd.model.shift_unet import ShiftUNet
from pdae_amd.model.representation_learning.encoder import FFHQEncoder
from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
from pdae_amd.trainer.fused_step import FusedRLStep
dev = torch.device("cuda")
enc = FFHQEncoder(device=dev, latent_dim=512); dec = ShiftUNet(device=dev, latent_dim=512, **bench.FFHQ128)
bench.randomize(enc, 1); bench.randomize(dec, 2); enc.train(); dec.set_train_mode()
gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, dev)
st = FusedRLStep(gd, enc, dec, copy.deepcopy(enc), copy.deepcopy(dec), 32, 128, 128)
x0 = torch.rand(32, 3, 128, 128, device=dev) * 2 - 1
st.step(x0); st.load_batch(x0)
durs = bench.profile_plan(st.plan, 0, st.n_bwd)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
names = {1: "fwd", 2: "dgrad", 3: "wgrad", 4: "gemm"}
for k, d in enumerate(durs):
    op = st.plan.arr[k]; i = op.i
    if op.kind in (1, 2, 3):
        patch = (i[8] == 3 and i[10] == 1 and i[4] == 0 and i[3] % 32 == 0 and i[5] % 8 == 0 and i[6] % 16 == 0)
        key = (names[op.kind], f"N{i[0]} {i[1]}x{i[2]} {i[3]}+{i[4]}->{i[7]} k{i[8]} s{i[10]} up{i[12]}", "P" if patch else "-")
    elif op.kind == 4:
        key = ("gemm", f"M{i[2]} N{i[3]} K{i[4]} b{i[14]*i[15]}", "-")
    else:
        key = (f"kind{op.kind}", "", "-")
    a = agg[key]; a[0] += 1; a[1] += d; a[2] += bench.op_flops(op)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(durs)
print(f"total {tot:.1f} ms")
for (kind, shape, p), (cnt, ms, fl) in rows[:70]:
    print(f"{kind:7s} {p} {shape:44s} x{cnt:3d} {ms:7.2f} ms {100*ms/tot:5.1f}%  {fl/ms/1e9 if ms else 0:7.1f} TF")
