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
enc = FFHQEncoder(device=dev, latent_dim=512); dec = ShiftUNet(device=dev, latent_dim=512, **bench.load_workload()[1])
bench.randomize(enc, 1); bench.randomize(dec, 2); enc.train(); dec.set_train_mode()
gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, dev)
st = FusedRLStep(gd, enc, dec, copy.deepcopy(enc), copy.deepcopy(dec), 32, 128, 128)
x0 = torch.rand(32, 3, 128, 128, device=dev) * 2 - 1
st.step(x0); st.load_batch(x0)
durs = bench.profile_plan(st.plan, 0, st.n_bwd)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
gb = 0.0
names = {1: "fwd", 2: "dgrad", 3: "wgrad", 4: "gemm", 30: "fwd", 31: "fwd+sk"}
for k, d in enumerate(durs):
    op = st.plan.arr[k]; i = op.i
    if op.kind in (1, 2, 3, 30, 31):
        patch = (i[8] == 3 and i[10] == 1 and i[4] == 0 and i[3] % 32 == 0 and i[5] % 8 == 0 and i[6] % 16 == 0)
        key = (names[op.kind], f"N{i[0]} {i[1]}x{i[2]} {i[3]}+{i[4]}->{i[7]} k{i[8]} s{i[10]} up{i[12]}", "P" if patch else "-")
    elif op.kind == 4:
        key = ("gemm", f"M{i[2]} N{i[3]} K{i[4]} b{i[14]*i[15]}", "-")
    elif op.kind == 5:
        key = ("gnstat", f"N{i[2]} HW{i[3]} C{i[0]}+{i[1]}", "-"); gb = 4.0 * i[2] * i[3] * (i[0] + i[1])
    elif op.kind == 7:
        key = ("gnapply", f"N{i[2]} {i[3]}x{i[4]} C{i[0]}+{i[1]} act{i[5]} mode{i[6]} drop{op.f[0]:.1f}", "-"); gb = 8.0 * i[2] * i[3] * i[4] * (i[0] + i[1])
    elif op.kind == 8:
        key = ("gnbwd", f"N{i[2]} {i[3]}x{i[4]} C{i[0]}+{i[1]} mode{i[7]} add{int(bool(op.p[9]))} dx{int(bool(op.p[10]))}{int(bool(op.p[11]))}", "-")
        gb = 4.0 * i[2] * i[3] * i[4] * (i[0] + i[1]) * 5
    elif op.kind == 24:
        key = ("colsum", f"M{i[0]} C{i[1]}", "-"); gb = 4.0 * i[0] * i[1]
    else:
        key = (f"kind{op.kind}", "", "-")
    a = agg[key]; a[0] += 1; a[1] += d; a[2] += bench.op_flops(op) if (op.kind < 5 or op.kind in (30, 31)) else gb * 1e3; gb = 0.0
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(durs)
print(f"total {tot:.1f} ms")
for (kind, shape, p), (cnt, ms, fl) in rows[:int(os.environ.get("TOPN", "70"))]:
    print(f"{kind:7s} {p} {shape:44s} x{cnt:3d} {ms:7.2f} ms {100*ms/tot:5.1f}%  {fl/ms/1e9 if ms else 0:7.1f} TF")
cat = collections.defaultdict(lambda: [0, 0.0, 0.0])
for (kind, shape, p), (cnt, ms, fl) in agg.items():
    res = shape.split()[1] if kind in ("fwd", "dgrad", "wgrad") else ""
    k3 = ("k3" if " k3 " in shape else "k1") if kind in ("fwd", "dgrad", "wgrad") else ""
    c = cat[(kind, k3, res)]; c[0] += cnt; c[1] += ms; c[2] += fl
print("---- by category")
for (kind, k3, res), (cnt, ms, fl) in sorted(cat.items(), key=lambda kv: -kv[1][1]):
    print(f"{kind:8s} {k3:3s} {res:9s} x{cnt:4d} {ms:7.2f} ms {100*ms/tot:5.1f}%  {fl/ms/1e9 if ms else 0:7.1f}")
