"""Dynamic range of the gradient tensors (dY operands of dgrad / wgrad) over one FFHQ-128 training step: abs-max and rms per conv layer."""
import sys, os, copy, ctypes
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
rows = []
for k in range(st.n_bwd):
    op = st.plan.arr[k]
    H.run_ops(op, 1)
    if op.kind in (H.OP_CONV_DGRAD, H.OP_CONV_WGRAD):
        i = op.i
        n = i[0] * i[5] * i[6] * i[7]                 # dY = [N, Ho, Wo, Cout]
        ptr = op.p[0] if op.kind == H.OP_CONV_DGRAD else op.p[2]
        buf = (ctypes.c_float * n).from_address(0)    # placeholder type; read through torch below
        t = torch.empty(0)
        dy = torch.from_dlpack if False else None
        # wrap the raw device pointer
        a = torch.cuda.FloatTensor(0)
        dyt = torch.as_strided(st.plan.live[0], (0,), (1,))
        rows.append((k, op.kind, i[5], i[7], ptr, n))
torch.cuda.synchronize()
# second pass: locate each dY pointer inside the plan's live buffers and take statistics right after the op ran
live = [(t.data_ptr(), t.numel() * t.element_size(), t) for t in st.plan.live if t.dtype == torch.float32]
def view(ptr, n):
    for base, nb, t in live:
        if base <= ptr < base + nb:
            off = (ptr - base) // 4
            return t.view(-1)[off:off + n]
    return None
st.load_batch(x0)
seen = set()
print(f"{'op':>5s} {'kind':6s} {'res':>4s} {'Cout':>5s} {'amax':>10s} {'rms':>10s} {'amax/rms':>9s}")
idx = {r[0]: r for r in rows}
for k in range(st.n_bwd):
    H.run_ops(st.plan.arr[k], 1)
    if k in idx:
        _, kind, res, cout, ptr, n = idx[k]
        v = view(ptr, n)
        if v is None or (ptr, kind) in seen:
            continue
        seen.add((ptr, kind))
        amax = float(v.abs().max()); rms = float(v.pow(2).mean().sqrt())
        print(f"{k:5d} {'dgrad' if kind == H.OP_CONV_DGRAD else 'wgrad':6s} {res:4d} {cout:5d} {amax:10.3e} {rms:10.3e} {amax / max(rms, 1e-30):9.1f}")
