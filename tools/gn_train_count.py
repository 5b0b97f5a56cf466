import sys, os, copy
sys.path.insert(0, os.getcwd())
import torch, bench
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
n = sum(1 for k in range(st.plan.n) if st.plan.arr[k].kind == H.OP_CONV_WGRAD and st.plan.arr[k].p[7])
ng = sum(1 for k in range(st.plan.n) if st.plan.arr[k].kind == H.OP_GN_APPLY)
print("wgrad with gn input:", n, "gn_apply ops:", ng, "plan ops", st.plan.n, "bytes", st.plan.bytes_alloc / 2**30)
