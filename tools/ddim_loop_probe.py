"""Where does a DDIM-100 loop's time go beyond the per-op sum of one denoising step?  (a) 100 back-to-back plan runs, (b) the sampler loop."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pdae_amd.model.shift_unet import ShiftUNet
from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
dev = torch.device("cuda")
B = int(os.environ.get("B", "100"))
dec = ShiftUNet(device=dev, latent_dim=512, **bench.load_workload()[1])
bench.randomize(dec, 2); dec.eval(); dec.set_eval_mode()
gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, dev)
p = dec.plan(B, 128, 128, False)
x = torch.randn(B, 3, 128, 128, device=dev); t = torch.full((B,), 500, device=dev, dtype=torch.long); z = torch.randn(B, 512, device=dev)
with torch.no_grad():
    dec(x, t, z)
    torch.cuda.synchronize()
    for n in (10, 100):
        t0 = time.perf_counter()
        for i in range(n):
            p.run(p.n_const if i else 0, p.n_fwd, prep=(i == 0))
        torch.cuda.synchronize()
        print(f"{n} plan runs: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per run", flush=True)
    t0 = time.perf_counter()
    for i in range(100):
        p.run(p.n_const if i else 0, p.n_fwd, prep=(i == 0))
        if i % 10 == 9:
            torch.cuda.synchronize(); print(f"  steps {i-9}..{i}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per run", flush=True); t0 = time.perf_counter()
    for style in ("ddim10", "ddim100"):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gd.representation_learning_ddim_sample(style, None, dec, None, x, z)
        torch.cuda.synchronize()
        n = int(style[4:])
        print(f"{style}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per step", flush=True)
