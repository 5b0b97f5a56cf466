"""Worker of tests/test_side_stream_gpu.py::test_second_stream_stress_...: the FFHQ-128 representation-learning step at its shipped topology, B = 32,
dropout ON, with the second stream at the HIGHEST priority (the side stream's priority is fixed when the library first creates it: this must be a
fresh process with PDAE_SIDE_STREAM=3 in its environment), a 64 MB parking budget (joins all over the backward and the shift branch) and -- second
variant -- the data-parallel driver's segmented runs (native communicator at world 1: every bucket boundary is a pdae_run_ops call).  Each variant
runs three optimizer steps; the same steps with the flag ignored (PDAE_SIDE_STREAM=0: one stream, in order) must give BIT-IDENTICAL parameters,
EMA copies and Adam moments.  Prints one JSON line."""
import copy
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                                    # noqa: E402


def main():
    from oracle import pdae_oracle as O
    from pdae_amd import hip as H
    from pdae_amd.utils import load_yaml
    from pdae_amd.model.representation_learning import decoder as decoder_module, encoder as encoder_module
    from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    from pdae_amd.trainer.fused_step import FusedRLStep
    assert os.environ.get("PDAE_SIDE_STREAM") == "3" and os.environ.get("PDAE_SIDE_BUDGET_MB") == "64"
    dev = torch.device("cuda", 0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    c = load_yaml(os.path.join(ROOT, "config/ffhq_representation_learning.yml"))
    dcfg = load_yaml(os.path.join(ROOT, c["trained_ddpm_config"]))["denoise_fn_config"]
    assert float(dcfg["dropout"]) > 0
    ename, latent = c["encoder_config"]["model"], c["encoder_config"]["latent_dim"]
    enc_sd = O.synth_state_dict(O.encoder_param_shapes(ename, latent), 1)
    dec_sd = O.synth_state_dict(O.unet_param_shapes(dcfg, shift=True, latent_dim=latent), 2)
    gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, dev)
    g = torch.Generator().manual_seed(0)
    x0 = (torch.rand(B, 3, 128, 128, generator=g) * 2 - 1).to(dev)
    ts = [torch.randint(0, 1000, (B,), generator=g).to(dev) for _ in range(3)]
    ns = [torch.randn(B, 3, 128, 128, generator=g).to(dev) for _ in range(3)]

    def run(side, native):
        H.set_knob("PDAE_SIDE_STREAM", side)
        enc = getattr(encoder_module, ename)(device=dev, **c["encoder_config"])
        dec = getattr(decoder_module, c["decoder_config"]["model"])(device=dev, latent_dim=c["decoder_config"]["latent_dim"], **dcfg)
        enc.load_state_dict(enc_sd); dec.load_state_dict(dec_sd)
        enc.train(); dec.set_train_mode()
        ee, ed = copy.deepcopy(enc), copy.deepcopy(dec)
        st = FusedRLStep(gd, enc, dec, ee, ed, B, 128, 128, lr=1e-4, ema_decay=0.9999, native_comm=native, bucket_mb=8)
        random.seed(1234)                                                       # the dropout seeds come from the host RNG
        losses = [float(st.step(x0, t=ts[k], noise=ns[k])) for k in range(3)]
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for tns in [enc.flat_train, dec.flat_train, ee.flat_train, ed.flat_train] + list(st.m) + list(st.v):
            h.update(tns.detach().cpu().numpy().tobytes())
        info = dict(side_ops=st.plan.n_side, joins=sum(1 for o in st.plan.recs if o.kind == H.OP_JOIN), check=getattr(st.plan, "check", None),
                    buckets=len(st.buckets), sat=st.saturation()[0])
        del st
        torch.cuda.empty_cache()
        return losses, h.hexdigest(), info
    out = {}
    for native in (False, True):
        l1, h1, i1 = run(3, native)
        l0, h0, i0 = run(0, native)
        out["native" if native else "plain"] = dict(identical=(h1 == h0 and l1 == l0), losses=l1, info=i1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
