"""One rank of a REAL multi-process data-parallel run of FusedRLStep (launched by tests/test_ddp_gpu.py under torch.distributed.run).

Every rank builds the same networks (set_seed(0), base_trainer.py:27-28), takes its own slice of one seeded global batch and runs `--steps`
fused steps: HIP forward / backward issued in segments, a real collective per gradient bucket (torch.distributed all-reduce -- gloo when the
ranks share one GPU, nccl = RCCL when every rank has its own -- or, with --native 1, the library's own RCCL communicator through
pdae_allreduce_bucket), 1/world folded into Adam.  Rank 0 then checks that parameters, EMA copies and Adam moments are BIT-IDENTICAL on all
ranks (DistributedDataParallel's invariant, trainer/train_representation_learning.py:29,39) and writes them, with the per-step mean loss,
to --out for the launching test to compare against a single process on the concatenated batch.
"""
import argparse
import copy
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build(dev, batch, cfg, native, bucket_mb=1):
    from pdae_amd.utils import set_seed
    from pdae_amd.model.shift_unet import ShiftUNet
    from pdae_amd.model.representation_learning.encoder import CELEBA64Encoder
    from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    from pdae_amd.trainer.fused_step import FusedRLStep
    set_seed(0)
    enc = CELEBA64Encoder(device=dev, latent_dim=512)
    dec = ShiftUNet(device=dev, latent_dim=512, **cfg)
    with torch.no_grad():                          # zero-initialised heads would make the gradients vacuous
        for net in (enc, dec):
            g = torch.Generator(device="cpu").manual_seed(5)
            for p in net.parameters():
                if float(p.abs().max()) == 0.0:
                    p.copy_((torch.randn(p.shape, generator=g) * 0.05).to(dev))
    enc.train(); dec.set_train_mode()
    gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, dev)
    st = FusedRLStep(gd, enc, dec, copy.deepcopy(enc), copy.deepcopy(dec), batch, 64, 64, bucket_mb=bucket_mb, native_comm=native)
    return enc, dec, st


def data(n, dev):
    g = torch.Generator().manual_seed(9)
    return ((torch.rand(n, 3, 64, 64, generator=g) * 2 - 1).to(dev), torch.randint(0, 1000, (n,), generator=g).to(dev),
            torch.randn(n, 3, 64, 64, generator=g).to(dev))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--native", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--per_rank", type=int, default=2)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local if a.backend == "nccl" else 0)          # gloo: the ranks share GPU 0 (1-GPU boxes)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend=a.backend)
    from tests.golden import make_fixtures_cfg as C
    cfg = dict(C.CFG_SHIFT_64, dropout=0.0)
    enc, dec, st = build(dev, a.per_rank, cfg, bool(a.native))
    assert st.world == world and len(st.buckets) >= 3, (st.world, len(st.buckets))
    assert (st.ncomm is not None) == bool(a.native)
    x0, t, noise = data(world * a.per_rank, dev)
    sl = slice(rank * a.per_rank, (rank + 1) * a.per_rank)
    losses = []
    for _ in range(a.steps):
        loss = st.step(x0[sl], t=t[sl], noise=noise[sl]).detach().clone()
        dist.all_reduce(loss)                                                 # mean of the rank losses = loss of the global batch
        losses.append(float(loss.item()) / world)
    # the time marks bench.py reports for the data-parallel exchange (round 5): one more step with them armed -- after the state below has been
    # captured, so that the comparison against the single-process run still sees `--steps` steps
    torch.cuda.synchronize()
    state = {"dec": dec.flat_train, "enc": enc.flat_train, "ema_dec": st.ema_dec.flat_train, "ema_enc": st.ema_enc.flat_train,
             "m_dec": st.m[0], "v_dec": st.v[0], "m_enc": st.m[1], "v_enc": st.v[1]}
    same = True
    for k, v in state.items():
        ref = v.clone()
        dist.broadcast(ref, src=0)
        same = same and bool(torch.equal(ref, v))
    flag = torch.tensor([int(same)], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    saved = {k: v.cpu().clone() for k, v in state.items()}
    st.enable_comm_timing()
    st.step(x0[sl], t=t[sl], noise=noise[sl])
    tm = st.comm_timing_ms()
    st.comm_events = None
    if st.ncomm is None:
        assert tm is not None and len(tm["bucket_enqueue_to_complete"]) == len(st.buckets) == len(tm["bucket_complete_after_bwd"]), tm
        assert tm["backward"] > 0 and tm["exposed"] >= 0 and all(v >= 0 for v in tm["bucket_enqueue_to_complete"]), tm
    if rank == 0:
        torch.save({"identical": bool(flag.item()), "losses": losses, "guard": st.saturation(), "backend": a.backend, "native": a.native,
                    "state": saved, "comm_timing": tm}, os.path.join(a.out, "result.pt"))
    dist.barrier()
    if st.ncomm is not None:
        st.ncomm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
