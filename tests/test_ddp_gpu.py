"""GPU: the data-parallel training step of FusedRLStep on the real HIP kernels, world_size 2 emulated in ONE process.

Two ranks cannot share one device under RCCL, so rank 1's backward is run first and its flat gradients are kept; rank 0's
step then runs with `torch.distributed.all_reduce` replaced by a stand-in that adds rank 1's slice of the same bucket (as a
stream-ordered device op, exactly where the RCCL all-reduce would be enqueued).  Everything else is the production path:
forward/backward issued in segments, one all-reduce(sum) per bucket as soon as its gradients are final, 1/world folded
into Adam.  The result must equal a single process that sees the concatenated batch (gradient of the global mean loss --
DistributedDataParallel semantics of trainer/train_representation_learning.py:29,39).  An all-reduce issued before its
bucket is final, a bucket missed or reduced twice, or a wrong 1/world all break the equality."""
import copy

import pytest
import torch

from tests.golden import make_fixtures_cfg as C

pytestmark = pytest.mark.gpu
CFG = dict(C.CFG_SHIFT_64, dropout=0.0)           # dropout masks are per-rank streams: switch them off for the equivalence


def _build(batch):
    from pdae_amd.utils import set_seed
    from pdae_amd.model.shift_unet import ShiftUNet
    from pdae_amd.model.representation_learning.encoder import CELEBA64Encoder
    from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    from pdae_amd.trainer.fused_step import FusedRLStep
    set_seed(0)
    dev = torch.device("cuda", 0)
    enc = CELEBA64Encoder(device=dev, latent_dim=512)
    dec = ShiftUNet(device=dev, latent_dim=512, **CFG)
    with torch.no_grad():                          # zero-initialised heads would make the gradients vacuous
        for net in (enc, dec):
            g = torch.Generator(device="cpu").manual_seed(5)
            for p in net.parameters():
                if float(p.abs().max()) == 0.0:
                    p.copy_((torch.randn(p.shape, generator=g) * 0.05).to(dev))
    enc.train(); dec.set_train_mode()
    gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, dev)
    st = FusedRLStep(gd, enc, dec, copy.deepcopy(enc), copy.deepcopy(dec), batch, 64, 64, bucket_mb=1)
    return enc, dec, st


def _data(n):
    g = torch.Generator().manual_seed(9)
    return (torch.rand(n, 3, 64, 64, generator=g) * 2 - 1).cuda(), torch.randint(0, 1000, (n,), generator=g).cuda(), torch.randn(n, 3, 64, 64, generator=g).cuda()


class _Work:
    def wait(self):
        return True


def test_two_rank_step_equals_single_process_on_concatenated_batch(monkeypatch):
    from pdae_amd.trainer import fused_step as FS
    x0, t, noise = _data(4)
    enc0, dec0, st0 = _build(2)                    # "rank 0"
    enc1, dec1, st1 = _build(2)                    # "rank 1": same initial weights (set_seed(0), base_trainer.py:27-28)
    assert torch.equal(dec0.flat_train, dec1.flat_train) and len(st0.buckets) >= 3
    st0.world = 2
    calls = []

    def fake_all_reduce(view, op=None, group=None, async_op=False):
        if view.dtype == torch.int32:              # the fp16-window guard word travels with the last bucket (MAX): nothing to add here
            return _Work()
        for mine, other in ((dec0.flat_grad, dec1.flat_grad), (enc0.flat_grad, enc1.flat_grad)):
            off = (view.data_ptr() - mine.data_ptr()) // 4
            if 0 <= off and off + view.numel() <= mine.numel() and view.data_ptr() >= mine.data_ptr():
                view += other[off:off + view.numel()]
                calls.append((id(mine), off, view.numel()))
                return _Work()
        raise AssertionError("all_reduce on a tensor that is not a slice of a flat gradient buffer")

    monkeypatch.setattr(FS.dist, "all_reduce", fake_all_reduce)
    losses, grads = [], []
    for _ in range(2):
        # rank 1: forward + backward only (its optimizer step is irrelevant here: weights are re-synchronised below)
        st1.load_batch(x0[2:], t[2:], noise[2:])
        st1.plan.run(0, st1.n_bwd)
        l1 = float(st1.loss.item())
        l0 = float(st0.step(x0[:2], t=t[:2], noise=noise[:2]).item())
        losses.append(0.5 * (l0 + l1))
        grads.append((0.5 * dec0.flat_grad.double(), 0.5 * enc0.flat_grad.double()))         # reduced sum x 1/world
        dec1.flat_train.copy_(dec0.flat_train); enc1.flat_train.copy_(enc0.flat_train)      # what rank 1 would hold after its own step
    # every gradient element was reduced exactly once per step
    for buf in (dec0.flat_grad, enc0.flat_grad):
        spans = sorted((o, n) for i, o, n in calls[:len(calls) // 2] if i == id(buf))
        assert spans[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:])) and spans[-1][0] + spans[-1][1] == buf.numel()
    monkeypatch.undo()
    # single process, batch 4 = the two rank batches concatenated
    enc, dec, st = _build(4)
    ref_losses = []
    for k in range(2):
        ref_losses.append(float(st.step(x0, t=t, noise=noise).item()))
        assert abs(losses[k] - ref_losses[k]) < 1e-5 * abs(ref_losses[k])
        for got, ref in zip(grads[k], (dec.flat_grad.double(), enc.flat_grad.double())):
            # step 0 starts from identical weights: reduction-order noise only.  Step 1 starts from weights that differ by Adam's
            # amplification of that noise on near-zero gradients (|g| ~ eps), hence the looser bound.
            assert float((got - ref).norm() / ref.norm()) < (1e-5 if k == 0 else 2e-3)
    # weights: Adam moves each weight by ~lr = 1e-4 per step in the direction sign(g), so elements whose gradient is at rounding-noise
    # level may differ by a fraction of a step; everything else must agree to 2% of one step
    for got, ref in ((dec0.flat_train, dec.flat_train), (enc0.flat_train, enc.flat_train), (st0.ema_dec.flat_train, st.ema_dec.flat_train)):
        diff = (got - ref).abs()
        assert float(diff.max()) < 1e-4 and float((diff > 2e-6).float().mean()) < 1e-3 and float(diff.mean()) < 1e-7


def test_native_rccl_communicator_single_rank():
    """The library's own RCCL communicator (pdae_comm_init / pdae_allreduce_bucket, RCCL dlopen'ed) driven through the fused step's bucket
    path on a side stream: with one rank the all-reduce is the identity, so the step must equal the plain step bit for bit -- what this pins
    is symbol binding, communicator creation, stream / event ordering and that every bucket goes through the C ABI."""
    from pdae_amd.trainer import fused_step as FS
    x0, t, noise = _data(2)
    enc_a, dec_a, st_a = _build(2)
    set_calls = []
    import pdae_amd.comm as comm_mod
    orig = comm_mod.NativeComm.all_reduce

    def counting(self, view, op="sum"):
        set_calls.append((view.data_ptr(), view.numel(), op))
        return orig(self, view, op)

    comm_mod.NativeComm.all_reduce = counting
    try:
        enc_b, dec_b = _build(2)[:2]
        from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
        st_b = FS.FusedRLStep(GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, dec_b.device), enc_b, dec_b, copy.deepcopy(enc_b),
                              copy.deepcopy(dec_b), 2, 64, 64, bucket_mb=1, native_comm=True)
        assert st_b.ncomm is not None and st_b.ncomm.world == 1
        for _ in range(2):
            la = float(st_a.step(x0, t=t, noise=noise).item())
            lb = float(st_b.step(x0, t=t, noise=noise).item())
            assert la == lb
        torch.cuda.synchronize()
    finally:
        comm_mod.NativeComm.all_reduce = orig
    assert torch.equal(dec_a.flat_train, dec_b.flat_train) and torch.equal(enc_a.flat_train, enc_b.flat_train)
    floats = [c for c in set_calls if c[2] == "sum"]
    assert len(floats) == 2 * len(st_b.buckets) and sum(n for _, n, _ in floats) == 2 * (dec_b.flat_grad.numel() + enc_b.flat_grad.numel())
    st_b.ncomm.close()


# ---------------------------------------------------------------------------------------------------------------------------------
# REAL multi-process runs: HIP kernels and a real collective together (tests/ddp_worker.py under torch.distributed.run)
# ---------------------------------------------------------------------------------------------------------------------------------
def _launch_ranks(tmp_path, backend, native, steps=3, per_rank=2, world=2):
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / f"{backend}_{native}"
    out.mkdir()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("PDAE_NATIVE_RCCL", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "tests", "ddp_worker.py"), "--backend", backend, "--native", str(int(native)), "--steps", str(steps),
           "--per_rank", str(per_rank), "--out", str(out)]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return torch.load(out / "result.pt", map_location="cpu")


def _single_process_reference(steps, n):
    x0, t, noise = _data(n)
    enc, dec, st = _build(n)
    losses = [float(st.step(x0, t=t, noise=noise).item()) for _ in range(steps)]
    torch.cuda.synchronize()
    return losses, {"dec": dec.flat_train.cpu(), "enc": enc.flat_train.cpu(), "ema_dec": st.ema_dec.flat_train.cpu()}


def _check_against_single_process(res, steps=3, n=4):
    assert res["identical"], "parameters / EMA / Adam moments differ between ranks after the all-reduced steps"
    assert res["guard"][0] == 0
    ref_losses, ref = _single_process_reference(steps, n)
    for a, b in zip(res["losses"], ref_losses):
        assert abs(a - b) < 2e-4 * abs(b), (res["losses"], ref_losses)
    for k, v in ref.items():                         # see the bounds of the emulated test above: Adam amplifies rounding noise on |g| ~ eps
        diff = (res["state"][k] - v).abs()
        assert float(diff.max()) < 3e-4 and float((diff > 2e-6).float().mean()) < 5e-3 and float(diff.mean()) < 2e-7, (k, float(diff.max()), float(diff.mean()))


def test_two_processes_one_gpu_real_collective_bucketed_step(tmp_path):
    """Two OS processes share GPU 0; the gradient buckets go through torch.distributed's gloo all-reduce (async, one per bucket, issued
    between the backward segments) -- the production code path of FusedRLStep.backward_with_allreduce with a real collective, on a 1-GPU
    box.  Ranks must end bit-identical and equal to one process on the concatenated batch."""
    res = _launch_ranks(tmp_path, "gloo", native=False)
    _check_against_single_process(res)
    # round 5: the per-bucket time marks bench.py reports (CUDA events around every bucket's enqueue and completion) were taken on one more step
    tm = res["comm_timing"]
    assert tm is not None and len(tm["bucket_enqueue_to_complete"]) >= 3 and tm["backward"] > 0 and tm["exposed"] >= 0


def test_four_processes_one_gpu_real_collective_bucketed_step(tmp_path):
    """World size 4 (bucket boundaries, the 1 / world folded into Adam and the saturation-word MAX with world != 2): four OS
    processes share GPU 0, one image each, gloo all-reduce per bucket.  Ranks bit-identical; equal to one process on the four-image batch."""
    res = _launch_ranks(tmp_path, "gloo", native=False, steps=2, per_rank=1, world=4)
    _check_against_single_process(res, steps=2, n=4)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL places one rank per device)")
@pytest.mark.parametrize("native", [False, True], ids=["torch_nccl", "native_rccl"])
def test_two_gpus_rccl_both_exchange_backends(tmp_path, native):
    """Two ranks on two GPUs over RCCL: ProcessGroupNCCL (default) and the library's own communicator (pdae_allreduce_bucket)."""
    _check_against_single_process(_launch_ranks(tmp_path, "nccl", native=native))
