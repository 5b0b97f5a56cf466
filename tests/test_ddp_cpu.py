"""CPU, world_size 2, gloo: the data-parallel exchange of FusedRLStep (flat-gradient buckets in backward order,
all-reduce(sum), 1/world folded into Adam) and the reference's seeding / dispatch semantics.
The kernels themselves cannot run here (HIP only): the plan is built on CPU tensors and `run` is replaced by a
recorder, so what is exercised is exactly the code path that issues collectives on the GPU box."""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.golden import make_fixtures_cfg as C


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pdae_amd.utils import set_seed
        from pdae_amd import hip as H
        from pdae_amd.model.shift_unet import ShiftUNet
        from pdae_amd.model.representation_learning.encoder import CELEBA64Encoder
        from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
        from pdae_amd.trainer.fused_step import FusedRLStep
        set_seed(0)                                   # same init on all ranks (base_trainer.py:27-28)
        enc = CELEBA64Encoder(device="cpu", latent_dim=512)
        dec = ShiftUNet(device="cpu", latent_dim=512, **C.CFG_SHIFT_64)
        dec.set_train_mode()
        w0 = [dec.flat_train.clone(), enc.flat_train.clone()]
        gathered = [torch.zeros_like(w0[0]) for _ in range(world)]
        dist.all_gather(gathered, w0[0])
        assert all(torch.equal(g, w0[0]) for g in gathered), "ranks must start from identical weights"
        set_seed(rank)                                # different data/noise streams afterwards (base_trainer.py:50-52)
        r = torch.rand(4)
        allr = [torch.zeros(4) for _ in range(world)]
        dist.all_gather(allr, r)
        assert not torch.equal(allr[0], allr[1])
        gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, torch.device("cpu"))
        st = FusedRLStep(gd, enc, dec, copy.deepcopy(enc), copy.deepcopy(dec), 2, 64, 64, bucket_mb=1)
        assert st.world == world
        # buckets: cover both flat gradient buffers exactly once, each in backward (descending offset) order; the decoder's come first, the
        # encoder's final Linear (more than half of its bytes, final first) is released before the end of the backward
        def inside(v, buf):
            return buf.data_ptr() <= v.data_ptr() < buf.data_ptr() + buf.numel() * 4
        dec_b = [v for _, v in st.buckets if inside(v, dec.flat_grad)]
        enc_b = [(i, v) for i, v in st.buckets if inside(v, enc.flat_grad)]
        assert len(dec_b) + len(enc_b) == len(st.buckets)
        assert sum(v.numel() for v in dec_b) == dec.flat_grad.numel() and len(dec_b) >= 3
        assert sum(v.numel() for _, v in enc_b) == enc.flat_grad.numel() and len(enc_b) >= 2
        for ptrs in ([v.data_ptr() for v in dec_b], [v.data_ptr() for _, v in enc_b]):
            assert ptrs == sorted(ptrs, reverse=True)
        assert enc_b[-1][1].data_ptr() == enc.flat_grad.data_ptr() and enc_b[0][0] < st.n_bwd
        ops_idx = [i for i, _ in st.buckets]
        # (round 6: the backward segment ends with the explicit join in front of the optimizer ops -- the last bucket is final right before it)
        assert ops_idx == sorted(ops_idx) and st.n_fwd < ops_idx[0] and ops_idx[-1] == st.n_bwd - 1 and st.plan.recs[st.n_bwd - 1].kind == H.OP_JOIN
        # fake "backward": every rank writes rank-dependent gradients, segments are recorded instead of launched
        dec.flat_grad.copy_(torch.arange(dec.flat_grad.numel(), dtype=torch.float32) % 7 + rank)
        enc.flat_grad.fill_(float(rank + 1))
        segs = []
        st.backward_with_allreduce(lambda a, b: segs.append((a, b)))
        assert segs[0][0] == 0 and segs[-1][1] == st.n_bwd and all(segs[i][1] == segs[i + 1][0] for i in range(len(segs) - 1))
        exp = sum((torch.arange(dec.flat_grad.numel(), dtype=torch.float32) % 7 + rk) for rk in range(world))
        assert torch.equal(dec.flat_grad, exp)
        assert torch.equal(enc.flat_grad, torch.full_like(enc.flat_grad, float(sum(range(1, world + 1)))))
        st._patch_adam()
        assert abs(st.plan.arr[st.adam_idx[0][0]].f[7] - 1.0 / world) < 1e-12      # mean = sum * (1/world) inside Adam
        q.put((rank, "ok"))
    except Exception as e:                        # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_exchange_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_dispatch_num_samples_matches_reference_table():
    """trainer/base_trainer.py:143-153 (remainder to the last rank)."""
    from pdae_amd.utils import dispatch_num_samples_for_process as d
    assert [d(36, 8, r) for r in range(8)] == [4] * 7 + [8]
    assert [d(10, 3, r) for r in range(3)] == [3, 3, 4]
    assert d(5, 5, 4) == 1 and d(7, 1, 0) == 7
    with pytest.raises(AssertionError):
        d(3, 4, 0)


def _retry_worker(q, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from pdae_amd.model.shift_unet import ShiftUNet
        from pdae_amd.model.representation_learning.encoder import CELEBA64Encoder
        from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
        from pdae_amd.trainer.fused_step import FusedRLStep
        gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, torch.device("cpu"))

        def make(num_iterations):
            enc = CELEBA64Encoder(device="cpu", latent_dim=512)
            dec = ShiftUNet(device="cpu", latent_dim=512, **C.CFG_SHIFT_64)
            dec.set_train_mode()
            st = FusedRLStep(gd, enc, dec, copy.deepcopy(enc), copy.deepcopy(dec), 2, 64, 64, bucket_mb=1, num_iterations=num_iterations)
            st.world = 2                                   # take the collective path (the group itself has one rank: all-reduce = identity)
            runs = []
            st.plan.run = lambda first=0, last=None, stream=None, prep=True: runs.append((first, last))
            boom = {"n": 0}
            orig = st._bucketed

            def failing(run):
                boom["n"] += 1
                if boom["n"] == 1:
                    run(0, 5)                                  # part of the backward has been issued when the collective raises
                    raise RuntimeError("simulated RCCL enqueue error")
                return orig(run)
            st._bucketed = failing
            return st, runs
        # num_iterations == 1: the plan overwrites its gradients -> the micro-batch is re-run through the post-backward fallback, nothing raised
        st, runs = make(1)
        st._run_micro()
        assert st._comm_fallback and st.comm_retries == 1 and st.step_count == 1
        assert (0, st.n_bwd) in runs and runs[-1] == (st.n_bwd, st.plan.n)          # whole forward + backward again, then the optimizer ops
        st._run_micro()                                                              # later steps stay on the fallback, no further retries
        assert st.comm_retries == 1 and st.step_count == 2
        # accumulating plans cannot be repaired: the error reaches the trainer
        st2, _ = make(2)
        st2._run_micro()
        try:
            st2._run_micro()
            q.put("accumulating plan did not raise")
            return
        except RuntimeError as e:
            assert "simulated" in str(e)
        q.put("ok")
    except Exception:                             # noqa: BLE001
        import traceback
        q.put(traceback.format_exc())
    finally:
        dist.destroy_process_group()


def test_failed_bucketed_exchange_is_retried_through_the_fallback_when_the_plan_overwrites_its_gradients():
    """backward_with_allreduce re-raises after switching to the fall-back; _run_micro re-runs the micro-batch through it when
    num_iterations == 1 (gradients are overwritten, same dropout seed) and lets the error through for accumulating plans."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_retry_worker, args=(q, _free_port()))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=60)
    assert res == "ok", res


def _two_rank_retry_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), PDAE_COMM_TIMEOUT_S="20")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        from pdae_amd.utils import set_seed
        from pdae_amd import hip as H
        from pdae_amd.model.shift_unet import ShiftUNet
        from pdae_amd.model.representation_learning.encoder import CELEBA64Encoder
        from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
        from pdae_amd.trainer.fused_step import FusedRLStep
        set_seed(0)
        gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, torch.device("cpu"))
        enc = CELEBA64Encoder(device="cpu", latent_dim=512)
        dec = ShiftUNet(device="cpu", latent_dim=512, **C.CFG_SHIFT_64)
        dec.set_train_mode()
        st = FusedRLStep(gd, enc, dec, copy.deepcopy(enc), copy.deepcopy(dec), 2, 64, 64, bucket_mb=1)
        runs = []

        def fake_run(first=0, last=None, stream=None, prep=True):       # the "backward": rank-dependent gradients, written by the segment that ends the backward
            runs.append((first, last))
            if last == st.n_bwd:
                dec.flat_grad.copy_(torch.arange(dec.flat_grad.numel(), dtype=torch.float32) % 5 + rank)
                enc.flat_grad.fill_(float(rank + 1))
        st.plan.run = fake_run
        calls = {"n": 0}

        def failing(run):
            calls["n"] += 1
            run(0, 5)
            if rank == 1:
                raise RuntimeError("simulated RCCL enqueue error on rank 1")                 # fails at once, mid-bucket
            time.sleep(1.0)                                                                  # its peer: stuck in the collective until the group's timeout
            raise RuntimeError("simulated collective timeout on rank 0")
        st._bucketed = failing
        t0 = time.time()
        st._run_micro()
        # both ranks met in the store BEFORE either issued a fall-back collective (rank 1 waited for rank 0's "timeout"), then re-ran the step together
        assert st._comm_fallback and st.comm_retries == 1 and st.step_count == 1 and calls["n"] == 1
        assert time.time() - t0 >= (0.9 if rank == 1 else 0.0)
        exp = sum((torch.arange(dec.flat_grad.numel(), dtype=torch.float32) % 5 + rk) for rk in range(world))
        assert torch.equal(dec.flat_grad, exp) and torch.equal(enc.flat_grad, torch.full_like(enc.flat_grad, 3.0))
        st._run_micro()                                                                      # later steps: fall-back, no rendezvous, no retry
        assert st.comm_retries == 1 and st.step_count == 2
        q.put((rank, "ok"))
    except Exception:                             # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_agree_on_the_fallback_before_either_retries():
    """one rank's bucket enqueue fails mid-backward, its peer leaves the collective later (timeout): the retry on the fall-back path starts
    only after BOTH have reached the rendezvous, so whole-buffer all-reduces never pair with bucket-sized ones; gradients are the exact sums."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_two_rank_retry_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
    assert res == {0: "ok", 1: "ok"}, res
