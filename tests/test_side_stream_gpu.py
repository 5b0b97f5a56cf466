"""GPU: the executor's second stream (include/pdae_hip.h: PDAE_OPF_SIDE, PDAE_OP_JOIN; pdae_amd/engine.py: Plan.emit_side / join).
(a) ordering of a hand-made op array: a side op sees what the ops in front of it wrote, a join makes its result visible to the ops behind it,
    and every pdae_run_ops call joins at its end;
(b) the representation-learning step: which ops of its plan carry the flag, and parameters / EMA / Adam moments after three steps BIT-IDENTICAL
    with the flag honoured (PDAE_SIDE_STREAM=1) and ignored (=0) -- the weight gradients and the shift branch of the
    forward pass only moved in time."""
import copy
import pytest
import torch

from tests.conftest import load_golden, T
from tests.golden import make_fixtures_cfg as C
from oracle import pdae_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_side_op_ordering_fork_and_join():
    from pdae_amd import hip as H
    n = 1 << 24                                            # 64 MB per buffer: the copies take long enough to expose a missing dependency
    a = torch.zeros(n, device=DEV); b = torch.zeros(n, device=DEV); c = torch.zeros(n, device=DEV); src = torch.full((n,), 3.0, device=DEV)
    side = H.make_op(H.OP_COPY, [a, b], [4 * n]); side.flags = H.OPF_SIDE
    ops = [H.make_op(H.OP_COPY, [src, a], [4 * n]),       # main: a = 3
           side,                                           # side: b = a        (must wait for the op in front of it)
           H.op_join(),
           H.make_op(H.OP_COPY, [b, c], [4 * n])]          # main: c = b        (must wait for the side op)
    for _ in range(5):
        for t in (a, b, c):
            t.zero_()
        H.run_ops(H.ops_array(ops), len(ops))
        torch.cuda.synchronize()
        assert float(c.min()) == 3.0 and float(c.max()) == 3.0
    # without an explicit join the call itself joins: the result of a trailing side op is visible to whatever the stream runs next
    for _ in range(5):
        b.zero_()
        H.run_ops(H.ops_array(ops[:2]), 2)
        d = b.clone()                                      # torch, same stream, right behind the call
        torch.cuda.synchronize()
        assert float(d.min()) == 3.0


def _three_steps(knob, mode):
    from pdae_amd.model.shift_unet import ShiftUNet
    from pdae_amd.model.representation_learning.encoder import CELEBA64Encoder
    from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    from pdae_amd.trainer.fused_step import FusedRLStep
    knob("PDAE_SIDE_STREAM", mode)
    g = load_golden("rl_step")
    cfg = C.CFG_SHIFT_64
    enc = CELEBA64Encoder(device=DEV, latent_dim=512); enc.load_state_dict(O.synth_state_dict(O.encoder_param_shapes("CELEBA64Encoder", 512), int(g["seed_enc"])))
    dec = ShiftUNet(device=DEV, latent_dim=512, **cfg); dec.load_state_dict(O.synth_state_dict(O.unet_param_shapes(cfg, shift=True, latent_dim=512), int(g["seed_dec"])))
    enc.train(); dec.set_train_mode()
    ema_enc, ema_dec = copy.deepcopy(enc), copy.deepcopy(dec)
    gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, torch.device(DEV))
    st = FusedRLStep(gd, enc, dec, ema_enc, ema_dec, 2, 64, 64, lr=1e-4, ema_decay=0.9999)
    x0 = T(g["x0"]).to(DEV)
    losses = [float(st.step(x0, t=T(g[f"t{s}"]).to(DEV), noise=T(g[f"noise{s}"]).to(DEV))) for s in range(3)]
    torch.cuda.synchronize()
    return st, losses, [enc.flat_train.clone(), dec.flat_train.clone(), ema_enc.flat_train.clone(), ema_dec.flat_train.clone()] + [m.clone() for m in st.m] + [v.clone() for v in st.v]


def test_training_step_is_bit_identical_with_and_without_the_second_stream(knob):
    from pdae_amd import hip as H
    st, l1, s1 = _three_steps(knob, 1)
    recs = st.plan.recs
    side = [k for k, o in enumerate(recs) if o.flags & H.OPF_SIDE]
    wg = [k for k, o in enumerate(recs) if o.kind == H.OP_CONV_WGRAD]
    assert wg and set(wg) <= set(side), "every convolution weight gradient runs on the second stream"
    fwd_side = [k for k in side if k not in set(wg)]
    # ... and the shift branch of the forward pass (model/graph.py), a contiguous-in-branch-order subset of the forward ops, joined before the loss
    assert fwd_side and max(fwd_side) < min(wg) and all(recs[k].kind != H.OP_CONV_WGRAD for k in fwd_side)
    joins = [k for k, o in enumerate(recs) if o.kind == H.OP_JOIN]
    assert any(max(fwd_side) < j < min(wg) for j in joins), "the forward's side branch is joined before anything reads its outputs"
    assert all(k < st.n_bwd for k in side)
    assert st.plan.ws_side is not None and not st.plan.side_parked
    _, l0, s0 = _three_steps(knob, 0)
    assert l0 == l1
    for x, y in zip(s0, s1):
        assert torch.equal(x, y)


@pytest.mark.timeout(1200)
def test_second_stream_stress_f128_b32_dropout_highest_priority_tiny_budget_and_ddp_segments():
    """the stress case.  tests/side_stress_worker.py in a fresh process (the side stream's priority is fixed at its creation):
    FFHQ-128 topology, B = 32, dropout on, side stream at the HIGHEST priority, 64 MB parking budget, plain step and the data-parallel driver's
    segmented backward -- three optimizer steps bit-identical to the one-stream order; the plan's schedule is proven by plancheck on the way."""
    import json
    import os
    import subprocess
    import sys
    from tests.conftest import ROOT
    env = dict(os.environ, PDAE_SIDE_STREAM="3", PDAE_SIDE_BUDGET_MB="64", PDAE_SIDE_BRANCH_BUDGET_MB="64")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "side_stress_worker.py"), "32"], capture_output=True, text=True, env=env, timeout=1100)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(next(ln for ln in r.stdout.splitlines() if ln.startswith('{"plain"')))          # (RCCL prints its banner behind it)
    for name in ("plain", "native"):
        o = out[name]
        assert o["identical"], (name, o)
        assert o["info"]["side_ops"] > 100 and o["info"]["joins"] > 10 and o["info"]["sat"] == 0, o["info"]
        assert o["info"]["check"]["pairs"] > 0
    assert out["native"]["info"]["buckets"] >= 4


def test_two_host_threads_on_their_own_streams_match_the_serial_result(knob):
    """VERDICT r5 #9 / SURVEY 8(b) "re-entrant": two host threads issue two independent training plans (each with side-stream ops) on their own HIP
    streams at the same time -- the library keeps one side stream + fork / join events per (device, calling stream) under a mutex, so the threads
    share nothing -- and after three steps each both end bit-identical to the same steps issued one after the other."""
    import threading
    from pdae_amd import hip as H
    knob("PDAE_SIDE_STREAM", 1)

    def make(seed):
        from pdae_amd.model.shift_unet import ShiftUNet
        from pdae_amd.model.representation_learning.encoder import CELEBA64Encoder
        from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
        from pdae_amd.trainer.fused_step import FusedRLStep
        cfg = dict(C.CFG_SHIFT_64, dropout=0.0)
        enc = CELEBA64Encoder(device=DEV, latent_dim=512); enc.load_state_dict(O.synth_state_dict(O.encoder_param_shapes("CELEBA64Encoder", 512), seed))
        dec = ShiftUNet(device=DEV, latent_dim=512, **cfg); dec.load_state_dict(O.synth_state_dict(O.unet_param_shapes(cfg, shift=True, latent_dim=512), seed + 1))
        enc.train(); dec.set_train_mode()
        gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, torch.device(DEV))
        st = FusedRLStep(gd, enc, dec, None, None, 4, 64, 64, lr=1e-4)
        g = torch.Generator().manual_seed(seed)
        data = [((torch.rand(4, 3, 64, 64, generator=g) * 2 - 1).to(DEV), torch.randint(0, 1000, (4,), generator=g).to(DEV), torch.randn(4, 3, 64, 64, generator=g).to(DEV)) for _ in range(3)]
        assert st.plan.n_side > 20
        return st, enc, dec, data

    def steps(job, stream, out, key):
        st, enc, dec, data = job
        with torch.cuda.stream(stream):
            losses = [st.step(x, t=t, noise=n) for x, t, n in data]
        stream.synchronize()
        out[key] = ([float(l) for l in losses], enc.flat_train.clone(), dec.flat_train.clone())
    torch.cuda.synchronize()
    serial, conc = {}, {}
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for key, seed, s in (("a", 11, s1), ("b", 23, s2)):
        steps(make(seed), s, serial, key)
    jobs = {"a": make(11), "b": make(23)}
    torch.cuda.synchronize()
    ths = [threading.Thread(target=steps, args=(jobs[k], s, conc, k)) for k, s in (("a", s1), ("b", s2))]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for k in ("a", "b"):
        assert serial[k][0] == conc[k][0], k
        assert torch.equal(serial[k][1], conc[k][1]) and torch.equal(serial[k][2], conc[k][2]), k
