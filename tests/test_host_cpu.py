"""CPU: host-side logic of the product that needs no kernel -- schedule tables, respacing and DDIM coefficient rows of
pdae_amd.diffusion against the vectors the reference emitted (tests/golden/schedules.npz), bucket planning, config loading."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import ROOT, load_golden


def test_product_schedule_tables_match_reference_vectors():
    from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    g = load_golden("schedules")
    gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, torch.device("cpu"))
    names = [k[4:] for k in g if k.startswith("lin_")]
    assert len(names) >= 13
    for n in names:
        assert np.array_equal(getattr(gd, n).numpy(), g["lin_" + n]), n
    gc = GaussianDiffusion({"timesteps": 1000, "betas_type": "cosine"}, torch.device("cpu"))
    for n in [k[4:] for k in g if k.startswith("cos_")]:
        assert np.array_equal(getattr(gc, n).numpy(), g["cos_" + n]), n
    for style in ["ddim10", "ddim20", "ddim100", "ddim1000"]:
        d = gd._ddim(style)
        assert np.array_equal(d.timestep_map.numpy(), g[style + "_map"])
        assert d.timesteps == len(g[style + "_map"]) - 1
        for n in d.TABLES:
            assert np.array_equal(getattr(d, n).numpy(), g[style + "_" + n]), (style, n)
        # the per-sample coefficient rows are the scalar coefficients of each step, for both directions
        for enc in (False, True):
            rows = d._coef_rows(torch.arange(d.timesteps + 1), enc).numpy()
            for i in (0, 1, d.timesteps // 2, d.timesteps):
                assert np.array_equal(rows[i], np.array(d._coefs(i, enc), dtype=np.float32)), (style, enc, i)
    assert gd._ddim("ddim1000").timesteps == 999          # duplicate integer steps collapse
    import pytest
    with pytest.raises(NotImplementedError):
        GaussianDiffusion({"timesteps": 10, "betas_type": "quadratic"}, torch.device("cpu"))


def test_latent_schedule_is_constant_beta():
    from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    cfg = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, torch.device("cpu")).latent_diffusion_config
    assert cfg["timesteps"] == 1000 and cfg["loss_type"] == "l1"
    ac = np.cumprod(1.0 - np.full(1000, 0.008))
    assert np.array_equal(cfg["alphas_cumprod"].numpy(), ac.astype(np.float32))
    assert np.array_equal(cfg["sqrt_one_minus_alphas_cumprod"].numpy(), np.sqrt(1.0 - ac).astype(np.float32))


def test_plans_take_groupnorm_statistics_from_the_producing_convolution(monkeypatch):
    """Plan construction (no GPU needed: only size queries reach the library): a GroupNorm whose input tensors were all written by single-launch
    3x3 patch convolutions is planned as pdae_gn_coef_from_conv_stats (op kind 42) on their partial sums -- forward convolutions carry the
    partial-sum pointer in slot 19 -- while the stem, stride-2 and attention outputs keep the statistics pass (kind 32); PDAE_FUSE_GN_STATS=0
    restores the statistics pass everywhere."""
    import collections
    import torch
    from pdae_amd import hip as H
    from pdae_amd.model.shift_unet import ShiftUNet
    cfg = dict(input_channel=3, base_channel=128, channel_multiplier=[1, 2], num_residual_blocks_of_a_block=1, attention_resolutions=[],
               num_heads=1, head_channel=-1, use_new_attention_order=False, dropout=0.0)

    def kinds(flag):
        monkeypatch.setenv("PDAE_FUSE_GN_STATS", flag)
        dec = ShiftUNet(device=torch.device("cpu"), latent_dim=64, **cfg)
        dec.eval()
        p = dec.plan(64, 64, 64, False)                   # B = 64 at 64 x 64: enough tiles that no forward launch splits K
        c = collections.Counter(op.kind for op in p.recs)
        armed = sum(1 for op in p.recs if op.kind in (H.OP_CONV_FWD, H.OP_CONV_FWD_GN, H.OP_CONV_FWD_SKIP) and op.p[19])
        return c, armed
    on, armed_on = kinds("1")
    off, armed_off = kinds("0")
    assert armed_off == 0 and off[H.OP_GN_COEF_FROM_CONV_STATS] == 0
    assert armed_on > 0 and on[H.OP_GN_COEF_FROM_CONV_STATS] > 0
    assert on[H.OP_GN_STATS_COEF] + on[H.OP_GN_COEF_FROM_CONV_STATS] == off[H.OP_GN_STATS_COEF]       # every GroupNorm is still there
    assert on[H.OP_GN_STATS_COEF] >= 1                                                               # e.g. the tensor behind the stem convolution


def test_epoch_order_partitions_every_epoch_across_ranks():
    """every data-parallel rank must walk its OWN share of one common per-epoch permutation (DistributedSampler,
    base_trainer.py:73-78) -- not the same images on every rank.  Shares are disjoint, equally long, cover the dataset, change per epoch,
    do not depend on `augmentation`; mirror coins differ between ranks; an evaluation order is the identity."""
    from pdae_amd.dataset import EpochOrder
    L, W = 1003, 4
    for epoch in (0, 1, 7):
        shares = [EpochOrder(L, r, W, seed=5).indices(epoch) for r in range(W)]
        assert len({len(s) for s in shares}) == 1 and len(shares[0]) == (L + W - 1) // W
        allidx = np.concatenate(shares)
        assert set(allidx.tolist()) == set(range(L))                              # covers the dataset ...
        assert len(allidx) - len(set(allidx.tolist())) == W * len(shares[0]) - L   # ... with only the wrap-around padding duplicated
        for a in range(W):
            for b in range(a + 1, W):
                assert len(set(shares[a].tolist()) & set(shares[b].tolist())) <= W * len(shares[0]) - L
    e0, e1 = EpochOrder(L, 1, W, seed=5).indices(0), EpochOrder(L, 1, W, seed=5).indices(1)
    assert not np.array_equal(e0, e1) and not np.array_equal(e0, np.sort(e0))     # shuffled, and reshuffled per epoch
    assert np.array_equal(EpochOrder(L, 1, W, seed=5, flip=False).indices(0), e0)  # the order does not depend on the augmentation flag
    assert np.array_equal(EpochOrder(L, 0, 1, shuffle=False).indices(3), np.arange(L))
    f0, f1 = EpochOrder(L, 0, W, seed=5).flips(0, 0, 256), EpochOrder(L, 1, W, seed=5).flips(0, 0, 256)
    assert not np.array_equal(f0, f1) and 64 < int(f0.sum()) < 192
    assert int(EpochOrder(L, 0, W, seed=5, flip=False).flips(0, 0, 64).sum()) == 0
    # training order: DistributedSampler(drop_last=True) of base_trainer.py:73-78 drops the len % world tail -- floor(len / world)
    # images per rank, no image twice in an epoch; only the evaluator's order pads
    for epoch in (0, 3):
        shares = [EpochOrder(L, r, W, seed=5, drop_tail=True).indices(epoch) for r in range(W)]
        assert all(len(s) == L // W for s in shares) and EpochOrder(L, 0, W, drop_tail=True).per_rank == L // W
        allidx = np.concatenate(shares)
        assert len(set(allidx.tolist())) == len(allidx) == (L // W) * W


def test_trainers_and_sampler_hand_their_rank_to_the_dataset(monkeypatch):
    """The four entry points that build a dataset pass device / rank / world_size (and a seed shared by all ranks)."""
    import ast
    import inspect
    import pdae_amd.sampler.autoencoding_eval as S
    import pdae_amd.trainer.train_latent_diffusion as TL
    import pdae_amd.trainer.train_regular_diffusion as TG
    import pdae_amd.trainer.train_representation_learning as TR
    n = 0
    for mod in (S, TL, TG, TR):
        for node in ast.walk(ast.parse(inspect.getsource(mod))):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "build" and getattr(node.func.value, "id", "") == "dataset_module":
                kws = {k.arg for k in node.keywords} | ({"**"} if any(k.arg is None for k in node.keywords) else set())
                assert {"rank", "world_size", "device"} <= kws or "**" in kws, (mod.__name__, kws)
                n += 1
    assert n >= 5
    assert TR.DATA_SEED == TG.DATA_SEED == TL.DATA_SEED


def test_encoder_plan_runs_stride2_convolutions_in_the_dense_grid_form(monkeypatch):
    """engine.Builder._dense_grid_desc: with the default switches the planned FFHQ encoder pass (B = 32) scatters dY of every eligible stride-2
    3x3 convolution onto the stride-1 grid once (pdae_zero_insert2), runs its weight / data gradient (and the forward of the small layers) as
    stride-1 records and marks those records for bench.py; PDAE_S2_DENSE=0 leaves only stride-2 records.  Plans are built on the CPU (records
    only, nothing is launched)."""
    import torch
    from pdae_amd import hip as H
    from pdae_amd.engine import Builder, Plan
    from pdae_amd.model import graph as G
    from pdae_amd.model.representation_learning.encoder import FFHQEncoder

    def build(flag):
        monkeypatch.setenv("PDAE_S2_DENSE", flag)
        enc = FFHQEncoder(device="cpu", latent_dim=512)
        enc.train()
        p = Plan("cpu")
        B = Builder(p, enc.P, enc.grads(), save=True, math="f16x3")
        x = torch.zeros(32, 128, 128, 3)
        z, ex = G.encoder_forward(B, enc.NAME, x)
        G.encoder_backward(B, ex, torch.zeros_like(z))
        return p

    def convs(plan):
        return [r for r in plan.recs if r.kind in (H.OP_CONV_FWD, H.OP_CONV_DGRAD, H.OP_CONV_WGRAD)]

    on, off = build("1"), build("0")
    kinds_on, kinds_off = [r.kind for r in on.recs], [r.kind for r in off.recs]
    assert H.OP_ZERO_INSERT2 not in kinds_off and H.OP_SUBSAMPLE2 not in kinds_off and not off.dense_grid
    n_s2_off = sum(1 for r in convs(off) if r.i[10] == 2)
    assert n_s2_off == 5 + 5 + 4                                            # five stride-2 layers: forward + dW each, dX of all but the first
    nz, ns = kinds_on.count(H.OP_ZERO_INSERT2), kinds_on.count(H.OP_SUBSAMPLE2)
    assert nz == 4 and ns == 3 and len(on.dense_grid) == 4 + 4 + 3           # Cin = 3 stays generic; outputs of <= 8192 pixels also forward
    for k in on.dense_grid:
        r = on.recs[k]
        assert r.kind in (H.OP_CONV_FWD, H.OP_CONV_DGRAD, H.OP_CONV_WGRAD) and r.i[8] == 3 and r.i[10] == 1      # 3x3, stride 1
    assert sum(1 for r in convs(on) if r.i[10] == 2) == n_s2_off - len(on.dense_grid)      # each marked record replaces one stride-2 record


def test_sampling_plan_has_a_pinned_latent_only_prefix(monkeypatch):
    """model/graph.py unet_forward: in a sampling plan of ShiftUNet the ops that depend only on the latent z (label_emb, its SiLU, the grouped
    emb_z_layers Linear) come first (Plan.n_const), their results are pinned -- never returned to the buffer pool, so no later op of a step can
    overwrite what the following steps still read -- and PDAE_DDIM_HOIST=0 restores the flat list.  Built on the CPU: records only."""
    import torch
    from pdae_amd import hip as H
    from pdae_amd.model.shift_unet import ShiftUNet
    from tests.golden import make_fixtures_cfg as C

    def plan(flag):
        monkeypatch.setenv("PDAE_DDIM_HOIST", flag)
        net = ShiftUNet(device="cpu", latent_dim=512, **C.CFG_SHIFT_64)
        net.set_eval_mode()
        return net.plan(2, 64, 64, False)

    on, off = plan("1"), plan("0")
    assert off.n_const == 0 and not off.pinned
    assert on.n_const >= 3 and len(on.recs) == len(off.recs) + 1                  # same ops reordered, the grouped Linear launch in two (z part, time part)
    kinds = [r.kind for r in on.recs[:on.n_const]]
    assert kinds[0] == H.OP_GEMM and H.OP_SILU in kinds and H.OP_LINEAR_GROUP in kinds and H.OP_TEMB not in kinds
    assert H.OP_TEMB in [r.kind for r in on.recs[on.n_const:on.n_fwd]]             # everything time-dependent runs every step
    pooled = {id(t) for ts in on.pool.values() for t in ts}
    assert on.pinned and not (on.pinned & pooled)                                 # pinned buffers were never recycled
    # the training plan keeps the reference's order: the prefix exists only where nothing is saved for a backward
    monkeypatch.setenv("PDAE_DDIM_HOIST", "1")
    net = ShiftUNet(device="cpu", latent_dim=512, **C.CFG_SHIFT_64)
    net.set_train_mode()
    assert net.plan(2, 64, 64, True).n_const == 0


def test_eps_only_sampling_plan_shares_the_input_buffers_and_has_no_shift_ops():
    """ShiftUNet.plan_eps (the steps of shift_ddim_sample_loop whose shift term is discarded, reference diffusion/ddim.py:94-96,115,119): the eps half
    alone on the x / t buffers of the full sampling plan -- same convolution records as the eps half of the full plan, none of the shift
    branch's, no latent input, nothing to prepare per run.  Built on the CPU: records only."""
    from pdae_amd import hip as H
    from pdae_amd.model.shift_unet import ShiftUNet
    from tests.golden import make_fixtures_cfg as C
    net = ShiftUNet(device="cpu", latent_dim=512, **C.CFG_SHIFT_64)
    net.set_eval_mode()
    full, pe = net.plan(2, 64, 64, False), net.plan_eps(2, 64, 64)
    assert pe is net.plan_eps(2, 64, 64) and pe.x is full.x and pe.t is full.t and pe.z is None and pe.shift is None
    assert pe.n_const == 0 and not pe.wprep_jobs and pe.pre_arr is None          # frozen weights only: prepared once (init ops), never per run
    conv_kinds = (H.OP_CONV_FWD, H.OP_CONV_FWD_GN, H.OP_CONV_FWD_SKIP)
    n_full = sum(1 for r in full.recs[:full.n_fwd] if r.kind in conv_kinds)
    n_eps = sum(1 for r in pe.recs if r.kind in conv_kinds)
    # the full plan runs trunk + eps branch + shift branch; the shift branch mirrors the eps branch (middle + output blocks + head)
    assert 0.5 * n_full < n_eps < 0.75 * n_full, (n_eps, n_full)
    assert pe.n_fwd == len(pe.recs) < 0.75 * (full.n_fwd - full.n_const)
    # x is an input of both op lists and must never have been handed back to either buffer pool
    for pl in (full, pe):
        assert all(t is not full.x and t is not full.t for ts in pl.pool.values() for t in ts)


def test_plan_parks_what_a_side_stream_op_reads_until_the_join():
    """Plan.emit_side / join (the executor's second stream, pdae_hip.h PDAE_OPF_SIDE / PDAE_OP_JOIN), on CPU tensors: a buffer that a pending side op
    reads is not handed out again by buf() -- neither as itself nor through a view of its storage -- until a join has been emitted; the join
    is an op of its own; side ops get the SECOND workspace; compile() leaves nothing parked."""
    import torch
    from pdae_amd import hip as H
    from pdae_amd.engine import Plan
    p = Plan(torch.device("cpu"))
    a, b = p.buf(8, 8), p.buf(8, 8)
    main_op = H.make_op(H.OP_MEMSET, [a], [256])
    side_op = H.make_op(H.OP_COPY, [a, b], [256])
    p.emit(main_op)
    k = p.emit_side(side_op, [a.view(64)], ws_slot=5, wsb_slot=3, ws_bytes=1 << 16)       # a VIEW of `a`: the storage is what counts
    assert p.recs[k].flags == H.OPF_SIDE and p.recs[0].flags == 0
    p.free(a)
    c = p.buf(8, 8)
    assert c.data_ptr() != a.data_ptr(), "the side op's input was recycled before the join"
    assert p.side_parked and p.side_parked_bytes == 256
    p.join(force=False)                                         # below the budget: nothing happens
    assert p.side_parked and p.recs[-1].kind != H.OP_JOIN
    p.join()
    assert p.recs[-1].kind == H.OP_JOIN and not p.side_parked and not p.side_busy
    d = p.buf(8, 8)
    assert d.data_ptr() == a.data_ptr()                         # now it is back in the pool
    p.free(b)                                                   # (b is an OUTPUT of the side op: not parked, the builder never reads it before a join)
    p.emit_side(H.make_op(H.OP_COPY, [c, d], [256]), [c])
    p.free(c)
    n = len(p.recs)
    p.compile()
    assert p.ws_side is not None and p.arr[k].p[5] == p.ws_side.data_ptr() and p.arr[k].i[3] == p.ws_side_bytes and p.ws_side_bytes == 1 << 16
    assert len(p.recs) == n + 1 and p.recs[-1].kind == H.OP_JOIN and not p.side_parked


def test_builder_switch_for_side_stream_weight_gradients(monkeypatch):
    """PDAE_SIDE_WGRAD is read when a Builder is created (the census over a real training plan runs in the GPU suite, tests/test_side_stream_gpu.py)."""
    import torch
    from pdae_amd.engine import Plan, Builder
    monkeypatch.setenv("PDAE_SIDE_WGRAD", "0")
    b0 = Builder(Plan(torch.device("cpu")), {}, grads={}, save=True)
    monkeypatch.setenv("PDAE_SIDE_WGRAD", "1")
    b1 = Builder(Plan(torch.device("cpu")), {}, grads={}, save=True)
    assert b0.side_wgrad is False and b1.side_wgrad is True


def test_plan_side_branch_parks_inputs_and_outputs_and_shiftunet_uses_it(monkeypatch):
    """Plan.side(): every op emitted inside carries PDAE_OPF_SIDE and every pool buffer one of its pointers falls into (inputs AND outputs, matched by
    address range) is parked when freed; ShiftUNet's plan puts exactly its shift branch there and joins behind it (PDAE_SIDE_SHIFT=0: nothing)."""
    import torch
    from pdae_amd import hip as H
    from pdae_amd.engine import Plan
    p = Plan(torch.device("cpu"))
    a, b, c = p.buf(16), p.buf(16), p.buf(16)
    with p.side():
        k = p.emit(H.make_op(H.OP_COPY, [a[4:], b], [48]))     # a pointer INTO a's buffer, and an output
    assert p.recs[k].flags == H.OPF_SIDE and not p.side_mode
    p.emit(H.make_op(H.OP_COPY, [c, c], [64]))
    assert p.recs[-1].flags == 0
    p.free(a, b, c)
    got = {p.buf(16).data_ptr() for _ in range(3)}
    assert c.data_ptr() in got and a.data_ptr() not in got and b.data_ptr() not in got       # only the main stream's buffer came back
    p.join()
    assert {p.buf(16).data_ptr(), p.buf(16).data_ptr()} == {a.data_ptr(), b.data_ptr()}

    from pdae_amd.model.shift_unet import ShiftUNet
    cfg = dict(input_channel=3, base_channel=32, channel_multiplier=[1, 2], num_residual_blocks_of_a_block=1, attention_resolutions=[],
               num_heads=1, head_channel=-1, use_new_attention_order=False, dropout=0.0)

    def census(flag):
        monkeypatch.setenv("PDAE_SIDE_SHIFT", flag)
        dec = ShiftUNet(device=torch.device("cpu"), latent_dim=64, **cfg)
        dec.eval()
        pl = dec.plan(2, 32, 32, False)
        side = [k for k, o in enumerate(pl.recs) if o.flags & H.OPF_SIDE]
        joins = [k for k, o in enumerate(pl.recs) if o.kind == H.OP_JOIN]
        return pl, side, joins
    pl, side, joins = census("1")
    assert side and joins and max(side) < max(joins) and not pl.side_parked and not pl.side_busy
    assert 0.2 < len(side) / len(pl.recs) < 0.6                 # roughly the shift half of what follows the input blocks
    _, side0, _ = census("0")
    assert not side0


def test_plan_checker_finds_every_kind_of_two_stream_hazard():
    """pdae_amd/plancheck.py on hand-made plans (CPU tensors): a main-stream op between a side op and its join that (a) overwrites the side op's
    input, (b) overwrites its output, (c) reads its output is refused by Plan.compile(); the same ops behind a join, and main-stream ops on
    unrelated buffers, pass.  Granularity is the pool buffer: a pointer INTO a buffer counts as the buffer."""
    import pytest
    import torch
    from pdae_amd import hip as H
    from pdae_amd.engine import Plan
    from pdae_amd.plancheck import PlanHazard, check_plan

    def plan(main_op_of, join_first=False):
        p = Plan(torch.device("cpu"))
        a, b, c, d = p.buf(16), p.buf(16), p.buf(16), p.buf(16)
        p.emit_side(H.make_op(H.OP_COPY, [a, b], [64]), [a])          # side: reads a, writes b
        if join_first:
            p.join()
        p.emit(main_op_of(a, b, c, d))
        return p
    bad = {"writes the side op's input": lambda a, b, c, d: H.make_op(H.OP_MEMSET, [a[4:]], [16]),
           "writes the side op's output": lambda a, b, c, d: H.make_op(H.OP_COPY, [c, b], [64]),
           "reads the side op's output": lambda a, b, c, d: H.make_op(H.OP_COPY, [b[8:], d], [32])}
    for why, f in bad.items():
        with pytest.raises(PlanHazard):
            plan(f).compile()
        ok = plan(f, join_first=True).compile()                      # ordered by the join: fine
        assert ok.check["side_ops"] == 1 and ok.check["joins"] >= 1, why
    ok = plan(lambda a, b, c, d: H.make_op(H.OP_COPY, [c, d], [64])).compile()      # unrelated buffers beside the side op
    assert ok.check["pairs"] == 1
    ok = plan(lambda a, b, c, d: H.make_op(H.OP_COPY, [a, d], [64])).compile()      # two READERS of one buffer do not conflict
    assert ok.check["pairs"] == 1
    # an op whose operands sit in a device table must bring them along (Builder.linear_group does): without, the schedule is unprovable
    p = Plan(torch.device("cpu"))
    a, b = p.buf(16), p.buf(16)
    p.emit_side(H.make_op(H.OP_COPY, [a, b], [64]), [a])
    p.emit(H.op_linear_group(a, b, 1, 1, 1, 8))
    with pytest.raises(PlanHazard):
        check_plan(p)
    p.recs[-1].rw = ([a], [b])                                       # ... and with them the conflict (it writes b) is seen
    with pytest.raises(PlanHazard):
        check_plan(p)
    # every op kind of the library has its write slots declared
    from pdae_amd import plancheck
    kinds = {getattr(H, k) for k in dir(H) if k.startswith("OP_") and isinstance(getattr(H, k), int)}
    assert kinds - set(plancheck.WRITES) - plancheck.TABLE_KINDS == {H.OP_JOIN}


@pytest.mark.timeout(900)
def test_training_and_sampling_plans_are_proven_hazard_free_on_the_cpu(monkeypatch):
    """Every plan is checked when it is compiled (Plan.compile -> plancheck.check_plan; PDAE_PLAN_CHECK=0 opts out): here the three trainers'
    step plans and ShiftUNet's sampling plans are BUILT on the CPU at the shipped FFHQ-128 topology (B = 1) and at a small one with dropout and
    attention, with the default second-stream switches and with a tiny parking budget (joins in the middle of the backward and of the shift
    branch), and the census of what was proven is non-trivial.  The optimizer ops sit behind an explicit join (found by this check: the
    step used to rely on the implicit join at the end of the backward's pdae_run_ops call)."""
    import copy
    import torch
    from pdae_amd import hip as H
    from pdae_amd.utils import load_yaml
    from pdae_amd.model.shift_unet import ShiftUNet
    from pdae_amd.model.unet import UNet
    from pdae_amd.model.representation_learning.encoder import CELEBA64Encoder, FFHQEncoder
    from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    from pdae_amd.trainer.fused_step import FusedRLStep, FusedRegularStep
    dev = torch.device("cpu")
    gd = GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, dev)
    small = dict(input_channel=3, base_channel=32, channel_multiplier=[1, 2], num_residual_blocks_of_a_block=1, attention_resolutions=[2],
                 num_heads=1, head_channel=-1, use_new_attention_order=False, dropout=0.1)
    f128 = load_yaml(os.path.join(ROOT, load_yaml(os.path.join(ROOT, "config/ffhq_representation_learning.yml"))["trained_ddpm_config"]))["denoise_fn_config"]

    def rl(cfg, Enc, latent, B, S):
        enc, dec = Enc(device=dev, latent_dim=latent), ShiftUNet(device=dev, latent_dim=latent, **cfg)
        enc.train(); dec.set_train_mode()
        st = FusedRLStep(gd, enc, dec, copy.deepcopy(enc), copy.deepcopy(dec), B, S, S)
        return st, dec
    for budget in ("6144", "1"):
        monkeypatch.setenv("PDAE_SIDE_BUDGET_MB", budget)
        monkeypatch.setenv("PDAE_SIDE_BRANCH_BUDGET_MB", "24576" if budget != "1" else "1")
        for cfg, Enc, latent, B, S in ((small, CELEBA64Encoder, 64, 2, 64), (f128, FFHQEncoder, 512, 1, 128)):
            st, dec = rl(cfg, Enc, latent, B, S)
            pl = st.plan
            chk = pl.check
            assert chk["side_ops"] >= 20 and chk["pairs"] > (10 if budget != "1" else 2) * chk["side_ops"] and chk["joins"] >= (2 if budget != "1" else 4), (budget, chk)
            adam = [k for k, o in enumerate(pl.recs) if o.kind == H.OP_ADAM_EMA]
            assert pl.recs[min(adam) - 1].kind == H.OP_JOIN and min(adam) == st.n_bwd
            dec.set_eval_mode()
            for p2 in (dec.plan(B, S, S, False), dec.plan_eps(B, S, S)):
                if p2.n_side:
                    assert p2.check["side_ops"] == p2.n_side and (p2.check["pairs"] > 0 or budget == "1")      # (a 1 MB budget joins behind nearly every side op)
            if budget == "1":
                assert dec.plan(B, S, S, False).side_parked_peak <= (2 << 20) + 4 * B * S * S * 4 * max(cfg["channel_multiplier"]) * cfg["base_channel"]
    un = UNet(device=dev, **dict(small, input_channel=1))
    un.train()
    st = FusedRegularStep(gd, un, copy.deepcopy(un), 2, 32, 32)
    assert st.plan.check["side_ops"] > 0 and st.plan.recs[st.n_bwd - 1].kind == H.OP_JOIN
    # PDAE_PLAN_CHECK=0 opts out (the attribute is then absent)
    monkeypatch.setenv("PDAE_PLAN_CHECK", "0")
    st = FusedRegularStep(gd, un, copy.deepcopy(un), 2, 32, 32)
    assert not hasattr(st.plan, "check")
