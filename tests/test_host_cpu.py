"""CPU: host-side logic of the product that needs no kernel -- schedule tables, respacing and DDIM coefficient rows of
pdae_amd.diffusion against the vectors the reference emitted (tests/golden/schedules.npz), bucket planning, config loading."""
import numpy as np
import torch

from tests.conftest import load_golden


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
