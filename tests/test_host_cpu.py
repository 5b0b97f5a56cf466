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
