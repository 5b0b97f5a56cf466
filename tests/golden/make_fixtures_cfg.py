"""Tiny network configs shared by the fixture generator and the tests (data only)."""
CFG_UNET_A = dict(input_channel=1, base_channel=32, channel_multiplier=[1, 2], num_residual_blocks_of_a_block=1,
                  attention_resolutions=[2], num_heads=2, head_channel=-1, use_new_attention_order=True,
                  dropout=0.0, num_class=10)
CFG_UNET_B = dict(input_channel=1, base_channel=32, channel_multiplier=[1, 2, 2], num_residual_blocks_of_a_block=2,
                  attention_resolutions=[], num_heads=1, head_channel=-1, use_new_attention_order=False, dropout=0.0)
CFG_SHIFT_T = dict(input_channel=3, base_channel=32, channel_multiplier=[1, 2], num_residual_blocks_of_a_block=1,
                   attention_resolutions=[2], num_heads=1, head_channel=-1, use_new_attention_order=False, dropout=0.0)
CFG_SHIFT_64 = dict(input_channel=3, base_channel=32, channel_multiplier=[1, 2, 2], num_residual_blocks_of_a_block=1,
                    attention_resolutions=[4], num_heads=1, head_channel=32, use_new_attention_order=False, dropout=0.0)
CFG_MLP = dict(input_channel=64, model_channel=128, num_layers=4, time_emb_channel=32, use_norm=True, dropout=0.0)
# ---- f3 fixtures (tests/golden/make_fixtures_f3.py): short chain so the 'all timesteps' loops stay cheap
F3_T = 100
CFG_UNET_SIGMA = dict(input_channel=3, base_channel=32, channel_multiplier=[1, 2], num_residual_blocks_of_a_block=1,
                      attention_resolutions=[2], num_heads=1, head_channel=-1, use_new_attention_order=False, dropout=0.0, learn_sigma=True)


def f3_noise(stream, i, shape, uniform=False):
    """Deterministic stand-in for the reference's internal RNG draws (numpy PCG64 streams are machine-independent): stream = which
    sampler, i = the loop's timestep index.  Returns a float32 numpy array."""
    import numpy as np
    rng = np.random.default_rng(7_000_000 + 10_000 * stream + i)
    a = rng.uniform(0.0, 1.0, shape) if uniform else rng.standard_normal(shape)
    return a.astype(np.float32)
