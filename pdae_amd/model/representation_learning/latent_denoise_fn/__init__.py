"""Latent-DPM denoiser aliases (model/representation_learning/latent_denoise_fn/*.py): every dataset uses MLPSkipNet."""
from ...mlp_skip_net import MLPSkipNet

FFHQLatentDenoiseFn = MLPSkipNet
CELEBA64LatentDenoiseFn = MLPSkipNet
BEDROOMLatentDenoiseFn = MLPSkipNet
HORSELatentDenoiseFn = MLPSkipNet
