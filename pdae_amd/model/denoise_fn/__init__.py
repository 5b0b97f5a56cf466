"""model/denoise_fn/mnist.py: MNISTDenoiseFn = UNet."""
from ..unet import UNet

MNISTDenoiseFn = UNet
