"""Decoder aliases (model/representation_learning/decoder/*.py:1-3): every dataset's decoder is ShiftUNet."""
from ...shift_unet import ShiftUNet

FFHQDecoder = ShiftUNet
CELEBAHQDecoder = ShiftUNet
CELEBA64Decoder = ShiftUNet
BEDROOMDecoder = ShiftUNet
HORSEDecoder = ShiftUNet
