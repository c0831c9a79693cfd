"""MI355X-native FFT convolution with the API of klae01/fft-conv-pytorch.

Exports mirror /root/reference/fft_conv_pytorch/__init__.py:1-9, plus the
top-level ``fft_conv`` the reference's README promises (README.md:22).
"""
from . import functional, nn
from .functional import fft_conv
from .nn import (
    FFTConv1d,
    FFTConv2d,
    FFTConv3d,
    FFTConvTranspose1d,
    FFTConvTranspose2d,
    FFTConvTranspose3d,
)

__version__ = "0.1.0"
