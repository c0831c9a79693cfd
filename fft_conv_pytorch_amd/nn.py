"""Drop-in ``nn.Conv{N}d`` subclasses whose forward runs the HIP FFT convolution.

Mirrors /root/reference/fft_conv_pytorch/nn.py:7-51: constructor, parameters
and ``state_dict`` come unchanged from ``torch.nn.Conv{N}d`` through the MRO;
only ``forward`` differs.  On top of the reference, the transformed kernel is
cached per (weight version, input geometry), so steady-state inference pays for
the kernel FFT once per weight update (the reference recomputes it every call,
functional.py:71).
"""
import os

from torch import Tensor, nn

from . import functional as F_
from .utils import to_ntuple  # noqa: F401  (the reference's nn.py imports it too)


class _FFTConvForward(nn.Module):
    """Shared ``forward`` for FFTConv1d/2d/3d (reference: nn.py:7-22)."""

    cache_kernel_spectrum = True

    def forward(self, signal: Tensor):
        assert signal.ndim == self.weight.ndim
        padding_mode = "constant" if self.padding_mode == "zeros" else self.padding_mode
        plan = self._plan(signal, padding_mode)
        tag = (id(plan), self.weight.data_ptr(), self.weight._version)
        cached = self.__dict__.get("_spectrum_cache")
        if not self.cache_kernel_spectrum or cached is None or cached[0] != tag:
            cached = (tag, F_.transform_kernel(plan, self.weight))
            self.__dict__["_spectrum_cache"] = cached
        return F_._fft_conv_impl(signal, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                 self.groups, padding_mode, cached[1], plan)

    def _plan(self, signal: Tensor, padding_mode: str):
        """Plan for this call; the argument validation and descriptor lookup are skipped while the call looks
        exactly like the previous one (same input geometry, devices, dtypes and hyper-parameters)."""
        bias = self.bias
        sig = (signal.shape, signal.device, signal.dtype, self.weight.device, self.weight.dtype,
               None if bias is None else (bias.device, bias.dtype), self.stride, self.padding, self.dilation,
               self.groups, padding_mode, os.environ.get("FFTCONV_TILE"))
        last = self.__dict__.get("_last_plan")
        if last is not None and last[0] == sig:
            return last[1]
        plan = F_._plan_for(signal, self.weight, bias, self.stride, self.padding, self.dilation, self.groups, padding_mode)
        self.__dict__["_last_plan"] = (sig, plan)
        return plan


class _FFTConvTransposeForward(nn.Module):
    """Shared ``forward`` for FFTConvTranspose1d/2d/3d (reference: nn.py:25-39)."""

    cache_kernel_spectrum = True

    def forward(self, signal: Tensor):
        assert signal.ndim == self.weight.ndim
        plan = F_._plan_for(signal, self.weight, self.bias, self.stride, self.padding, self.dilation,
                            self.groups, "constant", transposed=True, output_padding=self.output_padding)
        tag = (id(plan), self.weight.data_ptr(), self.weight._version)
        cached = self.__dict__.get("_spectrum_cache")
        if not self.cache_kernel_spectrum or cached is None or cached[0] != tag:
            cached = (tag, F_.transform_kernel(plan, self.weight))
            self.__dict__["_spectrum_cache"] = cached
        return F_._forward_native(signal, cached[1], self.bias)


class FFTConv1d(_FFTConvForward, nn.Conv1d):
    ...


class FFTConv2d(_FFTConvForward, nn.Conv2d):
    ...


class FFTConv3d(_FFTConvForward, nn.Conv3d):
    ...


class FFTConvTranspose1d(_FFTConvTransposeForward, nn.ConvTranspose1d):
    ...


class FFTConvTranspose2d(_FFTConvTransposeForward, nn.ConvTranspose2d):
    ...


class FFTConvTranspose3d(_FFTConvTransposeForward, nn.ConvTranspose3d):
    ...
