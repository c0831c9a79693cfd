"""Drop-in ``nn.Conv{N}d`` subclasses whose forward runs the HIP FFT convolution.

Mirrors /root/reference/fft_conv_pytorch/nn.py:7-51: constructor, parameters
and ``state_dict`` come unchanged from ``torch.nn.Conv{N}d`` through the MRO;
only ``forward`` differs.  On top of the reference, inference (``eval()`` or a
frozen weight) reuses the transformed kernel per (weight version, input geometry);
see ``_SpectrumCache`` for exactly when.
"""
import os

import torch
from torch import Tensor, nn

from . import functional as F_
from .utils import to_ntuple  # noqa: F401  (the reference's nn.py imports it too)


def _is_parametrized(module: nn.Module) -> bool:
    from torch.nn.utils import parametrize
    return parametrize.is_parametrized(module, "weight")


class _SpectrumCache:
    """Kernel-spectrum cache shared by the forward and the transposed modules.

    The transformed kernel is reused only while that is provably the same weight: the cache is consulted
    when the layer is in ``eval()`` mode, or the weight does not require grad, and the weight is a plain
    (un-parametrized) tensor; it is keyed on the plan, the weight's storage pointer and its version counter.
    A training step therefore always re-transforms (as the reference does on every call,
    /root/reference/fft_conv_pytorch/functional.py:71).  Writes that bypass the version counter
    (``layer.weight.data.copy_(...)``, ``dist.broadcast(layer.weight.data)``) are invisible to any key:
    call ``invalidate_kernel_spectrum()`` after them, or set ``cache_kernel_spectrum = False``."""

    cache_kernel_spectrum = True

    def invalidate_kernel_spectrum(self):
        self.__dict__.pop("_spectrum_cache", None)

    def _cached_spectrum(self, plan):
        weight = self.weight
        usable = (self.cache_kernel_spectrum and not _is_parametrized(self)
                  and not (self.training and weight.requires_grad and torch.is_grad_enabled()))
        if not usable:
            self.__dict__.pop("_spectrum_cache", None)
            return None
        tag = (id(plan), weight.data_ptr(), weight._version)
        cached = self.__dict__.get("_spectrum_cache")
        if cached is None or cached[0] != tag:
            cached = (tag, F_.transform_kernel(plan, weight))
            self.__dict__["_spectrum_cache"] = cached
        return cached[1]

    # The cached spectrum and the remembered plan hold native handles (ctypes pointers, a loaded library): they are
    # per-process acceleration state, not part of the module.  copy.deepcopy (EMA / AveragedModel, quantization flows),
    # pickle and torch.save(module) therefore see the module WITHOUT them -- exactly what a reference FFTConv module,
    # a plain nn.Conv subclass, carries -- and the copy rebuilds its own on first use.
    _TRANSIENT = ("_spectrum_cache", "_last_plan")

    def __getstate__(self):
        state = self.__dict__.copy()
        for key in self._TRANSIENT:
            state.pop(key, None)
        return state

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        clone = cls.__new__(cls)
        memo[id(self)] = clone
        for key, value in self.__dict__.items():
            if key not in self._TRANSIENT:
                clone.__dict__[key] = copy.deepcopy(value, memo)
        return clone

    def _apply(self, fn, *args, **kwargs):       # .to() / .cuda() / .float(): new storage, new spectrum
        self.invalidate_kernel_spectrum()
        self.__dict__.pop("_last_plan", None)
        return super()._apply(fn, *args, **kwargs)


class _FFTConvForward(_SpectrumCache, nn.Module):
    """Shared ``forward`` for FFTConv1d/2d/3d (reference: nn.py:7-22)."""

    def forward(self, signal: Tensor):
        assert signal.ndim == self.weight.ndim
        padding_mode = "constant" if self.padding_mode == "zeros" else self.padding_mode
        if isinstance(self.padding, str) or signal.dtype not in (torch.float32, torch.float64):
            # padding='same' / 'valid' (torch stores the string) and half-precision inputs: the functional resolves them
            return F_._fft_conv_impl(signal, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                     self.groups, padding_mode, None)
        plan = self._plan(signal, padding_mode)
        return F_._fft_conv_impl(signal, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                 self.groups, padding_mode, self._cached_spectrum(plan), plan)

    def _plan(self, signal: Tensor, padding_mode: str):
        """Plan for this call; the argument validation and descriptor lookup are skipped while the call looks
        exactly like the previous one (same input geometry, devices, dtypes and hyper-parameters)."""
        weight, bias = self.weight, self.bias
        sig = (signal.shape, signal.device, signal.dtype, weight.device, weight.dtype,
               None if bias is None else (bias.device, bias.dtype), self.stride, self.padding, self.dilation,
               self.groups, padding_mode, os.environ.get("FFTCONV_TILE"))
        last = self.__dict__.get("_last_plan")
        if last is not None and last[0] == sig:
            return last[1]
        plan = F_._plan_for(signal, weight, bias, self.stride, self.padding, self.dilation, self.groups, padding_mode)
        self.__dict__["_last_plan"] = (sig, plan)
        return plan


class _FFTConvTransposeForward(_SpectrumCache, nn.Module):
    """Shared ``forward`` for FFTConvTranspose1d/2d/3d (reference: nn.py:25-39)."""

    def forward(self, signal: Tensor):
        assert signal.ndim == self.weight.ndim
        if signal.dtype in F_._LOW_PRECISION:      # half-precision tensors: the functional casts (fp32 arithmetic)
            return F_._fft_conv_transpose_impl(signal, self.weight, self.bias, self.stride, self.padding,
                                               self.output_padding, self.dilation, self.groups, None, None)
        plan = F_._plan_for(signal, self.weight, self.bias, self.stride, self.padding, self.dilation,
                            self.groups, "constant", transposed=True, output_padding=self.output_padding)
        return F_._fft_conv_transpose_impl(signal, self.weight, self.bias, self.stride, self.padding,
                                           self.output_padding, self.dilation, self.groups,
                                           self._cached_spectrum(plan), plan)


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
