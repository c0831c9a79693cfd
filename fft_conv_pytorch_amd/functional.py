"""``fft_conv`` for MI355X: same call signature as the reference, HIP kernels underneath.

Mirrors /root/reference/fft_conv_pytorch/functional.py (fft_conv :19-89,
complex_matmul :11-16, to_ntuple re-export :8).  The Python layer only
normalises arguments, checks shapes, allocates the output and hands raw device
pointers + the current HIP stream to libfftconv_amd.so.  There is no CPU path
and no torch.fft / rocFFT / hipFFT call anywhere in this package.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional, Union

import torch
from torch import Tensor

from . import _native
from .utils import to_ntuple

__all__ = ["fft_conv", "fft_conv_transpose", "complex_matmul", "to_ntuple", "transform_kernel", "KernelSpectrum"]


_DTYPE_CODES = {torch.float32: 0, torch.float64: 1}      # enum fc_dtype


def _require_gpu_f32(name: str, t: Tensor, dtype: torch.dtype = torch.float32):
    """Device + dtype gate of everything handed to the library (``dtype``: the signal's, float32 or float64)."""
    if not t.is_cuda:
        raise RuntimeError(
            f"fft_conv_pytorch_amd: `{name}` is on {t.device}; this implementation runs on ROCm devices only "
            f"(no CPU fallback). Move the tensor to 'cuda'.")
    if t.dtype != dtype or dtype not in _DTYPE_CODES:
        raise TypeError(f"fft_conv_pytorch_amd: `{name}` has dtype {t.dtype}; signal, kernel and bias must share one of "
                        f"float32 (the FFT kernels), float64 (direct float64 kernel) or float16 / bfloat16 (computed in "
                        f"float32)")


class KernelSpectrum:
    """A weight tensor transformed for one plan (rows a2 + a6); reusable while the weight is unchanged.

    Read-only once built: the scratch area of the N-d passes is NOT part of it -- every forward call takes
    its own workspace from torch's (stream-ordered) allocator, so one spectrum can serve several streams."""

    __slots__ = ("plan", "buf")

    def __init__(self, plan, buf):
        self.plan, self.buf = plan, buf


def _same_device(**tensors) -> torch.device:
    """All tensors of one call must live on one device (else the kernels would be handed foreign pointers)."""
    dev = None
    for name, t in tensors.items():
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError(f"fft_conv_pytorch_amd: `{name}` is on {t.device} but the signal is on {dev}; "
                             f"all tensors of one call must be on the same device")
    return dev


def _device_index(dev: torch.device) -> int:
    return dev.index if dev.index is not None else torch.cuda.current_device()


def new_workspace(plan, device) -> Optional[Tensor]:
    """Scratch area of one forward / transform call of an N-d plan (None for 1-D plans)."""
    if not plan.workspace_bytes:
        return None
    return torch.empty(plan.workspace_bytes // 4, dtype=torch.float32, device=device)


def _plan_for(signal: Tensor, kernel: Tensor, bias, stride, padding, dilation, groups, padding_mode, tile_hint=None,
              transposed=False, output_padding=0):
    if tile_hint is None:   # debugging / tuning knob: force the FFT tile length
        tile_hint = int(os.environ.get("FFTCONV_TILE", "0"))
    n = signal.ndim - 2
    if n < 1 or n > 3:
        raise ValueError(f"fft_conv expects (batch, channels, *spatial) with 1-3 spatial dims, got shape {tuple(signal.shape)}")
    if kernel.ndim != signal.ndim:
        raise ValueError(f"kernel has {kernel.ndim} dims but signal has {signal.ndim}")
    padding_ = to_ntuple(padding, n=n)
    stride_ = to_ntuple(stride, n=n)
    dilation_ = to_ntuple(dilation, n=n)
    if padding_mode not in _native.PAD_MODES:
        raise ValueError(f"unknown padding_mode {padding_mode!r}; expected one of constant/zeros/reflect/replicate/circular")
    if not isinstance(groups, int) or groups < 1:
        raise ValueError(f"groups must be a positive int, got {groups!r}")
    output_padding_ = to_ntuple(output_padding, n=n)
    cin = int(signal.shape[1])
    if transposed:
        # kernel is (Cin, Cout/groups, *k) as in torch.nn.ConvTranspose{N}d (functional.py:109-114)
        cout = int(kernel.shape[1]) * groups
        if cin % groups or int(kernel.shape[0]) != cin:
            raise ValueError(
                f"channel mismatch: signal has {cin} channels, transposed kernel is {tuple(kernel.shape)} with "
                f"groups={groups} (need kernel.shape[0] == in_channels and in_channels % groups == 0)")
        if padding_mode not in ("constant", "zeros"):
            raise ValueError("fft_conv_transpose supports zero padding only")
    else:
        cout = int(kernel.shape[0])
        if cin % groups or cout % groups or int(kernel.shape[1]) * groups != cin:
            raise ValueError(
                f"channel mismatch: signal has {cin} channels, kernel is {tuple(kernel.shape)} with groups={groups} "
                f"(need kernel.shape[1] * groups == in_channels and out_channels % groups == 0)")
    if bias is not None and tuple(bias.shape) != (cout,):
        raise ValueError(f"bias must have shape ({cout},), got {tuple(bias.shape)}")
    dtype = signal.dtype if signal.dtype in _DTYPE_CODES else torch.float32
    key = (n, int(signal.shape[0]), cin, cout, groups,
           tuple(int(s) for s in signal.shape[2:]), tuple(int(k) for k in kernel.shape[2:]),
           tuple(int(s) for s in stride_), tuple(int(p) for p in padding_), tuple(int(d) for d in dilation_),
           _native.PAD_MODES[padding_mode], bias is not None, int(tile_hint), bool(transposed),
           tuple(int(o) for o in output_padding_), _DTYPE_CODES[dtype])
    # host-side validation is complete; only now touch the device library
    _require_gpu_f32("signal", signal, dtype)
    _require_gpu_f32("kernel", kernel, dtype)
    if bias is not None:
        _require_gpu_f32("bias", bias, dtype)
    dev = _same_device(signal=signal, kernel=kernel, bias=bias)
    index = _device_index(dev)
    plan = _native.lookup_plan(index, key)
    if plan is None:
        # the library allocates twiddle tables and work lists on the CURRENT HIP device and sizes the work
        # list for its CU count: create the plan with the tensors' device current
        with torch.cuda.device(index):
            plan = _native.get_plan(index, key)
    return plan


def transform_kernel(plan, kernel: Tensor) -> KernelSpectrum:
    """Kernel transform (dilate, zero-pad, FFT, conjugate) on the device; rows a2 + a6."""
    _require_gpu_f32("kernel", kernel, plan.dtype)
    kernel = kernel.detach().contiguous()
    if _device_index(kernel.device) != plan.device_index:
        raise ValueError(f"kernel is on {kernel.device} but the plan was made for cuda:{plan.device_index}")
    with torch.cuda.device(kernel.device):
        buf = torch.empty(max(plan.spectrum_bytes, 16) // 4, dtype=torch.float32, device=kernel.device)
        ws = new_workspace(plan, kernel.device)      # scratch of this call only
        stream = torch.cuda.current_stream(kernel.device).cuda_stream
        plan.transform_kernel(kernel.data_ptr(), buf.data_ptr(), ws.data_ptr() if ws is not None else None, stream)
    return KernelSpectrum(plan, buf)


def _launch_forward(signal: Tensor, spectrum: KernelSpectrum, bias_c: Optional[Tensor]) -> Tensor:
    plan = spectrum.plan
    out = torch.empty((signal.shape[0], plan.key[3]) + plan.out_spatial, dtype=plan.dtype, device=signal.device)
    ws = new_workspace(plan, signal.device)
    stream = torch.cuda.current_stream(signal.device).cuda_stream
    plan.forward(signal.data_ptr(), spectrum.buf.data_ptr(), bias_c.data_ptr() if bias_c is not None else None,
                 out.data_ptr(), ws.data_ptr() if ws is not None else None, stream)
    return out


def _forward_native(signal: Tensor, spectrum: KernelSpectrum, bias: Optional[Tensor]) -> Tensor:
    signal = signal.detach().contiguous()
    bias_c = bias.detach().contiguous() if bias is not None else None
    index = _device_index(signal.device)
    if signal.dtype != spectrum.plan.dtype:
        raise TypeError(f"signal is {signal.dtype} but the plan was made for {spectrum.plan.dtype}")
    if spectrum.plan.device_index != index or spectrum.buf.device != signal.device:
        raise ValueError(f"signal is on {signal.device} but the kernel spectrum / plan belong to "
                         f"cuda:{spectrum.plan.device_index}")
    if torch.cuda.current_device() == index:      # the common case: no device switch to pay for
        return _launch_forward(signal, spectrum, bias_c)
    with torch.cuda.device(signal.device):
        return _launch_forward(signal, spectrum, bias_c)


def fft_conv(
    signal: Tensor,
    kernel: Tensor,
    bias: Tensor = None,
    stride: Union[int, Iterable[int]] = 1,
    padding: Union[int, Iterable[int]] = 0,
    dilation: Union[int, Iterable[int]] = 1,
    groups: int = 1,
    padding_mode: str = "constant",
) -> Tensor:
    """N-d (1/2/3) cross-correlation through FFTs, equal to ``torch.nn.functional.conv{N}d``.

    Args and result as in the reference (functional.py:19-42): ``signal`` is
    (B, Cin, *spatial), ``kernel`` is (Cout, Cin/groups, *k), ``bias`` is (Cout,)
    or None; ``stride``/``padding``/``dilation`` are ints or per-axis iterables;
    ``padding_mode`` is one of constant | reflect | replicate | circular.
    The result is a fresh contiguous (B, Cout, *out) float32 tensor on the
    input's device.  Unlike the reference, a kernel larger than the padded
    input raises ``ValueError`` (torch's behaviour) instead of returning a
    wrongly shaped tensor.
    """
    return _fft_conv_impl(signal, kernel, bias, stride, padding, dilation, groups, padding_mode, None)


_LOW_PRECISION = (torch.float16, torch.bfloat16)


def _string_padding(padding: str, kernel: Tensor, stride, dilation, n: int):
    """``padding='valid' | 'same'`` as in torch.nn.functional.conv{N}d (the reference rejects strings: a str is
    Iterable for its to_ntuple).  Returns (symmetric padding per axis, leading output samples to
    drop per axis): 'same' needs d*(k-1) padded samples per axis, split floor/ceil like torch; an odd total is
    run with the larger half on both sides and the surplus leading output sample dropped."""
    if padding == "valid":
        return (0,) * n, (0,) * n
    if padding != "same":
        raise ValueError(f"invalid padding string {padding!r}; expected 'same' or 'valid'")
    if any(s != 1 for s in to_ntuple(stride, n)):
        raise ValueError("padding='same' is not supported for strided convolutions")
    total = [d * (int(k) - 1) for d, k in zip(to_ntuple(dilation, n), kernel.shape[2:])]
    return tuple(t - t // 2 for t in total), tuple(t % 2 for t in total)


def _needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _fft_conv_impl(signal, kernel, bias, stride, padding, dilation, groups, padding_mode, spectrum, plan=None):
    """Shared by the functional and the modules; ``spectrum`` is an optional cached kernel transform and
    ``plan`` the plan the caller already looked up (and validated) for exactly these arguments."""
    if isinstance(padding, str) and signal.ndim >= 3:
        n = signal.ndim - 2
        pads, drop = _string_padding(padding, kernel, stride, dilation, n)
        out = _fft_conv_impl(signal, kernel, bias, stride, pads, dilation, groups, padding_mode, spectrum, plan)
        if any(drop):
            # (a fresh contiguous tensor, as documented and as torch's convolutions return: a view would pin the
            # larger buffer and break downstream .view() calls)
            out = out[(slice(None), slice(None)) + tuple(slice(d, None) for d in drop)].contiguous()
        return out
    if signal.dtype in _LOW_PRECISION and kernel.dtype == signal.dtype and (bias is None or bias.dtype == signal.dtype):
        # half-precision tensors in, half-precision tensor out; the arithmetic is the fp32 path (one cast pass each way)
        out = _fft_conv_impl(signal.float(), kernel.float(), None if bias is None else bias.float(), stride, padding,
                             dilation, groups, padding_mode, None, None)
        return out.to(signal.dtype)
    if _needs_grad(signal, kernel, bias):
        from .autograd import FFTConvFunction        # backward built from the same kernels (row N1)
        n = signal.ndim - 2
        return FFTConvFunction.apply(signal, kernel, bias, to_ntuple(stride, n), to_ntuple(padding, n),
                                     to_ntuple(dilation, n), groups, padding_mode, spectrum)
    if plan is None:
        plan = _plan_for(signal, kernel, bias, stride, padding, dilation, groups, padding_mode)
    if spectrum is None or spectrum.plan is not plan:
        spectrum = transform_kernel(plan, kernel)   # the reference also re-transforms per call (functional.py:71)
    return _forward_native(signal, spectrum, bias)


def fft_conv_transpose(
    signal: Tensor,
    kernel: Tensor,
    bias: Tensor = None,
    stride: Union[int, Iterable[int]] = 1,
    padding: Union[int, Iterable[int]] = 0,
    output_padding: Union[int, Iterable[int]] = 0,
    dilation: Union[int, Iterable[int]] = 1,
    groups: int = 1,
) -> Tensor:
    """N-d transposed convolution through FFTs, equal to ``torch.nn.functional.conv_transpose{N}d``
    (reference: functional.py:92-176).

    ``kernel`` is (Cin, Cout/groups, *k).  The same HIP kernels as ``fft_conv`` run it: the stride
    becomes a zero-spread of the input folded into the load index map, the kernel flip and the
    in/out channel swap are folded into the kernel transform, padding / output_padding only move
    the window of kept samples -- no intermediate tensor is materialised.
    """
    return _fft_conv_transpose_impl(signal, kernel, bias, stride, padding, output_padding, dilation, groups, None)


def _fft_conv_transpose_impl(signal, kernel, bias, stride, padding, output_padding, dilation, groups, spectrum,
                             plan=None):
    """Shared by the functional and the transposed modules (``spectrum`` / ``plan``: see ``_fft_conv_impl``)."""
    if signal.dtype in _LOW_PRECISION and kernel.dtype == signal.dtype and (bias is None or bias.dtype == signal.dtype):
        # half-precision tensors: fp32 arithmetic, one cast pass each way (as the forward op)
        out = _fft_conv_transpose_impl(signal.float(), kernel.float(), None if bias is None else bias.float(), stride,
                                       padding, output_padding, dilation, groups, None, None)
        return out.to(signal.dtype)
    if _needs_grad(signal, kernel, bias):
        from .autograd import FFTConvTransposeFunction     # differentiable like the reference's op graph
        n = signal.ndim - 2
        return FFTConvTransposeFunction.apply(signal, kernel, bias, to_ntuple(stride, n), to_ntuple(padding, n),
                                              to_ntuple(output_padding, n), to_ntuple(dilation, n), groups, spectrum)
    if plan is None:
        plan = _plan_for(signal, kernel, bias, stride, padding, dilation, groups, "constant",
                         transposed=True, output_padding=output_padding)
    if spectrum is None or spectrum.plan is not plan:
        spectrum = transform_kernel(plan, kernel)
    return _forward_native(signal, spectrum, bias)


def complex_matmul(a: Tensor, b: Tensor, groups: int = 1) -> Tensor:
    """Grouped per-bin channel contraction ``einsum('bgi...,goi...->bgo...')`` (functional.py:11-16).

    In this implementation the contraction is fused into the convolution kernels
    (the "mix" step), so ``fft_conv`` never calls this function; it is kept for
    API compatibility and evaluates the same contraction on the tensors' device.
    """
    a_g = a.unflatten(1, [groups, a.size(1) // groups])
    b_g = b.unflatten(0, [groups, b.size(0) // groups])
    return torch.einsum("bgi...,goi...->bgo...", a_g, b_g).flatten(1, 2)
