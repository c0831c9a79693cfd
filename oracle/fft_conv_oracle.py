"""CPU oracle for the forward FFT-convolution hot path.  TEST INFRASTRUCTURE ONLY.

This module is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The shipped path (``fft_conv_pytorch_amd``) never imports anything
from ``oracle/`` and raises if its HIP library is missing.

It restates, independently, the algorithm of the reference's forward path
(``/root/reference/fft_conv_pytorch/functional.py:19-89``):

    a1  hyper-parameter normalisation ............ utils.py:4-20, functional.py:45-47
    a2  kernel dilation by zero stuffing ......... functional.py:49-57
    a3  signal padding (4 modes) ................. functional.py:60-62
    a4  even FFT extent per spatial dim .......... functional.py:66
    a5  real FFT of the signal ................... functional.py:70
    a6  real FFT of the kernel, conjugated ....... functional.py:71
    a7  grouped per-bin channel contraction ...... functional.py:11-16
    a8  inverse real FFT ......................... functional.py:68,74-75
    a9  valid window + stride decimation ......... functional.py:76-82
    a10 bias ..................................... functional.py:85-87

The arithmetic itself lives in a third-party dependency of the reference,
``torch`` (``torch>=1.8`` in /root/reference/setup.py:33; this image ships
torch 2.10.0): ``torch.fft.rfftn/irfftn`` + ``torch.einsum``.  Two backends are
provided:

* ``fft_conv_oracle_torch``  -- the same op sequence on torch CPU tensors (this
  is what the reference executes on a CPU; also used as the timed CPU baseline,
  ``cpu_baseline.kind == "port"``).
* ``fft_conv_oracle_numpy``  -- the same pipeline on numpy (pocketfft), in the
  dtype of the input or in float64.
* ``direct_conv_float64``    -- an FFT-free float64 sliding-window
  cross-correlation used as an independent cross-check (this is the role
  ``torch.nn.functional.conv{N}d`` plays in the reference's own tests,
  /root/reference/tests/test_functional.py:56-59).

Parity pinning: ``oracle/make_golden.py`` imports the real reference from
``/root/reference`` (in the build container only) and stores its outputs in
``tests/golden/*.npz``; ``tests/test_oracle.py`` checks every backend of this
file against those vectors.  Parity is therefore PINNED.
"""
from __future__ import annotations

import collections.abc
import itertools
from typing import Sequence, Tuple

import numpy as np

PAD_MODES = ("constant", "reflect", "replicate", "circular")


# --------------------------------------------------------------------------- a1
def ntuple(value, n: int) -> Tuple[int, ...]:
    """int-or-iterable -> tuple of length n (reference: utils.py:4-20).

    Mirrors the reference's behaviour including the error text and the fact
    that *any* iterable (a ``str`` too) is expanded element-wise.
    """
    if isinstance(value, collections.abc.Iterable):
        items = tuple(value)
        if len(items) != n:
            raise ValueError(f"Cannot cast tuple of length {len(items)} to length {n}.")
        return items
    return (value,) * n


def output_extent(size: int, k: int, stride: int, pad: int, dil: int) -> int:
    """Number of outputs along one axis (functional.py:76-82 slice arithmetic)."""
    span = size + 2 * pad - ((k - 1) * dil + 1)
    return span // stride + 1 if span >= 0 else 0


# ------------------------------------------------------------------ numpy path
def _stuff_zeros(kernel: np.ndarray, dilation: Sequence[int]) -> np.ndarray:
    """a2: place taps every ``d`` samples (functional.py:49-57)."""
    if all(d == 1 for d in dilation):
        return kernel
    spatial = [(k - 1) * d + 1 for k, d in zip(kernel.shape[2:], dilation)]
    out = np.zeros(kernel.shape[:2] + tuple(spatial), dtype=kernel.dtype)
    index = (slice(None), slice(None)) + tuple(slice(None, None, d) for d in dilation)
    out[index] = kernel
    return out


_NP_PAD = {"constant": "constant", "reflect": "reflect", "replicate": "edge", "circular": "wrap"}


def _pad_signal(signal: np.ndarray, padding: Sequence[int], mode: str) -> np.ndarray:
    """a3: symmetric padding of every spatial axis (functional.py:60-62)."""
    if all(p == 0 for p in padding):
        return signal
    if mode not in _NP_PAD:
        raise ValueError(f"unknown padding_mode {mode!r}")
    widths = [(0, 0), (0, 0)] + [(p, p) for p in padding]
    return np.pad(signal, widths, mode=_NP_PAD[mode])


def fft_conv_oracle_numpy(signal, kernel, bias=None, stride=1, padding=0, dilation=1,
                          groups=1, padding_mode="constant", compute_dtype=None):
    """numpy restatement of functional.py:19-89.  Arrays in, array out."""
    signal = np.asarray(signal)
    kernel = np.asarray(kernel)
    if compute_dtype is not None:
        signal = signal.astype(compute_dtype)
        kernel = kernel.astype(compute_dtype)
    nd = signal.ndim - 2
    pad = ntuple(padding, nd)
    strd = ntuple(stride, nd)
    dil = ntuple(dilation, nd)

    kernel = _stuff_zeros(kernel, dil)                       # a2
    signal = _pad_signal(signal, pad, padding_mode)          # a3
    extent = [(s + 1) // 2 * 2 for s in signal.shape[2:]]   # a4
    axes = tuple(range(-nd, 0))

    sig_f = np.fft.rfftn(signal, s=extent, axes=axes)        # a5
    ker_f = np.conj(np.fft.rfftn(kernel, s=extent, axes=axes))  # a6

    b, c_in = signal.shape[:2]
    c_out = kernel.shape[0]
    sig_g = sig_f.reshape((b, groups, c_in // groups) + sig_f.shape[2:])
    ker_g = ker_f.reshape((groups, c_out // groups, kernel.shape[1]) + ker_f.shape[2:])
    prod = np.einsum("bgi...,goi...->bgo...", sig_g, ker_g)  # a7
    prod = prod.reshape((b, c_out) + prod.shape[3:])

    full = np.fft.irfftn(prod, s=extent, axes=axes)          # a8
    window = (slice(None), slice(None)) + tuple(
        slice(0, signal.shape[2 + i] - kernel.shape[2 + i] + 1, strd[i]) for i in range(nd)
    )
    out = full[window]                                       # a9
    if bias is not None:                                     # a10
        out = out + np.asarray(bias).reshape((1, -1) + (1,) * nd).astype(out.dtype)
    return np.ascontiguousarray(out.astype(signal.dtype, copy=False))


def direct_conv_float64(signal, kernel, bias=None, stride=1, padding=0, dilation=1,
                        groups=1, padding_mode="constant"):
    """FFT-free float64 cross-correlation (independent cross-check).

    out[b,o,t] = bias[o] + sum_{i,k} x_pad[b, g*Ci/g + i, t*stride + k*dil] * w[o,i,k]
    which is what ``torch.nn.functional.conv{N}d`` computes and what the
    reference's tests compare against (tests/test_functional.py:56-59).
    """
    x = np.asarray(signal, dtype=np.float64)
    w = np.asarray(kernel, dtype=np.float64)
    nd = x.ndim - 2
    pad = ntuple(padding, nd)
    strd = ntuple(stride, nd)
    dil = ntuple(dilation, nd)
    x = _pad_signal(x, pad, padding_mode)
    b, c_in = x.shape[:2]
    c_out, cig = w.shape[:2]
    cog = c_out // groups
    outs = [output_extent(x.shape[2 + i], w.shape[2 + i], strd[i], 0, dil[i]) for i in range(nd)]
    y = np.zeros((b, c_out) + tuple(outs), dtype=np.float64)
    for taps in itertools.product(*[range(k) for k in w.shape[2:]]):
        window = tuple(
            slice(taps[i] * dil[i], taps[i] * dil[i] + (outs[i] - 1) * strd[i] + 1, strd[i])
            for i in range(nd)
        )
        xs = x[(slice(None), slice(None)) + window]          # (B, Ci, *outs)
        wt = w[(slice(None), slice(None)) + taps]            # (Co, Ci/g)
        for g in range(groups):
            y[:, g * cog:(g + 1) * cog] += np.einsum(
                "bi...,oi->bo...", xs[:, g * cig:(g + 1) * cig], wt[g * cog:(g + 1) * cog])
    if bias is not None:
        y += np.asarray(bias, dtype=np.float64).reshape((1, -1) + (1,) * nd)
    return y


# ------------------------------------------------------------------ torch path
def fft_conv_oracle_torch(signal, kernel, bias=None, stride=1, padding=0, dilation=1,
                          groups=1, padding_mode="constant"):
    """Same op sequence as the reference on torch CPU tensors (functional.py:19-89).

    torch tensors in, contiguous torch tensor out.  This is what the reference
    runs on a CPU and is the function timed for ``cpu_baseline`` in bench.py.
    """
    import torch
    import torch.nn.functional as F

    nd = signal.dim() - 2
    pad = ntuple(padding, nd)
    strd = ntuple(stride, nd)
    dil = ntuple(dilation, nd)
    axes = tuple(range(2, 2 + nd))

    if any(d != 1 for d in dil):                             # a2
        wide = [(k - 1) * d + 1 for k, d in zip(kernel.shape[2:], dil)]
        stuffed = torch.zeros(tuple(kernel.shape[:2]) + tuple(wide), dtype=kernel.dtype)
        stuffed[(Ellipsis,) + tuple(slice(None, None, d) for d in dil)] = kernel
        kernel = stuffed
    if any(p != 0 for p in pad):                             # a3
        flat = []
        for p in reversed(pad):
            flat += [p, p]
        signal = F.pad(signal, flat, mode=padding_mode)
    extent = [s + (s & 1) for s in signal.shape[2:]]         # a4

    sig_f = torch.fft.rfftn(signal, s=extent, dim=axes)      # a5
    ker_f = torch.fft.rfftn(kernel, s=extent, dim=axes).conj()  # a6
    b, c_in = signal.shape[:2]
    c_out, cig = kernel.shape[:2]
    sig_g = sig_f.reshape(b, groups, c_in // groups, *sig_f.shape[2:])
    ker_g = ker_f.reshape(groups, c_out // groups, cig, *ker_f.shape[2:])
    prod = torch.einsum("bgi...,goi...->bgo...", sig_g, ker_g)  # a7
    prod = prod.reshape(b, c_out, *prod.shape[3:])
    full = torch.fft.irfftn(prod, s=extent, dim=axes)        # a8
    window = (slice(None), slice(None)) + tuple(
        slice(0, signal.shape[2 + i] - kernel.shape[2 + i] + 1, strd[i]) for i in range(nd))
    out = full[window]                                       # a9
    if bias is not None:                                     # a10
        out = out + bias.reshape(1, -1, *([1] * nd))
    return out.contiguous()


# ------------------------------------------------------------- transposed convolution (row N2)
def transpose_output_extent(size: int, k: int, stride: int, pad: int, dil: int, out_pad: int) -> int:
    """(s - 1) * stride - 2 p + d (k - 1) + output_padding + 1   (functional.py:144-154)."""
    return (size - 1) * stride - 2 * pad + dil * (k - 1) + out_pad + 1


def fft_conv_transpose_oracle_torch(signal, kernel, bias=None, stride=1, padding=0, output_padding=0,
                                    dilation=1, groups=1):
    """Restatement of the reference's transposed convolution (functional.py:92-176) on torch CPU:
    flip the kernel and swap its in/out channels inside every group (:109-114), zero-stuff it for
    dilation (:115-124), spread the signal over a zero grid with the stride and a leading offset of
    K_d - 1 (:126-139), run the forward FFT pipeline on an even extent >= s_ + k - 1 (:143,155-162)
    and keep output_shape samples starting at the padding (:163-169); bias last (:172-174)."""
    import torch

    nd = signal.dim() - 2
    pad = ntuple(padding, nd)
    opad = ntuple(output_padding, nd)
    strd = ntuple(stride, nd)
    dil = ntuple(dilation, nd)
    axes = tuple(range(2, 2 + nd))
    cin, cog = kernel.shape[:2]

    w = torch.flip(kernel, dims=axes)                                   # spatial flip
    w = w.reshape(groups, cin // groups, cog, *kernel.shape[2:]).transpose(1, 2)
    w = w.reshape(groups * cog, cin // groups, *kernel.shape[2:])        # (Cout, Cin/g, *k)
    if any(d != 1 for d in dil):
        wide = [(k - 1) * d + 1 for k, d in zip(w.shape[2:], dil)]
        stuffed = torch.zeros(tuple(w.shape[:2]) + tuple(wide), dtype=w.dtype)
        stuffed[(Ellipsis,) + tuple(slice(None, None, d) for d in dil)] = w
        w = stuffed
    kd = w.shape[2:]
    grid = [(s - 1) * t + 1 + (k - 1) for s, k, t in zip(signal.shape[2:], kd, strd)]
    spread = torch.zeros(tuple(signal.shape[:2]) + tuple(grid), dtype=signal.dtype)
    spread[(Ellipsis,) + tuple(slice(k - 1, None, t) for k, t in zip(kd, strd))] = signal
    extent = [(s + k) // 2 * 2 for s, k in zip(grid, kd)]
    out_shape = [transpose_output_extent(s, k, t, p, d, o)
                 for s, k, t, p, d, o in zip(signal.shape[2:], kernel.shape[2:], strd, pad, dil, opad)]

    sig_f = torch.fft.rfftn(spread, s=extent, dim=axes)
    ker_f = torch.fft.rfftn(w, s=extent, dim=axes).conj()
    b = signal.shape[0]
    sig_g = sig_f.reshape(b, groups, cin // groups, *sig_f.shape[2:])
    ker_g = ker_f.reshape(groups, cog, cin // groups, *ker_f.shape[2:])
    prod = torch.einsum("bgi...,goi...->bgo...", sig_g, ker_g).reshape(b, groups * cog, *sig_f.shape[2:])
    full = torch.fft.irfftn(prod, s=extent, dim=axes)
    window = (slice(None), slice(None)) + tuple(slice(p, p + n) for p, n in zip(pad, out_shape))
    out = full[window]
    if bias is not None:
        out = out + bias.reshape(1, -1, *([1] * nd))
    return out.contiguous()


def rel_err(y, y_ref) -> float:
    """max|y - y_ref| / max|y_ref| -- the parity measure of BASELINE.md section 3."""
    y = np.asarray(y, dtype=np.float64)
    y_ref = np.asarray(y_ref, dtype=np.float64)
    denom = np.max(np.abs(y_ref))
    return float(np.max(np.abs(y - y_ref)) / (denom if denom > 0 else 1.0))
