"""Backward of ``fft_conv`` built from the same HIP kernels.

The reference has no custom backward: autograd differentiates its rfftn/einsum/irfftn graph
(/root/reference/tests/test_functional.py:111-117 pins dW and db).  A C-ABI forward is opaque to
autograd, so the three gradients are expressed as convolutions the native library already runs:

    dX = conv_transpose(dY, W)             -> ``fc_forward`` on a transposed plan
    dW = correlate(X, dY) over the batch   -> 1-D, <= 64 channels per group: ``fc_wgrad1d`` (cross-spectra
                                              accumulated on chip); 2-D / 3-D: ``fc_wgrad_nd`` (the role-swapped
                                              convolution below, run by the library on the tensors as they lie);
                                              otherwise ``fc_forward`` with the roles of
                                              batch and channels swapped: signal' = X^T (Cin/g, B, *S), kernel' =
                                              dY^T (Cout/g, B, *Lout), dilation' = stride, stride' = dilation;
                                              all channel groups ride the group axis of that one call, long 1-D
                                              rows are cut into chunks that ride it too
    db = sum of dY over batch and space    -> a plain reduction

torch is used for data movement only (transposes, unfold views, the final chunk sum and db).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import functional as F_

_DW_TILE = 2048      # FFT tile the dW correlation is sized for: the largest one that also fits the
                     # accumulating (> 8 channels', i.e. batch > 8) mode of the fused kernel in LDS


_BWD_PLANS: dict = {}      # (shapes, hyper-parameters, device) -> (transposed plan of dX, ...): skips the argument
                           # normalisation and descriptor lookup of the functional on every backward call


def _pad_adjoint(dxp: Tensor, in_spatial, padding, mode: str) -> Tensor:
    """Adjoint of ``F.pad(x, padding, mode)`` for reflect / replicate / circular: fold the border samples of the
    gradient w.r.t. the padded signal back onto their sources, one axis after the other (the padding is separable).
    A handful of slice-adds on border planes; the round-2 version back-propagated through torch's own pad with a
    zeros probe per call."""
    t = dxp
    for i, (n, p) in enumerate(zip(in_spatial, padding)):
        ax = 2 + i
        if p == 0:
            continue
        core = t.narrow(ax, p, n).clone()
        left, right = t.narrow(ax, 0, p), t.narrow(ax, n + p, p)
        if mode == "reflect":          # padded j < p reads source p - j; padded n + p + j reads source n - 2 - j
            core.narrow(ax, 1, p).add_(left.flip(ax))
            core.narrow(ax, n - 1 - p, p).add_(right.flip(ax))
        elif mode == "replicate":
            core.narrow(ax, 0, 1).add_(left.sum(dim=ax, keepdim=True))
            core.narrow(ax, n - 1, 1).add_(right.sum(dim=ax, keepdim=True))
        elif mode == "circular":       # padded j < p reads source n - p + j; padded n + p + j reads source j
            core.narrow(ax, n - p, p).add_(left)
            core.narrow(ax, 0, p).add_(right)
        else:
            raise ValueError(f"no adjoint for padding mode {mode!r}")
        t = core
    return t.contiguous()


def _grad_input(grad: Tensor, weight: Tensor, in_spatial, stride, padding, dilation, groups, padding_mode) -> Tensor:
    n = grad.ndim - 2
    key = ("dx", tuple(grad.shape), tuple(weight.shape), tuple(in_spatial), stride, padding, dilation, groups, padding_mode,
           grad.device, grad.dtype)
    plan = _BWD_PLANS.get(key)
    if plan is None:
        pad_t = padding if padding_mode == "constant" else (0,) * n
        full = tuple(s + 2 * (p if padding_mode != "constant" else 0) for s, p in zip(in_spatial, padding))
        out_pad = []
        for i in range(n):
            base = (grad.shape[2 + i] - 1) * stride[i] - 2 * pad_t[i] + dilation[i] * (weight.shape[2 + i] - 1) + 1
            out_pad.append(full[i] - base)
        # conv weight (Cout, Cin/g, *k) is exactly the transposed-conv layout (Cin_t = Cout, Cout_t/g = Cin/g)
        plan = F_._plan_for(grad, weight, None, stride, pad_t, dilation, groups, "constant", transposed=True,
                            output_padding=tuple(out_pad))
        if len(_BWD_PLANS) > 256:
            _BWD_PLANS.clear()
        _BWD_PLANS[key] = plan
    dx = F_._forward_native(grad, F_.transform_kernel(plan, weight), None)
    if padding_mode == "constant":
        return dx
    return _pad_adjoint(dx, in_spatial, padding, padding_mode)


def _grad_weight_plans(x: Tensor, grad: Tensor, wshape, stride, padding, dilation, groups, padding_mode) -> Tensor:
    """dW through forward plans with the roles of batch and channels swapped, ALL channel groups in one call
    (they ride the group axis of the swapped convolution).  x: (B, Cin, *S), grad: (B, Cout, *Lout) -> (Cout, Cin/g, *k)."""
    n = x.ndim - 2
    b, g = x.shape[0], groups
    cog, cig, ksize = wshape[0] // groups, wshape[1], tuple(wshape[2:])
    sp, lo = tuple(x.shape[2:]), tuple(grad.shape[2:])
    kext = [(lo[i] - 1) * stride[i] + 1 for i in range(n)]
    kd0 = (ksize[0] - 1) * dilation[0] + 1
    # one overlap-save tile must hold kernel' (extent kext) plus the kd - 1 further samples that give the
    # k taps of dW; a longer gradient would leave the tile (almost) no valid window (V = T - kext + 1)
    max_ext = max(_DW_TILE - kd0 + 1, _DW_TILE // 4)
    if n == 1 and kext[0] > max_ext:
        # Long rows: kernel' (the gradient) is cut into chunks of C taps; chunk j only meets the signal
        # window [j*C*s - p, j*C*s - p + (C-1)*s + 1 + Kd - 1).  Chunks (and channel groups) ride the group axis.
        s, p, d = stride[0], padding[0], dilation[0]
        kd = kd0
        c_taps = max(1, (max_ext - 1) // s + 1)
        lout = lo[0]
        nchunk = (lout + c_taps - 1) // c_taps
        seg = (c_taps - 1) * s + kd                                  # signal samples one chunk needs
        # one padded copy of each operand, then one gather through an overlapping strided view
        need = (nchunk - 1) * c_taps * s + seg
        if padding_mode == "constant" or p == 0:
            xp = F.pad(x, [p, max(0, need - x.shape[-1] - p)])       # (B, Cin, >= need)
        else:
            xp = F.pad(x, [p, p], mode=padding_mode)
            if xp.shape[-1] < need:
                xp = F.pad(xp, [0, need - xp.shape[-1]])
        xp = xp.contiguous()
        sb, sc, sl = xp.stride()
        # signal' (Cig, [g, chunk, b], seg): batch' = the input channel inside its group, channels' = (group, chunk, batch)
        sig = xp.as_strided((cig, g, nchunk, b, seg), (sc, cig * sc, c_taps * s * sl, sb, sl)).reshape(cig, g * nchunk * b, seg)
        gp = F.pad(grad, [0, nchunk * c_taps - lout]).contiguous()   # (B, Cout, nchunk*C)
        gb, gc, gl = gp.stride()
        # kernel' ([g, chunk, o], b, C): out' = (group, chunk, output channel inside the group), in' = batch
        ker = gp.as_strided((g, nchunk, cog, b, c_taps), (cog * gc, c_taps * gl, gc, gb, gl)).reshape(g * nchunk * cog, b, c_taps)
        part = F_.fft_conv(sig, ker.contiguous(), None, stride=d, padding=0, dilation=s, groups=g * nchunk)
        part = part.reshape(cig, g, nchunk, cog, -1).sum(dim=2)      # (Cig, g, Cog, >= k)
        return part[..., : ksize[0]].permute(1, 2, 0, 3).reshape(g * cog, cig, ksize[0]).contiguous()
    # signal' (Cig, [g, b], *S), kernel' ([g, o], b, *Lout), groups' = g
    xt = x.reshape((b, g, cig) + sp).permute((2, 1, 0) + tuple(range(3, 3 + n))).reshape((cig, g * b) + sp).contiguous()
    gt = grad.reshape((b, g, cog) + lo).permute((1, 2, 0) + tuple(range(3, 3 + n))).reshape((g * cog, b) + lo).contiguous()
    key = ("dwp", tuple(xt.shape), tuple(gt.shape), stride, padding, dilation, g, padding_mode, xt.device, xt.dtype)
    plan = _BWD_PLANS.get(key)
    if plan is None:
        plan = F_._plan_for(xt, gt, None, dilation, padding, stride, g, padding_mode)
        if len(_BWD_PLANS) > 256:
            _BWD_PLANS.clear()
        _BWD_PLANS[key] = plan
    out = F_._forward_native(xt, F_.transform_kernel(plan, gt), None)      # (Cig, g*Cog, >= k per axis)
    index = (slice(None), slice(None)) + tuple(slice(0, k) for k in ksize)
    return out[index].transpose(0, 1).contiguous()                  # (g*Cog, Cig, *k)


def _grad_weight_native(x: Tensor, grad: Tensor, wshape, stride, padding, dilation, groups, padding_mode,
                        want_db: bool = False):
    """dW by ``fc_wgrad1d`` (cross-spectra accumulated on chip over batch and row); None when not covered.  With
    ``want_db`` the bias gradient rides the same launch where the kernel offers it (bin 0 of the gradient spectra):
    the slices come as rows [dW | db] and ONE reduction sums both.  Returns (dW, db or None)."""
    if x.ndim != 3 or x.dtype != torch.float32:      # (float64 runs the direct kernel: dW through forward plans)
        return None
    from . import _native
    if grad.device != x.device:
        raise ValueError(f"gradient is on {grad.device} but the signal is on {x.device}")
    key = ("dw", tuple(x.shape), tuple(wshape), stride, padding, dilation, groups, padding_mode, x.device)
    hit = _BWD_PLANS.get(key)
    if hit is None:
        desc = _native.conv_desc(1, x.shape[0], x.shape[1], wshape[0], groups, (x.shape[2],), (wshape[2],), stride, padding,
                                 dilation, _native.PAD_MODES[padding_mode])
        # the library sizes the launch for, and keeps its twiddle tables on, the CURRENT device
        with torch.cuda.device(x.device):
            slices = _native.wgrad1d_slices(desc)
            db_ok = bool(slices) and _native.wgrad1d_db_supported(desc)
        hit = (desc, slices, db_ok)
        if len(_BWD_PLANS) > 256:
            _BWD_PLANS.clear()
        _BWD_PLANS[key] = hit
    desc, slices, db_ok = hit
    if slices == 0:
        return None
    with_db = want_db and db_ok
    n_w = wshape[0] * wshape[1] * wshape[2]
    row = n_w + (wshape[0] if with_db else 0)
    x = x.contiguous()
    grad = grad.contiguous()
    with torch.cuda.device(x.device):
        part = torch.empty((slices, row), device=x.device, dtype=torch.float32)
        _native.wgrad1d_db(desc, x.data_ptr(), grad.data_ptr(), part.data_ptr(),
                           part.data_ptr() + 4 * n_w if with_db else None, row, slices,
                           torch.cuda.current_stream(x.device).cuda_stream)
    total = part[0] if slices == 1 else part.sum(dim=0)
    return total[:n_w].view(wshape), (total[n_w:] if with_db else None)


def _grad_weight_nd_native(x: Tensor, grad: Tensor, wshape, stride, padding, dilation, groups, padding_mode):
    """2-D / 3-D dW by ``fc_wgrad_nd``: the role-swapped convolution run by the library on the tensors as they are
    (x (B, Cin, *S) and dY (B, Cout, *Lout) are read, dW (Cout, Cin/g, *k) is written, in these layouts) -- no transposed
    copies, no crop, no reduction on the torch side.  None when not covered (1-D, float64, a shape the plan refuses)."""
    n = x.ndim - 2
    if n < 2 or x.dtype != torch.float32:
        return None
    from . import _native
    if grad.device != x.device:
        raise ValueError(f"gradient is on {grad.device} but the signal is on {x.device}")
    key = ("dwn", tuple(x.shape), tuple(wshape), stride, padding, dilation, groups, padding_mode, x.device)
    plan = _BWD_PLANS.get(key)
    if plan is None:
        desc = _native.conv_desc(n, x.shape[0], x.shape[1], wshape[0], groups, tuple(x.shape[2:]), tuple(wshape[2:]), stride,
                                 padding, dilation, _native.PAD_MODES[padding_mode])
        try:
            with torch.cuda.device(x.device):
                plan = _native.WgradPlan(desc)
        except NotImplementedError:
            plan = False                      # (remembered: the forward-plan route below takes this shape)
        if len(_BWD_PLANS) > 256:
            _BWD_PLANS.clear()
        _BWD_PLANS[key] = plan
    if plan is False:
        return None
    x = x.contiguous()
    grad = grad.contiguous()
    with torch.cuda.device(x.device):
        dw = torch.empty(tuple(wshape), device=x.device, dtype=torch.float32)
        spec = torch.empty(plan.spectrum_bytes, device=x.device, dtype=torch.uint8)
        ws = torch.empty(max(plan.workspace_bytes, 1), device=x.device, dtype=torch.uint8)
        plan.run(x.data_ptr(), grad.data_ptr(), dw.data_ptr(), spec.data_ptr(), ws.data_ptr(),
                 torch.cuda.current_stream(x.device).cuda_stream)
    return dw


def _grad_weight(x: Tensor, grad: Tensor, wshape, stride, padding, dilation, groups, padding_mode) -> Tensor:
    native = _grad_weight_native(x, grad, wshape, stride, padding, dilation, groups, padding_mode)
    if native is not None:
        return native[0]
    nd = _grad_weight_nd_native(x, grad, wshape, stride, padding, dilation, groups, padding_mode)
    if nd is not None:
        return nd
    return _grad_weight_plans(x, grad, wshape, stride, padding, dilation, groups, padding_mode)


class FFTConvFunction(torch.autograd.Function):
    """``fft_conv`` with gradients; ``spectrum`` is an optional pre-transformed kernel (module cache)."""

    @staticmethod
    def forward(ctx, signal: Tensor, kernel: Tensor, bias: Optional[Tensor], stride: Tuple[int, ...],
                padding: Tuple[int, ...], dilation: Tuple[int, ...], groups: int, padding_mode: str, spectrum):
        plan = F_._plan_for(signal, kernel, bias, stride, padding, dilation, groups, padding_mode)
        if spectrum is None or spectrum.plan is not plan:
            spectrum = F_.transform_kernel(plan, kernel)
        ctx.save_for_backward(signal, kernel)
        ctx.conf = (stride, padding, dilation, groups, padding_mode, bias is not None)
        return F_._forward_native(signal, spectrum, bias)

    @staticmethod
    def backward(ctx, grad: Tensor):
        signal, kernel = ctx.saved_tensors
        stride, padding, dilation, groups, padding_mode, has_bias = ctx.conf
        grad = grad.contiguous()
        d_signal = d_kernel = d_bias = None
        if ctx.needs_input_grad[0]:
            d_signal = _grad_input(grad, kernel.detach(), tuple(signal.shape[2:]), stride, padding, dilation, groups,
                                   padding_mode)
        want_db = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            native = _grad_weight_native(signal.detach(), grad, tuple(kernel.shape), stride, padding, dilation, groups,
                                         padding_mode, want_db)
            if native is not None:
                d_kernel, d_bias = native          # (db rode the weight-gradient launch where the kernel offers it)
            else:
                d_kernel = _grad_weight_nd_native(signal.detach(), grad, tuple(kernel.shape), stride, padding, dilation,
                                                  groups, padding_mode)
                if d_kernel is None:
                    d_kernel = _grad_weight_plans(signal.detach(), grad, tuple(kernel.shape), stride, padding, dilation,
                                                  groups, padding_mode)
        if want_db and d_bias is None:
            d_bias = grad.sum(dim=[0] + list(range(2, grad.ndim)))
        return d_signal, d_kernel, d_bias, None, None, None, None, None, None


class FFTConvTransposeFunction(torch.autograd.Function):
    """``fft_conv_transpose`` with gradients (the reference's transposed op is
    differentiable through its rfftn/einsum/irfftn graph and its tests pin dW and db:
    /root/reference/tests/test_functional_transpose.py:73-124).  With y = convT(x, W), W of shape
    (Cin, Cout/g, *k):

        dX = conv(dY, W)                  the ordinary convolution with the same hyper-parameters: a forward plan
                                          (W read as a conv weight with Cin outputs and Cout/g inputs per group)
        dW = the weight gradient of that  convolution, with dY as its signal and x as its output gradient
        db = sum of dY over batch and space
    """

    @staticmethod
    def forward(ctx, signal: Tensor, kernel: Tensor, bias: Optional[Tensor], stride: Tuple[int, ...],
                padding: Tuple[int, ...], output_padding: Tuple[int, ...], dilation: Tuple[int, ...], groups: int,
                spectrum):
        plan = F_._plan_for(signal, kernel, bias, stride, padding, dilation, groups, "constant",
                            transposed=True, output_padding=output_padding)
        if spectrum is None or spectrum.plan is not plan:
            spectrum = F_.transform_kernel(plan, kernel)
        ctx.save_for_backward(signal, kernel)
        ctx.conf = (stride, padding, dilation, groups, bias is not None)
        return F_._forward_native(signal, spectrum, bias)

    @staticmethod
    def backward(ctx, grad: Tensor):
        signal, kernel = ctx.saved_tensors
        stride, padding, dilation, groups, has_bias = ctx.conf
        grad = grad.contiguous()
        n = grad.ndim - 2
        in_spatial = tuple(signal.shape[2:])
        # extent of conv(dY, W) per axis: >= the input extent; larger only when output_padding >= stride
        conv_sp = tuple((grad.shape[2 + i] + 2 * padding[i] - dilation[i] * (kernel.shape[2 + i] - 1) - 1) // stride[i] + 1
                        for i in range(n))
        d_signal = d_kernel = d_bias = None
        if ctx.needs_input_grad[0]:
            dx = F_.fft_conv(grad, kernel.detach(), None, stride=stride, padding=padding, dilation=dilation,
                             groups=groups)
            if conv_sp != in_spatial:
                dx = dx[(slice(None), slice(None)) + tuple(slice(0, s) for s in in_spatial)].contiguous()
            d_signal = dx
        if ctx.needs_input_grad[1]:
            xg = signal.detach()
            if conv_sp != in_spatial:     # rows past the input contribute nothing: zero-extend (data movement)
                flat = []
                for i in reversed(range(n)):
                    flat += [0, conv_sp[i] - in_spatial[i]]
                xg = F.pad(xg, flat)
            d_kernel = _grad_weight(grad, xg, tuple(kernel.shape), stride, padding, dilation, groups, "constant")
        if has_bias and ctx.needs_input_grad[2]:
            d_bias = grad.sum(dim=[0] + list(range(2, grad.ndim)))
        return d_signal, d_kernel, d_bias, None, None, None, None, None, None
