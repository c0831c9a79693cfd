#!/usr/bin/env python3
"""Depthwise (groups == channels) long convolutions: parity against the rocFFT formulation and timing of both."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_conv_pytorch_amd import FFTConv1d  # noqa: E402

dev = "cuda"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def rfft_dw(x, w, b, pad):
    x = F.pad(x, [pad, pad])
    n = x.shape[-1]
    y = torch.fft.irfft(torch.fft.rfft(x, n=n) * torch.fft.rfft(w[:, 0], n=n).conj().unsqueeze(0), n=n)[..., : n - w.shape[-1] + 1]
    return y + b.view(1, -1, 1)


for (B, C, L, K, pad) in [(8, 64, 65536, 1025, 0), (8, 256, 16384, 512, 100), (4, 512, 8192, 2048, 0), (16, 128, 32768, 129, 64),
                          (3, 24, 5000, 33, 5), (1, 8, 100000, 257, 0)]:
    torch.manual_seed(C)
    x = torch.randn(B, C, L, device=dev)
    m = FFTConv1d(C, C, K, groups=C, padding=pad).to(dev).eval()
    with torch.no_grad():
        y = m(x)
        ref = rfft_dw(x, m.weight, m.bias, pad)
        err = ((y - ref).norm() / ref.norm()).item()
        t = timed(lambda: m(x))
        tr = timed(lambda: rfft_dw(x, m.weight, m.bias, pad), 5)
    alg = 4 * (x.numel() + y.numel())
    print(f"depthwise B{B} C{C} L{L} K{K} pad{pad}: ours {t:.1f} us ({alg / t / 1e3 / 8000:.3f} of 8 TB/s, tile "
          f"{m.__dict__['_spectrum_cache'][1].plan.tile}), torch.fft {tr:.1f} us, rel err {err:.1e}", flush=True)
    assert err < 1e-4
