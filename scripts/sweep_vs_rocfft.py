#!/usr/bin/env python3
"""Forward time of this library against two same-GPU yardsticks over a spread of shapes: the reference's
algorithm written with torch.fft (rocFFT underneath; functional.py:60-87 restated inline) and torch's direct
convolution (MIOpen).  One JSON line per shape."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_conv_pytorch_amd import FFTConv1d, FFTConv2d, FFTConv3d  # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [  # ndim, B, Cin, Cout, groups, size, k
    (1, 32, 8, 8, 1, 32768, 512),
    (1, 8, 64, 64, 1, 16384, 129),
    (1, 5, 128, 96, 1, 4096, 257),
    (1, 16, 32, 32, 1, 8192, 65),
    (1, 4, 32, 32, 1, 65536, 1025),
    (1, 16, 16, 16, 1, 4096, 33),
    (1, 64, 4, 4, 1, 8192, 2049),
    (2, 16, 8, 8, 1, 512, 31),
    (2, 8, 16, 16, 1, 256, 15),
    (2, 4, 32, 32, 1, 128, 7),
    (3, 8, 8, 8, 1, 64, 9),
    (3, 2, 16, 16, 1, 48, 5),
    (1, 8, 64, 64, 8, 1 << 20, 257),     # last: its 2 GB tensors disturb the allocator for what follows
]


def rfft_conv(x, w, b):
    nd = x.ndim - 2
    dims = tuple(range(2, 2 + nd))
    s = x.shape[2:]
    xf = torch.fft.rfftn(x, s=s, dim=dims)
    wf = torch.fft.rfftn(w, s=s, dim=dims)
    yf = torch.einsum("bi...,oi...->bo...", xf, wf.conj())
    y = torch.fft.irfftn(yf, s=s, dim=dims)
    crop = (slice(None), slice(None)) + tuple(slice(0, n - k + 1) for n, k in zip(s, w.shape[2:]))
    return y[crop] + b.view(1, -1, *([1] * nd))


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


first = int(os.environ.get("SWEEP_FIRST", "0"))
for nd, B, ci, co, g, size, k in SHAPES[first:]:
    torch.manual_seed(0)
    x = torch.randn(B, ci, *([size] * nd), device=dev)
    layer = {1: FFTConv1d, 2: FFTConv2d, 3: FFTConv3d}[nd](ci, co, k, groups=g, bias=True).to(dev)
    w, b = layer.weight.detach(), layer.bias.detach()
    row = {"shape": f"{nd}D B{B} {ci}->{co} g{g} size{size} k{k}"}
    with torch.no_grad():
        y = layer(x)
        row["ours_us"] = round(timed(lambda: layer(x), 20), 1)
        if g == 1:
            ref = rfft_conv(x, w, b)
            row["rel_err_vs_rocfft_path"] = float(((y - ref).norm() / ref.norm()).item())
            row["torch_fft_us"] = round(timed(lambda: rfft_conv(x, w, b), 10), 1)
        try:
            conv = getattr(F, f"conv{nd}d")
            row["miopen_us"] = round(timed(lambda: conv(x, w, b, groups=g), 3), 1)
        except Exception as exc:  # noqa: BLE001
            row["miopen_us"] = str(exc)[:60]
    n_out = y.numel()
    row["ours_GSamples_s"] = round(n_out / row["ours_us"] / 1e3, 1)
    alg = 4 * (x.numel() + w.numel() + co + n_out)
    row["ours_frac_of_8TBps"] = round(alg / row["ours_us"] / 1e3 / 8000, 4)
    print(json.dumps(row), flush=True)
    del x, layer, y
    torch.cuda.empty_cache()
