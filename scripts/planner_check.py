#!/usr/bin/env python3
"""Forward time of a few 8->8 1-D shapes under the flavour forced by FFTCONV_TILE / FFTCONV_PERS (or the
planner's own choice when unset): used to check the planner's table (fc_api.cpp: choose_fast_path)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_conv_pytorch_amd import FFTConv1d  # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [(32, 32768, 512), (8, 262144, 129), (2, 1 << 20, 1025), (64, 4096, 65), (4, 16384, 33), (16, 65536, 257),
          (3, 100000, 700), (128, 2048, 200)]


def timed(fn, iters=30):
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


tag = f"TILE={os.environ.get('FFTCONV_TILE', '-')} PERS={os.environ.get('FFTCONV_PERS', '-')}"
out = []
for B, L, K in SHAPES:
    x = torch.randn(B, 8, L, device=dev)
    try:
        layer = FFTConv1d(8, 8, K).to(dev).eval()
        with torch.no_grad():
            layer(x)
            tile = layer.__dict__["_spectrum_cache"][1].plan.tile
            out.append(f"{timed(lambda: layer(x)):8.1f}({tile})")
    except Exception as exc:  # noqa: BLE001
        out.append(f"{'n/a':>8s}      ")
print(f"{tag:22s}" + " ".join(out), flush=True)
