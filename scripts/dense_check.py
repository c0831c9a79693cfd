#!/usr/bin/env python3
"""Parity and timing of the many-channel pipeline (dense1d.hpp: FFT -> MFMA GEMM per bin -> FFT) against the fused
kernels (FFTCONV_DENSE=0) and torch's direct convolution.  One JSON line per shape."""
import json
import os
import subprocess
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # batch, cin, cout, groups, L, k, kwargs
    (8, 64, 64, 1, 16384, 129, {}),
    (3, 24, 40, 1, 5000, 65, dict(padding=7, padding_mode="reflect")),
    (2, 32, 48, 2, 3000, 33, dict(dilation=3, padding=40, padding_mode="circular")),
    (1, 16, 16, 1, 1200, 200, dict(padding=3)),
    (5, 128, 96, 1, 4096, 257, dict(padding=128)),
    (16, 32, 32, 1, 8192, 65, {}),
    (4, 32, 32, 1, 65536, 1025, {}),           # 2048 tile
    (8, 64, 64, 8, 16384, 129, {}),          # 8 per group: stays on the fused kernel
]


def run_one(idx):
    import fft_conv_pytorch_amd as fca
    from fft_conv_pytorch_amd.functional import fft_conv
    B, ci, co, g, L, k, kw = SHAPES[idx]
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(idx)
    x = torch.randn(B, ci, L, generator=gen).to(dev)
    w = (torch.randn(co, ci // g, k, generator=gen) / (ci // g * k) ** 0.5).to(dev)
    b = torch.randn(co, generator=gen).to(dev)
    y = fft_conv(x, w, b, groups=g, **kw)
    kwt = {kk: v for kk, v in kw.items() if kk not in ("padding", "padding_mode")}
    pad = kw.get("padding", 0)
    mode = kw.get("padding_mode", "constant")
    xd = x.double()
    if mode != "constant":
        xd = F.pad(xd, (pad, pad), mode=mode)
        pad = 0
    ref = F.conv1d(xd, w.double(), b.double(), padding=pad, groups=g, **kwt)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    layer = fca.FFTConv1d(ci, co, k, groups=g, **kw).to(dev).eval()
    with torch.no_grad():
        for _ in range(3):
            layer(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            layer(x)
        e1.record()
        torch.cuda.synchronize()
    print(json.dumps({"shape": [B, ci, co, g, L, k, kw], "dense_env": os.environ.get("FFTCONV_DENSE", "1"),
                      "rel_err": err, "us": e0.elapsed_time(e1) * 1e3 / 20}))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_one(int(sys.argv[1]))
    else:
        for i in range(len(SHAPES)):
            for env in ("1", "0"):
                e = dict(os.environ, FFTCONV_DENSE=env)
                r = subprocess.run([sys.executable, __file__, str(i)], env=e, capture_output=True, text=True, timeout=300)
                out = [l for l in r.stdout.splitlines() if l.startswith("{")]
                print(out[0] if out else json.dumps({"shape": SHAPES[i][:6], "dense_env": env, "failed": r.stderr[-400:]}), flush=True)
