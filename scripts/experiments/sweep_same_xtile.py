"""'same'-padded 2-D / 3-D forward convolutions on power-of-two images (padded rows just past the power of two): the row
transform as one longer FFT against overlap-save x tiles (FFTCONV_XTILE), graph-replayed module forward, us per call."""
import itertools
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import fft_conv_pytorch_amd as fca
from fft_conv_pytorch_amd import _native

dev = "cuda:0"


def timed(fn, iters=30):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(5):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters // 5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters // 5 * 5)


CASES = [(2, 16, 8, 512, 7), (2, 16, 8, 512, 3), (2, 16, 8, 512, 31), (2, 4, 8, 256, 5), (2, 8, 8, 1024, 5), (3, 8, 8, 64, 3), (3, 2, 8, 128, 5)]
for nd, b, c, s, k in CASES:
    x = torch.randn(b, c, *([s] * nd), device=dev)
    for xt, planes in itertools.product((0, s // 2, s, s // 4), ("1", "0")):
        if xt and xt < 2 * k:
            continue
        os.environ["FFTCONV_XTILE"] = str(xt)
        os.environ["FFTCONV_PLANES"] = planes
        _native.clear_plan_cache()
        cls = fca.FFTConv2d if nd == 2 else fca.FFTConv3d
        layer = cls(c, c, k, padding=k // 2).to(dev).eval()
        try:
            us = timed(lambda: layer(x))
            plan = layer.__dict__["_spectrum_cache"][1].plan
            print(json.dumps({"shape": f"{nd}-D B{b} {c}ch {s}^{nd} k{k} same", "xtile": xt, "planes": planes, "layout": plan.layout, "us": round(us, 1)}), flush=True)
        except (NotImplementedError, ValueError) as exc:
            print(json.dumps({"shape": f"{nd}-D B{b} {s} k{k}", "xtile": xt, "error": str(exc)[:60]}), flush=True)
os.environ.pop("FFTCONV_XTILE"); os.environ.pop("FFTCONV_PLANES")
