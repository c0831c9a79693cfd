"""3-D problems whose padded planes exceed 64 x 64: the plane-major pipeline on 64 x 64 overlap-save tiles (default) against
the separable passes with the planner's x / y tiles (FFTCONV_PLANES=0), graph-replayed module forward, us per call."""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import fft_conv_pytorch_amd as fca
from fft_conv_pytorch_amd import _native

dev = "cuda:0"


def timed(fn, iters=20):
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


# B, C, size, k, pad
CASES = [(8, 8, (64, 64, 64), 3, 1), (8, 8, (64, 64, 64), 9, 4), (2, 8, (128, 128, 128), 5, 2), (4, 8, (40, 90, 300), 3, 1),
         (2, 8, (128, 128, 128), 9, 0), (8, 8, (32, 100, 100), 7, 0), (1, 8, (200, 200, 200), 5, 0), (8, 8, (64, 64, 64), 9, 0)]
for b, c, size, k, pad in CASES:
    x = torch.randn(b, c, *size, device=dev)
    res = {}
    for knob in ("1", "0"):
        os.environ["FFTCONV_PLANES"] = knob
        _native.clear_plan_cache()
        layer = fca.FFTConv3d(c, c, k, padding=pad).to(dev).eval()
        us = timed(lambda: layer(x))
        plan = layer.__dict__["_spectrum_cache"][1].plan
        res["planes" if knob == "1" else "separable"] = (round(us, 1), plan.layout[:3], plan.layout[7])
    print(json.dumps({"shape": f"B{b} {c}ch {'x'.join(map(str, size))} k{k} pad{pad}", **res}), flush=True)
os.environ.pop("FFTCONV_PLANES", None)
