"""The planner's row-axis choice (overlap-save x tiles where they save >= 15 % of the points) against one full-length transform
(FFTCONV_XTILE=0 keeps the round-2 behaviour): graph-replayed module forward, us per call."""
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


# nd, B, C, size, k, padding
CASES = [(2, 16, 8, (512, 512), 7, 3), (2, 16, 8, (512, 512), 3, 1), (2, 16, 8, (512, 512), 31, 15), (2, 4, 8, (256, 256), 5, 2),
         (2, 8, 8, (1024, 1024), 5, 2), (2, 8, 8, (300, 300), 5, 0), (2, 8, 8, (200, 600), 9, 0), (2, 2, 8, (700, 700), 11, 5),
         (2, 32, 8, (150, 150), 3, 1), (3, 2, 8, (128, 128, 128), 5, 2), (3, 4, 8, (40, 90, 300), 3, 1), (3, 8, 8, (64, 64, 64), 3, 1)]
for nd, b, c, size, k, pad in CASES:
    x = torch.randn(b, c, *size, device=dev)
    res = {}
    for knob in (None, "0"):
        if knob is None:
            os.environ.pop("FFTCONV_XTILE", None)
        else:
            os.environ["FFTCONV_XTILE"] = knob
        _native.clear_plan_cache()
        cls = fca.FFTConv2d if nd == 2 else fca.FFTConv3d
        layer = cls(c, c, k, padding=pad).to(dev).eval()
        us = timed(lambda: layer(x))
        plan = layer.__dict__["_spectrum_cache"][1].plan
        res["auto" if knob is None else "single"] = (round(us, 1), plan.layout[:3])
    print(json.dumps({"shape": f"{nd}-D B{b} {c}ch {'x'.join(map(str, size))} k{k} pad{pad}", **{k2: v for k2, v in res.items()}}), flush=True)
os.environ.pop("FFTCONV_XTILE", None)
