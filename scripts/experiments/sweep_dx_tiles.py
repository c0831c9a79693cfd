"""Input-gradient plans (transposed convolution on dY) of the 2-D / 3-D BASELINE shapes: every (x tile, y/z tile) choice forced in
turn (FFTCONV_XTILE / FFTCONV_YTILE at plan creation, tile_hint for the outermost axis), graph-replayed, us per call."""
import itertools
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from fft_conv_pytorch_amd import functional as F_, _native

dev = "cuda:0"


def timed(fn, iters=30):
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


CASES = [("cfgB dX", 2, 16, 8, 512, 31, [0, 256, 512], [0], [0, 128, 256, 512, 1024]),
         ("cfgC dX", 3, 8, 8, 64, 9, [0, 64], [0, 64], [0, 64, 128]),
         ("2-D B16 512^2 k7 dX", 2, 16, 8, 512, 7, [0, 256, 512], [0], [0, 128, 256, 512])]
for name, nd, b, c, s, k, xts, yts, hints in CASES:
    lo = s - k + 1
    gy = torch.randn(b, c, *([lo] * nd), device=dev)
    w = torch.randn(c, c, *([k] * nd), device=dev)
    one = (1,) * nd
    for xt, yt, hint in itertools.product(xts, yts, hints):
        os.environ["FFTCONV_XTILE"] = str(xt)
        os.environ["FFTCONV_YTILE"] = str(yt)
        _native.clear_plan_cache()
        try:
            plan = F_._plan_for(gy, w, None, one, (0,) * nd, one, 1, "constant", tile_hint=hint, transposed=True, output_padding=(0,) * nd)
            spec = F_.transform_kernel(plan, w)
            us = timed(lambda: F_._forward_native(gy, spec, None))
            print(json.dumps({"case": name, "xtile": xt, "ytile": yt, "tile_hint": hint, "plan_tile": plan.tile, "layout": plan.layout[:4], "us": round(us, 1)}), flush=True)
        except (NotImplementedError, ValueError) as exc:
            print(json.dumps({"case": name, "xtile": xt, "ytile": yt, "tile_hint": hint, "error": str(exc)[:80]}), flush=True)
os.environ.pop("FFTCONV_XTILE"); os.environ.pop("FFTCONV_YTILE")
