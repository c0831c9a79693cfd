"""A/B per shape: 2-D forward with the thread-per-sequence column pass (FFTCONV_PLANES=1: rows_r2c / colz / rows_c2r, rows as they
are) against the transposing passes with the LDS column pass (FFTCONV_PLANES=0).  Graph-replayed module forward, us per call."""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import fft_conv_pytorch_amd as fca
from fft_conv_pytorch_amd import _native

dev = "cuda:0"
SHAPES = [(16, 8, 8, 512, k) for k in (3, 5, 7, 9, 15, 23, 31)] + [(4, 8, 8, 256, 15), (4, 8, 8, 256, 3), (8, 4, 4, 128, 5),
                                                                  (2, 8, 8, 1024, 7), (32, 16, 16, 128, 3)]


def timed(layer, x, iters=60):
    with torch.no_grad():
        for _ in range(3):
            layer(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                layer(x)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters // 10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters // 10 * 10)


for b, ci, co, s, k in SHAPES:
    x = torch.randn(b, ci, s, s, device=dev)
    res = {}
    for knob in ("1", "0"):
        os.environ["FFTCONV_PLANES"] = knob
        _native.clear_plan_cache()
        layer = fca.FFTConv2d(ci, co, k).to(dev).eval()
        res[knob] = round(timed(layer, x), 1)
    print(json.dumps({"shape": f"B{b} {ci}->{co} {s}^2 k{k}", "colz_us": res["1"], "fusedc_us": res["0"]}), flush=True)
