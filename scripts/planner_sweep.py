#!/usr/bin/env python3
"""Planner regret on 8 -> 8 1-D shapes: every (tile, flavour) the planner may pick is forced in turn (FFTCONV_TILE /
FFTCONV_PERS, read at plan creation) and timed next to the planner's own choice.  Prints one JSON line per shape:
candidate times, the planner's pick, and its regret against the best forced candidate.
Usage: python scripts/planner_sweep.py > profiles/r03_planner_sweep.jsonl"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_conv_pytorch_amd import FFTConv1d, _native  # noqa: E402

dev = torch.device("cuda", 0)
# batch, length, taps: the metric shape, long rows at small batch, short rows at large batch, in-between kernels
SHAPES = [(32, 32768, 512), (4, 32768, 512), (8, 262144, 129), (2, 1 << 20, 1025), (1, 1 << 20, 257), (64, 4096, 65),
          (4, 16384, 33), (16, 65536, 257), (3, 100000, 700), (128, 2048, 200), (48, 32768, 512), (16, 8192, 1000)]
# (tile, FFTCONV_PERS): 0 = general kernel, n = batch-sharing kernel with n slots
CANDS = [(256, 0), (512, 0), (1024, 0), (2048, 0), (1024, 2), (1024, 4), (2048, 1), (2048, 2)]


def timed(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def run(x, K, tile, pers):
    for k in ("FFTCONV_TILE", "FFTCONV_PERS"):
        os.environ.pop(k, None)
    if tile:
        os.environ["FFTCONV_TILE"] = str(tile)
        os.environ["FFTCONV_PERS"] = str(pers)
    _native.clear_plan_cache()
    try:
        layer = FFTConv1d(8, 8, K).to(dev).eval()
        with torch.no_grad():
            layer(x)
            plan = layer.__dict__["_spectrum_cache"][1].plan
            if tile and (plan.tile != tile or plan.layout[7] != pers):
                return None, None            # the forced flavour is not available for this shape
            g = torch.cuda.CUDAGraph()       # graph replay: the host stays out of the timing of 10-us launches
            y = layer(x)
            with torch.cuda.graph(g):
                for _ in range(4):
                    y = layer(x)
            return timed(g.replay) / 4, (plan.tile, plan.layout[7], plan.layout[1])
    except Exception:  # noqa: BLE001
        return None, None


for B, L, K in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B, 8, L, device=dev)
    run(x, K, 0, 0)                    # (spin-up: the first launches of a shape see the clock ramp)
    cands = {}
    for tile, pers in CANDS:
        if tile < K:
            continue
        us, _ = run(x, K, tile, pers)
        if us is not None:
            cands[f"{tile}/{pers}"] = round(us, 2)
    auto_us, auto_pick = run(x, K, 0, 0)
    best = min(cands.values()) if cands else None
    print(json.dumps({"shape": {"batch": B, "length": L, "taps": K}, "planner_pick": auto_pick, "planner_us": round(auto_us, 2),
                      "forced_us": cands, "best_forced_us": best,
                      "regret": None if best is None else round(auto_us / best - 1.0, 4)}), flush=True)
for k in ("FFTCONV_TILE", "FFTCONV_PERS"):
    os.environ.pop(k, None)
