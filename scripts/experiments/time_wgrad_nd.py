"""A/B on one box: N-d weight gradient through fc_wgrad_nd against the round-2 route (forward plans fed by torch
transposes), wall time per call (eager launches) and device time per call (torch.profiler)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from fft_conv_pytorch_amd import autograd as A

dev = "cuda:0"
CASES = [
    ("2-D B4 8->8 256^2 k15", 4, 8, 8, 1, (256, 256), (15, 15), (1, 1), (0, 0), (1, 1)),
    ("2-D B16 8->8 512^2 k31 (cfgB)", 16, 8, 8, 1, (512, 512), (31, 31), (1, 1), (0, 0), (1, 1)),
    ("2-D B8 16->16 128^2 k5 s2", 8, 16, 16, 1, (128, 128), (5, 5), (2, 2), (2, 2), (1, 1)),
    ("3-D B8 8->8 64^3 k9 (cfgC)", 8, 8, 8, 1, (64, 64, 64), (9, 9, 9), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
]
for name, b, ci, co, g, size, k, st, pad, dil in CASES:
    x = torch.randn(b, ci, *size, device=dev)
    lout = tuple((s + 2 * p - d * (kk - 1) - 1) // s_ + 1 for s, p, d, kk, s_ in zip(size, pad, dil, k, st))
    gy = torch.randn(b, co, *lout, device=dev)
    wshape = (co, ci // g) + tuple(k)
    routes = {"fc_wgrad_nd": lambda: A._grad_weight_nd_native(x, gy, wshape, st, pad, dil, g, "constant"),
              "forward plans + torch transposes": lambda: A._grad_weight_plans(x, gy, wshape, st, pad, dil, g, "constant")}
    for rname, fn in routes.items():
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e6
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        devt = sum(e.device_time_total for e in prof.key_averages()) / 20
        kern = {e.key[:40]: round(e.device_time_total / 20, 1) for e in prof.key_averages() if e.device_time_total > 0}
        print(json.dumps({"case": name, "route": rname, "wall_us": round(wall, 1), "device_us": round(devt, 1), "kernels_us": kern}), flush=True)
