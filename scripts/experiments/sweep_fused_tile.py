"""Outermost-axis (fused column pass) tile of N-d forward plans: every tile forced in turn (tile_hint) against the planner's pick,
graph-replayed, us per call.  The rows / middle axis keep the planner's choice."""
import json
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


# nd, B, C, size, k, pad
CASES = [(2, 16, 8, (512, 512), 7, 3), (2, 16, 8, (512, 512), 31, 15), (2, 8, 8, (1024, 1024), 5, 2), (2, 8, 8, (300, 300), 5, 0),
         (2, 4, 8, (2000, 100), 9, 0), (3, 2, 8, (128, 128, 128), 5, 2), (3, 4, 8, (300, 40, 90), 3, 1), (2, 16, 8, (512, 512), 15, 0)]
for nd, b, c, size, k, pad in CASES:
    x = torch.randn(b, c, *size, device=dev)
    w = torch.randn(c, c, *([k] * nd), device=dev)
    one = (1,) * nd
    row = {}
    for hint in (0, 64, 128, 256, 512, 1024):
        _native.clear_plan_cache()
        try:
            plan = F_._plan_for(x, w, None, one, (pad,) * nd, one, 1, "constant", tile_hint=hint)
            spec = F_.transform_kernel(plan, w)
            row["auto->%d" % plan.tile if hint == 0 else str(hint)] = round(timed(lambda: F_._forward_native(x, spec, None)), 1)
        except (NotImplementedError, ValueError):
            pass
    print(json.dumps({"shape": f"{nd}-D B{b} {'x'.join(map(str, size))} k{k} pad{pad}", **row}), flush=True)
