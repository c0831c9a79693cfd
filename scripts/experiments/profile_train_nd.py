"""torch.profiler breakdown of one forward + backward step of the 2-D / 3-D BASELINE shapes (device time per kernel, us)."""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import fft_conv_pytorch_amd as fca

dev = "cuda:0"
for name, nd, b, c, s, k in (("cfgB", 2, 16, 8, 512, 31), ("cfgC", 3, 8, 8, 64, 9), ("B4 256^2 k15", 2, 4, 8, 256, 15)):
    cls = fca.FFTConv2d if nd == 2 else fca.FFTConv3d
    layer = cls(c, c, k, bias=True).to(dev)
    x = torch.randn(b, c, *([s] * nd), device=dev, requires_grad=True)

    def step():
        layer.zero_grad(set_to_none=True)
        x.grad = None
        layer(x).sum().backward()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    rows = [(e.key, e.device_time_total / 10, e.count / 10) for e in prof.key_averages() if e.device_time_total > 0 and not e.key.startswith("aten::")
            and not e.key.startswith("autograd::") and "Backward" not in e.key]
    rows.sort(key=lambda r: -r[1])
    print(name, "device total", round(sum(r[1] for r in rows), 1))
    for key, us, n in rows[:14]:
        print(f"   {us:9.1f} us  x{n:4.1f}  {key[:100]}")
