"""Per-kernel device time of the cfgB / cfgC input-gradient plan under a few forced tile choices (torch.profiler)."""
import os
import sys

import torch

sys.path.insert(0, ".")
from fft_conv_pytorch_amd import functional as F_, _native

dev = "cuda:0"
for nd, b, c, s, k, combos in ((2, 16, 8, 512, 31, [(0, 0, 0), (256, 0, 256), (512, 0, 512)]), (3, 8, 8, 64, 9, [(0, 0, 0)])):
    lo = s - k + 1
    gy = torch.randn(b, c, *([lo] * nd), device=dev)
    w = torch.randn(c, c, *([k] * nd), device=dev)
    one = (1,) * nd
    for xt, yt, hint in combos:
        os.environ["FFTCONV_XTILE"] = str(xt)
        os.environ["FFTCONV_YTILE"] = str(yt)
        _native.clear_plan_cache()
        plan = F_._plan_for(gy, w, None, one, (0,) * nd, one, 1, "constant", tile_hint=hint, transposed=True, output_padding=(0,) * nd)
        spec = F_.transform_kernel(plan, w)
        for _ in range(3):
            F_._forward_native(gy, spec, None)
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            for _ in range(10):
                F_._forward_native(gy, spec, None)
            torch.cuda.synchronize()
        rows = sorted(((e.key, e.device_time_total / 10, e.count / 10) for e in prof.key_averages() if e.device_time_total > 0), key=lambda r: -r[1])
        print(f"nd={nd} xtile={xt} ytile={yt} hint={hint} tile={plan.tile} layout={plan.layout[:4]} total={sum(r[1] for r in rows):.1f}")
        for key, us, n in rows[:8]:
            print(f"   {us:9.1f} us x{n:4.1f} {key[:90]}")
