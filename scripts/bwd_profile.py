"""Where does a training step (forward + backward) of FFTConv1d at the cfgA shape spend its time?  torch.profiler table
(host and device time per op) + wall time per step.  Usage: python scripts/bwd_profile.py [--nd 2]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_conv_pytorch_amd as fca  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nd", type=int, default=1)
args = ap.parse_args()
dev = "cuda:0"
if args.nd == 1:
    layer = fca.FFTConv1d(8, 8, 512, bias=True).to(dev)
    x = torch.randn(32, 8, 32768, device=dev, requires_grad=True)
else:
    layer = fca.FFTConv2d(8, 8, 15, bias=True).to(dev)
    x = torch.randn(4, 8, 256, 256, device=dev, requires_grad=True)


def step():
    y = layer(x)
    y.backward(gy)


y = layer(x)
gy = torch.randn_like(y)
for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print(f"fwd+bwd wall: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per step")
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(20):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
