#!/usr/bin/env python3
"""Per-phase timeline of the fused column pass of the 2-D / 3-D path (timestamp hook, diagnostic only)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_conv_pytorch_amd as fca  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nd", type=int, default=2)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--ch", type=int, default=8)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--k", type=int, default=31)
args = ap.parse_args()
dev = "cuda:0"
cls = fca.FFTConv2d if args.nd == 2 else fca.FFTConv3d
layer = cls(args.ch, args.ch, args.k).to(dev).eval()
x = torch.randn(args.batch, args.ch, *([args.size] * args.nd), device=dev)
for _ in range(3):
    y = layer(x)
plan = layer.__dict__["_spectrum_cache"][1].plan
grid = plan.debug_grid()
buf = torch.zeros(grid * 8, dtype=torch.int64, device=dev)
plan.debug_set_stamps(buf.data_ptr())
torch.cuda.synchronize()
y = layer(x)
torch.cuda.synchronize()
plan.debug_set_stamps(None)
st = buf.cpu().numpy().reshape(grid, 8).astype(np.float64) * 0.01   # 100 MHz ticks -> microseconds
st = st[st[:, 7] > 0]
t0 = st[:, 0].min()
names = ["start", "input landed", "fwd FFT done", "barrier", "mix done", "barrier", "inverse FFT done", "stores landed"]
if args.nd == 3 and os.environ.get("FFTCONV_PLANES", "1") != "0" and args.size <= 64:   # colz (planes3d.hpp)
    names = ["start", "input landed", "fwd FFT done", "mix done", "inverse FFT done", "stores issued", "stores landed", "-"]
print(f"workgroups={len(st)} tile={plan.tile}  kernel span = {st[:, 7].max() - t0:.2f} us")
for i in range(1, 8):
    dt = st[:, i] - st[:, i - 1]
    print(f"{names[i]:18s} median {np.median(dt):7.2f}  p10 {np.percentile(dt, 10):7.2f}  p90 {np.percentile(dt, 90):7.2f}")
life = st[:, 7] - st[:, 0]
print(f"workgroup lifetime p10/p50/p90 = {np.percentile(life, 10):.2f}/{np.median(life):.2f}/{np.percentile(life, 90):.2f} us")
hist, edges = np.histogram(st[:, 0] - t0, bins=12)
print("start-time histogram:", list(zip(np.round(edges[:-1], 1), hist)))
