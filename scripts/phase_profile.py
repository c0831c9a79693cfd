#!/usr/bin/env python3
"""Per-phase timeline of the fused 1-D kernel from its timestamp hook (fc_debug_set_stamps).

Diagnostic only: the stamped launch drains loads at two extra points, so read SHARES, not totals."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_conv_pytorch_amd as fca  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--ch", type=int, default=8)
ap.add_argument("--len", type=int, default=32768)
ap.add_argument("--k", type=int, default=512)
ap.add_argument("--tile", type=int, default=0)
ap.add_argument("--dil", type=int, default=1)
ap.add_argument("--groups", type=int, default=1)
args = ap.parse_args()
if args.tile:
    os.environ["FFTCONV_TILE"] = str(args.tile)
dev = "cuda:0"
layer = fca.FFTConv1d(args.ch, args.ch, args.k, dilation=args.dil, groups=args.groups).to(dev).eval()
x = torch.randn(args.batch, args.ch, args.len, device=dev)
for _ in range(3):
    y = layer(x)
plan = layer.__dict__["_spectrum_cache"][1].plan
nrec = plan.debug_grid()                 # records of 16 stamps: one per wave (16 slots) of every work item
buf = torch.zeros(nrec * 16, dtype=torch.int64, device=dev)
plan.debug_set_stamps(buf.data_ptr())
torch.cuda.synchronize()
y = layer(x)
torch.cuda.synchronize()
plan.debug_set_stamps(None)
ticks = buf.cpu().numpy().reshape(-1, 16, 16).astype(np.float64)
raw = ticks * 0.01   # [item][wave][stamp], 100 MHz ticks -> us
if (ticks[:, 0, 13] > 0).any():     # slots 12 / 13: shader clock counter at the item's start / end
    ghz = (ticks[:, 0, 13] - ticks[:, 0, 12]) / np.maximum(ticks[:, 0, 11] - ticks[:, 0, 0], 1) * 0.1
    print(f"shader clock held over an item (s_memtime / s_memrealtime): median {np.median(ghz):.3f} GHz, "
          f"p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}")
grid = raw.shape[0]
nw = int((raw[:, :, 0] > 0).sum(axis=1).max())
st = raw[:, 0, :]
t0 = st[:, 0].min()
names = ["start", "input landed", "passA done", "barrier1", "passB done", "barrier3", "mix done", "barrier4",
         "invA done", "barrier6", "stores issued", "stores landed"]
print(f"items={grid} waves/wg={nw} tile={plan.tile}  kernel span = {raw[:, :nw, 11].max() - t0:.2f} us")
print(f"{'phase':16s} {'median dt':>10s} {'p10':>8s} {'p90':>8s}   (us, per work item, lane 0 of wave 0)")
for i in range(1, 12):
    dt = st[:, i] - st[:, i - 1]
    print(f"{names[i]:16s} {np.median(dt):10.2f} {np.percentile(dt, 10):8.2f} {np.percentile(dt, 90):8.2f}")
# per-wave view: when does each wave reach each stamp, relative to the item's earliest start (median over items)
start = raw[:, :nw, 0].min(axis=1, keepdims=True)
print("per-wave arrival (us after the item's first wave started; median over items)")
print("stamp            " + " ".join(f"w{w:<5d}" for w in range(nw)))
for i in range(12):
    rel = raw[:, :nw, i] - start
    print(f"{names[i]:16s} " + " ".join(f"{np.median(rel[:, w]):6.2f}" for w in range(nw)))
life = st[:, 11] - st[:, 0]
print(f"work item lifetime median {np.median(life):.2f} us; start times: p50 {np.median(st[:, 0] - t0):.2f} "
      f"p90 {np.percentile(st[:, 0] - t0, 90):.2f} max {(st[:, 0] - t0).max():.2f} us")
half = st[:, 0] - t0 > 0.5 * np.median(life)
for name, sel in (("first round", ~half), ("later rounds", half), ("first round, slowest 10 %", ~half & (life > np.percentile(life[~half], 90)))):
    if sel.any():
        print(f"-- {name}: median dt per phase: " + "  ".join(f"{names[i]} {np.median(st[sel, i] - st[sel, i - 1]):.2f}" for i in range(1, 12)))
        print(f"{name}: n={int(sel.sum())} lifetime p10/p50/p90/max = "
              f"{np.percentile(life[sel], 10):.1f}/{np.median(life[sel]):.1f}/{np.percentile(life[sel], 90):.1f}/{life[sel].max():.1f} us, "
              f"end p50/max = {np.median(st[sel, 11] - t0):.1f}/{(st[sel, 11] - t0).max():.1f} us")
