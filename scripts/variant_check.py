#!/usr/bin/env python3
"""One tuning variant of the batch-sharing 1-D kernel (selected through FFTCONV_* environment knobs):
parity against torch's direct convolution on the same device, then HIP-event time per launch over
rotating cold buffers (same timing as bench.py).  Prints one JSON line."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_conv_pytorch_amd as fca  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--ch", type=int, default=8)
ap.add_argument("--len", type=int, default=32768)
ap.add_argument("--k", type=int, default=512)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--tag", default="")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
layer = fca.FFTConv1d(args.ch, args.ch, args.k, bias=True).to(dev).eval()
nbuf = 9
xs = [torch.randn(args.batch, args.ch, args.len, device=dev) for _ in range(nbuf)]
with torch.no_grad():
    y = layer(xs[0])
    ref = torch.nn.functional.conv1d(xs[0][:4], layer.weight, layer.bias)
err = float((y[:4] - ref).abs().max() / ref.abs().max())
spectrum = layer.__dict__["_spectrum_cache"][1]
plan = spectrum.plan
ys = [torch.empty_like(y) for _ in range(nbuf)]
bias_ptr = layer.bias.data_ptr()


def step(i):
    j = i % nbuf
    plan.forward(xs[j].data_ptr(), spectrum.buf.data_ptr(), bias_ptr, ys[j].data_ptr(), None,
                 torch.cuda.current_stream(dev).cuda_stream)


side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for i in range(nbuf):
        step(i)
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for i in range(nbuf):
        step(i)
for _ in range(5):
    graph.replay()
torch.cuda.synchronize()
best = 1e9
reps = max(1, args.steps // nbuf)
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * nbuf))
same = bool(torch.equal(ys[0], y))
knobs = {k: v for k, v in os.environ.items() if k.startswith("FFTCONV_")}
print(json.dumps({"tag": args.tag, "knobs": knobs, "tile": plan.tile, "rel_err": err, "replay_equal": same, "us": best}))
