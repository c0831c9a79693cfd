#!/usr/bin/env python3
"""Forward + backward (dX, dW, db) of the BASELINE shapes: this library's modules against torch's direct convolution autograd
(MIOpen) on the same GPU.  Eager launches, HIP events around `iters` steps, us per step; and the same step captured once into a
HIP graph and replayed (the host's ~50 launches and allocator calls per step leave the timed path)."""
import json
import sys
import os

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_conv_pytorch_amd as fca  # noqa: E402

dev = "cuda:0"
CASES = [("cfgA 1-D B32 8->8 L32768 k512", 1, 32, 8, 32768, 512, 100), ("cfgB 2-D B16 8->8 512^2 k31", 2, 16, 8, 512, 31, 30), ("cfgC 3-D B8 8->8 64^3 k9", 3, 8, 8, 64, 9, 30),
         ("2-D B16 8->8 512^2 k7", 2, 16, 8, 512, 7, 30), ("2-D B4 8->8 256^2 k15", 2, 4, 8, 256, 15, 100)]


def timed(fn, iters):
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


for name, nd, b, c, s, k, iters in CASES:
    cls = {1: fca.FFTConv1d, 2: fca.FFTConv2d, 3: fca.FFTConv3d}[nd]
    layer = cls(c, c, k, bias=True).to(dev)
    x = torch.randn(b, c, *([s] * nd), device=dev, requires_grad=True)
    wr = layer.weight.detach().clone().requires_grad_()
    br = layer.bias.detach().clone().requires_grad_()
    conv = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}[nd]

    def ours():
        layer.zero_grad(set_to_none=True)
        x.grad = None
        layer(x).sum().backward()

    def ref():
        wr.grad = br.grad = x.grad = None
        conv(x, wr, br).sum().backward()

    with torch.no_grad():
        fwd = timed(lambda: layer(x), iters)
    t_ours = timed(ours, iters)
    t_ref = timed(ref, max(3, iters // 10))

    def graph_step():                       # fixed gradient buffers: the graph writes where the warm-up steps allocated
        layer.zero_grad(set_to_none=False)
        x.grad.zero_()
        layer(x).sum().backward()

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            graph_step()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        graph_step()
    t_graph = timed(g.replay, iters)
    print(json.dumps({"shape": name, "forward_us": round(fwd, 1), "forward_backward_us": round(t_ours, 1),
                      "forward_backward_graph_replay_us": round(t_graph, 1), "torch_conv_autograd_us": round(t_ref, 1)}), flush=True)
