#!/usr/bin/env python3
"""Forward + backward (dX, dW, db) of the 2-D / 3-D BASELINE shapes: this library's modules against torch's direct
convolution autograd (MIOpen) on the same GPU.  Eager launches, HIP events around `iters` steps, us per step."""
import json
import sys
import os

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_conv_pytorch_amd as fca  # noqa: E402

dev = "cuda:0"
CASES = [("cfgB 2-D B16 8->8 512^2 k31", 2, 16, 8, 512, 31, 30), ("cfgC 3-D B8 8->8 64^3 k9", 3, 8, 8, 64, 9, 30),
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
    cls = fca.FFTConv2d if nd == 2 else fca.FFTConv3d
    layer = cls(c, c, k, bias=True).to(dev)
    x = torch.randn(b, c, *([s] * nd), device=dev, requires_grad=True)
    wr = layer.weight.detach().clone().requires_grad_()
    br = layer.bias.detach().clone().requires_grad_()
    conv = F.conv2d if nd == 2 else F.conv3d

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
    print(json.dumps({"shape": name, "forward_us": round(fwd, 1), "forward_backward_us": round(t_ours, 1),
                      "torch_conv_autograd_us": round(t_ref, 1)}), flush=True)
