"""Forward + backward of 'same'-padded 3-D layers (planes past 64 x 64 after padding: tiled plane-major pipeline for y and dX,
fc_wgrad_nd for dW) against torch's direct convolution autograd on the same GPU, eager, us per step."""
import json
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
import fft_conv_pytorch_amd as fca

dev = "cuda:0"


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for b, c, s, k in ((8, 8, 64, 3), (2, 8, 128, 5), (8, 8, 64, 9)):
    layer = fca.FFTConv3d(c, c, k, padding=k // 2).to(dev)
    x = torch.randn(b, c, s, s, s, device=dev, requires_grad=True)
    gy = torch.randn(b, c, s, s, s, device=dev)
    wr, br = layer.weight.detach().clone().requires_grad_(), layer.bias.detach().clone().requires_grad_()

    def ours():
        layer.zero_grad(set_to_none=True)
        x.grad = None
        layer(x).backward(gy)

    def ref():
        wr.grad = br.grad = x.grad = None
        F.conv3d(x, wr, br, padding=k // 2).backward(gy)

    with torch.no_grad():
        fwd = timed(lambda: layer(x), 20)
    print(json.dumps({"shape": f"B{b} {c}->{c} {s}^3 k{k} 'same'", "forward_us": round(fwd, 1), "forward_backward_us": round(timed(ours, 20), 1),
                      "torch_conv_autograd_us": round(timed(ref, 3), 1)}), flush=True)
