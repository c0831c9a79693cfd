"""2-D / 3-D forwards with many channels, groups and depthwise shapes against torch's direct convolution on the same GPU
(graph-replayed module forward, us per call) -- a look for pathologies outside the 8-channel BASELINE shapes."""
import json
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
import fft_conv_pytorch_amd as fca

dev = "cuda:0"


def timed(fn, iters=20):
    with torch.no_grad():
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


# nd, B, Cin, Cout, groups, size, k
CASES = [(2, 8, 64, 64, 1, 128, 15), (2, 8, 64, 64, 64, 128, 15), (2, 8, 64, 64, 8, 128, 15), (2, 4, 32, 32, 1, 256, 31),
         (2, 4, 3, 16, 1, 256, 31), (2, 4, 16, 3, 1, 256, 31), (3, 2, 32, 32, 1, 64, 9), (3, 2, 32, 32, 32, 64, 9), (2, 16, 128, 128, 128, 64, 31)]
for nd, b, ci, co, g, s, k in CASES:
    cls = fca.FFTConv2d if nd == 2 else fca.FFTConv3d
    layer = cls(ci, co, k, groups=g).to(dev).eval()
    x = torch.randn(b, ci, *([s] * nd), device=dev)
    conv = F.conv2d if nd == 2 else F.conv3d
    w, bias = layer.weight.detach(), layer.bias.detach()
    ours = timed(lambda: layer(x))
    ref = timed(lambda: conv(x, w, bias, groups=g), iters=5)
    plan = layer.__dict__["_spectrum_cache"][1].plan
    out_elems = b * co * (s - k + 1) ** nd
    alg = (x.numel() + out_elems) * 4
    print(json.dumps({"shape": f"{nd}-D B{b} {ci}->{co} g{g} {s}^{nd} k{k}", "ours_us": round(ours, 1), "torch_direct_us": round(ref, 1),
                      "frac_of_8TBps": round(alg / ours / 1e6 / 8000, 4), "layout": plan.layout}), flush=True)
