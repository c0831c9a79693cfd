"""Graph-replayed time per forward of the cfgA shape with whatever library FFTCONV_LIB names (diagnostic builds:
timing only).  Usage: FFTCONV_LIB=.../libfftconv_diag1.so python scripts/experiments/time_cfgA.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fft_conv_pytorch_amd import FFTConv1d  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
layer = FFTConv1d(8, 8, 512, bias=True).to(dev).eval()
xs = [torch.randn(32, 8, 32768, device=dev) for _ in range(9)]
with torch.no_grad():
    for x in xs:
        layer(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for x in xs:
            layer(x)
    for _ in range(300):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
print(f"{os.environ.get('FFTCONV_LIB', 'product')}: {e0.elapsed_time(e1) * 1e3 / 900:.2f} us per forward")
