"""Backward-only probe at the cfgA shape (for rocprofv3 --kernel-trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_conv_pytorch_amd as fca
dev = "cuda:0"
B, C, L, K = 32, 8, 32768, 512
conv = fca.FFTConv1d(C, C, K, bias=True).to(dev)
x = torch.randn(B, C, L, device=dev, requires_grad=True)
for i in range(6):
    x.grad = None
    conv.zero_grad(set_to_none=True)
    conv(x).sum().backward()
torch.cuda.synchronize()
