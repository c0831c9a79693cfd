#!/usr/bin/env python3
"""Per-iteration wall clock of one README-shape point whose mean was 25x its best (2-D transposed, k = 10)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fft_conv_pytorch_amd.functional import fft_conv_transpose, fft_conv
dev = torch.device("cuda", 0)
x = torch.randn(2, 8, 512, 512, device=dev, requires_grad=True)
for k in (4, 10, 16):
    w = torch.randn(8, 8, k, k, device=dev, requires_grad=True)
    b = torch.randn(8, device=dev, requires_grad=True)
    ts = []
    s0 = torch.cuda.memory_stats()
    for i in range(16):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = fft_conv_transpose(x, w, bias=b)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    s1 = torch.cuda.memory_stats()
    print(k, [round(t) for t in ts], "device allocs", s1["num_device_alloc"] - s0["num_device_alloc"], "frees", s1["num_device_free"] - s0["num_device_free"],
          "reserved MB", s1["reserved_bytes.all.current"] >> 20)
