"""Parity + timing of one batch-sharing kernel flavour (FFTCONV_PERS / FFTCONV_TILE from the environment)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fft_conv_pytorch_amd.functional import fft_conv

torch.manual_seed(0)
dev = "cuda"
cases = [  # B, Cin, Cout, L, K, padding, mode, groups, bias
    (32, 8, 8, 32768, 512, 0, "constant", 1, True),
    (3, 8, 8, 5000, 129, 64, "constant", 1, True),
    (5, 8, 16, 4097, 33, 16, "reflect", 1, False),
    (2, 16, 16, 3000, 200, 10, "circular", 2, True),
    (1, 6, 8, 2500, 100, 7, "replicate", 1, True),
    (7, 8, 24, 1500, 500, 0, "constant", 1, False),
]
worst = 0.0
for (B, ci, co, L, K, pad, mode, g, hb) in cases:
    x = torch.randn(B, ci, L, device=dev)
    w = torch.randn(co, ci // g, K, device=dev) / (K * ci) ** 0.5
    b = torch.randn(co, device=dev) if hb else None
    y = fft_conv(x, w, b, padding=pad, padding_mode=mode, groups=g)
    xp = F.pad(x, [pad, pad], mode=mode) if mode != "constant" else x
    ref = F.conv1d(xp.double(), w.double(), b.double() if hb else None, padding=pad if mode == "constant" else 0, groups=g)
    err = ((y.double() - ref).norm() / ref.norm()).item()
    worst = max(worst, err)
    print(f"case B{B} C{ci}->{co} L{L} K{K} pad{pad} {mode} g{g}: rel {err:.2e}", flush=True)
assert worst < 1e-4, worst
B, ci, co, L, K = 32, 8, 8, 32768, 512
xs = [torch.randn(B, ci, L, device=dev) for _ in range(9)]
w = torch.randn(co, ci, K, device=dev)
from fft_conv_pytorch_amd import FFTConv1d
m = FFTConv1d(ci, co, K, bias=True).to(dev)
with torch.no_grad():
    for i in range(30): m(xs[i % 9])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(300): m(xs[i % 9])
    e1.record(); torch.cuda.synchronize()
print(f"cfgA eager {e0.elapsed_time(e1) / 300 * 1000:.1f} us/launch  PERS={os.environ.get('FFTCONV_PERS')} TILE={os.environ.get('FFTCONV_TILE')}")
