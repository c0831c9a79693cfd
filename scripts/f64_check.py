"""float64 1-D convolution: FFT path (csrc/fft_f64.hip) against the direct kernel and torch.fft on the same GPU.
Usage: python scripts/f64_check.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_conv_pytorch_amd import _native  # noqa: E402
from fft_conv_pytorch_amd.functional import fft_conv  # noqa: E402

dev = "cuda:0"


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


for (B, C, L, K) in ((1, 8, 32768, 128), (1, 8, 32768, 512), (8, 8, 32768, 512), (32, 8, 32768, 512), (4, 16, 65536, 1025)):
    x = torch.randn(B, C, L, device=dev, dtype=torch.float64)
    w = torch.randn(C, C, K, device=dev, dtype=torch.float64)
    b = torch.randn(C, device=dev, dtype=torch.float64)
    res = {}
    from fft_conv_pytorch_amd import FFTConv1d
    layer = FFTConv1d(C, C, K, bias=True).to(dev).double().eval()     # module: kernel spectrum cached, one launch per call
    with torch.no_grad():
        layer.weight.copy_(w)
        layer.bias.copy_(b)
    for knob in ("1", "0"):
        os.environ["FFTCONV_F64_FFT"] = knob
        _native.clear_plan_cache()
        layer.invalidate_kernel_spectrum()
        layer.__dict__.pop("_last_plan", None)
        with torch.no_grad():
            y = layer(x)
            wall = timed(lambda: layer(x))
            g = torch.cuda.CUDAGraph()           # graph replay: the launches without the host (eager calls of a 20-us kernel are host-bound)
            with torch.cuda.graph(g):
                for _ in range(4):
                    layer(x)
            res[knob] = (wall, y, timed(g.replay) / 4)
    n = L
    yref = torch.fft.irfft(torch.einsum("bif,oif->bof", torch.fft.rfft(x, n), torch.fft.rfft(w, n).conj()), n)[..., : L - K + 1] + b[None, :, None]
    t_fft = timed(lambda: torch.fft.irfft(torch.einsum("bif,oif->bof", torch.fft.rfft(x, n), torch.fft.rfft(w, n).conj()), n))
    err = float((res["1"][1] - yref).abs().max() / yref.abs().max())
    print(f"B{B} {C}->{C} L{L} k{K}: FFT path {res['1'][0]:8.1f} us eager / {res['1'][2]:8.1f} us replayed, direct kernel "
          f"{res['0'][0]:9.1f} / {res['0'][2]:9.1f} us ({res['0'][2] / res['1'][2]:.1f}x replayed), torch.fft formulation {t_fft:8.1f} us; "
          f"rel err vs torch.fft {err:.2e}", flush=True)
os.environ.pop("FFTCONV_F64_FFT", None)
