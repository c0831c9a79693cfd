"""3-D plane-major pipeline (planes3d.hpp) against the separable passes and torch's direct convolution.

Usage (GPU box): python scripts/planes_check.py [--time]
Every case runs with FFTCONV_PLANES=1 and =0 (the knob is read at plan creation: the plan cache is cleared in
between) and is compared with torch.nn.functional.conv3d / conv_transpose3d in float64 on the CPU for small cases,
float32 on the GPU for the large ones."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_conv_pytorch_amd import _native  # noqa: E402
from fft_conv_pytorch_amd.functional import fft_conv, fft_conv_transpose  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


CASES = [
    # B, Cin, Cout, groups, size, k, stride, padding, dilation, mode
    (8, 8, 8, 1, (64, 64, 64), (9, 9, 9), 1, 0, 1, "constant"),          # cfgC
    (2, 8, 8, 1, (64, 64, 64), (2, 4, 8), 1, 0, 1, "constant"),          # README shapes
    (1, 8, 8, 1, (64, 64, 64), (9, 9, 9), 1, 0, 1, "constant"),          # cfgC shard (B = 1)
    (3, 5, 7, 1, (40, 50, 60), (3, 4, 5), 1, 1, 1, "constant"),          # odd batch, ragged channels
    (2, 8, 16, 1, (30, 33, 47), (5, 3, 2), (2, 1, 3), (2, 1, 0), 1, "constant"),   # strides, two out-chunks
    (2, 16, 8, 2, (20, 60, 62), (3, 3, 3), 1, 1, 1, "reflect"),          # groups, reflect
    (2, 8, 8, 1, (21, 40, 40), (3, 5, 5), 1, 2, (1, 2, 3), "circular"),  # dilation, circular
    (2, 6, 6, 1, (17, 33, 20), (2, 3, 3), (1, 2, 1), 1, 1, "replicate"),
    (2, 8, 8, 1, (150, 40, 40), (9, 3, 3), 1, 0, 1, "constant"),         # several z tiles
    (1, 8, 8, 1, (100, 20, 20), (33, 3, 3), (3, 1, 1), 4, 1, "constant"),  # z tiles, stride along z, longest z kernel
]
TCASES = [
    # B, Cin, Cout, groups, size, k, stride, padding, output_padding, dilation
    (2, 8, 8, 1, (20, 20, 20), (3, 3, 3), 2, 1, 1, 1),
    (2, 8, 6, 1, (30, 10, 15), (4, 3, 2), (1, 2, 3), (1, 0, 1), (0, 1, 2), 1),
]


def run(fn, env):
    os.environ["FFTCONV_PLANES"] = env
    _native.clear_plan_cache()
    out = fn()
    torch.cuda.synchronize()
    return out


def timeit(fn, env, iters=30):
    os.environ["FFTCONV_PLANES"] = env
    _native.clear_plan_cache()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    dev = "cuda:0"
    torch.manual_seed(0)
    worst = 0.0
    for (B, Ci, Co, g, size, k, s, p, d, mode) in CASES:
        x = torch.randn(B, Ci, *size, device=dev)
        w = torch.randn(Co, Ci // g, *k, device=dev)
        b = torch.randn(Co, device=dev)
        fn = lambda: fft_conv(x, w, b, stride=s, padding=p, dilation=d, groups=g, padding_mode=mode)  # noqa: E731
        y1 = run(fn, "1")
        y0 = run(fn, "0")
        big = x.numel() > 4_000_000
        xx, ww, bb = (x, w, b) if big else (x.double().cpu(), w.double().cpu(), b.double().cpu())
        if mode == "constant":
            ref = F.conv3d(xx, ww, bb, stride=s, padding=p, dilation=d, groups=g)
        else:
            pp = (p,) * 3 if isinstance(p, int) else p
            padl = [q for ax in reversed(pp) for q in (ax, ax)]
            ref = F.conv3d(F.pad(xx, padl, mode=mode), ww, bb, stride=s, dilation=d, groups=g)
        ref = ref.to(dev)
        e1, e0, e10 = rel(y1, ref), rel(y0, ref), rel(y1, y0)
        worst = max(worst, e1)
        print(f"conv  B{B} {Ci}->{Co} g{g} {size} k{k} s{s} p{p} d{d} {mode}: planes {e1:.2e} separable {e0:.2e} "
              f"planes-vs-separable {e10:.2e} shape {tuple(y1.shape)}", flush=True)
        assert y1.shape == ref.shape and e1 < 1e-4 and e0 < 1e-4, "parity"
    for (B, Ci, Co, g, size, k, s, p, op, d) in TCASES:
        x = torch.randn(B, Ci, *size, device=dev)
        w = torch.randn(Ci, Co // g, *k, device=dev)
        b = torch.randn(Co, device=dev)
        fn = lambda: fft_conv_transpose(x, w, b, stride=s, padding=p, output_padding=op, dilation=d, groups=g)  # noqa: E731
        y1 = run(fn, "1")
        y0 = run(fn, "0")
        ref = F.conv_transpose3d(x.double().cpu(), w.double().cpu(), b.double().cpu(), stride=s, padding=p, output_padding=op,
                                 dilation=d, groups=g).to(dev)
        e1, e0 = rel(y1, ref), rel(y0, ref)
        worst = max(worst, e1)
        print(f"convT B{B} {Ci}->{Co} {size} k{k} s{s} p{p} op{op}: planes {e1:.2e} separable {e0:.2e} shape {tuple(y1.shape)}", flush=True)
        assert y1.shape == ref.shape and e1 < 1e-4 and e0 < 1e-4, "parity (transposed)"
    print(f"all cases within 1e-4 (worst {worst:.2e})")
    if "--time" in sys.argv:
        from fft_conv_pytorch_amd import FFTConv3d
        for B in (8, 1):
            layer = FFTConv3d(8, 8, 9, bias=True).to(dev).eval()
            x = torch.randn(B, 8, 64, 64, 64, device=dev)
            with torch.no_grad():
                t1 = timeit(lambda: layer(x), "1")
                t0 = timeit(lambda: layer(x), "0")
            print(f"cfgC B={B}: eager module call planes {t1:.1f} us, separable {t0:.1f} us")


if __name__ == "__main__":
    main()
