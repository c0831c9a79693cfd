#!/usr/bin/env python3
"""The reference's README benchmark (doc/scripts/generate_benchmark_plot.py:128-159) on one MI355X: batch 2,
8 -> 8 channels, groups 1, bias, fp32; 1-D L = 32768, 2-D 512 x 512, 3-D 64^3, swept over its kernel sizes,
forward and transposed.  Method as in the reference's harness (benchmark_utils.py:23-50): eager calls of the
FUNCTIONAL (kernel transformed on every call, output allocated per call, inputs with requires_grad so the
autograd node is recorded), wall clock bracketed by synchronize, 16 iterations, the first dropped.
Beside it: the reference's algorithm written with torch.fft (rocFFT underneath -- what the reference itself
would run on this GPU) and torch's direct convolution (MIOpen).  One JSON line per point.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_conv_pytorch_amd.functional import fft_conv, fft_conv_transpose  # noqa: E402

HBM_GBPS = 8000.0
ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/readme_shapes.jsonl")
ap.add_argument("--iters", type=int, default=16)
ap.add_argument("--quick", action="store_true", help="three kernel sizes per dimension")
args = ap.parse_args()
dev = torch.device("cuda", 0)


def measure(fn, iters):
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[1:])
    # (mean without the slowest sample: one call in a series now and then pays a hipMalloc of the caching allocator -- tens of
    #  milliseconds for the 67 MB spectrum of a 2-D shape -- which is the harness's allocation pattern, not the call's cost)
    kept = ts[:-1] if len(ts) > 3 else ts
    return sum(kept) / len(kept), ts[0]


def torch_fft_conv(x, w, b):
    """functional.py:60-87 of the reference with torch.fft on the GPU (yardstick only)."""
    n = x.ndim - 2
    shape = [s + (s % 2) for s in x.shape[2:]]
    dims = tuple(range(2, x.ndim))
    xf = torch.fft.rfftn(x, shape, dim=dims)
    wf = torch.fft.rfftn(w, shape, dim=dims).conj()
    yf = torch.einsum("bi...,oi...->bo...", xf, wf)
    y = torch.fft.irfftn(yf, shape, dim=dims)
    idx = (slice(None), slice(None)) + tuple(slice(0, x.shape[2 + i] - w.shape[2 + i] + 1) for i in range(n))
    return y[idx] + b.view(1, -1, *([1] * n))


CONFIGS = [
    (1, 32768, [1] + list(range(256, 4096, 512))),
    (2, 512, [1] + list(range(4, 49, 6))),
    (3, 64, [1, 2, 4, 6, 8]),
]
lines = []
for ndim, size, ks in CONFIGS:
    if args.quick:
        ks = [ks[0], ks[len(ks) // 2], ks[-1]]
    x = torch.randn(2, 8, *([size] * ndim), device=dev, requires_grad=True)
    for k in ks:
        w = torch.randn(8, 8, *([k] * ndim), device=dev, requires_grad=True)
        b = torch.randn(8, device=dev, requires_grad=True)
        direct = getattr(F, f"conv{ndim}d")
        direct_t = getattr(F, f"conv_transpose{ndim}d")
        for name, ours, theirs in (("fft_conv", lambda: fft_conv(x, w, bias=b), lambda: direct(x, w, b)),
                                   ("fft_conv_transpose", lambda: fft_conv_transpose(x, w, bias=b), lambda: direct_t(x, w, b))):
            y = ours()
            ref = theirs()
            err = float((y - ref).abs().max() / ref.abs().max())
            mean_s, best_s = measure(ours, args.iters)
            rec = {"op": name, "ndim": ndim, "size": size, "k": k, "out_elems": y.numel(), "rel_err_vs_direct": err,
                   "us_mean": mean_s * 1e6, "us_best": best_s * 1e6, "gsamples_per_s": y.numel() / mean_s / 1e9}
            alg = 4 * (x.numel() + w.numel() + b.numel() + y.numel())
            rec["hbm_roofline_frac"] = alg / mean_s / 1e9 / HBM_GBPS
            rec["direct_us_mean"] = measure(theirs, max(4, args.iters // 2))[0] * 1e6
            if name == "fft_conv":
                try:
                    with torch.no_grad():
                        rec["torch_fft_us_mean"] = measure(lambda: torch_fft_conv(x, w, b), max(4, args.iters // 2))[0] * 1e6
                except Exception as e:           # (out of memory on the largest 3-D spectra is possible)
                    rec["torch_fft_us_mean"] = None
                    rec["torch_fft_error"] = str(e)[:80]
            lines.append(rec)
            print(json.dumps(rec), flush=True)
with open(args.out, "w") as fh:
    for rec in lines:
        fh.write(json.dumps(rec) + "\n")
