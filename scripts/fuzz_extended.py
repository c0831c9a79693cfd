#!/usr/bin/env python3
"""Longer seeded fuzz run than tests/test_gpu_fuzz.py (same case generator, other seeds, more cases).
Prints every case before it runs so that a GPU fault can be attributed; exits non-zero on any mismatch."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as tf  # noqa: E402
from fft_conv_pytorch_amd.functional import fft_conv  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
counts = {1: 150, 2: 60, 3: 24}
worst = 0.0
for ndim, count in counts.items():
    rng = random.Random(seed * 1000 + ndim)
    gen = torch.Generator().manual_seed(seed * 77 + ndim)
    for n in range(count):
        c = tf._case(rng, ndim)
        print(ndim, n, c, flush=True)
        x = torch.randn(c["batch"], c["cin"], *c["size"], generator=gen, dtype=torch.float64)
        w = torch.randn(c["cout"], c["cin"] // c["groups"], *([c["k"]] * ndim), generator=gen, dtype=torch.float64)
        b = torch.randn(c["cout"], generator=gen, dtype=torch.float64)
        with_grad = n % 2 == 0
        xr, wr, br = (t.clone().requires_grad_(with_grad) for t in (x, w, b))
        want = tf._reference(c, xr, wr, br)
        xd, wd, bd = (t.float().to("cuda").requires_grad_(with_grad) for t in (x, w, b))
        try:
            got = fft_conv(xd, wd, bias=bd, stride=c["stride"], padding=c["pad"], dilation=c["dil"], groups=c["groups"],
                           padding_mode=c["mode"])
            errs = [tf._rel(got.detach(), want.detach())]
            if with_grad:
                gy = torch.randn(want.shape, generator=gen, dtype=torch.float64)
                want.backward(gy)
                got.backward(gy.float().to("cuda"))
                errs += [tf._rel(xd.grad, xr.grad), tf._rel(wd.grad, wr.grad), tf._rel(bd.grad, br.grad)]
        except NotImplementedError as exc:      # documented limits
            print("  unsupported:", str(exc)[:120], flush=True)
            continue
        if max(errs) >= 1e-4:
            print("MISMATCH", errs, flush=True)
            sys.exit(1)
        if max(errs) > 3e-6:
            print("  errs (y, dX, dW, db):", ["%.1e" % e for e in errs], flush=True)
        worst = max(worst, max(errs))
print(f"extended fuzz seed {seed}: worst rel err {worst:.2e}")

# transposed convolutions
import torch.nn.functional as F  # noqa: E402
from fft_conv_pytorch_amd.functional import fft_conv_transpose  # noqa: E402
worst_t = 0.0
for ndim, count in {1: 80, 2: 30, 3: 12}.items():
    rng = random.Random(seed * 333 + ndim)
    gen = torch.Generator().manual_seed(seed * 5 + ndim)
    convt = getattr(F, f"conv_transpose{ndim}d")
    for n in range(count):
        groups = rng.choice([1, 1, 2])
        cig, cog = rng.choice([1, 2, 4, 8, 9, 16]), rng.choice([1, 3, 8, 16])
        batch = rng.choice([1, 2, 3, 5, 8])
        k = rng.choice([1, 2, 3, 5, 33 if ndim == 1 else 4, 200 if ndim == 1 else 2])
        stride = rng.choice([1, 1, 2, 3])
        dil = rng.choice([1, 2, 3])
        pad = rng.choice([0, 1, (k - 1) * dil // 2])
        opad = rng.randint(0, max(stride, dil) - 1)
        size = [rng.randint(2, {1: 4000, 2: 60, 3: 14}[ndim]) for _ in range(ndim)]
        kw = dict(stride=stride, padding=pad, output_padding=opad, dilation=dil, groups=groups)
        print("T", ndim, n, batch, cig * groups, cog * groups, k, size, kw, flush=True)
        x = torch.randn(batch, cig * groups, *size, generator=gen, dtype=torch.float64)
        w = torch.randn(cig * groups, cog, *([k] * ndim), generator=gen, dtype=torch.float64)
        b = torch.randn(cog * groups, generator=gen, dtype=torch.float64)
        try:
            want = convt(x, w, b, **kw)
        except RuntimeError:
            continue
        try:
            got = fft_conv_transpose(x.float().to("cuda"), w.float().to("cuda"), b.float().to("cuda"), **kw)
        except NotImplementedError as exc:
            print("  unsupported:", str(exc)[:120], flush=True)
            continue
        err = tf._rel(got, want)
        if got.shape != want.shape or err >= 1e-4:
            print("MISMATCH", err, flush=True)
            sys.exit(1)
        worst_t = max(worst_t, err)
print(f"extended transposed fuzz seed {seed}: worst rel err {worst_t:.2e}")
