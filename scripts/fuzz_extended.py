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
        worst = max(worst, max(errs))
print(f"extended fuzz seed {seed}: worst rel err {worst:.2e}")
