#!/usr/bin/env python3
"""Seeded random 2-D / 3-D cases LARGER than tests/test_gpu_fuzz.py draws (planes past 64 x 64, rows past a power of two, 8-channel
blocks): the shapes that take the planner's x / y tiles, the tiled plane-major pipeline and the thread-per-sequence 2-D column pass.
Forward, dX, dW, db against torch float64 on the CPU; prints every case first; exits non-zero on a mismatch."""
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from fft_conv_pytorch_amd.functional import fft_conv, _plan_for  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = random.Random(seed)
gen = torch.Generator().manual_seed(seed)
worst = 0.0


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


for n in range(int(sys.argv[2]) if len(sys.argv) > 2 else 36):
    nd = rng.choice([2, 2, 3])
    if nd == 2:
        size = [rng.randint(60, 640) for _ in range(2)]
        k = [rng.choice([1, 3, 5, 7, 9, 15, 31]) for _ in range(2)]
        batch = rng.choice([1, 2, 4, 9])
    else:
        size = [rng.randint(20, 150) for _ in range(3)]
        k = [rng.choice([1, 2, 3, 5, 9]) for _ in range(3)]
        batch = rng.choice([1, 2, 3])
    groups = rng.choice([1, 1, 2])
    cin, cout = 8 * groups, rng.choice([8, 16]) * groups
    stride = [rng.choice([1, 1, 1, 2]) for _ in range(nd)]
    dil = [rng.choice([1, 1, 2]) for _ in range(nd)]
    mode = rng.choice(["constant", "constant", "constant", "reflect", "circular", "replicate"])
    pad = [rng.choice([0, (kk - 1) * d // 2, min(kk, s - 1)]) for kk, d, s in zip(k, dil, size)]
    if any((kk - 1) * d + 1 > s + 2 * p for kk, d, s, p in zip(k, dil, size, pad)):
        continue
    case = dict(nd=nd, batch=batch, cin=cin, cout=cout, groups=groups, size=size, k=k, stride=stride, dil=dil, pad=pad, mode=mode)
    print(n, case, flush=True)
    x = torch.randn(batch, cin, *size, generator=gen, dtype=torch.float64)
    w = torch.randn(cout, cin // groups, *k, generator=gen, dtype=torch.float64) / (cin // groups * max(1, int(torch.tensor(k).prod()))) ** 0.5
    b = torch.randn(cout, generator=gen, dtype=torch.float64)
    xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
    conv = F.conv2d if nd == 2 else F.conv3d
    if mode == "constant":
        want = conv(xr, wr, br, stride=stride, padding=pad, dilation=dil, groups=groups)
    else:
        want = conv(F.pad(xr, [q for p in reversed(pad) for q in (p, p)], mode=mode), wr, br, stride=stride, dilation=dil, groups=groups)
    gy = torch.randn(want.shape, generator=gen, dtype=torch.float64)
    want.backward(gy)
    xd, wd, bd = (t.float().to("cuda").requires_grad_() for t in (x, w, b))
    got = fft_conv(xd, wd, bias=bd, stride=tuple(stride), padding=tuple(pad), dilation=tuple(dil), groups=groups, padding_mode=mode)
    lay = _plan_for(xd, wd, bd, tuple(stride), tuple(pad), tuple(dil), groups, mode).layout
    got.backward(gy.float().to("cuda"))
    errs = [rel(got, want), rel(xd.grad, xr.grad), rel(wd.grad, wr.grad), rel(bd.grad, br.grad)]
    print("   layout", lay[:4], "pipeline", lay[7], "errs", ["%.1e" % e for e in errs], flush=True)
    if max(errs) >= 1e-4:
        print("MISMATCH", flush=True)
        sys.exit(1)
    worst = max(worst, max(errs))
print(f"large N-d fuzz seed {seed}: worst rel err {worst:.2e}")
