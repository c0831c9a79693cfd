"""Seeded random sweep of forward / transposed / backward shapes against torch's direct convolutions
(float64 on the CPU).  Complements the reference's grids with batch sizes, lengths and channel counts that
exercise the batch-sharing remainders, border tiles, channel chunking and the weight-gradient kernel."""
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REL_TOL = 1e-4          # BASELINE.json: within 1e-4 rel fp32


def _rel(a, b):
    """Relative L2 error; a reference that is identically zero (e.g. a gradient that only ever meets the zero
    padding) is compared absolutely instead."""
    diff = a.double().cpu() - b
    ref = b.norm().item()
    if ref < 1e-9:
        return diff.abs().max().item()
    return diff.norm().item() / ref


def _case(rng, ndim):
    groups = rng.choice([1, 1, 1, 2, 4])
    cig = rng.choice([1, 2, 3, 4, 8, 8, 9, 16])
    cog = rng.choice([1, 2, 4, 8, 8, 12, 16])
    batch = rng.choice([1, 2, 3, 4, 5, 7, 8, 9])
    if ndim == 1:
        k = rng.choice([1, 2, 3, 16, 33, 100, 257, 512, 700])
        dil = rng.choice([1, 1, 2, 3])
        size = [rng.randint((k - 1) * dil + 1, 6000)]
    elif ndim == 2:
        k = rng.choice([1, 2, 3, 5, 8])
        dil = rng.choice([1, 1, 2])
        size = [rng.randint((k - 1) * dil + 1, 90) for _ in range(2)]
        batch = min(batch, 5)
    else:
        k = rng.choice([1, 2, 3, 4])
        dil = rng.choice([1, 1, 2])
        size = [rng.randint((k - 1) * dil + 1, 24) for _ in range(3)]
        batch = min(batch, 3)
        cig, cog = min(cig, 8), min(cog, 8)
    stride = rng.choice([1, 1, 1, 2, 3])
    pad = rng.choice([0, 0, 1, (k - 1) * dil // 2, k])
    mode = rng.choice(["constant", "constant", "reflect", "replicate", "circular"])
    if mode == "reflect":
        pad = min(pad, min(size) - 1)
    if mode == "circular":
        pad = min(pad, min(size))
    return dict(ndim=ndim, batch=batch, cin=cig * groups, cout=cog * groups, groups=groups, k=k, dil=dil, size=size,
                stride=stride, pad=pad, mode=mode)


def _reference(c, x, w, b):
    conv = getattr(F, f"conv{c['ndim']}d")
    if c["mode"] != "constant" and c["pad"]:
        x = F.pad(x, [c["pad"]] * (2 * c["ndim"]), mode=c["mode"])
        return conv(x, w, b, stride=c["stride"], dilation=c["dil"], groups=c["groups"])
    return conv(x, w, b, stride=c["stride"], padding=c["pad"], dilation=c["dil"], groups=c["groups"])


@pytest.mark.parametrize("ndim,count", [(1, 60), (2, 30), (3, 16)])
def test_forward_and_backward_fuzz(ndim, count):
    from fft_conv_pytorch_amd.functional import fft_conv
    rng = random.Random(20260 + ndim)
    gen = torch.Generator().manual_seed(77 + ndim)
    worst = 0.0
    for n in range(count):
        c = _case(rng, ndim)
        x = torch.randn(c["batch"], c["cin"], *c["size"], generator=gen, dtype=torch.float64)
        w = torch.randn(c["cout"], c["cin"] // c["groups"], *([c["k"]] * ndim), generator=gen, dtype=torch.float64)
        b = torch.randn(c["cout"], generator=gen, dtype=torch.float64)
        with_grad = n % 3 == 0
        xr, wr, br = (t.clone().requires_grad_(with_grad) for t in (x, w, b))
        want = _reference(c, xr, wr, br)
        xd, wd, bd = (t.float().to(DEV).requires_grad_(with_grad) for t in (x, w, b))
        got = fft_conv(xd, wd, bias=bd, stride=c["stride"], padding=c["pad"], dilation=c["dil"], groups=c["groups"],
                       padding_mode=c["mode"])
        assert got.shape == want.shape, c
        errs = [_rel(got.detach(), want.detach())]
        if with_grad:
            gy = torch.randn(want.shape, generator=gen, dtype=torch.float64)
            want.backward(gy)
            got.backward(gy.float().to(DEV))
            errs += [_rel(xd.grad, xr.grad), _rel(wd.grad, wr.grad), _rel(bd.grad, br.grad)]
        assert max(errs) < REL_TOL, (c, errs)
        worst = max(worst, max(errs))
    print(f"fuzz {ndim}-D: {count} cases, worst rel err {worst:.2e}")


@pytest.mark.parametrize("ndim,count", [(1, 30), (2, 12), (3, 6)])
def test_transposed_fuzz(ndim, count):
    from fft_conv_pytorch_amd.functional import fft_conv_transpose
    rng = random.Random(9090 + ndim)
    gen = torch.Generator().manual_seed(5 + ndim)
    convt = getattr(F, f"conv_transpose{ndim}d")
    worst = 0.0
    for _ in range(count):
        groups = rng.choice([1, 1, 2])
        cig, cog = rng.choice([1, 2, 4, 8, 9]), rng.choice([1, 3, 8])
        batch = rng.choice([1, 2, 3, 5, 8])
        k = rng.choice([1, 2, 3, 5, 33 if ndim == 1 else 4])
        stride = rng.choice([1, 1, 2, 3])
        dil = rng.choice([1, 2])
        pad = rng.choice([0, 1, (k - 1) * dil // 2])
        opad = rng.randint(0, max(stride, dil) - 1)
        size = [rng.randint(2, {1: 3000, 2: 60, 3: 14}[ndim]) for _ in range(ndim)]
        x = torch.randn(batch, cig * groups, *size, generator=gen, dtype=torch.float64)
        w = torch.randn(cig * groups, cog, *([k] * ndim), generator=gen, dtype=torch.float64)
        b = torch.randn(cog * groups, generator=gen, dtype=torch.float64)
        kw = dict(stride=stride, padding=pad, output_padding=opad, dilation=dil, groups=groups)
        try:
            want = convt(x, w, b, **kw)
        except RuntimeError:
            continue                      # torch rejects the combination (e.g. non-positive output size)
        got = fft_conv_transpose(x.float().to(DEV), w.float().to(DEV), b.float().to(DEV), **kw)
        assert got.shape == want.shape, kw
        err = _rel(got, want)
        assert err < REL_TOL, (kw, size, err)
        worst = max(worst, err)
    print(f"transposed fuzz {ndim}-D: worst rel err {worst:.2e}")


@pytest.mark.parametrize("size,k,stride,pad", [
    ([1, 18, 10], 1, 2, 1),     # found by scripts/fuzz_extended.py seed 22: the only z plane of dX is cropped away
    ([1, 7, 9], 1, 3, 1),
    ([2, 6, 5], 1, 2, 1),
    ([1, 1, 12], 1, 2, 1),
])
def test_input_gradient_when_an_axis_only_meets_padding(size, k, stride, pad):
    """Strided 3-D convolution whose taps along a one-sample axis only ever land on the zero padding: the forward
    output is the bias and dX is exactly zero there.  The dX plan (transposed, left crop -1) has a padded z extent
    of one plane, which used to skip the z-axis source map and read the plane it should have cropped."""
    from fft_conv_pytorch_amd.functional import fft_conv
    gen = torch.Generator().manual_seed(2222)
    x = torch.randn(3, 3, *size, generator=gen, dtype=torch.float64)
    w = torch.randn(8, 3, k, k, k, generator=gen, dtype=torch.float64)
    b = torch.randn(8, generator=gen, dtype=torch.float64)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    want = F.conv3d(xr, wr, br, stride=stride, padding=pad)
    xd, wd, bd = (t.float().to(DEV).requires_grad_(True) for t in (x, w, b))
    got = fft_conv(xd, wd, bias=bd, stride=stride, padding=pad)
    gy = torch.randn(want.shape, generator=gen, dtype=torch.float64)
    want.backward(gy)
    got.backward(gy.float().to(DEV))
    errs = [_rel(got.detach(), want.detach()), _rel(xd.grad, xr.grad), _rel(wd.grad, wr.grad), _rel(bd.grad, br.grad)]
    assert max(errs) < REL_TOL, errs
