"""GPU tests of the many-channel 1-D pipeline (csrc/dense1d.hpp): forward transforms -> one real GEMM per frequency
bin on v_mfma_f32_16x16x4_f32 -> inverse transforms, against torch's direct convolution in float64 (the truth the
reference's own tests use, /root/reference/tests/test_functional.py:62-117) and against the fused kernels.
Every call goes through the C ABI (fft_conv_pytorch_amd._native)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4      # north_star bound (fp32, relative to the tensor's max magnitude)
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _truth(x, w, b, groups=1, padding=0, padding_mode="constant", dilation=1):
    xd = x.detach().double()
    if padding_mode != "constant" and padding:
        xd = F.pad(xd, (padding, padding), mode=padding_mode)
        padding = 0
    return F.conv1d(xd, w.detach().double(), None if b is None else b.detach().double(), padding=padding, groups=groups,
                    dilation=dilation)


# batch, cin, cout, groups, L, k, bias, kwargs -- channel counts that are / are not multiples of 8 and 64, one and two
# K chunks (Cin/g > 64), every column-block width of the GEMM (Cout/g 16..31 -> 2 tiles, 32..63 -> 4, >= 64 -> 8), a
# partly filled last row block (M = batch x tiles not a multiple of 128) and last column block
CASES = [
    (8, 64, 64, 1, 4000, 129, True, {}),
    (3, 24, 40, 1, 5000, 65, True, dict(padding=7, padding_mode="reflect")),
    (2, 32, 48, 2, 3000, 33, False, dict(dilation=3, padding=40, padding_mode="circular")),
    (1, 16, 16, 1, 1200, 200, True, dict(padding=3)),
    (2, 144, 96, 1, 2048, 257, True, dict(padding=128)),                      # two K chunks (64 + 64 + 16), two column blocks
    (5, 17, 19, 1, 1500, 31, True, dict(padding=15, padding_mode="replicate")),   # odd channel counts: padded to 24 / 24
    (40, 32, 32, 1, 4096, 65, True, {}),                                        # M = 200 rows: two row blocks
    (2, 80, 72, 1, 1024, 769, False, dict(padding=400)),                        # the longest kernel the 1024 tile takes
    (3, 32, 40, 1, 6000, 1025, True, dict(padding=512)),                        # 2048 tile (lane-split transforms)
    (2, 48, 32, 2, 5000, 385, True, dict(dilation=4, padding=100, padding_mode="reflect")),   # dilated extent 1537: its longest
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[1]}to{c[2]}g{c[3]}k{c[5]}" for c in CASES])
def test_many_channel_pipeline_matches_direct_convolution(case, monkeypatch):
    from fft_conv_pytorch_amd import _native
    from fft_conv_pytorch_amd.functional import fft_conv, _plan_for  # noqa: F401
    B, ci, co, g, L, k, has_bias, kw = case
    gen = torch.Generator().manual_seed(ci * 1000 + co)
    x = torch.randn(B, ci, L, generator=gen).to(DEV)
    w = (torch.randn(co, ci // g, k, generator=gen) / math.sqrt(ci // g * k)).to(DEV)
    b = torch.randn(co, generator=gen).to(DEV) if has_bias else None
    want = _truth(x, w, b, groups=g, **kw)
    outs = {}
    for mode in ("2", "0"):                     # forced many-channel pipeline / fused kernels only
        monkeypatch.setenv("FFTCONV_DENSE", mode)
        _native.clear_plan_cache()
        outs[mode] = fft_conv(x, w, b, groups=g, **kw)
    monkeypatch.delenv("FFTCONV_DENSE", raising=False)
    _native.clear_plan_cache()
    assert _rel(outs["2"], want) < REL_TOL, case
    assert _rel(outs["0"], want) < REL_TOL, case
    assert _rel(outs["2"], outs["0"]) < 5e-6, case      # two fp32 formulations of the same sums


def test_many_channel_plan_reports_its_layout_and_runs_in_slabs(monkeypatch):
    """The plan says so in its layout words (a spectrum of the fused kernels must not be fed to it), and a workspace
    of a few rows at a time (FFTCONV_DENSE_SLAB: 3 rows per slab, 29 slabs here) gives the same result bit for bit."""
    from fft_conv_pytorch_amd import _native
    from fft_conv_pytorch_amd.functional import fft_conv
    import fft_conv_pytorch_amd as fca
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(6, 64, 9000, generator=gen).to(DEV)
    layer = fca.FFTConv1d(64, 64, 257, padding=100, bias=True).to(DEV).eval()
    monkeypatch.setenv("FFTCONV_DENSE", "1")
    _native.clear_plan_cache()
    with torch.no_grad():
        y1 = layer(x)
    plan = layer.__dict__["_spectrum_cache"][1].plan
    assert plan.layout[6] == 2 and plan.workspace_bytes > 0, plan.layout       # picked on its own: 64 x 64 channels, M = 72
    monkeypatch.setenv("FFTCONV_DENSE_SLAB", "3")
    _native.clear_plan_cache()
    layer2 = fca.FFTConv1d(64, 64, 257, padding=100, bias=True).to(DEV).eval()
    layer2.load_state_dict(layer.state_dict())
    with torch.no_grad():
        y2 = layer2(x)
    monkeypatch.delenv("FFTCONV_DENSE_SLAB", raising=False)
    monkeypatch.delenv("FFTCONV_DENSE", raising=False)
    _native.clear_plan_cache()
    assert torch.equal(y1, y2)
    assert _rel(y1, _truth(x, layer.weight, layer.bias, padding=100)) < REL_TOL
    # 8 channels per group stay with the fused batch-sharing kernel
    small = fca.FFTConv1d(64, 64, 257, groups=8).to(DEV).eval()
    with torch.no_grad():
        small(x)
    assert small.__dict__["_spectrum_cache"][1].plan.layout[6] == 0


def test_many_channel_pipeline_gradients_and_transposed_form(monkeypatch):
    """Autograd through plans that take the many-channel pipeline (dX = a transposed plan over the same weights, dW =
    the fused weight-gradient kernel or the plan fallback), and fft_conv_transpose with stride 1; truth = torch."""
    from fft_conv_pytorch_amd import _native
    from fft_conv_pytorch_amd.functional import fft_conv, fft_conv_transpose
    monkeypatch.setenv("FFTCONV_DENSE", "2")
    _native.clear_plan_cache()
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(4, 48, 3000, generator=gen).to(DEV).requires_grad_()
    w = (torch.randn(40, 24, 97, generator=gen) / math.sqrt(24 * 97)).to(DEV).requires_grad_()
    b = torch.randn(40, generator=gen).to(DEV).requires_grad_()
    y = fft_conv(x, w, b, groups=2, padding=48)
    gy = torch.randn(y.shape, generator=gen).to(DEV)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), gy)
    xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
    yd = F.conv1d(xd, wd, bd, padding=48, groups=2)
    gxd, gwd, gbd = torch.autograd.grad(yd, (xd, wd, bd), gy.double())
    assert _rel(y, yd) < REL_TOL and _rel(gx, gxd) < REL_TOL and _rel(gw, gwd) < REL_TOL and _rel(gb, gbd) < REL_TOL
    xt = torch.randn(3, 32, 2000, generator=gen).to(DEV)
    wt = (torch.randn(32, 20, 65, generator=gen) / math.sqrt(32 * 65)).to(DEV)      # (Cin, Cout / groups, k)
    yt = fft_conv_transpose(xt, wt, None, groups=2, padding=10)
    want = F.conv_transpose1d(xt.double(), wt.double(), None, groups=2, padding=10)
    monkeypatch.delenv("FFTCONV_DENSE", raising=False)
    _native.clear_plan_cache()
    assert _rel(yt, want) < REL_TOL
