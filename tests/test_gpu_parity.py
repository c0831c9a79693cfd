"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, the committed golden
vectors from the real reference, and torch's direct convolution.  Tolerance: 1e-4 relative
(max|y - y_ref| / max|y_ref|), the bar BASELINE.json's north_star states for fp32."""
import itertools
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fft_conv_oracle as orc
from tests import golden_util as gu

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4
DEV = "cuda:0"


def _hip(x, w, b, **kw):
    from fft_conv_pytorch_amd.functional import fft_conv
    t = lambda a: None if a is None else torch.as_tensor(a).to(DEV)
    y = fft_conv(t(x), t(w), bias=t(b), **kw)
    torch.cuda.synchronize()
    return y.cpu().numpy()


def test_native_library_is_loaded():
    from fft_conv_pytorch_amd import _native
    lib = _native.load_library()
    assert lib.fc_version() == _native.ABI_VERSION
    with open("/proc/self/maps") as f:
        assert _native.LIB_NAME in f.read()


@pytest.mark.parametrize("ndim", [1, 2, 3])
def test_golden_g1_reference_grid(ndim):
    worst, n_cases = 0.0, 0
    for n, x, w, b, kw, y_ref in gu.g1_cases():
        if x.ndim - 2 != ndim:
            continue
        worst = max(worst, gu.check_against(_hip(x, w, b, **kw), y_ref, REL_TOL))
        n_cases += 1
    assert n_cases > 0
    print(f"G1 ndim={ndim}: {n_cases} cases, worst rel err {worst:.2e}")


def test_golden_g2_extended():
    for n, x, w, b, kw, gold in gu.g2_cases():
        err = gu.check_against(_hip(x, w, b, **kw), gold, REL_TOL)
        print(f"G2 case {n}: rel err {err:.2e}")


@pytest.mark.parametrize("name", ["cfg0", "cfgA", "cfgB", "cfgC", "cfgD_b1"])
def test_golden_g3_baseline_configs(name):
    x, w, b, kw, gold = gu.g3_case(name)
    err = gu.check_against(_hip(x, w, b, **kw), gold, REL_TOL)
    print(f"G3 {name}: rel err {err:.2e}")


def _gcd3(a, b, c):
    return math.gcd(a, math.gcd(b, c))


@pytest.mark.parametrize("ndim", [1, 2, 3])
def test_full_reference_grid_against_torch_direct(ndim):
    """The reference's whole 1,152-case grid (tests/test_functional.py:11-20), seeded, against
    torch's direct convolution on the CPU -- the same ground truth the reference uses -- with the
    reference's own absolute tolerance (benchmark_utils.py:53-57)."""
    from fft_conv_pytorch_amd.functional import fft_conv
    gen = torch.Generator().manual_seed(1234 + ndim)
    conv = getattr(F, f"conv{ndim}d")
    grid = itertools.product([2, 3], [2, 3], [1, 2, 3], [2, 3], [0, 1], [1, 2], [1, 2], [7, 8])
    count = 0
    for cin, cout, groups, k, pad, stride, dil, size in grid:
        g = _gcd3(cin, cout, groups)
        x = torch.randn(2, cin, *([size] * ndim), generator=gen)
        w = torch.randn(cout, cin // g, *([k] * ndim), generator=gen)
        b = torch.randn(cout, generator=gen)
        kw = dict(stride=stride, padding=pad, dilation=dil, groups=g)
        y = fft_conv(x.to(DEV), w.to(DEV), bias=b.to(DEV), **kw).cpu()
        y_ref = conv(x, w, bias=b, **kw)
        assert y.shape == y_ref.shape
        err = (y - y_ref).abs()
        assert err.max().item() < 1e-4 and err.mean().item() < 5e-5, (cin, cout, g, k, pad, stride, dil, size)
        count += 1
    assert count == 384


@pytest.mark.parametrize("mode", ["constant", "reflect", "replicate", "circular"])
@pytest.mark.parametrize("shape", [((300,), (17,), (8,)), ((2500,), (65,), (40,)), ((40, 37), (5, 4), (3, 2)),
                                   ((9, 12, 11), (3, 2, 3), (2, 1, 2))])
def test_padding_modes_against_oracle(mode, shape):
    spatial, ksz, pad = shape
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 4) + spatial, dtype=np.float32)
    w = rng.standard_normal((6, 2) + ksz, dtype=np.float32)
    b = rng.standard_normal(6, dtype=np.float32)
    kw = dict(padding=pad, groups=2, padding_mode=mode)
    y_ref = orc.fft_conv_oracle_torch(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), **kw).numpy()
    assert orc.rel_err(_hip(x, w, b, **kw), y_ref) < REL_TOL


@pytest.mark.parametrize("cin,cout,groups", [(1, 1, 1), (1, 5, 1), (5, 1, 1), (7, 9, 1), (16, 24, 1), (12, 20, 4),
                                             (64, 64, 8), (20, 10, 5), (9, 9, 9)])
def test_channel_blocking_1d(cin, cout, groups):
    """Odd channel counts (zero phantom channels), several input chunks (LDS accumulate path) and
    several output chunks; multi-channel groups are never exercised by the reference's own tests."""
    rng = np.random.default_rng(cin * 100 + cout)
    x = rng.standard_normal((2, cin, 3000), dtype=np.float32)
    w = rng.standard_normal((cout, cin // groups, 33), dtype=np.float32)
    b = rng.standard_normal(cout, dtype=np.float32)
    kw = dict(padding=16, groups=groups)
    y_ref = orc.fft_conv_oracle_torch(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), **kw).numpy()
    assert orc.rel_err(_hip(x, w, b, **kw), y_ref) < REL_TOL


@pytest.mark.parametrize("tile", [64, 128, 256, 512, 1024, 2048, 4096])
@pytest.mark.parametrize("stride,dilation", [(1, 1), (3, 2)])
def test_every_tile_geometry_1d(tile, stride, dilation, monkeypatch):
    """Force each FFT tile length (all (P, S) register/lane-split geometries) on a multi-tile row."""
    monkeypatch.setenv("FFTCONV_TILE", str(tile))
    k = 9
    rng = np.random.default_rng(tile)
    length = 3 * tile + 123
    x = rng.standard_normal((2, 3, length), dtype=np.float32)
    w = rng.standard_normal((5, 3, k), dtype=np.float32)
    b = rng.standard_normal(5, dtype=np.float32)
    kw = dict(stride=stride, dilation=dilation, padding=4)
    y_ref = orc.fft_conv_oracle_torch(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), **kw).numpy()
    assert orc.rel_err(_hip(x, w, b, **kw), y_ref) < REL_TOL


def test_edge_cases_1d():
    rng = np.random.default_rng(3)
    # kernel as long as the (padded) input -> single output sample
    x = rng.standard_normal((1, 2, 9), dtype=np.float32)
    w = rng.standard_normal((2, 2, 9), dtype=np.float32)
    y_ref = orc.direct_conv_float64(x, w)
    assert orc.rel_err(_hip(x, w, None), y_ref) < REL_TOL
    # 1-tap kernel, no bias
    w1 = rng.standard_normal((3, 2, 1), dtype=np.float32)
    assert orc.rel_err(_hip(x, w1, None), orc.direct_conv_float64(x, w1)) < REL_TOL
    # non-contiguous input views
    xb = rng.standard_normal((2, 4, 200), dtype=np.float32)
    from fft_conv_pytorch_amd.functional import fft_conv
    xt = torch.from_numpy(xb).to(DEV)[:, ::2, ::2]
    wt = torch.from_numpy(rng.standard_normal((3, 2, 5), dtype=np.float32)).to(DEV)
    y = fft_conv(xt, wt).cpu().numpy()
    assert orc.rel_err(y, orc.direct_conv_float64(xb[:, ::2, ::2], wt.cpu().numpy())) < REL_TOL
    # kernel larger than padded input: torch raises, the reference silently mis-shapes; we raise
    with pytest.raises(ValueError):
        fft_conv(torch.zeros(1, 1, 4, device=DEV), torch.zeros(1, 1, 6, device=DEV))


def test_linearity_and_shift_at_full_size():
    """Size-independent properties at the metric configuration (cfgA): linearity in the signal and
    translation equivariance -- no oracle needed at this size."""
    from fft_conv_pytorch_amd.functional import fft_conv
    g = torch.Generator(device=DEV).manual_seed(7)
    x1 = torch.randn(32, 8, 32768, device=DEV, generator=g)
    x2 = torch.randn(32, 8, 32768, device=DEV, generator=g)
    w = torch.randn(8, 8, 512, device=DEV, generator=g)
    y1, y2 = fft_conv(x1, w), fft_conv(x2, w)
    y12 = fft_conv(2.0 * x1 - 3.0 * x2, w)
    scale = y12.abs().max().item()
    assert (y12 - (2.0 * y1 - 3.0 * y2)).abs().max().item() / scale < REL_TOL
    ys = fft_conv(torch.roll(x1, 100, dims=-1), w)
    assert (ys[..., 100:] - y1[..., :-100]).abs().max().item() / scale < REL_TOL


def test_module_matches_functional_and_tracks_weight_updates():
    from fft_conv_pytorch_amd import FFTConv1d
    torch.manual_seed(0)
    layer = FFTConv1d(4, 6, 31, padding=15, padding_mode="reflect").to(DEV)
    x = torch.randn(3, 4, 2000, device=DEV)
    ref = lambda: F.conv1d(F.pad(x, (15, 15), mode="reflect"), layer.weight, layer.bias)
    y = layer(x)
    assert (y - ref()).abs().max().item() / ref().abs().max().item() < REL_TOL
    with torch.no_grad():
        layer.weight.mul_(0.5)          # bumps weight._version -> spectrum cache must refresh
    y2 = layer(x)
    assert (y2 - ref()).abs().max().item() / ref().abs().max().item() < REL_TOL
    sd = layer.state_dict()
    assert set(sd) == {"weight", "bias"}
    with pytest.raises(AssertionError):
        layer(x[0])                     # unbatched input is rejected like the reference (nn.py:11)


# ----------------------------------------------------------------------------- transposed convolution (row N2)
def _hip_t(x, w, b, **kw):
    from fft_conv_pytorch_amd.functional import fft_conv_transpose
    t = lambda a: None if a is None else torch.as_tensor(a).to(DEV)
    y = fft_conv_transpose(t(x), t(w), bias=t(b), **kw)
    torch.cuda.synchronize()
    return y.cpu().numpy()


def test_transpose_golden_g4():
    worst, count = 0.0, 0
    for n, x, w, b, kw, y_ref in gu.g4_cases():
        worst = max(worst, gu.check_against(_hip_t(x, w, b, **kw), y_ref, REL_TOL))
        count += 1
    print(f"G4: {count} transposed cases, worst rel err {worst:.2e}")


@pytest.mark.parametrize("ndim", [1, 2, 3])
def test_transpose_reference_grid_against_torch_direct(ndim):
    """The reference's transposed grid (tests/test_functional_transpose.py:11-21, with its
    `dilation += output_padding; stride += output_padding` rule), seeded, against torch's
    conv_transpose on the CPU with the reference's absolute tolerance."""
    from fft_conv_pytorch_amd.functional import fft_conv_transpose
    gen = torch.Generator().manual_seed(4321 + ndim)
    conv = getattr(F, f"conv_transpose{ndim}d")
    grid = itertools.product([2, 3], [2, 3], [1, 2, 3], [2, 3], [0, 1], [1, 2], [1, 2], [0, 1, 2], [7, 8])
    count = 0
    for cin, cout, groups, k, pad, stride, dil, opad, size in grid:
        g = _gcd3(cin, cout, groups)
        if ndim == 3 and (count % 3):      # thin the 3-D sweep (same coverage of every axis value)
            count += 1
            continue
        x = torch.randn(2, cin, *([size] * ndim), generator=gen)
        w = torch.randn(cin, cout // g, *([k] * ndim), generator=gen)
        b = torch.randn(cout, generator=gen)
        kw = dict(stride=stride + opad, padding=pad, output_padding=opad, dilation=dil + opad, groups=g)
        y = fft_conv_transpose(x.to(DEV), w.to(DEV), bias=b.to(DEV), **kw).cpu()
        y_ref = conv(x, w, bias=b, **kw)
        assert y.shape == y_ref.shape
        err = (y - y_ref).abs()
        assert err.max().item() < 1e-4 and err.mean().item() < 5e-5, (cin, cout, g, k, pad, stride, dil, opad, size)
        count += 1
    assert count == 1152


def test_transpose_module_and_long_rows():
    from fft_conv_pytorch_amd import FFTConvTranspose1d, FFTConvTranspose2d
    torch.manual_seed(1)
    layer = FFTConvTranspose1d(8, 8, 129, stride=2, padding=10, output_padding=1).to(DEV)
    x = torch.randn(2, 8, 5000, device=DEV)
    y = layer(x)
    ref = F.conv_transpose1d(x.cpu(), layer.weight.detach().cpu(), layer.bias.detach().cpu(), stride=2, padding=10,
                             output_padding=1)
    assert y.shape == ref.shape
    assert (y.cpu() - ref).abs().max().item() / ref.abs().max().item() < REL_TOL
    layer2 = FFTConvTranspose2d(4, 6, (5, 3), stride=(2, 3), padding=(2, 1), groups=2).to(DEV)
    x2 = torch.randn(2, 4, 33, 40, device=DEV)
    y2 = layer2(x2)
    ref2 = F.conv_transpose2d(x2.cpu(), layer2.weight.detach().cpu(), layer2.bias.detach().cpu(), stride=(2, 3),
                              padding=(2, 1), groups=2)
    assert y2.shape == ref2.shape
    assert (y2.cpu() - ref2).abs().max().item() / ref2.abs().max().item() < REL_TOL
    assert set(layer2.state_dict()) == {"weight", "bias"}


# ----------------------------------------------------------------------------- backward (row N1)
@pytest.mark.parametrize("ndim", [1, 2, 3])
def test_backward_reference_grid(ndim):
    """dW and db as the reference pins them (tests/test_functional.py:62-117), plus dX, against torch's
    direct convolution on the CPU, over the reference's grid (thinned for 2-D/3-D)."""
    from fft_conv_pytorch_amd.functional import fft_conv
    gen = torch.Generator().manual_seed(99 + ndim)
    conv = getattr(F, f"conv{ndim}d")
    grid = list(itertools.product([2, 3], [2, 3], [1, 2, 3], [2, 3], [0, 1], [1, 2], [1, 2], [7, 8]))
    step = {1: 1, 2: 3, 3: 11}[ndim]
    for cin, cout, groups, k, pad, stride, dil, size in grid[::step]:
        g = _gcd3(cin, cout, groups)
        x = torch.randn(2, cin, *([size] * ndim), generator=gen)
        w = torch.randn(cout, cin // g, *([k] * ndim), generator=gen)
        b = torch.randn(cout, generator=gen)
        kw = dict(stride=stride, padding=pad, dilation=dil, groups=g)
        xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, w, b))
        xc, wc, bc = (t.clone().requires_grad_() for t in (x, w, b))
        fft_conv(xd, wd, bias=bd, **kw).sum().backward()
        conv(xc, wc, bias=bc, **kw).sum().backward()
        for got, ref in ((xd.grad, xc.grad), (wd.grad, wc.grad), (bd.grad, bc.grad)):
            err = (got.cpu() - ref).abs()
            assert err.max().item() < 1e-4 * max(1.0, ref.abs().max().item()), (cin, cout, g, k, pad, stride, dil, size)


@pytest.mark.parametrize("mode", ["constant", "reflect", "replicate", "circular"])
def test_backward_long_rows_and_padding_modes(mode):
    """Chunked dW path (gradient longer than one tile), multi-channel groups, every padding mode."""
    from fft_conv_pytorch_amd import FFTConv1d
    torch.manual_seed(5)
    layer = FFTConv1d(8, 12, 65, stride=2, padding=20, dilation=2, groups=4, padding_mode="zeros" if mode == "constant" else mode).to(DEV)
    ref = torch.nn.Conv1d(8, 12, 65, stride=2, padding=20, dilation=2, groups=4, padding_mode="zeros" if mode == "constant" else mode)
    ref.load_state_dict({k: v.cpu() for k, v in layer.state_dict().items()})
    x = torch.randn(3, 8, 9000)
    xd = x.to(DEV).requires_grad_()
    xc = x.clone().requires_grad_()
    gy = torch.randn_like(ref(xc))
    layer(xd).backward(gy.to(DEV))
    ref(xc).backward(gy)
    for got, want in ((xd.grad, xc.grad), (layer.weight.grad, ref.weight.grad), (layer.bias.grad, ref.bias.grad)):
        assert (got.cpu() - want).abs().max().item() / want.abs().max().item() < REL_TOL


WGRAD_CASES = [  # B, Cin, Cout, groups, L, K, padding, dilation, mode
    (32, 8, 8, 1, 32768, 512, 0, 1, "constant"),      # cfgA
    (3, 3, 5, 1, 2000, 17, 8, 1, "constant"),         # odd channel counts inside one 4 x 4 block pair
    (1, 8, 8, 1, 700, 33, 0, 3, "constant"),          # single tile, dilation 3
    (5, 16, 24, 4, 5000, 129, 64, 2, "reflect"),      # groups, 4 in / 6 out per group
    (2, 6, 6, 2, 4096, 1, 0, 1, "constant"),          # 1-tap kernel
    (7, 8, 8, 1, 3000, 385, 100, 2, "circular"),      # dilated extent 769: two segments of taps
    (4, 7, 8, 1, 1500, 200, 30, 1, "replicate"),
    (9, 8, 8, 8, 10000, 65, 5, 1, "constant"),        # depthwise-like: one channel per group
    (2, 24, 40, 1, 3000, 65, 0, 1, "constant"),       # 6 x 10 blocks of 4 x 4 channels
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_weight_gradient_kernel(case):
    """fc_wgrad1d (cross-spectra accumulated on chip; long kernels in segments of taps) against autograd through
    torch's direct convolution in float64."""
    from fft_conv_pytorch_amd import autograd as ag
    B, cin, cout, groups, L, K, pad, dil, mode = case
    gen = torch.Generator().manual_seed(1234 + L)
    x = torch.randn(B, cin, L, generator=gen)
    w = torch.randn(cout, cin // groups, K, generator=gen, dtype=torch.float64, requires_grad=True)
    xp = F.pad(x.double(), [pad, pad], mode=mode) if (mode != "constant" and pad) else x.double()
    y = F.conv1d(xp, w, None, padding=pad if mode == "constant" else 0, dilation=dil, groups=groups)
    gy = torch.randn(y.shape, generator=gen, dtype=torch.float64)
    (want,) = torch.autograd.grad(y, w, gy)
    got = ag._grad_weight(x.to(DEV), gy.float().to(DEV), tuple(w.shape), (1,), (pad,), (dil,), groups, mode)
    covered = ag._grad_weight_native(x.to(DEV), gy.float().to(DEV), tuple(w.shape), (1,), (pad,), (dil,), groups, mode)
    assert covered is not None
    assert got.shape == want.shape
    err = (got.double().cpu() - want).norm().item() / want.norm().item()
    print(f"wgrad {case}: rel err {err:.2e}")
    assert err < REL_TOL


WIDE_CASES = [  # B, Cin, Cout, groups, L, K, padding, mode  (more than 8 input channels per group, k >= 97)
    (8, 64, 64, 1, 16384, 129, 0, "constant"),
    (3, 16, 8, 1, 5000, 100, 16, "reflect"),
    (5, 24, 16, 1, 3000, 200, 7, "circular"),
    (2, 20, 8, 1, 2500, 700, 0, "constant"),          # 20 channels: the last chunk is half empty
    (4, 32, 32, 2, 9000, 1025, 100, "replicate"),     # 2048 tile
    (7, 9, 8, 1, 1200, 97, 2, "constant"),            # odd batch: a one-item remainder
]


@pytest.mark.parametrize("case", WIDE_CASES)
def test_wide_input_kernel(case):
    """conv1d_wide_kernel (running sums of the out-chunk in registers over the input chunks) against torch's
    direct convolution in float64, and against the general kernel it replaces (FFTCONV_WIDE=0 plans)."""
    from fft_conv_pytorch_amd.functional import fft_conv
    B, cin, cout, groups, L, K, pad, mode = case
    gen = torch.Generator().manual_seed(4321 + L)
    x = torch.randn(B, cin, L, generator=gen)
    w = torch.randn(cout, cin // groups, K, generator=gen) / (K * cin) ** 0.5
    b = torch.randn(cout, generator=gen)
    xp = F.pad(x.double(), [pad, pad], mode=mode) if (mode != "constant" and pad) else x.double()
    want = F.conv1d(xp, w.double(), b.double(), padding=pad if mode == "constant" else 0, groups=groups)
    got = fft_conv(x.to(DEV), w.to(DEV), b.to(DEV), padding=pad, padding_mode=mode, groups=groups)
    assert got.shape == want.shape
    err = (got.double().cpu() - want).norm().item() / want.norm().item()
    print(f"wide {case}: rel err {err:.2e}")
    assert err < REL_TOL


def test_large_grouped_dilated_rows_sampled():
    """cfgD at twice the per-GPU shard (B=16, 64->64 channels in 8 groups, 2^20 samples, k=257, dilation 4;
    4.3 GB in, 4.3 GB out): exercises the 64-bit addressing of the phase-decomposed batch-sharing kernel.
    Checked on 300 sampled outputs against a float64 dot product."""
    from fft_conv_pytorch_amd.functional import fft_conv
    B, C, G, L, K, D = 16, 64, 8, 1 << 20, 257, 4
    gen = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(B, C, L, device=DEV, generator=gen)
    w = torch.randn(C, C // G, K, device=DEV, generator=gen) / (K * C // G) ** 0.5
    b = torch.randn(C, device=DEV, generator=gen)
    y = fft_conv(x, w, b, dilation=D, groups=G)
    Lout = L - (K - 1) * D
    assert y.shape == (B, C, Lout)
    rng = torch.Generator().manual_seed(12)
    bs = torch.randint(0, B, (300,), generator=rng)
    cs = torch.randint(0, C, (300,), generator=rng)
    ts = torch.randint(0, Lout, (300,), generator=rng)
    ts[:8] = torch.tensor([0, 1, 2, 3, Lout - 1, Lout - 2, Lout - 3, Lout - 4])     # both ends, every phase
    cig = C // G
    worst = 0.0
    taps = torch.arange(K, device=DEV) * D
    for bi, co, t in zip(bs.tolist(), cs.tolist(), ts.tolist()):
        g = co // cig
        seg = x[bi, g * cig:(g + 1) * cig][:, t + taps].double()                  # (cig, K)
        want = (seg * w[co].double()).sum().item() + b[co].double().item()
        got = y[bi, co, t].item()
        worst = max(worst, abs(got - want) / max(1.0, abs(want)))
    print(f"large rows: worst sampled error {worst:.2e}")
    assert worst < REL_TOL


def test_long_dilated_kernel_with_many_channels_forward_and_backward():
    """Dilated extent 2098 needs the 4096 tile, which has no room for the running sums of a second input
    chunk: the plan launches the general kernel once per chunk (later chunks add into y).  The backward of
    a 4 -> 16 convolution needs exactly that for dX (16 'input' channels of the transposed plan)."""
    from fft_conv_pytorch_amd.functional import fft_conv
    gen = torch.Generator().manual_seed(31)
    for cin, cout in ((4, 16), (20, 8)):
        x = torch.randn(2, cin, 5317, generator=gen, dtype=torch.float64)
        w = torch.randn(cout, cin, 700, generator=gen, dtype=torch.float64) / (700 * cin) ** 0.5
        b = torch.randn(cout, generator=gen, dtype=torch.float64)
        xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
        want = F.conv1d(xr, wr, br, dilation=3, padding=100)
        xd, wd, bd = (t.float().to(DEV).requires_grad_() for t in (x, w, b))
        got = fft_conv(xd, wd, bias=bd, dilation=3, padding=100)
        gy = torch.randn(want.shape, generator=gen, dtype=torch.float64)
        want.backward(gy)
        got.backward(gy.float().to(DEV))
        for a_, b_ in ((got, want), (xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
            err = (a_.detach().double().cpu() - b_.detach()).norm().item() / b_.detach().norm().item()
            assert err < REL_TOL, (cin, cout, err)


@pytest.mark.parametrize("batch,mode", [(1, "constant"), (3, "reflect")])
def test_small_batch_long_rows_use_tile_slots(batch, mode):
    """Fewer batch items than slots of a work item: the slots become consecutive tiles of one batch item
    (same spectrum sharing).  Long 8 -> 8 rows against torch's direct convolution in float64."""
    from fft_conv_pytorch_amd.functional import fft_conv
    gen = torch.Generator().manual_seed(71 + batch)
    L, K, pad = 150001, 257, 100
    x = torch.randn(batch, 8, L, generator=gen)
    w = torch.randn(8, 8, K, generator=gen) / (8 * K) ** 0.5
    b = torch.randn(8, generator=gen)
    xp = F.pad(x.double(), [pad, pad], mode=mode) if mode != "constant" else x.double()
    want = F.conv1d(xp, w.double(), b.double(), padding=pad if mode == "constant" else 0)
    got = fft_conv(x.to(DEV), w.to(DEV), b.to(DEV), padding=pad, padding_mode=mode)
    assert got.shape == want.shape
    assert (got.double().cpu() - want).norm().item() / want.norm().item() < REL_TOL


DEPTHWISE_CASES = [  # B, C, L, K, padding, dilation, mode
    (8, 64, 20000, 1025, 0, 1, "constant"),
    (3, 24, 5000, 33, 5, 1, "reflect"),
    (1, 8, 100000, 257, 0, 1, "constant"),       # batch 1: tile slots
    (5, 16, 9000, 129, 64, 3, "circular"),       # dilation as phases on depthwise blocks
    (2, 40, 3000, 700, 10, 1, "replicate"),
    (4, 12, 2000, 65, 0, 1, "constant"),         # 12 channels: the second block is half empty
    (3, 5, 3000, 129, 7, 1, "reflect"),          # one block, 5 of 8 channels (odd count: a lone channel in a pair)
    (2, 21, 2500, 300, 0, 2, "constant"),        # 21 channels, dilation phases
]


@pytest.mark.parametrize("case", DEPTHWISE_CASES)
def test_depthwise_blocks(case):
    """groups == channels: 8-channel blocks on the batch-sharing kernel with a per-channel mix; forward,
    transposed (dX) and weight gradient against torch's direct convolution in float64."""
    from fft_conv_pytorch_amd.functional import fft_conv
    B, C, L, K, pad, dil, mode = case
    gen = torch.Generator().manual_seed(900 + C + K)
    x = torch.randn(B, C, L, generator=gen, dtype=torch.float64)
    w = torch.randn(C, 1, K, generator=gen, dtype=torch.float64) / K ** 0.5
    b = torch.randn(C, generator=gen, dtype=torch.float64)
    xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
    xp = F.pad(xr, [pad, pad], mode=mode) if (mode != "constant" and pad) else xr
    want = F.conv1d(xp, wr, br, padding=pad if mode == "constant" else 0, dilation=dil, groups=C)
    xd, wd, bd = (t.float().to(DEV).requires_grad_() for t in (x, w, b))
    got = fft_conv(xd, wd, bias=bd, padding=pad, dilation=dil, groups=C, padding_mode=mode)
    gy = torch.randn(want.shape, generator=gen, dtype=torch.float64)
    want.backward(gy)
    got.backward(gy.float().to(DEV))
    for a_, b_ in ((got, want), (xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
        err = (a_.detach().double().cpu() - b_.detach()).norm().item() / b_.detach().norm().item()
        assert err < REL_TOL, (case, err)


LONG_KERNEL_CASES = [  # B, Cin, Cout, groups, L, K, padding, dilation, mode
    (2, 8, 8, 1, 20000, 5000, 0, 1, "constant"),          # batch-sharing kernel, 5 segments
    (3, 4, 6, 1, 12000, 3000, 50, 2, "reflect"),          # general kernel, dilated extent 5999
    (2, 16, 16, 16, 9000, 2048, 0, 1, "constant"),        # depthwise blocks, 2 segments
    (2, 12, 8, 1, 15000, 4500, 100, 1, "circular"),       # two input chunks + segments
    (1, 8, 8, 1, 8192, 8192, 4096, 1, "constant"),        # kernel as long as the row
]


@pytest.mark.parametrize("case", LONG_KERNEL_CASES)
def test_long_kernels_run_in_segments(case):
    """Kernels longer than the largest FFT tile (and 8-channel shapes beyond 1537 taps) run as segments of taps
    that accumulate into y; forward and all three gradients against torch's direct convolution in float64."""
    from fft_conv_pytorch_amd.functional import fft_conv
    B, cin, cout, groups, L, K, pad, dil, mode = case
    gen = torch.Generator().manual_seed(333 + K)
    x = torch.randn(B, cin, L, generator=gen, dtype=torch.float64)
    w = torch.randn(cout, cin // groups, K, generator=gen, dtype=torch.float64) / (K * cin // groups) ** 0.5
    b = torch.randn(cout, generator=gen, dtype=torch.float64)
    xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
    xp = F.pad(xr, [pad, pad], mode=mode) if (mode != "constant" and pad) else xr
    want = F.conv1d(xp, wr, br, padding=pad if mode == "constant" else 0, dilation=dil, groups=groups)
    xd, wd, bd = (t.float().to(DEV).requires_grad_() for t in (x, w, b))
    got = fft_conv(xd, wd, bias=bd, padding=pad, dilation=dil, groups=groups, padding_mode=mode)
    assert got.shape == want.shape
    gy = torch.randn(want.shape, generator=gen, dtype=torch.float64)
    want.backward(gy)
    got.backward(gy.float().to(DEV))
    for name, a_, b_ in (("y", got, want), ("dX", xd.grad, xr.grad), ("dW", wd.grad, wr.grad), ("db", bd.grad, br.grad)):
        err = (a_.detach().double().cpu() - b_.detach()).norm().item() / b_.detach().norm().item()
        assert err < REL_TOL, (case, name, err)


SMALL_GROUP_CASES = [  # B, C, group size, L, K, padding, dilation, mode
    (4, 16, 2, 20000, 257, 0, 1, "constant"),
    (3, 32, 4, 9000, 129, 30, 1, "reflect"),
    (5, 8, 4, 5000, 513, 0, 2, "constant"),
    (2, 24, 2, 3000, 65, 10, 1, "circular"),
    (2, 12, 4, 3000, 65, 0, 1, "constant"),        # 3 groups of 4: not a whole number of 8-channel blocks -> generic plan
]


@pytest.mark.parametrize("case", SMALL_GROUP_CASES)
def test_small_groups_as_block_diagonal_blocks(case):
    """Groups of 2 or 4 channels are regrouped into 8 x 8 blocks with a block-diagonal spectrum and run on the
    batch-sharing kernel; forward and gradients against torch's direct convolution in float64."""
    from fft_conv_pytorch_amd.functional import fft_conv
    B, C, gs, L, K, pad, dil, mode = case
    groups = C // gs
    gen = torch.Generator().manual_seed(77 + C + K)
    x = torch.randn(B, C, L, generator=gen, dtype=torch.float64)
    w = torch.randn(C, gs, K, generator=gen, dtype=torch.float64) / (K * gs) ** 0.5
    b = torch.randn(C, generator=gen, dtype=torch.float64)
    xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
    xp = F.pad(xr, [pad, pad], mode=mode) if (mode != "constant" and pad) else xr
    want = F.conv1d(xp, wr, br, padding=pad if mode == "constant" else 0, dilation=dil, groups=groups)
    xd, wd, bd = (t.float().to(DEV).requires_grad_() for t in (x, w, b))
    got = fft_conv(xd, wd, bias=bd, padding=pad, dilation=dil, groups=groups, padding_mode=mode)
    gy = torch.randn(want.shape, generator=gen, dtype=torch.float64)
    want.backward(gy)
    got.backward(gy.float().to(DEV))
    for name, a_, b_ in (("y", got, want), ("dX", xd.grad, xr.grad), ("dW", wd.grad, wr.grad), ("db", bd.grad, br.grad)):
        err = (a_.detach().double().cpu() - b_.detach()).norm().item() / b_.detach().norm().item()
        assert err < REL_TOL, (case, name, err)
