"""GPU tests added in round 3: the plane-major 3-D pipeline against the separable passes and torch, the bias gradient
folded into the weight-gradient launch, the explicit padding adjoints of the backward, plan creation versus stream
capture, the 32-bit guards of the many-channel plan, the float64 FFT path and the planner's small-batch choice.
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


# ----------------------------------------------------------------------------- plane-major 3-D pipeline (row N3)
PLANE_CASES = [
    # B, Cin, Cout, groups, size, k, stride, padding, dilation, mode
    (2, 8, 8, 1, (64, 64, 64), (9, 9, 9), 1, 0, 1, "constant"),              # cfgC at batch 2
    (1, 8, 8, 1, (64, 64, 64), (2, 4, 8), 1, 0, 1, "constant"),              # README shape, one batch item (NB = 1 build)
    (3, 5, 7, 1, (40, 50, 60), (3, 4, 5), 1, 1, 1, "constant"),              # odd batch, ragged channel counts
    (2, 8, 16, 1, (30, 33, 47), (5, 3, 2), (2, 1, 3), (2, 1, 0), 1, "constant"),   # strides, two output chunks
    (2, 16, 8, 2, (20, 60, 62), (3, 3, 3), 1, 1, 1, "reflect"),              # groups, index-map padding
    (2, 8, 8, 1, (21, 40, 40), (3, 5, 5), 1, 2, (1, 2, 3), "circular"),      # dilation
    (2, 6, 6, 1, (17, 33, 20), (2, 3, 3), (1, 2, 1), 1, 1, "replicate"),
    (2, 8, 8, 1, (150, 40, 40), (9, 3, 3), 1, 0, 1, "constant"),             # several overlap-save tiles along z
    (1, 8, 8, 1, (100, 20, 20), (33, 3, 3), (3, 1, 1), 4, 1, "constant"),    # longest z kernel of the path, z stride
    (2, 8, 8, 1, (40, 100, 130), (3, 5, 5), 1, 2, 1, "constant"),            # planes larger than 64 x 64: 2 x 3 overlap-save tiles per plane
    (1, 8, 8, 1, (64, 64, 64), (3, 3, 3), 1, 1, 1, "constant"),              # 'same' padding on 64^3: 66 samples = two tiles on every axis
    (2, 8, 16, 1, (30, 70, 90), (5, 3, 7), (1, 2, 3), (2, 1, 3), 1, "reflect"),   # tiles with strides and index-map padding
    (1, 8, 8, 1, (20, 200, 70), (3, 9, 3), 1, (1, 0, 4), (1, 2, 1), "circular"),  # four y tiles, dilated y kernel
]


@pytest.mark.parametrize("case", PLANE_CASES, ids=[f"{c[4]}k{c[5]}{c[9]}" for c in PLANE_CASES])
def test_planes3d_matches_separable_passes_and_torch(case, monkeypatch):
    """planes_fwd / colz / planes_inv (csrc/planes3d.hpp) against the five separable passes (FFTCONV_PLANES=0) and
    torch's direct convolution in float64: same function, three launches, plane-major intermediates."""
    from fft_conv_pytorch_amd import _native
    from fft_conv_pytorch_amd.functional import fft_conv
    B, Ci, Co, g, size, k, s, p, d, mode = case
    gen = torch.Generator().manual_seed(1000 + sum(case[4]) + case[0])
    x = torch.randn(B, Ci, *size, generator=gen).to(DEV)
    w = (torch.randn(Co, Ci // g, *k, generator=gen) / math.sqrt(Ci // g * math.prod(k))).to(DEV)
    b = torch.randn(Co, generator=gen).to(DEV)
    outs = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("FFTCONV_PLANES", knob)          # (read at plan creation)
        _native.clear_plan_cache()
        outs[knob] = fft_conv(x, w, b, stride=s, padding=p, dilation=d, groups=g, padding_mode=mode)
    xd, wd, bd = x.double().cpu(), w.double().cpu(), b.double().cpu()
    if mode == "constant":
        want = F.conv3d(xd, wd, bd, stride=s, padding=p, dilation=d, groups=g)
    else:
        pp = (p,) * 3 if isinstance(p, int) else p
        want = F.conv3d(F.pad(xd, [q for ax in reversed(pp) for q in (ax, ax)], mode=mode), wd, bd, stride=s, dilation=d, groups=g)
    assert outs["1"].shape == want.shape and outs["1"].is_contiguous()
    assert _rel(outs["1"], want) < REL_TOL and _rel(outs["0"], want) < REL_TOL
    assert _rel(outs["1"], outs["0"]) < 5e-6                 # the two pipelines agree to fp32 rounding
    monkeypatch.delenv("FFTCONV_PLANES", raising=False)
    _native.clear_plan_cache()


def test_planes3d_transposed_and_backward():
    """The transposed op and the gradients ride the same plans (dX of a 3-D convolution is a transposed plan whose
    padded extent still fits the 64-point planes)."""
    from fft_conv_pytorch_amd.functional import fft_conv, fft_conv_transpose
    gen = torch.Generator().manual_seed(33)
    x = torch.randn(2, 8, 20, 20, 20, generator=gen)
    w = torch.randn(8, 6, 3, 3, 3, generator=gen) / 10
    b = torch.randn(6, generator=gen)
    want = F.conv_transpose3d(x.double(), w.double(), b.double(), stride=2, padding=1, output_padding=1)
    got = fft_conv_transpose(x.to(DEV), w.to(DEV), b.to(DEV), stride=2, padding=1, output_padding=1)
    assert _rel(got, want) < REL_TOL
    xs = torch.randn(2, 8, 40, 48, 56, generator=gen)
    ws = torch.randn(8, 8, 3, 5, 4, generator=gen) / 20
    bs = torch.randn(8, generator=gen)
    xr, wr, br = (t.clone().double().requires_grad_() for t in (xs, ws, bs))
    F.conv3d(xr, wr, br, padding=2).square().sum().backward()
    xg, wg, bg = (t.to(DEV).requires_grad_() for t in (xs, ws, bs))
    fft_conv(xg, wg, bg, padding=2).square().sum().backward()
    for got_g, want_g in ((xg.grad, xr.grad), (wg.grad, wr.grad), (bg.grad, br.grad)):
        assert _rel(got_g, want_g) < REL_TOL


# ----------------------------------------------------------------------------- backward: db in the dW launch, padding adjoints
@pytest.mark.parametrize("mode,pad", [("constant", 5), ("reflect", 7), ("replicate", 4), ("circular", 9)])
def test_bias_gradient_rides_the_weight_gradient_launch(mode, pad):
    """1-D, stride 1: fc_wgrad1d_db returns rows [dW | db] per slice (db = bin 0 of the gradient spectra) and ONE
    reduction sums both; the input gradient of a non-zero padding mode folds its border samples back explicitly.
    Against torch's autograd in float64, long rows (many tiles and slices), a kernel that runs in tap segments too."""
    from fft_conv_pytorch_amd import FFTConv1d, _native
    for cin, cout, g, L, k, dil in ((8, 8, 1, 20000, 129, 1), (16, 24, 2, 9000, 33, 3), (8, 8, 1, 6000, 900, 1)):
        torch.manual_seed(L + k)
        layer = FFTConv1d(cin, cout, k, padding=pad, dilation=dil, groups=g, padding_mode="zeros" if mode == "constant" else mode).to(DEV)
        x = torch.randn(3, cin, L, device=DEV, requires_grad=True)
        desc = _native.conv_desc(1, 3, cin, cout, g, (L,), (k,), (1,), (pad,), (dil,), _native.PAD_MODES[mode])
        assert _native.wgrad1d_db_supported(desc)             # (dense kernel: the fold is on)
        y = layer(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        xr = x.detach().double().cpu().requires_grad_()
        wr = layer.weight.detach().double().cpu().requires_grad_()
        br = layer.bias.detach().double().cpu().requires_grad_()
        xp = F.pad(xr, (pad, pad), mode=mode) if mode != "constant" else F.pad(xr, (pad, pad))
        F.conv1d(xp, wr, br, dilation=dil, groups=g).backward(gy.double().cpu())
        assert _rel(layer.bias.grad, br.grad) < REL_TOL
        assert _rel(layer.weight.grad, wr.grad) < REL_TOL
        assert _rel(x.grad, xr.grad) < REL_TOL


def test_padding_adjoint_helper_matches_autograd():
    """autograd._pad_adjoint (slice-adds on the border planes) against torch's own backward of F.pad, 1-3 axes."""
    from fft_conv_pytorch_amd.autograd import _pad_adjoint
    gen = torch.Generator().manual_seed(5)
    for mode in ("reflect", "replicate", "circular"):
        for size, pad in (((19,), (4,)), ((9, 12), (3, 0)), ((7, 8, 6), (2, 3, 1))):
            probe = torch.randn(2, 3, *size, generator=gen, dtype=torch.float64, requires_grad=True)
            flat = [q for p in reversed(pad) for q in (p, p)]
            padded = F.pad(probe, flat, mode=mode)
            g = torch.randn(padded.shape, generator=gen, dtype=torch.float64)
            want, = torch.autograd.grad(padded, probe, g)
            got = _pad_adjoint(g.to(DEV), size, pad, mode)
            assert torch.allclose(got.cpu(), want, rtol=0, atol=1e-12), (mode, size, pad)


# ----------------------------------------------------------------------------- plan creation and stream capture
def test_warm_plans_capture_and_cold_plans_say_why_not():
    """Plan creation allocates device tables with synchronous calls, so it must happen before a capture; a shape that
    has run once is captured and replayed like any kernel launch (bench.py does exactly that).  A COLD shape inside a
    capture fails with an error that names the fix instead of a bare HIP code (checked in a child process: a failed
    capture leaves torch's capture stream unusable for the rest of the process)."""
    import os
    import subprocess
    import sys
    from fft_conv_pytorch_amd import FFTConv1d
    torch.manual_seed(0)
    layer = FFTConv1d(8, 8, 65, bias=True).to(DEV).eval()
    x = torch.randn(4, 8, 9000, device=DEV)
    with torch.no_grad():
        want = layer(x).clone()                                # warm: plan, twiddles, work list, spectrum
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = layer(x)
        y.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, want)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "from fft_conv_pytorch_amd import FFTConv2d\n"
        "cold = FFTConv2d(8, 8, 5, bias=False).to('cuda:0').eval()\n"
        "x = torch.randn(2, 8, 300, 700, device='cuda:0')\n"
        "g = torch.cuda.CUDAGraph()\n"
        "try:\n"
        "    with torch.no_grad(), torch.cuda.graph(g):\n"
        "        cold(x)\n"
        "    print('CAPTURED')\n"
        "except Exception as exc:\n"
        "    while exc is not None:\n"
        "        print('MSG', str(exc).replace(chr(10), ' ')[:400])\n"
        "        exc = exc.__context__\n"
    )
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    out = res.stdout
    assert "CAPTURED" in out or "before the capture" in out, (out[-2000:], res.stderr[-2000:])


# ----------------------------------------------------------------------------- many-channel plan: 32-bit guards
def test_dense_plan_is_refused_beyond_its_32_bit_offsets():
    """dense_inv marks dead stores with bit 31 of an offset into ONE group's output rows: Cog * Lout * 4 bytes must stay
    below 2 GiB.  A shape beyond it keeps the fused kernels (plan layout word 6 != 2); just below it the pipeline runs.
    Plan creation only: no tensor of that size is needed."""
    from fft_conv_pytorch_amd import _native

    def layout_word6(cout, length):
        key = (1, 1, 32, cout, 1, (length,), (65,), (1,), (0,), (1,), 0, False, 0, False, (0,), 0)
        with torch.cuda.device(0):
            plan = _native.Plan(key, 0)
        return plan.layout[6]

    assert layout_word6(256, 1 << 20) == 2                   # 256 x 2^20 x 4 = 1 GiB per group: many-channel pipeline
    assert layout_word6(256, (1 << 21) + 64) != 2            # 2 GiB and a bit: refused, fused kernels instead


# ----------------------------------------------------------------------------- planner: small batches (bench cfgA_shard)
def test_small_batch_plans_spread_over_more_workgroups():
    """B = 4 rows of 32768 (one GPU's share of cfgA on an 8-GPU node): 63 four-slot items would leave three quarters of
    the CUs idle; the launch-time model picks the two-slot kernel (126 workgroups).  The result is the same function."""
    from fft_conv_pytorch_amd import FFTConv1d
    torch.manual_seed(1)
    layer = FFTConv1d(8, 8, 512, bias=True).to(DEV).eval()
    x = torch.randn(4, 8, 32768, device=DEV)
    with torch.no_grad():
        y = layer(x)
    plan = layer.__dict__["_spectrum_cache"][1].plan
    assert plan.tile == 1024 and plan.layout[7] == 2, plan.layout
    want = F.conv1d(x.double(), layer.weight.double(), layer.bias.double())
    assert _rel(y, want) < REL_TOL


# ----------------------------------------------------------------------------- float64 through an FFT path (row N4)
F64_CASES = [
    # B, Cin, Cout, groups, L, k, stride, padding, dilation, mode
    (1, 8, 8, 1, 32768, 128, 1, 0, 1, "constant"),        # cfg0 in float64
    (2, 6, 10, 2, 5000, 33, 2, 7, 3, "reflect"),          # groups, stride, dilation, 97-sample extent
    (3, 5, 3, 1, 1234, 16, 1, 15, 1, "circular"),         # shortest kernel of the path, ragged channels
    (2, 12, 12, 1, 9000, 513, 1, 256, 2, "replicate"),    # 1025-sample extent: the 2048-point tile
    (1, 4, 4, 4, 700, 100, 3, 0, 1, "constant"),          # depthwise, stride 3
]


@pytest.mark.parametrize("case", F64_CASES, ids=[f"L{c[4]}k{c[5]}{c[9]}" for c in F64_CASES])
def test_float64_fft_path_matches_torch_and_the_direct_kernel(case, monkeypatch):
    """float64 tensors of a 1-D convolution run overlap-save tiles through a double-precision transform
    (csrc/fft_f64.hip) like the reference's complex128 FFTs (functional.py:19-89), not the O(taps) direct kernel:
    within 1e-10 of torch's float64 convolution, and of the direct kernel (FFTCONV_F64_FFT=0)."""
    from fft_conv_pytorch_amd import _native
    from fft_conv_pytorch_amd.functional import _plan_for, fft_conv
    B, Ci, Co, g, L, k, s, p, d, mode = case
    gen = torch.Generator().manual_seed(L + k)
    x = torch.randn(B, Ci, L, generator=gen, dtype=torch.float64).to(DEV)
    w = torch.randn(Co, Ci // g, k, generator=gen, dtype=torch.float64).to(DEV)
    b = torch.randn(Co, generator=gen, dtype=torch.float64).to(DEV)
    outs = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("FFTCONV_F64_FFT", knob)          # (read at plan creation)
        _native.clear_plan_cache()
        outs[knob] = fft_conv(x, w, b, stride=s, padding=p, dilation=d, groups=g, padding_mode=mode)
        plan = _plan_for(x, w, b, s, p, d, g, mode)
        assert (plan.tile > 0) == (knob == "1")              # an FFT tile, or the direct kernel
    xc = x.cpu()
    if mode != "constant" and p:
        want = F.conv1d(F.pad(xc, (p, p), mode=mode), w.cpu(), b.cpu(), stride=s, dilation=d, groups=g)
    else:
        want = F.conv1d(xc, w.cpu(), b.cpu(), stride=s, padding=p, dilation=d, groups=g)
    assert outs["1"].dtype == torch.float64 and outs["1"].shape == want.shape
    assert _rel(outs["1"], want) < 1e-10 and _rel(outs["0"], want) < 1e-10
    monkeypatch.delenv("FFTCONV_F64_FFT", raising=False)
    _native.clear_plan_cache()


def test_float64_module_trains_through_the_fft_path():
    from fft_conv_pytorch_amd import FFTConv1d
    torch.manual_seed(3)
    layer = FFTConv1d(6, 6, 65, padding=32, bias=True).to(DEV).double()
    x = torch.randn(2, 6, 3000, device=DEV, dtype=torch.float64, requires_grad=True)
    layer(x).square().sum().backward()
    xr = x.detach().cpu().requires_grad_()
    wr, br = layer.weight.detach().cpu().requires_grad_(), layer.bias.detach().cpu().requires_grad_()
    F.conv1d(xr, wr, br, padding=32).square().sum().backward()
    assert _rel(x.grad, xr.grad) < 1e-10 and _rel(layer.weight.grad, wr.grad) < 1e-10 and _rel(layer.bias.grad, br.grad) < 1e-10


# ----------------------------------------------------------------------------- native weight gradients beyond 1-D / stride 1
def _torch_dw(x, gy, wshape, stride, padding, dilation, groups, mode, want_db=False):
    """dW (and db) of the reference convolution by torch's autograd in float64 on the CPU."""
    n = x.ndim - 2
    xr = x.detach().double().cpu()
    wr = torch.zeros(wshape, dtype=torch.float64, requires_grad=True)
    br = torch.zeros(wshape[0], dtype=torch.float64, requires_grad=True)
    flat = [q for p in reversed(padding) for q in (p, p)]
    xp = F.pad(xr, flat, mode=mode) if mode != "constant" else F.pad(xr, flat)
    conv = (F.conv1d, F.conv2d, F.conv3d)[n - 1]
    conv(xp, wr, br, stride=stride, dilation=dilation, groups=groups).backward(gy.detach().double().cpu())
    return (wr.grad, br.grad) if want_db else wr.grad


WGRAD_ND_CASES = [
    # B, Cin, Cout, groups, size, k, stride, padding, dilation, mode
    (4, 8, 8, 1, (96, 100), (15, 15), (1, 1), (0, 0), (1, 1), "constant"),
    (3, 6, 9, 3, (41, 37), (3, 5), (2, 1), (1, 2), (1, 1), "constant"),           # groups, stride, ragged channels
    (2, 4, 4, 1, (50, 64), (4, 3), (3, 2), (2, 2), (2, 3), "reflect"),            # stride with unreached tail samples, dilation
    (12, 3, 5, 1, (33, 40), (5, 5), (1, 2), (2, 0), (1, 2), "circular"),          # batch > 8: the contraction runs in chunks
    (2, 2, 3, 1, (20, 24, 28), (3, 2, 4), (1, 1, 1), (1, 0, 1), (1, 1, 1), "constant"),
    (3, 4, 6, 2, (17, 30, 21), (2, 3, 3), (2, 1, 2), (0, 1, 1), (1, 2, 1), "replicate"),
    (2, 8, 8, 1, (40, 40, 40), (5, 5, 5), (1, 1, 1), (0, 0, 0), (1, 1, 1), "constant"),   # a shape the forward runs plane-major
]


@pytest.mark.parametrize("case", WGRAD_ND_CASES, ids=[f"{len(c[4])}d-k{'x'.join(map(str, c[5]))}-s{'x'.join(map(str, c[6]))}-{c[9]}" for c in WGRAD_ND_CASES])
def test_wgrad_nd_runs_in_the_library_and_matches_torch(case):
    """2-D / 3-D dW through fc_wgrad_nd (the tensors as they lie: no transposed copies, no crop) against torch's autograd in
    float64, and against the round-2 route (forward plans fed by torch transposes) it replaces."""
    from fft_conv_pytorch_amd import autograd as A
    b, cin, cout, g, size, k, stride, padding, dilation, mode = case
    torch.manual_seed(sum(size) + sum(k))
    x = torch.randn(b, cin, *size, device=DEV)
    lout = tuple((s + 2 * p - d * (kk - 1) - 1) // st + 1 for s, p, d, kk, st in zip(size, padding, dilation, k, stride))
    gy = torch.randn(b, cout, *lout, device=DEV)
    wshape = (cout, cin // g) + tuple(k)
    got = A._grad_weight_nd_native(x, gy, wshape, stride, padding, dilation, g, mode)
    assert got is not None and tuple(got.shape) == wshape
    want = _torch_dw(x, gy, wshape, stride, padding, dilation, g, mode)
    assert _rel(got, want) < REL_TOL
    old = A._grad_weight_plans(x, gy, wshape, stride, padding, dilation, g, mode)
    assert _rel(got, old) < REL_TOL
    # the operands were read in place: still intact
    torch.manual_seed(sum(size) + sum(k))
    assert torch.equal(x, torch.randn(b, cin, *size, device=DEV))


def test_wgrad_nd_is_what_the_module_backward_calls(monkeypatch):
    """FFTConv2d.backward takes dW from fc_wgrad_nd: the torch-transpose route is not entered."""
    from fft_conv_pytorch_amd import FFTConv2d, autograd as A

    def boom(*a, **k):
        raise AssertionError("N-d dW went through the forward-plan route")
    monkeypatch.setattr(A, "_grad_weight_plans", boom)
    torch.manual_seed(3)
    layer = FFTConv2d(4, 6, (5, 3), stride=(2, 1), padding=(1, 2), groups=2).to(DEV)
    x = torch.randn(3, 4, 30, 44, device=DEV, requires_grad=True)
    y = layer(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    want_w, want_b = _torch_dw(x, gy, tuple(layer.weight.shape), (2, 1), (1, 2), (1, 1), 2, "constant", want_db=True)
    assert _rel(layer.weight.grad, want_w) < REL_TOL
    assert _rel(layer.bias.grad, want_b) < REL_TOL


@pytest.mark.parametrize("stride,dil,mode,pad", [(2, 1, "constant", 3), (3, 2, "reflect", 5), (5, 1, "circular", 4), (2, 3, "replicate", 2)])
def test_strided_1d_weight_gradient_runs_in_fc_wgrad1d(stride, dil, mode, pad):
    """fc_wgrad1d covers strided convolutions (round 3: the gradient row is read spread over the stride's grid); dW and the
    folded db against torch's autograd in float64 -- long rows (several tiles and slices), tails the stride never reaches."""
    from fft_conv_pytorch_amd import autograd as A, _native
    for cin, cout, g, L, k in ((8, 8, 1, 20001, 65), (6, 9, 3, 7003, 17), (4, 4, 1, 5000, 700)):
        torch.manual_seed(L + k + stride)
        x = torch.randn(3, cin, L, device=DEV)
        lout = (L + 2 * pad - dil * (k - 1) - 1) // stride + 1
        gy = torch.randn(3, cout, lout, device=DEV)
        desc = _native.conv_desc(1, 3, cin, cout, g, (L,), (k,), (stride,), (pad,), (dil,), _native.PAD_MODES[mode])
        assert _native.wgrad1d_slices(desc) > 0
        got = A._grad_weight_native(x, gy, (cout, cin // g, k), (stride,), (pad,), (dil,), g, mode, want_db=True)
        assert got is not None
        want_w, want_b = _torch_dw(x, gy, (cout, cin // g, k), (stride,), (pad,), (dil,), g, mode, want_db=True)
        assert _rel(got[0], want_w) < REL_TOL, (cin, L, k)
        assert got[1] is not None and _rel(got[1], want_b) < REL_TOL


# ----------------------------------------------------------------------------- 2-D: thread-per-sequence column pass between row passes
ROWS2D_CASES = [
    # B, Cin, Cout, groups, size, k, stride, padding, dilation, mode
    (4, 8, 8, 1, (512, 512), (31, 31), 1, 0, 1, "constant"),              # cfgB at batch 4: 15 overlap-save tiles along y
    (1, 8, 8, 1, (100, 96), (3, 3), 1, 1, 1, "constant"),                 # one batch item (NB = 1 build), 3 x 3
    (3, 5, 7, 1, (70, 130), (4, 5), 1, (2, 0), 1, "constant"),            # odd batch, ragged channel counts
    (2, 8, 24, 1, (90, 61), (5, 3), (2, 3), (2, 1), 1, "constant"),       # strides, three output chunks
    (2, 16, 8, 2, (64, 200), (3, 7), 1, (1, 3), 1, "reflect"),            # groups, index-map padding
    (2, 8, 8, 1, (131, 75), (9, 5), 1, 4, (4, 2), "circular"),            # dilated y kernel of 33: the longest of the path
    (5, 16, 8, 2, (33, 40), (2, 3), (1, 2), 1, 1, "replicate"),           # two groups of 8 -> 4 channels
    (2, 8, 8, 1, (300, 20), (17, 3), (3, 1), 0, 1, "constant"),           # narrow rows (Tx = 32: 16 bin columns), y stride
    (2, 8, 8, 1, (256, 512), (7, 7), 1, 3, 1, "constant"),                # 'same' padding on a power of two: rows in nine 64-point x tiles
    (3, 8, 16, 1, (40, 300), (5, 9), (1, 2), (2, 4), 1, "constant"),      # x tiles, two output chunks, x stride
]


@pytest.mark.parametrize("case", ROWS2D_CASES, ids=[f"{c[4]}k{c[5]}{c[9]}" for c in ROWS2D_CASES])
def test_rows2d_pipeline_matches_separable_passes_and_torch(case, monkeypatch):
    """2-D with a y kernel of at most 33 dilated taps: rows_r2c (rows as they are) / colz (one thread per 64-point y
    sequence, lanes over bin columns) / rows_c2r against the transposing passes with the LDS column pass
    (FFTCONV_PLANES=0) and torch's direct convolution in float64."""
    from fft_conv_pytorch_amd import _native
    from fft_conv_pytorch_amd.functional import fft_conv, _plan_for
    B, Ci, Co, g, size, k, s, p, d, mode = case
    gen = torch.Generator().manual_seed(2000 + sum(size) + B)
    x = torch.randn(B, Ci, *size, generator=gen).to(DEV)
    w = (torch.randn(Co, Ci // g, *k, generator=gen) / math.sqrt(Ci // g * math.prod(k))).to(DEV)
    b = torch.randn(Co, generator=gen).to(DEV)
    outs, kinds = {}, {}
    for knob in ("2", "0"):                                  # 2: wherever possible (the planner takes it on large problems only)
        monkeypatch.setenv("FFTCONV_PLANES", knob)          # (read at plan creation)
        _native.clear_plan_cache()
        outs[knob] = fft_conv(x, w, b, stride=s, padding=p, dilation=d, groups=g, padding_mode=mode)
        tup = lambda v: (v, v) if isinstance(v, int) else tuple(v)
        kinds[knob] = _plan_for(x, w, b, tup(s), tup(p), tup(d), g, mode).layout[7]
    assert kinds == {"2": 2, "0": 0}                         # the new pipeline really ran (and the knob really turns it off)
    xd, wd, bd = x.double().cpu(), w.double().cpu(), b.double().cpu()
    if mode == "constant":
        want = F.conv2d(xd, wd, bd, stride=s, padding=p, dilation=d, groups=g)
    else:
        pp = (p,) * 2 if isinstance(p, int) else p
        want = F.conv2d(F.pad(xd, [q for ax in reversed(pp) for q in (ax, ax)], mode=mode), wd, bd, stride=s, dilation=d, groups=g)
    assert outs["2"].shape == want.shape and outs["2"].is_contiguous()
    assert _rel(outs["2"], want) < REL_TOL and _rel(outs["0"], want) < REL_TOL
    assert _rel(outs["2"], outs["0"]) < 5e-6                 # the two pipelines agree to fp32 rounding
    monkeypatch.delenv("FFTCONV_PLANES", raising=False)
    _native.clear_plan_cache()


def test_rows2d_pipeline_is_the_planners_choice_on_large_images_only():
    from fft_conv_pytorch_amd.functional import _plan_for
    from fft_conv_pytorch_amd import _native
    _native.clear_plan_cache()
    pick = lambda b, s, k: _plan_for(torch.empty(b, 8, s, s, device=DEV), torch.empty(8, 8, k, k, device=DEV), None, (1, 1), (0, 0),
                                     (1, 1), 1, "constant").layout[7]
    assert pick(16, 512, 7) == 2 and pick(2, 1024, 5) == 2            # large: thread-per-sequence column pass
    assert pick(4, 256, 7) == 0 and pick(16, 512, 31) == 0            # small problem / long y kernel: LDS column pass


# ----------------------------------------------------------------------------- zero padding absorbs the cyclic wrap: shorter transforms
ZEROWRAP_CASES = [
    # nd, B, C, size, k, stride, padding, transposed
    (2, 2, 8, (98, 114), (31, 15), (1, 1), (0, 0), True),        # dX of an unpadded convolution: out = 128 x 128 exactly
    (3, 2, 8, (56, 56, 56), (9, 9, 9), (1, 1, 1), (0, 0, 0), True),   # cfgC's dX: 64^3, the plane-major pipeline becomes eligible
    (2, 2, 4, (30, 41), (5, 4), (2, 3), (1, 0), True),           # strides spread the source, padding crops: one y tile of 64 instead of two
    (2, 3, 8, (250, 120), (11, 9), (1, 1), (4, 5), False),       # padded forward convolution: size + pad fits where size + 2 pad does not
    (3, 1, 3, (20, 30, 61), (3, 5, 4), (1, 2, 1), (1, 1, 3), False),
]


@pytest.mark.parametrize("case", ZEROWRAP_CASES, ids=[f"{c[0]}d-{'x'.join(map(str, c[3]))}-k{'x'.join(map(str, c[4]))}-{'T' if c[7] else 'F'}" for c in ZEROWRAP_CASES])
def test_zero_padding_absorbs_the_wrap_of_a_shorter_transform(case, monkeypatch):
    """With zero padding a cyclic transform shorter than the padded axis is exact when all data and the wrapped-in range lie
    inside the tile (fc_api.cpp set_need): same results as with FFTCONV_ZEROWRAP=0 and as torch float64, shorter plans."""
    from fft_conv_pytorch_amd import _native
    from fft_conv_pytorch_amd.functional import fft_conv, fft_conv_transpose, _plan_for
    nd, B, C, size, k, stride, padding, transposed = case
    gen = torch.Generator().manual_seed(3000 + sum(size))
    x = torch.randn(B, C, *size, generator=gen).to(DEV)
    w = (torch.randn(C, C, *k, generator=gen) / math.sqrt(C * math.prod(k))).to(DEV)
    b = torch.randn(C, generator=gen).to(DEV)
    outs, tiles = {}, {}
    monkeypatch.setenv("FFTCONV_XTILE", "0")                  # (single row / middle transforms: the comparison below is about their
    monkeypatch.setenv("FFTCONV_YTILE", "0")                  #  length, which the planner's overlap-save tiles would hide)
    for knob in ("1", "0"):
        monkeypatch.setenv("FFTCONV_ZEROWRAP", knob)          # (read at plan creation)
        _native.clear_plan_cache()
        if transposed:
            outs[knob] = fft_conv_transpose(x, w, b, stride=stride, padding=padding)
            plan = _plan_for(x, w, b, stride, padding, (1,) * nd, 1, "constant", transposed=True, output_padding=(0,) * nd)
        else:
            outs[knob] = fft_conv(x, w, b, stride=stride, padding=padding)
            plan = _plan_for(x, w, b, stride, padding, (1,) * nd, 1, "constant")
        tiles[knob] = plan.layout[:3]
    for name in ("FFTCONV_ZEROWRAP", "FFTCONV_XTILE", "FFTCONV_YTILE"):
        monkeypatch.delenv(name, raising=False)
    _native.clear_plan_cache()
    xd, wd, bd = x.double().cpu(), w.double().cpu(), b.double().cpu()
    if transposed:
        want = (F.conv_transpose2d if nd == 2 else F.conv_transpose3d)(xd, wd, bd, stride=stride, padding=padding)
    else:
        want = (F.conv2d if nd == 2 else F.conv3d)(xd, wd, bd, stride=stride, padding=padding)
    assert outs["1"].shape == want.shape
    assert _rel(outs["1"], want) < REL_TOL and _rel(outs["0"], want) < REL_TOL
    assert _rel(outs["1"], outs["0"]) < 5e-6
    # (outermost tile, row transform, middle transform): the row / middle transforms never grow, and something got shorter
    # (the outermost axis may trade several short overlap-save tiles for one longer cyclic tile)
    assert tiles["1"][1] <= tiles["0"][1] and tiles["1"][2] <= tiles["0"][2], tiles
    if tuple(size) != (30, 41):          # (there the saving is the tile COUNT of the outermost axis, which the layout words do not show)
        assert tiles["1"] != tiles["0"], tiles


AUTOTILE_CASES = [
    # nd, B, Cin, Cout, groups, size, k, stride, padding, mode
    (2, 2, 8, 8, 1, (90, 300), (5, 5), (1, 1), (2, 2), "constant"),          # 304 samples -> five 64-point x tiles instead of 512 points
    (2, 3, 4, 6, 2, (150, 150), (3, 7), (1, 2), (1, 3), "reflect"),          # index-map padding, x stride
    (3, 1, 4, 4, 1, (20, 150, 170), (3, 3, 5), (1, 1, 1), (1, 1, 2), "constant"),   # 3-D: middle axis AND rows in tiles
    (3, 2, 8, 8, 1, (30, 140, 40), (2, 9, 3), (1, 2, 1), (0, 4, 1), "circular"),    # middle axis only, stride on it
]


@pytest.mark.parametrize("case", AUTOTILE_CASES, ids=[f"{c[0]}d-{'x'.join(map(str, c[5]))}-{c[9]}" for c in AUTOTILE_CASES])
def test_rows_just_past_a_power_of_two_run_in_tiles_and_match_torch(case, monkeypatch):
    """The planner cuts rows / the middle axis into overlap-save tiles where that saves >= 15 % of the points (fc_api.cpp
    plan_nd): forward, dX, dW and db against torch float64, and the forward against the single-transform plan (knobs = 0)."""
    from fft_conv_pytorch_amd import _native
    from fft_conv_pytorch_amd.functional import fft_conv, _plan_for
    nd, B, Ci, Co, g, size, k, stride, padding, mode = case
    gen = torch.Generator().manual_seed(4000 + sum(size))
    x = torch.randn(B, Ci, *size, generator=gen).to(DEV).requires_grad_()
    w = (torch.randn(Co, Ci // g, *k, generator=gen) / math.sqrt(Ci // g * math.prod(k))).to(DEV).requires_grad_()
    b = torch.randn(Co, generator=gen).to(DEV).requires_grad_()
    _native.clear_plan_cache()
    y = fft_conv(x, w, b, stride=stride, padding=padding, groups=g, padding_mode=mode)
    tiled = _plan_for(x, w, b, stride, padding, (1,) * nd, g, mode).layout[:3]
    gy = torch.randn(y.shape, generator=gen).to(DEV)
    y.backward(gy)
    monkeypatch.setenv("FFTCONV_XTILE", "0")
    monkeypatch.setenv("FFTCONV_YTILE", "0")
    _native.clear_plan_cache()
    with torch.no_grad():
        y_single = fft_conv(x, w, b, stride=stride, padding=padding, groups=g, padding_mode=mode)
    single = _plan_for(x, w, b, stride, padding, (1,) * nd, g, mode).layout[:3]
    monkeypatch.delenv("FFTCONV_XTILE")
    monkeypatch.delenv("FFTCONV_YTILE")
    _native.clear_plan_cache()
    assert tiled != single and tiled[1] <= single[1] and tiled[2] <= single[2], (tiled, single)     # (tiles really were taken)
    xr, wr, br = (t.detach().double().cpu().requires_grad_() for t in (x, w, b))
    conv = F.conv2d if nd == 2 else F.conv3d
    if mode == "constant":
        want = conv(xr, wr, br, stride=stride, padding=padding, groups=g)
    else:
        want = conv(F.pad(xr, [q for p_ in reversed(padding) for q in (p_, p_)], mode=mode), wr, br, stride=stride, groups=g)
    want.backward(gy.double().cpu())
    assert _rel(y, want) < REL_TOL and _rel(y_single, want) < REL_TOL and _rel(y, y_single) < 5e-6
    assert _rel(x.grad, xr.grad) < REL_TOL and _rel(w.grad, wr.grad) < REL_TOL and _rel(b.grad, br.grad) < REL_TOL


@pytest.mark.parametrize("nd", [1, 2, 3])
def test_a_whole_training_step_replays_from_a_hip_graph(nd):
    """Forward + backward (dX, dW, db) of a module captured into ONE HIP graph after a warm-up step (plans, tables and the
    allocator's blocks exist by then) and replayed on new data: the gradients of the replay equal an eager step's."""
    from fft_conv_pytorch_amd import FFTConv1d, FFTConv2d, FFTConv3d
    cls, size, k = {1: (FFTConv1d, (5000,), 129), 2: (FFTConv2d, (70, 150), (5, 7)), 3: (FFTConv3d, (20, 30, 70), 3)}[nd]
    torch.manual_seed(10 + nd)
    layer = cls(8, 8, k, padding=1, stride=1).to(DEV)
    x_static = torch.randn(2, 8, *size, device=DEV, requires_grad=True)

    def step():
        layer.zero_grad(set_to_none=False)
        if x_static.grad is not None:
            x_static.grad.zero_()
        loss = (layer(x_static) ** 2).sum()
        loss.backward()

    # warm-up on a side stream (torch's capture rule), grads allocated once so the graph writes into fixed buffers
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream(DEV).wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    new = torch.randn_like(x_static)
    with torch.no_grad():
        x_static.copy_(new)
    graph.replay()
    torch.cuda.synchronize()
    got = [layer.weight.grad.clone(), layer.bias.grad.clone(), x_static.grad.clone()]
    step()                                              # the same data, eagerly
    torch.cuda.synchronize()
    want = [layer.weight.grad, layer.bias.grad, x_static.grad]
    for g_, w_ in zip(got, want):
        assert _rel(g_, w_) < 1e-6
