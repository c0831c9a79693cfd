"""GPU tests added in round 2: backward of the transposed convolution (functional + modules), gradients pinned
to the reference's own autograd (golden family G5), the phantom-channel fault guard, the multi-GPU helpers at
world size 1 over RCCL, and the host-layer guarantees (device handling, spectrum cache, shared spectra).
Every call goes through the C ABI (fft_conv_pytorch_amd._native)."""
import itertools
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import golden_util as gu

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4      # north_star bound (fp32, relative to the tensor's max magnitude)
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _gcd3(a, b, c):
    return math.gcd(a, math.gcd(b, c))


# ----------------------------------------------------------------------------- transposed backward (row N2)
@pytest.mark.parametrize("ndim", [1, 2, 3])
def test_transpose_backward_reference_grid(ndim):
    """w.grad and b.grad as the reference pins them (tests/test_functional_transpose.py:73-124: its grid with
    the `dilation += output_padding; stride += output_padding` rule, y.sum().backward()), plus x.grad, against
    torch's conv_transpose autograd on the CPU with the reference's absolute tolerance
    (benchmark_utils.py:53-57).  2-D / 3-D are thinned; every axis value still appears."""
    from fft_conv_pytorch_amd.functional import fft_conv_transpose
    gen = torch.Generator().manual_seed(977 + ndim)
    conv = getattr(F, f"conv_transpose{ndim}d")
    grid = itertools.product([2, 3], [2, 3], [1, 2, 3], [2, 3], [0, 1], [1, 2], [1, 2], [0, 1, 2], [7, 8])
    keep_every = {1: 1, 2: 5, 3: 11}[ndim]
    ran = 0
    for idx, (cin, cout, groups, k, pad, stride, dil, opad, size) in enumerate(grid):
        if idx % keep_every:
            continue
        g = _gcd3(cin, cout, groups)
        x = torch.randn(2, cin, *([size] * ndim), generator=gen)
        w = torch.randn(cin, cout // g, *([k] * ndim), generator=gen)
        b = torch.randn(cout, generator=gen)
        kw = dict(stride=stride + opad, padding=pad, output_padding=opad, dilation=dil + opad, groups=g)
        xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
        conv(xr, wr, bias=br, **kw).sum().backward()
        xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, w, b))
        y = fft_conv_transpose(xd, wd, bias=bd, **kw)
        assert y.grad_fn is not None
        y.sum().backward()
        for name, got, want in (("dW", wd.grad, wr.grad), ("db", bd.grad, br.grad), ("dX", xd.grad, xr.grad)):
            err = (got.cpu() - want).abs()
            # the reference's bound is absolute at its toy sizes; gradients of a sum grow with the output count
            scale = max(1.0, want.abs().max().item())
            assert err.max().item() < 1e-4 * scale and err.mean().item() < 5e-5 * scale, (name, idx, kw, err.max().item())
        ran += 1
    assert ran >= 100


def test_transpose_backward_long_rows_float64_truth():
    """Several overlap-save tiles per row, grouped and strided, random output gradient; truth = torch's
    conv_transpose autograd in float64."""
    from fft_conv_pytorch_amd.functional import fft_conv_transpose
    gen = torch.Generator().manual_seed(5)
    cases = [
        dict(B=2, cin=8, cout=8, g=1, L=3000, k=129, stride=1, padding=64, output_padding=0, dilation=1),
        dict(B=2, cin=8, cout=16, g=2, L=1500, k=33, stride=3, padding=5, output_padding=2, dilation=2),
        dict(B=3, cin=16, cout=8, g=1, L=2000, k=65, stride=2, padding=3, output_padding=1, dilation=1),
        dict(B=2, cin=6, cout=6, g=6, L=2500, k=17, stride=2, padding=0, output_padding=3, dilation=4),   # output_padding >= stride
    ]
    for c in cases:
        x = torch.randn(c["B"], c["cin"], c["L"], generator=gen, dtype=torch.float64)
        w = torch.randn(c["cin"], c["cout"] // c["g"], c["k"], generator=gen, dtype=torch.float64) / c["k"] ** 0.5
        b = torch.randn(c["cout"], generator=gen, dtype=torch.float64)
        kw = dict(stride=c["stride"], padding=c["padding"], output_padding=c["output_padding"], dilation=c["dilation"],
                  groups=c["g"])
        xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
        want = F.conv_transpose1d(xr, wr, br, **kw)
        gy = torch.randn(want.shape, generator=gen, dtype=torch.float64)
        want.backward(gy)
        xd, wd, bd = (t.float().to(DEV).requires_grad_() for t in (x, w, b))
        got = fft_conv_transpose(xd, wd, bias=bd, **kw)
        got.backward(gy.float().to(DEV))
        for name, a_, b_ in (("y", got, want), ("dX", xd.grad, xr.grad), ("dW", wd.grad, wr.grad), ("db", bd.grad, br.grad)):
            assert _rel(a_, b_) < REL_TOL, (name, c, _rel(a_, b_))


@pytest.mark.parametrize("ndim", [1, 2, 3])
def test_transpose_modules_train_like_torch(ndim):
    """FFTConvTranspose{N}d inside a tiny model: gradients reach the layer's parameters AND the layer in front
    of it (reference: tests/test_module_transpose.py:88-144 pins weight.grad / bias.grad of the module)."""
    import fft_conv_pytorch_amd as fca
    torch.manual_seed(3 + ndim)
    Tr = getattr(fca, f"FFTConvTranspose{ndim}d")
    Fw = getattr(fca, f"FFTConv{ndim}d")
    TorchTr = getattr(torch.nn, f"ConvTranspose{ndim}d")
    TorchFw = getattr(torch.nn, f"Conv{ndim}d")
    size = {1: 300, 2: 24, 3: 10}[ndim]
    ours = torch.nn.Sequential(Fw(3, 4, 3, padding=1), Tr(4, 6, 3, stride=2, padding=1, output_padding=1, groups=2)).to(DEV)
    theirs = torch.nn.Sequential(TorchFw(3, 4, 3, padding=1), TorchTr(4, 6, 3, stride=2, padding=1, output_padding=1, groups=2))
    theirs.load_state_dict({k: v.cpu() for k, v in ours.state_dict().items()})     # state_dicts interchange
    x = torch.randn(2, 3, *([size] * ndim))
    y = ours(x.to(DEV))
    y_ref = theirs(x)
    assert _rel(y, y_ref) < REL_TOL
    gy = torch.randn_like(y_ref)
    y.backward(gy.to(DEV))
    y_ref.backward(gy)
    for (name, p), (_, q) in zip(ours.named_parameters(), theirs.named_parameters()):
        assert p.grad is not None, name
        assert _rel(p.grad, q.grad) < REL_TOL, (name, _rel(p.grad, q.grad))


# ----------------------------------------------------------------------------- G5: gradients pinned to the reference
def test_golden_g5_reference_gradients():
    """dX / dW / db of sum(y * gy) from the REFERENCE's autograd (oracle/make_golden.py: make_g5), forward and
    transposed, including every BASELINE config at reduced batch / length."""
    from fft_conv_pytorch_amd.functional import fft_conv, fft_conv_transpose
    worst = {"y": 0.0, "dx": 0.0, "dw": 0.0, "db": 0.0}
    count = {"fwd": 0, "tr": 0}
    for n, kind, meta, x, w, b, gy, gold in gu.g5_cases():
        xd, wd, bd = (torch.from_numpy(t).to(DEV).requires_grad_() for t in (x, w, b))
        fn = fft_conv if kind == "fwd" else fft_conv_transpose
        y = fn(xd, wd, bias=bd, **gu.g5_kwargs(kind, meta))
        y.backward(torch.from_numpy(gy).to(DEV))
        torch.cuda.synchronize()
        for key, t in (("y", y), ("dx", xd.grad), ("dw", wd.grad), ("db", bd.grad)):
            worst[key] = max(worst[key], gu.check_entry(t.detach().cpu().numpy(), gold[key], REL_TOL))
        count[kind] += 1
    print("G5:", count, "worst rel errs", {k: f"{v:.1e}" for k, v in worst.items()})
    assert count["fwd"] >= 20 and count["tr"] >= 12


# ----------------------------------------------------------------------------- guard for the round-1 memory fault
def _tail_carved(shape, gen, pad_elems=4096):
    """A tensor whose LAST element is the last element of its allocation (NaN canaries in front of it): any read
    past the tensor leaves the allocation, any write before it lands in the canaries."""
    n = int(np.prod(shape))
    buf = torch.full((pad_elems + n,), float("nan"), device=DEV)
    t = buf[pad_elems:].view(*shape)
    t.copy_(torch.randn(shape, generator=gen).to(DEV))
    return buf, t


PHANTOM_CASES = [  # cin, cout, groups, L, k, padding: odd Cin/groups so the last chunk of the last group has phantom channels
    (12, 48, 4, 3000, 33, 16),      # the fuzz case that faulted in round 1 (3 channels per group), general kernel
    (9, 9, 3, 700, 9, 4),           # border-only rows
    (10, 8, 2, 5000, 129, 0),       # 5 channels per group, interior + border tiles
    (8, 8, 1, 5000, 129, 64),       # batch-sharing kernel (8 -> 8), last item at the end of x
    (24, 16, 1, 3000, 200, 7),      # wide-input kernel
    (7, 7, 7, 4000, 65, 32),        # depthwise with a partly empty last 8-channel block
]


@pytest.mark.parametrize("case", PHANTOM_CASES)
def test_no_access_outside_the_tensors(case):
    """Regression guard for the fault fixed in d5d194d (masked lanes dereferenced row 0 of phantom channels past
    the tensor): x, w and the output gradient end exactly at the end of their allocations, y / dX / dW are carved
    the same way by torch; results must be finite and right, canaries in front untouched."""
    from fft_conv_pytorch_amd.functional import fft_conv
    cin, cout, groups, L, k, pad = case
    gen = torch.Generator().manual_seed(cin * 131 + L)
    xbuf, x = _tail_carved((3, cin, L), gen)
    wbuf, w = _tail_carved((cout, cin // groups, k), gen)
    b = torch.randn(cout, generator=gen).to(DEV)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    want = F.conv1d(xr.double(), wr.double(), b.double(), padding=pad, groups=groups)
    xg, wg = x.requires_grad_(), w.requires_grad_()
    got = fft_conv(xg, wg, bias=b, padding=pad, groups=groups)
    assert _rel(got, want) < REL_TOL
    gbuf, gy = _tail_carved(tuple(want.shape), gen)
    want.backward(gy.double())
    got.backward(gy)
    torch.cuda.synchronize()
    assert _rel(xg.grad, xr.grad) < REL_TOL and _rel(wg.grad, wr.grad) < REL_TOL
    for buf in (xbuf, wbuf, gbuf):
        assert torch.isnan(buf[:4096]).all(), "canary in front of an input was overwritten"


# ----------------------------------------------------------------------------- multi-GPU helpers at world size 1 (RCCL)
def test_sharded_forward_and_spectrum_broadcast_world1_nccl():
    """the GPU box has one MI355X, so the multi-GPU path runs at world_size = 1 over the
    real backend (nccl = RCCL): fft_conv_sharded + broadcast_kernel_spectrum against plain fft_conv."""
    import torch.distributed as dist
    from fft_conv_pytorch_amd.distributed import broadcast_kernel_spectrum, fft_conv_sharded, shard_range
    from fft_conv_pytorch_amd import functional as F_
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        torch.manual_seed(9)
        x = torch.randn(5, 8, 6000, device=DEV)
        w = torch.randn(8, 8, 257, device=DEV)
        b = torch.randn(8, device=DEV)
        lo, hi = shard_range(x.shape[0], 1, 0)
        y = fft_conv_sharded(x[lo:hi], w, b, padding=3, padding_mode="reflect")
        y_plain = F_.fft_conv(x, w, b, padding=3, padding_mode="reflect")
        assert torch.equal(y, y_plain)
        plan = F_._plan_for(x, w, b, 1, 3, 1, 1, "reflect")
        spec = broadcast_kernel_spectrum(plan, w)
        assert spec.plan is plan and spec.buf.numel() * 4 >= plan.spectrum_bytes
        assert torch.equal(F_._forward_native(x, spec, b), y_plain)
        # 2-D plan: the broadcast spectrum carries no scratch area of its own
        x2, w2 = torch.randn(2, 4, 40, 50, device=DEV), torch.randn(6, 4, 5, 3, device=DEV)
        y2 = fft_conv_sharded(x2, w2, None, stride=(1, 2))
        assert _rel(y2, F.conv2d(x2, w2, stride=(1, 2))) < REL_TOL
    finally:
        dist.destroy_process_group()


# ----------------------------------------------------------------------------- host-layer guarantees on the device
def test_module_cache_follows_the_weight_on_device():
    import fft_conv_pytorch_amd as fca
    torch.manual_seed(2)
    layer = fca.FFTConv1d(4, 4, 33, padding=16).to(DEV)
    x = torch.randn(2, 4, 1500, device=DEV)
    ref = lambda: F.conv1d(x, layer.weight, layer.bias, padding=16)
    opt = torch.optim.SGD(layer.parameters(), lr=0.5)
    for _ in range(2):                                   # train mode: every step sees the updated weight
        opt.zero_grad()
        y = layer(x)
        assert _rel(y, ref()) < REL_TOL
        y.square().mean().backward()
        opt.step()
    layer.weight.data.mul_(3.0)                          # bypasses the version counter: fine in train mode
    assert _rel(layer(x), ref()) < REL_TOL
    layer.eval()
    with torch.no_grad():
        y0 = layer(x)
        assert "_spectrum_cache" in layer.__dict__ and _rel(y0, ref()) < REL_TOL
        layer.weight.data.mul_(0.5)
        layer.invalidate_kernel_spectrum()               # documented requirement after a .data write in eval mode
        assert _rel(layer(x), ref()) < REL_TOL
    wn = torch.nn.utils.parametrizations.weight_norm(fca.FFTConv1d(4, 4, 9, padding=4).to(DEV))
    for _ in range(2):
        y = wn(x)
        assert _rel(y, F.conv1d(x, wn.weight, wn.bias, padding=4)) < REL_TOL
        y.sum().backward()
        with torch.no_grad():
            wn.parametrizations.weight.original0.mul_(1.5)
    wn.eval()
    with torch.no_grad():
        assert _rel(wn(x), F.conv1d(x, wn.weight, wn.bias, padding=4)) < REL_TOL


def test_one_spectrum_serves_two_streams_nd():
    """The cached kernel spectrum is read-only: two streams run the same 2-D module at once, each call takes its
    own scratch area (round 1 kept the scratch inside the cached spectrum)."""
    import fft_conv_pytorch_amd as fca
    torch.manual_seed(4)
    layer = fca.FFTConv2d(4, 4, 7, padding=3).to(DEV).eval()
    xs = [torch.randn(2, 4, 96, 96, device=DEV) for _ in range(2)]
    with torch.no_grad():
        refs = [F.conv2d(x, layer.weight, layer.bias, padding=3) for x in xs]
        layer(xs[0])                                    # builds the spectrum once
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
        outs = [None, None]
        for rep in range(20):
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    outs[i] = layer(xs[i])
            torch.cuda.synchronize()
            for i in range(2):
                assert _rel(outs[i], refs[i]) < REL_TOL, (rep, i)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_tensor_on_a_non_current_device():
    from fft_conv_pytorch_amd.functional import fft_conv
    torch.cuda.set_device(0)
    dev1 = torch.device("cuda", 1)
    x = torch.randn(2, 8, 5000, device=dev1, requires_grad=True)
    w = torch.randn(8, 8, 129, device=dev1, requires_grad=True)
    y = fft_conv(x, w, padding=64)
    assert y.device == dev1 and _rel(y, F.conv1d(x, w, padding=64)) < REL_TOL
    y.sum().backward()
    assert x.grad.device == dev1 and w.grad.device == dev1
    with pytest.raises(ValueError, match="same device"):
        fft_conv(x, w.to("cuda:0"))


def test_nd_block_of_bin_columns_beyond_2_gib_is_refused(monkeypatch):
    """The row passes address one block of bin columns per workgroup with 32-bit byte offsets: a second-to-last axis so long
    that Tx/2 bin columns of it span 2 GiB is refused at plan creation (a documented limit) instead of
    computing with wrapped offsets.  Just below the limit the same shape family still runs.  (Since round 3 the planner cuts a
    600-sample row into 64-point tiles -- 32 bin columns, far below the limit -- so the refusal is provoked with the single
    1024-point transform of round 2, FFTCONV_XTILE=0.)"""
    from fft_conv_pytorch_amd import _native
    from fft_conv_pytorch_amd.functional import fft_conv
    w = torch.randn(1, 1, 3, 3, device=DEV)
    x = torch.empty(1, 1, 540000, 600, device=DEV)             # padded row 600 -> 1024-point transforms, 512 bin columns
    monkeypatch.setenv("FFTCONV_XTILE", "0")
    _native.clear_plan_cache()
    with pytest.raises(NotImplementedError, match="2 GiB"):
        fft_conv(x, w)
    monkeypatch.delenv("FFTCONV_XTILE")
    _native.clear_plan_cache()
    del x
    x = torch.randn(1, 1, 300000, 40, device=DEV)              # 64-point transforms: 32 columns x 300000 rows = 77 MB
    y = fft_conv(x, w)
    want = F.conv2d(x.double().cpu(), w.double().cpu())
    assert _rel(y, want) < REL_TOL


# ----------------------------------------------------------------------------- rows longer than the largest FFT (N-d)
def test_nd_rows_longer_than_4096():
    """The reference has no size limit (functional.py:66-70).  The last axis of a 2-D / 3-D problem runs in
    overlap-save tiles once its padded extent passes 4096 (round 1 raised NotImplementedError)."""
    from fft_conv_pytorch_amd.functional import fft_conv, fft_conv_transpose
    gen = torch.Generator().manual_seed(77)
    cases = [
        ((2, 3, 64, 8192), (4, 3, 5, 33), dict(padding=(2, 7), padding_mode="reflect")),
        ((1, 2, 8192, 64), (3, 2, 9, 5), dict(stride=(2, 1))),                                  # long OUTER axis (already tiled)
        ((1, 4, 6, 10, 5000), (4, 2, 3, 3, 17), dict(groups=2, padding=(1, 1, 8), dilation=(1, 2, 3))),
        ((2, 2, 40, 4500), (2, 2, 3, 1200), dict(padding=(1, 100), padding_mode="circular")),     # 4096-point x tiles
        ((1, 3, 33, 6000), (5, 3, 4, 9), dict(stride=(1, 3), padding=(0, 4), padding_mode="replicate")),
        ((1, 2, 4, 4300, 12), (2, 2, 2, 9, 3), dict(stride=(1, 2, 1), padding=(0, 4, 1), padding_mode="replicate")),   # long MIDDLE axis
        ((1, 2, 3, 4200, 4200), (2, 1, 2, 5, 7), dict(groups=2)),                                  # middle and last axis tiled
    ]
    for xs, ws, kw in cases:
        x = torch.randn(*xs, generator=gen, dtype=torch.float64)
        w = torch.randn(*ws, generator=gen, dtype=torch.float64) / math.sqrt(np.prod(ws[1:]))
        b = torch.randn(ws[0], generator=gen, dtype=torch.float64)
        nd = len(xs) - 2
        mode = kw.get("padding_mode", "constant")
        pad = kw.get("padding", 0)
        pads = (pad,) * nd if isinstance(pad, int) else tuple(pad)
        kwt = {k: v for k, v in kw.items() if k not in ("padding", "padding_mode")}
        if mode == "constant":
            want = getattr(F, f"conv{nd}d")(x, w, b, padding=pads, **kwt)
        else:
            flat = []
            for p_ in reversed(pads):
                flat += [p_, p_]
            want = getattr(F, f"conv{nd}d")(F.pad(x, flat, mode=mode), w, b, **kwt)
        got = fft_conv(x.float().to(DEV), w.float().to(DEV), b.float().to(DEV), **kw)
        assert got.shape == want.shape
        assert _rel(got, want) < REL_TOL, (xs, ws, kw, _rel(got, want))
    # transposed, long rows
    x = torch.randn(1, 3, 20, 2500, generator=gen, dtype=torch.float64)
    w = torch.randn(3, 2, 3, 9, generator=gen, dtype=torch.float64) / 5
    want = F.conv_transpose2d(x, w, stride=(1, 2), padding=(1, 3), output_padding=(0, 1))
    got = fft_conv_transpose(x.float().to(DEV), w.float().to(DEV), stride=(1, 2), padding=(1, 3), output_padding=(0, 1))
    assert got.shape == want.shape and _rel(got, want) < REL_TOL


@pytest.mark.parametrize("xtile", [64, 128, 256])
def test_nd_forced_x_tiles_match_single_transform(xtile, monkeypatch):
    """The same problems with and without x tiles (FFTCONV_XTILE forces tiles on rows that would fit one FFT;
    FFTCONV_YTILE the same for the middle axis of 3-D plans): 2-D / 3-D, strides, padding modes, backward through the
    tiled plans."""
    from fft_conv_pytorch_amd import _native
    from fft_conv_pytorch_amd.functional import fft_conv
    gen = torch.Generator().manual_seed(78 + xtile)
    cases = [
        ((2, 4, 30, 300), (6, 2, 5, 7), dict(groups=2, padding=(2, 3), stride=(1, 2))),
        ((2, 3, 25, 500), (3, 3, 3, 31), dict(padding=(1, 15), padding_mode="reflect")),
        ((1, 2, 7, 9, 400), (4, 2, 3, 2, 11), dict(dilation=(1, 1, 2), padding=(1, 0, 10), padding_mode="circular")),
        ((2, 2, 6, 300, 70), (2, 2, 2, 9, 5), dict(stride=(1, 2, 1), padding=(1, 4, 2), padding_mode="reflect")),
    ]
    for xs, ws, kw in cases:
        x = torch.randn(*xs, generator=gen).to(DEV).requires_grad_()
        w = torch.randn(*ws, generator=gen).to(DEV).requires_grad_()
        b = torch.randn(ws[0], generator=gen).to(DEV)
        for var in ("FFTCONV_XTILE", "FFTCONV_YTILE"):
            monkeypatch.delenv(var, raising=False)
        _native.clear_plan_cache()
        y0 = fft_conv(x, w, b, **kw)
        gy = torch.randn(y0.shape, generator=gen).to(DEV)
        gx0, gw0 = torch.autograd.grad(y0, (x, w), gy)
        for forced in (("FFTCONV_XTILE",), ("FFTCONV_YTILE",), ("FFTCONV_XTILE", "FFTCONV_YTILE")):
            for var in forced:
                monkeypatch.setenv(var, str(xtile))
            _native.clear_plan_cache()
            y1 = fft_conv(x, w, b, **kw)
            gx1, gw1 = torch.autograd.grad(y1, (x, w), gy)
            for var in forced:
                monkeypatch.delenv(var, raising=False)
            _native.clear_plan_cache()
            assert _rel(y1, y0) < 1e-5 and _rel(gx1, gx0) < 1e-5 and _rel(gw1, gw0) < 1e-5, (xs, kw, forced)


# ----------------------------------------------------------------------------- N4 supersets: padding strings, half precision
def test_string_padding_and_half_precision_inputs():
    """``padding='same' | 'valid'`` as torch's direct convolution takes them (the reference rejects strings), for
    even and odd dilated extents, functional and module; float16 / bfloat16 tensors go through the fp32 kernels."""
    import fft_conv_pytorch_amd as fca
    from fft_conv_pytorch_amd.functional import fft_conv
    gen = torch.Generator().manual_seed(91)
    for xs, ws, dil, mode in [((2, 4, 301), (6, 4, 8), 1, "constant"), ((2, 4, 300), (6, 2, 5), 3, "constant"),
                              ((1, 3, 40, 51), (5, 3, 4, 6), (1, 2), "constant"), ((2, 2, 9, 10, 11), (2, 2, 2, 3, 4), 1, "constant"),
                              ((2, 3, 120), (3, 3, 6), 2, "reflect")]:
        x = torch.randn(*xs, generator=gen).to(DEV)
        w = torch.randn(*ws, generator=gen).to(DEV)
        b = torch.randn(ws[0], generator=gen).to(DEV)
        nd = len(xs) - 2
        groups = xs[1] // ws[1]
        conv = getattr(F, f"conv{nd}d")
        for pad in ("same", "valid"):
            if mode == "constant":
                want = conv(x.double(), w.double(), b.double(), padding=pad, dilation=dil, groups=groups)
            else:
                layer = getattr(torch.nn, f"Conv{nd}d")(xs[1], ws[0], ws[2:], padding=pad, dilation=dil, groups=groups,
                                                        padding_mode=mode).to(DEV).double()
                with torch.no_grad():
                    layer.weight.copy_(w.double())
                    layer.bias.copy_(b.double())
                want = layer(x.double()).detach()
            got = fft_conv(x, w, b, padding=pad, dilation=dil, groups=groups, padding_mode=mode)
            assert got.shape == want.shape, (xs, ws, pad, got.shape, want.shape)
            assert _rel(got, want) < REL_TOL, (xs, ws, pad)
    layer = fca.FFTConv2d(3, 5, (4, 3), padding="same", bias=True).to(DEV)
    x = torch.randn(2, 3, 20, 21, device=DEV, requires_grad=True)
    y = layer(x)
    want = F.conv2d(x, layer.weight, layer.bias, padding="same")
    assert y.shape == want.shape and _rel(y, want) < REL_TOL
    gy = torch.randn_like(y)
    gx, gw = torch.autograd.grad(y, (x, layer.weight), gy)
    gx_r, gw_r = torch.autograd.grad(want, (x, layer.weight), gy)
    assert _rel(gx, gx_r) < REL_TOL and _rel(gw, gw_r) < REL_TOL
    for dt, tol in ((torch.bfloat16, 2e-2), (torch.float16, 3e-3)):
        xh = torch.randn(2, 8, 3000, device=DEV).to(dt)
        wh = (torch.randn(8, 8, 65, device=DEV) / 20).to(dt)
        bh = torch.randn(8, device=DEV).to(dt)
        yh = fft_conv(xh, wh, bh, padding=32)
        assert yh.dtype == dt
        want = F.conv1d(xh.double(), wh.double(), bh.double(), padding=32)
        assert _rel(yh, want) < tol
    with pytest.raises(TypeError, match="must share"):
        fft_conv(torch.zeros(1, 2, 16, device=DEV, dtype=torch.float64), torch.zeros(2, 2, 3, device=DEV))


def test_float64_tensors_direct_kernel():
    """float64 in, float64 out (the reference is dtype-agnostic: functional.py:19-89 runs in float64 on the CPU): the
    FC_F64 plans' direct kernel against torch's float64 convolutions at 1e-10 -- forward with every padding mode,
    stride, dilation, groups in 1-D / 2-D / 3-D, transposed with output_padding, gradients, modules."""
    import fft_conv_pytorch_amd as fca
    from fft_conv_pytorch_amd.functional import fft_conv, fft_conv_transpose
    gen = torch.Generator().manual_seed(64)
    tol = 1e-10
    cases = [
        ((2, 4, 200), (6, 2, 7), dict(groups=2, stride=3, padding=5, dilation=2, padding_mode="reflect")),
        ((2, 3, 20, 33), (5, 3, 3, 5), dict(stride=(1, 2), padding=(2, 1), dilation=(2, 1), padding_mode="circular")),
        ((1, 4, 9, 12, 10), (2, 2, 2, 3, 4), dict(groups=2, stride=(1, 2, 1), padding=(1, 0, 2), dilation=(2, 1, 1), padding_mode="constant")),
        ((2, 2, 64), (3, 2, 6), dict(padding=3, padding_mode="replicate")),
        ((2, 8, 3000), (8, 8, 129), dict(padding=64)),
    ]
    for xs, ws, kw in cases:
        nd = len(xs) - 2
        x = torch.randn(*xs, generator=gen, dtype=torch.float64).to(DEV).requires_grad_()
        w = torch.randn(*ws, generator=gen, dtype=torch.float64).to(DEV).requires_grad_()
        b = torch.randn(ws[0], generator=gen, dtype=torch.float64).to(DEV).requires_grad_()
        mode = kw.get("padding_mode", "constant")
        pad = kw.get("padding", 0)
        pads = (pad,) * nd if isinstance(pad, int) else tuple(pad)
        kwt = {k: v for k, v in kw.items() if k not in ("padding", "padding_mode")}
        xr, wr, br = (t.detach().clone().requires_grad_() for t in (x, w, b))
        if mode == "constant":
            want = getattr(F, f"conv{nd}d")(xr, wr, br, padding=pads, **kwt)
        else:
            flat = []
            for p_ in reversed(pads):
                flat += [p_, p_]
            want = getattr(F, f"conv{nd}d")(F.pad(xr, flat, mode=mode), wr, br, **kwt)
        got = fft_conv(x, w, b, **kw)
        assert got.dtype == torch.float64 and got.shape == want.shape
        assert _rel(got, want) < tol, (xs, kw, _rel(got, want))
        gy = torch.randn(want.shape, generator=gen, dtype=torch.float64).to(DEV)
        got.backward(gy)
        want.backward(gy)
        for name, a_, b_ in (("dX", x.grad, xr.grad), ("dW", w.grad, wr.grad), ("db", b.grad, br.grad)):
            assert _rel(a_, b_) < tol, (name, xs, kw, _rel(a_, b_))
    # transposed, with output_padding >= stride on one axis
    x = torch.randn(2, 4, 9, 11, generator=gen, dtype=torch.float64).to(DEV).requires_grad_()
    w = torch.randn(4, 3, 3, 4, generator=gen, dtype=torch.float64).to(DEV).requires_grad_()
    kw = dict(stride=(2, 3), padding=(1, 2), output_padding=(2, 1), dilation=(3, 1), groups=2)
    xr, wr = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    want = F.conv_transpose2d(xr, wr, None, **kw)
    got = fft_conv_transpose(x, w, None, **kw)
    assert got.shape == want.shape and _rel(got, want) < tol
    gy = torch.randn(want.shape, generator=gen, dtype=torch.float64).to(DEV)
    got.backward(gy)
    want.backward(gy)
    assert _rel(x.grad, xr.grad) < tol and _rel(w.grad, wr.grad) < tol
    # modules follow .double()
    layer = fca.FFTConv1d(3, 5, 9, padding=4).to(DEV).double().eval()
    xin = torch.randn(2, 3, 100, device=DEV, dtype=torch.float64)
    with torch.no_grad():
        assert _rel(layer(xin), F.conv1d(xin, layer.weight, layer.bias, padding=4)) < tol
    # G5 fixtures (float32 reference outputs) are reproduced to float32 accuracy by the float64 kernel too
    for n, kind, meta, xg, wg, bg, gyg, gold in gu.g5_cases():
        if xg.size > 5000:
            continue
        fn = fft_conv if kind == "fwd" else fft_conv_transpose
        y = fn(torch.from_numpy(xg).double().to(DEV), torch.from_numpy(wg).double().to(DEV),
               bias=torch.from_numpy(bg).double().to(DEV), **gu.g5_kwargs(kind, meta))
        gu.check_entry(y.cpu().numpy(), gold["y"], 2e-6)


# ----------------------------------------------------------------------------- dilation phases in pairs (8-byte accesses)
def test_paired_dilation_phases_match_single_phases(monkeypatch):
    """An even dilation runs its phases in pairs on the batch-sharing kernel: a lane loads / stores both phases of a
    position as 8 bytes and trades halves with its partner lane (v_permlane32_swap).  Against the one-phase-per-slot
    build (FFTCONV_PH2=0) and torch, for row ends where the odd phase is one sample shorter, odd paddings (unaligned
    8-byte accesses), every padding mode (border tiles take the per-sample path), groups and both work-item sizes."""
    from fft_conv_pytorch_amd import _native
    from fft_conv_pytorch_amd.functional import fft_conv
    gen = torch.Generator().manual_seed(2024)
    cases = [  # batch, channels, groups, L, k, dilation, padding, mode
        (8, 8, 1, 40001, 129, 4, 0, "constant"),        # Lfull odd: phase lengths differ
        (4, 16, 2, 30000, 257, 2, 255, "constant"),     # odd padding
        (3, 8, 1, 20011, 65, 6, 33, "reflect"),
        (5, 8, 1, 9000, 33, 4, 64, "circular"),
        (2, 24, 3, 70000, 200, 2, 7, "replicate"),
        (16, 8, 1, 5000, 17, 4, 3, "constant"),          # short rows: mostly border tiles
    ]
    for B, C, g, L, k, dil, pad, mode in cases:
        x = torch.randn(B, C, L, generator=gen).to(DEV)
        w = (torch.randn(C, C // g, k, generator=gen) / math.sqrt(C // g * k)).to(DEV)
        b = torch.randn(C, generator=gen).to(DEV)
        outs = {}
        for ph2 in ("1", "0"):
            monkeypatch.setenv("FFTCONV_PH2", ph2)       # (read at plan creation; "1" = pairs even where quads would run)
            _native.clear_plan_cache()
            outs[ph2] = fft_conv(x, w, b, padding=pad, padding_mode=mode, dilation=dil, groups=g)
        xd = x.double()
        if mode != "constant" and pad:
            xd = F.pad(xd, (pad, pad), mode=mode)
            want = F.conv1d(xd, w.double(), b.double(), dilation=dil, groups=g)
        else:
            want = F.conv1d(xd, w.double(), b.double(), padding=pad, dilation=dil, groups=g)
        assert _rel(outs["1"], want) < REL_TOL, (B, C, g, L, k, dil, pad, mode)
        assert _rel(outs["0"], want) < REL_TOL
    monkeypatch.delenv("FFTCONV_PH2", raising=False)
    _native.clear_plan_cache()


# ----------------------------------------------------------------------------- dilation phases in quads (16-byte accesses)
def test_dilation_phase_quads_match_pairs_and_torch(monkeypatch):
    """A dilation that is a multiple of 4 on a full 8 -> 8 channel block runs its phases in QUADS (round 3, conv1d_pers.hpp
    PH4): a complex sequence carries two phases of one channel, a lane loads / stores all four phases of a position as 16
    bytes.  Against the pairs build (FFTCONV_PH2=1), the single-phase build (=0) and torch in float64: rows whose length
    is not a multiple of 4 (the 16-byte path must fall back), odd paddings (unaligned tile positions), phases of unequal
    length at the row end, every padding mode (border tiles), dilation 8 (two quads per batch item), groups of 8 and
    batch sizes that leave tail items."""
    from fft_conv_pytorch_amd import _native
    from fft_conv_pytorch_amd.functional import fft_conv
    gen = torch.Generator().manual_seed(404)
    cases = [  # batch, channels, groups, L, k, dilation, padding, mode
        (8, 8, 1, 40000, 129, 4, 0, "constant"),         # aligned: the 16-byte path everywhere
        (8, 8, 1, 40001, 129, 4, 0, "constant"),         # L % 4 != 0: 4-byte fallback on loads and stores
        (3, 16, 2, 36864, 257, 4, 256, "constant"),      # aligned padding, two groups, odd batch
        (4, 8, 1, 30002, 65, 4, 31, "reflect"),          # odd padding, border tiles by index map
        (2, 8, 1, 50000, 33, 8, 64, "circular"),         # dilation 8: two quads per batch item
        (5, 64, 8, 20480, 100, 4, 6, "replicate"),       # cfgD-like grouping
        (16, 8, 1, 4099, 17, 4, 4, "constant"),          # short rows, unequal phase lengths
    ]
    for B, C, g, L, k, dil, pad, mode in cases:
        x = torch.randn(B, C, L, generator=gen).to(DEV)
        w = (torch.randn(C, C // g, k, generator=gen) / math.sqrt(C // g * k)).to(DEV)
        b = torch.randn(C, generator=gen).to(DEV)
        outs = {}
        for knob in ("2", "1", "0"):
            monkeypatch.setenv("FFTCONV_PH2", knob)
            _native.clear_plan_cache()
            outs[knob] = fft_conv(x, w, b, padding=pad, padding_mode=mode, dilation=dil, groups=g)
        xd = x.double()
        if mode != "constant" and pad:
            want = F.conv1d(F.pad(xd, (pad, pad), mode=mode), w.double(), b.double(), dilation=dil, groups=g)
        else:
            want = F.conv1d(xd, w.double(), b.double(), padding=pad, dilation=dil, groups=g)
        for knob in ("2", "1", "0"):
            assert _rel(outs[knob], want) < REL_TOL, (knob, B, C, g, L, k, dil, pad, mode)
    monkeypatch.delenv("FFTCONV_PH2", raising=False)
    _native.clear_plan_cache()
