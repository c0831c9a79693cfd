"""CPU-only checks of the host layer: argument normalisation, error behaviour, exports, and that the
C-ABI library loads and exports every symbol include/fftconv_amd.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_to_ntuple_contract():
    from fft_conv_pytorch_amd.functional import to_ntuple
    assert to_ntuple(2, 3) == (2, 2, 2)
    assert to_ntuple((1, 2), 2) == (1, 2)
    assert to_ntuple([4], 1) == (4,)
    with pytest.raises(ValueError, match=r"Cannot cast tuple of length 3 to length 2\."):
        to_ntuple((1, 2, 3), 2)
    with pytest.raises(ValueError, match=r"Cannot cast tuple of length 4 to length 1\."):
        to_ntuple("same", 1)


def test_package_exports_mirror_reference():
    import fft_conv_pytorch_amd as pkg
    for name in ("functional", "nn", "FFTConv1d", "FFTConv2d", "FFTConv3d", "FFTConvTranspose1d",
                 "FFTConvTranspose2d", "FFTConvTranspose3d", "fft_conv"):
        assert hasattr(pkg, name), name
    for name in ("fft_conv", "fft_conv_transpose", "complex_matmul", "to_ntuple"):
        assert hasattr(pkg.functional, name), name
    assert issubclass(pkg.FFTConv2d, torch.nn.Conv2d)
    layer = pkg.FFTConv3d(4, 6, 3, groups=2, padding=1, padding_mode="circular")
    assert set(layer.state_dict()) == {"weight", "bias"}
    assert tuple(layer.weight.shape) == (6, 2, 3, 3, 3)


def test_library_exports_every_declared_symbol():
    from fft_conv_pytorch_amd import _native
    lib = _native.load_library()
    header = open(os.path.join(ROOT, "include", "fftconv_amd.h")).read()
    declared = set(re.findall(r"\b(fc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_native.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.fc_version() == _native.ABI_VERSION
    assert ctypes.sizeof(_native.FcDesc) == 8 + 4 * 8 + 5 * 24 + 16 + 24


def test_host_validation_happens_before_any_device_call():
    from fft_conv_pytorch_amd.functional import fft_conv
    x, w = torch.zeros(2, 4, 30), torch.zeros(6, 2, 5)
    with pytest.raises(ValueError, match="Cannot cast tuple of length 2 to length 1"):
        fft_conv(x, w, groups=2, stride=(1, 2))
    with pytest.raises(ValueError, match="channel mismatch"):
        fft_conv(x, w, groups=1)
    with pytest.raises(ValueError, match="bias must have shape"):
        fft_conv(x, w, bias=torch.zeros(5), groups=2)
    with pytest.raises(ValueError, match="padding_mode"):
        fft_conv(x, w, groups=2, padding_mode="mirror")
    with pytest.raises(ValueError, match="1-3 spatial dims"):
        fft_conv(torch.zeros(2, 4), torch.zeros(6, 2), groups=2)


def test_no_cpu_fallback():
    """The product path must refuse CPU tensors instead of silently computing elsewhere."""
    from fft_conv_pytorch_amd import FFTConv1d
    from fft_conv_pytorch_amd.functional import fft_conv
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fft_conv(torch.zeros(1, 2, 16), torch.zeros(2, 2, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FFTConv1d(2, 2, 3)(torch.zeros(1, 2, 16))
    with pytest.raises(AssertionError):
        FFTConv1d(2, 2, 3)(torch.zeros(2, 16))     # ndim check first, like nn.py:11


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fft_conv_pytorch_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no CPU", ""), fn
            assert "torch.fft" not in src.replace("no torch.fft", ""), fn


def test_wgrad_descriptor_query_is_safe_without_a_device():
    """fc_wgrad1d_slices is a pure query: it must answer (0 = not covered / no device) and never raise."""
    from fft_conv_pytorch_amd import _native
    ok = _native.conv_desc(1, 4, 8, 8, 1, (4096,), (33,), (1,), (0,), (1,), 0)
    strided = _native.conv_desc(1, 4, 8, 8, 1, (4096,), (33,), (2,), (0,), (1,), 0)
    two_d = _native.conv_desc(2, 4, 8, 8, 1, (64, 64), (3, 3), (1, 1), (0, 0), (1, 1), 0)
    has_gpu = torch.cuda.is_available()
    assert (_native.wgrad1d_slices(ok) > 0) == has_gpu
    assert _native.wgrad1d_slices(strided) == 0
    assert _native.wgrad1d_slices(two_d) == 0


def test_scripts_and_entry_points_compile():
    """Helper scripts are not exercised by the suite; at least keep them syntactically valid."""
    import py_compile
    for name in sorted(os.listdir(os.path.join(ROOT, "scripts"))):
        if name.endswith(".py"):
            py_compile.compile(os.path.join(ROOT, "scripts", name), doraise=True)
    for name in ("bench.py", "__graft_entry__.py"):
        py_compile.compile(os.path.join(ROOT, name), doraise=True)


# ---------------------------------------------------------------- round 2: device handling, caches, signatures
class _FakeCudaTensor:
    """Just enough of a tensor for functional._plan_for's host-side checks (no device is touched)."""

    def __init__(self, shape, index):
        self.shape, self.ndim = torch.Size(shape), len(shape)
        self.is_cuda, self.dtype = True, torch.float32
        self.device = torch.device("cuda", index)


def test_plan_is_created_on_the_tensors_device_not_the_current_one(monkeypatch):
    """plans own device allocations, so they must be built with the TENSORS' device current
    and cached under that device's ordinal -- also when another device is current."""
    from fft_conv_pytorch_amd import _native, functional as F_
    seen = {}

    class Guard:
        def __init__(self, index):
            seen["guard"] = index

        def __enter__(self):
            seen["inside"] = True

        def __exit__(self, *a):
            seen["inside"] = False

    def fake_get_plan(index, key):
        seen["created_on"] = index
        seen["created_inside_guard"] = seen.get("inside", False)
        return "plan"

    monkeypatch.setattr(F_.torch.cuda, "device", Guard)
    monkeypatch.setattr(_native, "lookup_plan", lambda index, key: None)
    monkeypatch.setattr(_native, "get_plan", fake_get_plan)
    x, w = _FakeCudaTensor((2, 4, 64), 1), _FakeCudaTensor((6, 4, 5), 1)
    assert F_._plan_for(x, w, None, 1, 0, 1, 1, "constant") == "plan"
    assert seen["guard"] == 1 and seen["created_on"] == 1 and seen["created_inside_guard"]
    # a cached plan is returned without switching devices
    seen.clear()
    monkeypatch.setattr(_native, "lookup_plan", lambda index, key: ("cached", index))
    assert F_._plan_for(x, w, None, 1, 0, 1, 1, "constant") == ("cached", 1)
    assert "guard" not in seen
    # tensors of one call on different devices are refused before anything is launched
    with pytest.raises(ValueError, match="same device"):
        F_._plan_for(x, _FakeCudaTensor((6, 4, 5), 0), None, 1, 0, 1, 1, "constant")


def test_plan_cache_is_bounded_lru(monkeypatch):
    from fft_conv_pytorch_amd import _native

    class FakePlan:
        def __init__(self, key, device_index=0):
            self.key, self.device_index = key, device_index

    monkeypatch.setattr(_native, "Plan", FakePlan)
    monkeypatch.setattr(_native, "PLAN_CACHE_SIZE", 3)
    _native.clear_plan_cache()
    try:
        plans = [_native.get_plan(0, (i,)) for i in range(3)]
        assert _native.get_plan(0, (0,)) is plans[0]            # hit: becomes the most recent
        _native.get_plan(0, (3,))                               # evicts the least recently used = key 1
        assert _native.lookup_plan(0, (1,)) is None
        assert _native.lookup_plan(0, (0,)) is plans[0] and _native.lookup_plan(0, (2,)) is plans[2]
        assert _native.lookup_plan(1, (0,)) is None             # the device ordinal is part of the key
    finally:
        _native.clear_plan_cache()


def test_module_spectrum_cache_policy(monkeypatch):
    """the spectrum cache may only be used when the weight provably did not change."""
    from fft_conv_pytorch_amd import FFTConv1d, FFTConvTranspose1d, functional as F_
    calls = []
    monkeypatch.setattr(F_, "transform_kernel", lambda plan, w: calls.append(1) or ("spec", len(calls)))
    for layer in (FFTConv1d(4, 4, 3), FFTConvTranspose1d(4, 4, 3)):
        calls.clear()
        plan = object()
        layer.train()
        assert layer._cached_spectrum(plan) is None and not calls       # training step: re-transform like the reference
        with torch.no_grad():
            s1 = layer._cached_spectrum(plan)                            # no grad: cache is usable
            assert s1 is not None and layer._cached_spectrum(plan) is s1 and len(calls) == 1
        layer.eval()
        assert layer._cached_spectrum(plan) is s1                        # eval: same weight version -> reuse
        with torch.no_grad():
            layer.weight.mul_(2.0)                                       # version counter bumps
        s2 = layer._cached_spectrum(plan)
        assert s2 is not s1 and len(calls) == 2
        layer.weight.data.mul_(2.0)                                      # invisible to any key ...
        assert layer._cached_spectrum(plan) is s2
        layer.invalidate_kernel_spectrum()                               # ... hence the explicit hook
        assert layer._cached_spectrum(plan) is not s2 and len(calls) == 3
        assert layer._cached_spectrum(object()) is not None and len(calls) == 4   # another plan: another spectrum
        layer.cache_kernel_spectrum = False
        assert layer._cached_spectrum(plan) is None
        layer.cache_kernel_spectrum = True
        torch.nn.utils.parametrizations.weight_norm(layer)               # weight is now a fresh temporary per access
        assert layer._cached_spectrum(plan) is None


def test_transposed_ops_route_through_autograd_when_grad_is_needed(monkeypatch):
    """fft_conv_transpose must not silently detach."""
    from fft_conv_pytorch_amd import autograd as A, functional as F_
    hit = {}

    class Probe:
        @staticmethod
        def apply(*args):
            hit["args"] = args
            return "routed"

    monkeypatch.setattr(A, "FFTConvTransposeFunction", Probe)
    x = torch.zeros(1, 2, 8)
    w = torch.zeros(2, 3, 3, requires_grad=True)
    assert F_.fft_conv_transpose(x, w, stride=2, output_padding=1) == "routed"
    assert hit["args"][3:8] == ((2,), (0,), (1,), (1,), 1)
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        F_.fft_conv_transpose(x, w)          # without grad it takes the plain (device) path


def test_modules_copy_and_pickle_without_their_native_handles():
    """A module that has run carries a cached kernel spectrum and its last plan (ctypes handles) in __dict__; deepcopy
    (EMA / AveragedModel, quantization flows), pickle and torch.save(module) must see a plain nn.Conv subclass, as the
    reference's FFTConv modules are."""
    import copy
    import io
    import pickle

    import fft_conv_pytorch_amd as pkg

    class FakeHandle:                       # stands in for a _native.Plan / KernelSpectrum: refuses both protocols
        def __reduce__(self):
            raise ValueError("ctypes objects containing pointers cannot be pickled")

        def __deepcopy__(self, memo):
            raise ValueError("ctypes objects containing pointers cannot be pickled")

    for cls, args in ((pkg.FFTConv2d, (4, 6, 3)), (pkg.FFTConvTranspose1d, (4, 6, 3))):
        layer = cls(*args)
        layer.__dict__["_spectrum_cache"] = (("tag",), FakeHandle())
        layer.__dict__["_last_plan"] = (("sig",), FakeHandle())
        twin = copy.deepcopy(layer)
        assert "_spectrum_cache" not in twin.__dict__ and "_last_plan" not in twin.__dict__
        assert torch.equal(twin.weight, layer.weight) and twin.weight is not layer.weight
        assert "_spectrum_cache" in layer.__dict__          # the original keeps its cache
        back = pickle.loads(pickle.dumps(layer))
        assert torch.equal(back.weight, layer.weight) and "_last_plan" not in back.__dict__
        buf = io.BytesIO()
        torch.save(layer, buf)
        buf.seek(0)
        again = torch.load(buf, weights_only=False)
        assert torch.equal(again.bias, layer.bias)
        # inside a container, as model-averaging utilities copy whole models
        model = torch.nn.Sequential(layer, torch.nn.ReLU())
        assert torch.equal(copy.deepcopy(model)[0].weight, layer.weight)


def test_unit_map_division_constants_are_exact():
    """csrc/fft_engine.hpp make_fastdiv / fdiv: the pass kernels split blockIdx by launch constants with a host-prepared
    multiply-shift (m = floor(2^(31+l)/d) + 1, l = ceil(log2 d)); the formula restated here must be exact for every
    dividend below 2^31 -- a wrong quotient would send a workgroup to another image's rows."""
    import random

    def make(d):
        l = 0
        while (1 << l) < d:
            l += 1
        m = ((1 << (31 + l)) // d) + 1
        assert m < (1 << 32)
        return m, 31 + l

    rng = random.Random(7)
    divisors = list(range(1, 600)) + [(1 << k) + e for k in range(1, 31) for e in (-1, 0, 1)] + [rng.randrange(1, 1 << 31) for _ in range(3000)]
    top = (1 << 31) - 1
    for d in divisors:
        m, sh = make(d)
        q = top // d
        probes = {0, 1, d - 1, d, min(d + 1, top), top, top - 1, q * d, max(q * d - 1, 0)}
        probes |= {rng.randrange(0, top + 1) for _ in range(8)}
        probes |= {min(rng.randrange(0, q + 1) * d + e, top) for e in (0, d - 1) for _ in range(4)}
        for n in probes:
            assert (n * m) >> sh == n // d, (n, d)
