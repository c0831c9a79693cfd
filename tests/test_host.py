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
