"""ctypes binding of libfftconv_amd.so (C ABI in include/fftconv_amd.h).

There is deliberately no fallback: if the library is missing or a call fails,
an exception is raised.  The product path never computes on the CPU.
"""
from __future__ import annotations

import collections
import ctypes
import os
import threading
from typing import Optional, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libfftconv_amd.so"
LIB_PATH = os.environ.get("FFTCONV_LIB") or os.path.join(_HERE, LIB_NAME)   # env: A/B builds while tuning

FC_OK, FC_ERR_INVALID, FC_ERR_UNSUPPORTED, FC_ERR_HIP = 0, 1, 2, 3
PAD_MODES = {"constant": 0, "zeros": 0, "reflect": 1, "replicate": 2, "circular": 3}
ABI_VERSION = 6

EXPORTS = (
    "fc_version", "fc_last_error", "fc_plan_create", "fc_plan_destroy", "fc_output_shape",
    "fc_kernel_spectrum_bytes", "fc_workspace_bytes", "fc_plan_tile", "fc_plan_layout", "fc_transform_kernel",
    "fc_forward", "fc_forward_stamped", "fc_wgrad1d_slices", "fc_wgrad1d", "fc_wgrad1d_db", "fc_wgrad1d_db_supported",
    "fc_debug_grid", "fc_wgrad_nd_plan_create", "fc_wgrad_nd",
)


class FcDesc(ctypes.Structure):
    """Mirror of ``struct fc_desc``."""
    _fields_ = [
        ("ndim", ctypes.c_int32), ("dtype", ctypes.c_int32),
        ("batch", ctypes.c_int64), ("in_channels", ctypes.c_int64), ("out_channels", ctypes.c_int64),
        ("groups", ctypes.c_int64),
        ("spatial", ctypes.c_int64 * 3), ("kernel", ctypes.c_int64 * 3), ("stride", ctypes.c_int64 * 3),
        ("padding", ctypes.c_int64 * 3), ("dilation", ctypes.c_int64 * 3),
        ("padding_mode", ctypes.c_int32), ("has_bias", ctypes.c_int32), ("tile_hint", ctypes.c_int32),
        ("transposed", ctypes.c_int32),
        ("output_padding", ctypes.c_int64 * 3),
    ]


class NativeLibraryMissing(ImportError):
    pass


_lib = None
_lib_lock = threading.Lock()


def load_library() -> ctypes.CDLL:
    """Load libfftconv_amd.so (built in-tree by ``__graft_entry__.build()`` / ``make``)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C fft_conv_pytorch_amd/csrc`). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        lib.fc_version.restype = i32
        lib.fc_last_error.restype = ctypes.c_char_p
        lib.fc_plan_create.argtypes = [ctypes.POINTER(FcDesc), ctypes.POINTER(vp)]
        lib.fc_plan_create.restype = i32
        lib.fc_plan_destroy.argtypes = [vp]
        lib.fc_plan_destroy.restype = None
        lib.fc_output_shape.argtypes = [vp, ctypes.POINTER(ctypes.c_int64 * 3)]
        lib.fc_output_shape.restype = i32
        lib.fc_kernel_spectrum_bytes.argtypes = [vp]
        lib.fc_kernel_spectrum_bytes.restype = sz
        lib.fc_workspace_bytes.argtypes = [vp]
        lib.fc_workspace_bytes.restype = sz
        lib.fc_plan_tile.argtypes = [vp]
        lib.fc_plan_tile.restype = i32
        lib.fc_transform_kernel.argtypes = [vp, vp, vp, vp, vp]
        lib.fc_transform_kernel.restype = i32
        lib.fc_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        lib.fc_forward.restype = i32
        lib.fc_wgrad1d_slices.argtypes = [ctypes.POINTER(FcDesc)]
        lib.fc_wgrad1d_slices.restype = i32
        lib.fc_wgrad1d.argtypes = [ctypes.POINTER(FcDesc), vp, vp, vp, i32, vp]
        lib.fc_wgrad1d.restype = i32
        lib.fc_wgrad1d_db.argtypes = [ctypes.POINTER(FcDesc), vp, vp, vp, vp, ctypes.c_longlong, i32, vp]
        lib.fc_wgrad1d_db.restype = i32
        lib.fc_wgrad1d_db_supported.argtypes = [ctypes.POINTER(FcDesc)]
        lib.fc_wgrad1d_db_supported.restype = i32
        lib.fc_wgrad_nd_plan_create.argtypes = [ctypes.POINTER(FcDesc), ctypes.POINTER(vp)]
        lib.fc_wgrad_nd_plan_create.restype = i32
        lib.fc_wgrad_nd.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        lib.fc_wgrad_nd.restype = i32
        lib.fc_forward_stamped.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        lib.fc_forward_stamped.restype = i32
        lib.fc_plan_layout.argtypes = [vp, ctypes.POINTER(ctypes.c_int32 * 8)]
        lib.fc_plan_layout.restype = i32
        lib.fc_debug_grid.argtypes = [vp]
        lib.fc_debug_grid.restype = ctypes.c_longlong
        if lib.fc_version() != ABI_VERSION:
            raise ImportError(f"{LIB_NAME}: ABI version {lib.fc_version()} != {ABI_VERSION}")
        _lib = lib
    return _lib


def conv_desc(ndim, batch, cin, cout, groups, spatial, kernel, stride, padding, dilation, mode) -> FcDesc:
    """``struct fc_desc`` of a forward convolution (used by the calls that take a descriptor, not a plan)."""
    d = FcDesc()
    d.ndim, d.dtype = ndim, 0
    d.batch, d.in_channels, d.out_channels, d.groups = batch, cin, cout, groups
    for i in range(3):
        d.spatial[i] = spatial[i] if i < ndim else 1
        d.kernel[i] = kernel[i] if i < ndim else 1
        d.stride[i] = stride[i] if i < ndim else 1
        d.padding[i] = padding[i] if i < ndim else 0
        d.dilation[i] = dilation[i] if i < ndim else 1
        d.output_padding[i] = 0
    d.padding_mode, d.has_bias, d.tile_hint, d.transposed = mode, 0, 0, 0
    return d


def wgrad1d_slices(desc: FcDesc) -> int:
    """Partial-result count of ``fc_wgrad1d`` for this shape on the current device; 0 = not covered."""
    return int(load_library().fc_wgrad1d_slices(ctypes.byref(desc)))


def wgrad1d(desc: FcDesc, x_ptr: int, dy_ptr: int, partial_ptr: int, slices: int, stream: int):
    lib = load_library()
    st = lib.fc_wgrad1d(ctypes.byref(desc), x_ptr, dy_ptr, partial_ptr, slices, stream)
    if st != FC_OK:
        _raise(lib, st)


def wgrad1d_db_supported(desc: FcDesc) -> bool:
    return bool(load_library().fc_wgrad1d_db_supported(ctypes.byref(desc)))


def wgrad1d_db(desc: FcDesc, x_ptr: int, dy_ptr: int, partial_ptr: int, db_ptr: Optional[int], slice_stride: int,
               slices: int, stream: int):
    """fc_wgrad1d with the bias gradient folded in: rows [dW | db] of ``slice_stride`` floats per slice."""
    lib = load_library()
    st = lib.fc_wgrad1d_db(ctypes.byref(desc), x_ptr, dy_ptr, partial_ptr, db_ptr, slice_stride, slices, stream)
    if st != FC_OK:
        _raise(lib, st)


class WgradPlan:
    """Owns the plan of ``fc_wgrad_nd`` for one convolution descriptor (2-D / 3-D, float32): created on the CURRENT HIP
    device.  ``run`` reads x (B, Cin, *S) and dY (B, Cout, *Lout) and writes dW (Cout, Cin/g, *k) -- no copies around it."""

    def __init__(self, desc: FcDesc):
        lib = load_library()
        handle = ctypes.c_void_p()
        st = lib.fc_wgrad_nd_plan_create(ctypes.byref(desc), ctypes.byref(handle))
        if st != FC_OK:
            _raise(lib, st)
        self._lib, self._h = lib, handle
        self.spectrum_bytes = int(lib.fc_kernel_spectrum_bytes(handle))
        self.workspace_bytes = int(lib.fc_workspace_bytes(handle))
        self.tile = int(lib.fc_plan_tile(handle))

    def run(self, x_ptr: int, dy_ptr: int, dw_ptr: int, spectrum_ptr: int, workspace_ptr: Optional[int], stream: int):
        st = self._lib.fc_wgrad_nd(self._h, x_ptr, dy_ptr, dw_ptr, spectrum_ptr, workspace_ptr, stream)
        if st != FC_OK:
            _raise(self._lib, st)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.fc_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _raise(lib, status: int):
    msg = lib.fc_last_error().decode("utf-8", "replace")
    if status == FC_ERR_INVALID:
        raise ValueError(msg)
    if status == FC_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(f"libfftconv_amd: {msg}")


class Plan:
    """Owns one ``fc_plan`` (immutable after creation; shareable between threads and streams).

    The library builds a plan for the HIP device that is current at creation (twiddle tables, work list,
    CU count): create it under ``torch.cuda.device(index)`` and pass that index as ``device_index``."""

    def __init__(self, key: Tuple, device_index: int = 0):
        (ndim, batch, cin, cout, groups, spatial, kernel, stride, padding, dilation, mode, has_bias, tile_hint,
         transposed, output_padding, dtype_code) = key
        lib = load_library()
        d = FcDesc()
        d.ndim, d.dtype = ndim, dtype_code
        d.batch, d.in_channels, d.out_channels, d.groups = batch, cin, cout, groups
        for i in range(3):
            d.spatial[i] = spatial[i] if i < ndim else 1
            d.kernel[i] = kernel[i] if i < ndim else 1
            d.stride[i] = stride[i] if i < ndim else 1
            d.padding[i] = padding[i] if i < ndim else 0
            d.dilation[i] = dilation[i] if i < ndim else 1
            d.output_padding[i] = output_padding[i] if i < ndim else 0
        d.padding_mode, d.has_bias, d.tile_hint, d.transposed = mode, int(has_bias), tile_hint, int(transposed)
        handle = ctypes.c_void_p()
        st = lib.fc_plan_create(ctypes.byref(d), ctypes.byref(handle))
        if st != FC_OK:
            _raise(lib, st)
        self._lib, self._h, self.key = lib, handle, key
        self.device_index = int(device_index)
        import torch
        self.dtype = torch.float64 if dtype_code == 1 else torch.float32
        out = (ctypes.c_int64 * 3)()
        lib.fc_output_shape(handle, ctypes.byref(out))
        self.out_spatial = tuple(int(out[i]) for i in range(ndim))
        self.spectrum_bytes = int(lib.fc_kernel_spectrum_bytes(handle))
        self.workspace_bytes = int(lib.fc_workspace_bytes(handle))
        self.tile = int(lib.fc_plan_tile(handle))
        lay = (ctypes.c_int32 * 8)()
        lib.fc_plan_layout(handle, ctypes.byref(lay))
        # what the byte layout of the kernel spectrum depends on besides the descriptor: two plans with
        # equal signatures accept each other's spectra (used by the multi-GPU broadcast)
        self.layout = tuple(int(v) for v in lay)
        self._stamps = None

    def signature(self) -> Tuple:
        """(spectrum bytes, layout words) -- equal on two plans iff a kernel spectrum is interchangeable."""
        return (self.spectrum_bytes,) + self.layout

    def transform_kernel(self, weight_ptr: int, w_hat_ptr: int, workspace_ptr: Optional[int], stream: int):
        st = self._lib.fc_transform_kernel(self._h, weight_ptr, w_hat_ptr, workspace_ptr, stream)
        if st != FC_OK:
            _raise(self._lib, st)

    def forward(self, x_ptr: int, w_hat_ptr: int, bias_ptr: Optional[int], y_ptr: int,
                workspace_ptr: Optional[int], stream: int):
        if self._stamps is not None:      # profiling run (scripts/phase_profile.py)
            st = self._lib.fc_forward_stamped(self._h, x_ptr, w_hat_ptr, bias_ptr, y_ptr, workspace_ptr, stream,
                                              self._stamps)
        else:
            st = self._lib.fc_forward(self._h, x_ptr, w_hat_ptr, bias_ptr, y_ptr, workspace_ptr, stream)
        if st != FC_OK:
            _raise(self._lib, st)

    def debug_set_stamps(self, ptr: Optional[int]):
        """Profiling hook of this Python handle (the native plan stays immutable): while set, ``forward``
        goes through ``fc_forward_stamped`` with this device buffer."""
        self._stamps = ptr

    def debug_grid(self) -> int:
        return int(self._lib.fc_debug_grid(self._h))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.fc_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass


# Plan cache keyed on (device, descriptor): least-recently-used, bounded -- variable-length inputs would
# otherwise keep one plan (and its device work list) alive per shape ever seen.  An evicted plan is
# destroyed when the last KernelSpectrum / module that still refers to it lets go.
PLAN_CACHE_SIZE = int(os.environ.get("FFTCONV_PLAN_CACHE", "128"))
_plans: "collections.OrderedDict[Tuple, Plan]" = collections.OrderedDict()
_plans_lock = threading.Lock()


def lookup_plan(device_index: int, key: Tuple) -> Optional[Plan]:
    full = (device_index,) + key
    with _plans_lock:
        plan = _plans.get(full)
        if plan is not None:
            _plans.move_to_end(full)
        return plan


def get_plan(device_index: int, key: Tuple) -> Plan:
    """Cached plan for (device, descriptor); creates it on the CURRENT HIP device, which the caller
    must have set to ``device_index`` (``functional._plan_for`` does)."""
    full = (device_index,) + key
    with _plans_lock:
        plan = _plans.get(full)
        if plan is None:
            plan = Plan(key, device_index)
            _plans[full] = plan
            while len(_plans) > max(1, PLAN_CACHE_SIZE):
                _plans.popitem(last=False)
        else:
            _plans.move_to_end(full)
    return plan


def clear_plan_cache():
    with _plans_lock:
        _plans.clear()
