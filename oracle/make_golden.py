#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference in the build container.

Runs only where ``/root/reference`` exists (never on the GPU box).  It imports
the reference package from its read-only location under an alias module name
(so it can coexist with this repo's own packages), feeds it seeded inputs and
stores inputs + outputs as small ``.npz`` fixtures.  Nothing of the reference's
source is copied; the fixtures are data only.

    python oracle/make_golden.py            # rewrites tests/golden/*.npz

Fixture families:
  G1  the reference's own test grid (tests/test_functional.py:11-20), thinned
  G2  multi-channel groups, per-axis hyper-parameters, the four padding modes
  G3  the BASELINE.json configs: seed, input checksum, sampled outputs
  G4  transposed convolution (forward outputs)
  G5  gradients: the reference's own autograd (dX, dW, db of sum(y * gy) with a seeded gy) for forward and
      transposed cases, including every BASELINE config at reduced batch / length
"""
from __future__ import annotations

import importlib.util
import itertools
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
REF_ROOT = "/root/reference/fft_conv_pytorch"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")


def load_reference():
    spec = importlib.util.spec_from_file_location(
        "_reference_fft_conv_pytorch", os.path.join(REF_ROOT, "__init__.py"),
        submodule_search_locations=[REF_ROOT])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def seeded(seed, *shape):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def make_inputs(seed, batch, cin, cout, groups, spatial, ksize):
    x = seeded(seed, batch, cin, *spatial)
    w = seeded(seed + 1, cout, cin // groups, *ksize)
    b = seeded(seed + 2, cout)
    return x, w, b


def run_ref_transpose(ref, x, w, b, **kw):
    import torch
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = ref.functional.fft_conv_transpose(torch.from_numpy(x), torch.from_numpy(w),
                                              bias=None if b is None else torch.from_numpy(b), **kw)
    return y.contiguous().numpy()


def g4_cases():
    """Transposed convolution: thinned copy of the reference grid (tests/test_functional_transpose.py:11-21,
    including its `dilation += output_padding; stride += output_padding` adjustment) plus longer rows."""
    grid = itertools.product([1, 2, 3], [7, 8], [2, 3], [2, 3], [1, 2, 3], [2, 3], [0, 1], [1, 2], [1, 2], [0, 1, 2])
    cases = []
    for idx, (nd, size, cin, cout, groups, k, pad, stride, dil, opad) in enumerate(grid):
        g = int(np.gcd(cin, np.gcd(cout, groups)))
        keep = (idx % 17 == 0) if nd == 1 else ((idx % 61 == 0) if nd == 2 else (idx % 331 == 0))
        if not keep:
            continue
        cases.append(dict(batch=2, cin=cin, cout=cout, groups=g, spatial=(size,) * nd, k=(k,) * nd, padding=pad,
                          stride=stride + opad, dilation=dil + opad, output_padding=opad))
    cases += [
        dict(batch=2, cin=8, cout=8, groups=1, spatial=(3000,), k=(129,), padding=64, stride=1, dilation=1, output_padding=0),
        dict(batch=1, cin=8, cout=16, groups=2, spatial=(1000,), k=(33,), padding=5, stride=3, dilation=2, output_padding=2),
        dict(batch=2, cin=6, cout=4, groups=2, spatial=(40, 37), k=(5, 4), padding=(2, 1), stride=(2, 1), dilation=(1, 2), output_padding=(1, 0)),
        dict(batch=1, cin=4, cout=6, groups=1, spatial=(9, 10, 8), k=(3, 2, 3), padding=(1, 0, 2), stride=(2, 1, 2), dilation=1, output_padding=(1, 0, 1)),
        dict(batch=2, cin=3, cout=5, groups=1, spatial=(50,), k=(4,), padding=7, stride=2, dilation=3, output_padding=1),
    ]
    return cases


def run_ref(ref, x, w, b, **kw):
    import torch
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = ref.functional.fft_conv(torch.from_numpy(x), torch.from_numpy(w),
                                    bias=None if b is None else torch.from_numpy(b), **kw)
    return y.contiguous().numpy()


def g1_cases():
    """Thinned copy of the reference grid: every value of every axis appears."""
    grid = itertools.product([1, 2, 3], [7, 8], [2, 3], [2, 3], [1, 2, 3], [2, 3], [0, 1], [1, 2], [1, 2])
    cases = []
    for idx, (nd, size, cin, cout, groups, k, pad, stride, dil) in enumerate(grid):
        g = np.gcd(cin, np.gcd(cout, groups))
        if idx % 7 not in (0, 3) and not (nd == 1 and idx % 2 == 0):
            continue
        if nd == 3 and idx % 21 != 0:
            continue
        cases.append(dict(ndim=nd, size=size, cin=cin, cout=cout, groups=int(g), k=k,
                          padding=pad, stride=stride, dilation=dil))
    return cases


def g2_cases():
    return [
        dict(batch=2, cin=8, cout=8, groups=2, spatial=(257,), k=(9,), stride=1, padding=4, dilation=1, mode="constant"),
        dict(batch=1, cin=64, cout=64, groups=8, spatial=(4096,), k=(33,), stride=1, padding=0, dilation=4, mode="constant"),
        dict(batch=3, cin=6, cout=4, groups=2, spatial=(100,), k=(7,), stride=3, padding=5, dilation=2, mode="reflect"),
        dict(batch=2, cin=4, cout=6, groups=1, spatial=(64,), k=(5,), stride=2, padding=3, dilation=1, mode="replicate"),
        dict(batch=2, cin=5, cout=3, groups=1, spatial=(50,), k=(4,), stride=1, padding=6, dilation=3, mode="circular"),
        dict(batch=2, cin=4, cout=4, groups=2, spatial=(20, 33), k=(3, 5), stride=(1, 2), padding=(2, 1), dilation=(2, 1), mode="constant"),
        dict(batch=1, cin=3, cout=5, groups=1, spatial=(31, 18), k=(4, 3), stride=(2, 1), padding=(1, 2), dilation=(1, 2), mode="reflect"),
        dict(batch=2, cin=2, cout=2, groups=1, spatial=(16, 16), k=(5, 5), stride=1, padding=2, dilation=1, mode="circular"),
        dict(batch=2, cin=2, cout=4, groups=2, spatial=(16, 17), k=(3, 3), stride=1, padding=1, dilation=1, mode="replicate"),
        dict(batch=1, cin=4, cout=2, groups=2, spatial=(9, 12, 10), k=(2, 3, 4), stride=(1, 2, 1), padding=(1, 0, 2), dilation=(2, 1, 1), mode="constant"),
        dict(batch=2, cin=3, cout=3, groups=3, spatial=(8, 8, 8), k=(3, 3, 3), stride=1, padding=1, dilation=1, mode="replicate"),
        dict(batch=1, cin=2, cout=3, groups=1, spatial=(10, 9, 11), k=(3, 2, 3), stride=2, padding=(2, 1, 2), dilation=1, mode="circular"),
        dict(batch=1, cin=2, cout=2, groups=1, spatial=(12, 10, 9), k=(2, 2, 2), stride=1, padding=1, dilation=2, mode="reflect"),
        # long 1D rows: several overlap-save tiles in the HIP path
        dict(batch=2, cin=8, cout=8, groups=1, spatial=(9000,), k=(300,), stride=1, padding=0, dilation=1, mode="constant"),
        dict(batch=1, cin=3, cout=5, groups=1, spatial=(5000,), k=(129,), stride=2, padding=64, dilation=1, mode="reflect"),
        dict(batch=1, cin=16, cout=24, groups=1, spatial=(3000,), k=(65,), stride=1, padding=32, dilation=1, mode="constant"),
    ]


BASELINE_CONFIGS = {
    # name: (batch, cin, cout, groups, spatial, kernel, dilation)   -- SURVEY.md section 8d
    "cfg0": (1, 8, 8, 1, (32768,), (128,), 1),
    "cfgA": (32, 8, 8, 1, (32768,), (512,), 1),
    "cfgB": (16, 8, 8, 1, (512, 512), (31, 31), 1),
    "cfgC": (8, 8, 8, 1, (64, 64, 64), (9, 9, 9), 1),
    "cfgD_b1": (1, 64, 64, 8, (1 << 20,), (257,), 4),   # one batch item of cfgD
}


def make_g4(ref):
    store = {}
    cases = g4_cases()
    for n, c in enumerate(cases):
        seed = 7000 + 3 * n
        x = seeded(seed, c["batch"], c["cin"], *c["spatial"])
        w = seeded(seed + 1, c["cin"], c["cout"] // c["groups"], *c["k"])      # transposed layout (Cin, Cout/g, *k)
        b = seeded(seed + 2, c["cout"])
        y = run_ref_transpose(ref, x, w, b, stride=c["stride"], padding=c["padding"],
                              output_padding=c["output_padding"], dilation=c["dilation"], groups=c["groups"])
        store[f"x{n}"], store[f"w{n}"], store[f"b{n}"], store[f"y{n}"] = x, w, b, y
        store[f"meta{n}"] = np.array(repr(c))
    store["count"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT_DIR, "g4_transpose.npz"), **store)
    print("G4:", len(cases), "cases")


def g5_cases():
    """(kind, case): kind 'fwd' uses fft_conv (kernel (Cout, Cin/g, *k)), 'tr' fft_conv_transpose (kernel (Cin, Cout/g, *k))."""
    fwd = [
        # the reference's grid values (tests/test_functional.py:11-20), one per axis combination of interest
        dict(batch=2, cin=2, cout=3, groups=1, spatial=(7,), k=(2,), stride=1, padding=0, dilation=1, mode="constant"),
        dict(batch=2, cin=3, cout=3, groups=3, spatial=(8,), k=(3,), stride=2, padding=1, dilation=2, mode="constant"),
        dict(batch=2, cin=2, cout=2, groups=2, spatial=(8, 7), k=(3, 2), stride=(1, 2), padding=(1, 0), dilation=(2, 1), mode="constant"),
        dict(batch=2, cin=3, cout=2, groups=1, spatial=(7, 8, 7), k=(2, 3, 2), stride=1, padding=1, dilation=1, mode="constant"),
        # multi-channel groups, padding modes, strides (the reference never tests these)
        dict(batch=2, cin=8, cout=8, groups=2, spatial=(300,), k=(9,), stride=1, padding=4, dilation=1, mode="constant"),
        dict(batch=3, cin=6, cout=4, groups=2, spatial=(100,), k=(7,), stride=3, padding=5, dilation=2, mode="reflect"),
        dict(batch=2, cin=4, cout=6, groups=1, spatial=(64,), k=(5,), stride=2, padding=3, dilation=1, mode="replicate"),
        dict(batch=2, cin=5, cout=3, groups=1, spatial=(50,), k=(4,), stride=1, padding=6, dilation=3, mode="circular"),
        dict(batch=2, cin=4, cout=4, groups=2, spatial=(20, 33), k=(3, 5), stride=(1, 2), padding=(2, 1), dilation=(2, 1), mode="constant"),
        dict(batch=1, cin=3, cout=5, groups=1, spatial=(31, 18), k=(4, 3), stride=(2, 1), padding=(1, 2), dilation=(1, 2), mode="reflect"),
        dict(batch=1, cin=4, cout=2, groups=2, spatial=(9, 12, 10), k=(2, 3, 4), stride=(1, 2, 1), padding=(1, 0, 2), dilation=(2, 1, 1), mode="constant"),
        dict(batch=2, cin=3, cout=3, groups=3, spatial=(8, 8, 8), k=(3, 3, 3), stride=1, padding=1, dilation=1, mode="replicate"),
        # several overlap-save tiles; the on-chip weight-gradient kernel (stride 1) and the plan fallback (stride 2)
        dict(batch=2, cin=8, cout=8, groups=1, spatial=(5000,), k=(129,), stride=1, padding=64, dilation=1, mode="constant"),
        dict(batch=2, cin=16, cout=24, groups=1, spatial=(3000,), k=(65,), stride=2, padding=32, dilation=1, mode="constant"),
        dict(batch=2, cin=16, cout=16, groups=16, spatial=(4000,), k=(33,), stride=1, padding=16, dilation=1, mode="circular"),
        # BASELINE.json configs at reduced batch / length (same channels, kernel, dilation, groups)
        dict(batch=1, cin=8, cout=8, groups=1, spatial=(32768,), k=(128,), stride=1, padding=0, dilation=1, mode="constant", name="cfg0"),
        dict(batch=2, cin=8, cout=8, groups=1, spatial=(32768,), k=(512,), stride=1, padding=0, dilation=1, mode="constant", name="cfgA_b2"),
        dict(batch=1, cin=8, cout=8, groups=1, spatial=(256, 256), k=(31, 31), stride=1, padding=0, dilation=1, mode="constant", name="cfgB_b1_256"),
        dict(batch=1, cin=8, cout=8, groups=1, spatial=(64, 64, 64), k=(9, 9, 9), stride=1, padding=0, dilation=1, mode="constant", name="cfgC_b1"),
        dict(batch=1, cin=64, cout=64, groups=8, spatial=(1 << 16,), k=(257,), stride=1, padding=0, dilation=4, mode="constant", name="cfgD_b1_64k"),
    ]
    tr = [
        dict(batch=2, cin=2, cout=3, groups=1, spatial=(7,), k=(2,), stride=1, padding=0, dilation=1, output_padding=0),
        dict(batch=2, cin=3, cout=3, groups=3, spatial=(8,), k=(3,), stride=3, padding=1, dilation=2, output_padding=1),
        dict(batch=2, cin=2, cout=2, groups=2, spatial=(8,), k=(3,), stride=4, padding=1, dilation=4, output_padding=2),
        dict(batch=2, cin=2, cout=2, groups=1, spatial=(7,), k=(2,), stride=2, padding=0, dilation=3, output_padding=2),    # output_padding >= stride
        dict(batch=2, cin=3, cout=2, groups=1, spatial=(7, 8), k=(2, 3), stride=(2, 3), padding=(1, 0), dilation=(1, 3), output_padding=(1, 2)),
        dict(batch=2, cin=2, cout=3, groups=1, spatial=(7, 8, 7), k=(3, 2, 3), stride=2, padding=1, dilation=2, output_padding=1),
        dict(batch=2, cin=6, cout=4, groups=2, spatial=(40, 37), k=(5, 4), stride=(2, 1), padding=(2, 1), dilation=(1, 2), output_padding=(1, 0)),
        dict(batch=1, cin=4, cout=6, groups=1, spatial=(9, 10, 8), k=(3, 2, 3), stride=(2, 1, 2), padding=(1, 0, 2), dilation=1, output_padding=(1, 0, 1)),
        dict(batch=2, cin=8, cout=8, groups=1, spatial=(3000,), k=(129,), stride=1, padding=64, dilation=1, output_padding=0),
        dict(batch=1, cin=8, cout=16, groups=2, spatial=(1000,), k=(33,), stride=3, padding=5, dilation=2, output_padding=2),
        dict(batch=2, cin=16, cout=8, groups=1, spatial=(2000,), k=(65,), stride=2, padding=3, dilation=1, output_padding=1),
        dict(batch=2, cin=3, cout=5, groups=1, spatial=(50,), k=(4,), stride=2, padding=7, dilation=3, output_padding=1),
    ]
    return [("fwd", c) for c in fwd] + [("tr", c) for c in tr]


G5_FULL_LIMIT = 20000      # tensors up to this many elements are stored whole, larger ones as 4096 samples + sum


def _pack(store, key, arr, seed):
    arr = np.ascontiguousarray(arr)
    store[f"{key}_shape"] = np.array(arr.shape, dtype=np.int64)
    store[f"{key}_absmax"] = np.array(float(np.abs(arr).max()))
    if arr.size <= G5_FULL_LIMIT:
        store[f"{key}_full"] = arr
    else:
        idx = np.random.default_rng(seed).integers(0, arr.size, 4096)
        store[f"{key}_idx"], store[f"{key}_samp"] = idx, arr.reshape(-1)[idx]
        store[f"{key}_sum"] = np.array(arr.astype(np.float64).sum())


def make_g5(ref):
    """Gradients from the REFERENCE's autograd graph (it has no custom backward: torch differentiates its
    rfftn / einsum / irfftn ops; the reference's tests pin dW and db the same way,
    tests/test_functional.py:62-117, tests/test_functional_transpose.py:73-124)."""
    import torch
    store = {}
    cases = g5_cases()
    for n, (kind, c) in enumerate(cases):
        seed = 11000 + 5 * n
        if kind == "fwd":
            wshape = (c["cout"], c["cin"] // c["groups"]) + tuple(c["k"])
        else:
            wshape = (c["cin"], c["cout"] // c["groups"]) + tuple(c["k"])
        x = torch.from_numpy(seeded(seed, c["batch"], c["cin"], *c["spatial"])).requires_grad_()
        w = torch.from_numpy(seeded(seed + 1, *wshape)).requires_grad_()
        b = torch.from_numpy(seeded(seed + 2, c["cout"])).requires_grad_()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if kind == "fwd":
                y = ref.functional.fft_conv(x, w, bias=b, stride=c["stride"], padding=c["padding"], dilation=c["dilation"],
                                            groups=c["groups"], padding_mode=c["mode"])
            else:
                y = ref.functional.fft_conv_transpose(x, w, bias=b, stride=c["stride"], padding=c["padding"],
                                                      output_padding=c["output_padding"], dilation=c["dilation"],
                                                      groups=c["groups"])
        gy = torch.from_numpy(seeded(seed + 3, *y.shape))
        y.backward(gy)
        meta = dict(c)
        meta.update(kind=kind, seed=seed, yshape=tuple(int(v) for v in y.shape))
        store[f"meta{n}"] = np.array(repr(meta))
        _pack(store, f"y{n}", y.detach().numpy(), seed + 10)
        _pack(store, f"dx{n}", x.grad.numpy(), seed + 11)
        _pack(store, f"dw{n}", w.grad.numpy(), seed + 12)
        _pack(store, f"db{n}", b.grad.numpy(), seed + 13)
        print("G5:", n, kind, c.get("name", ""), tuple(y.shape))
    store["count"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT_DIR, "g5_gradients.npz"), **store)
    print("G5:", len(cases), "cases")


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    ref = load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "g4":
        make_g4(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g5":
        make_g5(ref)
        return
    make_g4(ref)
    make_g5(ref)

    # ---- G1
    store = {}
    cases = g1_cases()
    for n, c in enumerate(cases):
        nd = c["ndim"]
        x, w, b = make_inputs(1000 + 3 * n, 2, c["cin"], c["cout"], c["groups"],
                              (c["size"],) * nd, (c["k"],) * nd)
        y = run_ref(ref, x, w, b, stride=c["stride"], padding=c["padding"],
                    dilation=c["dilation"], groups=c["groups"])
        store[f"x{n}"], store[f"w{n}"], store[f"b{n}"], store[f"y{n}"] = x, w, b, y
        store[f"p{n}"] = np.array([nd, c["size"], c["cin"], c["cout"], c["groups"], c["k"],
                                   c["padding"], c["stride"], c["dilation"]], dtype=np.int64)
    store["count"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT_DIR, "g1_reference_grid.npz"), **store)
    print("G1:", len(cases), "cases")

    # ---- G2
    store = {}
    cases = g2_cases()
    for n, c in enumerate(cases):
        x, w, b = make_inputs(5000 + 3 * n, c["batch"], c["cin"], c["cout"], c["groups"],
                              c["spatial"], c["k"])
        y = run_ref(ref, x, w, b, stride=c["stride"], padding=c["padding"],
                    dilation=c["dilation"], groups=c["groups"], padding_mode=c["mode"])
        big = x.size > 200000 or y.size > 200000
        if big:   # keep the seed, a checksum and samples only
            idx = np.random.default_rng(77 + n).integers(0, y.size, 4096)
            store[f"yi{n}"], store[f"ys{n}"] = idx, y.reshape(-1)[idx]
            store[f"ysum{n}"] = np.array(y.astype(np.float64).sum())
            store[f"xsum{n}"] = np.array(x.astype(np.float64).sum())
            store[f"yshape{n}"] = np.array(y.shape)
        else:
            store[f"x{n}"], store[f"w{n}"], store[f"b{n}"], store[f"y{n}"] = x, w, b, y
        nd = len(c["spatial"])
        meta = dict(c)
        meta["seed"] = 5000 + 3 * n
        meta["big"] = bool(big)
        store[f"meta{n}"] = np.array(repr(meta))
    store["count"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT_DIR, "g2_extended.npz"), **store)
    print("G2:", len(cases), "cases")

    # ---- G3
    store = {}
    for name, (batch, cin, cout, groups, spatial, ksize, dil) in BASELINE_CONFIGS.items():
        seed = 9000 + sum(map(ord, name))
        x, w, b = make_inputs(seed, batch, cin, cout, groups, spatial, ksize)
        y = run_ref(ref, x, w, b, dilation=dil, groups=groups)
        idx = np.random.default_rng(seed + 5).integers(0, y.size, 4096)
        store[f"{name}_seed"] = np.array(seed)
        store[f"{name}_xsum"] = np.array(x.astype(np.float64).sum())
        store[f"{name}_x64"] = x.reshape(-1)[:64].copy()
        store[f"{name}_yshape"] = np.array(y.shape)
        store[f"{name}_yidx"] = idx
        store[f"{name}_ysamp"] = y.reshape(-1)[idx]
        store[f"{name}_ysum"] = np.array(y.astype(np.float64).sum())
        store[f"{name}_yabsmax"] = np.array(np.abs(y).max())
        print("G3:", name, y.shape, "absmax", float(np.abs(y).max()))
        del x, y
    np.savez_compressed(os.path.join(OUT_DIR, "g3_baseline_configs.npz"), **store)


if __name__ == "__main__":
    main()
