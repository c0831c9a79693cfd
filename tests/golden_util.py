"""Helpers to read the committed golden fixtures (tests/golden/*.npz)."""
import ast
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def seeded(seed, *shape):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def make_inputs(seed, batch, cin, cout, groups, spatial, ksize):
    """Same generator as oracle/make_golden.py (inputs are re-derived from the seed)."""
    x = seeded(seed, batch, cin, *spatial)
    w = seeded(seed + 1, cout, cin // groups, *ksize)
    b = seeded(seed + 2, cout)
    return x, w, b


def g1_cases():
    z = np.load(os.path.join(GOLDEN, "g1_reference_grid.npz"))
    for n in range(int(z["count"])):
        nd, size, cin, cout, groups, k, pad, stride, dil = (int(v) for v in z[f"p{n}"])
        kw = dict(stride=stride, padding=pad, dilation=dil, groups=groups)
        yield n, z[f"x{n}"], z[f"w{n}"], z[f"b{n}"], kw, z[f"y{n}"]


def g2_cases():
    z = np.load(os.path.join(GOLDEN, "g2_extended.npz"))
    for n in range(int(z["count"])):
        meta = ast.literal_eval(str(z[f"meta{n}"]))
        kw = dict(stride=meta["stride"], padding=meta["padding"], dilation=meta["dilation"],
                  groups=meta["groups"], padding_mode=meta["mode"])
        if meta["big"]:
            x, w, b = make_inputs(meta["seed"], meta["batch"], meta["cin"], meta["cout"],
                                  meta["groups"], meta["spatial"], meta["k"])
            assert abs(float(x.astype(np.float64).sum()) - float(z[f"xsum{n}"])) < 1e-6 * x.size
            yield n, x, w, b, kw, dict(idx=z[f"yi{n}"], samples=z[f"ys{n}"], total=float(z[f"ysum{n}"]),
                                       shape=tuple(int(v) for v in z[f"yshape{n}"]))
        else:
            yield n, z[f"x{n}"], z[f"w{n}"], z[f"b{n}"], kw, z[f"y{n}"]


def g4_cases():
    z = np.load(os.path.join(GOLDEN, "g4_transpose.npz"))
    for n in range(int(z["count"])):
        meta = ast.literal_eval(str(z[f"meta{n}"]))
        kw = dict(stride=meta["stride"], padding=meta["padding"], output_padding=meta["output_padding"],
                  dilation=meta["dilation"], groups=meta["groups"])
        yield n, z[f"x{n}"], z[f"w{n}"], z[f"b{n}"], kw, z[f"y{n}"]


def _unpack(z, key):
    """Golden entry written by oracle/make_golden.py:_pack -- a full array, or samples + sum of a large one."""
    shape = tuple(int(v) for v in z[f"{key}_shape"])
    absmax = float(z[f"{key}_absmax"])
    if f"{key}_full" in z.files:
        return dict(full=z[f"{key}_full"], shape=shape, absmax=absmax)
    return dict(idx=z[f"{key}_idx"], samples=z[f"{key}_samp"], total=float(z[f"{key}_sum"]), shape=shape, absmax=absmax)


def g5_cases():
    """Gradient fixtures from the reference's own autograd: yields (n, kind, meta, x, w, b, gy, gold) with
    gold = dict(y=..., dx=..., dw=..., db=...) of _unpack entries; inputs are re-derived from the seed."""
    z = np.load(os.path.join(GOLDEN, "g5_gradients.npz"))
    for n in range(int(z["count"])):
        meta = ast.literal_eval(str(z[f"meta{n}"]))
        kind, seed = meta["kind"], meta["seed"]
        if kind == "fwd":
            wshape = (meta["cout"], meta["cin"] // meta["groups"]) + tuple(meta["k"])
        else:
            wshape = (meta["cin"], meta["cout"] // meta["groups"]) + tuple(meta["k"])
        x = seeded(seed, meta["batch"], meta["cin"], *meta["spatial"])
        w = seeded(seed + 1, *wshape)
        b = seeded(seed + 2, meta["cout"])
        gy = seeded(seed + 3, *meta["yshape"])
        gold = {k: _unpack(z, f"{k}{n}") for k in ("y", "dx", "dw", "db")}
        yield n, kind, meta, x, w, b, gy, gold


def check_entry(arr, gold, tol):
    """Compare an array with an _unpack entry; error is relative to the golden tensor's max magnitude."""
    arr = np.asarray(arr)
    assert tuple(arr.shape) == gold["shape"], (arr.shape, gold["shape"])
    scale = max(gold["absmax"], 1e-30)
    if "full" in gold:
        err = float(np.abs(arr.astype(np.float64) - gold["full"].astype(np.float64)).max() / scale)
    else:
        got = arr.reshape(-1)[gold["idx"]].astype(np.float64)
        err = float(np.abs(got - gold["samples"].astype(np.float64)).max() / scale)
        tot = float(arr.astype(np.float64).sum())
        assert abs(tot - gold["total"]) <= 1e-4 * scale * np.sqrt(arr.size) + 1e-3 * abs(gold["total"])
    assert err <= tol, f"rel err {err:.3e} > {tol}"
    return err


def g5_kwargs(kind, meta):
    if kind == "fwd":
        return dict(stride=meta["stride"], padding=meta["padding"], dilation=meta["dilation"], groups=meta["groups"],
                    padding_mode=meta["mode"])
    return dict(stride=meta["stride"], padding=meta["padding"], output_padding=meta["output_padding"],
                dilation=meta["dilation"], groups=meta["groups"])


BASELINE_CONFIGS = {
    "cfg0": (1, 8, 8, 1, (32768,), (128,), 1),
    "cfgA": (32, 8, 8, 1, (32768,), (512,), 1),
    "cfgB": (16, 8, 8, 1, (512, 512), (31, 31), 1),
    "cfgC": (8, 8, 8, 1, (64, 64, 64), (9, 9, 9), 1),
    "cfgD_b1": (1, 64, 64, 8, (1 << 20,), (257,), 4),
}


def g3_case(name):
    z = np.load(os.path.join(GOLDEN, "g3_baseline_configs.npz"))
    batch, cin, cout, groups, spatial, ksize, dil = BASELINE_CONFIGS[name]
    seed = int(z[f"{name}_seed"])
    x, w, b = make_inputs(seed, batch, cin, cout, groups, spatial, ksize)
    assert np.array_equal(x.reshape(-1)[:64], z[f"{name}_x64"]), "numpy RNG stream drifted"
    kw = dict(dilation=dil, groups=groups)
    gold = dict(idx=z[f"{name}_yidx"], samples=z[f"{name}_ysamp"], total=float(z[f"{name}_ysum"]),
                shape=tuple(int(v) for v in z[f"{name}_yshape"]), absmax=float(z[f"{name}_yabsmax"]))
    return x, w, b, kw, gold


def check_against(y, gold, tol):
    """Compare a full output array with a golden entry (full array or samples)."""
    y = np.asarray(y)
    if isinstance(gold, dict):
        assert tuple(y.shape) == gold["shape"], (y.shape, gold["shape"])
        got = y.reshape(-1)[gold["idx"]].astype(np.float64)
        ref = gold["samples"].astype(np.float64)
        scale = gold.get("absmax", np.abs(ref).max())
        err = np.abs(got - ref).max() / scale
        assert err <= tol, f"sampled rel err {err:.3e} > {tol}"
        # checksum: loose (fp32 accumulation of ~1e7..1e9 terms), catches gross holes
        tot = float(y.astype(np.float64).sum())
        assert abs(tot - gold["total"]) <= 1e-4 * scale * np.sqrt(y.size) + 1e-3 * abs(gold["total"])
        return err
    assert y.shape == gold.shape, (y.shape, gold.shape)
    scale = max(np.abs(gold).max(), 1e-30)
    err = np.abs(y.astype(np.float64) - gold.astype(np.float64)).max() / scale
    assert err <= tol, f"rel err {err:.3e} > {tol}"
    return err
