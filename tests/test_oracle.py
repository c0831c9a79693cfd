"""The oracle (oracle/fft_conv_oracle.py) against the golden vectors produced by the real
reference (oracle/make_golden.py) and against an FFT-free float64 direct convolution."""
import numpy as np
import pytest
import torch

from oracle import fft_conv_oracle as orc
from tests import golden_util as gu

REL_TOL = 1e-4      # north_star: outputs within 1e-4 rel fp32
TIGHT = 2e-5        # oracle backends vs the reference itself: same algorithm, fp32


def test_ntuple_matches_reference_semantics():
    assert orc.ntuple(3, 2) == (3, 3)
    assert orc.ntuple([1, 2, 3], 3) == (1, 2, 3)
    with pytest.raises(ValueError, match="Cannot cast tuple of length 2 to length 3."):
        orc.ntuple((1, 2), 3)
    with pytest.raises(ValueError, match="Cannot cast tuple of length 4 to length 1."):
        orc.ntuple("same", 1)   # a str is Iterable: reference quirk


def test_oracle_backends_match_reference_grid_g1():
    worst = 0.0
    for n, x, w, b, kw, y_ref in gu.g1_cases():
        y_t = orc.fft_conv_oracle_torch(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), **kw).numpy()
        y_n = orc.fft_conv_oracle_numpy(x, w, b, **kw)
        y_d = orc.direct_conv_float64(x, w, b, **kw)
        worst = max(worst, gu.check_against(y_t, y_ref, TIGHT), gu.check_against(y_n, y_ref, TIGHT),
                    gu.check_against(y_d, y_ref, TIGHT))
        # the reference's own absolute tolerance (benchmark_utils.py:53-57) holds at these sizes
        assert np.abs(y_t - y_d).max() < 1e-4 and np.abs(y_t - y_d).mean() < 5e-5
    assert worst < TIGHT


def test_oracle_backends_match_extended_g2():
    for n, x, w, b, kw, gold in gu.g2_cases():
        y_t = orc.fft_conv_oracle_torch(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), **kw).numpy()
        gu.check_against(y_t, gold, TIGHT)
        if x.size < 50000:
            gu.check_against(orc.fft_conv_oracle_numpy(x, w, b, **kw), gold, TIGHT)
            gu.check_against(orc.direct_conv_float64(x, w, b, **kw), gold, TIGHT)


@pytest.mark.parametrize("name", ["cfg0", "cfgA"])
def test_oracle_matches_baseline_config_samples_g3(name):
    x, w, b, kw, gold = gu.g3_case(name)
    y = orc.fft_conv_oracle_torch(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), **kw).numpy()
    gu.check_against(y, gold, TIGHT)


def test_any_longer_fft_gives_same_valid_outputs():
    """Freedom (1) of SURVEY section 0: padding the signal with trailing zeros (= a longer FFT)
    leaves the valid outputs unchanged -- the HIP path relies on this (power-of-two tiles)."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 3, 50), dtype=np.float32)
    w = rng.standard_normal((4, 3, 7), dtype=np.float32)
    y = orc.fft_conv_oracle_numpy(x, w)
    x_long = np.concatenate([x, np.zeros((2, 3, 78), np.float32)], axis=-1)
    y_long = orc.fft_conv_oracle_numpy(x_long, w)[..., : y.shape[-1]]
    assert orc.rel_err(y_long, y) < 1e-5


def test_output_extent_formula():
    for size, k, s, p, d in [(7, 2, 1, 0, 1), (8, 3, 2, 1, 2), (32768, 512, 1, 0, 1), (100, 7, 3, 5, 2)]:
        y = torch.nn.functional.conv1d(torch.zeros(1, 1, size), torch.zeros(1, 1, k), stride=s, padding=p, dilation=d)
        assert orc.output_extent(size, k, s, p, d) == y.shape[-1]


def test_transpose_oracle_matches_reference_g4():
    """Row N2: the transposed-convolution restatement against the real reference's outputs and against
    torch's direct conv_transpose (the ground truth of tests/test_functional_transpose.py:62-66)."""
    import torch.nn.functional as F
    count = 0
    for n, x, w, b, kw, y_ref in gu.g4_cases():
        xt, wt, bt = torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b)
        y = orc.fft_conv_transpose_oracle_torch(xt, wt, bt, **kw).numpy()
        gu.check_against(y, y_ref, TIGHT)
        y_direct = getattr(F, f"conv_transpose{x.ndim - 2}d")(xt, wt, bt, **kw).numpy()
        gu.check_against(y_direct, y_ref, TIGHT)
        count += 1
    assert count > 40


def test_oracle_autograd_matches_reference_gradients_g5():
    """G5: autograd through the oracle's restated op sequence gives the
    reference's own dX / dW / db (the reference has no custom backward either), forward and transposed."""
    count = {"fwd": 0, "tr": 0}
    for n, kind, meta, x, w, b, gy, gold in gu.g5_cases():
        if x.size > 600000:
            continue        # the reduced BASELINE configs are checked on the GPU path; keep the CPU suite quick
        xt = torch.from_numpy(x).requires_grad_()
        wt = torch.from_numpy(w).requires_grad_()
        bt = torch.from_numpy(b).requires_grad_()
        fn = orc.fft_conv_oracle_torch if kind == "fwd" else orc.fft_conv_transpose_oracle_torch
        y = fn(xt, wt, bt, **gu.g5_kwargs(kind, meta))
        y.backward(torch.from_numpy(gy))
        gu.check_entry(y.detach().numpy(), gold["y"], TIGHT)
        gu.check_entry(xt.grad.numpy(), gold["dx"], TIGHT)
        gu.check_entry(wt.grad.numpy(), gold["dw"], TIGHT)
        gu.check_entry(bt.grad.numpy(), gold["db"], TIGHT)
        count[kind] += 1
    assert count["fwd"] >= 15 and count["tr"] >= 12


def test_odd_frequency_real_axis_gives_same_valid_outputs():
    """The scheme of the HIP row passes (csrc/nd_passes.hpp, rows_r2c / rows_c2r), restated in numpy: samples turned by
    e^{-i*pi*n/T} before a complex FFT (bin k at frequency k + 1/2), two real rows per transform unpacked with the partner
    bin T-1-k, T/2 bins kept per row, the product with the conjugated kernel bins, the mirrored pack, the inverse FFT and the
    turn back.  The result is the negacyclic correlation: identical to the reference's rfft/irfft formulation
    (/root/reference/fft_conv_pytorch/functional.py:66-82) on the valid window, which is all that overlap-save keeps."""
    rng = np.random.default_rng(7)
    T, K = 64, 9
    n = np.arange(T)
    turn = np.exp(-1j * np.pi * n / T)
    rows = rng.standard_normal((2, T))
    taps = np.zeros(T)
    taps[:K] = rng.standard_normal(K)

    def half_bins_of_two_rows(a, b):
        z = np.fft.fft((a + 1j * b) * turn)
        zp = np.conj(z[T - 1 - np.arange(T // 2)])            # partner of bin k: T-1-k
        zk = z[: T // 2]
        return 0.5 * (zk + zp), -0.5j * (zk - zp)             # odd-frequency bins k < T/2 of row a and of row b

    xa, xb = half_bins_of_two_rows(rows[0], rows[1])
    ha, _ = half_bins_of_two_rows(taps, np.zeros(T))
    # each row's own spectrum really is the odd-frequency DFT of that row
    direct = np.array([np.sum(rows[0] * np.exp(-2j * np.pi * n * (k + 0.5) / T)) for k in range(T // 2)])
    assert np.allclose(xa, direct, atol=1e-10)
    ya, yb = xa * np.conj(ha), xb * np.conj(ha)               # correlation, as the reference's conj(kernel_fr)
    v = np.zeros(T, dtype=complex)
    v[: T // 2] = ya + 1j * yb
    v[T - 1 - np.arange(T // 2)] = np.conj(ya) + 1j * np.conj(yb)
    out = np.fft.ifft(v) * np.conj(turn)
    want = np.stack([np.correlate(r, taps[:K], mode="valid") for r in rows])        # T - K + 1 valid samples
    assert np.allclose(out.real[: T - K + 1], want[0], atol=1e-10)
    assert np.allclose(out.imag[: T - K + 1], want[1], atol=1e-10)
    # ... and the wrapped samples are the negated wrap-around (negacyclic), not the cyclic one: never kept
    cyc = np.fft.irfft(np.fft.rfft(rows[0]) * np.conj(np.fft.rfft(taps)), T)
    assert np.allclose(cyc[: T - K + 1], want[0], atol=1e-10)
    assert not np.allclose(cyc[T - K + 1:], out.real[T - K + 1:], atol=1e-6)
