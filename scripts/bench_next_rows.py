#!/usr/bin/env python3
"""Timings for the rows next to the headline path: transposed convolution and the
backward pass, at the cfgA shape, next to two same-GPU yardsticks: the reference's algorithm written with
torch.fft (rocFFT underneath; functional.py:60-87 restated inline) and torch's direct convolution (MIOpen).
Prints one JSON line per measurement.  Not part of bench.py's contract."""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_conv_pytorch_amd as fca  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--ch", type=int, default=8)
ap.add_argument("--len", type=int, default=32768)
ap.add_argument("--k", type=int, default=512)
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
dev = torch.device("cuda", 0)
B, C, L, K = args.batch, args.ch, args.len, args.k
torch.manual_seed(0)
nbuf = 9
xs = [torch.randn(B, C, L, device=dev) for _ in range(nbuf)]
w = torch.randn(C, C, K, device=dev) / (C * K) ** 0.5
b = torch.randn(C, device=dev)


def rfft_conv(x, w, b):
    """the reference's op sequence: pad, rfft both, conj-multiply-sum over channels, irfft, crop, bias"""
    n = x.shape[-1]
    xf = torch.fft.rfft(x, n=n)
    wf = torch.fft.rfft(w, n=n)
    yf = torch.einsum("bif,oif->bof", xf, wf.conj())
    y = torch.fft.irfft(yf, n=n)[..., : n - w.shape[-1] + 1]
    return y + b.view(1, -1, 1)


def timed(fn, iters):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters   # us


def report(name, us, n_out, alg_bytes):
    print(json.dumps({"measurement": name, "us_per_call": round(us, 1), "GSamples_per_s": round(n_out / us / 1e3, 1),
                      "algorithmic_GB_per_s": round(alg_bytes / us / 1e3, 1), "frac_of_8TBps": round(alg_bytes / us / 1e3 / 8000, 4)}),
          flush=True)


Lout = L - K + 1
n_out = B * C * Lout
fwd_bytes = 4 * (B * C * L + C * C * K + C + n_out)

conv = fca.FFTConv1d(C, C, K, bias=True).to(dev)
with torch.no_grad():
    conv.weight.copy_(w)
    conv.bias.copy_(b)

with torch.no_grad():
    report("forward, this library (module, eager launches)", timed(lambda i: conv(xs[i % nbuf]), args.iters), n_out, fwd_bytes)
    report("forward, reference algorithm via torch.fft (rocFFT), same GPU", timed(lambda i: rfft_conv(xs[i % nbuf], w, b), args.iters), n_out, fwd_bytes)
    try:
        report("forward, torch F.conv1d (MIOpen direct), same GPU", timed(lambda i: F.conv1d(xs[i % nbuf], w, b), max(3, args.iters // 10)), n_out, fwd_bytes)
    except Exception as exc:  # noqa: BLE001
        print(json.dumps({"measurement": "forward, torch F.conv1d", "error": str(exc)[:200]}))

# transposed convolution (row N2): (B, C, Lout) -> (B, C, L)
tconv = fca.FFTConvTranspose1d(C, C, K, bias=True).to(dev)
gs = [torch.randn(B, C, Lout, device=dev) for _ in range(nbuf)]
with torch.no_grad():
    t_bytes = 4 * (B * C * Lout + C * C * K + C + B * C * L)
    report("transposed forward, this library", timed(lambda i: tconv(gs[i % nbuf]), args.iters), B * C * L, t_bytes)
    wt = tconv.weight.detach()
    report("transposed forward, torch F.conv_transpose1d (MIOpen), same GPU",
           timed(lambda i: F.conv_transpose1d(gs[i % nbuf], wt, tconv.bias), max(3, args.iters // 10)), B * C * L, t_bytes)

# backward (row N1): forward + dX + dW + db
xr = [x.clone().requires_grad_(True) for x in xs[:3]]
bwd_bytes = fwd_bytes + 4 * (n_out + B * C * L + C * C * K + C) + 4 * (B * C * L + n_out)   # + read dY, X; write dX, dW, db


def ours_step(i):
    x = xr[i % 3]
    x.grad = None
    conv.zero_grad(set_to_none=True)
    conv(x).sum().backward()


wr = w.clone().requires_grad_(True)
br = b.clone().requires_grad_(True)


def ref_step(i):
    x = xr[i % 3]
    x.grad = None
    wr.grad = None
    br.grad = None
    rfft_conv(x, wr, br).sum().backward()


report("forward + backward, this library (autograd.Function)", timed(ours_step, args.iters), n_out, bwd_bytes)
report("forward + backward, reference algorithm via torch.fft autograd, same GPU", timed(ref_step, args.iters), n_out, bwd_bytes)

# the same with a materialised output gradient (what a following layer hands back) instead of .sum(): without the harness's
# loss reduction and the copy that makes its broadcast gradient contiguous
gys = [torch.randn(B, C, Lout, device=dev) for _ in range(3)]


def ours_step_gy(i):
    x = xr[i % 3]
    x.grad = None
    conv.zero_grad(set_to_none=True)
    conv(x).backward(gys[i % 3])


def ref_step_gy(i):
    x = xr[i % 3]
    x.grad = None
    wr.grad = None
    br.grad = None
    rfft_conv(x, wr, br).backward(gys[i % 3])


report("forward + backward from a given output gradient, this library", timed(ours_step_gy, args.iters), n_out, bwd_bytes)
report("forward + backward from a given output gradient, reference algorithm via torch.fft autograd, same GPU",
       timed(ref_step_gy, args.iters), n_out, bwd_bytes)

# transposed convolution forward + backward (differentiable since round 2): dX is a forward plan, dW the weight gradient
# with the roles of signal and gradient swapped
gr = [g.clone().requires_grad_(True) for g in gs[:3]]


def ours_t_step(i):
    g = gr[i % 3]
    g.grad = None
    tconv.zero_grad(set_to_none=True)
    tconv(g).sum().backward()


wtr = tconv.weight.detach().clone().requires_grad_(True)
btr = tconv.bias.detach().clone().requires_grad_(True)


def torch_t_step(i):
    g = gr[i % 3]
    g.grad = None
    wtr.grad = None
    btr.grad = None
    F.conv_transpose1d(g, wtr, btr).sum().backward()


t_bwd_bytes = t_bytes + 4 * (B * C * L + B * C * Lout + C * C * K + C) + 4 * (B * C * Lout + B * C * L)
report("transposed forward + backward, this library", timed(ours_t_step, args.iters), B * C * L, t_bwd_bytes)
report("transposed forward + backward, torch conv_transpose1d autograd (MIOpen), same GPU",
       timed(torch_t_step, max(3, args.iters // 10)), B * C * L, t_bwd_bytes)

# 2-D forward + backward (weight gradient by fc_wgrad_nd: the batch / channel-swapped convolution run by the library on the tensors as they lie)
B2, S2, K2 = 4, 256, 15
conv2 = fca.FFTConv2d(C, C, K2, bias=True).to(dev)
x2 = [torch.randn(B2, C, S2, S2, device=dev, requires_grad=True) for _ in range(3)]
o2 = S2 - K2 + 1
n2 = B2 * C * o2 * o2
b2_bytes = 4 * (B2 * C * S2 * S2 + C * C * K2 * K2 + C + n2) * 2 + 4 * (B2 * C * S2 * S2 + n2 + C * C * K2 * K2 + C)


def ours_2d_step(i):
    x = x2[i % 3]
    x.grad = None
    conv2.zero_grad(set_to_none=True)
    conv2(x).sum().backward()


w2r = conv2.weight.detach().clone().requires_grad_(True)
b2r = conv2.bias.detach().clone().requires_grad_(True)


def torch_2d_step(i):
    x = x2[i % 3]
    x.grad = None
    w2r.grad = None
    b2r.grad = None
    F.conv2d(x, w2r, b2r).sum().backward()


report(f"2-D forward + backward B{B2} {C}->{C} {S2}x{S2} k{K2}, this library", timed(ours_2d_step, args.iters), n2, b2_bytes)
report(f"2-D forward + backward B{B2} {C}->{C} {S2}x{S2} k{K2}, torch conv2d autograd (MIOpen), same GPU",
       timed(torch_2d_step, max(3, args.iters // 10)), n2, b2_bytes)
