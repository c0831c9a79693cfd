import sys, os, torch
sys.path.insert(0, ''+os.environ.get('GRAFT_REPO_ROOT','/root/repo')+''); sys.path.insert(0, '/root/repo/tests')
import test_gpu_fuzz as tf
from fft_conv_pytorch_amd.functional import fft_conv
c={'ndim': 3, 'batch': 3, 'cin': 3, 'cout': 8, 'groups': 1, 'k': 1, 'dil': 1, 'size': [1, 18, 10], 'stride': 2, 'pad': 1, 'mode': 'constant'}
gen = torch.Generator().manual_seed(5)
x = torch.randn(c["batch"], c["cin"], *c["size"], generator=gen, dtype=torch.float64)
w = torch.randn(c["cout"], c["cin"] // c["groups"], *([c["k"]] * 3), generator=gen, dtype=torch.float64)
b = torch.randn(c["cout"], generator=gen, dtype=torch.float64)
xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
want = tf._reference(c, xr, wr, br)
xd, wd, bd = (t.float().to("cuda").requires_grad_(True) for t in (x, w, b))
got = fft_conv(xd, wd, bias=bd, stride=c["stride"], padding=c["pad"], dilation=c["dil"], groups=c["groups"], padding_mode=c["mode"])
gy = torch.randn(want.shape, generator=gen, dtype=torch.float64)
want.backward(gy); got.backward(gy.float().to("cuda"))
print('env', os.environ.get('FFTCONV_DENSE'), 'y', tf._rel(got.detach(), want.detach()), 'dx', tf._rel(xd.grad, xr.grad), 'dw', tf._rel(wd.grad, wr.grad), 'db', tf._rel(bd.grad, br.grad))
print(xd.grad.shape, (xd.grad.cpu().double()-xr.grad).abs().max().item(), xr.grad.abs().max().item())
