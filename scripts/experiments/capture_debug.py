import sys, os
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from fft_conv_pytorch_amd import FFTConv2d, FFTConv1d, _native
DEV="cuda:0"
torch.manual_seed(0)
cold = FFTConv2d(8, 8, 5, bias=False).to(DEV).eval()
x2 = torch.randn(2, 8, 300, 700, device=DEV)
g2 = torch.cuda.CUDAGraph()
with torch.no_grad():
    try:
        with torch.cuda.graph(g2):
            cold(x2)
        print("capture succeeded")
    except Exception as exc:
        print("EXC:", type(exc).__name__, str(exc)[:400])
        c = exc.__context__
        while c is not None:
            print("  CTX:", type(c).__name__, str(c)[:400]); c = c.__context__
    torch.cuda.synchronize()
    print("cache keys", list(cold.__dict__.keys())[-3:])
    out = cold(x2)
    ref = F.conv2d(x2.double(), cold.weight.double())
    print("after: rel", float((out.double()-ref).abs().max()/ref.abs().max()))
    cold.invalidate_kernel_spectrum()
    out = cold(x2)
    print("after invalidate: rel", float((out.double()-ref).abs().max()/ref.abs().max()))
    _native.clear_plan_cache()
    cold.__dict__.pop("_last_plan", None); cold.invalidate_kernel_spectrum()
    out = cold(x2)
    print("after plan cache clear: rel", float((out.double()-ref).abs().max()/ref.abs().max()))
