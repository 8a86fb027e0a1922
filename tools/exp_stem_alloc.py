import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
from bnn_amd import hipops, native
from tests.golden import gen
dev = torch.device("cuda:0")
lib = native.require()
N = 256
x = torch.from_numpy(gen.normal(1, (8, 3, 224, 224))).to(dev).repeat(N // 8, 1, 1, 1)
w = torch.from_numpy(gen.conv_weight("kaiming", 3, (64, 3, 7, 7))).to(dev)
a = torch.rand(64, device=dev) + 0.5; b = torch.randn(64, device=dev) * 0.3
y = torch.empty((N, 64, 56, 56), device=dev)
P = torch.empty((N, 1, 56, 56), dtype=torch.int64, device=dev); M = torch.empty_like(P)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(yp, pp, mp):
    rc = lib.bnn_hip_stem7x7_bn_relu_pool_pack_f32(x.data_ptr(), w.data_ptr(), a.data_ptr(), b.data_ptr(), N, 224, 224, 0,
                                                   yp, pp, mp, st)
    assert rc == 0
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("static full      %.1f" % t(lambda: run(y.data_ptr(), P.data_ptr(), M.data_ptr())))
print("static f32-only  %.1f" % t(lambda: run(y.data_ptr(), None, None)))
print("static pack-only %.1f" % t(lambda: run(None, P.data_ptr(), M.data_ptr())))
print("static full      %.1f" % t(lambda: run(y.data_ptr(), P.data_ptr(), M.data_ptr())))
print("hipops full      %.1f" % t(lambda: hipops.stem7x7(x, w, a, b)))
print("hipops f32-only  %.1f" % t(lambda: hipops.stem7x7(x, w, a, b, out_packed=False)))
