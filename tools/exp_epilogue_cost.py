"""Where the conv2-type (BN + residual + ReLU -> fp32 + packed) kernels of layer1/layer2 spend their time: the same
convolution with pieces of the epilogue switched off (runtime-flag kernel variant unless the combination has a
compiled profile)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import numpy as np, torch
from bnn_amd import hipops
from tests.golden import gen
dev = torch.device("cuda:0")
def t(fn, n=100):
    for _ in range(30): fn()  # also keeps the clocks up (the first timed region after idle runs ~15 % slow)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (C, HW) in ((64, 56), (128, 28), (256, 14), (512, 7)):
    N = 256
    x = torch.from_numpy(gen.activation("relu", 7, (8, C, HW, HW))).to(dev).repeat(N // 8, 1, 1, 1)
    w = torch.from_numpy(gen.conv_weight("kaiming", 8, (C, C, 3, 3))).to(dev)
    act = hipops.pack_act(x); act.nonneg = True
    pw = hipops.pack_weight(w)
    res = torch.randn(N, C, HW, HW, device=dev)
    a = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev) * 0.3
    kw = dict(bn_scale=a, bn_shift=b, relu=True, stride=1, padding=1)
    rows = [
        ("OUT  bn+res+relu -> f32+pack", dict(residual=res, out_f32=True, out_packed=True)),
        ("LAST bn+res+relu -> f32     ", dict(residual=res, out_f32=True, out_packed=False)),
        ("     bn+relu     -> f32+pack", dict(out_f32=True, out_packed=True)),
        ("     bn+res+relu -> pack    ", dict(residual=res, out_f32=False, out_packed=True)),
        ("MID  bn+relu     -> pack    ", dict(out_f32=False, out_packed=True)),
    ]
    ops = 2.0 * (C * 9 // 32) * N * C * HW * HW
    print("== %d ch %dx%d  (int-ALU floor %.0f us; residual or fp32 tensor %.0f MB)" % (C, HW, HW, ops / 39.3e12 * 1e6, N * C * HW * HW * 4 / 1e6))
    for name, k in rows:
        us = t(lambda: hipops.bconv2d_fused(act, pw, **kw, **k))
        print("  %s %7.1f us" % (name, us))
