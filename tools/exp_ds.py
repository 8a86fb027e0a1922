import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
from bnn_amd import hipops
from tests.golden import gen
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (C, H, O) in [(64, 28, 128), (128, 14, 256), (256, 7, 512)]:
    x = torch.from_numpy(gen.activation("relu", 7, (4, C, H, H))).to(dev).repeat(64, 1, 1, 1)
    w = torch.from_numpy(gen.conv_weight("kaiming", 8, (O, C, 1, 1))).to(dev)
    a = torch.rand(O, device=dev) + 0.5; b = torch.randn(O, device=dev)
    pw, act = hipops.pack_weight(w), hipops.pack_act(x)
    print(C, H, O, "plain %.1f us" % t(lambda: hipops.bconv2d(act, pw)),
          "bn %.1f us" % t(lambda: hipops.bconv2d_fused(act, pw, bn_scale=a, bn_shift=b)),
          "bn+relu+pack %.1f us" % t(lambda: hipops.bconv2d_fused(act, pw, bn_scale=a, bn_shift=b, relu=True, out_packed=True)),
          "alloc-only %.1f us" % t(lambda: torch.empty((256, O, H, H), device=dev)))
