#!/usr/bin/env python3
"""Time one binary conv shape given on the command line:  exp_shape.py C H W O k stride pad N [weights]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
from bnn_amd import hipops, native
from tests.golden import gen

C, H, W, O, k, s, p, N = map(int, sys.argv[1:9])
wsel = sys.argv[9] if len(sys.argv) > 9 else None
dev = torch.device("cuda:0")
info = native.device_info(0)
peak = info["compute_units"] * 64 * info["clock_khz"] * 1e3
x = torch.from_numpy(gen.activation("relu", 7, (4, C, H, W))).to(dev).repeat(N // 4, 1, 1, 1)
w = torch.from_numpy(gen.conv_weight("kaiming", 8, (O, C, k, k))).to(dev)
pw, act = hipops.pack_weight(w), hipops.pack_act(x)
for _ in range(3):
    out = hipops.bconv2d(act, pw, stride=s, padding=p, weights=wsel)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20):
    out = hipops.bconv2d(act, pw, stride=s, padding=p, weights=wsel)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
ops = 2.0 * ((C * k * k + 31) // 32) * out.numel()
print("C%d %dx%d O%d k%d s%d N%d %s: %.1f us  frac %.3f" % (C, H, W, O, k, s, N, wsel or "", us, ops / (us * 1e-6) / peak))
