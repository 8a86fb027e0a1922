import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
from bnn_amd import hipops
from tests.golden import gen
dev = torch.device("cuda:0")
def run(name, N, C, H, W, O, k, s, p, **kw):
    x = torch.from_numpy(gen.activation("relu", 7, (4, C, H, W))).to(dev).repeat(N // 4, 1, 1, 1)
    w = torch.from_numpy(gen.conv_weight("kaiming", 8, (O, C, k, k))).to(dev)
    pw, act = hipops.pack_weight(w), hipops.pack_act(x)
    for _ in range(3): out = hipops.bconv2d(act, pw, stride=s, padding=p, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): out = hipops.bconv2d(act, pw, stride=s, padding=p, **kw)
    e1.record(); torch.cuda.synchronize()
    print(name, N, kw, round(e0.elapsed_time(e1) * 100, 1), "us")
for N in (64, 256, 1024):
    for kw in (dict(weights="sgpr"), dict(weights="lds"), dict(force_generic=True)):
        run("l4_512x7", N, 512, 7, 7, 512, 3, 1, 1, **kw)
for O in (32, 128, 512):
    run("C512_O%d" % O, 256, 512, 7, 7, O, 3, 1, 1, weights="sgpr")
for C in (128, 256, 512):
    run("C%d_O512" % C, 256, C, 7, 7, 512, 3, 1, 1, weights="sgpr")
for HW in (7, 14, 28):
    run("C512_HW%d" % HW, 1024 * 49 // (HW * HW) // 4 * 4 or 4, 512, HW, HW, 512, 3, 1, 1, weights="sgpr")
