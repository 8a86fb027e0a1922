"""Does a convolution of ANOTHER stream make progress beside the stem?  Stream 0: the stem kernel N times; stream 1: the
64-channel conv1-type kernel (42 VGPRs: the only one that fits beside a 230-VGPR stem) M times; alone and together.
BNN_AMD_LIB selects the library (default stem: 246 VGPRs x 2 waves = the whole register file; BNN_ROWS_LEAN: 230)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import numpy as np, torch
from bnn_amd import hipops
from tests.golden import gen
dev = torch.device("cuda:0")
N = 256
x = torch.from_numpy(gen.normal(1, (8, 3, 224, 224))).to(dev).repeat(N // 8, 1, 1, 1)
w = torch.from_numpy(gen.conv_weight("kaiming", 3, (64, 3, 7, 7))).to(dev)
a = torch.rand(64, device=dev) + 0.5; b = torch.randn(64, device=dev) * 0.3
xa = torch.from_numpy(np.maximum(gen.normal(2, (8, 64, 56, 56)), 0)).to(dev).repeat(N // 8, 1, 1, 1)
act = hipops.pack_act(xa); act.nonneg = True
pw = hipops.pack_weight(torch.from_numpy(gen.conv_weight("kaiming", 4, (64, 64, 3, 3))).to(dev))
thr = hipops.sign_thresholds(pw, a, b)
kw = dict(bn_scale=a, bn_shift=b, relu=True, out_f32=False, out_packed=True, stride=1, padding=1, sign_thresholds=thr)
s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
stem = lambda: hipops.stem7x7(x, w, a, b)
conv = lambda: hipops.bconv2d_fused(act, pw, **kw)
for _ in range(300):
    stem()
torch.cuda.synchronize()


def run(ns, per):
    """ns stems on stream 0 and, beside each, `per` convolutions on stream 1 (host order: stem, per x conv, stem, ...)."""
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(ns):
        with torch.cuda.stream(s0): stem()
        for _ in range(per):
            with torch.cuda.stream(s1): conv()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


NS, PER = 100, 5
for rep in range(2):
    ta = run(NS, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(NS * PER):
        with torch.cuda.stream(s1): conv()
    torch.cuda.synchronize(); tb = (time.perf_counter() - t0) * 1e3
    tc = run(NS, PER)
    print("stems %.1f ms (%.1f us each)  convs %.1f ms (%.1f us each)  together %.1f ms = %.2f of the sum" % (
        ta, ta / NS * 1e3, tb, tb / (NS * PER) * 1e3, tc, tc / (ta + tb)))
