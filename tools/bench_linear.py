"""Linear / tiny-image shapes: the one-launch layer (bnn_hip_bconv2d_direct) against pack_act + bconv2d (ADVICE round 3:
with H*W == 1 a pack item of the one-launch kernel holds one valid lane in 64).  Prints us per call for both routes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
from bnn_amd import hipops
from tests.golden import gen
dev = torch.device("cuda:0")


def t(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


SHAPES = [  # (N, C, H, W, O, k, pad)
    (256, 512, 1, 1, 1000, 1, 0), (1024, 512, 1, 1, 1000, 1, 0), (4096, 512, 1, 1, 1000, 1, 0),
    (256, 4096, 1, 1, 4096, 1, 0), (4096, 1024, 1, 1, 1024, 1, 0), (32768, 256, 1, 1, 256, 1, 0),
    (64, 2048, 1, 1, 1000, 1, 0), (256, 512, 2, 2, 512, 1, 0), (256, 512, 2, 2, 512, 3, 1),
    (256, 512, 4, 4, 512, 3, 1), (256, 256, 7, 7, 256, 3, 1), (256, 512, 1, 7, 512, 1, 0)]
for N, C, H, W, O, k, pad in SHAPES:
    x = torch.from_numpy(gen.normal(1, (min(N, 64), C, H, W))).to(dev).repeat((N + 63) // 64, 1, 1, 1)[:N].contiguous()
    pw = hipops.pack_weight(torch.from_numpy(gen.conv_weight("kaiming", 2, (O, C, k, k))).to(dev))
    a = hipops.bconv2d_direct(x, pw, padding=pad, route="direct")
    b = hipops.bconv2d(hipops.pack_act(x), pw, padding=pad)
    assert torch.equal(a, b)
    td = t(lambda: hipops.bconv2d_direct(x, pw, padding=pad, route="direct"))
    tp = t(lambda: hipops.bconv2d(hipops.pack_act(x), pw, padding=pad))
    ta = t(lambda: hipops.bconv2d_direct(x, pw, padding=pad))
    print(f"N={N:6d} C={C:5d} {H}x{W} O={O:5d} k={k}: one-launch {td:8.1f} us   pack+conv {tp:8.1f} us   "
          f"default route {ta:8.1f} us", flush=True)
