"""pack_act (BasicInputBinarizer on device) on the config-2 tensor: time and achieved HBM read rate."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
from bnn_amd import hipops
dev = torch.device("cuda:0")
x = torch.randn(int(os.environ.get("BATCH", "256")), 128, 56, 56, device=dev).relu_()
for _ in range(300): hipops.pack_act(x)
def t(fn, n=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for _ in range(3):
    us = t(lambda: hipops.pack_act(x))
    print("pack_act %s: %.1f us  %.2f TB/s read" % (tuple(x.shape), us, x.numel() * 4 / us / 1e6))
