"""Warm-clock time of the default stem kernel at batch 256 (three output modes); BNN_AMD_LIB selects a variant build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
from bnn_amd import hipops
from tests.golden import gen
dev = torch.device("cuda:0")
N = int(os.environ.get("BATCH", "256"))
x = torch.from_numpy(gen.normal(1, (8, 3, 224, 224))).to(dev).repeat(N // 8, 1, 1, 1)
w = torch.from_numpy(gen.conv_weight("kaiming", 3, (64, 3, 7, 7))).to(dev)
a = torch.rand(64, device=dev) + 0.5; b = torch.randn(64, device=dev) * 0.3
def t(fn, n=100):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for _ in range(600):
    hipops.stem7x7(x, w, a, b)
torch.cuda.synchronize()
for name, kw in (("split", {}), ("fp16", {"fp16": True})):
    print("%-14s %-6s full %.1f us   packed-only %.1f us   f32-only %.1f us" % (
        os.environ.get("TAG", ""), name, t(lambda: hipops.stem7x7(x, w, a, b, **kw)),
        t(lambda: hipops.stem7x7(x, w, a, b, out_f32=False, **kw)),
        t(lambda: hipops.stem7x7(x, w, a, b, out_packed=False, **kw))), flush=True)
