"""Stem kernel at batch 256, 224x224: its three arithmetic modes, all output modes."""
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
MODES = (("split fp16 hi+lo (default)", {}), ("plain fp16", {"fp16": True}), ("exact fp32", {"exact_fp32": True}))
if os.environ.get("ONLY") == "default":   # PMC passes: only the default kernel, both outputs
    for _ in range(5):
        hipops.stem7x7(x, w, a, b)
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(600):   # ~0.2 s of work first: the first timed region after idle runs at ramping clocks (+15 %)
    hipops.stem7x7(x, w, a, b)
torch.cuda.synchronize()
for name, kw in MODES:
    print("%-26s full %.1f us   packed-only %.1f us   f32-only %.1f us" % (
        name, t(lambda: hipops.stem7x7(x, w, a, b, **kw)), t(lambda: hipops.stem7x7(x, w, a, b, out_f32=False, **kw)),
        t(lambda: hipops.stem7x7(x, w, a, b, out_packed=False, **kw))))
