"""The stem kernel back to back for SECONDS seconds (power / clock sampling beside it: tools/stem_power.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
from bnn_amd import hipops
from tests.golden import gen
dev = torch.device("cuda:0")
x = torch.from_numpy(gen.normal(1, (8, 3, 224, 224))).to(dev).repeat(32, 1, 1, 1)
w = torch.from_numpy(gen.conv_weight("kaiming", 3, (64, 3, 7, 7))).to(dev)
a = torch.rand(64, device=dev) + 0.5; b = torch.randn(64, device=dev) * 0.3
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < float(os.environ.get("SECONDS", "4")):
    for _ in range(200):
        hipops.stem7x7(x, w, a, b)
    torch.cuda.synchronize(); n += 200
print("%d launches, %.1f us each" % (n, (time.perf_counter() - t0) / n * 1e6))
