import os, sys
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
for _ in range(3):
    y, pk = hipops.stem7x7(x, w, a, b)
torch.cuda.synchronize()
t = pk.M.cpu().numpy().reshape(-1)[:256 * 8 * 6].astype(np.float64).reshape(256, 8, 6)
names = ["wait barrier0", "fetch-issue+matrix", "epilogue", "wait barrier1", "commit", "pool+stores"]
tiles = 56
print("cycles per tile (mean over workgroups), by wave:")
for k, nm in enumerate(names):
    print("  %-20s" % nm, " ".join("%7.0f" % (t[:, wv, k].mean() / tiles) for wv in range(8)))
print("  %-20s" % "total", " ".join("%7.0f" % (t[:, wv, :].sum(axis=1).mean() / tiles) for wv in range(8)))
