"""Where does the stem kernel differ from the torch sequence?  (debug helper)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch, torch.nn.functional as F
from bnn_amd import hipops
dev = torch.device("cuda:0")
torch.manual_seed(0)
N, H, W = 1, int(os.environ.get("H", 224)), int(os.environ.get("W", 224))
x = torch.randn(N, 3, H, W, device=dev)
w = torch.randn(64, 3, 7, 7, device=dev) * 0.1
a = torch.rand(64, device=dev) + 0.5; b = torch.randn(64, device=dev) * 0.3
y, _ = hipops.stem7x7(x, w, a, b)
ref = F.max_pool2d(F.relu(F.conv2d(x.double(), w.double(), None, 2, 3) * a.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)), 3, 2, 1)
err = (y.double() - ref).abs().amax(dim=(0, 1))          # [Hp, Wp]
print("max err", float(err.max()), "ref max", float(ref.abs().max()))
bad = (err > 1e-4)
print("bad pooled pixels:", int(bad.sum()), "of", bad.numel())
ys, xs = torch.nonzero(bad, as_tuple=True)
if len(ys):
    print("rows", sorted(set((ys // 4).tolist()))[:20], "tile cols", sorted(set((xs // 8).tolist()))[:20])
    print("x within tile", sorted(set((xs % 8).tolist())), "y within tile", sorted(set((ys % 4).tolist())))
