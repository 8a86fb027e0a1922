import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import numpy as np, torch
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests.test_gpu_fused import _r18, dev, gen
from bnn_amd.inference import FusedResNet
g = np.load(os.path.join(ROOT, "tests", "golden", "resnet18.npz"))
net = _r18()
for tag, shape in [("32", (4, 3, 32, 32)), ("64", (2, 3, 64, 64)), ("224", (2, 3, 224, 224))]:
    x = dev(gen.normal(gen.seed_of("r18", tag), shape))
    ref = g["logits_" + tag]
    for mfma in (True, False):
        y = FusedResNet(net, use_mfma_stem=mfma)(x).cpu().numpy()
        print(tag, "mfma-split-stem" if mfma else "library-stem   ", "max|err|/max|ref| = %.2e" % (np.abs(y - ref).max() / np.abs(ref).max()))
    with torch.no_grad():
        y = net(x).cpu().numpy()
    print(tag, "per-layer path ", "max|err|/max|ref| = %.2e" % (np.abs(y - ref).max() / np.abs(ref).max()))
