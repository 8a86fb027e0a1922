import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
import bench
from bnn_amd.inference import FusedResNet
from tests.golden import gen
dev = torch.device("cuda:0")
net = bench.build_model(dev)
x = torch.from_numpy(gen.normal(100, (8, 3, 224, 224))).to(dev).repeat(32, 1, 1, 1)
def run(ov):
    f = FusedResNet(net, overlap_shortcut=ov).capture(x)
    xi = f.static_input
    for _ in range(10): f(xi)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): f(xi)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 50 * 1e3
for rep in range(3):
    print("overlap %.4f ms   serial %.4f ms" % (run(True), run(False)))
