"""Two batches in flight: does the phase between the two streams matter?  Stream 1 starts DELAY_US after stream 0
(a device-side spin), then both free-run for STEPS batches each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
import bench  # noqa: F401  (model builder)
from bnn_amd.inference import PipelinedInference
from bnn_amd.models import resnet18
dev = torch.device("cuda:0")
net = bench.build_model(dev, resnet18)
x = torch.randn(256, 3, 224, 224, device=dev)
pipe = PipelinedInference(net, x, n_streams=2)
STEPS = 40
def run(delay_us):
    for i in range(20): pipe.launch(i)
    pipe.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(pipe.streams[1]):
        if delay_us: torch.cuda._sleep(int(delay_us * 2100))   # ~2.1 GHz cycles
    for i in range(2 * STEPS): pipe.launch(i)
    pipe.synchronize()
    dt = time.perf_counter() - t0
    return 256 * 2 * STEPS / dt
for d in (0, 300, 650, 1000, 0, 650):
    print("delay %4d us: %.0f images/s" % (d, run(d)))
