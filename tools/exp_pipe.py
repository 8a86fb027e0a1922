"""Two batches in flight with different executor options:  KW='{"overlap_shortcut": false}' python tools/exp_pipe.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
sys.argv = sys.argv[:1]
import torch
import bench
from bnn_amd.inference import PipelinedInference
from bnn_amd.models import resnet18
dev = torch.device("cuda:0")
net = bench.build_model(dev, resnet18)
x = torch.randn(256, 3, 224, 224, device=dev)
NS = int(os.environ.get("STREAMS", "2"))
STEPS = 60
def run(kw):
    pipe = PipelinedInference(net, x, n_streams=NS, **kw)
    for i in range(30): pipe.launch(i)
    pipe.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(NS * STEPS): pipe.launch(i)
    pipe.synchronize()
    return 256 * NS * STEPS / (time.perf_counter() - t0)
variants = [json.loads(v) for v in os.environ.get("KW", "{}").split(";")]
for rep in range(2):
    for kw in variants:
        print(kw, "%.0f images/s" % run(kw))
