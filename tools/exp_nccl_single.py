"""RCCL + HIP graphs + side streams in one process (world_size 1): the call pattern bench.py uses per rank."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch, torch.distributed as dist
import bench
from bnn_amd.inference import FusedResNet, PipelinedInference
from tests.golden import gen
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(4, device=dev); dist.all_reduce(t)               # communicator + watchdog thread are live
net = bench.build_model(dev)
x = torch.from_numpy(gen.normal(100, (8, 3, 224, 224))).to(dev).repeat(8, 1, 1, 1)
pipe = PipelinedInference(net, x, n_streams=2)                  # capture AFTER the process group exists
ref = FusedResNet(net)(x)
outs = []
for i in range(6):
    with torch.cuda.stream(pipe.stream(i)):
        y = pipe.engines[i % 2](pipe.input(i))
        out = y.new_empty(y.shape)
        dist.all_gather_into_tensor(out, y.contiguous())        # world 1: a device-to-device copy through RCCL
        outs.append(out)
dist.barrier()
torch.cuda.synchronize()
assert all(torch.equal(o, ref) for o in outs)
print("rccl + graphs + two streams ok:", len(outs), "gathers")
dist.destroy_process_group()
