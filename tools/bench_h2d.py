"""PCIe-inclusive rate: every batch starts in pinned host memory and is copied into the graph's static input
on the stream that will run it (copies of batch i+1 overlap the compute of batch i)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
import bench
from bnn_amd.inference import PipelinedInference
from tests.golden import gen
dev = torch.device("cuda:0")
net = bench.build_model(dev)
B = 256
x = torch.from_numpy(gen.normal(100, (8, 3, 224, 224))).to(dev).repeat(B // 8, 1, 1, 1)
for ns in (2, 3):
    pipe = PipelinedInference(net, x, n_streams=ns)
    for dtype, name in ((torch.float32, "fp32 host batches (154 MB)"), (torch.uint8, "uint8 host batches (38.5 MB) + on-GPU convert")):
        host = [torch.empty((B, 3, 224, 224), dtype=dtype).pin_memory() for _ in range(ns)]
        stage = [torch.empty((B, 3, 224, 224), dtype=dtype, device=dev) for _ in range(ns)] if dtype != torch.float32 else None
        def step(i):
            k = i % ns
            with torch.cuda.stream(pipe.stream(i)):
                if stage is None:
                    pipe.input(i).copy_(host[k], non_blocking=True)
                else:
                    stage[k].copy_(host[k], non_blocking=True)
                    pipe.input(i).copy_(stage[k])          # uint8 -> fp32 on the GPU (normalisation would go here)
            return pipe.launch(i)
        for i in range(6): step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 40
        for i in range(n): step(i)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print("%d streams, %-46s %.3f ms/batch -> %.0f img/s (%.1f GB/s over PCIe)" % (ns, name, dt * 1e3, B / dt, host[0].numel() * host[0].element_size() / dt / 1e9))
