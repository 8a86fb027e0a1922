"""Worker of tests/test_gpu_dist.py: launched as `python -m torch.distributed.run --nproc-per-node 2 dist_worker.py`
with both ranks on ONE GPU (cuda:0) under gloo — RCCL refuses two ranks on one device, so this is how the N > 1 code
(batch shards, ragged shards, two streams + collective ordering, DDP gradient all-reduce) meets the real HIP executors
before it meets an 8-GPU node.  Mirrors examples/imagenet.py:139-177 (one process per rank, DDP) and
examples/cifar10.py:74-77 (gather of the replicas' outputs).  Prints DIST_WORKER_OK on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402

import bnn_amd as bnn  # noqa: E402
from bnn_amd import fastpath, training  # noqa: E402
from bnn_amd.inference import FusedResNet, PipelinedInference, auto_fusion  # noqa: E402
from bnn_amd.models import resnet18  # noqa: E402
from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer  # noqa: E402
from bnn_amd.parallel import ShardedInference, shard_batch  # noqa: E402
from tests.golden import gen  # noqa: E402

DEV = torch.device("cuda:0")


def _cfg():
    return bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                       weight_pre_process=XNORWeightBinarizer)


def _r18():
    net = bnn.prepare_binary_model(resnet18(), _cfg(), custom_config_layers_name={
        "conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 1).items()})
    return net.to(DEV).eval()


def check_inference(rank, world):
    net = _r18()
    single = FusedResNet(net)
    # even and ragged shards (8 ranks: 16 even, 61 = 5 shards of 8 + 3 of 7, 13 = shards of 2 and 1)
    for total in ((8, 7, 5) if world <= 4 else (16, 61, 13)):
        x = torch.from_numpy(gen.normal(90 + total, (total, 3, 64, 64))).to(DEV)
        want = single(x).clone()                          # what one process computes for the whole batch
        mine = shard_batch(x, rank, world).contiguous()
        with torch.no_grad():
            got = ShardedInference(net)(mine)             # net(x): AutoFusion -> the HIP executors, then the gather
        assert got.shape == want.shape and torch.equal(got, want), f"rank {rank}: total {total}"
    assert auto_fusion(net).calls["eager"] + auto_fusion(net).calls["graph"] >= 3
    # two batches in flight, each batch's gather issued on the stream that computed it, same host order on all ranks
    per = 8 // min(world, 8) if world <= 4 else 2         # images per rank and batch
    xs = [torch.from_numpy(gen.normal(60 + i, (per * world, 3, 64, 64))).to(DEV) for i in range(5)]
    want = [single(x).clone() for x in xs]
    pipe = PipelinedInference(net, xs[0][:per].contiguous(), n_streams=2, fresh_input=True)
    models = [ShardedInference(_Fresh(e)) for e in pipe.engines]
    got = []
    for i, x in enumerate(xs):
        k = i % 2
        with torch.cuda.stream(pipe.stream(i)):
            got.append(models[k].forward_even(shard_batch(x, rank, world).contiguous()).clone())
    pipe.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b)


class _Fresh(nn.Module):
    def __init__(self, engine):
        super().__init__()
        self.engine = engine

    def forward(self, x):
        return self.engine.forward_fresh(x, clone=False)


def check_ddp(rank, world):
    """DDP gradient all-reduce around HIP forward + HIP gradient kernels: every rank ends with the gradients one
    process computes for the whole batch (examples/imagenet.py:146-147,172-177)."""
    def small():
        net = nn.Sequential(nn.Conv2d(3, 64, 3, padding=1, bias=False), nn.BatchNorm2d(64), nn.ReLU(),
                            nn.Conv2d(64, 64, 3, padding=1, bias=False), nn.BatchNorm2d(64), nn.ReLU(),
                            nn.Conv2d(64, 128, 3, stride=2, padding=1, bias=True), nn.BatchNorm2d(128), nn.ReLU(),
                            nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(128, 10))
        net = bnn.prepare_binary_model(net, _cfg(), custom_config_layers_name={"0": bnn.BConfig(), "11": bnn.BConfig()})
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 4).items()})
        return net.to(DEV).eval()      # eval-mode BN: per-sample independent, so shard gradients add up exactly

    x = torch.from_numpy(gen.normal(31, (8, 3, 16, 16))).to(DEV)
    t = (torch.arange(8) % 10).to(DEV)

    def loss(m, xb, tb):
        return nn.functional.cross_entropy(m(xb), tb, reduction="sum") / 8.0

    ref = small()
    before = fastpath.stats()["conv2d_train"]
    loss(ref, x, t).backward()
    assert fastpath.stats()["conv2d_train"] == before + 2          # the two binary convs ran the HIP training path
    ddp = training.make_ddp(small(), DEV)
    lo, hi = rank * 8 // world, (rank + 1) * 8 // world
    (loss(ddp, x[lo:hi], t[lo:hi]) * world).backward()            # DDP averages over ranks
    for (n, p), (_, q) in zip(ref.named_parameters(), ddp.module.named_parameters()):
        a, b = p.grad, q.grad
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-6 * float(a.abs().max()) + 1e-9), (n, float((a - b).abs().max()))
    # identical on every rank (bitwise: the all-reduce result)
    flat = torch.cat([p.grad.flatten() for p in ddp.module.parameters()]).cpu()
    every = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(every, flat)
    assert all(torch.equal(every[0], e) for e in every)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # every rank of this worker shares cuda:0 on purpose (one GPU on the test box).  The mapping the real launch uses —
    # LOCAL_RANK -> cuda:LOCAL_RANK, no reliance on HIP_VISIBLE_DEVICES — is bench.py's and is asserted there
    # (tests/test_gpu_dist.py::test_rank_to_device_mapping).
    torch.cuda.set_device(DEV)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        what = sys.argv[1] if len(sys.argv) > 1 else "all"
        if what in ("all", "inference"):
            check_inference(rank, world)
        if what in ("all", "ddp"):
            check_ddp(rank, world)
        dist.barrier()
        if rank == 0:
            print("DIST_WORKER_OK", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
