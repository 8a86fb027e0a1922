"""The N>1 path (batch shards + one all-gather of logits) on CPU: world_size 2, gloo backend.

Covers bnn_amd.parallel: shard_bounds / shard_batch, ShardedInference.forward (ragged shards) and
forward_even (the single-collective fast path bench.py uses).  The model runs its CPU composition
path here; on the GPU box the same wrapper drives the fused HIP executor over RCCL.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

import bnn_amd as bnn
from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer
from bnn_amd.parallel import ShardedInference, shard_batch, shard_bounds
from tests.golden import gen


def small_binary_net():
    net = nn.Sequential(nn.Conv2d(3, 16, 3, padding=1, bias=False), nn.BatchNorm2d(16), nn.ReLU(),
                        nn.Conv2d(16, 32, 3, padding=1, bias=False), nn.BatchNorm2d(32), nn.ReLU(),
                        nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(32, 10))
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    net = bnn.prepare_binary_model(net, cfg, custom_config_layers_name={"0": bnn.BConfig(), "8": bnn.BConfig()})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 3).items()})
    return net.eval()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, batch, even, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        x = torch.from_numpy(gen.normal(11, (batch, 3, 12, 12)))
        model = ShardedInference(small_binary_net())
        local = shard_batch(x, rank, world)
        y = model.forward_even(local) if even else model(local)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), y.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch,even", [(8, True), (8, False), (7, False)])
def test_sharded_inference_equals_single_process(tmp_path, batch, even):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), batch, even, str(tmp_path)), nprocs=world,
                       join=True, start_method="spawn")
    x = torch.from_numpy(gen.normal(11, (batch, 3, 12, 12)))
    with torch.no_grad():
        ref = small_binary_net()(x).numpy()
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), f"rank {r}: gathered logits differ from the single-process run"


def test_shard_bounds_partition_the_batch():
    for total in (0, 1, 7, 8, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_unwrapped_single_process_path():
    m = ShardedInference(small_binary_net())
    x = torch.from_numpy(gen.normal(11, (4, 3, 12, 12)))
    assert torch.equal(m(x), m.forward_even(x))
