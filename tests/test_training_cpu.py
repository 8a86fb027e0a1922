"""Data-parallel training wiring on CPU (world_size 2, gloo): `bnn_amd.training.make_ddp` around a
binarised model; gradients after the bucketed all-reduce equal the single-process gradients of the
whole batch, and the straight-through estimator matches the reference's (bnn/ops.py:63-73).
On the GPU box the same wrapper runs over RCCL with the HIP forward (tests/test_gpu_training.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

import bnn_amd as bnn
from bnn_amd import training
from bnn_amd.ops import BasicInputBinarizer, SignActivation, XNORWeightBinarizer
from tests.golden import gen


def _net():
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8), nn.ReLU(),
                        nn.Conv2d(8, 16, 3, padding=1, bias=True), nn.BatchNorm2d(16), nn.ReLU(),
                        nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(16, 4))
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    net = bnn.prepare_binary_model(net, cfg, custom_config_layers_name={"0": bnn.BConfig(), "8": bnn.BConfig()})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 4).items()})
    return net.eval()        # eval-mode BN: per-sample independent, so shard gradients add up exactly


def _loss(net, x, t):
    return nn.functional.cross_entropy(net(x), t, reduction="sum") / 8.0   # global-batch mean


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        x = torch.from_numpy(gen.normal(31, (8, 3, 10, 10)))
        t = torch.arange(8) % 4
        lo, hi = rank * 4, rank * 4 + 4
        ddp = training.make_ddp(_net())
        # DDP averages over ranks: scale the local (global-mean) loss by world to recover the sum
        (_loss(ddp, x[lo:hi], t[lo:hi]) * world).backward()
        grads = {n: p.grad.numpy() for n, p in ddp.module.named_parameters()}
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **grads)
    finally:
        dist.destroy_process_group()


def test_ddp_gradients_equal_single_process(tmp_path):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True,
                       start_method="spawn")
    net = _net()
    x = torch.from_numpy(gen.normal(31, (8, 3, 10, 10)))
    t = torch.arange(8) % 4
    _loss(net, x, t).backward()
    r0, r1 = (np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world))
    for n, p in net.named_parameters():
        assert np.array_equal(r0[n], r1[n]), n                     # all-reduced: identical on both ranks
        assert np.allclose(r0[n], p.grad.numpy(), rtol=1e-5, atol=1e-7), n
    assert any(np.abs(r0[n]).max() > 0 for n in ("3.weight", "3.bias"))   # the binary layer does learn


def test_straight_through_estimator_is_the_hard_tanh_mask():
    x = torch.tensor([-2.0, -1.0, -0.999, -0.0, 0.0, 0.5, 1.0, 3.0], requires_grad=True)
    y = SignActivation.apply(x)
    assert torch.equal(y.detach(), torch.tensor([-1.0, -1.0, -1.0, 0.0, 0.0, 1.0, 1.0, 1.0]))
    y.backward(torch.arange(1.0, 9.0))
    assert torch.equal(x.grad, torch.tensor([0.0, 0.0, 3.0, 4.0, 5.0, 6.0, 0.0, 0.0]))   # |x| >= 1 -> 0


def test_training_fast_paths_decline_cpu_tensors_and_other_modules():
    """Dispatch rules of the training-side fused ops (bnn_amd/training.py) on a box without a GPU: CPU tensors, other
    window sizes and modules with hooks take the modules themselves — same values, torch's autograd graph."""
    x = torch.from_numpy(gen.normal(71, (2, 3, 16, 16))).requires_grad_(True)
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
    assert not training.stem_conv_applies(conv, x)                      # CPU tensor
    y = training.stem_conv(x, conv)
    assert torch.equal(y, conv(x)) and "Convolution" in type(y.grad_fn).__name__
    pool = nn.AvgPool2d(kernel_size=2, stride=2, ceil_mode=True, count_include_pad=False)
    p = training.shortcut_pool(x, pool)
    assert torch.equal(p, pool(x)) and "AvgPool2x2Fn" not in type(p.grad_fn).__name__
    bn, act, mp_ = nn.BatchNorm2d(64).train(), nn.ReLU(), nn.MaxPool2d(3, 2, 1)
    assert not training.bn_act_applies(bn, act, y)
    z = training.stem_tail(y, bn, act, mp_)
    bn2 = nn.BatchNorm2d(64).train()
    assert torch.allclose(z, mp_(act(bn2(conv(x)))))
    z.sum().backward()
    assert x.grad is not None and conv.weight.grad is not None
