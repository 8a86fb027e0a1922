"""Thread re-entrancy of the boundary (SURVEY §8(b) "Threading / streams").

The reference's multi-GPU wrapper `nn.DataParallel` (examples/cifar10.py:74-77, examples/imagenet.py:187) calls every
layer's forward from one Python thread per replica.  The C-ABI keeps no mutable state besides an atomic launch counter,
but the Python host side has shared state: the per-layer packed-weight cache (`fastpath.packed_weight`), the stats
counters, the torch-op pack cache.  Here two threads, each on its own HIP stream, drive (a) ONE converted layer and
(b) the two slots of a PipelinedInference, and the results must be bit-equal to a serial run.
"""
import threading

import numpy as np
import pytest
import torch
import torch.nn as nn

import bnn_amd as bnn
from bnn_amd import fastpath, native
from bnn_amd.inference import FusedResNet, PipelinedInference
from bnn_amd.models import resnet18
from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer
from tests.golden import gen

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _net_call_means_per_layer():
    """In this module `net(x)` means the per-layer drop-in path (one launch per binary layer + torch BN / ReLU / add);
    the fused executor is built explicitly (`FusedResNet(net)`).  What `net(x)` does by default — the fused executor,
    bnn_amd/inference.py: AutoFusion — is tested in tests/test_gpu_dropin.py."""
    from bnn_amd.inference import per_layer_forward
    with per_layer_forward():
        yield
DEV = "cuda:0"


def _cfg():
    return bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                       weight_pre_process=XNORWeightBinarizer)


def _run_threads(workers):
    errors = []

    def guard(fn):
        def run():
            try:
                fn()
            except BaseException as exc:  # noqa: BLE001 - reported to the main thread
                errors.append(exc)
        return run
    ts = [threading.Thread(target=guard(w)) for w in workers]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ts), "a worker thread hung"
    if errors:
        raise errors[0]


def test_two_threads_two_streams_through_one_layer():
    conv = nn.Conv2d(96, 80, 3, padding=1, bias=True)
    conv.weight.data.copy_(torch.from_numpy(gen.conv_weight("kaiming", 311, (80, 96, 3, 3))))
    layer = bnn.prepare_binary_model(conv, _cfg()).to(DEV).eval()
    rounds, n_threads = 24, 2
    xs = [[torch.from_numpy(gen.activation("normal", 400 + 31 * t + r, (3 + r % 3, 96, 17, 13))).to(DEV)
           for r in range(rounds)] for t in range(n_threads)]
    # first use of the layer happens INSIDE the threads: both may find the weight cache empty at the same time
    assert "_bnn_packed" not in layer.__dict__
    outs = [[None] * rounds for _ in range(n_threads)]
    launches0, calls0 = native.launch_count(), fastpath.stats()["conv2d"]
    barrier = threading.Barrier(n_threads)

    def worker(t):
        def run():
            stream = torch.cuda.Stream(device=DEV)
            barrier.wait()
            with torch.no_grad(), torch.cuda.stream(stream):
                for r in range(rounds):
                    outs[t][r] = layer(xs[t][r])
            stream.synchronize()
        return run
    _run_threads([worker(t) for t in range(n_threads)])
    torch.cuda.synchronize()
    assert fastpath.stats()["conv2d"] == calls0 + n_threads * rounds        # the counters lost no update
    assert native.launch_count() >= launches0 + n_threads * rounds
    with torch.no_grad():
        for t in range(n_threads):
            for r in range(rounds):
                assert torch.equal(outs[t][r], layer(xs[t][r])), (t, r)       # serial, default stream


def test_two_threads_drive_the_two_slots_of_a_pipelined_executor():
    net = bnn.prepare_binary_model(resnet18(), _cfg(), custom_config_layers_name={
        "conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 1).items()})
    net = net.to(DEV).eval()
    rounds = 6
    batches = [[torch.from_numpy(gen.normal(700 + 10 * k + r, (8, 3, 64, 64))).to(DEV) for r in range(rounds)]
               for k in range(2)]
    serial = FusedResNet(net)
    want = [[serial(b).clone() for b in batches[k]] for k in range(2)]
    pipe = PipelinedInference(net, batches[0][0], n_streams=2)
    got = [[None] * rounds for _ in range(2)]
    barrier = threading.Barrier(2)

    def worker(k):
        def run():
            barrier.wait()
            for r in range(rounds):
                with torch.cuda.stream(pipe.stream(k)):
                    pipe.input(k).copy_(batches[k][r], non_blocking=True)
                out = pipe.launch(k)
                # the slot's buffer is overwritten by the slot's next launch: copy it out IN THE SLOT'S STREAM ORDER (a
                # clone on the thread's default stream can sit behind the other slot's replay in a shared hardware queue
                # and read the buffer after the next launch has rewritten it)
                with torch.cuda.stream(pipe.stream(k)):
                    got[k][r] = out.clone()
                pipe.stream(k).synchronize()
        return run
    _run_threads([worker(0), worker(1)])
    for k in range(2):
        for r in range(rounds):
            assert torch.equal(got[k][r], want[k][r]), (k, r)
    assert np.isfinite(want[0][0].cpu().numpy()).all()
