"""The exact multi-GPU bench path on ONE GPU: a one-rank RCCL process group (backend "nccl" is RCCL on ROCm),
PipelinedInference (HIP-graph replay on two streams) + ShardedInference.forward_even with the all-gather
issued on the stream that computed the batch — so the code an 8-GPU node runs has been through RCCL before
it ever sees one.  Plus bench.py itself under torch.distributed.run (world size 1) vs the plain run."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

import bnn_amd as bnn
from bnn_amd.inference import FusedResNet, PipelinedInference
from bnn_amd.models import resnet18
from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer
from bnn_amd.parallel import ShardedInference
from tests.golden import gen

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _r18():
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    net = bnn.prepare_binary_model(resnet18(), cfg, custom_config_layers_name={
        "conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 1).items()})
    return net.to(DEV).eval()


@pytest.fixture
def rccl_world1():
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        yield
    finally:
        dist.destroy_process_group()


def test_pipelined_sharded_inference_over_rccl_world1(rccl_world1):
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    net = _r18()
    xs = [torch.from_numpy(gen.normal(60 + i, (8, 3, 64, 64))).to(DEV) for i in range(4)]
    single = FusedResNet(net)
    want = [single(x).clone() for x in xs]
    pipe = PipelinedInference(net, xs[0], n_streams=2)
    models = [ShardedInference(e, force_collective=True) for e in pipe.engines]
    got = []
    for i, x in enumerate(xs):
        k = i % 2
        with torch.cuda.stream(pipe.stream(i)):
            pipe.input(i).copy_(x)
            out = models[k].forward_even(pipe.engines[k].static_input)   # graph replay + all_gather, same stream
            assert out.shape == (8, 1000)
            got.append(out.clone())
    pipe.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # the ragged path (sizes exchanged first) through RCCL as well
    y = ShardedInference(single, force_collective=True)(xs[0][:5].contiguous())
    assert torch.equal(y, single(xs[0][:5].contiguous()))


def _run_bench(cmd, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_line_plain_vs_one_rank_under_the_launcher():
    """`python bench.py` and the driver's launcher form at N = 1 run the same step; the launcher form has an
    initialised RCCL group and says so in the JSON."""
    common = ["--steps", "3", "--warmup", "1", "--batch", "32", "--no-cpu-baseline", "--no-roofline", "--no-extras"]
    plain = _run_bench([sys.executable, "bench.py", "--gpus", "1"] + common)
    launched = _run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py",
                           "--gpus", "1"] + common)
    for rec in (plain, launched):
        assert rec["n_gpus"] == 1 and rec["unit"] == "images/s" and rec["value"] > 0
        assert rec["config"]["global_batch"] == 32 and rec["scaling"] == "weak"
    assert plain["dist"] == {"world_size": 1, "initialized": False, "launcher": "none"}
    assert launched["dist"]["initialized"] and launched["dist"]["backend"] == "nccl"
    assert launched["dist"]["world_size"] == 1 and launched["dist"]["launcher"] == "torchrun"
    # the N > 1 run validates itself: every rank recomputes the first images of every rank and compares them with its
    # gathered copy (here: one rank, through RCCL all the same); per-rank step times are reported
    assert launched["gather_check"] == {"ranks_checked": 1, "images_per_rank": 8, "bit_equal": True,
                                        "how": launched["gather_check"]["how"]}
    assert "gather_check" not in plain
    for rec in (plain, launched):
        pr = rec["per_rank_ms_per_step"]
        assert len(pr["all"]) == 1 and pr["min"] == pr["max"] == pr["all"][0] > 0
    assert 0.5 < launched["value"] / plain["value"] < 2.0
