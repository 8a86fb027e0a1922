"""The exact multi-GPU bench path on ONE GPU: a one-rank RCCL process group (backend "nccl" is RCCL on ROCm),
PipelinedInference (HIP-graph replay on two streams) + ShardedInference.forward_even with the all-gather
issued on the stream that computed the batch — so the code an 8-GPU node runs has been through RCCL before
it ever sees one.  Plus bench.py itself under torch.distributed.run (world size 1) vs the plain run.

Round 4: the N > 1 code itself on the one GPU there is — TWO ranks sharing cuda:0 under gloo (RCCL refuses two ranks
on one device): `bench.py --gpus 2 --backend gloo` (validate_gather with world = 2, the distinct-blocks assertion,
per-rank timing gather, self-launch) in three engines, tests/helpers/dist_worker.py (even + ragged shards through
net(x) and the HIP executors, two streams + collective ordering, DDP gradients == single process), and a one-rank
RCCL DDP training step.  The first 8-GPU run is then only new in link topology."""
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
    # the overlapped form bench.py uses: the gather is asynchronous, the slot's next launch waits for it
    models = [ShardedInference(e, force_collective=True) for e in pipe.engines]
    outs = []
    for rep in range(3):
        for i, x in enumerate(xs):
            k = i % 2
            with torch.cuda.stream(pipe.stream(i)):
                pipe.input(i).copy_(x)
                out = models[k].forward_even(pipe.engines[k].static_input, overlap=True)
                if rep == 2 and i >= 2:            # the last launch of each slot: read after wait()
                    models[k].wait()
                    outs.append((i, out.clone()))
    pipe.synchronize()
    for i, o in outs:
        assert torch.equal(o, want[i])
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
    assert plain["dist"] == {"world_size": 1, "initialized": False, "launcher": "none", "rank_devices": [0]}
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


# ---- round 4: two ranks on ONE GPU (gloo), DDP over RCCL ------------------------------------------------------------

@pytest.mark.parametrize("engine", ["graph", "graph_fresh", "net_call", "layerwise"])
def test_bench_with_two_ranks_sharing_the_gpu(engine):
    """`bench.py --gpus 2 --backend gloo`: self-launch of two ranks, each a full copy of the bench on cuda:0; the line
    must carry the world-2 gather check (every rank recomputed both ranks' first images and compared them with its
    gathered copy, and the two rank blocks are distinct) and one step time per rank."""
    rec = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--engine", engine, "--steps", "3",
                      "--warmup", "1", "--spinup", "2", "--batch", "16", "--sustain", "0", "--no-cpu-baseline",
                      "--no-roofline", "--no-extras"])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 32 and rec["config"]["engine"] == engine
    assert rec["dist"]["world_size"] == 2 and rec["dist"]["backend"] == "gloo" and rec["dist"]["launcher"] == "self"
    assert rec["dist"]["gpus_visible"] >= 1
    assert rec["gather_check"]["ranks_checked"] == 2 and rec["gather_check"]["bit_equal"] is True
    pr = rec["per_rank_ms_per_step"]
    assert len(pr["all"]) == 2 and 0 < pr["min"] <= pr["max"]
    assert rec["value"] > 0 and rec["scaling"] == "weak"


def _run_worker(what, nproc=2):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "helpers", "dist_worker.py"), what]
    out = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="4" if nproc <= 2 else "1"),
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and "DIST_WORKER_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_two_ranks_even_and_ragged_shards_through_the_hip_executors():
    """tests/helpers/dist_worker.py `inference`: gathered logits == one process's logits bit for bit, for even and
    ragged shards through `net(x)` (AutoFusion -> fused HIP executor) and for two batches in flight with the gather of
    each batch issued on its own stream (same host order on both ranks)."""
    _run_worker("inference")


def test_two_ranks_ddp_gradients_with_hip_kernels_equal_one_process():
    _run_worker("ddp")


# ---- round 5: world size 8 rehearsed on the one GPU ------------------------------------------------------------------

@pytest.mark.parametrize("engine", ["graph", "net_call"])
def test_bench_with_eight_ranks_sharing_the_gpu(engine):
    """`bench.py --gpus 8 --backend gloo --batch 8`: what the driver's 8-GPU run executes, minus xGMI — the self-launch
    of 8 ranks, `validate_gather` over 8 rank blocks on every rank, 8 per-rank step times, the rank -> device map and
    the scaling-efficiency record against a stored N = 1 line."""
    rec = _run_bench([sys.executable, "bench.py", "--gpus", "8", "--backend", "gloo", "--engine", engine, "--steps", "3",
                      "--warmup", "1", "--spinup", "2", "--batch", "8", "--sustain", "0", "--no-cpu-baseline",
                      "--no-roofline", "--no-extras"])
    assert rec["n_gpus"] == 8 and rec["config"]["global_batch"] == 64 and rec["config"]["engine"] == engine
    assert rec["dist"]["world_size"] == 8 and rec["dist"]["backend"] == "gloo" and rec["dist"]["launcher"] == "self"
    assert rec["gather_check"]["ranks_checked"] == 8 and rec["gather_check"]["bit_equal"] is True
    assert rec["gather_check"]["images_per_rank"] == 8
    pr = rec["per_rank_ms_per_step"]
    assert len(pr["all"]) == 8 and 0 < pr["min"] <= pr["max"]
    # LOCAL_RANK -> cuda:(LOCAL_RANK % visible GPUs): on this box every rank lands on the one GPU
    n_dev = torch.cuda.device_count()
    assert rec["dist"]["rank_devices"] == [r % n_dev for r in range(8)]
    assert rec["value"] > 0 and rec["scaling"] == "weak"
    # round 6: the line says where an N-rank step's time goes — per rank, the step without its collective; the collective
    # alone, idle and behind graph replays; the preflight record (peer-access matrix, collective smoke test)
    assert len(pr["compute_only"]) == 8 and len(pr["collective_share"]) == 8
    assert rec["collective"]["bytes_per_rank"] == 8 * 4000 and len(rec["collective"]["per_rank"]["idle_us"]) == 8
    assert rec["collective"]["under_graph_replay"]["added_us_per_gather"] == rec["collective"]["under_graph_replay"]["added_us_per_gather"]
    pre = rec["dist"]["preflight"]
    assert pre["world_size"] == 8 and pre["collective_smoke"]["ok"] is True and len(pre["peer_access"]) == n_dev


def test_rank_to_device_mapping(tmp_path):
    """The launcher form of the driver (`torch.distributed.run`, RCCL): rank r of a node runs on cuda:LOCAL_RANK — set
    from LOCAL_RANK alone, HIP_VISIBLE_DEVICES untouched — and an N > 1 line carries `scaling_efficiency` against a
    stored N = 1 line of the same engine and per-GPU batch.  (One GPU here: world 1 through RCCL for the mapping, world
    2 through gloo for the efficiency record.)"""
    common = ["--steps", "3", "--warmup", "1", "--spinup", "2", "--batch", "16", "--sustain", "0", "--no-cpu-baseline",
              "--no-roofline", "--no-extras"]
    one = _run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                      "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "1"] + common,
                     env_extra={"HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES", "0")})
    assert one["dist"]["backend"] == "nccl" and one["dist"]["rank_devices"] == [0] and "scaling_efficiency" not in one
    n1 = tmp_path / "n1.json"
    n1.write_text(json.dumps(one))
    two = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo"] + common,
                     env_extra={"BNN_BENCH_N1_JSON": str(n1)})
    eff = two["scaling_efficiency"]
    assert eff["n1_value"] == one["value"] and eff["n1_source"].endswith("n1.json")
    assert abs(eff["efficiency"] - two["value"] / 2 / one["value"]) < 1e-9 and 0.1 < eff["efficiency"] < 1.5


def test_eight_ranks_even_and_ragged_shards_through_the_hip_executors():
    """tests/helpers/dist_worker.py `inference` with 8 ranks on the one GPU: 16 images (even), 61 (five shards of 8 and
    three of 7) and 13 (shards of 2 and 1) through `net(x)` + the gather == one process's logits bit for bit, and two
    batches in flight with each batch's gather on its own stream."""
    _run_worker("inference", nproc=8)


def test_overlapped_gather_over_rccl_fifty_steps_with_changing_inputs(rccl_world1):
    """`forward_even(overlap=True)` as bench.py's graph engines use it — the asynchronous RCCL all-gather behind the
    slot's graph replay, the slot's NEXT replay waiting for it — for 50 steps on two slots with a different input every
    step, EVERY step's gathered logits checked (round 4 checked the last launch of three repetitions)."""
    net = _r18()
    xs = [torch.from_numpy(gen.normal(300 + i, (8, 3, 64, 64))).to(DEV) for i in range(10)]
    single = FusedResNet(net)
    want = [single(x).clone() for x in xs]
    pipe = PipelinedInference(net, xs[0], n_streams=2)
    models = [ShardedInference(e, force_collective=True) for e in pipe.engines]
    kept = []
    for step in range(50):
        k, j = step % 2, (7 * step) % 10
        with torch.cuda.stream(pipe.stream(step)):
            pipe.input(step).copy_(xs[j])
            out = models[k].forward_even(pipe.engines[k].static_input, overlap=True)
            models[k].wait()                       # this stream waits for the collective; the host does not
            kept.append((j, out.clone()))          # copied out in stream order: the slot's next gather overwrites `out`
    pipe.synchronize()
    for step, (j, o) in enumerate(kept):
        assert torch.equal(o, want[j]), step


def test_ddp_training_step_over_rccl_world1(rccl_world1):
    """`training.make_ddp` on the GPU over RCCL (examples/imagenet.py:146-147): a DDP-wrapped binary ResNet-18 does one
    SGD step with the HIP forward + HIP gradient kernels; gradients and updated weights equal the un-wrapped model's
    (one rank: the all-reduce is the identity)."""
    from bnn_amd import fastpath, training
    x = torch.from_numpy(gen.normal(77, (4, 3, 64, 64))).to(DEV)
    t = torch.tensor([1, 5, 9, 13], device=DEV)

    def step(model, params):
        opt = torch.optim.SGD(params, lr=0.1, momentum=0.9)
        before = fastpath.stats()["conv2d_train"]
        loss = torch.nn.functional.cross_entropy(model(x), t)
        loss.backward()
        assert fastpath.stats()["conv2d_train"] == before + 19
        grads = [p.grad.clone() for p in params]
        opt.step()
        return float(loss.detach()), grads

    ref = _r18().train()
    ddp_net = _r18().train()
    ddp = training.make_ddp(ddp_net, torch.device(DEV))
    assert type(ddp).__name__ == "DistributedDataParallel" and dist.get_backend() == "nccl"
    l0, g0 = step(ref, list(ref.parameters()))
    l1, g1 = step(ddp, list(ddp_net.parameters()))
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    for a, b in zip(g0, g1):        # (library BatchNorm backward may reduce with atomics: tight tolerance, not bits)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 * float(a.abs().max()) + 1e-12)
    for p, q in zip(ref.parameters(), ddp_net.parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-6 * float(p.abs().max()) + 1e-12)


def test_data_parallel_replicas_share_packed_weights_per_device_and_version():
    """`nn.DataParallel` (examples/cifar10.py:74-77) replicates the model on EVERY forward: `__dict__` copied shallowly,
    parameters replaced by broadcast copies.  A replica's forward must not re-pack (and block on the zero-weight flag)
    every time: packs are cached on the master layer per (device, weight version)."""
    import torch.nn as nn
    from bnn_amd import fastpath
    conv = nn.Conv2d(64, 64, 3, padding=1, bias=False)
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    layer = bnn.prepare_binary_model(conv, cfg).to(DEV).eval()
    x = torch.from_numpy(gen.normal(3, (2, 64, 12, 12))).to(DEV)

    def replica_forward():
        # what torch.nn.parallel.replicate does for one module: shallow copy + a NEW tensor holding the same values
        r = layer._replicate_for_data_parallel()
        r._parameters = {}
        r.weight = layer.weight.detach().clone()
        r.bias = None
        with torch.no_grad():
            return r(x), r
    with torch.no_grad():
        want = layer(x)
    packs = fastpath.stats()["weight_packs"]
    y1, r1 = replica_forward()
    assert fastpath.stats()["weight_packs"] == packs + 1 and torch.equal(y1, want)
    assert r1.__dict__["_bnn_master"] is layer and "_bnn_packed" not in r1.__dict__
    for _ in range(3):                                       # later forwards: new replicas, new storage — no re-pack
        y, r = replica_forward()
        assert torch.equal(y, want)
    assert fastpath.stats()["weight_packs"] == packs + 1
    r2 = r._replicate_for_data_parallel()                    # a replica of a replica still points at the master
    assert r2.__dict__["_bnn_master"] is layer
    with torch.no_grad():
        layer.weight.neg_()                                  # optimiser step on the master: version counter moves
    y, _ = replica_forward()
    assert fastpath.stats()["weight_packs"] == packs + 2 and torch.equal(y, -want)
    # the real thing on the devices there are: DataParallel's replicate() onto [0], and the wrapper on one GPU
    net = _r18()
    xs = torch.from_numpy(gen.normal(5, (4, 3, 64, 64))).to(DEV)
    with torch.no_grad():
        want = FusedResNet(net)(xs)
        from bnn_amd.inference import auto_fusion, per_layer_forward
        rep = nn.parallel.replicate(net, [0])[0]
        assert getattr(rep, "_is_replica", False) and rep.__dict__["_bnn_auto"] is auto_fusion(net)
        per_layer0 = fastpath.stats()["conv2d"]
        y_rep = rep(xs)                                      # the replica runs the fused executor of ITS device ...
        st = auto_fusion(net)
        assert torch.equal(y_rep, want) and fastpath.stats()["conv2d"] == per_layer0
        assert list(st.replica_engines) == [xs.device] and st.engine is None
        eng = st.replica_engines[xs.device][1]
        packs = fastpath.stats()["weight_packs"]
        for _ in range(2):                                   # ... and so do the replicas of later forwards: same executor
            assert torch.equal(nn.parallel.replicate(net, [0])[0](xs), want)
        assert st.replica_engines[xs.device][1] is eng and fastpath.stats()["weight_packs"] == packs
        assert st.calls["graph"] >= 1
        net.layer3[0].conv1.weight.neg_()                    # a master parameter changes: the executor is re-derived
        y_new = nn.parallel.replicate(net, [0])[0](xs)
        assert st.replica_engines[xs.device][1] is not eng and not torch.equal(y_new, want)
        assert torch.equal(y_new, FusedResNet(net)(xs))
        with per_layer_forward():                            # the per-layer path of a replica: packs cached on the master
            y_lw = nn.parallel.replicate(net, [0])[0](xs)
            packs = fastpath.stats()["weight_packs"]
            y_lw2 = nn.parallel.replicate(net, [0])[0](xs)
        assert fastpath.stats()["weight_packs"] == packs and torch.equal(y_lw, y_lw2)
        assert torch.allclose(y_lw, y_new, rtol=1e-3, atol=1e-3 * float(y_new.abs().max()))
        assert torch.equal(nn.DataParallel(net, device_ids=[0])(xs), y_new)


# ---- round 6: the first N-GPU run describes itself ------------------------------------------------------------------

def test_two_entries_in_the_per_device_executor_table():
    """`nn.DataParallel` over more than one device (examples/cifar10.py:76) keeps ONE executor per device in
    `AutoFusion.replica_engines`.  With one GPU in the box the second device is a key of its own (`cuda:1`) holding a
    stand-in: calls on cuda:0 must find THEIR entry, leave the other one alone, and a changed master parameter must
    re-derive only the entry of the device that is asked next (each entry carries the signature it was derived for)."""
    import torch.nn as nn
    from bnn_amd.inference import auto_fusion
    net = _r18()
    xs = torch.from_numpy(gen.normal(5, (4, 3, 64, 64))).to(DEV)
    other = torch.device("cuda", 1)
    with torch.no_grad():
        want = FusedResNet(net)(xs)
        st = auto_fusion(net)
        assert torch.equal(nn.parallel.replicate(net, [0])[0](xs), want)
        sig0, eng0 = st.replica_engines[xs.device]
        stand_in = object()
        st.replica_engines[other] = (sig0, stand_in)                     # "the replica on the second GPU"
        assert torch.equal(nn.parallel.replicate(net, [0])[0](xs), want)
        assert set(st.replica_engines) == {xs.device, other}
        assert st.replica_engines[xs.device][1] is eng0 and st.replica_engines[other] == (sig0, stand_in)
        net.layer2[0].conv1.weight.neg_()                                # optimiser step on the master
        y_new = nn.parallel.replicate(net, [0])[0](xs)
        assert not torch.equal(y_new, want) and torch.equal(y_new, FusedResNet(net)(xs))
        sig1, eng1 = st.replica_engines[xs.device]
        assert eng1 is not eng0 and sig1 != sig0
        assert st.replica_engines[other] == (sig0, stand_in)             # stale by ITS signature: re-derived when asked
        st.reset()
        assert not st.replica_engines


def test_bench_preflight_and_the_self_describing_fields_of_an_n_rank_line():
    """`bench.py --preflight` (one JSON line, exit code 0, the peer-access matrix and a collective smoke test) and the
    fields an N > 1 line carries so that a slow first multi-GPU run says where the time went: the same checks under
    `dist.preflight`, the step without its collective per rank, the collective alone (idle / behind graph replays)."""
    pre = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--preflight"])
    assert pre["ok"] is True and pre["problems"] == []
    p = pre["preflight"]
    assert p["world_size"] == 2 and p["gpus_visible"] >= 1 and p["collective_smoke"]["ok"] is True
    assert len(p["peer_access"]) == p["gpus_visible"] and all(p["peer_access"][i][i] for i in range(p["gpus_visible"]))
    # RCCL with more ranks than GPUs is refused by the preflight itself, with a reason, before any process group exists
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "bench.py", "--gpus", str(torch.cuda.device_count() + 1), "--preflight"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert line and json.loads(line[-1])["ok"] is False and "one GPU per rank" in json.loads(line[-1])["problems"][0]
    rec = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                      "--spinup", "2", "--batch", "16", "--sustain", "0", "--no-cpu-baseline", "--no-roofline", "--no-extras"])
    assert rec["dist"]["preflight"]["collective_smoke"]["ok"] is True and "peer_access" in rec["dist"]["preflight"]
    pr = rec["per_rank_ms_per_step"]
    assert len(pr["compute_only"]) == 2 and len(pr["collective_share"]) == 2 and all(v > 0 for v in pr["compute_only"])
    c = rec["collective"]
    assert c["bytes_per_rank"] == 16 * 4000 and c["idle_us"] > 0 and c["under_graph_replay"]["steps"] >= 10
    assert len(c["per_rank"]["idle_us"]) == 2 and len(c["per_rank"]["added_us_per_gather"]) == 2
