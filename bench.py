#!/usr/bin/env python3
"""bench.py — images/sec of the binary ResNet-18 224x224 forward on N MI355X (BASELINE.json metric),
with the int-ALU roofline of the dominant kernel (3x3 XNOR-popcount conv, BASELINE config 2) and a
CPU baseline of the reference's op sequence beside it.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 5

One "step" = one forward pass of the whole network over one synthetic batch (256 images per GPU,
resident in HBM before the timed region).  Weak scaling: every rank processes its own 256 images
and the [256,1000] logits are all-gathered over RCCL at the end of every step.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402

import bnn_amd as bnn  # noqa: E402
from bnn_amd import fastpath, hipops, native  # noqa: E402
from bnn_amd.inference import FusedResNet, PipelinedInference  # noqa: E402
from bnn_amd.models import resnet18  # noqa: E402
from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer  # noqa: E402
from bnn_amd.parallel import ShardedInference  # noqa: E402
from tests.golden import gen  # noqa: E402  (portable synthetic-data generator, no reference code)

# ResNet-18 @224: algorithmic int lane-ops per image over all binary convs (SURVEY §A.2 / BASELINE.md §4)
R18_LANE_OPS_PER_IMG = 105.97e6
R18_BINARY_MAC_PER_IMG = 1.6955e9


def xnor_cfg():
    return bnn.BConfig(activation_pre_process=BasicInputBinarizer,
                       activation_post_process=bnn.Identity,
                       weight_pre_process=XNORWeightBinarizer)


def build_model(device):
    """examples/cifar10.py:61-71 model: resnet18, XNOR recipe, conv1 and fc real-valued."""
    net = resnet18()
    net = bnn.prepare_binary_model(net, xnor_cfg(), custom_config_layers_name={
        "conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 1).items()})
    return net.to(device).eval()


def int_alu_peak(info) -> float:
    """Roofline denominator of BASELINE.md §4: CUs x 64 lanes x f_clk 32-bit lane-ops/s.

    This is also the measured issue limit of the two instructions the path is made of: on gfx950
    v_bitop3_b32 / v_xor_b32 and v_bcnt_u32_b32 each issue once per 4 cycles per wave64
    (bench field ``int_alu_probe_Tlane_ops``: the register-only pair sustains 39.2 T at 2.4 GHz),
    unlike v_add_u32 / v_fma_f32 which run at the SIMD-32 rate of one per 2 cycles."""
    return info["compute_units"] * 64 * info["clock_khz"] * 1e3


def simd32_peak(info) -> float:
    """Full-rate VALU bound (4 SIMD x 32 lanes): what a 2-cycle op such as v_add_u32 reaches."""
    return info["compute_units"] * 4 * 32 * info["clock_khz"] * 1e3


def conv_c2_roofline(device, info, batch=256, iters=20, act_kind="relu", nonneg=False):
    """BASELINE config 2: 3x3 Conv2d 128->128, 56x56, batch 256 — the graded kernel.
    Times `iters` launches of bnn_hip_bconv2d with events on the launch stream.
    ``nonneg``: promise the kernel that the input has no negative value (true for a ReLU output):
    it then runs the P-plane-only variant the fused ResNet executor uses."""
    N, C, H, W, O = batch, 128, 56, 56, 128
    x = torch.from_numpy(gen.activation(act_kind, 7, (8, C, H, W))).to(device).repeat(N // 8, 1, 1, 1)
    w = torch.from_numpy(gen.conv_weight("kaiming", 8, (O, C, 3, 3))).to(device)
    pw = hipops.pack_weight(w)
    act = hipops.pack_act(x)
    act.nonneg = bool(nonneg)
    for _ in range(3):
        out = hipops.bconv2d(act, pw, stride=1, padding=1)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = hipops.bconv2d(act, pw, stride=1, padding=1)
    e1.record()
    torch.cuda.synchronize(device)
    t_conv = e0.elapsed_time(e1) * 1e-3 / iters
    # pack kernel (HBM-bound) timed the same way
    e0.record()
    for _ in range(iters):
        hipops.pack_act(x)
    e1.record()
    torch.cuda.synchronize(device)
    t_pack = e0.elapsed_time(e1) * 1e-3 / iters
    K = C * 9
    lane_ops = 2.0 * ((K + 31) // 32) * N * O * H * W           # algorithmic: xor + popcount per 32 MACs
    peak = int_alu_peak(info)
    in_bytes = N * H * W * (2 * 2 * 8)                           # two planes x 2 uint64 words per pixel
    out_bytes = N * O * H * W * 4
    del out
    traffic, traffic_note = pmc_traffic()
    return {
        "bound": "int_alu", "kernel": "bconv_sgpr_kernel<3,3,4>", "workload": "conv3x3 128->128 56x56 b256",
        "achieved": lane_ops / t_conv / 1e12, "peak": peak / 1e12, "unit": "Tlane-op/s",
        "frac": lane_ops / t_conv / peak, "traffic": traffic, "traffic_note": traffic_note,
        "algorithmic_bytes": in_bytes + out_bytes + O * K // 8,
        "avg_kernel_us": t_conv * 1e6, "images_per_s_kernel": N / t_conv,
        "images_per_s_fp32_in_out": N / (t_conv + t_pack),
        "hbm": {"conv_GBps": (in_bytes + out_bytes) / t_conv / 1e9,
                "pack_us": t_pack * 1e6, "pack_GBps": (N * C * H * W * 4 + in_bytes) / t_pack / 1e9,
                "peak_GBps": 8000.0},
        "peak_basis": f"{info['compute_units']} CU x 64 lanes x {info['clock_khz'] / 1e6:.2f} GHz "
                      "(BASELINE.md §4; = issue rate of v_bitop3/v_bcnt, 4 cycles per wave64)",
        "frac_of_simd32_peak": lane_ops / t_conv / simd32_peak(info),
        "act": act_kind,
    }


def pmc_traffic():
    """HBM bytes per launch of the graded kernel from the committed rocprofv3 PMC passes
    (profiles/r01_c2_pmc_counters.json, collected by tools/gpu_profile.sh with separate --pmc runs).
    FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes
    (checked on this box: pack_act's 411 MB float4 stream reads as 205 MB)."""
    path = os.path.join(ROOT, "profiles", "r01_c2_pmc_counters.json")
    try:
        with open(path) as fh:
            pmc = json.load(fh)
        k = next(v for name, v in pmc.items() if "bconv_sgpr_kernel" in name)
        return (2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024, \
            "2*FETCH_SIZE + WRITE_SIZE from profiles/r01_c2_pmc_counters.json (same kernel, same shape)"
    except (OSError, StopIteration, KeyError, ValueError):
        return None, "no PMC summary committed"


def cpu_baseline(sample_batch=64, iters=6):
    """Reference op sequence (torch CPU: sign -> sign(W)*alpha -> conv2d) on the host cores."""
    from oracle import torch_ref  # checker / baseline only
    shapes = torch_ref.resnet18_state_shapes()
    sd = {k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 1).items()}
    x = torch.from_numpy(gen.normal(3, (sample_batch, 3, 224, 224)))
    cores = torch.get_num_threads()
    with torch.no_grad():
        torch_ref.resnet18_forward(sd, x[:4])  # warm-up
        t0 = time.perf_counter()
        for _ in range(iters):
            torch_ref.resnet18_forward(sd, x)
        dt = time.perf_counter() - t0
    return {"value": sample_batch * iters / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{iters} x batch {sample_batch} of the same ResNet-18 224x224 forward "
                      f"(oracle/torch_ref.py, torch {torch.__version__} CPU, fp32, {dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--engine", choices=("graph", "fused", "layerwise"), default="graph",
                    help="graph: fused executor replayed as a HIP graph (default); fused: same, eager "
                         "launches; layerwise: the drop-in per-layer path (pack -> conv -> torch BN/ReLU)")
    ap.add_argument("--streams", type=int, default=2,
                    help="graph engine: batches in flight per GPU (graph-captured executors on their own HIP "
                         "streams, replayed round-robin; 1 = strictly one batch at a time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the GPU path has no CPU stand-in)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    native.require()
    info = native.device_info(local_rank)
    net = build_model(device)
    B = args.batch
    x = torch.from_numpy(gen.normal(100 + rank, (8, 3, 224, 224))).to(device).repeat(B // 8, 1, 1, 1)
    x = x + 0.01 * torch.arange(B, device=device, dtype=torch.float32).view(B, 1, 1, 1)  # distinct images
    if args.engine == "layerwise":
        engine = net
    else:
        engine = FusedResNet(net)
    n_streams = max(1, args.streams) if args.engine == "graph" else 1
    if args.engine == "graph":
        # every stream owns a graph-captured executor whose static input buffer holds its batch (filled by
        # capture): no per-step device-to-device copy of the 154 MB input, and `n_streams` batches in flight
        pipe = PipelinedInference(net, x, n_streams=n_streams)
        models = [ShardedInference(e) for e in pipe.engines]

        def step(i, k_streams=n_streams):
            k = i % k_streams
            with torch.cuda.stream(pipe.streams[k]):
                return models[k].forward_even(pipe.engines[k].static_input)
    else:
        model = ShardedInference(engine)

        def step(i, k_streams=1):
            return model.forward_even(x)

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(steps, warmup, k_streams):
        for i in range(warmup):
            out = step(i, k_streams)
        torch.cuda.synchronize(device)
        barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(steps):
            out = step(warmup + i, k_streams)
        torch.cuda.synchronize(device)
        barrier()
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0, out

    launches0 = native.launch_count()
    with torch.no_grad():
        dt, logits = timed(args.steps, args.warmup, n_streams)
        dt1 = timed(args.steps, 2, 1)[0] if n_streams > 1 else None   # same steps, one batch at a time
    assert logits.shape == (world * B, 1000) and torch.isfinite(logits).all()
    hip_launches = native.launch_count() - launches0
    if world > 1:
        tt = torch.tensor([dt, dt1 or 0.0], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dt1 = float(tt[0].item()), (float(tt[1].item()) if dt1 is not None else None)

    if rank == 0:
        value = world * B * args.steps / dt
        rec = {
            "metric": "images/sec binary ResNet-18 224x224 forward", "value": value, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int1 xnor-popcount (binary convs) + fp32 (stem, BN, fc)",
            "data": "synthetic",
            "config": {"workload": "binary ResNet-18 (bnn.models resnet18, XNOR recipe of examples/cifar10.py, "
                                   "conv1+fc real-valued) 224x224 full forward, batch 256 per GPU",
                       "engine": args.engine, "batches_in_flight": n_streams,
                       "global_batch": world * B, "parallelism": f"dp{world} (batch shards, RCCL all-gather of logits)",
                       "hip_kernel_launches_per_step": hip_launches // max(args.steps + args.warmup, 1)},
            "device": {k: info[k] for k in ("name", "arch", "compute_units", "clock_khz")},
            "net_int_alu_frac": value / world * R18_LANE_OPS_PER_IMG / int_alu_peak(info),
        }
        if dt1 is not None:
            rec["one_batch_at_a_time"] = {"value": world * B * args.steps / dt1, "ms_per_step": dt1 / args.steps * 1e3}
        if not args.no_roofline:
            rec["roofline"] = conv_c2_roofline(device, info, act_kind="relu")
            rec["roofline_normal_input"] = {k: v for k, v in conv_c2_roofline(device, info, act_kind="normal").items()
                                            if k in ("achieved", "frac", "avg_kernel_us")}
            rec["roofline_relu_input_nonneg_kernel"] = {
                k: v for k, v in conv_c2_roofline(device, info, act_kind="relu", nonneg=True).items()
                if k in ("achieved", "frac", "avg_kernel_us")}
            rec["int_alu_probe_Tlane_ops"] = {
                name: round(hipops.probe_int_alu(4096, device, mode)["lane_ops_per_s"] / 1e12, 2)
                for mode, name in hipops.PROBE_MODES.items()}
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline()
        print(json.dumps(rec), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
