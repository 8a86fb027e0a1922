#!/usr/bin/env python3
"""Time bnn_hip_hblock_forward alone on the config-5 block shapes (batch 128 by default), plan by plan:
   python tools/bench_hblock.py [N]      ->  us per launch, fraction of the integer-ALU floor of the block"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
from bnn_amd import hipops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
DEV = "cuda:0"
SHAPES = [(64, 64, 56), (64, 128, 28), (128, 128, 28), (128, 256, 14), (256, 256, 14), (256, 512, 7), (512, 512, 7)]
PLANS = [("default", {}), ("whole", dict(throughput=True)), ("cl12", dict(channel_lanes=True)), ("cl8", dict(channel_lanes=True, waves=8))]
if os.environ.get("PLANS"):
    PLANS = [(p, eval("dict(%s)" % p)) for p in os.environ["PLANS"].split(";")]


def lane_ops(c_in, planes):
    w = lambda k: -(-k // 32)
    return 2 * (w(9 * c_in) * planes // 2 + w(9 * planes // 2) * planes // 4 + w(9 * planes // 4) * planes // 4)


g = torch.Generator().manual_seed(0)
for c_in, planes, hw in SHAPES:
    ws = [torch.randn(planes // 2, c_in, 3, 3, generator=g).to(DEV), torch.randn(planes // 4, planes // 2, 3, 3, generator=g).to(DEV),
          torch.randn(planes // 4, planes // 4, 3, 3, generator=g).to(DEV)]
    bn = lambda c: ((torch.rand(c, generator=g) + 0.5).to(DEV), (torch.randn(c, generator=g) * 0.3).to(DEV))
    x = torch.randn(N, c_in, hw, hw, generator=g).to(DEV)
    res = torch.randn(N, planes, hw, hw, generator=g).to(DEV)
    p_in = hipops.bn_act_pack(x, *bn(c_in), relu=True)
    pack = hipops.hblock_pack(*[hipops.pack_weight(w) for w in ws], bn(planes // 2), bn(planes // 4), bn(planes))
    floor_us = lane_ops(c_in, planes) * hw * hw * N / 39.3216e12 * 1e6
    line = "%3d->%3d %2dx%2d  floor %5.1f us |" % (c_in, planes, hw, hw, floor_us)
    for name, plan in PLANS:
        if not hipops.hblock_supported(N, c_in, hw, hw, planes, **plan):
            line += " %s: n/a |" % name
            continue
        for _ in range(3):
            hipops.hblock_forward(p_in, pack, res, **plan)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20
        e0.record()
        for _ in range(iters):
            hipops.hblock_forward(p_in, pack, res, **plan)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        line += " %s: %6.1f us (%.2f) |" % (name, us, floor_us / us)
    print(line, flush=True)
