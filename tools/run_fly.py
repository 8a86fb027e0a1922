#!/usr/bin/env python3
"""A few launches of the one-launch layer on BASELINE config 2 (for rocprofv3 --pmc / --kernel-trace runs).
   PLAN="images,rows,waves,obw,ahead" selects a band plan (default: the planner's); ITERS launches (default 5)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch  # noqa: E402

from bnn_amd import hipops, native  # noqa: E402
from tests.golden import gen  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("BATCH", "256"))
x = torch.from_numpy(gen.activation("relu", 7, (8, 128, 56, 56))).to(dev).repeat(N // 8, 1, 1, 1)
pw = hipops.pack_weight(torch.from_numpy(gen.conv_weight("kaiming", 8, (128, 128, 3, 3))).to(dev))
plan = None
if os.environ.get("PLAN"):
    v = [int(t) for t in os.environ["PLAN"].split(",")]
    plan = native.FlyPlan()
    v = (v + [-1, -1, -1, -1])[:8]
    (plan.images_per_band, plan.rows_per_band, plan.waves, plan.blocks_per_unit, plan.pack_ahead, plan.fine_head,
     plan.fine_tail, plan.producers) = v
for _ in range(int(os.environ.get("ITERS", "5"))):
    out = hipops.bconv2d_direct(x, pw, padding=1, plan=plan)
torch.cuda.synchronize()
print("ok", float(out[0, 0, 0, 0]))
