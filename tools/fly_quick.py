#!/usr/bin/env python3
"""Quick timing of the one-launch layer on config 2 for a list of plans (PLANS="1,56,16,1,-1;1,56,16,2,-1"),
   for A/B runs of variant libraries (BNN_AMD_LIB)."""
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
act = hipops.pack_act(x)
for _ in range(500):
    hipops.bconv2d(act, pw, padding=1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40):
    hipops.bconv2d(act, pw, padding=1)
e1.record()
torch.cuda.synchronize()
res = ["packed kernel: %.1f us" % (e0.elapsed_time(e1) * 1e3 / 40)]
for ps in os.environ.get("PLANS", "1,56,16,1,-1;1,56,16,2,-1").split(";"):
    v = [int(t) for t in ps.split(",")]
    plan = native.FlyPlan()
    v = (v + [-1, -1, -1, -1])[:8]
    (plan.images_per_band, plan.rows_per_band, plan.waves, plan.blocks_per_unit, plan.pack_ahead, plan.fine_head,
     plan.fine_tail, plan.producers) = v
    for _ in range(100):
        hipops.bconv2d_direct(x, pw, padding=1, plan=plan)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        hipops.bconv2d_direct(x, pw, padding=1, plan=plan)
    e1.record()
    torch.cuda.synchronize()
    res.append("%s: %.1f us" % (ps, e0.elapsed_time(e1) * 1e3 / 40))
print(os.environ.get("BNN_AMD_LIB", "default").split("/")[-2] if os.environ.get("BNN_AMD_LIB") else "default", " | ".join(res))
