#!/usr/bin/env python3
"""Interleaved, repeated timing of band plans of the one-launch layer on config 2 (median of REPS rounds)."""
import itertools
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch  # noqa: E402

from bnn_amd import hipops, native  # noqa: E402
from tests.golden import gen  # noqa: E402

dev = torch.device("cuda:0")
N = 256
x = torch.from_numpy(gen.activation("relu", 7, (8, 128, 56, 56))).to(dev).repeat(N // 8, 1, 1, 1)
pw = hipops.pack_weight(torch.from_numpy(gen.conv_weight("kaiming", 8, (128, 128, 3, 3))).to(dev))


def mk(v):
    p = native.FlyPlan()
    (p.images_per_band, p.rows_per_band, p.waves, p.blocks_per_unit, p.pack_ahead, p.fine_head, p.fine_tail,
     p.producers) = v
    return p


plans = [(1, 56, 16, o, -1, h, t, pr) for o, h, t, pr in itertools.product((2,), (0, 1), (1, 2), (1, 2))] + \
    [(1, 56, w, 2, -1, 1, 2, pr) for w, pr in ((12, 2), (14, 2), (15, 2), (16, -1))]
for _ in range(300):
    hipops.bconv2d_direct(x, pw, padding=1)
times = {v: [] for v in plans}
for rep in range(int(os.environ.get("REPS", "5"))):
    for v in plans:
        pl = mk(v)
        for _ in range(10):
            hipops.bconv2d_direct(x, pw, padding=1, plan=pl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            hipops.bconv2d_direct(x, pw, padding=1, plan=pl)
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) * 1e3 / 30)
for v, t in sorted(times.items(), key=lambda kv: statistics.median(kv[1])):
    print(v[3], v[5], v[6], v[7], "median %.1f  min %.1f  max %.1f" % (statistics.median(t), min(t), max(t)))
