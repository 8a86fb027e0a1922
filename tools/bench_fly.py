#!/usr/bin/env python3
"""Band-plan sweep of the one-launch layer (bnn_hip_bconv2d_direct) on BASELINE config 2 and on the ResNet-18 layer
shapes, next to the two-launch form (pack_act + bconv2d) and the packed-input kernel alone.

    python tools/bench_fly.py [--batch 256] [--quick] > gpurun_out/bench_fly.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bnn_amd import hipops, native  # noqa: E402
from tests.golden import gen  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--quick", action="store_true")
ap.add_argument("--c2-only", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
info = native.device_info(0)
peak = info["compute_units"] * 64 * info["clock_khz"] * 1e3


def ev(fn, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def plan(images, rows, waves, obw, ahead=-1, head=-1, tail=-1, prod=-1):
    p = native.FlyPlan()
    p.images_per_band, p.rows_per_band, p.waves, p.blocks_per_unit, p.pack_ahead = images, rows, waves, obw, ahead
    p.fine_head, p.fine_tail, p.producers = head, tail, prod
    return p


def run_shape(N, C, H, W, O, k, s, p, plans, spin=300, iters=30):
    x = torch.from_numpy(gen.activation("relu", 7, (8, C, H, W))).to(dev).repeat(N // 8, 1, 1, 1)
    pw = hipops.pack_weight(torch.from_numpy(gen.conv_weight("kaiming", 8, (O, C, k, k))).to(dev))
    ho, wo = hipops.conv_out_hw(H, W, k, k, s, p, 1)
    lane_ops = 2.0 * ((C * k * k + 31) // 32) * N * O * ho * wo
    rec = {"shape": [N, C, H, W, O, k, s, p], "floor_us": lane_ops / peak * 1e6,
           "hbm_floor_us": (N * C * H * W * 4 + N * O * ho * wo * 4) / 6.3e12 * 1e6}
    act = hipops.pack_act(x)
    for _ in range(spin):
        hipops.bconv2d(act, pw, stride=s, padding=p)
    t = ev(lambda: hipops.bconv2d(act, pw, stride=s, padding=p), iters)
    rec["packed_kernel"] = {"us": t * 1e6, "frac": lane_ops / t / peak}
    t = ev(lambda: hipops.bconv2d(hipops.pack_act(x), pw, stride=s, padding=p), iters)
    rec["two_launch"] = {"us": t * 1e6, "frac": lane_ops / t / peak}
    d = hipops.direct_plan(x.shape, pw, s, p, 1)
    rec["default_plan"] = None if d is None else [d.images_per_band, d.rows_per_band, d.waves, d.blocks_per_unit,
                                                  d.lds_bytes, d.n_bands]
    ref = hipops.bconv2d(act, pw, stride=s, padding=p)
    rec["plans"] = []
    for pl in [None] + plans:
        try:
            out = hipops.bconv2d_direct(x, pw, stride=s, padding=p, plan=pl)
        except native.NativeError as e:
            rec["plans"].append({"plan": None if pl is None else [pl.images_per_band, pl.rows_per_band, pl.waves,
                                                                  pl.blocks_per_unit, pl.pack_ahead, pl.fine_head, pl.fine_tail, pl.producers], "error": str(e)})
            continue
        ok = bool(torch.equal(out, ref))
        for _ in range(spin // 3):
            hipops.bconv2d_direct(x, pw, stride=s, padding=p, plan=pl)
        t = ev(lambda: hipops.bconv2d_direct(x, pw, stride=s, padding=p, plan=pl), iters)
        rec["plans"].append({"plan": "default" if pl is None else [pl.images_per_band, pl.rows_per_band, pl.waves,
                                                                   pl.blocks_per_unit, pl.pack_ahead, pl.fine_head, pl.fine_tail, pl.producers],
                             "us": t * 1e6, "frac": lane_ops / t / peak, "bit_identical": ok})
    return rec


out = {"device": info, "results": []}
B = args.batch
c2_plans = [plan(1, 56, 16, 2, prod=1), plan(1, 56, 16, 2, prod=2), plan(1, 56, 16, 2, prod=3)]
out["results"].append(run_shape(B, 128, 56, 56, 128, 3, 1, 1, c2_plans if not args.quick else c2_plans[:3]))
if not args.quick and not args.c2_only:
    for sh, pls in [
        ((B, 64, 56, 56, 64, 3, 1, 1), [plan(1, 56, 8, 2, prod=1), plan(1, 56, 8, 2, prod=2), plan(1, 56, 16, 2, prod=2), plan(1, 56, 8, 1, prod=1), plan(1, 28, 8, 2, prod=1)]),
        ((B, 64, 56, 56, 128, 3, 2, 1), [plan(1, 28, 8, 2, prod=1), plan(1, 28, 8, 4, prod=1), plan(1, 28, 16, 2, prod=2), plan(1, 14, 8, 2, prod=1)]),
        ((B, 128, 28, 28, 128, 3, 1, 1), [plan(1, 28, 8, 2, prod=1), plan(1, 28, 8, 2, prod=0), plan(1, 28, 16, 2, prod=2), plan(1, 28, 8, 4, prod=1), plan(1, 28, 4, 2, prod=1)]),
        ((B, 64, 28, 28, 128, 1, 1, 0), [plan(1, 28, 8, 1, prod=1), plan(1, 28, 8, 4, prod=1), plan(1, 28, 4, 4, prod=1), plan(1, 28, 8, 2, prod=0)]),
        ((B, 128, 28, 28, 256, 3, 2, 1), [plan(1, 14, 8, 1, prod=1), plan(1, 14, 8, 2, prod=1), plan(1, 14, 8, 4, prod=1), plan(1, 14, 4, 2, prod=1)]),
        ((B, 256, 14, 14, 256, 3, 1, 1), [plan(1, 14, 8, 1, prod=1), plan(1, 14, 8, 2, prod=1), plan(1, 14, 8, 1, prod=0), plan(1, 14, 4, 1, prod=1), plan(1, 14, 16, 1, prod=2)]),
        ((B, 128, 14, 14, 256, 1, 1, 0), [plan(1, 14, 8, 1, prod=1), plan(1, 14, 4, 2, prod=1), plan(1, 14, 4, 4, prod=0)]),
        ((B, 256, 14, 14, 512, 3, 2, 1), [plan(1, 7, 8, 1, prod=1), plan(1, 7, 4, 1, prod=1), plan(1, 7, 8, 2, prod=1)]),
        ((B, 512, 7, 7, 512, 3, 1, 1), [plan(1, 7, 8, 1, prod=1), plan(1, 7, 8, 1, prod=0), plan(1, 7, 4, 1, prod=1), plan(1, 7, 8, 2, prod=1), plan(1, 7, 16, 1, prod=2)]),
        ((B, 256, 7, 7, 512, 1, 1, 0), [plan(1, 7, 8, 1, prod=1), plan(1, 7, 4, 2, prod=1), plan(1, 7, 4, 4, prod=0), plan(1, 7, 2, 4, prod=0)]),
    ]:
        out["results"].append(run_shape(*sh, pls, spin=150, iters=20))
print(json.dumps(out, indent=1))
