#!/usr/bin/env python3
"""Per-wave cycle stamps of the one-launch layer (a -DBNN_FLY_TIMING build of bconv_fly.hip, see tools/fly_variants.sh):
   BNN_AMD_LIB=.../variants/timing/libbnn_hip.so PLAN=1,56,16,2,-1 python tools/exp_fly_timing.py"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import numpy as np  # noqa: E402
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
for _ in range(300):
    hipops.bconv2d_direct(x, pw, padding=1, plan=plan)
torch.cuda.synchronize()
lib = native.require()
nb = N * (1 if plan is None else -(-56 // plan.rows_per_band))
waves = 16 if plan is None else plan.waves
buf = np.zeros(8 * 16 * 4096, np.uint64)
assert lib.bnn_hip_debug_fly_timing(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size)) == 0
d = buf.reshape(4096, 16, 8)[:nb, :waves].astype(np.float64)
t0 = d[..., 0].min()
entry, zero, first, end = (d[..., i] - t0 for i in range(4))
wait, conv, pack = d[..., 4], d[..., 5], d[..., 6]
units = (buf.reshape(4096, 16, 8)[:nb, :waves, 7] >> np.uint64(32)).astype(np.float64)
items = (buf.reshape(4096, 16, 8)[:nb, :waves, 7] & np.uint64(0xFFFFFFFF)).astype(np.float64)
kernel = end.max()


def st(a):
    return {"min": float(a.min()), "mean": float(a.mean()), "max": float(a.max())}


rec = {"plan": os.environ.get("PLAN", "default"), "ticks_kernel": float(kernel),
       "entry": st(entry), "after_zero_fill": st(zero - entry), "first_conv_minus_entry": st(first - entry),
       "wave_end": st(end), "wg_end": st(end.max(axis=1)), "tail_idle_per_wave": st(end.max(axis=1, keepdims=True) - end),
       "kernel_minus_wave_end": st(kernel - end),
       "wait_total": st(wait), "pack_inside_wait": st(pack), "conv_total": st(conv), "units": st(units), "items": st(items),
       "lifetime": st(end - entry), "other": st((end - entry) - wait - conv)}
print(json.dumps(rec, indent=1))
