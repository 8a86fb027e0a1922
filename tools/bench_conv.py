#!/usr/bin/env python3
"""Kernel-level timing of the binary conv for BASELINE config 2 and every binary-conv shape of
ResNet-18 @224, batch 256.  Honour BNN_AMD_LIB to A/B differently built libraries:

    BNN_AMD_LIB=build/variants/libA.so python tools/bench_conv.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]

import torch  # noqa: E402

from bnn_amd import hipops, native  # noqa: E402
from tests.golden import gen  # noqa: E402

SHAPES = [  # name, C, H, W, O, k, stride, pad, count in R18
    ("c2_128x56", 128, 56, 56, 128, 3, 1, 1, 0),
    ("l1_64x56", 64, 56, 56, 64, 3, 1, 1, 4),
    ("l2_0_c1_s2", 64, 56, 56, 128, 3, 2, 1, 1),
    ("l2_128x28", 128, 28, 28, 128, 3, 1, 1, 3),
    ("l2_ds", 64, 28, 28, 128, 1, 1, 0, 1),
    ("l3_0_c1_s2", 128, 28, 28, 256, 3, 2, 1, 1),
    ("l3_256x14", 256, 14, 14, 256, 3, 1, 1, 3),
    ("l3_ds", 128, 14, 14, 256, 1, 1, 0, 1),
    ("l4_0_c1_s2", 256, 14, 14, 512, 3, 2, 1, 1),
    ("l4_512x7", 512, 7, 7, 512, 3, 1, 1, 3),
    ("l4_ds", 256, 7, 7, 512, 1, 1, 0, 1),
]


def main():
    N = int(os.environ.get("BATCH", "256"))
    iters = int(os.environ.get("ITERS", "20"))
    only = os.environ.get("ONLY")
    wsel = os.environ.get("WEIGHTS") or None
    dev = torch.device("cuda:0")
    info = native.device_info(0)
    peak = info["compute_units"] * 64 * info["clock_khz"] * 1e3  # CUs x 64 x f_clk
    rows, net_us, net_ops = [], 0.0, 0.0
    for name, C, H, W, O, k, s, p, cnt in SHAPES:
        if only and only not in name:
            continue
        x = torch.from_numpy(gen.activation("relu", 7, (4, C, H, W))).to(dev).repeat(N // 4, 1, 1, 1)
        w = torch.from_numpy(gen.conv_weight("kaiming", 8, (O, C, k, k))).to(dev)
        pw, act = hipops.pack_weight(w), hipops.pack_act(x)
        act.nonneg = os.environ.get("NONNEG") == "1"  # the synthetic input is a ReLU output: M == 0
        for _ in range(3):
            out = hipops.bconv2d(act, pw, stride=s, padding=p, weights=wsel)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            out = hipops.bconv2d(act, pw, stride=s, padding=p, weights=wsel)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        e0.record()
        for _ in range(iters):
            hipops.pack_act(x)
        e1.record()
        torch.cuda.synchronize()
        pack_us = e0.elapsed_time(e1) * 1e3 / iters
        ops = 2.0 * ((C * k * k + 31) // 32) * out.numel()
        rows.append(dict(name=name, us=round(us, 1), frac=round(ops / (us * 1e-6) / peak, 3),
                         out_GBps=round(out.numel() * 4 / us / 1e3, 0), pack_us=round(pack_us, 1),
                         pack_GBps=round(x.numel() * 4 / pack_us / 1e3, 0)))
        net_us += cnt * us
        net_ops += cnt * ops
        del out, act, x
    for r in rows:
        print(json.dumps(r))
    if net_us:
        print(json.dumps(dict(r18_binary_conv_us=round(net_us, 1),
                              r18_frac=round(net_ops / (net_us * 1e-6) / peak, 3), lib=native.lib_path())))


if __name__ == "__main__":
    main()
