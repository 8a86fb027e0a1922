#!/usr/bin/env python3
"""Per-wave cycle stamps of the one-launch hierarchical block (a -DHB_TIMING build of hblock.hip):
   make -C binary-networks-pytorch_amd/csrc OUTDIR=/tmp/hbt OBJDIR=/tmp/hbt/obj EXTRA=-DHB_TIMING   (or tools/hblock_timing.sh)
   BNN_AMD_LIB=/tmp/hbt/libbnn_hip.so python tools/exp_hblock_timing.py
s_memtime ticks are 100 MHz (10 ns)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import numpy as np
import torch
from bnn_amd import hipops, native

N = int(os.environ.get("BATCH", "128"))
DEV = "cuda:0"
SHAPES = [(64, 64, 56), (128, 128, 28), (256, 256, 14), (512, 512, 7)]
PLANS = [("default", {}), ("whole", dict(throughput=True))]
g = torch.Generator().manual_seed(0)
lib = native.require()
for c_in, planes, hw in SHAPES:
    ws = [torch.randn(planes // 2, c_in, 3, 3, generator=g).to(DEV), torch.randn(planes // 4, planes // 2, 3, 3, generator=g).to(DEV),
          torch.randn(planes // 4, planes // 4, 3, 3, generator=g).to(DEV)]
    bn = lambda c: ((torch.rand(c, generator=g) + 0.5).to(DEV), (torch.randn(c, generator=g) * 0.3).to(DEV))
    x = torch.randn(N, c_in, hw, hw, generator=g).to(DEV)
    res = torch.randn(N, planes, hw, hw, generator=g).to(DEV)
    p_in = hipops.bn_act_pack(x, *bn(c_in), relu=True)
    pack = hipops.hblock_pack(*[hipops.pack_weight(w) for w in ws], bn(planes // 2), bn(planes // 4), bn(planes))
    for name, plan in PLANS:
        for _ in range(20):
            hipops.hblock_forward(p_in, pack, res, **plan)
        torch.cuda.synchronize()
        buf = np.zeros(8 * 16 * 4096, np.uint64)
        assert lib.bnn_hip_debug_hblock_timing(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size)) == 0
        d = buf.reshape(4096, 16, 8).astype(np.float64)
        nb = int((d[:, 0, 0] > 0).sum())
        d = d[:nb]
        t0 = d[..., 0].min()
        d = (d[..., :7] - t0) / 100.0       # us
        entry, zero, p0, c1, c2, c3, end = (d[..., i] for i in range(7))
        wg_end = end.max(axis=1, keepdims=True)
        m = lambda a: round(float(a.mean()), 1)
        print(json.dumps({"shape": f"{c_in}->{planes} {hw}x{hw}", "plan": name, "workgroups": nb, "kernel_us": round(float(end.max()), 1),
                          "entry": m(entry), "zero_fill": m(zero - entry), "load_planes": m(p0 - zero), "conv1": m(c1 - p0),
                          "conv2": m(c2 - c1), "conv3": m(c3 - c2), "copy_out_and_tail": m(end - c3),
                          "wave_lifetime": m(end - entry), "wg_lifetime": m(wg_end[:, 0] - entry.min(axis=1)),
                          "last_wg_start": round(float(entry.min(axis=1).max()), 1)}))
