"""Per-phase shader-clock counts of the stem kernel (needs a -DBNN_ROWS_TIMING build: tools/stem_variants.sh
timing:-DBNN_ROWS_TIMING; BNN_AMD_LIB=.../variants/timing/libbnn_hip.so python tools/exp_rows_timing.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import numpy as np, torch
from bnn_amd import hipops
from tests.golden import gen
dev = torch.device("cuda:0")
N = 256
x = torch.from_numpy(gen.normal(1, (8, 3, 224, 224))).to(dev).repeat(N // 8, 1, 1, 1)
w = torch.from_numpy(gen.conv_weight("kaiming", 3, (64, 3, 7, 7))).to(dev)
a = torch.rand(64, device=dev) + 0.5; b = torch.randn(64, device=dev) * 0.3
for name, kw in (("split", {}), ("fp16", {"fp16": True})):
    for _ in range(300):
        y, pk = hipops.stem7x7(x, w, a, b, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y, pk = hipops.stem7x7(x, w, a, b, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    t = pk.M.cpu().numpy().reshape(-1)[:512 * 4 * 16].astype(np.float64).reshape(512, 4, 16)
    tiles = 28
    names = ["barrier wait", "flush+fetch+row0", "mfma row 0", "mfma row 1", "mfma row 2", "mfma row 3", "-",
             "finish 0", "finish 1", "finish 2", "finish 3", "-", "commit", "loop tail", "flush bits", "next tile + fetch"]
    print("%s: %.1f us per launch (instrumented); cycles per tile, mean over workgroups, by wave (mg,nh)=(0,0),(1,0),(0,1),(1,1)" % (name, us))
    for k, nm in enumerate(names):
        print("  %-18s" % nm, " ".join("%7.0f" % (t[:, wv, k].mean() / tiles) for wv in range(4)))
    print("  %-18s" % "total", " ".join("%7.0f" % (t[:, wv, :16].sum(axis=1).mean() / tiles) for wv in range(4)),
          "  -> %.2f GHz" % (t[:, 0, :16].sum(axis=1).mean() / us / 1e3))
