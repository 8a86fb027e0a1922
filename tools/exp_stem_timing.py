"""Per-role, per-segment cycle counts of the wave-specialised stem (needs a -DBNN_STEM_TIMING build:
   make -C binary-networks-pytorch_amd/csrc OUTDIR=../bnn_amd/_lib/variants/timing OBJDIR=../../build/obj_timing EXTRA=-DBNN_STEM_TIMING
   BNN_AMD_LIB=.../variants/timing/libbnn_hip.so python tools/exp_stem_timing.py)"""
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
for mode, kw in (("full", {}), ("f32-only", {"out_packed": False})):
    if mode != "full":
        continue  # the dump needs the M plane
    for _ in range(3):
        y, pk = hipops.stem7x7(x, w, a, b, **kw)
    torch.cuda.synchronize()
    t = pk.M.cpu().numpy().reshape(-1)[:256 * 8 * 6].astype(np.float64).reshape(256, 8, 6)
    tiles = 56
    print("cycles per tile (mean over workgroups), by wave; waves 0-3 matrix, 4-7 helper")
    mn = ["fetch+matrix", "wait X", "epilogue+commit", "wait Y", "-", "loop top"]
    hn = ["-", "pool+stores", "wait X", "flush", "wait Y", "loop top"]
    for k in range(6):
        print("  %-16s" % mn[k], " ".join("%7.0f" % (t[:, wv, k].mean() / tiles) for wv in range(4)),
              "   | %-14s" % hn[k], " ".join("%7.0f" % (t[:, wv, k].mean() / tiles) for wv in range(4, 8)))
    print("  %-14s" % "total", " ".join("%7.0f" % (t[:, wv, :].sum(axis=1).mean() / tiles) for wv in range(8)))
