"""Is the stand-alone stem loop clock-limited by the power budget?  The same launch back to back, and behind 0.3 / 1 / 3 ms
of a one-workgroup spin kernel (GPU busy, almost no power): duration of the stem launch alone (events around it) and,
with a -DBNN_ROWS_TIMING library, the shader cycles a wave counted per tile -> the clock the kernel ran at."""
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
timing = "tim" in os.environ.get("BNN_AMD_LIB", "")
for _ in range(600):
    hipops.stem7x7(x, w, a, b)
torch.cuda.synchronize()
for gap_us in (0, 300, 1000, 3000, 0):
    durs, cyc = [], []
    for i in range(60):
        if gap_us:
            torch.cuda._sleep(int(gap_us * 2400))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y, pk = hipops.stem7x7(x, w, a, b); e1.record()
        durs.append((e0, e1))
    torch.cuda.synchronize()
    us = np.array([e0.elapsed_time(e1) * 1e3 for e0, e1 in durs])[10:]
    line = "gap %5d us: stem %.1f us (min %.1f)" % (gap_us, np.median(us), us.min())
    if timing:
        t = pk.M.cpu().numpy().reshape(-1)[:512 * 4 * 16].astype(np.float64).reshape(512, 4, 16)
        per_tile = t[:, 0, :14].sum(axis=1).mean() / 28
        line += "   %.0f cycles per tile -> %.2f GHz" % (per_tile, per_tile * 28 / np.median(us) / 1e3)
    print(line)
