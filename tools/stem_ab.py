"""Stem kernel A/B: the pooled-in-accumulators kernel (stem_rows.hip) against the round-2 kernel (test-only library, tests/helpers/legacy.py):
bitwise comparison of both outputs over ragged shapes, then warm-clock timings of the three output modes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch, torch.nn.functional as F
from bnn_amd import hipops
from tests.golden import gen
from tests.helpers import legacy as legacy_lib
dev = torch.device("cuda:0")


def run(legacy, *a, **kw):
    y, pk = legacy_lib.stem_staged(*a, **kw) if legacy else hipops.stem7x7(*a, **kw)
    torch.cuda.synchronize()
    return y, pk


w = torch.from_numpy(gen.conv_weight("kaiming", 3, (64, 3, 7, 7))).to(dev)
a = (torch.rand(64, device=dev) + 0.5) * torch.where(torch.arange(64, device=dev) % 7 == 0, -1.0, 1.0)
b = torch.randn(64, device=dev) * 0.3
bad = 0
for shape in [(2, 3, 224, 224), (3, 3, 64, 64), (1, 3, 32, 32), (2, 3, 50, 38), (1, 3, 97, 131), (5, 3, 33, 65),
              (1, 3, 7, 9), (1, 3, 225, 223), (9, 3, 112, 112)]:
    x = torch.from_numpy(gen.normal(gen.seed_of("stemab", shape), shape)).to(dev)
    for kw in ({}, {"fp16": True}):
        y0, p0 = run(True, x, w, a, b, **kw)
        y1, p1 = run(False, x, w, a, b, **kw)
        ref = F.max_pool2d(F.relu(F.conv2d(x.double(), w.double(), None, 2, 3) * a.double().view(1, -1, 1, 1)
                                  + b.double().view(1, -1, 1, 1)), 3, 2, 1)
        e0 = float((y0.double() - ref).abs().max() / ref.abs().max())
        e1 = float((y1.double() - ref).abs().max() / ref.abs().max())
        same = torch.equal(y0, y1) and torch.equal(p0.P, p1.P) and torch.equal(p0.M, p1.M)
        nd = int((y0 != y1).sum())
        print(shape, kw, "bit-identical" if same else "DIFFERENT (%d values, planes equal %s)" % (
            nd, torch.equal(p0.P, p1.P)), "rel err legacy %.2e new %.2e" % (e0, e1), flush=True)
        if e1 > (2e-3 if kw else 2e-6):
            bad += 1
            err = (y1.double() - ref).abs().amax(dim=(0, 1))
            ys, xs = torch.nonzero(err > 1e-4 * float(ref.abs().max()), as_tuple=True)
            print("   bad pooled rows", sorted(set(ys.tolist()))[:24], "cols", sorted(set(xs.tolist()))[:24])
            ch = (y1.double() - ref).abs().amax(dim=(0, 2, 3))
            print("   bad channels", torch.nonzero(ch > 1e-4 * float(ref.abs().max())).flatten().tolist()[:64])
print("BAD SHAPES:", bad)

N = int(os.environ.get("BATCH", "256"))
x = torch.from_numpy(gen.normal(1, (8, 3, 224, 224))).to(dev).repeat(N // 8, 1, 1, 1)


def t(fn, n=100):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for _ in range(600):
    hipops.stem7x7(x, w, a, b)
torch.cuda.synchronize()
for rep in range(2):
    for legacy in (True, False):
        for name, kw in (("split", {}), ("fp16", {"fp16": True})):
            if legacy:      # (the test-only binding always writes both outputs)
                print("%-7s %-6s full %.1f us" % ("legacy", name, t(lambda: legacy_lib.stem_staged(x, w, a, b, **kw))),
                      flush=True)
                continue
            print("%-7s %-6s full %.1f us   packed-only %.1f us   f32-only %.1f us" % (
                "rows", name, t(lambda: hipops.stem7x7(x, w, a, b, **kw)),
                t(lambda: hipops.stem7x7(x, w, a, b, out_f32=False, **kw)),
                t(lambda: hipops.stem7x7(x, w, a, b, out_packed=False, **kw))), flush=True)
