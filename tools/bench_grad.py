"""Gradient kernels of the binary 3x3 conv, one line per ResNet-18 layer shape at batch 256 (BATCH=...):
dgrad / wgrad time, fraction of the bf16 MFMA peak (3 products per MAC), and the library's fp32 backward beside it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
from bnn_amd import hipops

dev = torch.device("cuda:0")
N = int(os.environ.get("BATCH", "256"))
LIB = os.environ.get("LIB", "1") == "1"
SHAPES = [  # (C, O, H, stride)
    (64, 64, 56, 1), (64, 128, 56, 2), (128, 128, 28, 1), (128, 256, 28, 2), (256, 256, 14, 1),
    (256, 512, 14, 2), (512, 512, 7, 1)]


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


x0 = torch.randn(64, 64, 56, 56, device=dev)
for _ in range(200):  # warm clocks
    x0 = x0 * 1.0001
torch.cuda.synchronize()
print("%-28s %9s %7s %9s %7s %9s %9s" % ("shape", "dgrad us", "frac", "wgrad us", "frac", "lib dgrad", "lib wgrad"))
if os.environ.get("ONLY"):  # e.g. ONLY=0,6 : PMC passes on a few shapes
    SHAPES = [SHAPES[int(i)] for i in os.environ["ONLY"].split(",")]
for C, O, H, st in SHAPES:
    x = torch.randn(N, C, H, H, device=dev)
    Ho = (H - 1) // st + 1
    g = torch.randn(N, O, Ho, Ho, device=dev)
    w = torch.randn(O, C, 3, 3, device=dev)
    alpha = w.abs().mean(dim=(1, 2, 3), keepdim=True)
    what = torch.sign(w) * alpha
    packed, al = hipops.grad_pack_weight(what)
    td = t(lambda: hipops.bconv_grad_input(g, x, packed, al, 3, st))
    tw = t(lambda: hipops.bconv_grad_weight(g, x, 3, st))
    if os.environ.get("PACKED") == "1":   # the same gradients from the 3-bit saved state (round 4)
        sv = hipops.pack_act_ste(x)
        print("   packed state: pack_ste %.1f us (pack_act %.1f)   dgrad %.1f (fp32 x: %.1f)   wgrad %.1f (fp32 x: %.1f)" % (
            t(lambda: hipops.pack_act_ste(x)), t(lambda: hipops.pack_act(x)),
            t(lambda: hipops.bconv_grad_input(g, sv, packed, al, 3, st)), td,
            t(lambda: hipops.bconv_grad_weight(g, sv, 3, st)), tw))
    flop = 2.0 * N * C * O * 9 * Ho * Ho * 3  # three bf16 products per MAC
    fd = flop / (td * 1e-6) / 2.5e15  # (stride 2: the four parity classes together do exactly these MACs)
    fw = flop / (tw * 1e-6) / 2.5e15
    ld = lw = float("nan")
    if LIB:
        sx = torch.sign(x)
        ld = t(lambda: torch.ops.aten.convolution_backward(g, sx, what, None, [st, st], [1, 1], [1, 1], False, [0, 0], 1,
                                                            [True, False, False]), 5)
        lw = t(lambda: torch.ops.aten.convolution_backward(g, sx, what, None, [st, st], [1, 1], [1, 1], False, [0, 0], 1,
                                                            [False, True, False]), 5)
    print("%3d->%3d %2dx%-2d s%d            %9.1f %7.3f %9.1f %7.3f %9.1f %9.1f" % (C, O, H, H, st, td, fd, tw, fw, ld, lw))
