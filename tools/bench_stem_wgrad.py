"""Stem weight gradient at batch 256, 224 x 224: bnn_hip_stem7x7_wgrad_f32 vs aten::convolution_backward (weight only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
from bnn_amd import hipops
dev = torch.device("cuda:0")
N = int(os.environ.get("BATCH", "256"))
x = torch.randn(N, 3, 224, 224, device=dev); dy = torch.randn(N, 64, 112, 112, device=dev); w = torch.randn(64, 3, 7, 7, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
lib = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])
print("batch %d: stem7x7_wgrad %.1f us   library %.1f us   (2 * 64 * 147 * N * 112 * 112 = %.1f GFLOP; dy %.0f MB)" % (
    N, t(lambda: hipops.stem7x7_wgrad(x, dy)), t(lib), 2 * 64 * 147 * N * 112 * 112 / 1e9, dy.numel() * 4 / 1e6))
