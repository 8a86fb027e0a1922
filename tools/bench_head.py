"""Head kernel (avgpool + fc) at batch 256."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch
from bnn_amd import native
lib = native.require()
N, C, HW, O = int(os.environ.get("BATCH", "256")), 512, 49, 1000
dev = torch.device("cuda:0")
x = torch.rand(N, C, 7, 7, device=dev); wt = torch.randn(C, O, device=dev) * 0.05; b = torch.randn(O, device=dev)
out = torch.empty(N, O, device=dev)
s = torch.cuda.current_stream().cuda_stream
def t(fn, n=200):
    for _ in range(50): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
nbytes = int(lib.bnn_hip_avgpool_fc_workspace_bytes(N, C))
ws = torch.empty(nbytes // 4, device=dev)
out2 = torch.empty(N, O, device=dev)
for _ in range(3):
    a = t(lambda: lib.bnn_hip_avgpool_fc_f32(x.data_ptr(), N, C, HW, wt.data_ptr(), b.data_ptr(), O, out.data_ptr(), s))
    a2 = t(lambda: lib.bnn_hip_avgpool_fc_ws_f32(x.data_ptr(), N, C, HW, wt.data_ptr(), b.data_ptr(), O, out2.data_ptr(),
                                                 ws.data_ptr(), nbytes, s))
    print("batch %d: avgpool + fc, one kernel %.1f us   two launches through a workspace %.1f us   (max |diff| %.2e)"
          % (N, a, a2, float((out - out2).abs().max())))
