"""Training-step time of binary ResNet-18 @224 on one GPU: HIP forward + library backward vs pure composition."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch, torch.nn as nn
import bnn_amd as bnn
from bnn_amd import training, fastpath
from bnn_amd.models import resnet18
from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer
dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", "64"))
def run(enabled, steps=6, binary_grads=True):
    training.ENABLED = enabled
    training.BINARY_GRADS = binary_grads
    torch.manual_seed(0)
    net = resnet18(num_classes=1000)
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    net = bnn.prepare_binary_model(net, cfg, ignore_layers_name=["conv1", "fc"]).to(dev).train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
    x = torch.randn(B, 3, 224, 224, device=dev); t = torch.randint(0, 1000, (B,), device=dev)
    for i in range(steps + 2):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = nn.functional.cross_entropy(net(x), t)
        loss.backward(); opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
if os.environ.get("PACKED") is not None:     # PACKED=0: keep the fp32 input for the backward (round 3), 1: three bit planes
    training.PACKED_STATE = os.environ["PACKED"] == "1"
if os.environ.get("ONLY") == "mfma":
    print("batch %d: HIP forward + MFMA gradient kernels %.1f ms" % (B, run(True)))
    sys.exit(0)
if os.environ.get("PACKED") is None:   # both forms of the saved input state, with the bytes they keep per step
    training.PACKED_STATE = True
    training.saved_input_bytes(reset=True); ap = run(True, steps=4); kp = training.saved_input_bytes(reset=True) / 6
    training.PACKED_STATE = False
    training.saved_input_bytes(reset=True); af = run(True, steps=4); kf = training.saved_input_bytes(reset=True) / 6
    print("batch %d: saved input state of the 19 binary convs per step: fp32 x %.0f MB (step %.1f ms) | 3 bits per element "
          "(BNN_AMD_TRAIN_PACKED_STATE=1) %.0f MB (step %.1f ms): %.1fx less" % (B, kf / 1e6, af, kp / 1e6, ap, kf / kp))
a = run(True); a2 = run(True, binary_grads=False); b = run(False)
print("batch %d: training step  HIP forward + MFMA gradient kernels %.1f ms (%.0f img/s) | HIP forward + library backward "
      "%.1f ms (%.0f img/s) | composition %.1f ms (%.0f img/s)" % (B, a, B / a * 1e3, a2, B / a2 * 1e3, b, B / b * 1e3))
