"""images/s of the fused executor on the other block families (BASELINE config 5 style nets), batch 128, 224x224."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import torch, torch.nn as nn
import bnn_amd as bnn
from bnn_amd.inference import FusedResNet
from bnn_amd.models import HBlock, PreBasicBlock, ResNet, resnet18, resnet50
from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer
from tests.golden import gen
dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", "128"))
NETS = {
    "resnet18 BasicBlock": lambda: resnet18(),
    "resnet18 PreBasicBlock+PReLU": lambda: resnet18(block_type=PreBasicBlock, activation=nn.PReLU),
    "resnet50 Bottleneck": lambda: resnet50(),
    "ResNet(HBlock,[3,4,6,3])": lambda: ResNet(HBlock, [3, 4, 6, 3]),
}
cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                  weight_pre_process=XNORWeightBinarizer)
x = torch.from_numpy(gen.normal(1, (8, 3, 224, 224))).to(dev).repeat(B // 8, 1, 1, 1)
for name, ctor in NETS.items():
    if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
        continue
    net = bnn.prepare_binary_model(ctor(), cfg, custom_config_layers_name={"conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 1).items()})
    net = net.to(dev).eval()
    def timeit(f, n):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
    with torch.no_grad():
        t_layer = timeit(lambda: net(x), 5)
    fused = FusedResNet(net).capture(x)
    t_fused = timeit(lambda: fused(fused.static_input), 20)
    print("%-32s batch %d: fused graph %8.0f img/s (%.2f ms)   per-layer path %7.0f img/s" % (name, B, B / t_fused, t_fused * 1e3, B / t_layer))
