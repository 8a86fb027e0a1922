import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "binary-networks-pytorch_amd")]
import numpy as np, torch, torch.nn as nn
import bnn_amd as bnn
from bnn_amd.models import ResNet, HBlock
from bnn_amd.ops import *
from tests.golden import gen
net = ResNet(HBlock, [1, 2, 1, 1], num_classes=100)
cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=BasicScaleBinarizer,
                  weight_pre_process=XNORWeightBinarizer.with_args(center_weights=True))
net = bnn.prepare_binary_model(net, cfg, ignore_layers_name=["_first_", "_last_"])
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 5).items()})
net.eval()
x = torch.from_numpy(gen.normal(gen.seed_of("c5", "resnet_hblock"), (2, 3, 64, 64)))
rec = {}
def hook(name):
    def f(m, i, o): rec.setdefault(name, []).append((i[0].detach().cpu().clone(), o.detach().cpu().clone()))
    return f
for n, m in net.named_modules():
    if isinstance(m, bnn.layers.Conv2d): m.register_forward_hook(hook(n))
with torch.no_grad():
    net(x); net.cuda()(x.cuda())
for n, (a, b) in rec.items():
    din = (a[0] - b[0]).abs().max().item(); dout = (a[1] - b[1]).abs().max().item()
    m = dict(net.named_modules())[n]
    print(f"{n:28s} in_diff {din:.3e} out_diff {dout:.3e} |out| {a[1].abs().max():.2f} w{tuple(m.weight.shape)} s{m.stride} p{m.padding} in{tuple(a[0].shape)}")
print("---- isolate layer2.1.downsample.0")
import oracle
from bnn_amd import hipops, fastpath
m = dict(net.named_modules())["layer2.1.downsample.0"]
a_in = rec["layer2.1.downsample.0"][0][0]
w = m.weight.detach().cpu().numpy(); sc = m.activation_post_process.alpha.detach().cpu().numpy().reshape(-1)
ref, dot = oracle.binary_conv2d_int(a_in.numpy(), w, None, sc, center=True)
cpu_out = rec["layer2.1.downsample.0"][0][1].numpy(); gpu_out = rec["layer2.1.downsample.0"][1][1].numpy()
print("oracle vs cpu", np.abs(ref - cpu_out).max(), "oracle vs gpu", np.abs(ref - gpu_out).max())
plan = fastpath._recognise(m, 128); print(plan.center, plan.compute_alpha, None if plan.scale is None else plan.scale.shape)
pw = fastpath.packed_weight(m, plan); print("has_zero", pw.has_zero)
wb, wz, al, anyz = oracle.pack_weight(w, True, True)
print("bits eq", np.array_equal(pw.wbits.cpu().numpy().view(np.uint32), wb), "alpha eq", np.array_equal(pw.alpha.cpu().numpy(), al), anyz)
act = hipops.pack_act(a_in.cuda())
for kw in (dict(), dict(force_generic=True)):
    o = hipops.bconv2d(act, pw, None, plan.scale, **kw).cpu().numpy()
    print(kw, np.abs(o - ref).max())
o = hipops.bconv2d(act, pw, None, None).cpu().numpy() * sc.reshape(1, -1, 1, 1)
print("no-scale then mul", np.abs(o - ref).max())
d = np.abs(gpu_out - cpu_out)
idx = np.argwhere(d > 1e-3)
print("n mismatches", len(idx), "of", d.size)
print("channels:", sorted(set(idx[:,1].tolist()))[:40])
print("n:", sorted(set(idx[:,0].tolist())), "ys:", sorted(set(idx[:,2].tolist())), "xs:", sorted(set(idx[:,3].tolist())))
print("ratio gpu/cpu at mismatches:", (gpu_out[d>1e-3] / cpu_out[d>1e-3])[:10])
# run the net on GPU a second time: same result?
rec.clear()
with torch.no_grad(): net(x.cuda())
g2 = rec["layer2.1.downsample.0"][0][1].numpy()
print("second GPU run vs oracle", np.abs(g2 - ref).max(), "vs first GPU run", np.abs(g2 - gpu_out).max())
