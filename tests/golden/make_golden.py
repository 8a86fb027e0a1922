#!/usr/bin/env python3
"""Generate the golden fixtures by running the REFERENCE implementation.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite use the
committed ``*.npz`` / ``*.json`` outputs.  Nothing of the reference's source is stored — only
inputs' generator parameters (tests/golden/cases.py, gen.py) and the reference's numeric outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("BNN_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REFERENCE)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import bnn  # the reference package  # noqa: E402
from bnn.ops import BasicInputBinarizer, BasicScaleBinarizer, XNORWeightBinarizer  # noqa: E402
from bnn.models.resnet import resnet18 as ref_resnet18  # noqa: E402
from bnn.models.layers import Bottleneck as RefBottleneck, HBlock as RefHBlock, PreBasicBlock as RefPreBasicBlock  # noqa: E402

from tests.golden import gen  # noqa: E402
from tests.golden.cases import LAYER_CASES, LINEAR_CASES, grad_case  # noqa: E402

assert os.path.realpath(bnn.__file__).startswith(os.path.realpath(REFERENCE)), bnn.__file__
torch.set_num_threads(8)


def t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a))


def ref_layer_forward(case):
    """One reference bnn.layers.Conv2d forward (bnn/layers/conv.py:90-97) for a LayerCase."""
    x, w, b, sc = case.tensors()
    conv = nn.Conv2d(case.C, case.O, case.k, stride=case.stride, padding=case.pad,
                     dilation=case.dilation, bias=case.bias)
    conv.weight.data.copy_(t(w))
    if b is not None:
        conv.bias.data.copy_(t(b))
    cfg = bnn.BConfig(
        activation_pre_process=BasicInputBinarizer,
        activation_post_process=BasicScaleBinarizer if case.post == "scale" else bnn.Identity,
        weight_pre_process=XNORWeightBinarizer.with_args(compute_alpha=case.compute_alpha,
                                                         center_weights=case.center))
    layer = bnn.prepare_binary_model(conv, cfg)
    assert type(layer) is bnn.layers.Conv2d
    if sc is not None:
        layer.activation_post_process.alpha.data.copy_(t(sc).view(1, -1, 1, 1))
    with torch.no_grad():
        out = layer(t(x)).numpy().copy()
        # the integer dot straight from the reference's own ops: conv(sign(x), sign(W - mean))
        xs = layer.activation_pre_process(t(x))
        wsgn = XNORWeightBinarizer(compute_alpha=False, center_weights=case.center)(layer.weight)
        dot = torch.nn.functional.conv2d(xs.double(), wsgn.double(), None, case.stride, case.pad,
                                         case.dilation).numpy()
    assert np.array_equal(dot, np.round(dot)) or np.isnan(dot).any()
    return out, dot.astype(np.int32)


def make_layers():
    blob = {}
    for case in LAYER_CASES:
        out, dot = ref_layer_forward(case)
        blob[case.name + "/out"] = out
        blob[case.name + "/dot"] = dot.astype(np.int16)
        print(f"layer {case.name:22s} out{out.shape} |max|={np.nanmax(np.abs(out)):.4f}")
    for case in LINEAR_CASES:
        x, w, b, sc = case.tensors()
        lin = nn.Linear(case.F, case.O, bias=case.bias)
        lin.weight.data.copy_(t(w))
        if b is not None:
            lin.bias.data.copy_(t(b))
        cfg = bnn.BConfig(
            activation_pre_process=BasicInputBinarizer,
            activation_post_process=BasicScaleBinarizer if case.post == "scale" else bnn.Identity,
            weight_pre_process=XNORWeightBinarizer.with_args(center_weights=case.center))
        layer = bnn.prepare_binary_model(lin, cfg)
        if sc is not None:
            layer.activation_post_process.alpha.data.copy_(t(sc).view(1, -1))
        with torch.no_grad():
            blob["linear/" + case.name + "/out"] = layer(t(x)).numpy().copy()
        print(f"linear {case.name}")
    np.savez_compressed(os.path.join(HERE, "layers.npz"), **blob)


def make_ref_test_layers():
    """G1: the reference's own known-answer vectors (test/test_layers.py:22-25, :37, :47-49, :59-66),
    re-evaluated through the reference so that expected == what its CI asserts (atol 1e-4)."""
    data = np.array([-0.05263, -0.05068, -0.03849, 0.03104, 0.0772, 0.03038, -0.06640, 0.05894,
                     0.13059, 0.03433, -0.25811, 0.13785], np.float32).reshape(1, 3, 2, 2)
    weights = np.array([-0.0252, 0.0084, -0.0676, 0.0891, -0.0010, 0.0518, 0.0380, 0.2866, -0.0050],
                       np.float32)
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer,
                      activation_post_process=BasicScaleBinarizer,
                      weight_pre_process=XNORWeightBinarizer)
    lin = nn.Linear(3, 3, bias=False); lin.weight.data.copy_(t(weights).view(3, 3))
    c1 = nn.Conv1d(3, 3, 1, bias=False); c1.weight.data.copy_(t(weights).view(3, 3, 1))
    c2 = nn.Conv2d(3, 3, 1, bias=False); c2.weight.data.copy_(t(weights).view(3, 3, 1, 1))
    with torch.no_grad():
        o_lin = bnn.prepare_binary_model(lin, cfg)(t(data)[:, :, 0, 0].reshape(1, 3)).numpy()
        o_c1 = bnn.prepare_binary_model(c1, cfg)(t(data)[:, :, :, 0].reshape(1, 3, 2)).numpy()
        o_c2 = bnn.prepare_binary_model(c2, cfg)(t(data)).numpy()
    # the literals asserted by the reference's test file
    exp_lin = np.array([[0.0337, -0.0473, -0.1099]], np.float32)
    exp_c1 = np.array([[[0.0337, 0.0337], [-0.0473, -0.0473], [-0.1099, -0.1099]]], np.float32)
    exp_c2 = np.array([[[[0.0337, 0.0337], [0.0337, -0.0337]], [[-0.0473, -0.0473], [-0.0473, 0.0473]],
                        [[-0.1099, -0.1099], [-0.1099, 0.1099]]]], np.float32)
    assert np.allclose(o_lin, exp_lin, atol=1e-4) and np.allclose(o_c1, exp_c1, atol=1e-4) \
        and np.allclose(o_c2, exp_c2, atol=1e-4)
    np.savez_compressed(os.path.join(HERE, "ref_test_layers.npz"), data=data, weights=weights,
                        linear=o_lin, conv1d=o_c1, conv2d=o_c2, lit_linear=exp_lin, lit_conv1d=exp_c1,
                        lit_conv2d=exp_c2)
    print("ref_test_layers ok")


def load_state(model, seed):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    st = gen.model_state(shapes, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    return model


def xnor_cfg():
    return bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                       weight_pre_process=XNORWeightBinarizer)


def make_resnet18():
    """G4: examples/cifar10.py:61-71 model (resnet18 + XNOR BConfig, conv1/fc real-valued)."""
    net = ref_resnet18()
    net = bnn.prepare_binary_model(net, xnor_cfg(), custom_config_layers_name={"conv1": bnn.BConfig(),
                                                                               "fc": bnn.BConfig()})
    load_state(net, seed=1)
    net.eval()
    blob = {}
    for tag, shape in (("32", (4, 3, 32, 32)), ("64", (2, 3, 64, 64)), ("224", (2, 3, 224, 224))):
        x = gen.normal(gen.seed_of("r18", tag), shape)
        with torch.no_grad():
            y = net(t(x)).numpy()
        blob["logits_" + tag] = y
        print("resnet18", tag, y.shape, float(np.abs(y).max()))
    # BASELINE config 1 at its stated size (examples/cifar10.py model, 32 x 32 inputs, batch 32): the four batches
    # bench.py's gpu_c1 leg feeds (gen.normal(40 + j, ...)), logits + sign checksums in front of the 19 binary convolutions
    for j in range(4):
        y, names, h = _binary_inputs(net, gen.normal(40 + j, (32, 3, 32, 32)), batch=32)
        blob["c1_logits_%d" % j] = y
        blob["c1_sign_hash_%d" % j] = h
        blob["c1_layers"] = np.array(names)
        print("resnet18 c1 batch", j, y.shape, h.shape)
    blob["state_keys"] = np.array(list(net.state_dict().keys()))
    blob["module_types"] = np.array([f"{n}:{type(m).__name__}" for n, m in net.named_modules()])
    np.savez_compressed(os.path.join(HERE, "resnet18.npz"), **blob)


def _binary_inputs(net, x, batch=32):
    """Forward in mini-batches, returning (logits, {layer name: per-image sign checksum}) where the
    checksums are tests/golden/sighash.py of every BINARY convolution's input, in execution order."""
    from tests.golden import sighash
    names, hashes = [], {}
    hooks = []
    for name, mod in net.named_modules():
        if isinstance(mod, bnn.layers.Conv2d) and isinstance(mod.activation_pre_process, BasicInputBinarizer):
            def pre(m, inp, name=name):
                if name not in hashes:
                    names.append(name)
                    hashes[name] = []
                hashes[name].append(sighash.sign_hash_torch(inp[0]).numpy())
            hooks.append(mod.register_forward_pre_hook(pre))
    outs = []
    with torch.no_grad():
        for i in range(0, x.shape[0], batch):
            outs.append(net(t(x[i:i + batch])).numpy())
    for h in hooks:
        h.remove()
    return np.concatenate(outs), names, np.stack([np.concatenate(hashes[n]) for n in names], 1)


def make_resnet18_b256():
    """G4 at the size the headline metric is quoted on (BASELINE config 3): reference logits of 256
    distinct 224x224 images + the discrete state (sign checksums of all 19 binary-conv inputs) per image.
    Also measures how far the reference moves from ITSELF when only its convolution backend changes
    (oneDNN vs ATen's native im2col+GEMM): the noise floor any other implementation is judged against."""
    net = ref_resnet18()
    net = bnn.prepare_binary_model(net, xnor_cfg(), custom_config_layers_name={"conv1": bnn.BConfig(),
                                                                               "fc": bnn.BConfig()})
    load_state(net, seed=1)
    net.eval()
    x = gen.normal(gen.seed_of("r18", "b256"), (256, 3, 224, 224))
    y, names, h = _binary_inputs(net, x)
    with torch.backends.mkldnn.flags(enabled=False):
        y2, _, h2 = _binary_inputs(net, x)
    y64, _, h64 = _binary_inputs(net.double(), x.astype(np.float64), batch=16)   # the same model evaluated in fp64
    net.float()
    y64 = y64.astype(np.float32)

    def compare(ya, ha, yb, hb):
        ok = np.all(np.abs(ya - yb) <= 1e-3 * np.abs(yb).max() + 1e-3 * np.abs(yb), 1)
        flipped = np.any(ha != hb, 1)
        first = [int(np.argmax(ha[i] != hb[i])) for i in np.nonzero(flipped)[0]]
        return {"images": int(ya.shape[0]), "within_tol": int(ok.sum()), "images_with_a_sign_flip": int(flipped.sum()),
                "max_abs_logit_dev": float(np.abs(ya - yb).max()),
                "max_dev_without_flip": float(np.abs(ya - yb)[~flipped].max()) if (~flipped).any() else 0.0,
                "first_diverging_layer_histogram": {names[k]: first.count(k) for k in sorted(set(first))}}
    self_check = {
        "max_abs_logit": float(np.abs(y).max()), "torch": torch.__version__,
        "tolerance": "per image: |a - b| <= 1e-3*max|b| + 1e-3*|b| on all 1000 logits",
        "ref_onednn_vs_ref_native_conv": compare(y2, h2, y, h),
        "ref_fp32_vs_ref_fp64": compare(y, h, y64, h64),
        "ref_native_conv_vs_ref_fp64": compare(y2, h2, y64, h64),
    }
    print(json.dumps(self_check, indent=1))
    np.savez_compressed(os.path.join(HERE, "resnet18_b256.npz"), logits=y, sign_hash=h, logits_f64=y64,
                        sign_hash_f64=h64, layers=np.array(names), self_check=np.array(json.dumps(self_check)))
    print("resnet18_b256", y.shape, h.shape, names)


def make_blocks():
    """G5: module-level outputs of the blocks that call the hot path (config 5 building blocks)."""
    blob = {}
    specs = {
        "hblock_256": (lambda: RefHBlock(256, 256, norm_layer=nn.BatchNorm2d), (2, 256, 8, 8)),
        "bottleneck_256_64": (lambda: RefBottleneck(256, 64), (2, 256, 8, 8)),
        "prebasic_64": (lambda: RefPreBasicBlock(64, 64), (2, 64, 10, 10)),
        "prebasic_64_prelu": (lambda: RefPreBasicBlock(64, 64, activation=nn.PReLU), (2, 64, 10, 10)),
    }
    for name, (ctor, shape) in specs.items():
        blk = bnn.prepare_binary_model(ctor(), xnor_cfg())
        load_state(blk, seed=gen.seed_of("block", name))
        blk.eval()
        x = gen.normal(gen.seed_of("blockx", name), shape)
        with torch.no_grad():
            blob[name] = blk(t(x)).numpy()
        print("block", name, blob[name].shape)
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **blob)


def make_convert():
    """G6: conversion behaviour of prepare_binary_model incl. the crossed _first_/_last_ words."""
    def small():
        return nn.Sequential(nn.Conv2d(3, 16, 1, 1), nn.BatchNorm2d(16), nn.ReLU(inplace=True),
                             nn.Conv2d(16, 16, 1, 1), nn.BatchNorm2d(16), nn.ReLU(inplace=True),
                             nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Linear(16, 3))
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer,
                      activation_post_process=BasicScaleBinarizer, weight_pre_process=XNORWeightBinarizer)

    def describe(m):
        return {n: [type(c).__module__.split(".")[0] == "bnn", type(c).__name__,
                    type(getattr(c, "activation_pre_process", None)).__name__]
                for n, c in m.named_modules() if isinstance(c, (nn.Conv2d, nn.Linear))}
    out = {}
    out["plain"] = describe(bnn.prepare_binary_model(small(), cfg))
    out["ignore_first_word"] = describe(bnn.prepare_binary_model(small(), cfg, ignore_layers_name=["_first_"]))
    out["ignore_last_word"] = describe(bnn.prepare_binary_model(small(), cfg, ignore_layers_name=["_last_"]))
    out["ignore_regex"] = describe(bnn.prepare_binary_model(small(), cfg, ignore_layers_name=["$^[03]$$"]))
    out["ignore_literal"] = describe(bnn.prepare_binary_model(small(), cfg, ignore_layers_name=["8"]))
    fp32 = bnn.BConfig(activation_pre_process=nn.Identity, activation_post_process=nn.Identity,
                       weight_pre_process=nn.Identity)
    out["custom_fp32_8"] = describe(bnn.prepare_binary_model(small(), cfg, custom_config_layers_name={"8": fp32}))
    out["state_keys"] = list(bnn.prepare_binary_model(small(), cfg).state_dict().keys())
    with open(os.path.join(HERE, "convert.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("convert ok")


IMAGENET_STEP1_IGNORED = ["_last_", "_first_", "layer2.0.downsample.1", "layer3.0.downsample.1",
                          "layer4.0.downsample.1"]


def imagenet_step1_model():
    """The reference's OTHER shipping dataflow (examples/imagenet.py:153-155 with step 1 of
    examples/recepies/imagenet-baseline.yaml:16-31): resnet18(PreBasicBlock, PReLU), BasicInputBinarizer /
    XNORWeightBinarizer(compute_alpha=False, center_weights=False) / BasicScaleBinarizer, first / last layer and
    the three down-sampling 1x1 convolutions real-valued.  (BinaryChef itself needs `easydict`, which this
    container lacks; what it evaluates for step 1 is exactly this prepare_binary_model call: bnn/engine.py:51-75.)"""
    net = ref_resnet18(stem_type="basic", num_classes=1000, block_type=RefPreBasicBlock, activation=nn.PReLU)
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=BasicScaleBinarizer,
                      weight_pre_process=XNORWeightBinarizer.with_args(compute_alpha=False, center_weights=False))
    net = bnn.prepare_binary_model(net, cfg, ignore_layers_name=list(IMAGENET_STEP1_IGNORED))
    load_state(net, seed=7)
    return net.eval()


def make_prenet():
    """G7: logits + per-image sign checksums of every binary convolution's input for the pre-activation PReLU
    ResNet-18 of examples/imagenet.py, at 64x64 and 224x224 (2 images each)."""
    net = imagenet_step1_model()
    blob = {}
    for tag, shape in (("64", (2, 3, 64, 64)), ("224", (2, 3, 224, 224))):
        x = gen.normal(gen.seed_of("prenet18", tag), shape)
        y, names, h = _binary_inputs(net, x)
        blob["logits_" + tag], blob["sign_hash_" + tag] = y, h
        print("prenet18", tag, y.shape, h.shape, float(np.abs(y).max()))
    blob["layers"] = np.array(names)
    blob["state_keys"] = np.array(list(net.state_dict().keys()))
    blob["float_convs"] = np.array([n for n, m in net.named_modules()
                                    if type(m) is nn.Conv2d or type(m) is nn.Linear])
    np.savez_compressed(os.path.join(HERE, "prenet18.npz"), **blob)


def make_stacks():
    """G8: the cross-block packed dataflow of config 5's building blocks: nn.Sequential of three reference
    HBlock(256, 256) (hierarchical_block.py:38-60) and of two Bottleneck(256, 64) (res_block.py:98-118) at 28x28,
    outputs + the sign checksums in front of every binary convolution."""
    blob = {}
    specs = {
        "hblock_x3": lambda: nn.Sequential(*[RefHBlock(256, 256, norm_layer=nn.BatchNorm2d) for _ in range(3)]),
        "bottleneck_x2": lambda: nn.Sequential(*[RefBottleneck(256, 64) for _ in range(2)]),
    }
    for name, ctor in specs.items():
        net = bnn.prepare_binary_model(ctor(), xnor_cfg())
        load_state(net, seed=gen.seed_of("stack", name))
        net.eval()
        x = gen.normal(gen.seed_of("stackx", name), (2, 256, 28, 28))
        y, names, h = _binary_inputs(net, x)
        # (fixture size: every 4th output channel in full + the per-channel sums of all of them)
        blob[name + "/out_c4"], blob[name + "/out_sum"] = y[:, ::4].copy(), y.astype(np.float64).sum((2, 3))
        blob[name + "/sign_hash"], blob[name + "/layers"] = h, np.array(names)
        print("stack", name, y.shape, h.shape, float(np.abs(y).max()))
    np.savez_compressed(os.path.join(HERE, "stacks.npz"), **blob)


class RefHBlockNet(nn.Module):
    """BASELINE config 5's network, which the reference cannot construct itself (HBlock has no
    `expansion` and rejects stride > 1), assembled here from the REFERENCE's own modules in the layout the build
    defines (bnn_amd/models/resnet.py: ResNet(HBlock, [3, 4, 6, 3])): the reference's stem (resnet.py:93-96,150-153),
    per stage an AvgPool2d(2, ceil, no pad count) in front of stride-2 stages, the reference's HBlock
    (hierarchical_block.py:8-60) with a BN -> conv1x1 shortcut where the width changes, the reference's head
    (resnet.py:160-164).  Same attribute names and Sequential indices as the build's model, so that the seeded state of
    tests/golden/gen.py maps key by key."""

    def __init__(self, depths=(3, 4, 6, 3)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = 64
        for i, (planes, blocks) in enumerate(zip((64, 128, 256, 512), depths)):
            stride = 1 if i == 0 else 2
            stage = []
            if stride != 1:
                stage.append(nn.AvgPool2d(kernel_size=stride, stride=stride, ceil_mode=True, count_include_pad=False))
            shortcut = None
            if stride != 1 or inplanes != planes:
                shortcut = nn.Sequential(nn.BatchNorm2d(inplanes), nn.Conv2d(inplanes, planes, 1, bias=False))
            stage.append(RefHBlock(inplanes, planes, 1, shortcut, norm_layer=nn.BatchNorm2d))
            inplanes = planes
            for _ in range(1, blocks):
                stage.append(RefHBlock(inplanes, planes, norm_layer=nn.BatchNorm2d))
            setattr(self, f"layer{i + 1}", nn.Sequential(*stage))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def make_hblock_net():
    """G10: config 5 at its real size pinned to the reference's modules (task 6; round 5: 32 images): 32 images 224x224
    through RefHBlockNet([3, 4, 6, 3]) — logits + the sign checksums in front of all 51 binary convolutions, the same
    scheme as resnet18_b256.npz — and the reference against itself (other conv backend, fp64)."""
    net = bnn.prepare_binary_model(RefHBlockNet(), xnor_cfg(), custom_config_layers_name={"conv1": bnn.BConfig(),
                                                                                         "fc": bnn.BConfig()})
    load_state(net, seed=1)
    net.eval()
    x = gen.normal(gen.seed_of("c5", "b128"), (128, 3, 224, 224))[:32].copy()     # the first 32 images of the c5 test batch
    y, names, h = _binary_inputs(net, x, batch=4)
    with torch.backends.mkldnn.flags(enabled=False):
        y2, _, h2 = _binary_inputs(net, x, batch=4)
    y64, _, h64 = _binary_inputs(net.double(), x.astype(np.float64), batch=4)
    net.float()
    y64 = y64.astype(np.float32)

    def compare(ya, ha, yb, hb):
        flipped = np.any(ha != hb, 1)
        ok = np.all(np.abs(ya - yb) <= 1e-3 * np.abs(yb).max() + 1e-3 * np.abs(yb), 1)
        return {"images": int(ya.shape[0]), "within_tol": int(ok.sum()), "images_with_a_sign_flip": int(flipped.sum()),
                "max_abs_logit_dev": float(np.abs(ya - yb).max()),
                "max_dev_without_flip": float(np.abs(ya - yb)[~flipped].max()) if (~flipped).any() else 0.0}
    self_check = {"max_abs_logit": float(np.abs(y).max()), "torch": torch.__version__,
                  "ref_onednn_vs_ref_native_conv": compare(y2, h2, y, h), "ref_fp32_vs_ref_fp64": compare(y, h, y64, h64)}
    print(json.dumps(self_check, indent=1))
    assert len(names) == 3 * 16 + 3, names
    np.savez_compressed(os.path.join(HERE, "hblock_net_b32.npz"), logits=y, sign_hash=h, logits_f64=y64, sign_hash_f64=h64,
                        layers=np.array(names), state_keys=np.array(list(net.state_dict().keys())),
                        self_check=np.array(json.dumps(self_check)))
    print("hblock_net_b32", y.shape, h.shape)


GRAD_CASES = ("c2_relu", "l2_0_c1_s2", "l3_ds_1x1")


def make_grads():
    """G9: the reference's OWN autograd through one binary layer (straight-through estimator on the activations
    bnn/ops.py:68-73, on the weights through bnn/ops.py:136, the alpha path bnn/ops.py:116-127, the post scale
    bnn/ops.py:200-202): dL/dx, dL/dW, dL/d(post alpha) for L = sum(out * G), at N = 2."""
    blob = {}
    for name in GRAD_CASES:
        case = grad_case(name)
        x, w, b, sc = case.tensors()
        x = (0.9 * x).astype(np.float32)            # part of the activations inside the STE window |x| < 1
        conv = nn.Conv2d(case.C, case.O, case.k, stride=case.stride, padding=case.pad, bias=False)
        conv.weight.data.copy_(t(w))
        cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=BasicScaleBinarizer,
                          weight_pre_process=XNORWeightBinarizer)
        layer = bnn.prepare_binary_model(conv, cfg).train()
        layer.activation_post_process.alpha.data.copy_(t(sc).view(1, -1, 1, 1))
        xt = t(x).requires_grad_(True)
        out = layer(xt)
        G = gen.normal(gen.seed_of("gradG", name), tuple(out.shape))
        (out * t(G)).sum().backward()
        # (x, G and the weights are regenerated by the tests from the same seeds; dW: every 2nd output channel)
        blob[name + "/dx"] = xt.grad.numpy()
        blob[name + "/dw_o2"] = layer.weight.grad.numpy()[::2].copy()
        blob[name + "/dw_sum"] = layer.weight.grad.numpy().astype(np.float64).sum((1, 2, 3))
        blob[name + "/dscale"] = layer.activation_post_process.alpha.grad.numpy()
        print("grad", name, out.shape, float(np.abs(blob[name + "/dx"]).max()), float(np.abs(blob[name + "/dw_o2"]).max()))
    np.savez_compressed(os.path.join(HERE, "grads.npz"), **blob)


ALL = {"ref_test_layers": make_ref_test_layers, "layers": make_layers, "resnet18": make_resnet18,
       "resnet18_b256": make_resnet18_b256, "blocks": make_blocks, "convert": make_convert,
       "prenet": make_prenet, "stacks": make_stacks, "grads": make_grads, "hblock_net": make_hblock_net}

if __name__ == "__main__":
    for which in (sys.argv[1:] or list(ALL)):   # no argument: every fixture
        ALL[which]()
