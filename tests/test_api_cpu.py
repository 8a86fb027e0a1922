"""Host-side mirror of the reference API (BConfig / prepare_binary_model / layers / ops).

Modelled on the reference's own test/test_layers.py, test/test_binarize.py, plus conversion
fixtures captured from the reference (tests/golden/convert.json) and module outputs
(tests/golden/blocks.npz, resnet18.npz).  CPU tensors take the torch composition path.
"""
import copy
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import bnn_amd as bnn
from bnn_amd.layers import Conv1d, Conv2d, Linear
from bnn_amd.models import Bottleneck, HBlock, PreBasicBlock, resnet18
from bnn_amd.ops import (AdvancedInputBinarizer, BasicInputBinarizer, BasicScaleBinarizer,
                         SignActivation, StochasticInputBinarizer, XNORWeightBinarizer)
from tests.golden import gen
from tests.golden.cases import LAYER_CASES_BY_NAME

XNOR_SCALE = dict(activation_pre_process=BasicInputBinarizer,
                  activation_post_process=BasicScaleBinarizer,
                  weight_pre_process=XNORWeightBinarizer)


def small_net():
    return nn.Sequential(nn.Conv2d(3, 16, 1, 1), nn.BatchNorm2d(16), nn.ReLU(inplace=True),
                         nn.Conv2d(16, 16, 1, 1), nn.BatchNorm2d(16), nn.ReLU(inplace=True),
                         nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Linear(16, 3))


# ------------------------------------------------------------------ reference test_layers.py
@pytest.fixture(scope="module")
def ka(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_test_layers.npz"))


def test_linear_known_answer(ka):
    layer = nn.Linear(3, 3, bias=False)
    layer.weight.data.copy_(torch.from_numpy(ka["weights"]).view(3, 3))
    layer = bnn.prepare_binary_model(layer, bconfig=bnn.BConfig(**XNOR_SCALE))
    out = layer(torch.from_numpy(ka["data"])[:, :, 0, 0].reshape(1, 3))
    assert torch.allclose(out, torch.from_numpy(ka["lit_linear"]), atol=1e-4)
    assert torch.allclose(out, torch.from_numpy(ka["linear"]), atol=1e-6)


def test_conv1d_known_answer(ka):
    layer = nn.Conv1d(3, 3, 1, bias=False)
    layer.weight.data.copy_(torch.from_numpy(ka["weights"]).view(3, 3, 1))
    layer = bnn.prepare_binary_model(layer, bconfig=bnn.BConfig(**XNOR_SCALE))
    out = layer(torch.from_numpy(ka["data"])[:, :, :, 0].reshape(1, 3, 2))
    assert torch.allclose(out, torch.from_numpy(ka["lit_conv1d"]), atol=1e-4)


def test_conv2d_known_answer(ka):
    layer = nn.Conv2d(3, 3, 1, bias=False)
    layer.weight.data.copy_(torch.from_numpy(ka["weights"]).view(3, 3, 1, 1))
    layer = bnn.prepare_binary_model(layer, bconfig=bnn.BConfig(**XNOR_SCALE))
    out = layer(torch.from_numpy(ka["data"]))
    assert torch.allclose(out, torch.from_numpy(ka["lit_conv2d"]), atol=1e-4)
    assert torch.allclose(out, torch.from_numpy(ka["conv2d"]), atol=1e-6)


# ------------------------------------------------------------------ reference test_binarize.py
def test_single_layers_are_swapped():
    cfg = bnn.BConfig(**XNOR_SCALE)
    assert type(bnn.prepare_binary_model(nn.Linear(10, 3), bconfig=cfg)) is Linear
    assert type(bnn.prepare_binary_model(nn.Conv2d(3, 16, 1, 1), bconfig=cfg)) is Conv2d
    assert type(bnn.prepare_binary_model(nn.Conv1d(3, 16, 1, 1), bconfig=cfg)) is Conv1d


def test_skip_binarization_with_custom_config():
    fp32 = bnn.BConfig(activation_pre_process=nn.Identity, activation_post_process=nn.Identity,
                       weight_pre_process=nn.Identity)
    model = bnn.prepare_binary_model(small_net(), bconfig=bnn.BConfig(**XNOR_SCALE),
                                     custom_config_layers_name={"8": fp32})
    convs = [m for m in model.modules() if isinstance(m, Conv2d)]
    lins = [m for m in model.modules() if isinstance(m, Linear)]
    assert len(convs) == 2 and len(lins) == 1
    assert isinstance(lins[0].activation_pre_process, nn.Identity)
    assert all(isinstance(c.activation_pre_process, BasicInputBinarizer) for c in convs)


def test_state_dict_roundtrip_bit_equal():
    torch.manual_seed(0)
    net = small_net()
    x = torch.rand(1, 3, 8, 8)
    model = bnn.prepare_binary_model(copy.deepcopy(net), bconfig=bnn.BConfig(**XNOR_SCALE))
    out1 = model(x)
    sd = model.state_dict()
    fresh = copy.deepcopy(net)
    for m in fresh.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            m.reset_parameters()
    fresh = bnn.prepare_binary_model(fresh, bconfig=bnn.BConfig(**XNOR_SCALE))
    fresh.load_state_dict(sd)
    assert torch.equal(out1, fresh(x))


def test_basic_input_binarizer_is_torch_sign():
    x = torch.tensor([0.3, 0.1, -2, -0.001, 0.01, 0.0, -0.0, float("nan"), 1e-45, -1e-45])
    assert torch.equal(BasicInputBinarizer()(x.clone()), torch.sign(x))


def test_op_constructors():
    BasicScaleBinarizer(nn.Conv2d(3, 16, 1, 1))
    XNORWeightBinarizer()
    with pytest.raises(Exception):
        BasicScaleBinarizer(nn.ReLU())
    with pytest.raises(ValueError):
        XNORWeightBinarizer()._compute_alpha(torch.zeros(3))


# ------------------------------------------------------------------ BConfig / with_args contract
def test_bconfig_rejects_instances():
    with pytest.raises(ValueError, match="received an instance"):
        bnn.BConfig(activation_pre_process=BasicInputBinarizer())
    with pytest.raises(ValueError):
        bnn.BConfig(weight_pre_process=XNORWeightBinarizer())
    cfg = bnn.BConfig()
    assert cfg.activation_pre_process is nn.Identity and cfg.activation_post_process is bnn.Identity


def test_with_args_factories_chain_and_build_fresh_modules():
    f = XNORWeightBinarizer.with_args(compute_alpha=False).with_args(center_weights=True)
    a, b = f(), f()
    assert a is not b and a.compute_alpha is False and a.center_weights is True
    assert "XNORWeightBinarizer" in repr(f) and "center_weights=True" in repr(f)
    layer = bnn.prepare_binary_model(nn.Conv2d(4, 4, 3), bnn.BConfig(
        activation_pre_process=BasicInputBinarizer, weight_pre_process=f))
    assert layer.weight_pre_process.center_weights is True


def test_bconfig_is_required_and_shared_parameters():
    with pytest.raises(AssertionError):
        Conv2d(3, 3, 1)
    conv = nn.Conv2d(3, 8, 3, bias=True)
    twin = Conv2d.from_module(conv, bnn.BConfig(**XNOR_SCALE))
    assert twin.weight is conv.weight and twin.bias is conv.bias
    again = Conv2d.from_module(twin)  # re-binarise with its own bconfig
    assert again.weight is conv.weight
    with pytest.raises(AssertionError):
        Conv2d.from_module(nn.Linear(3, 3), bnn.BConfig(**XNOR_SCALE))


def test_forward_does_not_mutate_weights():
    conv = bnn.prepare_binary_model(nn.Conv2d(8, 8, 3, padding=1), bnn.BConfig(**XNOR_SCALE))
    before = conv.weight.detach().clone()
    conv(torch.randn(2, 8, 5, 5))
    assert torch.equal(before, conv.weight)


# ------------------------------------------------------------------ conversion quirks (G6)
def describe(m):
    return {n: [type(c).__module__.split(".")[0] == "bnn_amd", type(c).__name__,
                type(getattr(c, "activation_pre_process", None)).__name__]
            for n, c in m.named_modules() if isinstance(c, (nn.Conv2d, nn.Linear))}


def test_conversion_matches_reference_behaviour(golden_dir):
    with open(os.path.join(golden_dir, "convert.json")) as f:
        ref = json.load(f)
    cfg = bnn.BConfig(**XNOR_SCALE)
    fp32 = bnn.BConfig(activation_pre_process=nn.Identity, activation_post_process=nn.Identity,
                       weight_pre_process=nn.Identity)
    got = {
        "plain": describe(bnn.prepare_binary_model(small_net(), cfg)),
        "ignore_first_word": describe(bnn.prepare_binary_model(small_net(), cfg, ignore_layers_name=["_first_"])),
        "ignore_last_word": describe(bnn.prepare_binary_model(small_net(), cfg, ignore_layers_name=["_last_"])),
        "ignore_regex": describe(bnn.prepare_binary_model(small_net(), cfg, ignore_layers_name=["$^[03]$$"])),
        "ignore_literal": describe(bnn.prepare_binary_model(small_net(), cfg, ignore_layers_name=["8"])),
        "custom_fp32_8": describe(bnn.prepare_binary_model(small_net(), cfg, custom_config_layers_name={"8": fp32})),
    }
    for key, val in got.items():
        assert val == ref[key], key
    # the crossed special words: '_first_' leaves the LAST layer float (reference binarize.py:47-50)
    assert got["ignore_first_word"]["8"][0] is False and got["ignore_first_word"]["0"][0] is True
    assert got["ignore_last_word"]["0"][0] is False and got["ignore_last_word"]["8"][0] is True
    assert list(bnn.prepare_binary_model(small_net(), cfg).state_dict().keys()) == ref["state_keys"]


def test_custom_modules_mapping_plugin_point():
    class MyConv(Conv2d):
        pass
    MyConv._FLOAT_MODULE = nn.Conv2d
    model = bnn.prepare_binary_model(small_net(), bnn.BConfig(**XNOR_SCALE),
                                     modules_mapping={nn.Conv2d: MyConv})
    assert type(model[0]) is MyConv and type(model[8]) is nn.Linear


# ------------------------------------------------------------------ layers vs reference fixtures
@pytest.mark.parametrize("name", ["l2_ds_1x1", "c2_relu", "tail_c3", "tail_c16_1x1", "center_bias_scale",
                                  "no_alpha", "special_vals", "k3_dil2_generic", "zero_weights"])
def test_composition_path_matches_reference(name, golden_layers):
    case = LAYER_CASES_BY_NAME[name]
    x, w, b, sc = case.tensors()
    conv = nn.Conv2d(case.C, case.O, case.k, stride=case.stride, padding=case.pad,
                     dilation=case.dilation, bias=case.bias)
    conv.weight.data.copy_(torch.from_numpy(w))
    if b is not None:
        conv.bias.data.copy_(torch.from_numpy(b))
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer,
                      activation_post_process=BasicScaleBinarizer if case.post == "scale" else bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer.with_args(compute_alpha=case.compute_alpha,
                                                                       center_weights=case.center))
    layer = bnn.prepare_binary_model(conv, cfg)
    if sc is not None:
        layer.activation_post_process.alpha.data.copy_(torch.from_numpy(sc).view(1, -1, 1, 1))
    with torch.no_grad():
        out = layer(torch.from_numpy(x)).numpy()
    ref = golden_layers[name + "/out"]
    assert np.allclose(out, ref, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(ref).max()))


def _load_state(model, seed):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, seed).items()})
    return model


def xnor_cfg():
    return bnn.BConfig(activation_pre_process=BasicInputBinarizer,
                       activation_post_process=bnn.Identity, weight_pre_process=XNORWeightBinarizer)


def build_r18():
    net = bnn.prepare_binary_model(resnet18(), xnor_cfg(),
                                   custom_config_layers_name={"conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    return _load_state(net, 1).eval()


def test_resnet18_cifar_config_matches_reference(golden_dir):
    """G4 — examples/cifar10.py:61-71 model; logits vs the reference at 32x32 and 64x64."""
    g = np.load(os.path.join(golden_dir, "resnet18.npz"))
    net = build_r18()
    assert list(net.state_dict().keys()) == list(g["state_keys"])
    types = [f"{n}:{type(m).__name__}" for n, m in net.named_modules()]
    assert types == list(g["module_types"])
    for tag, shape in (("32", (4, 3, 32, 32)), ("64", (2, 3, 64, 64))):
        x = torch.from_numpy(gen.normal(gen.seed_of("r18", tag), shape))
        with torch.no_grad():
            y = net(x).numpy()
        ref = g["logits_" + tag]
        assert np.allclose(y, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
        assert (y.argmax(1) == ref.argmax(1)).all()


@pytest.mark.parametrize("name,ctor,shape", [
    ("hblock_256", lambda: HBlock(256, 256), (2, 256, 8, 8)),
    ("bottleneck_256_64", lambda: Bottleneck(256, 64), (2, 256, 8, 8)),
    ("prebasic_64", lambda: PreBasicBlock(64, 64), (2, 64, 10, 10)),
    ("prebasic_64_prelu", lambda: PreBasicBlock(64, 64, activation=nn.PReLU), (2, 64, 10, 10)),
])
def test_blocks_match_reference(name, ctor, shape, golden_dir):
    """G5 — callers of the hot path for config 5 (HBlock / Bottleneck) and the imagenet.py block."""
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    blk = _load_state(bnn.prepare_binary_model(ctor(), xnor_cfg()), gen.seed_of("block", name)).eval()
    x = torch.from_numpy(gen.normal(gen.seed_of("blockx", name), shape))
    with torch.no_grad():
        y = blk(x).numpy()
    assert np.allclose(y, g[name], rtol=1e-3, atol=1e-4 * np.abs(g[name]).max())


# ------------------------------------------------------------------ training-side hooks still work
def test_ste_backward_and_other_binarizers():
    x = torch.tensor([-2.0, -0.5, 0.0, 0.5, 2.0], requires_grad=True)
    SignActivation.apply(x).sum().backward()
    assert x.grad.tolist() == [0, 1, 1, 1, 0]
    y = AdvancedInputBinarizer()(torch.tensor([-0.3, 0.2], requires_grad=True))
    assert y.tolist() == [-1.0, 1.0] and not y.requires_grad      # as upstream (bnn/ops.py:174-176): sign under no_grad
    xs = torch.tensor([-0.3, 0.2], requires_grad=True)
    ys = AdvancedInputBinarizer(soft_gradient=True)(xs)
    assert ys.tolist() == [-1.0, 1.0] and ys.requires_grad
    ys.sum().backward()
    assert torch.allclose(xs.grad, 5 * (1 - torch.tanh(5 * xs.detach()) ** 2))
    z = StochasticInputBinarizer()(torch.randn(100))
    assert set(z.unique().tolist()) <= {-1.0, 1.0}


def test_cpu_never_uses_native_path():
    from bnn_amd import fastpath
    before = fastpath.stats()
    conv = bnn.prepare_binary_model(nn.Conv2d(8, 8, 3, padding=1), xnor_cfg())
    with torch.no_grad():
        conv(torch.randn(1, 8, 4, 4))
    assert fastpath.stats() == before


def test_fast_path_recognises_exact_hook_classes_only():
    from bnn_amd import fastpath

    class Lookalike(BasicInputBinarizer):   # a subclass may change semantics: not accelerated
        pass
    good = bnn.prepare_binary_model(nn.Conv2d(8, 8, 3), xnor_cfg())
    assert fastpath._recognise(good, 8) is not None
    odd = bnn.prepare_binary_model(nn.Conv2d(8, 8, 3), bnn.BConfig(
        activation_pre_process=Lookalike, weight_pre_process=XNORWeightBinarizer))
    assert fastpath._recognise(odd, 8) is None
    scaled = bnn.prepare_binary_model(nn.Conv2d(8, 8, 3), bnn.BConfig(**XNOR_SCALE))
    assert fastpath._recognise(scaled, 8).scale is scaled.activation_post_process.alpha
    adv = bnn.prepare_binary_model(nn.Conv2d(8, 8, 3), bnn.BConfig(
        activation_pre_process=AdvancedInputBinarizer, weight_pre_process=XNORWeightBinarizer))
    plan = fastpath._recognise(adv, 8)          # value == sign(x) with the default tanh: inference only
    assert plan is not None and plan.ste is False and fastpath._recognise(good, 8).ste is True
    adv2 = bnn.prepare_binary_model(nn.Conv2d(8, 8, 3), bnn.BConfig(
        activation_pre_process=AdvancedInputBinarizer.with_args(derivative_funct=torch.sigmoid),
        weight_pre_process=XNORWeightBinarizer))
    assert fastpath._recognise(adv2, 8) is None
