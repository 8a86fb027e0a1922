"""Host logic of the `net(x)` dispatch (bnn_amd/inference.py: AutoFusion, install_auto_fusion) that needs no GPU: it
never changes what a CPU model computes, what `state_dict()` / `repr` show, or whether the model can be copied / pickled."""
import copy
import io
import pickle

import torch

import bnn_amd as bnn
from bnn_amd.inference import (AutoFusion, auto_fusion, install_auto_fusion, is_native_model, resnet_shaped,
                               uninstall_auto_fusion)
from bnn_amd.models import resnet18
from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer
from tests.golden import gen
from tests.helpers import foreign_resnet


def _cfg():
    return bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                       weight_pre_process=XNORWeightBinarizer)


def _prepare(net):
    net = bnn.prepare_binary_model(net, _cfg(), custom_config_layers_name={"conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 1).items()})
    return net.eval()


def test_native_model_on_the_cpu_is_untouched():
    net = _prepare(resnet18())
    assert is_native_model(net) and resnet_shaped(net)
    x = torch.from_numpy(gen.normal(3, (2, 3, 32, 32)))
    with torch.no_grad():
        y = net(x)
    assert "_bnn_auto" not in net.__dict__            # a CPU tensor never reaches the dispatch
    st = auto_fusion(net)
    assert isinstance(st, AutoFusion) and st.run(net, x) is None and st.calls["declined"] == 1
    assert not any("_bnn" in k for k in net.state_dict())
    twin = pickle.loads(pickle.dumps(net))
    assert twin.__dict__["_bnn_auto"] is not st and twin.__dict__["_bnn_auto"].engine is None
    with torch.no_grad():
        assert torch.equal(twin(x), y) and torch.equal(copy.deepcopy(net)(x), y)


def test_foreign_resnet_class_swap_is_invisible_copyable_and_reversible():
    plain = foreign_resnet.ResNet()
    keys, text = list(plain.state_dict().keys()), None
    net = _prepare(plain)
    assert type(net) is not foreign_resnet.ResNet and isinstance(net, foreign_resnet.ResNet)
    assert type(net).__name__ == "ResNet" and type(net).__module__ == foreign_resnet.__name__
    assert list(net.state_dict().keys()) == keys and not install_auto_fusion(net)
    x = torch.from_numpy(gen.normal(4, (2, 3, 32, 32)))
    with torch.no_grad():
        y = net(x)
        assert torch.equal(y, foreign_resnet.ResNet.forward(net, x))      # CPU: the class's own forward
        for twin in (copy.deepcopy(net), pickle.loads(pickle.dumps(net))):
            assert type(twin) is type(net) and twin is not net and torch.equal(twin(x), y)
        buf = io.BytesIO()
        torch.save(net, buf)
        buf.seek(0)
        assert torch.equal(torch.load(buf, weights_only=False)(x), y)
        # what DataParallel does to make a replica: the replica dispatches on itself, not on the original
        replica = net._replicate_for_data_parallel()
        assert type(replica) is type(net) and "forward" not in replica.__dict__
    text = repr(net)
    uninstall_auto_fusion(net)
    assert type(net) is foreign_resnet.ResNet and repr(net) == text
    # a model that is not laid out like the reference's ResNet is left alone
    assert not install_auto_fusion(torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3)))


def test_the_cached_batchnorm_fold_follows_its_tensors_and_stays_out_of_the_module():
    """bnn_amd/inference.py: cached_fold — the eval-mode BatchNorm constants of the per-layer tails are derived once per
    state of the module's four tensors and live outside the module (nothing in __dict__ / state_dict / pickles)."""
    import copy
    import pickle
    import torch
    import torch.nn as nn
    from bnn_amd.inference import cached_fold, fold_bn
    bn = nn.BatchNorm2d(6).eval()
    with torch.no_grad():
        bn.running_var.uniform_(0.5, 2.0)
        bn.running_mean.normal_()
        bn.weight.normal_()
        bn.bias.normal_()
    keys0 = set(bn.__dict__)
    a = cached_fold(bn)
    assert cached_fold(bn) is a and set(bn.__dict__) == keys0
    x = torch.randn(3, 6, 4, 4)
    want = bn(x)
    got = torch.addcmul(a[1].view(1, 6, 1, 1), x, a[0].view(1, 6, 1, 1))
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)
    with torch.no_grad():
        bn.running_var.mul_(4.0)                                  # in-place write: version counter
    b = cached_fold(bn)
    assert b is not a and torch.allclose(b[0], a[0] * 0.5, rtol=1e-4)
    bn.bias = nn.Parameter(torch.zeros(6))                        # replaced tensor: identity
    c = cached_fold(bn)
    assert c is not b and torch.equal(c[0], b[0]) and torch.allclose(c[1], -bn.running_mean * c[0], rtol=1e-6, atol=1e-7)
    clone = pickle.loads(pickle.dumps(copy.deepcopy(bn)))
    assert set(clone.state_dict()) == set(bn.state_dict())
    assert all(torch.equal(u, v) for u, v in zip(cached_fold(clone), fold_bn(bn)))


def test_the_folded_batchnorm_is_the_reference_forward_bit_for_bit():
    """bnn_amd/inference.py: fold_bn — `fma(x, scale, shift)` with the constants rounded as ATen's CPU kernel rounds them
    IS the reference's eval-mode BatchNorm (torch on the host, what `examples/cifar10.py` evaluates): every element equal,
    2.6 M of them over four shapes incl. small variances.  (A last-bit difference in front of a sign() is a flipped
    activation: the HIP epilogues and the one-launch tails evaluate exactly this expression.)"""
    import torch
    import torch.nn as nn
    from bnn_amd.inference import fold_bn
    torch.manual_seed(0)
    for C, H in ((64, 56), (128, 28), (512, 7), (3, 33)):
        bn = nn.BatchNorm2d(C).eval()
        with torch.no_grad():
            bn.running_var.uniform_(0.05, 4.0)
            bn.running_mean.normal_()
            bn.weight.normal_()
            bn.bias.normal_()
        x = torch.randn(8, C, H, H) * 3
        with torch.no_grad():
            want = bn(x)
        scale, shift = fold_bn(bn)
        # the product of two fp32 values is exact in fp64; one rounding to fp32 behind the sum: an fma
        got = (x.double() * scale.double().view(1, C, 1, 1) + shift.double().view(1, C, 1, 1)).float()
        assert torch.equal(got, want), (C, H, int((got != want).sum()))


def test_the_one_launch_tails_keep_out_of_everything_they_do_not_cover():
    """bnn_amd/inference.py: eval_tail / eval_stem / eval_head return None — the caller then runs the modules themselves —
    for CPU tensors, under autograd, with the library switch on, and for modules that carry hooks; the helpers that decide
    it are plain host logic."""
    import torch
    import torch.nn as nn
    import bnn_amd as bnn
    from bnn_amd import inference
    from bnn_amd.models import resnet18
    from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    net = bnn.prepare_binary_model(resnet18(num_classes=10), cfg, custom_config_layers_name={
        "conv1": bnn.BConfig(), "fc": bnn.BConfig()}).eval()
    x = torch.randn(2, 3, 32, 32)
    bn, act = net.layer1[0].bn1, net.layer1[0].act1
    with torch.no_grad():
        assert inference.eval_tail(torch.randn(2, 64, 8, 8), bn, act) is None        # not on a HIP device
        assert inference.eval_stem(net, x) is None and inference.eval_head(net, torch.randn(2, 512, 1, 1)) is None
        y = net(x)                                                                    # the modules themselves
    assert y.shape == (2, 10)
    # real-valued layers kept by an all-Identity recipe count as float layers; binary ones do not
    assert inference._is_float_layer(net.conv1) and inference._is_float_layer_linear(net.fc)
    assert not inference._is_float_layer(net.layer1[0].conv1)
    assert inference._is_float_layer(nn.Conv2d(3, 8, 3)) and inference._is_float_layer_linear(nn.Linear(4, 4))
    # hooks: on the module itself, or global ones
    assert inference._no_hooks(bn, act, None)
    h = bn.register_forward_hook(lambda m, i, o: None)
    assert not inference._no_hooks(bn, act)
    h.remove()
    g = nn.modules.module.register_module_forward_hook(lambda m, i, o: None)
    assert not inference._no_hooks(bn)
    g.remove()
    assert inference._no_hooks(bn)
    # the library switch nests
    assert inference._LIBRARY_TAILS == 0
    with inference.library_tails():
        with inference.library_tails():
            assert inference._LIBRARY_TAILS == 2
        assert inference._LIBRARY_TAILS == 1
    assert inference._LIBRARY_TAILS == 0
    # two halves in flight: from 128 images of 224 x 224 on, by pixel count
    assert inference.TwoHalves.wanted(torch.empty(128, 3, 224, 224, device="meta"))
    assert not inference.TwoHalves.wanted(torch.empty(127, 3, 224, 224, device="meta"))
    assert inference.TwoHalves.wanted(torch.empty(512, 3, 112, 112, device="meta"))
    assert not inference.TwoHalves.wanted(torch.empty(1, 3, 4096, 4096, device="meta"))


def test_switching_between_training_and_evaluation_drops_derived_data():
    """`model.train()` / `model.eval()` drop the packed-weight caches of the binary layers (and the fused executor of a
    ResNet): writes through `.data` made while training — weight clipping after the optimizer step, which autograd's
    version counter does not see — reach the first evaluation forward without `fastpath.invalidate`."""
    import torch.nn as nn
    from bnn_amd.models import resnet18
    from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    layer = bnn.prepare_binary_model(nn.Conv2d(8, 8, 3), cfg)
    assert layer.training
    layer.__dict__["_bnn_packed"] = ("key", "pack")
    layer.__dict__["_bnn_packed_replicas"] = {"cuda:1": ("key", "pack")}
    layer.train()                                   # no switch: nothing dropped
    assert "_bnn_packed" in layer.__dict__
    layer.eval()
    assert "_bnn_packed" not in layer.__dict__ and "_bnn_packed_replicas" not in layer.__dict__ and not layer.training
    layer.__dict__["_bnn_packed"] = ("key", "pack")
    layer.eval()
    assert "_bnn_packed" in layer.__dict__
    layer.train()
    assert "_bnn_packed" not in layer.__dict__ and layer.training
    net = bnn.prepare_binary_model(resnet18(), cfg, custom_config_layers_name={"conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    st = auto_fusion(net)
    st.engine, st.verified = object(), True
    net.layer1[0].conv1.__dict__["_bnn_packed"] = ("key", "pack")
    net.eval()
    assert st.engine is None and not st.verified and "_bnn_packed" not in net.layer1[0].conv1.__dict__


def test_mode_switch_and_invalidate_also_drop_the_block_tier_and_the_tails_constants():
    """the block tier's executor (`_bnn_auto_block`) and the per-layer tails' folded BatchNorms / transposed
    head weight are derived data as well — a mode switch of a block (or of the whole model) and `fastpath.invalidate` drop
    them, so that `.data` writes made while training reach the first evaluation forward on EVERY tier."""
    from bnn_amd import fastpath, tails
    net = _prepare(resnet18())
    blk = net.layer1[0]
    blk.__dict__["_bnn_auto_block"] = object()
    tails._FOLDS[blk.bn1] = ("key", "fold")
    tails._HEAD_WEIGHTS[net.fc] = ("key", "wt")
    blk.eval()                                      # no switch: nothing dropped
    assert "_bnn_auto_block" in blk.__dict__ and blk.bn1 in tails._FOLDS
    net.train()
    assert "_bnn_auto_block" not in blk.__dict__ and blk.bn1 not in tails._FOLDS
    assert net.fc in tails._HEAD_WEIGHTS            # (the head belongs to the model, not to a block ...)
    blk.__dict__["_bnn_auto_block"] = object()
    tails._FOLDS[blk.bn2] = ("key", "fold")
    assert fastpath.invalidate(net) == 0
    assert "_bnn_auto_block" not in blk.__dict__ and blk.bn2 not in tails._FOLDS and net.fc not in tails._HEAD_WEIGHTS


def test_strict_weights_mode_keeps_nothing_derived(monkeypatch):
    """`BNN_AMD_STRICT_WEIGHTS=1`: the reference's behaviour (re-binarise on every forward, bnn/layers/conv.py:92) — the
    fused tiers decline, and the packed-weight lookups ask for a fresh, synchronous pack every time."""
    from bnn_amd import dispatch, fastpath
    assert not fastpath.strict_weights() and AutoFusion.enabled()
    monkeypatch.setenv("BNN_AMD_STRICT_WEIGHTS", "1")
    assert fastpath.strict_weights() and not AutoFusion.enabled()
    calls = []
    monkeypatch.setattr(fastpath.hipops, "pack_weight", lambda w, c, a, sync=True: calls.append(sync) or
                        type("PW", (), {"has_zero": False, "zero_probe": None})())
    import torch.nn as nn
    layer = bnn.prepare_binary_model(nn.Conv2d(8, 8, 3), _cfg()).eval()
    plan = fastpath._recognise(layer, 8)
    fastpath.packed_weight(layer, plan, sync=False)
    fastpath.packed_weight(layer, plan, sync=False)
    assert calls == [True, True]                    # never cached, never on an unverified "no exact zero" assumption
    monkeypatch.delenv("BNN_AMD_STRICT_WEIGHTS")
    calls.clear()
    fastpath.invalidate(layer)
    fastpath.packed_weight(layer, plan)
    fastpath.packed_weight(layer, plan)
    assert calls == [True]                          # the default: packed once per weight version
    blk = dispatch.BlockFusion()
    monkeypatch.setenv("BNN_AMD_STRICT_WEIGHTS", "1")
    assert blk.run(_prepare(resnet18()).layer1[0], torch.zeros(1, 64, 8, 8)) is None
