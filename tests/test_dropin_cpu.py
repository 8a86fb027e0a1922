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
