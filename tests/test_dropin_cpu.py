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
