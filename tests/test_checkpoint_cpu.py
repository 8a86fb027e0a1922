"""Packed checkpoint format: 1 bit/weight + alpha on disk, fp32-keyed
state_dict back, forward unchanged.  Mirrors the reference's state-dict round trip
(test/test_binarize.py:95-110) for the packed container."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import bnn_amd as bnn
from bnn_amd import checkpoint
from bnn_amd.models import resnet18
from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer


def _net(center, compute_alpha=True, seed=0):
    torch.manual_seed(seed)
    net = resnet18(num_classes=10)
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer.with_args(compute_alpha=compute_alpha,
                                                                       center_weights=center))
    net = bnn.prepare_binary_model(net, cfg, ignore_layers_name=["conv1", "fc"])
    g = torch.Generator().manual_seed(seed + 1)
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data = torch.rand(m.num_features, generator=g) + 0.5
            m.bias.data = torch.randn(m.num_features, generator=g) * 0.3
            m.running_mean = torch.randn(m.num_features, generator=g) * 0.5
            m.running_var = torch.rand(m.num_features, generator=g) + 0.5
    return net.eval()


@pytest.mark.parametrize("center,compute_alpha", [(False, True), (True, True), (False, False)])
def test_round_trip_keeps_schema_signs_alpha_and_forward(tmp_path, center, compute_alpha):
    net = _net(center, compute_alpha)
    path = str(tmp_path / "r18.bnnpack")
    stats = checkpoint.save_packed(net, path)
    sd0 = net.state_dict()
    sd1 = checkpoint.load_packed(path)
    assert list(sd1.keys()) == list(sd0.keys())
    hooks = checkpoint._binary_weight_hooks(net)
    assert len(hooks) == 19                                   # every conv but conv1 (and not fc)
    for k in sd0:
        assert sd1[k].dtype == sd0[k].dtype and sd1[k].shape == sd0[k].shape
        if k in hooks:
            a, b = hooks[k](sd0[k]), hooks[k](sd1[k])         # what the layer's forward uses
            assert torch.equal(torch.sign(a), torch.sign(b))
            assert torch.allclose(a, b, rtol=2e-6, atol=0)
        else:
            assert torch.equal(sd1[k], sd0[k])
    # x32 on the binary weights (+ one fp32 alpha per output channel), whole file far below fp32
    assert stats["binary_weights_packed"] * 30 < stats["binary_weights_fp32"]
    assert stats["file"] < stats["fp32_state_dict"] / 8
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        y0 = net(x)
        net2 = _net(center, compute_alpha, seed=99)           # different weights, same architecture
        missing = net2.load_state_dict(sd1)
        assert not missing.missing_keys and not missing.unexpected_keys
        y1 = net2(x)
    assert torch.allclose(y0, y1, rtol=1e-4, atol=1e-5 * float(y0.abs().max()))


def test_zero_weights_and_linear_and_raw_tensors(tmp_path):
    torch.manual_seed(3)
    net = nn.Sequential(nn.Conv2d(8, 16, 3, padding=1), nn.Flatten(), nn.Linear(16 * 4 * 4, 5))
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    net = bnn.prepare_binary_model(net, cfg).eval()
    with torch.no_grad():
        net[0].weight[1, 2] = 0.0                              # exact zeros -> zero mask is stored
        net[2].weight[0, :7] = 0.0
    path = str(tmp_path / "m.bnnpack")
    checkpoint.save_packed(net, path)
    signs = checkpoint.load_packed_signs(path)
    assert set(signs) == {"0.weight", "2.weight"}
    s, alpha, meta = signs["0.weight"]
    assert np.array_equal(s, np.sign(net[0].weight.detach().numpy()).astype(np.int8))
    assert (s[1, 2] == 0).all() and meta == {"center": False, "compute_alpha": True}
    ref_alpha = net[0].weight.detach().abs().double().mean(dim=(1, 2, 3)).float().numpy()
    assert np.array_equal(alpha, ref_alpha)
    sd = checkpoint.load_packed(path)
    assert torch.equal(sd["0.bias"], net[0].bias) and torch.equal(sd["2.bias"], net[2].bias)
    x = torch.randn(3, 8, 4, 4)
    with torch.no_grad():
        y0 = net(x)
        net.load_state_dict(sd)
        assert torch.allclose(net(x), y0, rtol=1e-5, atol=1e-6)


def test_bad_files_are_rejected(tmp_path):
    p = tmp_path / "x.bnnpack"
    p.write_bytes(b"not a checkpoint")
    with pytest.raises(ValueError):
        checkpoint.load_packed(str(p))
    net = _net(False)
    good = tmp_path / "g.bnnpack"
    checkpoint.save_packed(net, str(good))
    raw = good.read_bytes()
    (tmp_path / "t.bnnpack").write_bytes(raw[: len(raw) // 2])
    with pytest.raises(ValueError):
        checkpoint.load_packed(str(tmp_path / "t.bnnpack"))
    assert not os.path.exists(str(good) + ".tmp")
