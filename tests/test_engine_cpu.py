"""Recipe engine (BinaryChef) — modelled on the reference's test/test_engine.py:23-66."""
import os

import pytest
import torch
import torch.nn as nn

import bnn_amd as bnn
from bnn_amd.engine import BinaryChef
from bnn_amd.ops import BasicInputBinarizer, BasicScaleBinarizer, XNORWeightBinarizer

ASSET = os.path.join(os.path.dirname(__file__), "assets", "recipe_three_steps.yaml")


def net():
    return nn.Sequential(nn.Conv2d(3, 16, 1, 1), nn.BatchNorm2d(16), nn.ReLU(inplace=True),
                         nn.Conv2d(16, 16, 1, 1), nn.BatchNorm2d(16), nn.ReLU(inplace=True),
                         nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Linear(16, 3))


def test_step_length():
    chef = BinaryChef(ASSET)
    assert len(chef) == 3 and chef.get_num_steps() == 3


def test_engine_steps_swap_hooks_like_the_reference():
    model, chef = net(), BinaryChef(ASSET)
    model = chef.next(model)                                   # step 0: float weights, binary inputs
    assert type(model[0]) is nn.Conv2d and type(model[8]) is nn.Linear     # _first_/_last_ ignored
    assert hasattr(model[3], "bconfig") and isinstance(model[3].weight_pre_process, nn.Identity)
    assert isinstance(model[3].activation_pre_process, BasicInputBinarizer)
    model = chef.next(model)                                   # step 1: centred XNOR weights
    w = model[3].weight_pre_process
    assert isinstance(w, XNORWeightBinarizer) and w.center_weights and w.compute_alpha
    alpha = model[3].activation_post_process.alpha
    assert isinstance(model[3].activation_post_process, BasicScaleBinarizer)
    model = chef.next(model)                                   # step 2: 'NAME' key, no ignore list
    assert isinstance(model[3].weight_pre_process, XNORWeightBinarizer)
    assert not model[3].weight_pre_process.center_weights
    assert isinstance(model[3].activation_post_process, bnn.Identity)
    assert isinstance(model[0], bnn.layers.Conv2d) and isinstance(model[8], bnn.layers.Linear)
    assert torch.equal(alpha, torch.ones_like(alpha))
    out = model(torch.randn(2, 3, 4, 4))
    assert out.shape == (2, 3)


def test_user_modules_and_errors(tmp_path):
    class MyBinarizer(BasicInputBinarizer):
        pass
    recipe = tmp_path / "r.yaml"
    recipe.write_text("s0:\n  pre_activation: {name: MyBinarizer}\n  post_activation: {name: Identity}\n"
                      "  weight: {name: XNORWeightBinarizer, args: {center_weights: true}}\n")
    with pytest.raises(NameError, match="MyBinarizer"):
        BinaryChef(str(recipe)).run_step(net(), 0)
    model = BinaryChef(str(recipe), user_modules=[MyBinarizer]).run_step(net(), 0)
    assert isinstance(model[3].activation_pre_process, MyBinarizer)
    recipe.write_text('s0:\n  pre_activation: {name: "print(1)"}\n'
                      "  post_activation: {name: Identity}\n  weight: {name: nn.Identity}\n")
    with pytest.raises(NameError):                             # recipes are data, never code
        BinaryChef(str(recipe)).run_step(net(), 0)
    with pytest.raises(AssertionError):
        BinaryChef(ASSET).run_step(net(), 3)
