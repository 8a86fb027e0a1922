"""The reference's own call at the speed the bench quotes:

    net = prepare_binary_model(resnet18(), bconfig, custom_config_layers_name={'conv1': BConfig(), 'fc': BConfig()})
    net.eval();  with torch.no_grad(): outputs = net(inputs)           # examples/cifar10.py:61-71,140-149

takes the fused executor — first batch of a shape as eager launches (19 for ResNet-18), from the second one on as
"stem launch on the caller's tensor + HIP graph of the rest" (bnn_amd/inference.py: AutoFusion, forward_fresh) — and
is bit-identical to FusedResNet(net)(x).  Everything that has to keep the per-layer path (training, autograd, hooks on
inner modules, uncovered models, foreign classes whose fused result does not check out) is covered too."""
import copy
import warnings

import numpy as np
import pytest
import torch
import torch.nn as nn

import bnn_amd as bnn
from bnn_amd import fastpath, native
from bnn_amd.inference import (AutoFusion, FusedBlocks, FusedResNet, PipelinedInference, auto_fusion,
                               install_auto_fusion, no_model_fusion, per_layer_forward)
from bnn_amd.models import BasicBlock, Bottleneck, HBlock, PreBasicBlock, resnet18
from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer
from tests.golden import gen

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(a).to(DEV)


def _cfg():
    return bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                       weight_pre_process=XNORWeightBinarizer)


def _load(net, seed=1):
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, seed).items()})
    return net.to(DEV).eval()


def _r18(**kw):
    net = bnn.prepare_binary_model(resnet18(**kw), _cfg(), custom_config_layers_name={
        "conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    return _load(net)


def test_net_call_runs_the_fused_executor_and_equals_it_bit_for_bit():
    net = _r18()
    xs = [dev(gen.normal(70 + i, (8, 3, 64, 64))) for i in range(4)]      # a NEW tensor every call
    want = [FusedResNet(net)(x).clone() for x in xs]
    st = auto_fusion(net)
    with torch.no_grad():
        per_layer0 = fastpath.stats()["conv2d"]
        net(xs[0][:3].contiguous())       # the very first call also BUILDS the executor (weight packs, thresholds)
        assert st.calls == {"graph": 0, "eager": 1, "declined": 0} and st.engine is not None
        launches0 = native.launch_count()
        y0 = net(xs[0])
        # first batch of this shape: the 19 eager launches of the fused executor (stem, 16 convs with the shortcut
        # convs and their OR-pools folded in, head) — and nothing through the per-layer path
        assert native.launch_count() - launches0 == 19 and fastpath.stats()["conv2d"] == per_layer0
        assert st.calls == {"graph": 0, "eager": 2, "declined": 0}
        ys = [y0]
        for x in xs[1:]:
            launches0 = native.launch_count()
            ys.append(net(x))
            # the stem reads the caller's tensor (the capture call runs the network's launches eagerly first)
            assert native.launch_count() - launches0 >= 1
        assert st.calls["graph"] == 3 and st.calls["eager"] == 2 and len(st.engine._split) == 1
        launches0 = native.launch_count()
        y_again = net(xs[1])
        assert native.launch_count() - launches0 == 1        # steady state: ONE launch through the C-ABI + a graph replay
    for y, w in zip(ys, want):
        assert torch.equal(y, w)                             # results are the caller's own tensors: later calls did
    assert torch.equal(y_again, want[1])                     # not overwrite earlier ones
    assert len({y.data_ptr() for y in ys}) == len(ys)
    assert fastpath.stats()["conv2d"] == per_layer0


def test_ragged_last_batch_and_other_shapes_take_eager_fused_launches():
    net = _r18()
    fused = FusedResNet(net)
    with torch.no_grad():
        for shape in [(8, 3, 64, 64), (8, 3, 64, 64), (5, 3, 64, 64), (2, 3, 96, 80), (8, 3, 64, 64)]:
            x = dev(gen.normal(sum(shape), shape))
            assert torch.equal(net(x), fused(x))
    st = auto_fusion(net)
    assert st.calls == {"graph": 2, "eager": 3, "declined": 0}


def test_what_keeps_the_per_layer_path():
    net = _r18()
    x = dev(gen.normal(5, (4, 3, 64, 64)))
    st = auto_fusion(net)
    n0 = fastpath.stats()["conv2d"]
    y_grad = net(x)                                  # autograd recording: the per-layer path (training kernels)
    assert st.calls["declined"] == 0 and st.engine is None      # (ResNet.forward does not even ask)
    with torch.no_grad():
        with per_layer_forward():
            y_layer = net(x)
        assert fastpath.stats()["conv2d"] == n0 + 19 and st.calls["declined"] == 1
        handle = net.layer2[0].conv1.register_forward_hook(lambda m, i, o: None)
        y_hooked = net(x)                            # builds the executor, then sees the hook: declined as a whole —
        # the model's own forward runs, in which every block WITHOUT a hook fuses itself (BlockFusion) and the hooked
        # block runs layer by layer: its two convs and its shortcut conv are the only per-layer calls
        assert fastpath.stats()["conv2d"] == n0 + 19 + 3 and st.calls["declined"] == 2
        assert torch.allclose(y_hooked, y_layer, rtol=1e-3, atol=1e-3 * float(y_layer.abs().max()))
        handle.remove()
        y_fused = net(x)
        assert st.calls["eager"] == 1 and fastpath.stats()["conv2d"] == n0 + 22
        net.train()
        net(x)
        net.eval()
        assert st.calls["eager"] == 1
    assert torch.allclose(y_fused, y_layer, rtol=1e-3, atol=1e-3 * float(y_layer.abs().max()))
    assert torch.allclose(y_grad.detach(), y_layer, rtol=1e-3, atol=1e-3 * float(y_layer.abs().max()))


def test_parameter_updates_are_seen_by_the_next_call():
    net = _r18()
    x = dev(gen.normal(6, (4, 3, 64, 64)))
    with torch.no_grad():
        y0 = net(x)
        y0b = net(x)                                  # graph captured
        assert torch.equal(y0, y0b)
        _load(net, seed=2)                            # load_state_dict writes in place: version counters move
        y1 = net(x)
        assert not torch.equal(y0, y1)
        assert torch.equal(y1, FusedResNet(net)(x))
        assert torch.equal(net(x), y1)
        with per_layer_forward():
            y_layer = net(x)
    assert torch.allclose(y1, y_layer, rtol=1e-3, atol=1e-3 * float(y_layer.abs().max()))


def test_weights_written_through_dot_data_while_training_reach_the_next_evaluation():
    """`p.data.clamp_()` after the optimizer step does not move autograd's version counter — the one thing the fused
    executor's packed weights are keyed on.  `net.train()` / `net.eval()` drop the executor and the layers' packs
    (round 5), so the usual loop — train with clipping, then evaluate — sees the clipped weights without
    `fastpath.invalidate`."""
    net = _r18()
    x = dev(gen.normal(16, (4, 3, 64, 64)))
    with torch.no_grad():
        y0 = net(x)
        assert auto_fusion(net).engine is not None
        net.train()
        assert auto_fusion(net).engine is None
        w = net.layer2[1].conv1.weight
        v = w._version
        w.data.mul_(-1.0)                             # through .data: invisible to the version counter
        assert w._version == v
        net.eval()
        y1 = net(x)
        assert not torch.equal(y1, y0) and torch.equal(y1, FusedResNet(net)(x))
        with per_layer_forward():
            y_layer = net(x)
        assert torch.allclose(y1, y_layer, rtol=1e-3, atol=1e-3 * float(y_layer.abs().max()))


def test_data_parallel_wrapper_and_deepcopy_and_state_dict():
    net = _r18()
    keys = list(net.state_dict().keys())
    x = dev(gen.normal(8, (4, 3, 64, 64)))
    with torch.no_grad():
        y = net(x)
        wrapped = nn.DataParallel(net, device_ids=[0])          # examples/cifar10.py:74-77 on a one-GPU box
        assert torch.equal(wrapped(x), y)
        twin = copy.deepcopy(net)
        assert auto_fusion(twin) is not auto_fusion(net) and auto_fusion(twin).engine is None
        assert torch.equal(twin(x), y)
    assert list(net.state_dict().keys()) == keys and "_bnn_auto" not in repr(net)


def test_dabnn_stem_runs_as_modules_in_front_of_fused_blocks():
    """The cheaper stem of daBNN (bnn/models/resnet.py:10-47): the stem runs as the torch modules it is, the residual
    blocks behind it are fused (no graph: the part that reads the caller's tensor is not one kernel)."""
    net = bnn.prepare_binary_model(resnet18(stem_type="dabnn"), _cfg(), ignore_layers_name=["_first_", "_last_"])
    net = _load(net)
    x = dev(gen.normal(14, (4, 3, 64, 64)))
    with torch.no_grad():
        n0 = fastpath.stats()["conv2d"]
        y = net(x)
        st = auto_fusion(net)
        assert st.engine is not None and not st.engine.reads_caller_tensor and st.calls["eager"] == 1
        assert fastpath.stats()["conv2d"] - n0 == 3          # the stem's three binary convs (its first conv stays float)
        assert torch.equal(net(x), y) and st.calls == {"graph": 0, "eager": 2, "declined": 0}
        with per_layer_forward():
            y_layer = net(x)
    assert torch.allclose(y, y_layer, rtol=1e-3, atol=1e-3 * float(y_layer.abs().max()))


def test_uncovered_model_keeps_its_own_forward_silently():
    net = bnn.prepare_binary_model(resnet18(norm_layer=lambda c: nn.InstanceNorm2d(c, affine=True)), _cfg(),
                                   ignore_layers_name=["_first_", "_last_"])
    net = net.to(DEV).eval()
    x = dev(gen.normal(9, (2, 3, 64, 64)))
    with torch.no_grad():
        y = net(x)
        y2 = net(x)
    st = auto_fusion(net)
    assert st.engine is None and st.reason and st.calls["declined"] == 2 and torch.equal(y, y2)


# ---- a ResNet of ANOTHER package, laid out like the reference's bnn.models.resnet.ResNet ---------------------------

def _foreign_classes(residual: bool):
    from tests.helpers import foreign_resnet
    return lambda: foreign_resnet.ResNet(residual)


def test_foreign_resnet_gets_the_dispatch_from_prepare_binary_model():
    net = bnn.prepare_binary_model(_foreign_classes(True)(), _cfg(), custom_config_layers_name={
        "conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    assert hasattr(type(net), "_bnn_base") and type(net).__name__ == "ResNet"     # install_auto_fusion ran
    assert not install_auto_fusion(net)                  # ... once
    net = _load(net)
    x = dev(gen.normal(11, (4, 3, 64, 64)))
    # same state_dict keys as bnn_amd's resnet18: the same weights give the same logits
    ours = _r18()
    with torch.no_grad():
        n0 = fastpath.stats()["conv2d"]
        y = net(x)                                       # fused + checked against the class's own forward on 2 images
        st = auto_fusion(net)
        assert st.verified and st.calls["eager"] == 1 and fastpath.stats()["conv2d"] == n0 + 19
        n0, l0 = fastpath.stats()["conv2d"], native.launch_count()
        y2 = net(x)
        assert fastpath.stats()["conv2d"] == n0 and st.calls["graph"] == 1
        assert torch.equal(y, y2) and torch.equal(y, FusedResNet(ours)(x))
        with per_layer_forward():
            y_layer = net(x)
    assert torch.allclose(y, y_layer, rtol=1e-3, atol=1e-3 * float(y_layer.abs().max()))


def test_foreign_class_with_the_same_names_but_other_arithmetic_is_not_fused():
    net = bnn.prepare_binary_model(_foreign_classes(False)(), _cfg(), custom_config_layers_name={
        "conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    net = _load(net)
    x = dev(gen.normal(12, (4, 3, 64, 64)))
    with torch.no_grad():
        with per_layer_forward():
            want = net(x)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            y = net(x)
        assert any("differs from the model's own forward" in str(m.message) for m in w)
        st = auto_fusion(net)
        assert st.engine is None and not st.verified
        assert torch.equal(y, want) and torch.equal(net(x), want)      # its own forward, now and later


def test_prebasicblock_prelu_net_through_the_model_call():
    """The reference's other shipping dataflow (examples/imagenet.py:153-155) through `net(x)`."""
    net = bnn.prepare_binary_model(resnet18(block_type=PreBasicBlock, activation=nn.PReLU), _cfg(),
                                   ignore_layers_name=["_first_", "_last_"])
    net = _load(net)
    x = dev(gen.normal(13, (4, 3, 64, 64)))
    with torch.no_grad():
        y = net(x)
        assert auto_fusion(net).calls["eager"] == 1
        assert torch.equal(net(x), y) and torch.equal(y, FusedResNet(net)(x))
        with per_layer_forward():
            y_layer = net(x)
    assert torch.allclose(y, y_layer, rtol=1e-3, atol=1e-3 * float(y_layer.abs().max()))


def test_pipelined_inference_with_fresh_inputs():
    """Two batches in flight where every launch reads a NEW caller tensor (no static input buffer, no copy)."""
    net = _r18()
    xs = [dev(gen.normal(80 + i, (8, 3, 64, 64))) for i in range(5)]
    single = FusedResNet(net)
    want = [single(x).clone() for x in xs]
    pipe = PipelinedInference(net, xs[0], n_streams=2, fresh_input=True)
    got = []
    for i, x in enumerate(xs):
        y = pipe.launch(i, x)
        with torch.cuda.stream(pipe.stream(i)):
            got.append(y.clone())
    pipe.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_forward_fresh_keeps_a_bounded_number_of_graphs():
    net = _r18()
    eng = FusedResNet(net)
    eager = FusedResNet(net)
    for n in range(1, 8):
        x = dev(gen.normal(n, (n, 3, 32, 32)))
        assert torch.equal(eng.forward_fresh(x), eager(x))
    assert len(eng._split) == FusedResNet.MAX_SPLIT_GRAPHS
    assert isinstance(auto_fusion(net), AutoFusion)


# ---- second tier: residual blocks fuse themselves when the whole model is not covered -------------------------------

class Cifar20(nn.Module):
    """A CIFAR-style three-stage ResNet (the "ResNet-20" BASELINE.json's config 1 names; 16 / 32 / 64 channels, no
    max-pool): built from bnn_amd's BasicBlock but NOT laid out like the reference's ImageNet ResNet."""

    def __init__(self, n=3, width=16):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, width, 3, padding=1, bias=False), nn.BatchNorm2d(width), nn.ReLU())
        blocks, inp = [], width
        for planes, stride in ((width, 1), (2 * width, 2), (4 * width, 2)):
            for i in range(n):
                ds = None
                if i == 0 and (stride != 1 or inp != planes):
                    ds = nn.Sequential(nn.AvgPool2d(stride, stride, ceil_mode=True, count_include_pad=False),
                                       nn.Conv2d(inp, planes, 1, bias=False), nn.BatchNorm2d(planes))
                blocks.append(BasicBlock(inp, planes, stride if i == 0 else 1, ds))
                inp = planes
        self.blocks = nn.Sequential(*blocks)
        self.head = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(inp, 10))

    def forward(self, x):
        return self.head(self.blocks(self.stem(x)))


def test_blocks_of_a_custom_network_fuse_themselves():
    net = bnn.prepare_binary_model(Cifar20(), _cfg(), ignore_layers_name=["_first_", "_last_"])
    assert "_bnn_auto" not in net.__dict__ and not hasattr(type(net), "_bnn_base")     # not ResNet-shaped: no model dispatch
    net = _load(net)
    x = dev(gen.normal(21, (16, 3, 32, 32)))
    with torch.no_grad():
        with per_layer_forward():
            n0 = fastpath.stats()["conv2d"]
            want = net(x)
            assert fastpath.stats()["conv2d"] - n0 == 9 * 2 + 2          # every binary conv on its own
        n0, l0 = fastpath.stats()["conv2d"], native.launch_count()
        y = net(x)
        assert fastpath.stats()["conv2d"] == n0                          # nothing through the per-layer path
        y2 = net(x)
    blocks = list(net.blocks)
    # (declined once: the per_layer_forward() call above)
    assert all(b.__dict__["_bnn_auto_block"].calls == {"fused": 2, "declined": 1} for b in blocks)
    assert torch.equal(y, y2)
    assert torch.allclose(y, want, rtol=1e-3, atol=1e-3 * float(want.abs().max()))
    # one block on its own == the executor for a Sequential of it, bit for bit
    blk = blocks[3]
    t = dev(gen.activation("relu", 22, (4, 16, 32, 32)))
    with torch.no_grad():
        assert torch.equal(blk(t), FusedBlocks(nn.Sequential(blk).eval())(t))
    assert list(net.state_dict().keys()) == [k for k in net.state_dict().keys() if "_bnn" not in k]


@pytest.mark.parametrize("kind", ["bottleneck", "prebasic", "hblock"])
def test_other_block_families_fuse_themselves_too(kind):
    if kind == "bottleneck":
        blk = Bottleneck(64, 16)
    elif kind == "prebasic":
        blk = PreBasicBlock(64, 64, activation=nn.PReLU)
    else:
        blk = HBlock(64, 64)
    blk = _load(bnn.prepare_binary_model(blk, _cfg()))
    x = dev(gen.normal(23, (4, 64, 14, 14)))
    with torch.no_grad():
        with per_layer_forward():
            want = blk(x)
        n0 = fastpath.stats()["conv2d"]
        y = blk(x)
        assert fastpath.stats()["conv2d"] == n0 and blk.__dict__["_bnn_auto_block"].calls["fused"] == 1
        with no_model_fusion():                 # (switches whole-model fusion off, not the blocks)
            assert torch.equal(blk(x), y)
        blk.train()
        assert "_bnn_auto_block" not in blk.__dict__    # the mode switch drops the block's executor (derived data) ...
        blk(x)
        blk.eval()
        # ... and a `.data` write made while training reaches the first evaluation forward of the block tier
        w = blk.conv1.weight
        w.data.neg_()
        y_neg = blk(x)
        assert blk.__dict__["_bnn_auto_block"].calls["fused"] == 1 and not torch.equal(y_neg, y)
        with per_layer_forward():               # (the training-mode call above moved the BatchNorm statistics: a new reference)
            want_neg = blk(x)
        assert torch.allclose(y_neg, want_neg, rtol=1e-3, atol=1e-3 * float(want_neg.abs().max()))
    assert torch.allclose(y, want, rtol=1e-3, atol=1e-3 * float(want.abs().max()))


def test_resnet_with_model_fusion_off_runs_blockwise():
    net = _r18()
    x = dev(gen.normal(24, (4, 3, 64, 64)))
    with torch.no_grad():
        want = net(x)
        n0 = fastpath.stats()["conv2d"]
        with no_model_fusion():
            y = net(x)                          # torch stem + 8 self-fused blocks + torch head
        assert fastpath.stats()["conv2d"] == n0
        assert all(b.__dict__["_bnn_auto_block"].calls["fused"] == 1 for st_ in (net.layer1, net.layer2, net.layer3, net.layer4)
                   for b in st_)
    assert torch.allclose(y, want, rtol=1e-3, atol=1e-3 * float(want.abs().max()))


@pytest.mark.parametrize("n", [64, 65])
def test_large_batches_run_as_two_halves_in_flight(n):
    """From TwoHalves.MIN_PIXELS input pixels on, `net(x)` cuts the batch in two halves on two streams (each: stem launch
    on its part of the caller's tensor + HIP graph of the rest): same bits, a new tensor every call, odd batches too."""
    from bnn_amd.inference import TwoHalves
    monkey = TwoHalves.MIN_PIXELS
    TwoHalves.MIN_PIXELS = 64 * 32 * 32            # (the test's images are small)
    try:
        _two_halves_case(n, TwoHalves)
    finally:
        TwoHalves.MIN_PIXELS = monkey


def _two_halves_case(n, TwoHalves):
    net = _r18()
    eng = FusedResNet(net)
    xs = [dev(gen.normal(120 + i, (n, 3, 32, 32))) for i in range(3)]
    want = [eng(x).clone() for x in xs]
    with torch.no_grad():
        ys = [net(x) for x in xs]                  # eager, capture (two halves), replay
        st = auto_fusion(net)
        assert st.calls == {"graph": 2, "eager": 1, "declined": 0} and len(st.halves) == 1
        two = next(iter(st.halves.values()))
        assert isinstance(two, TwoHalves) and two.captured(xs[0]) and all(e.throughput_mode for e in two.engines)
        launches0 = native.launch_count()
        y = net(xs[0])
        assert native.launch_count() - launches0 == 2            # two stem launches + two graph replays
        import os
        os.environ["BNN_AMD_SPLIT_BATCH"] = "0"
        try:
            y_single = net(xs[1])
        finally:
            del os.environ["BNN_AMD_SPLIT_BATCH"]
    for a, b in zip(ys, want):
        assert torch.equal(a, b)
    assert torch.equal(y, want[0]) and torch.equal(y_single, want[1])


# ---- the per-layer path's tails (bnn_amd/inference.py: eval_tail, eval_stem; include/bnn_hip.h: bnn_hip_bn_act_f32) ----
@pytest.mark.parametrize("shape", [(3, 64, 14, 14), (2, 40, 7, 5), (5, 3, 1, 1), (1, 130, 9, 9)])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("res", [False, True])
def test_the_batchnorm_tail_is_the_modules_arithmetic(shape, relu, res):
    """relu?(bn(x) (+ residual)) in one launch == fma(x, scale, shift) with the constants rounded as ATen's CPU kernel
    rounds them (fold_bn), then the add, then the clamp — bit for bit; and the library's own modules to rounding."""
    from bnn_amd.inference import cached_fold
    from bnn_amd.models.blocks import _bn_act
    N, C, H, W = shape
    bn = nn.BatchNorm2d(C).to(DEV).eval()
    with torch.no_grad():
        bn.running_mean.copy_(dev(gen.normal(1, (C,))) * 0.5)
        bn.running_var.copy_(dev(gen.normal(2, (C,))).abs() + 0.3)
        bn.weight.copy_(dev(gen.normal(3, (C,))))
        bn.bias.copy_(dev(gen.normal(4, (C,))))
    x = dev(gen.normal(5, shape)) * 2.0
    x[0, 0, 0, 0] = float("nan")
    r = dev(gen.normal(6, shape)) if res else None
    act = nn.ReLU(inplace=True) if relu else None
    with torch.no_grad():
        n0 = native.launch_count()
        y = _bn_act(x, bn, act, r)
        assert native.launch_count() == n0 + 1
        scale, shift = cached_fold(bn)
        want = (x.double() * scale.double().view(1, C, 1, 1) + shift.double().view(1, C, 1, 1)).float()   # one rounding: fma
        if res:
            want = want + r
        if relu:
            want = torch.relu(want)
        lib = bn(x)
        if res:
            lib = lib + r
        if relu:
            lib = torch.relu(lib)
    assert torch.isnan(y[0, 0, 0, 0]) and torch.isnan(want[0, 0, 0, 0])      # ReLU keeps NaN, like torch.relu
    assert torch.equal(torch.nan_to_num(y, nan=7.0), torch.nan_to_num(want, nan=7.0))
    assert torch.allclose(torch.nan_to_num(y, nan=7.0), torch.nan_to_num(lib, nan=7.0), rtol=1e-5, atol=1e-5)


def test_the_fold_on_the_module_follows_its_tensors():
    from bnn_amd.inference import cached_fold
    bn = nn.BatchNorm2d(8).to(DEV).eval()
    a = cached_fold(bn)
    assert cached_fold(bn) is a
    with torch.no_grad():
        bn.running_var.mul_(4.0)
    b = cached_fold(bn)
    assert b is not a and torch.allclose(b[0], a[0] * 0.5, rtol=1e-3)
    bn.weight = nn.Parameter(torch.full((8,), 3.0, device=DEV))
    assert torch.allclose(cached_fold(bn)[0], b[0] * 3.0, rtol=1e-6)


def test_the_per_layer_path_runs_its_tails_as_single_launches_and_agrees_with_the_fused_executor():
    from bnn_amd import hipops
    from bnn_amd.inference import library_tails
    net = _r18()
    x = dev(gen.normal(31, (6, 3, 96, 96)))
    calls = {"bn_act": 0, "stem": 0, "head": 0}
    real_bn, real_stem, real_head = hipops.bn_act, hipops.stem7x7, hipops.avgpool_fc

    def bn_act(*a, **k):
        calls["bn_act"] += 1
        return real_bn(*a, **k)

    def stem(*a, **k):
        calls["stem"] += 1
        return real_stem(*a, **k)
    def head(*a, **k):
        calls["head"] += 1
        return real_head(*a, **k)
    try:
        hipops.bn_act, hipops.stem7x7, hipops.avgpool_fc = bn_act, stem, head
        with torch.no_grad(), per_layer_forward():
            n0 = fastpath.stats()["conv2d"]
            y = net(x)
            assert fastpath.stats()["conv2d"] - n0 == 19                  # every binary conv on its own
            assert calls == {"bn_act": 16 + 3, "stem": 1, "head": 1}      # 16 block tails + 3 shortcut BatchNorms, stem, head
            with library_tails():
                y_lib = net(x)
            assert calls == {"bn_act": 19, "stem": 1, "head": 1}          # the library's modules: none of ours
    finally:
        hipops.bn_act, hipops.stem7x7, hipops.avgpool_fc = real_bn, real_stem, real_head
    with torch.no_grad():
        y_fused = FusedResNet(net)(x)
    # the same stem kernel, the same float operations in the tails, the same head kernel: the fused executor's bits; the
    # library's BatchNorm rounds differently
    assert torch.equal(y, y_fused)
    assert torch.allclose(y, y_lib, rtol=1e-3, atol=1e-3 * float(y_lib.abs().max()))


def test_a_hook_on_a_batchnorm_keeps_that_module_a_module():
    net = _r18()
    x = dev(gen.normal(32, (2, 3, 64, 64)))
    seen = []
    h = net.layer1[0].bn2.register_forward_hook(lambda m, i, o: seen.append(tuple(o.shape)))
    try:
        with torch.no_grad():
            y = net(x)                      # (hooks inside: the model tier declines, the block tier declines for layer1[0])
    finally:
        h.remove()
    assert seen == [(2, 64, 16, 16)]
    with torch.no_grad():
        assert torch.allclose(y, net(x), rtol=1e-3, atol=1e-3 * float(y.abs().max()))


@pytest.mark.parametrize("tier", ["model", "block", "layer"])
def test_net_call_inside_a_callers_own_graph_capture(tier):
    """A caller that captures `net(x)` into a HIP graph of its own (after the usual warm-up calls): no graph replay inside
    a capture, no second stream, nothing built or synchronised — the ready executor's launches (model / block tier) or the
    per-layer launches are recorded, and the caller's graph replays to the same logits."""
    from bnn_amd.inference import no_model_fusion
    import contextlib
    net = _r18()
    xs = [dev(gen.normal(60 + i, (4, 3, 64, 64))) for i in range(3)]
    want = [FusedResNet(net)(x).clone() for x in xs]
    ctx = {"model": contextlib.nullcontext, "block": no_model_fusion, "layer": per_layer_forward}[tier]
    static_x = xs[0].clone()
    side = torch.cuda.Stream()
    with torch.no_grad(), ctx():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                net(static_x)                              # warm-up on a side stream, as for any captured callable
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = net(static_x)
    for x, w in zip(xs, want):
        static_x.copy_(x)
        g.replay()
        assert torch.equal(y, w)
    if tier == "model":
        assert auto_fusion(net).calls["eager"] >= 2         # the first call + the captured one


def test_config1_batch32_against_the_reference_fixture():
    """BASELINE config 1 at its stated size on the GPU (examples/cifar10.py model, 32 x 32 inputs, batch 32): the four
    batches bench.py's `gpu_c1` leg times, through the reference's own call `net(x)`, against the reference's logits and
    the sign checksums in front of its 19 binary convolutions (tests/golden/resnet18.npz: c1_*, from
    tests/golden/make_golden.py `resnet18`).  Same strict / counted split as config 3: an image whose checksums all equal
    the reference's is within 1e-3 (measured 1e-6); images with a flipped sign are counted."""
    import os
    import numpy as np
    from tests.golden import gen, sighash
    from bnn_amd import inference
    from bnn_amd.inference import FusedResNet
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resnet18.npz"))
    names = [str(n) for n in fx["c1_layers"]]
    net = _r18()
    flipped_total = 0
    for j in range(4):
        x = torch.from_numpy(gen.normal(40 + j, (32, 3, 32, 32))).to(DEV)
        with torch.no_grad():
            y = net(x)                                   # the drop-in call: fused executor (eager, then stem + graph)
            y2 = net(x)
        assert torch.equal(y, y2)
        hashes = {}
        with inference.tap_binary_inputs(lambda n, a: hashes.__setitem__(n, sighash.sign_hash_planes(a.P, a.M, a.shape[1]))):
            yt = FusedResNet(net)(x)
        assert torch.equal(yt, y)
        h = torch.stack([hashes[n] for n in names], 1).cpu().numpy()
        ref, href = fx["c1_logits_%d" % j], fx["c1_sign_hash_%d" % j]
        flipped = np.any(h != href, 1)
        dev_ = np.abs(y.cpu().numpy() - ref)
        ok = np.all(dev_ <= 1e-3 * np.abs(ref).max() + 1e-3 * np.abs(ref), 1)
        assert ok[~flipped].all()
        assert dev_[~flipped].max() <= 1e-4 * np.abs(ref).max()
        flipped_total += int(flipped.sum())
    print({"c1_images_with_a_sign_flip_of_128": flipped_total})
    assert flipped_total <= C1_FLIPPED


C1_FLIPPED = 0          # measured (round 6, MI355X): all 128 images carry the reference's integers at every layer
