"""The fused executors: ``FusedResNet`` / ``FusedBlocks``.

The drop-in path (``prepare_binary_model`` + ``model(x)`` layer by layer) evaluates every binary conv as
pack -> XNOR/popcount -> fp32 NCHW and leaves BatchNorm / ReLU / residual adds to torch: each of
those is a full fp32 round trip through HBM.  For the block structure of the reference's ResNets
(``bnn/models/layers/res_block.py:40-56``: conv-BN-ReLU-conv-BN-(+identity)-ReLU) all of that
folds into the conv kernel's epilogue (``bnn_hip_epilogue``), so activations travel between
binary layers as bit planes and only the residual stream is ever written in fp32:

    conv1 -> BN1 -> ReLU            -> packed only                (no fp32 tensor at all)
    conv2 -> BN2 -> +identity -> ReLU -> fp32 (next identity) + packed (next conv1 input)
    shortcut: AvgPool(ceil) -> sign -> 1x1 binary conv -> BN, computed inside the block's conv2 launch

``FusedResNet`` shares the weights of the model it wraps; packed weights and folded BN constants
are derived once (call ``refresh()`` after changing parameters).  Eval mode only.  19 launches per ResNet-18
forward, eager or as HIP graphs (``capture``: whole forward over a static input; ``forward_fresh``: stem launch on the
caller's tensor + graph of the rest).

Part of what used to be one module, ``bnn_amd/inference.py`` (now a facade): executor.py (this file), pipeline.py
(several batches in flight), tails.py (the per-layer path's one-launch tails), dispatch.py (what ``model(x)`` /
``block(x)`` pick by themselves).
"""
from __future__ import annotations

import collections
import contextlib
import itertools
import os
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn as nn

from . import fastpath, hipops, native
from .layers import Conv2d as BinaryConv2d
from .models.blocks import BasicBlock, Bottleneck, HBlock, PreBasicBlock
from .models.resnet import ResNet


class FusionError(RuntimeError):
    """The model (or a layer's recipe) is outside what the fused executor covers."""


def fold_bn(bn: nn.BatchNorm2d):
    """Eval-mode BatchNorm as one fused multiply-add per channel, ``y = fma(x, scale, shift)``, with the
    constants rounded exactly as the reference's forward rounds them.

    The reference evaluates ``bnN(...)`` with ATen's CPU kernel, which computes (all fp32)
    ``scale = weight * (1 / sqrt(var + eps))``, ``shift = fma(-mean, scale, bias)`` and
    ``out = fma(x, scale, shift)`` — measured: bit-identical on 1.6 M elements, whereas constants folded
    in double precision differ in the last bit for 34 % of the elements.  A last-bit difference in front
    of a ``sign()`` is a flipped activation, so the fold follows the reference's rounding, not the
    "more accurate" one.  Done on the host in numpy (IEEE-correct fp32 sqrt / divide), 64..512 values."""
    if not isinstance(bn, nn.BatchNorm2d) or bn.running_mean is None:
        raise FusionError(f"cannot fold {type(bn).__name__} (needs BatchNorm2d with running stats)")
    import numpy as np
    dev = bn.running_var.device
    var = bn.running_var.detach().float().cpu().numpy()
    mean = bn.running_mean.detach().float().cpu().numpy()
    gamma = bn.weight.detach().float().cpu().numpy() if bn.weight is not None else np.ones_like(var)
    beta = bn.bias.detach().float().cpu().numpy() if bn.bias is not None else np.zeros_like(var)
    inv = np.float32(1.0) / np.sqrt(var + np.float32(bn.eps), dtype=np.float32)
    scale = (gamma * inv).astype(np.float32)
    # fma(-mean, scale, bias): the product of two fp32 values is exact in fp64, one rounding to fp32 after the add
    shift = (beta.astype(np.float64) - mean.astype(np.float64) * scale.astype(np.float64)).astype(np.float32)
    return torch.from_numpy(scale).to(dev), torch.from_numpy(shift).to(dev)


_TAP = None


@contextlib.contextmanager
def tap_binary_inputs(fn):
    """Debug/test hook: while active, ``fn(layer_name, PackedAct)`` is called with the bit planes every
    binary convolution of a ``FusedResNet`` reads (eager launches only — not during graph replay)."""
    global _TAP
    prev, _TAP = _TAP, fn
    try:
        yield
    finally:
        _TAP = prev


@dataclass
class _Conv:
    layer: BinaryConv2d
    plan: fastpath.Plan
    weight: hipops.PackedWeight
    bn_scale: Optional[torch.Tensor]
    bn_shift: Optional[torch.Tensor]
    relu: bool
    prelu: Optional[torch.Tensor]
    name: str = ""
    throughput: bool = False   # BNN_HIP_FLAG_THROUGHPUT: several batches in flight (PipelinedInference)
    thr: Optional[torch.Tensor] = None   # integer sign thresholds of a BN + ReLU -> planes-only epilogue

    def run(self, act: hipops.PackedAct, *, residual=None, out_f32: bool, out_packed: bool, **epi):
        """``epi``: the pre-activation switches of ``hipops.bconv2d_fused`` (late residual, pack affine, ...)."""
        lay = self.layer
        if _TAP is not None:
            _TAP(self.name, act)
        # BN + ReLU -> planes only (conv1 of a BasicBlock): the sign bit is an integer compare of the dot
        thr = self.thr if (out_packed and not out_f32 and residual is None and not epi) else None
        return hipops.bconv2d_fused(
            act, self.weight, bias=lay.bias, post_scale=self.plan.scale, bn_scale=self.bn_scale,
            bn_shift=self.bn_shift, residual=residual, prelu=self.prelu, relu=self.relu,
            out_f32=out_f32, out_packed=out_packed, stride=lay.stride, padding=lay.padding,
            dilation=lay.dilation, throughput=self.throughput, sign_thresholds=thr, **epi)


def _plan_of(conv: nn.Module) -> fastpath.Plan:
    if not isinstance(conv, BinaryConv2d):
        raise FusionError(f"{type(conv).__name__} is not a binary Conv2d (run prepare_binary_model first)")
    plan = fastpath._recognise(conv, conv.out_channels)
    if plan is None or not fastpath._numeric_padding(conv):
        raise FusionError("layer recipe is not BasicInputBinarizer + XNORWeightBinarizer "
                          "(+ Identity | BasicScaleBinarizer)")
    return plan


def _is_float_layer(conv: nn.Module) -> bool:
    """True for a stock conv or a binary-class conv whose recipe is all-Identity (kept real-valued
    the way examples/cifar10.py:71 does it: custom_config_layers_name={'conv1': BConfig()})."""
    if not isinstance(conv, BinaryConv2d):
        return type(conv) is nn.Conv2d
    return (type(conv.activation_pre_process) is nn.Identity and type(conv.weight_pre_process) is nn.Identity
            and type(conv.activation_post_process).__name__ == "Identity")


_BLOCK_KINDS = {"BasicBlock": BasicBlock, "PreBasicBlock": PreBasicBlock, "Bottleneck": Bottleneck, "HBlock": HBlock}
_BLOCK_ATTRS = {BasicBlock: ("conv1", "bn1", "act1", "conv2", "bn2", "act2", "downsample"),
                PreBasicBlock: ("conv1", "bn1", "act1", "conv2", "bn2", "act2", "downsample"),
                Bottleneck: ("conv1", "bn1", "act1", "conv2", "bn2", "act2", "conv3", "bn3", "act3", "downsample"),
                HBlock: ("conv1", "bn1", "act1", "conv2", "bn2", "act2", "conv3", "bn3", "act3", "downsample")}


def _block_kind(blk: nn.Module):
    """The block family ``blk`` belongs to: one of this package's classes (exact type), or a class of the same NAME
    and attribute layout from another package — the reference's own ``bnn.models.layers`` blocks, which these mirror
    attribute for attribute (res_block.py:8-56,59-118,121-167, hierarchical_block.py:8-60).  Foreign classes are
    only ever fused after ``AutoFusion`` has checked the fused result against the model's own forward."""
    t = type(blk)
    if t in _BLOCK_ATTRS:
        return t
    kind = _BLOCK_KINDS.get(t.__name__)
    if kind is not None and all(hasattr(blk, a) for a in _BLOCK_ATTRS[kind]):
        return kind
    return None


def is_native_model(model: nn.Module) -> bool:
    """True when ``model`` and all its residual blocks are this package's classes (their forward is known)."""
    return isinstance(model, ResNet) and all(
        type(b) in _BLOCK_ATTRS or isinstance(b, nn.AvgPool2d)
        for stage in (model.layer1, model.layer2, model.layer3, model.layer4) for b in stage)


def resnet_shaped(model: nn.Module) -> bool:
    """The module layout of the reference's ``bnn.models.resnet.ResNet`` (resnet.py:93-101,147-164)."""
    stem = getattr(model, "stem_type", "basic")
    need = ("conv1", "maxpool", "layer1", "layer2", "layer3", "layer4", "avgpool", "fc") + (("bn1",) if stem == "basic" else ())
    return stem in ("basic", "dabnn") and all(isinstance(getattr(model, a, None), nn.Module) for a in need) and \
        all(isinstance(getattr(model, a), nn.Sequential) for a in ("layer1", "layer2", "layer3", "layer4"))


def _activation(act: nn.Module):
    if isinstance(act, nn.ReLU):
        return True, None
    if isinstance(act, nn.PReLU):
        return False, act.weight.detach().float().contiguous()
    raise FusionError(f"unsupported activation {type(act).__name__}")


class FusedResNet(nn.Module):
    """Inference executor for ``bnn_amd.models.ResNet`` built from ``BasicBlock`` (ResNet-18/34),
    ``Bottleneck`` (ResNet-50/101/152: mixed 1x1 / 3x3 binary convolutions), ``PreBasicBlock`` (the
    pre-activation dataflow of examples/imagenet.py) or ``HBlock`` (hierarchical blocks)."""

    def __init__(self, model: ResNet, use_mfma_stem: bool = True, overlap_shortcut: bool = True,
                 stem_fp16: bool = False, stem_exact_fp32: bool = False, throughput_mode: bool = False,
                 int_thresholds: bool = True, skip_dead_f32: bool = True, fold_shortcut: bool = True,
                 fuse_hblock: bool = True) -> None:
        super().__init__()
        # a hierarchical block as ONE launch (bnn_hip_hblock_forward) instead of a packing pass + three convolutions
        self.fuse_hblock = fuse_hblock and os.environ.get("BNN_AMD_FUSE_HBLOCK", "1") != "0"   # ("0": launch by launch, for A/B profiles)
        # the last conv of a block writes no fp32 tensor when the next block consumes sign planes only
        self.skip_dead_f32 = skip_dead_f32
        # BN + ReLU + sign of the conv1-type layers as an integer compare of the dot (same bits, fewer instructions)
        self.int_thresholds = int_thresholds
        # several batches in flight on other streams: kernels prefer fewer, longer waves (BNN_HIP_FLAG_THROUGHPUT)
        self.throughput_mode = throughput_mode
        self.stem_fp16 = stem_fp16           # opt-in: plain fp16 stem operands (~5e-4 relative error)
        self.use_mfma_stem = use_mfma_stem
        self.stem_exact_fp32 = stem_exact_fp32   # v_mfma_f32_16x16x4_f32: bit-for-bit an fp32 fmaf chain (slower)
        self.overlap_shortcut = overlap_shortcut
        # a down-sampling block's shortcut conv (AvgPool -> binary 1x1 -> BN) computed inside its last conv
        self.fold_shortcut = fold_shortcut
        self._side = {}
        if not resnet_shaped(model):
            raise FusionError("FusedResNet covers ResNets laid out like bnn.models.resnet.ResNet")
        self.model = model
        self._blocks: List[dict] = []
        self._graph = None
        self._split = collections.OrderedDict()   # (input shape, stream) -> _Split (forward_fresh)
        self.refresh()

    def _conv(self, conv, bn, act) -> _Conv:
        plan = _plan_of(conv)
        relu, prelu = (False, None) if act is None else _activation(act)
        if prelu is not None and prelu.numel() != conv.out_channels:
            prelu = prelu.expand(conv.out_channels).contiguous()
        scale, shift = (None, None) if bn is None else fold_bn(bn)
        pw = fastpath.packed_weight(conv, plan)
        thr = None
        if (self.int_thresholds and scale is not None and relu and prelu is None and conv.bias is None
                and plan.scale is None and not pw.has_zero):
            thr = hipops.sign_thresholds(pw, scale, shift)
        return _Conv(conv, plan, pw, scale, shift, relu, prelu, self._names.get(id(conv), ""), self.throughput_mode, thr)

    @staticmethod
    def _sign_through(act: nn.Module):
        """How ``sign(act(v))`` is produced from ``v``: (relu_planes, ok).  ReLU: P = v > 0, M = 0.
        PReLU with positive slopes: sign(prelu(v)) == sign(v).  Anything else is not fused."""
        if isinstance(act, nn.ReLU):
            return True
        if isinstance(act, nn.PReLU) and bool((act.weight.detach() > 0).all()):
            return False
        raise FusionError(f"cannot binarise through {type(act).__name__} in a fused epilogue")

    def refresh(self) -> None:
        """(Re)derive packed weights and folded BN constants from the wrapped model.  Runs by itself when a
        parameter/buffer was replaced or written in place (version counters); call it by hand after writes
        through ``.data`` (they bypass the counters), then ``capture`` again if a graph was captured."""
        native.require()
        m = self.model
        fastpath.invalidate(m, executors=False)
        if m.training:
            raise FusionError("FusedResNet is inference-only: call model.eval() first")
        dev = m.fc.weight.device
        if dev.type != "cuda":
            raise FusionError("FusedResNet needs the model on a HIP device")
        self._blocks = []
        self._split.clear()     # graphs behind the stem hold pointers to the derived data rebuilt below
        self._stem = None
        self._names = {id(mod): name for name, mod in m.named_modules()}
        mp = m.maxpool
        self._stem_module = getattr(m, "stem_type", "basic") != "basic"   # daBNN stem (resnet.py:10-47): m.conv1 is all of it
        if not self._stem_module and isinstance(mp, nn.MaxPool2d) and isinstance(m.bn1, nn.BatchNorm2d) \
                and mp.dilation in (1, (1, 1)) \
                and not mp.ceil_mode and isinstance(mp.kernel_size, int) and isinstance(mp.stride, int) \
                and isinstance(mp.padding, int):
            self._stem = (*fold_bn(m.bn1), (mp.kernel_size, mp.stride, mp.padding))
        c1 = m.conv1
        # the whole stem as one fp32-MFMA kernel when it is the canonical 7x7/2/3 conv + 3/2/1 pool
        self._stem_mfma = (self.use_mfma_stem and self._stem is not None and self._stem[2] == (3, 2, 1)
                           and isinstance(c1, nn.Conv2d) and c1.weight.shape == (64, 3, 7, 7)
                           and c1.stride == (2, 2) and c1.padding == (3, 3) and c1.dilation == (1, 1)
                           and c1.groups == 1 and c1.bias is None and c1.weight.dtype == torch.float32
                           and _is_float_layer(c1))
        # real-valued head (avgpool -> flatten -> fc, resnet.py:160-164) as one kernel when it is the canonical one
        fc, ap = m.fc, m.avgpool
        fc_float = _is_float_layer_linear(fc)
        self._head = None
        if fc_float and isinstance(ap, nn.AdaptiveAvgPool2d) and ap.output_size in (1, (1, 1)) \
                and fc.weight.dtype == torch.float32 and fc.in_features * 16 <= 160 * 1024:
            self._head = (fc.weight.detach().t().contiguous(), None if fc.bias is None else fc.bias.detach())
        for stage in (m.layer1, m.layer2, m.layer3, m.layer4):
            for blk in stage:
                self._add_block(blk)
        self._link_hblocks()
        self._graph = None
        self._sig = self._signature()

    def _link_hblocks(self) -> None:
        """One-launch form of every hierarchical block that qualifies (csrc/hblock.hip): ReLU activations (all sign
        planes non-negative), bias-free 3x3 / stride 1 / padding 1 convolutions without post scale or zero weights.  The
        launch also writes the NEXT block's input planes when that block is a hierarchical block behind a ReLU."""
        if not getattr(self, "fuse_hblock", True):
            return
        for i, b in enumerate(self._blocks):
            if b["kind"] != "h":
                continue
            convs = b["convs"]
            ok = all(b["relu"]) and all(
                c.layer.bias is None and c.plan.scale is None and not c.weight.has_zero and c.prelu is None and not c.relu
                and tuple(c.layer.kernel_size) == (3, 3) and tuple(c.layer.stride) == (1, 1)
                and tuple(c.layer.padding) == (1, 1) and tuple(c.layer.dilation) == (1, 1) for c in convs)
            planes = b["planes"]
            ok = ok and convs[0].layer.out_channels * 2 == planes and convs[1].layer.out_channels * 4 == planes \
                and convs[2].layer.out_channels * 4 == planes and planes % 64 == 0
            if not ok:
                continue
            nxt = self._blocks[i + 1] if i + 1 < len(self._blocks) else None
            nbn = nxt["bn"][0] if (nxt is not None and nxt["kind"] == "h" and nxt["relu"][0]
                                   and nxt["convs"][0].layer.in_channels == planes) else None
            try:
                b["hpack"] = hipops.hblock_pack(convs[0].weight, convs[1].weight, convs[2].weight, b["bn"][1], b["bn"][2], nbn)
            except native.NativeError:      # a width the one-launch kernel has no instance for
                b["hpack"] = None
            b["hgeo"] = {}
            # the shortcut convolution of a stage's first block inside the block's launch (bnn_hip_hblock_shortcut_forward):
            # a plain binary 1x1 (no bias, no scale, no zero weights) behind its own BatchNorm, and a next block in the stage
            b["hsc"] = None
            if b["hpack"] is not None and b["ds"] is not None and nbn is not None:
                sc = b["ds"][1]
                lay = sc.layer
                if (lay.bias is None and sc.plan.scale is None and not sc.weight.has_zero and sc.prelu is None and not sc.relu
                        and sc.bn_scale is None and tuple(lay.kernel_size) == (1, 1) and tuple(lay.stride) == (1, 1)
                        and tuple(lay.padding) == (0, 0) and lay.in_channels * 2 == planes
                        and os.environ.get("BNN_AMD_HBLOCK_SHORTCUT", "1") != "0"):
                    try:
                        b["hsc"] = hipops.hblock_shortcut_pack(sc.weight)
                    except native.NativeError:
                        b["hsc"] = None

    def _add_block(self, blk) -> None:
        """Derive the fused form of one residual block (appends to ``self._blocks``)."""
        if isinstance(blk, nn.AvgPool2d):   # HBlock stages pool in front (bnn_amd/models/resnet.py)
            self._blocks.append({"kind": "pool", "mod": blk})
            return
        kind = _block_kind(blk)
        if kind is PreBasicBlock:           # BN-conv-act, BN-conv-act, (+id)   (res_block.py:147-152)
            entry = {"kind": "pre", "bn1": fold_bn(blk.bn1), "bn2": fold_bn(blk.bn2),
                     "convs": [self._conv(blk.conv1, None, blk.act1), self._conv(blk.conv2, None, blk.act2)],
                     "ds": None, "pool": 0}
            self._shortcut(blk, entry)
            self._blocks.append(entry)
            return
        if kind is HBlock:                  # three BN-act-conv stages, cat, (+id)  (hierarchical_block.py:38-60)
            entry = {"kind": "h", "planes": blk.conv1.out_channels * 2,
                     "bn": [fold_bn(blk.bn1), fold_bn(blk.bn2), fold_bn(blk.bn3)],
                     "relu": [self._sign_through(a) for a in (blk.act1, blk.act2, blk.act3)],
                     "convs": [self._conv(c, None, None) for c in (blk.conv1, blk.conv2, blk.conv3)],
                     "ds": None}
            if blk.downsample is not None:  # BN -> binary 1x1 (no BN behind it)
                bn, conv = blk.downsample[0], blk.downsample[1]
                entry["ds"] = (fold_bn(bn), self._conv(conv, None, None))
            self._blocks.append(entry)
            return
        if kind is BasicBlock:           # conv-BN-act, conv-BN-(+id)-act
            convs = [self._conv(blk.conv1, blk.bn1, blk.act1), self._conv(blk.conv2, blk.bn2, blk.act2)]
        elif kind is Bottleneck:         # 1x1-BN-act, 3x3-BN-act, 1x1-BN-(+id)-act  (res_block.py:98-118)
            convs = [self._conv(blk.conv1, blk.bn1, blk.act1), self._conv(blk.conv2, blk.bn2, blk.act2),
                     self._conv(blk.conv3, blk.bn3, blk.act3)]
        else:
            raise FusionError(f"unsupported block {type(blk).__name__}")
        entry = {"kind": "post", "convs": convs, "ds": None, "pool": 0}
        self._shortcut(blk, entry)
        self._blocks.append(entry)

    def _shortcut(self, blk, entry) -> None:
        """AvgPool(ceil) -> binary 1x1 -> BN shortcut of bnn/models/resnet.py:128-133."""
        if blk.downsample is None:
            return
        pool, conv, bn = blk.downsample[0], blk.downsample[1], blk.downsample[2]
        if _is_float_layer(conv):
            # a real-valued shortcut convolution (examples/recepies/imagenet-baseline.yaml keeps
            # layerN.0.downsample.1 out of the binarisation): the branch runs as the torch modules it is
            entry["ds_float"] = blk.downsample
            return
        k = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
        if not (isinstance(pool, nn.AvgPool2d) and pool.ceil_mode and not pool.count_include_pad
                and pool.padding in (0, (0, 0))):
            raise FusionError("shortcut pooling must be AvgPool2d(k, k, ceil_mode=True, "
                              "count_include_pad=False)")
        entry["ds"] = self._conv(conv, bn, None)
        entry["pool"] = k

    @torch.no_grad()
    def _forward_impl(self, x: torch.Tensor) -> torch.Tensor:
        return self._back(*self._front(x))

    def _front(self, x: torch.Tensor, out=None):
        """The part that reads the input tensor: the real-valued stem (first layer stays float: examples/cifar10.py:71)
        -> (fp32 NCHW, sign planes).  ``out``: the results of an earlier call to overwrite (one-kernel stem only)."""
        m = self.model
        if self._stem_mfma:
            # a hierarchical block behind the stem reads sign(relu(bn1(t))): the stem kernel writes THOSE planes (its own
            # packing pass over the fp32 tensor disappears); the tag tells _run_h whose planes they are
            b0 = self._blocks[0] if self._blocks else None
            aff = None
            if (b0 is not None and b0["kind"] == "h" and b0.get("hpack") is not None and b0["relu"][0] and b0["ds"] is None
                    and not self.stem_exact_fp32 and _TAP is None and b0["hpack"].c_in == 64):
                aff = b0["bn"][0]
            t, pk = hipops.stem7x7(x, m.conv1.weight, self._stem[0], self._stem[1], exact_fp32=self.stem_exact_fp32,
                                   fp16=self.stem_fp16, out=out, pack_affine=aff)
            if aff is not None:
                pk._h_for = b0
            return t, pk
        if out is not None:
            raise FusionError("only the one-kernel stem writes into preallocated buffers")
        if self._stem is not None:   # the conv runs in the vendor library, its BN -> ReLU -> MaxPool -> sign tail in one pass
            t = m.conv1(x)
            return hipops.bn_relu_maxpool_pack(t, self._stem[0], self._stem[1], True, *self._stem[2])
        # any other stem runs as the torch modules it is (binary layers inside it one launch each); the residual blocks
        # behind it are fused all the same
        t = m.conv1(x) if self._stem_module else m.maxpool(m.relu(m.bn1(m.conv1(x))))
        return t, hipops.pack_act(t)

    def _back(self, t, packed) -> torch.Tensor:
        """Everything behind the stem: residual blocks + real-valued head (last layer stays float)."""
        m = self.model
        t = self._run_blocks(t, packed)
        if self._head is not None:
            return hipops.avgpool_fc(t, *self._head)
        return m.fc(torch.flatten(m.avgpool(t), 1))

    def _run_blocks(self, t, packed):
        """The residual blocks: ``t`` fp32 NCHW (may be None when only planes exist), ``packed`` its sign planes
        or None.  Returns the fp32 output of the last block."""
        last = len(self._blocks) - 1
        for i, b in enumerate(self._blocks):
            nxt = self._blocks[i + 1] if i < last else None
            if b["kind"] == "pool":
                if t is None and packed is not None and getattr(packed, "_h_for", None) is nxt:
                    continue                # the previous block's launch pooled and binarised its own output (_run_h)
                fused = self._pool_into_hblock(b["mod"], nxt, t)
                if fused is not None:       # the pool + both sign planes the next stage's first block reads, one pass
                    t, packed = fused
                    continue
                t, packed = b["mod"](t), None
                continue
            if b["kind"] == "pre":
                t, packed = self._run_pre(b, nxt, t, packed)
                continue
            if b["kind"] == "h":
                t, packed = self._run_h(b, t, packed, nxt, self._blocks[i + 2] if i + 2 <= last else None)
                continue
            if packed is None:
                packed = hipops.pack_act(t)
            side = None
            fold = None   # (PackedAct, PackedWeight, bn_scale, bn_shift) of a shortcut conv folded into the last conv
            if b["ds"] is not None and self._fold_applies(b, packed):
                # the shortcut conv (1x1 over the OR-pooled sign planes) is computed inside the block's last conv: no
                # fp32 shortcut tensor, no 1x1 launch (bnn_hip_epilogue.sc_*)
                # (pool 2: the kernel ORs the 2 x 2 windows of the block's input planes itself — no OR-pool launch)
                sc_in = packed if b["pool"] in (0, 1, 2) else hipops.orpool_packed(packed, b["pool"])
                if _TAP is not None:
                    _TAP(b["ds"].name, hipops.orpool_packed(packed, b["pool"]) if b["pool"] > 1 else packed)
                fold = (sc_in, b["ds"].weight, b["ds"].bn_scale, b["ds"].bn_shift)
                idn = None
            elif b["ds"] is not None:
                # the shortcut branch (HBM-bound avg-pool + a small 1x1 conv) is independent of the block's
                # first convs (ALU-bound): run it on a second stream and join before the residual is needed
                dev_ = packed.P.device      # (t is None when the previous block skipped its dead fp32 output)
                cur = torch.cuda.current_stream(dev_)
                side = self._side_stream(dev_) if self.overlap_shortcut else None
                if side is not None:
                    side.wait_stream(cur)
                with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                    if b["pool"] > 1 and packed.nonneg:   # sign(avg of non-negative values) = OR of the sign bits
                        sc_in = hipops.orpool_packed(packed, b["pool"])
                    elif b["pool"] > 1:
                        sc_in = hipops.avgpool_pack(t, b["pool"])
                    else:
                        sc_in = packed
                    idn, _ = b["ds"].run(sc_in, out_f32=True, out_packed=False)
                if side is not None:
                    if t is not None:
                        t.record_stream(side)
                    packed.P.record_stream(side)
                    sc_in.P.record_stream(side)
                    sc_in.M.record_stream(side)
            elif b.get("ds_float") is not None:
                idn = b["ds_float"](t)
            else:
                idn = t
            for c in b["convs"][:-1]:           # activations travel between binary layers as bit planes
                _, packed = c.run(packed, out_f32=False, out_packed=True)
            if side is not None:
                cur.wait_stream(side)
                idn.record_stream(cur)
            # the fp32 output is dead when the next block reads sign planes only: its convs always do, its shortcut
            # does when it is AvgPool -> binary 1x1 (-> OR-pool of the planes of a non-negative tensor, or no pooling)
            c2 = b["convs"][-1]
            dead_f32 = (self.skip_dead_f32 and nxt is not None and nxt["kind"] == "post" and nxt["ds"] is not None
                        and (nxt["pool"] <= 1 or (c2.relu and c2.prelu is None)))
            if fold is not None:
                t, packed = c2.run(packed, out_f32=True, out_packed=True, shortcut=fold)
            else:
                t, packed = c2.run(packed, residual=idn, out_f32=not dead_f32, out_packed=i != last)
        return t

    def _fold_applies(self, b, packed) -> bool:
        """Whether the block's shortcut convolution can be computed inside its last convolution.  The recipe part is
        decided once per block; the kernel part (``hipops.shortcut_fold_supported``) depends on the geometry — image
        size and the images one launch covers (large batches are split) — and is cached per geometry.  Needs:
        non-negative sign planes in front of the block (an OR-pool then IS the avg-pool's sign), a bias-free 1x1 /
        stride-1 shortcut conv without post scale or zero weights, a last conv whose kernel takes the fold and whose
        fp32 output and sign planes are both wanted (a block in the middle of the net)."""
        if not self.fold_shortcut or not packed.nonneg:
            return False
        ds, c2 = b["ds"], b["convs"][-1]
        lay, c2l = ds.layer, c2.layer
        if "fold_recipe" not in b:
            b["fold_recipe"] = bool(
                lay.bias is None and ds.plan.scale is None and not ds.relu and ds.prelu is None
                and tuple(lay.kernel_size) == (1, 1) and tuple(lay.stride) == (1, 1) and tuple(lay.padding) == (0, 0)
                and not ds.weight.has_zero and b is not self._blocks[-1] and len(b["convs"]) >= 2
                and all(c.relu and c.prelu is None for c in b["convs"][:-1])
                and c2.relu and c2.prelu is None and c2.layer.bias is None and c2.plan.scale is None)
            b["fold_geo"] = {}
        if not b["fold_recipe"]:
            return False
        N, _, H, W = packed.shape
        k = max(b["pool"], 1)
        ho, wo = -(-H // k), -(-W // k)
        n_launch = hipops.fused_launch_images(N, c2l.in_channels, ho, wo, c2l.out_channels, c2l.kernel_size,
                                              c2l.stride, c2l.padding, c2l.dilation)
        key = (n_launch, ho, wo)
        ok = b["fold_geo"].get(key)
        if ok is None:
            probe = hipops.PackedAct(packed.P, packed.M, (n_launch, c2l.in_channels, ho, wo), nonneg=True)
            ok = b["fold_geo"][key] = bool(hipops.shortcut_fold_supported(
                probe, c2.weight, lay.in_channels, c2l.stride, c2l.padding, c2l.dilation, throughput=c2.throughput))
        return ok

    def _side_stream(self, device) -> torch.cuda.Stream:
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=device)
        return self._side[key]

    def _run_pre(self, b, nxt, t, packed):
        """PreBasicBlock: the block input travels as fp32 ``t`` (shortcut) and as ``sign(bn1(t))``; the
        latter comes out of the previous block's last epilogue (its ``pack_scale`` = this ``bn1``)."""
        if packed is None or not getattr(packed, "_pre_bn_of", None) is b:
            packed = hipops.bn_act_pack(t, *b["bn1"], relu=False)
        if b["ds"] is not None:
            sc_in = hipops.avgpool_pack(t, b["pool"]) if b["pool"] > 1 else hipops.pack_act(t)
            idn, _ = b["ds"].run(sc_in, out_f32=True, out_packed=False)
        elif b.get("ds_float") is not None:
            idn = b["ds_float"](t)
        else:
            idn = t
        c1, c2 = b["convs"]
        _, p1 = c1.run(packed, out_f32=False, out_packed=True, pack_scale=b["bn2"][0], pack_shift=b["bn2"][1])
        if nxt is not None and nxt["kind"] == "pre":   # binarise for the next block's conv1 right here
            t, pk = c2.run(p1, residual=idn, out_f32=True, out_packed=True, residual_after_act=True,
                           pack_scale=nxt["bn1"][0], pack_shift=nxt["bn1"][1])
            pk._pre_bn_of = nxt
            return t, pk
        t, _ = c2.run(p1, residual=idn, out_f32=True, out_packed=False, residual_after_act=True)
        return t, None

    def _pool_into_hblock(self, pool, nxt, t):
        """``AvgPool2d(2, 2)`` in front of a hierarchical block that runs as one launch: the pooled tensor is only ever
        binarised — by the block's bn1 -> ReLU and by its shortcut's BatchNorm — so one pass writes both sets of planes
        and (when the block has a shortcut convolution) no fp32 tensor at all.  None: not that case."""
        if (_TAP is not None or nxt is None or nxt["kind"] != "h" or nxt.get("hpack") is None or t is None
                or not isinstance(pool, nn.AvgPool2d) or pool.kernel_size not in (2, (2, 2))
                or pool.stride not in (2, (2, 2)) or pool.padding not in (0, (0, 0))
                or t.shape[2] % 2 or t.shape[3] % 2 or not nxt["relu"][0]):
            return None
        N, C, H, W = t.shape
        hp = nxt["hpack"]
        if C != hp.c_in or not self._hblock_ok(nxt, N, H // 2, W // 2):
            return None
        ds_bn = nxt["ds"][0] if nxt["ds"] is not None else None
        p1, p2, tp = hipops.avgpool2_bn_pack2(t, nxt["bn"][0], True, ds_bn, False, out_f32=ds_bn is None)
        p1._h_for = nxt
        p1._ds_planes = p2
        return tp, p1

    def _hblock_ok(self, b, N, H, W) -> bool:
        """Whether the one-launch kernel covers this geometry; also decides which form of it runs (``b["hcl"]``): the
        small-image form (lanes = output channels, csrc/hblock_cl.hip) on 7 x 7 images, and on 14 x 14 images when other
        work shares the GPU — there whole images per workgroup win; alone, the pixel-lane kernel splits a 14 x 14 image over
        two workgroups and fills the chip (measured at batch 128, us per block: 7 x 7 53.6 vs 67.1; 14 x 14 shared 60.4 vs
        72.0, alone 60.4 vs 44.8)."""
        ok = b["hgeo"].get((N, H, W))
        if ok is None:
            hp = b["hpack"]
            cl = (os.environ.get("BNN_AMD_HBLOCK_CL", "1") != "0" and (H == 7 or self.throughput_mode)
                  and hipops.hblock_supported(N, hp.c_in, H, W, hp.planes, self.throughput_mode, channel_lanes=True))
            ok = cl or hipops.hblock_supported(N, hp.c_in, H, W, hp.planes, self.throughput_mode)
            b["hgeo"][(N, H, W)] = ok
            b.setdefault("hcl", {})[(N, H, W)] = cl
        return ok

    def _pooled_form(self, b, nxt, nxt2, N, H, W):
        """The constants of ``hipops.hblock_pool_forward`` when block ``b`` ends a stage — ``nxt`` is ``AvgPool2d(2, 2)`` and
        ``nxt2`` a one-launch hierarchical block with a shortcut convolution, so that nobody reads ``b``'s fp32 output —
        and the kernel covers the geometry; else None."""
        key = ("pool", N, H, W)
        hit = b["hgeo"].get(key)
        if hit is None:
            hit = False
            pool = nxt["mod"] if nxt is not None and nxt["kind"] == "pool" else None
            hp = b["hpack"]
            if (os.environ.get("BNN_AMD_HBLOCK_POOL", "1") != "0" and _TAP is None and pool is not None and nxt2 is not None
                    and nxt2["kind"] == "h" and nxt2.get("hpack") is not None and nxt2["ds"] is not None and nxt2["relu"][0]
                    and isinstance(pool, nn.AvgPool2d) and pool.kernel_size in (2, (2, 2)) and pool.stride in (2, (2, 2))
                    and pool.padding in (0, (0, 0)) and H % 2 == 0 and W % 2 == 0 and hp.c_in == hp.planes
                    and H * W >= int(os.environ.get("BNN_AMD_HBLOCK_POOL_MIN", "784"))
                    and nxt2["hpack"].c_in == hp.planes and self._hblock_ok(nxt2, N, H // 2, W // 2)
                    and hipops.hblock_pool_supported(N, hp.c_in, H, W, hp.planes, self.throughput_mode)):
                try:
                    hit = hipops.hblock_pool_consts(nxt2["bn"][0], nxt2["ds"][0], hp.planes)
                except native.NativeError:      # (a scale that cannot take the average's 1 / 4 exactly)
                    hit = False
            b["hgeo"][key] = hit
        return None if hit is False else hit

    def _run_h(self, b, t, packed=None, nxt=None, nxt2=None):
        """HBlock: three BN-act-conv stages write their slice of the concatenated output in place, each
        adds its slice of the shortcut and hands ``sign(act(bn_next(o_k)))`` to the next stage.  Returns
        ``(y, planes of the next block's input | None)``; ``packed``: what the previous block's launch left for this one.
        The last block of a stage returns ``(None, planes of the next stage's first block)``: pooled and binarised in the
        same launch (``_pooled_form``)."""
        mine = packed is not None and getattr(packed, "_h_for", None) is b    # planes the previous launch left for this block
        if b["ds"] is not None:
            (sa, sb), conv = b["ds"]
            sp = getattr(packed, "_ds_planes", None) if mine else None
            if sp is None:
                sp = hipops.bn_act_pack(t, sa, sb, relu=False)
            hp = b.get("hpack")
            if b.get("hsc") is not None and _TAP is None:
                N, _, H, W = sp.shape
                key = ("sc", N, H, W)
                ok = b["hgeo"].get(key)
                if ok is None:     # (in the form of the block that _hblock_ok chose for this geometry)
                    ok = b["hgeo"][key] = (self._hblock_ok(b, N, H, W) and
                                           hipops.hblock_shortcut_supported(N, hp.c_in, H, W, hp.planes, self.throughput_mode,
                                                                            channel_lanes=b["hcl"][(N, H, W)]))
                if ok:
                    if not mine:
                        packed = hipops.bn_act_pack(t, *b["bn"][0], relu=True)
                    y, pk = hipops.hblock_shortcut_forward(packed, hp, sp, b["hsc"], throughput=self.throughput_mode,
                                                           channel_lanes=b["hcl"][(N, H, W)])
                    pk._h_for = nxt
                    return y, pk
            idn, _ = conv.run(sp, out_f32=True, out_packed=False)
        else:
            idn = t
        c1, c2, c3 = b["convs"]
        hp = b.get("hpack")
        if hp is not None and _TAP is None:     # (a tap wants the planes in front of every convolution: launch by launch)
            N, _, H, W = idn.shape
            if self._hblock_ok(b, N, H, W):
                if not mine:
                    packed = hipops.bn_act_pack(t, *b["bn"][0], relu=True)
                kp = self._pooled_form(b, nxt, nxt2, N, H, W)
                if kp is not None:
                    p1, p2 = hipops.hblock_pool_forward(packed, hp, idn, kp, throughput=self.throughput_mode)
                    p1._h_for = nxt2
                    p1._ds_planes = p2
                    return None, p1
                y, pk = hipops.hblock_forward(packed, hp, idn, out_packed=hp.has_next, throughput=self.throughput_mode,
                                              channel_lanes=b["hcl"][(N, H, W)])
                if pk is not None:
                    pk._h_for = nxt
                return y, pk
        half = b["planes"] // 2
        quarter = c2.layer.out_channels
        p = hipops.bn_act_pack(t, *b["bn"][0], relu=b["relu"][0])
        y = torch.empty((t.shape[0], b["planes"], t.shape[2], t.shape[3]), dtype=torch.float32, device=t.device)
        late = dict(residual=idn, residual_after_act=True, pack_before_residual=True, out=y, out_f32=True)
        _, p = c1.run(p, out_packed=True, out_c_offset=0, pack_scale=b["bn"][1][0], pack_shift=b["bn"][1][1],
                      pack_relu=b["relu"][1], **late)
        _, p = c2.run(p, out_packed=True, out_c_offset=half, pack_scale=b["bn"][2][0], pack_shift=b["bn"][2][1],
                      pack_relu=b["relu"][2], **late)
        c3.run(p, out_packed=False, out_c_offset=half + quarter, **late)
        return y, None

    def _signature(self):
        """Changes whenever a parameter or buffer of the wrapped model is replaced or written in place
        (optimizer step, ``load_state_dict``, ``.to()``): the derived data must then be rebuilt."""
        # a fresh walk every time: a Parameter that was REPLACED (setattr, a swapped sub-module) is a new object with
        # its own storage, which a list captured at refresh() time would never see
        return tuple((id(t), t.data_ptr(), t._version)
                     for t in itertools.chain(self.model.parameters(), self.model.buffers()))

    def _slots(self):
        """(module dict, name) of every parameter / buffer slot of the wrapped model in the order of ``_signature()``
        (all parameters in module order, then all buffers; shared tensors once), collected once per refresh: the
        per-call staleness check reads the slots directly instead of walking the module tree (a ``net(x)`` call at
        batch 32 is host-bound: the tree walks were most of its 0.27 ms)."""
        mods = list(self.model.modules())
        slots, seen = [], set()
        for kind in ("_parameters", "_buffers"):
            for m in mods:
                d = getattr(m, kind)
                for k, t in d.items():
                    if t is not None and id(t) not in seen:
                        seen.add(id(t))
                        slots.append((d, k))
        return mods, slots

    def _unchanged(self) -> bool:
        """Cheap form of ``self._signature() == self._sig``: same triples read through the cached slots (no tree walk,
        early exit), plus the identity of every module's children (a swapped sub-module has other slots)."""
        cache = self.__dict__.get("_fast")
        if cache is None or cache[0] is not self._sig:
            if self._signature() != self._sig:
                return False
            mods, slots = self._slots()
            aligned = tuple((id(d[k]), d[k].data_ptr(), d[k]._version) for d, k in slots) == self._sig
            cache = self.__dict__["_fast"] = (self._sig, mods, slots if aligned else None,
                                              [(m._modules, tuple(m._modules.values())) for m in mods])
            return True
        _, mods, slots, children = cache
        if slots is None:                                   # (an unusual module tree: keep the plain comparison)
            return self._signature() == self._sig
        for (d, k), want in zip(slots, self._sig):
            t = d.get(k)
            if t is None or id(t) != want[0] or t._version != want[2] or t.data_ptr() != want[1]:
                return False
        for ch, snap in children:                           # a replaced / added / removed sub-module
            if len(ch) != len(snap) or any(a is not b for a, b in zip(ch.values(), snap)):
                return False
        return True

    def hooked(self) -> bool:
        """Forward (pre-)hooks on inner modules of the wrapped model (they would not fire in the fused executor)."""
        import torch.nn.modules.module as _mm
        if _mm._global_forward_hooks or _mm._global_forward_pre_hooks:
            return True
        cache = self.__dict__.get("_fast")
        mods = cache[1] if cache is not None and cache[0] is self._sig else list(self.model.modules())
        return any(m._forward_hooks or m._forward_pre_hooks for m in mods if m is not self.model)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_current()
        if self._graph is not None and x.shape == self._gx.shape:
            if x.data_ptr() != self._gx.data_ptr():   # callers that fill `static_input` in place skip the copy
                self._gx.copy_(x, non_blocking=True)
            self._graph.replay()
            return self._gy
        return self._forward_impl(x)

    @property
    def static_input(self):
        """The graph's input buffer (None before ``capture``).  Writing the batch into it in place
        (e.g. as the destination of the host-to-device copy) and passing it to ``forward`` replays
        the graph without the extra device-to-device copy of the input (154 MB at batch 256)."""
        return getattr(self, "_gx", None) if self._graph is not None else None

    def capture(self, example: torch.Tensor) -> "FusedResNet":
        """Record the whole forward (for this input shape) into a HIP graph; later calls with the
        same shape replay it — no per-kernel launch cost on the host."""
        self._graph = None
        self._gx = example.clone()
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(torch.cuda.current_stream(example.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._forward_impl(self._gx)
        torch.cuda.current_stream(example.device).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        # thread_local: HIP calls of OTHER threads (the RCCL watchdog of an initialised process group, data-loader
        # pinning threads) must not invalidate the capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._gy = self._forward_impl(self._gx)
        self._graph = g
        return self

    # ---- fresh input every call: the stem reads the CALLER's tensor, a HIP graph replays the rest ------------------
    MAX_SPLIT_GRAPHS = 4

    @property
    def reads_caller_tensor(self) -> bool:
        """True when ``forward_fresh`` applies (the stem is the one-kernel MFMA stem)."""
        return bool(self._stem_mfma)

    def _check_current(self) -> None:
        if not self._unchanged():                 # weights changed since the packed forms were derived
            recapture = self._graph is not None
            self.refresh()
            if recapture:
                self.capture(self._gx)

    @torch.no_grad()
    def capture_fresh(self, example: torch.Tensor) -> "_Split":
        """HIP graph of everything BEHIND the stem for inputs shaped like ``example``.  The stem stays an ordinary
        launch that reads whatever tensor the caller passes and writes the graph's two static inputs (its fp32 output
        and sign planes) — so a new input tensor per call costs no staging copy (154 MB at batch 256) and no
        re-capture, and the host issues two calls per forward instead of 21."""
        if not self._stem_mfma:
            raise FusionError("forward_fresh needs the one-kernel stem (7x7/2/3 conv + BN + ReLU + 3/2/1 max-pool)")
        dev = example.device
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            t0, pk0 = self._front(example)          # allocated on the stream that will replay (the key holds it)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._back(t0, pk0)
            cur.wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                gy = self._back(t0, pk0)
        return _Split(g, t0, pk0, gy)

    @torch.no_grad()
    def forward_fresh(self, x: torch.Tensor, clone: bool = True) -> torch.Tensor:
        """``forward`` for a caller that brings a NEW tensor every call (the reference's eval loop,
        examples/cifar10.py:147-149): stem launch on ``x`` + graph replay of the rest.  The graph for an input shape is
        captured on its first use (per stream; at most ``MAX_SPLIT_GRAPHS`` are kept).  Returns a fresh tensor unless
        ``clone=False`` (then: the graph's output buffer, overwritten by the next call with this shape)."""
        self._check_current()
        x = hipops._require_cuda_f32(x, "stem input")
        dev = x.device
        with torch.cuda.device(dev):
            key = (tuple(x.shape), torch.cuda.current_stream(dev).cuda_stream)
            sp = self._split.get(key)
            if sp is None:
                sp = self._split[key] = self.capture_fresh(x)
                while len(self._split) > self.MAX_SPLIT_GRAPHS:
                    self._split.popitem(last=False)
            else:
                self._split.move_to_end(key)
            self._front(x, out=(sp.t0, sp.pk0))
            sp.graph.replay()
            return sp.gy.clone() if clone else sp.gy


@dataclass
class _Split:
    graph: "torch.cuda.CUDAGraph"
    t0: torch.Tensor            # static fp32 output of the stem
    pk0: hipops.PackedAct       # static sign planes of the stem
    gy: torch.Tensor            # static logits


class FusedBlocks(FusedResNet):
    """The fused executor for a bare ``nn.Sequential`` of residual blocks (``BasicBlock`` / ``Bottleneck`` /
    ``PreBasicBlock`` / ``HBlock``, optionally ``nn.AvgPool2d`` between them): fp32 NCHW in, fp32 NCHW out, the
    activations between the binary layers travel as bit planes exactly as inside ``FusedResNet``.  For custom
    networks that keep their own stem / head, and for testing the cross-block dataflow on its own."""

    def __init__(self, blocks: nn.Sequential, throughput_mode: bool = False, int_thresholds: bool = True) -> None:
        nn.Module.__init__(self)
        self.skip_dead_f32 = True
        self.int_thresholds = int_thresholds
        self.throughput_mode = throughput_mode
        self.overlap_shortcut = True
        self.fold_shortcut = True
        self.fuse_hblock = True
        self._side = {}
        self.model = blocks
        self._blocks = []
        self._graph = None
        self._split = collections.OrderedDict()
        self._stem_mfma = False
        self.refresh()

    def refresh(self) -> None:
        native.require()
        fastpath.invalidate(self.model, executors=False)
        if self.model.training:
            raise FusionError("FusedBlocks is inference-only: call .eval() first")
        self._blocks = []
        self._names = {id(mod): name for name, mod in self.model.named_modules()}
        for blk in self.model:
            self._add_block(blk)
        self._link_hblocks()
        self._graph = None
        self._sig = self._signature()

    @torch.no_grad()
    def _forward_impl(self, x: torch.Tensor) -> torch.Tensor:
        return self._run_blocks(hipops._require_cuda_f32(x, "activation"), None)


def _is_float_layer_linear(fc: nn.Module) -> bool:
    """A stock ``nn.Linear``, or a binary-class Linear whose recipe is all-Identity (examples/cifar10.py:71 keeps ``fc``
    real-valued that way)."""
    return type(fc) is nn.Linear or (
        isinstance(fc, nn.Linear) and type(getattr(fc, "activation_pre_process", None)) is nn.Identity
        and type(getattr(fc, "weight_pre_process", None)) is nn.Identity
        and type(getattr(fc, "activation_post_process", None)).__name__ == "Identity")
