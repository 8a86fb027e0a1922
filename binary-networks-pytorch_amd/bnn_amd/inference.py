"""Fused inference executor for binary ResNets.

The drop-in path (``prepare_binary_model`` + ``model(x)``) evaluates every binary conv as
pack -> XNOR/popcount -> fp32 NCHW and leaves BatchNorm / ReLU / residual adds to torch: each of
those is a full fp32 round trip through HBM.  For the block structure of the reference's ResNets
(``bnn/models/layers/res_block.py:40-56``: conv-BN-ReLU-conv-BN-(+identity)-ReLU) all of that
folds into the conv kernel's epilogue (``bnn_hip_epilogue``), so activations travel between
binary layers as bit planes and only the residual stream is ever written in fp32:

    conv1 -> BN1 -> ReLU            -> packed only                (no fp32 tensor at all)
    conv2 -> BN2 -> +identity -> ReLU -> fp32 (next identity) + packed (next conv1 input)
    shortcut: AvgPool(ceil) -> sign  fused into one kernel; 1x1 binary conv -> BN -> fp32

``FusedResNet`` shares the weights of the model it wraps; packed weights and folded BN constants
are derived once (call ``refresh()`` after changing parameters).  Eval mode only.

What is in this module, top to bottom:

    fold_bn, tap_binary_inputs          eval-mode BatchNorm as one fma per channel (ATen's rounding); debug tap
    FusedResNet / FusedBlocks           the executors: 18 launches per ResNet-18 forward, eager or as HIP graphs
                                        (``capture``: whole forward over a static input; ``forward_fresh``: stem launch
                                        on the caller's tensor + graph of the rest)
    concurrent_streams                  streams that are checked to really run beside each other
    PipelinedInference                  several batches in flight across calls (what ``bench.py`` replays)
    per_layer_forward, no_model_fusion  switches for the tiers below (tests, bench engines)
    eval_tail / eval_stem / eval_head   the per-layer path's one-launch tails (``library_tails`` switches them off)
    TwoHalves                           two halves of one batch in flight inside one call
    BlockFusion, AutoFusion             what ``block(x)`` / ``model(x)`` dispatch to by themselves (the drop-in tiers)
    install_auto_fusion                 the same dispatch for ResNets of other packages (``prepare_binary_model``)
"""
from __future__ import annotations

import collections
import contextlib
import itertools
import os
import threading
import types
import warnings
import weakref
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
                 int_thresholds: bool = True, skip_dead_f32: bool = True, fold_shortcut: bool = True) -> None:
        super().__init__()
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
        fastpath.invalidate(m)
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
        self._graph = None
        self._sig = self._signature()

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
            return hipops.stem7x7(x, m.conv1.weight, self._stem[0], self._stem[1],
                                  exact_fp32=self.stem_exact_fp32, fp16=self.stem_fp16, out=out)
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
                t, packed = b["mod"](t), None
                continue
            if b["kind"] == "pre":
                t, packed = self._run_pre(b, nxt, t, packed)
                continue
            if b["kind"] == "h":
                t, packed = self._run_h(b, t), None
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

    def _run_h(self, b, t):
        """HBlock: three BN-act-conv stages write their slice of the concatenated output in place, each
        adds its slice of the shortcut and hands ``sign(act(bn_next(o_k)))`` to the next stage."""
        if b["ds"] is not None:
            (sa, sb), conv = b["ds"]
            idn, _ = conv.run(hipops.bn_act_pack(t, sa, sb, relu=False), out_f32=True, out_packed=False)
        else:
            idn = t
        c1, c2, c3 = b["convs"]
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
        return y

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
        self._side = {}
        self.model = blocks
        self._blocks = []
        self._graph = None
        self._split = collections.OrderedDict()
        self._stem_mfma = False
        self.refresh()

    def refresh(self) -> None:
        native.require()
        fastpath.invalidate(self.model)
        if self.model.training:
            raise FusionError("FusedBlocks is inference-only: call .eval() first")
        self._blocks = []
        self._names = {id(mod): name for name, mod in self.model.named_modules()}
        for blk in self.model:
            self._add_block(blk)
        self._graph = None
        self._sig = self._signature()

    @torch.no_grad()
    def _forward_impl(self, x: torch.Tensor) -> torch.Tensor:
        return self._run_blocks(hipops._require_cuda_f32(x, "activation"), None)


STREAM_PROBE_LOG: List[float] = []      # (chain time of candidate + chosen stream) / (one chain), per candidate probed


def concurrent_streams(device: torch.device, n: int, against=(), tries: int = 12) -> List["torch.cuda.Stream"]:
    """``n`` HIP streams whose work really runs beside each other's and beside that of the streams in ``against``.  The
    runtime maps streams onto a handful of hardware queues (four by default), two streams that share a queue run back to
    back, and which queue a stream gets depends on how many streams the process used before (`tools/exp_stream_queues.py`).
    So every candidate is checked against the streams already chosen — a chain of four ~0.1 ms spin kernels
    (`torch.cuda._sleep`) on each, submitted alternately, must take about as long as one chain — and dropped if it is
    serialised behind one of them (kernels of ONE stream carry the packet barrier bit, which orders them behind
    everything in front of them in the hardware queue, the other stream's kernels included; a single kernel per stream
    would overlap even on a shared queue); after ``tries`` candidates the best effort is kept.  Synchronises the device
    (a few milliseconds, once per executor)."""
    import time as _time
    with torch.cuda.device(device):
        if not hasattr(torch.cuda, "_sleep"):
            return [torch.cuda.Stream(device=device) for _ in range(n)]
        cyc, reps = 200_000, 4

        def spin(streams) -> float:
            torch.cuda.synchronize(device)
            t0 = _time.perf_counter()
            for _ in range(reps):
                for st in streams:
                    with torch.cuda.stream(st):
                        torch.cuda._sleep(cyc)
            torch.cuda.synchronize(device)
            return _time.perf_counter() - t0
        fixed = list(against)
        chosen: List["torch.cuda.Stream"] = []
        if not fixed:
            chosen.append(torch.cuda.Stream(device=device))
        ref = (fixed + chosen)[0]
        spin([ref])                                    # (first use of a stream: lazy initialisation)
        one = min(spin([ref]) for _ in range(2))
        keep = []                                      # rejected candidates stay alive until the end: the pool hands out others
        for _ in range(tries):
            if len(chosen) == n:
                break
            cand = torch.cuda.Stream(device=device)
            spin([cand])
            ratio = max(min(spin([c, cand]) for _ in range(2)) for c in fixed + chosen) / one
            STREAM_PROBE_LOG.append(round(ratio, 2))
            if ratio < 1.3:
                chosen.append(cand)
            else:
                keep.append(cand)
        while len(chosen) < n:                         # nothing better found: serialised streams still give correct results
            chosen.append(keep.pop() if keep else torch.cuda.Stream(device=device))
        return chosen


class PipelinedInference:
    """Several batches in flight: ``n_streams`` graph-captured copies of the fused executor (sharing the
    model's weights, each with its own static input / activations) replayed round-robin on their own HIP
    streams.  One forward is a chain of kernels bound by different units — the stem by the matrix cores and
    LDS, the 64-channel convs by HBM, the rest by the integer ALU — so two batches interleave well:
    +20 % images/s over one stream on MI355X (three streams: +12 %).

        pipe = PipelinedInference(model, example_batch)
        for i, batch in enumerate(loader):
            pipe.input(i).copy_(batch, non_blocking=True)      # e.g. the H2D copy target
            logits = pipe.launch(i)                            # valid after pipe.wait(i) / synchronize()
    """

    def __init__(self, model: nn.Module, example: torch.Tensor, n_streams: int = 2, fresh_input: bool = False,
                 **fused_kwargs) -> None:
        """``fresh_input``: every ``launch(i, x)`` reads the tensor the caller passes (stem launch on ``x`` + HIP graph
        of the rest, ``FusedResNet.forward_fresh``) instead of a static input buffer the caller has to fill."""
        if n_streams < 1:
            raise ValueError("n_streams must be >= 1")
        dev = example.device
        env = os.environ.get("BNN_AMD_THROUGHPUT")   # "0" / "1": override for experiments
        fused_kwargs.setdefault("throughput_mode", n_streams > 1 if env is None else env == "1")
        self.fresh_input = fresh_input
        self.streams = concurrent_streams(dev, n_streams)
        self.engines: List[FusedResNet] = []
        cur = torch.cuda.current_stream(dev)
        for s in self.streams:
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                eng = FusedResNet(model, **fused_kwargs)
                if fresh_input:
                    eng.forward_fresh(example, clone=False)      # captures the graph behind the stem for this stream
                else:
                    eng.capture(example)
                self.engines.append(eng)
        for s in self.streams:
            cur.wait_stream(s)

    def __len__(self) -> int:
        return len(self.engines)

    def input(self, i: int) -> torch.Tensor:
        return self.engines[i % len(self.engines)].static_input

    def stream(self, i: int) -> torch.cuda.Stream:
        return self.streams[i % len(self.streams)]

    def launch(self, i: int, x: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Replay slot ``i % n`` on its stream — with whatever its static input holds, or (``fresh_input``) reading
        ``x`` directly (the caller keeps ``x`` alive and unchanged until the slot's stream has passed the launch; work
        that produced ``x`` on another stream must be ordered before it by the caller, e.g.
        ``pipe.stream(i).wait_stream(...)``).  Returns the slot's logits buffer (overwritten by the next launch of
        the same slot)."""
        k = i % len(self.engines)
        with torch.cuda.stream(self.streams[k]):
            if self.fresh_input:
                if x is None:
                    raise ValueError("PipelinedInference(fresh_input=True).launch needs the input tensor")
                return self.engines[k].forward_fresh(x, clone=False)
            if x is not None:
                self.engines[k].static_input.copy_(x, non_blocking=True)
            return self.engines[k](self.engines[k].static_input)

    def wait(self, i: int) -> None:
        torch.cuda.current_stream(self.streams[0].device).wait_stream(self.streams[i % len(self.streams)])

    def synchronize(self) -> None:
        for s in self.streams:
            s.synchronize()


_PER_LAYER = 0


@contextlib.contextmanager
def per_layer_forward():
    """While active (process-wide), ``model(x)`` never takes the fused executor: every layer runs on its own, the way
    the reference evaluates a model (tests and ``bench.py --engine layerwise`` compare the two paths with it)."""
    global _PER_LAYER
    _PER_LAYER += 1
    try:
        yield
    finally:
        _PER_LAYER -= 1


# ---------------------------------------------------------------------------------------------------------------------
# The per-layer path's tails: what surrounds the binary convolutions when no fused executor takes the model or the block
# (forward hooks on inner layers, `per_layer_forward()`, a block the executors do not cover).  The reference evaluates
# `act(bn(conv(x)) + shortcut)` as four library passes over the fp32 tensor (res_block.py:40-56) and the stem as
# conv -> bn -> relu -> maxpool (resnet.py:150-153); on a HIP device the blocks of `bnn_amd.models` call these helpers
# instead: one launch per tail, the stem as its MFMA kernel — the same float operations as the fused executors, so
# the per-layer path, the block tier and the whole-model tier agree bit for bit on everything but the real-valued
# layers' library kernels.  `BNN_AMD_EVAL_TAILS=0` / `library_tails()` give the library's own modules back.
# ---------------------------------------------------------------------------------------------------------------------
EVAL_TAILS = os.environ.get("BNN_AMD_EVAL_TAILS", "1") != "0"
_LIBRARY_TAILS = 0


@contextlib.contextmanager
def library_tails():
    """While active (process-wide), BatchNorm / residual add / ReLU / the stem of the per-layer path are the library's
    own modules (A/B runs and cross-checks; ``bench.py --engine layerwise_library``)."""
    global _LIBRARY_TAILS
    _LIBRARY_TAILS += 1
    try:
        yield
    finally:
        _LIBRARY_TAILS -= 1


def _no_hooks(*mods) -> bool:
    import torch.nn.modules.module as _mm
    if _mm._global_forward_hooks or _mm._global_forward_pre_hooks:
        return False
    return not any(m is not None and (m._forward_hooks or m._forward_pre_hooks) for m in mods)


# derived constants of the tails, per module: kept OUTSIDE the modules (weak keys), so that `state_dict()`, pickling and
# `copy.deepcopy` of a model see nothing of them
_FOLDS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
_HEAD_WEIGHTS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def cached_fold(bn: nn.BatchNorm2d):
    """``fold_bn(bn)`` on ``bn``'s device, kept until one of the module's four tensors is written or replaced."""
    ts = (bn.running_mean, bn.running_var, bn.weight, bn.bias)
    key = tuple((id(t), t._version, t.data_ptr()) if t is not None else None for t in ts) + (bn.eps,)
    c = _FOLDS.get(bn)
    if c is None or c[0] != key:
        c = _FOLDS[bn] = (key, fold_bn(bn))
    return c[1]


def _tails_wanted(x: torch.Tensor) -> bool:
    return (EVAL_TAILS and not _LIBRARY_TAILS and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and not torch.is_grad_enabled())


def eval_tail(x: torch.Tensor, bn: nn.Module, act: Optional[nn.Module] = None,
              residual: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """``act(bn(x) (+ residual))`` of an eval-mode block in one launch, or None (not applicable: the caller runs the
    modules).  A parametric activation runs as its own module behind the fused BatchNorm + add."""
    if not _tails_wanted(x) or type(bn) is not nn.BatchNorm2d or bn.training or bn.running_mean is None:
        return None
    relu = type(act) is nn.ReLU
    if not _no_hooks(bn, act if relu else None):
        return None
    if residual is not None and (residual.shape != x.shape or residual.dtype != x.dtype or residual.device != x.device):
        return None
    scale, shift = cached_fold(bn)
    y = hipops.bn_act(x, scale, shift, relu=relu, residual=residual)
    return y if act is None or relu else act(y)


def eval_stem(model: nn.Module, x: torch.Tensor) -> Optional[torch.Tensor]:
    """``maxpool(relu(bn1(conv1(x))))`` of a ``ResNet`` with the basic stem (resnet.py:93-96,150-153) as the MFMA stem
    kernel — fp32 out, no sign planes — or None (not applicable)."""
    if not _tails_wanted(x) or getattr(model, "stem_type", None) != "basic":
        return None
    conv, bn, relu, pool = model.conv1, model.bn1, model.relu, model.maxpool
    if (not isinstance(conv, nn.Conv2d) or not _is_float_layer(conv) or tuple(conv.weight.shape) != (64, 3, 7, 7)
            or conv.bias is not None
            or conv.stride != (2, 2) or conv.padding != (3, 3) or conv.dilation != (1, 1) or conv.groups != 1
            or conv.padding_mode != "zeros" or conv.weight.dtype != torch.float32 or x.shape[1] != 3
            or type(bn) is not nn.BatchNorm2d or bn.training or bn.running_mean is None or type(relu) is not nn.ReLU
            or type(pool) is not nn.MaxPool2d or _pair2(pool.kernel_size) != (3, 3) or _pair2(pool.stride) != (2, 2)
            or _pair2(pool.padding) != (1, 1) or _pair2(pool.dilation) != (1, 1) or pool.ceil_mode
            or pool.return_indices or not _no_hooks(conv, bn, relu, pool)):
        return None
    scale, shift = cached_fold(bn)
    y, _ = hipops.stem7x7(x, conv.weight, scale, shift, out_f32=True, out_packed=False)
    return y


def eval_head(model: nn.Module, x: torch.Tensor) -> Optional[torch.Tensor]:
    """``fc(flatten(avgpool(x), 1))`` of a ``ResNet`` (resnet.py:160-164) as the head kernel (``bnn_hip_avgpool_fc_f32``:
    global average pool + real-valued Linear in one launch), or None (not applicable).  The transposed weight is kept
    until the weight is written or replaced."""
    if not _tails_wanted(x):
        return None
    ap, fc = model.avgpool, model.fc
    if (not isinstance(ap, nn.AdaptiveAvgPool2d) or ap.output_size not in (1, (1, 1)) or not isinstance(fc, nn.Linear)
            or not _is_float_layer_linear(fc) or fc.weight.dtype != torch.float32 or fc.in_features != x.shape[1]
            or fc.in_features * 16 > 160 * 1024 or not _no_hooks(ap, fc)):
        return None
    w = fc.weight
    key = (id(w), w._version, w.data_ptr())
    c = _HEAD_WEIGHTS.get(fc)
    if c is None or c[0] != key:
        c = _HEAD_WEIGHTS[fc] = (key, w.detach().t().contiguous())
    return hipops.avgpool_fc(x, c[1], None if fc.bias is None else fc.bias.detach())


def _is_float_layer_linear(fc: nn.Module) -> bool:
    """A stock ``nn.Linear``, or a binary-class Linear whose recipe is all-Identity (examples/cifar10.py:71 keeps ``fc``
    real-valued that way)."""
    return type(fc) is nn.Linear or (
        isinstance(fc, nn.Linear) and type(getattr(fc, "activation_pre_process", None)) is nn.Identity
        and type(getattr(fc, "weight_pre_process", None)) is nn.Identity
        and type(getattr(fc, "activation_post_process", None)).__name__ == "Identity")


def _pair2(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class TwoHalves:
    """One call, two batches in flight: ``model(x)`` with the batch cut in two halves that run on two HIP streams (the
    caller's and one more), each as "stem launch on its half of the caller's tensor + HIP graph of the rest" (``FusedResNet.forward_fresh``) on
    an executor in throughput mode.  A forward is a chain of kernels bound by different units (stem: matrix cores, the
    64-channel convs: HBM, the rest: integer ALU) with a tail after every launch; a second half-batch fills what the
    first leaves idle — what ``PipelinedInference`` does across calls, done inside one call, so that the reference's
    own ``net(x)`` loop (examples/cifar10.py:147-149) gets it without knowing.  Bit-identical logits (images are
    independent).  ``AutoFusion`` uses it from ``MIN_PIXELS`` input pixels on (``BNN_AMD_SPLIT_BATCH=0`` turns it off)."""

    # from this many input pixels on (batch x height x width): measured on MI355X at 224 x 224 (k images/s, two halves vs
    # one batch) — batch 64: 132 vs 143, batch 128: 183 vs 175, batch 256: 227 vs 213
    MIN_PIXELS = 128 * 224 * 224

    def __init__(self, model: nn.Module, device: torch.device) -> None:
        # Half 0 runs on the CALLER's stream, half 1 on one side stream.  (Until late round 4 both halves had a side
        # stream of their own.  Which hardware queues the three streams then sat on — a matter of how many streams the
        # process had used before — made the same call 1.10, 1.22, 1.26 or 1.5-1.7 ms per batch, the last slower than the
        # halves one after the other; the fast layouts were those where half 0's stream shared the caller's queue
        # (`tools/exp_stream_queues.py`).  One stream less, two cross-stream waits less, and nothing left to chance but
        # the side stream's queue, which `concurrent_streams` checks against the caller's.)
        cur = torch.cuda.current_stream(device)
        self.side = concurrent_streams(device, 1, against=[cur])[0]
        self.engines = [FusedResNet(model, throughput_mode=True)]
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            self.engines.append(FusedResNet(model, throughput_mode=True))
        cur.wait_stream(self.side)
        self._lock = threading.Lock()       # one call at a time enqueues its two halves (callers on different streams)
        self._done: Optional[torch.cuda.Event] = None   # the previous call has read both halves' output buffers

    @staticmethod
    def wanted(x: torch.Tensor) -> bool:
        return (x.shape[0] >= 2 and x.shape[0] * x.shape[2] * x.shape[3] >= TwoHalves.MIN_PIXELS
                and os.environ.get("BNN_AMD_SPLIT_BATCH", "1") != "0")

    def _streams(self, device):
        return torch.cuda.current_stream(device), self.side

    def captured(self, x: torch.Tensor) -> bool:
        h = (x.shape[0] + 1) // 2
        keys = [((n,) + tuple(x.shape[1:]), st.cuda_stream) for n, st in zip((h, x.shape[0] - h), self._streams(x.device))]
        return all(k in e._split for k, e in zip(keys, self.engines))

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        dev = x.device
        cur, side = self._streams(dev)
        h = (x.shape[0] + 1) // 2
        with self._lock:
            if self._done is not None:           # a caller on ANOTHER stream may still be reading the halves' static
                cur.wait_event(self._done)       # output buffers (its torch.cat): overwrite them only behind it
            side.wait_stream(cur)                # the caller's tensor is ready on the caller's stream
            with torch.cuda.stream(side):        # (the side half first: it is under way while the host issues the other)
                y1 = self.engines[1].forward_fresh(x[h:], clone=False)
            y0 = self.engines[0].forward_fresh(x[:h], clone=False)
            cur.wait_stream(side)                # (also orders the caller's later reuse of x behind both halves)
            out = torch.cat((y0, y1), 0)
            self._done = torch.cuda.Event()
            self._done.record(cur)
        return out


_NO_MODEL_FUSION = 0


@contextlib.contextmanager
def no_model_fusion():
    """While active, whole-model fusion (``AutoFusion``) is off but residual blocks still fuse themselves
    (``BlockFusion``): what a network that is NOT laid out like the reference's ResNet gets (``bench.py --engine
    blockwise``)."""
    global _NO_MODEL_FUSION
    _NO_MODEL_FUSION += 1
    try:
        yield
    finally:
        _NO_MODEL_FUSION -= 1


class BlockFusion:
    """The second tier of the drop-in dispatch: a residual block of ``bnn_amd.models`` (``BasicBlock``, ``Bottleneck``,
    ``PreBasicBlock``, ``HBlock``) called on its own — inside a network that is not laid out like the reference's
    ``ResNet`` (a CIFAR-style three-stage ResNet-20, a custom backbone), or behind a stem the whole-model executor does
    not cover — evaluates itself as ``FusedBlocks([block])``: fp32 NCHW in -> ``pack_act`` -> the block's convolutions
    with BatchNorm / activation / residual add in their epilogues (activations between them as bit planes) -> fp32 NCHW
    out; 3 launches and 3 fp32 passes over HBM for a ``BasicBlock`` instead of 8 kernels and 13 passes.  Same conditions
    as ``AutoFusion`` (eval, no autograd, fp32 on a HIP device, no hooks on inner modules, not a replica); one instance
    per block in ``block.__dict__['_bnn_auto_block']``."""

    def __init__(self) -> None:
        self.engine: Optional["FusedBlocks"] = None
        self.failed_sig = None
        self.calls = {"fused": 0, "declined": 0}
        self.lock = threading.Lock()

    def __deepcopy__(self, memo):
        return BlockFusion()

    def __reduce__(self):
        return (BlockFusion, ())

    def run(self, block: nn.Module, x: torch.Tensor) -> Optional[torch.Tensor]:
        if (block.training or torch.is_grad_enabled() or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4
                or x.shape[0] == 0 or getattr(block, "_is_replica", False) or _PER_LAYER
                or os.environ.get("BNN_AMD_AUTOFUSE", "1") == "0" or not native.available()):
            self.calls["declined"] += 1
            return None
        with self.lock:
            eng = self.engine
            # (inside a caller's own graph capture nothing may be built or re-derived: an executor that is ready runs —
            # its launches are plain kernels on the capturing stream — anything else falls to the per-layer path)
            if torch.cuda.is_current_stream_capturing() and (eng is None or not eng._unchanged()):
                self.calls["declined"] += 1
                return None
            if eng is None:
                sig = _param_signature(block)
                if self.failed_sig == sig:
                    self.calls["declined"] += 1
                    return None
                try:
                    seq = nn.Sequential(block)
                    seq.training = False            # (a new container starts in training mode; the block is in eval mode)
                    eng = self.engine = FusedBlocks(seq)
                except FusionError:
                    self.failed_sig = sig
                    self.calls["declined"] += 1
                    return None
            if AutoFusion._hooked(block) or next(block.parameters()).device != x.device:
                self.calls["declined"] += 1
                return None
        try:
            y = eng(x)
        except FusionError:
            with self.lock:
                self.engine, self.failed_sig = None, _param_signature(block)
            self.calls["declined"] += 1
            return None
        self.calls["fused"] += 1
        return y


def auto_block_forward(block: nn.Module, x: torch.Tensor) -> Optional[torch.Tensor]:
    """Called at the top of the residual blocks' ``forward``: the fused block's output, or None -> its own forward."""
    st = block.__dict__.get("_bnn_auto_block")
    if st is None:
        st = block.__dict__["_bnn_auto_block"] = BlockFusion()
    return st.run(block, x)


class AutoFusion:
    """What makes the reference's own call fast: ``net = prepare_binary_model(...)``, ``net.eval()``, ``net(x)`` under
    ``torch.no_grad()`` (examples/cifar10.py:71,140-149) runs the fused executor instead of one launch per layer plus
    torch BatchNorm / ReLU / add round trips through HBM.

    One instance lives in ``model.__dict__['_bnn_auto']`` (not a sub-module: ``state_dict`` and ``repr`` are those of
    the reference).  ``run(model, x)`` returns the logits, or ``None`` when the call has to take the model's own
    per-layer forward: training mode or autograd recording, CPU / non-fp32 input, forward hooks registered on inner
    modules (they would not fire), a
    model the executor does not cover (``FusionError``, remembered until the parameters change), or
    ``BNN_AMD_AUTOFUSE=0``.

    Policy: the first batch of a given shape runs the fused launches eagerly (18 for ResNet-18); from the second one
    on the stem reads the caller's tensor and a HIP graph replays the rest (``FusedResNet.forward_fresh``) — the last,
    ragged batch of an epoch never pays for a capture.  A model built from classes of another package (same names
    and layout: the reference's ``bnn.models``) is fused only after its first fused result has been checked against
    its own forward on the same input (logits within ``VERIFY_TOL`` relative to the largest one).

    ``nn.DataParallel`` (examples/cifar10.py:74-77) replicates the model on every forward; the replicas share this object
    (``replicate`` copies ``__dict__``) and get ONE executor per device, derived from the first replica seen there
    and valid until a parameter of the master changes — so the reference's multi-GPU script runs the fused executor on
    every GPU, not the per-layer path."""

    VERIFY_TOL = 2e-2       # a flipped sign() moves a logit by a few per cent; a wrong graph by O(1)
    CAPTURE_AFTER = 1       # eager calls of a shape before its graph is captured

    def __init__(self, owner: Optional[nn.Module] = None) -> None:
        self.engine: Optional[FusedResNet] = None
        self.failed_sig = None          # parameter signature for which fusion was refused
        self.reason: Optional[str] = None
        self.verified = False
        self.seen = collections.Counter()
        self.lock = threading.RLock()   # re-entrant: the first-call check runs the model's own forward under it
        self._verifying = False
        self.calls = {"graph": 0, "eager": 0, "declined": 0}
        # the model this state belongs to.  nn.DataParallel replicas (``replicate`` copies ``__dict__`` shallowly) share
        # the object with the model they were made from: their executors live here, one per device
        self.owner = None if owner is None else weakref.ref(owner)
        self.replica_engines = {}       # device -> (master parameter signature, FusedResNet of the first replica there)
        self.halves = {}                # id(engine) -> TwoHalves of the same model (large batches: two halves in flight)

    def __deepcopy__(self, memo):       # copy.deepcopy(model): the copy derives its own executor
        return AutoFusion()

    def __reduce__(self):               # pickling / torch.save(model): derived data is not saved
        return (AutoFusion, ())

    def reset(self) -> None:
        with self.lock:
            self.engine, self.failed_sig, self.reason, self.verified = None, None, None, False
            self.seen.clear()
            self.replica_engines.clear()
            self.halves.clear()

    @staticmethod
    def enabled() -> bool:
        return _PER_LAYER == 0 and _NO_MODEL_FUSION == 0 and os.environ.get("BNN_AMD_AUTOFUSE", "1") != "0"

    @staticmethod
    def _hooked(model: nn.Module) -> bool:
        import torch.nn.modules.module as _mm
        if _mm._global_forward_hooks or _mm._global_forward_pre_hooks:
            return True
        return any(m._forward_hooks or m._forward_pre_hooks for m in model.modules() if m is not model)

    def _decline(self):
        self.calls["declined"] += 1
        return None

    def _engine_for(self, model: nn.Module, x: torch.Tensor) -> Optional["FusedResNet"]:
        """The executor for this call (built on first use), or None.  Called with the lock held."""
        if getattr(model, "_is_replica", False):
            # a DataParallel replica (examples/cifar10.py:74-77): its parameters are broadcast copies that are new on
            # every forward, but their VALUES are the master's — one executor per device, derived from the first
            # replica seen there (which it keeps alive), valid until a master parameter changes
            master = self.owner() if self.owner is not None else None
            if master is None or master is model:
                return None
            sig = _param_signature(master)
            if self.failed_sig == sig:
                return None
            ent = self.replica_engines.get(x.device)
            if ent is not None and ent[0] == sig:
                return ent[1]
            try:
                eng = FusedResNet(model)
            except FusionError as exc:
                self.failed_sig, self.reason = sig, str(exc)
                return None
            self.replica_engines[x.device] = (sig, eng)
            self.verified = self.verified or is_native_model(master)
            if not self.verified:
                return None     # a foreign class is verified on the master first (one un-replicated call)
            return eng
        eng = self.engine
        if eng is None:
            sig = _param_signature(model)
            if self.failed_sig == sig:
                return None
            try:
                eng = FusedResNet(model)
            except FusionError as exc:
                self.failed_sig, self.reason = sig, str(exc)
                return None
            self.engine = eng
            self.verified = self.verified or is_native_model(model)
            if self.owner is None:
                self.owner = weakref.ref(model)
        return eng

    def run(self, model: nn.Module, x: torch.Tensor) -> Optional[torch.Tensor]:
        if (model.training or torch.is_grad_enabled() or not isinstance(x, torch.Tensor) or not x.is_cuda
                or x.dtype != torch.float32 or x.dim() != 4 or x.shape[0] == 0
                or not self.enabled() or not native.available()):
            return self._decline()
        try:
            with self.lock:     # lookup / build / verification / graph capture; the steady-state launches run outside
                if self._verifying:         # the model's own forward, run by _verify: not a call to dispatch
                    return self._decline()
                eng = self._engine_for(model, x)
                if eng is None or eng.model.fc.weight.device != x.device or (
                        eng.hooked() if eng.model is model else self._hooked(model)):
                    return self._decline()
                if torch.cuda.is_current_stream_capturing():
                    # the caller is capturing a HIP graph of its own around `net(x)`: no graph replay inside a capture, no
                    # second stream, nothing that synchronises — the executor's eager launches are plain kernels on the
                    # capturing stream (an executor that is not built and checked yet cannot be built here: per layer)
                    if not self.verified or eng._sig is None or not eng._unchanged():
                        return self._decline()
                    self.calls["eager"] += 1
                    return eng._forward_impl(x)
                if not self.verified:
                    return self._verify(eng, model, x)
                key = (tuple(x.shape), torch.cuda.current_stream(x.device).cuda_stream)
                graph = eng.reads_caller_tensor and (key in eng._split or self.seen[(id(eng),) + key] >= self.CAPTURE_AFTER)
                if graph and TwoHalves.wanted(x):
                    eng._check_current()                     # (a parameter change drops the half-batch executors too)
                    two = self.halves.get(id(eng))
                    if two is None or two.engines[0]._sig != eng._sig:
                        two = self.halves[id(eng)] = TwoHalves(eng.model, x.device)
                    self.calls["graph"] += 1
                    if not two.captured(x):
                        return two(x)                        # captures: under the lock
                    graph = two
                elif graph:
                    self.calls["graph"] += 1
                    if key not in eng._split:
                        return eng.forward_fresh(x)          # captures: under the lock
                else:
                    self.seen[(id(eng),) + key] += 1
                    if len(self.seen) > 64:
                        self.seen.clear()
                    self.calls["eager"] += 1
            if isinstance(graph, TwoHalves):
                return graph(x)
            return eng.forward_fresh(x) if graph else eng(x)
        except FusionError as exc:      # e.g. parameters moved to the CPU since the executor was built
            with self.lock:
                self.engine, self.failed_sig, self.reason = None, _param_signature(model), str(exc)
                self.replica_engines.clear()
                self.halves.clear()
            return self._decline()

    def _verify(self, eng: "FusedResNet", model: nn.Module, x: torch.Tensor) -> Optional[torch.Tensor]:
        """First fused call of a model built from another package's classes: check it against the class's own forward."""
        y = eng(x)
        n = min(2, x.shape[0])
        self._verifying = True
        try:
            want = getattr(type(model), "_bnn_base", type(model)).forward(model, x[:n])   # the class's OWN forward
        finally:
            self._verifying = False
        err = float((y[:n] - want).abs().max() / want.abs().max().clamp_min(1e-30))
        if not err <= self.VERIFY_TOL:
            self.engine, self.failed_sig = None, _param_signature(model)
            self.reason = f"fused result differs from the model's own forward (relative {err:.3g})"
            warnings.warn(f"bnn_amd: {type(model).__name__}: {self.reason}; keeping the per-layer path",
                          RuntimeWarning)
            return self._decline()
        self.verified = True
        self.calls["eager"] += 1
        self.seen[(id(eng), tuple(x.shape), torch.cuda.current_stream(x.device).cuda_stream)] += 1
        return y


def _param_signature(model: nn.Module):
    return tuple((id(t), t.data_ptr(), t._version) for t in itertools.chain(model.parameters(), model.buffers()))


def auto_fusion(model: nn.Module) -> AutoFusion:
    """The model's ``AutoFusion`` state (created on first use)."""
    st = model.__dict__.get("_bnn_auto")
    if st is None:
        st = model.__dict__["_bnn_auto"] = AutoFusion(model)
    return st


def auto_forward(model: nn.Module, x: torch.Tensor) -> Optional[torch.Tensor]:
    """Called at the top of ``bnn_amd.models.ResNet.forward``: fused logits, or None -> the caller's own forward."""
    return auto_fusion(model).run(model, x)


_AUTO_CLASSES: dict = {}


def _auto_class(base: type) -> type:
    """``base`` with the dispatch of ``bnn_amd.models.ResNet.forward`` in front of its own ``forward``: a subclass made
    on the fly (one per base class), the way ``torch.nn.utils.parametrize`` injects behaviour into a module instance.
    A class — not an instance attribute — so that ``DataParallel`` replicas (``replicate`` copies ``__dict__``), deep
    copies and pickles of the model each dispatch on THEMSELVES; name, module and repr stay those of ``base``."""
    dyn = _AUTO_CLASSES.get(base)
    if dyn is None:
        def forward(self, x, *args, **kwargs):
            if not args and not kwargs and isinstance(x, torch.Tensor) and x.is_cuda and not self.training \
                    and not torch.is_grad_enabled():
                y = auto_forward(self, x)
                if y is not None:
                    return y
            return base.forward(self, x, *args, **kwargs)

        def __reduce_ex__(self, protocol):      # pickle / deepcopy: rebuilt from the importable base class
            return (_rebuild_auto, (base,), self.__dict__)

        def _replicate_for_data_parallel(self):  # replicas share the master's AutoFusion (one executor per device)
            auto_fusion(self)
            return base._replicate_for_data_parallel(self)

        dyn = type(base.__name__, (base,), {"forward": forward, "__reduce_ex__": __reduce_ex__, "_bnn_base": base,
                                            "_replicate_for_data_parallel": _replicate_for_data_parallel,
                                            "__module__": base.__module__, "__qualname__": base.__qualname__,
                                            "__doc__": base.__doc__})
        _AUTO_CLASSES[base] = dyn
    return dyn


def _rebuild_auto(base: type):
    cls = _auto_class(base)
    return cls.__new__(cls)


def install_auto_fusion(model: nn.Module) -> bool:
    """Give a ResNet of ANOTHER package (laid out like the reference's ``bnn.models.resnet.ResNet``) the same
    dispatch ``bnn_amd.models.ResNet.forward`` has: "fused executor when it applies, else the class's own forward".
    ``prepare_binary_model`` calls this for the model it converted; returns whether the model was recognised (and was
    not dispatching already).  Undo with ``uninstall_auto_fusion``."""
    if isinstance(model, ResNet) or hasattr(type(model), "_bnn_base") or not resnet_shaped(model):
        return False
    model.__class__ = _auto_class(type(model))
    return True


def uninstall_auto_fusion(model: nn.Module) -> None:
    base = getattr(type(model), "_bnn_base", None)
    if base is not None:
        model.__class__ = base
    model.__dict__.pop("_bnn_auto", None)


def optimize_for_inference(model: nn.Module) -> nn.Module:
    """Return the fused executor for ``model`` when it is covered, else ``model`` unchanged."""
    try:
        return FusedResNet(model)
    except FusionError:
        return model
