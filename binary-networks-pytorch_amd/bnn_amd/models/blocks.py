"""Residual blocks that CALL the binary-conv hot path.

These are plain float ``nn.Module`` graphs; their ``nn.Conv2d`` leaves become binary layers only
after ``prepare_binary_model``.  Attribute names and forward order equal the reference's
(``bnn/models/layers/res_block.py``, ``hierarchical_block.py``), so ``state_dict`` keys are
interchangeable and outputs can be pinned against fixtures generated from the reference.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn as nn


def conv3x3(cin: int, cout: int, stride: int = 1, groups: int = 1, dilation: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=dilation, groups=groups, bias=False,
                     dilation=dilation)


def conv1x1(cin: int, cout: int, stride: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, 1, stride=stride, bias=False)


def _act(activation, channels: int) -> nn.Module:
    """``nn.ReLU(inplace=True)`` or a per-channel parametric activation (``nn.PReLU``)."""
    return activation(inplace=True) if activation == nn.ReLU else activation(num_parameters=channels)


def _fused(block, x):
    from ..inference import auto_block_forward      # (inference imports this module)
    return auto_block_forward(block, x)


def _bn_act(x, bn, act=None, residual=None):
    """``act(bn(x) (+ residual))``: on a HIP device in training mode one fused op (bnn_amd/training.py: bn_act — batch
    statistics, normalisation, residual add and ReLU in three launches instead of four library passes), in eval mode
    without autograd one launch (bnn_amd/tails.py: eval_tail), else the modules themselves in the reference's
    order."""
    if x.is_cuda and bn.training:
        from .. import training
        return training.bn_act(x, bn, act, residual)
    if x.is_cuda and not torch.is_grad_enabled():
        from ..inference import eval_tail           # (inference imports this module)
        y = eval_tail(x, bn, act, residual)
        if y is not None:
            return y
    y = bn(x)
    if residual is not None:
        y += residual
    return y if act is None else act(y)


class _Residual(nn.Module):
    """Base of the residual blocks.  ``forward`` is the reference's op sequence (``_forward``), except that a block
    evaluated for inference on a HIP device first offers itself to the fused block executor
    (``bnn_amd/dispatch.py: BlockFusion``) — the blocks of a ``ResNet`` are normally fused as part of the whole model
    (``AutoFusion``) and never get here."""
    expansion = 1

    def _shortcut(self, x: torch.Tensor) -> torch.Tensor:
        ds = self.downsample
        if ds is None:
            return x
        if (isinstance(ds, nn.Sequential) and len(ds) == 3 and isinstance(ds[2], nn.BatchNorm2d)
                and not ds._forward_hooks and not ds._forward_pre_hooks):
            if x.is_cuda and self.training and torch.is_grad_enabled():
                from .. import training
                pooled = training.shortcut_pool(x, ds[0])   # (the same forward; a streaming kernel for its backward)
            else:
                pooled = ds[0](x)
            return _bn_act(ds[1](pooled), ds[2])         # AvgPool -> conv1x1 -> BN (bnn/models/resnet.py:128-133)
        return ds(x)

    def train(self, mode: bool = True):
        # train() <-> eval(): what was derived from the block's parameters and buffers goes with the mode — the block
        # tier's executor (packed weights, folded BatchNorm constants, thresholds) and the per-layer tails' folded
        # BatchNorms: `.data` writes made while training (clamps, EMA swaps) reach the first evaluation forward
        # (layers/_base.py: BinaryLayerMixin.train gives the same guarantee for the packed weights)
        if bool(mode) != self.training:
            self.__dict__.pop("_bnn_auto_block", None)
            from ..tails import drop_derived
            drop_derived(self)
        return super().train(mode)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.training and x.is_cuda and not torch.is_grad_enabled():
            y = _fused(self, x)
            if y is not None:
                return y
        return self._forward(x)


class BasicBlock(_Residual):
    """conv-BN-act-conv-BN-(+id)-act   (reference: ``res_block.py:8-56``)."""
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1,
                 downsample: Optional[nn.Module] = None, groups: int = 1, base_width: int = 64,
                 dilation: int = 1, norm_layer: Optional[Callable[..., nn.Module]] = None,
                 activation=nn.ReLU) -> None:
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.act1 = _act(activation, planes)
        self.act2 = _act(activation, planes)
        self.downsample = downsample
        self.stride = stride

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        y = _bn_act(self.conv1(x), self.bn1, self.act1)
        return _bn_act(self.conv2(y), self.bn2, self.act2, residual=self._shortcut(x))


class PreBasicBlock(_Residual):
    """BN-conv-act-BN-conv-act-(+id)   (reference: ``res_block.py:121-167``)."""
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1,
                 downsample: Optional[nn.Module] = None, groups: int = 1, base_width: int = 64,
                 dilation: int = 1, norm_layer: Optional[Callable[..., nn.Module]] = None,
                 activation=nn.ReLU) -> None:
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(inplanes)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.act1 = _act(activation, planes)
        self.act2 = _act(activation, planes)
        self.downsample = downsample
        self.stride = stride

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.act1(self.conv1(_bn_act(x, self.bn1)))
        y = self.act2(self.conv2(_bn_act(y, self.bn2)))
        y += self._shortcut(x)
        return y


class Bottleneck(_Residual):
    """1x1-BN-act-3x3(stride)-BN-act-1x1-BN-(+id)-act   (reference: ``res_block.py:59-118``)."""
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1,
                 downsample: Optional[nn.Module] = None, groups: int = 1, base_width: int = 64,
                 dilation: int = 1, norm_layer: Optional[Callable[..., nn.Module]] = None,
                 activation=nn.ReLU) -> None:
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = norm_layer(width)
        self.conv2 = conv3x3(width, width, stride, groups, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.act1 = _act(activation, width)
        self.act2 = _act(activation, width)
        self.act3 = _act(activation, planes * self.expansion)
        self.downsample = downsample
        self.stride = stride

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        y = _bn_act(self.conv1(x), self.bn1, self.act1)
        y = _bn_act(self.conv2(y), self.bn2, self.act2)
        return _bn_act(self.conv3(y), self.bn3, self.act3, residual=self._shortcut(x))


class PreBottleneck(_Residual):
    """BN-1x1-act-BN-3x3-act-BN-1x1-act-(+id)   (reference: ``res_block.py:170-229``)."""
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1,
                 downsample: Optional[nn.Module] = None, groups: int = 1, base_width: int = 64,
                 dilation: int = 1, norm_layer: Optional[Callable[..., nn.Module]] = None,
                 activation=nn.ReLU) -> None:
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = norm_layer(inplanes)
        self.conv2 = conv3x3(width, width, stride, groups, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, planes * self.expansion)
        self.bn3 = norm_layer(width)
        self.act1 = _act(activation, width)
        self.act2 = _act(activation, width)
        self.act3 = _act(activation, planes * self.expansion)
        self.downsample = downsample
        self.stride = stride

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.act1(self.conv1(self.bn1(x)))
        y = self.act2(self.conv2(self.bn2(y)))
        y = self.act3(self.conv3(self.bn3(y)))
        y += self._shortcut(x)
        return y


class HBlock(_Residual):
    """Hierarchical block: three BN-act-3x3 stages of widths C/2, C/4, C/4, concatenated, + id.

    Reference: ``bnn/models/layers/hierarchical_block.py:8-60`` (stride/dilation > 1 rejected
    there too).  ``expansion = 1`` is added so the block can be placed in :class:`ResNet`.
    """
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1,
                 downsample: Optional[nn.Module] = None, groups: int = 1, base_width: int = 64,
                 dilation: int = 1, norm_layer: Optional[Callable[..., nn.Module]] = None,
                 activation=nn.ReLU) -> None:
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in HBlock")
        if stride > 1:
            raise NotImplementedError("Stride > 1 not supported in HBlock")
        half, quarter = int(planes / 2), int(planes / 4)
        self.bn1 = norm_layer(inplanes)
        self.conv1 = conv3x3(inplanes, half, groups=groups)
        self.bn2 = norm_layer(half)
        self.conv2 = conv3x3(half, quarter, groups=groups)
        self.bn3 = norm_layer(quarter)
        self.conv3 = conv3x3(quarter, quarter, groups=groups)
        self.act1 = _act(activation, half)
        self.act2 = _act(activation, half)
        self.act3 = _act(activation, quarter)
        self.downsample = downsample

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        o1 = self.conv1(_bn_act(x, self.bn1, self.act1))
        o2 = self.conv2(_bn_act(o1, self.bn2, self.act2))
        o3 = self.conv3(_bn_act(o2, self.bn3, self.act3))
        y = torch.cat((o1, o2, o3), 1)
        y += self._shortcut(x)
        return y
