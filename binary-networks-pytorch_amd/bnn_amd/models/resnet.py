"""ResNet graphs that feed the binary-conv hot path.

Same module names / forward order as the reference's ``bnn/models/resnet.py:50-167`` so that
``state_dict`` keys line up: 7x7-s2 stem (real-valued) -> maxpool -> 4 stages -> avgpool -> fc
(real-valued).  Stage transitions down-sample the shortcut with
``AvgPool2d(stride, ceil_mode=True, count_include_pad=False) -> conv1x1 -> BN``
(``resnet.py:128-133``).

Deliberate fix: the classifier width is ``planes * block.expansion``; the reference uses ``planes``
(``resnet.py:143,101``), which makes every Bottleneck ResNet crash at ``fc``.
"""
from __future__ import annotations

from typing import Any, Callable, List, Optional, Type

import torch
import torch.nn as nn

from .blocks import BasicBlock, Bottleneck, HBlock, PreBasicBlock, PreBottleneck, conv1x1


def _auto_forward(model, x):
    from ..inference import auto_forward      # (inference imports this module)
    return auto_forward(model, x)


class DaBNNStem(nn.Module):
    """Cheaper stem of daBNN (``resnet.py:10-47``): 3x3-s2 -> (1x1 -> 3x3-s2 | maxpool) -> 1x1."""

    def __init__(self, planes: int, norm_layer: Optional[Callable[..., nn.Module]] = None,
                 activation=nn.ReLU) -> None:
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d

        def unit(cin, cout, k, s):
            return nn.Sequential(nn.Conv2d(cin, cout, k, s, padding=k // 2, bias=False),
                                 norm_layer(cout), activation())

        self.conv1 = unit(3, planes // 2, 3, 2)
        self.conv2_1 = unit(planes // 2, planes // 4, 1, 1)
        self.conv2_2 = unit(planes // 4, planes // 2, 3, 2)
        self.conv3 = unit(planes, planes, 1, 1)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.conv1(x)
        x = torch.cat([self.conv2_2(self.conv2_1(x)), self.maxpool(x)], dim=1)
        return self.conv3(x)


class ResNet(nn.Module):
    def __init__(self, block: Type[nn.Module], layers: List[int], num_classes: int = 1000,
                 zero_init_residual: bool = False, groups: int = 1, width_per_group: int = 64,
                 replace_stride_with_dilation: Optional[List[bool]] = None,
                 norm_layer: Optional[Callable[..., nn.Module]] = None,
                 activation: Optional[Callable[..., nn.Module]] = None,
                 stem_type: str = "basic") -> None:
        super().__init__()
        self._norm_layer = norm_layer or nn.BatchNorm2d
        self._activation = activation or nn.ReLU
        self.stem_type = stem_type
        self.inplanes = 64
        self.dilation = 1
        self.groups = groups
        self.base_width = width_per_group
        dilate = replace_stride_with_dilation or [False, False, False]
        if len(dilate) != 3:
            raise ValueError("replace_stride_with_dilation should be None "
                             "or a 3-element tuple, got {}".format(dilate))

        if stem_type == "basic":
            self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
            self.bn1 = self._norm_layer(self.inplanes)
        elif stem_type == "dabnn":
            self.conv1 = DaBNNStem(self.inplanes, norm_layer=self._norm_layer)
        else:
            raise ValueError(f"unknown stem_type {stem_type!r}")
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dilate=dilate[0])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dilate=dilate[1])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dilate=dilate[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(self.outplanes, num_classes)

        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes: int, blocks: int, stride: int = 1,
                    dilate: bool = False) -> nn.Sequential:
        prev_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        out_ch = planes * block.expansion
        shortcut = None
        if stride != 1 or self.inplanes != out_ch:
            shortcut = nn.Sequential(
                nn.AvgPool2d(kernel_size=stride, stride=stride, ceil_mode=True,
                             count_include_pad=False),
                conv1x1(self.inplanes, out_ch, stride=1),
                self._norm_layer(out_ch))
        # HBlock cannot stride (hierarchical_block.py:23-24): pool in front of the stage instead.
        # Its shortcut is pre-activation style like the block itself (BN -> conv): a binary conv
        # must not binarise a raw residual sum, whose exactly-cancelling entries are 0 in exact
        # arithmetic but +-1e-9 in any float convolution ("exact zeros").
        pre: List[nn.Module] = []
        block_stride = stride
        if block is HBlock and stride != 1:
            pre.append(nn.AvgPool2d(kernel_size=stride, stride=stride, ceil_mode=True,
                                    count_include_pad=False))
            block_stride = 1
        if block is HBlock and shortcut is not None:
            shortcut = nn.Sequential(self._norm_layer(self.inplanes),
                                     conv1x1(self.inplanes, out_ch, stride=1))
        stage = pre + [block(self.inplanes, planes, block_stride, shortcut, self.groups,
                             self.base_width, prev_dilation, self._norm_layer,
                             activation=self._activation)]
        self.inplanes = out_ch
        for _ in range(1, blocks):
            stage.append(block(self.inplanes, planes, groups=self.groups,
                               base_width=self.base_width, dilation=self.dilation,
                               norm_layer=self._norm_layer, activation=self._activation))
        self.outplanes = out_ch
        return nn.Sequential(*stage)

    def train(self, mode: bool = True):
        # train() <-> eval(): the fused executor's packed weights and folded BatchNorm constants are derived data of
        # the parameters; drop them with the mode (layers/_base.py: BinaryLayerMixin.train — `.data` writes made while
        # training are then seen by the first evaluation forward)
        if bool(mode) != self.training:
            st = self.__dict__.get("_bnn_auto")
            if st is not None:
                st.reset()
        return super().train(mode)

    def _replicate_for_data_parallel(self):
        # nn.DataParallel replicas share the master's AutoFusion state (`replicate` copies __dict__): it must exist
        # before the copy is made, or every replica of every forward would start from scratch
        from ..inference import auto_fusion
        auto_fusion(self)
        return super()._replicate_for_data_parallel()

    def _stem(self, x: torch.Tensor, inference: bool) -> torch.Tensor:
        """conv1 -> bn1 -> relu -> maxpool (the daBNN stem is all in conv1)."""
        if inference:                                # the per-layer / per-block tiers: the stem as its MFMA kernel
            from ..inference import eval_stem
            y = eval_stem(self, x)
            if y is not None:
                return y
        if self.stem_type == "basic" and x.is_cuda and self.training:
            # training on a HIP device: the 7x7 convolution on the matrix cores, bn1 -> relu -> maxpool as one fused op
            from .. import training
            return training.stem_tail(training.stem_conv(x, self.conv1), self.bn1, self.relu, self.maxpool)
        x = self.conv1(x)
        if self.stem_type != "basic":
            return x
        if x.is_cuda and self.bn1.training:          # (bn1 alone in training mode)
            from .. import training
            return training.stem_tail(x, self.bn1, self.relu, self.maxpool)
        return self.maxpool(self.relu(self.bn1(x)))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # eval + no_grad on a HIP device: the fused executor (bnn_amd/dispatch.py: AutoFusion) — what makes the
        # reference's own call `outputs = net(inputs)` (examples/cifar10.py:140-149) the fast path.  None -> per layer.
        inference = not self.training and x.is_cuda and not torch.is_grad_enabled()
        if inference:
            y = _auto_forward(self, x)
            if y is not None:
                return y
        x = self._stem(x, inference)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        if inference:                                # avgpool + fc as the head kernel
            from ..inference import eval_head
            y = eval_head(self, x)
            if y is not None:
                return y
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)


def resnet18(block_type: Optional[Type[nn.Module]] = None, **kwargs: Any) -> ResNet:
    return ResNet(block_type or BasicBlock, [2, 2, 2, 2], **kwargs)


def resnet34(block_type: Optional[Type[nn.Module]] = None, **kwargs: Any) -> ResNet:
    return ResNet(block_type or BasicBlock, [3, 4, 6, 3], **kwargs)


def resnet50(block_type: Optional[Type[nn.Module]] = None, **kwargs: Any) -> ResNet:
    return ResNet(block_type or Bottleneck, [3, 4, 6, 3], **kwargs)
