"""Binary ``Conv2d`` / ``Conv1d`` (API of the reference's ``bnn/layers/conv.py``).

``Conv2d.forward`` has two implementations of the same function:

* the **HIP path** (``fastpath.conv2d``): bit-packed activations/weights, XNOR/popcount kernels on
  gfx950 — taken for CUDA(HIP) float32 tensors when autograd is not recording and the hooks are
  the recognised XNOR-Net combination.  If it applies and the native library is missing, the call
  raises; it never silently degrades.
* the **composition path**: ``post(conv(pre(x), wpre(W), b), x)`` out of torch ops — training,
  CPU tensors, or hook combinations outside the accelerated path.  This is the reference's own
  formulation (``bnn/layers/conv.py:90-97``).
"""
from __future__ import annotations

from typing import Optional, Union

import torch
import torch.nn as nn
from torch.nn.common_types import _size_1_t, _size_2_t

from .. import fastpath
from ..bconfig import BConfig
from ._base import BinaryLayerMixin


def _conv_kwargs(mod: nn.Module) -> dict:
    return dict(in_channels=mod.in_channels, out_channels=mod.out_channels,
                kernel_size=mod.kernel_size, stride=mod.stride, padding=mod.padding,
                dilation=mod.dilation, groups=mod.groups, bias=mod.bias is not None,
                padding_mode=mod.padding_mode)


class Conv2d(BinaryLayerMixin, nn.Conv2d):
    _FLOAT_MODULE = nn.Conv2d

    def __init__(self, in_channels: int, out_channels: int, kernel_size: _size_2_t,
                 stride: _size_2_t = 1, padding: Union[str, _size_2_t] = 0,
                 dilation: _size_2_t = 1, groups: int = 1, bias: bool = True,
                 padding_mode: str = "zeros", bconfig: Optional[BConfig] = None) -> None:
        nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size, stride=stride,
                           padding=padding, dilation=dilation, groups=groups, bias=bias,
                           padding_mode=padding_mode)
        self._init_hooks(bconfig)

    _ctor_kwargs = classmethod(lambda cls, mod: _conv_kwargs(mod))

    def forward(self, input: torch.Tensor) -> torch.Tensor:  # noqa: A002
        plan = fastpath.plan_conv2d(self, input)
        if plan is not None:
            return fastpath.conv2d(self, input, plan)
        plan = fastpath.plan_conv2d_train(self, input)
        if plan is not None:
            return fastpath.conv2d_train(self, input, plan)
        x = self.activation_pre_process(input)
        out = self._conv_forward(x, self.weight_pre_process(self.weight), self.bias)
        return self.activation_post_process(out, input)


class Conv1d(BinaryLayerMixin, nn.Conv1d):
    """1-D variant; evaluated on the HIP path as an ``H == 1`` convolution."""

    _FLOAT_MODULE = nn.Conv1d

    def __init__(self, in_channels: int, out_channels: int, kernel_size: _size_1_t,
                 stride: _size_1_t = 1, padding: Union[str, _size_1_t] = 0,
                 dilation: _size_1_t = 1, groups: int = 1, bias: bool = True,
                 padding_mode: str = "zeros", bconfig: Optional[BConfig] = None) -> None:
        nn.Conv1d.__init__(self, in_channels, out_channels, kernel_size, stride=stride,
                           padding=padding, dilation=dilation, groups=groups, bias=bias,
                           padding_mode=padding_mode)
        self._init_hooks(bconfig)

    _ctor_kwargs = classmethod(lambda cls, mod: _conv_kwargs(mod))

    def forward(self, input: torch.Tensor) -> torch.Tensor:  # noqa: A002
        plan = fastpath.plan_conv1d(self, input)
        if plan is not None:
            return fastpath.conv1d(self, input, plan)
        x = self.activation_pre_process(input)
        out = self._conv_forward(x, self.weight_pre_process(self.weight), self.bias)
        return self.activation_post_process(out, input)
