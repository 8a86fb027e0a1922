"""Binary ``Linear`` (API of the reference's ``bnn/layers/linear.py:9-44``)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import fastpath
from ..bconfig import BConfig
from ._base import BinaryLayerMixin


class Linear(BinaryLayerMixin, nn.Linear):
    _FLOAT_MODULE = nn.Linear

    def __init__(self, in_features: int, out_features: int, bias: bool = True,
                 bconfig: Optional[BConfig] = None) -> None:
        nn.Linear.__init__(self, in_features, out_features, bias)
        self._init_hooks(bconfig)

    @classmethod
    def _ctor_kwargs(cls, mod: nn.Module) -> dict:
        return dict(in_features=mod.in_features, out_features=mod.out_features,
                    bias=mod.bias is not None)

    def forward(self, input: torch.Tensor) -> torch.Tensor:  # noqa: A002
        plan = fastpath.plan_linear(self, input)
        if plan is not None:
            return fastpath.linear(self, input, plan)
        x = self.activation_pre_process(input)
        out = F.linear(x, self.weight_pre_process(self.weight), self.bias)
        return self.activation_post_process(out, input)
