"""Per-layer binarisation recipe object.

Mirrors the public surface of the reference's ``bnn/bconfig.py:6-25``: ``BConfig`` carries three
*factories* (classes or ``with_args`` partials, never module instances) and ``Identity`` is the
two-argument no-op used as default post-processing hook.
"""
from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Any, Callable

import torch
import torch.nn as nn


class Identity(nn.Identity):
    """``post(layer_out, layer_in) -> layer_out`` (reference: ``bnn/bconfig.py:6-8``)."""

    def forward(self, layer_out: torch.Tensor, layer_in: torch.Tensor = None) -> torch.Tensor:  # type: ignore[override]
        return layer_out


@dataclass
class BConfig:
    """Three hook factories of a binary layer.

    * ``activation_pre_process()``      -> module applied to the layer input
    * ``activation_post_process(layer)`` -> module applied to ``(layer_out, layer_in)``
    * ``weight_pre_process()``          -> module applied to the weight

    Passing an ``nn.Module`` *instance* raises ``ValueError`` exactly like the reference
    (``bnn/bconfig.py:17-25``): every converted layer must own fresh hook modules.
    """

    activation_pre_process: Callable[..., nn.Module] = nn.Identity
    activation_post_process: Callable[..., nn.Module] = Identity
    weight_pre_process: Callable[..., nn.Module] = nn.Identity

    def __post_init__(self) -> None:
        for f in fields(self):
            if isinstance(getattr(self, f.name), nn.Module):
                raise ValueError("BConfig received an instance, please pass the class instead.")
