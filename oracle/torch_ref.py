"""Float restatement of the reference forward out of stock torch CPU ops — TEST INFRASTRUCTURE ONLY.

This is the op sequence the reference itself executes: ``aten::sign`` ->
``mean|W|`` -> ``sign(W)*alpha`` -> ``aten::conv2d`` (oneDNN on CPU), written functionally so it does
not depend on the product package.  Used (a) as an independent checker in tests/, validated there
against the fixtures generated from the reference, and (b) as ``bench.py``'s ``cpu_baseline``
(kind "port": the reference's Python cannot travel to the GPU box).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F


def xnor_weight(w: torch.Tensor, compute_alpha: bool = True, center: bool = False) -> torch.Tensor:
    """bnn/ops.py:129-140 (XNORWeightBinarizer.forward) incl. :116-127 (_compute_alpha)."""
    if center:
        w = w - w.mean(1, keepdim=True)
    s = torch.sign(w)
    if compute_alpha:
        n = w[0].nelement()
        alpha = w.abs().flatten(1).sum(1).div(n).view(-1, *([1] * (w.dim() - 1)))
        s = s * alpha
    return s


def binary_conv2d(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, stride=1,
                  padding=0, dilation=1, post_scale: Optional[torch.Tensor] = None,
                  compute_alpha: bool = True, center: bool = False) -> torch.Tensor:
    """bnn/layers/conv.py:90-97 with BasicInputBinarizer / XNORWeightBinarizer / Identity|BasicScale."""
    out = F.conv2d(torch.sign(x), xnor_weight(w, compute_alpha, center), bias, stride, padding, dilation)
    if post_scale is not None:
        out = out * post_scale.view(1, -1, 1, 1)
    return out


def binary_linear(x, w, bias=None, post_scale=None, compute_alpha=True, center=False):
    """bnn/layers/linear.py:22-27."""
    out = F.linear(torch.sign(x), xnor_weight(w, compute_alpha, center), bias)
    if post_scale is not None:
        out = out * post_scale.view(1, -1)
    return out


def _bn(x, sd: Dict[str, torch.Tensor], p: str):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.0, 1e-5)


def resnet18_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """Eval forward of the examples/cifar10.py:61-71 model (binary ResNet-18, conv1 + fc float).

    Graph: bnn/models/resnet.py:147-164 (_forward_impl), BasicBlock bnn/models/layers/res_block.py:40-56,
    shortcut AvgPool2d(ceil, no pad count) -> 1x1 (binary) -> BN, resnet.py:128-133.
    """
    x = F.conv2d(x, sd["conv1.weight"], None, 2, 3)                      # real-valued stem
    x = F.max_pool2d(F.relu(_bn(x, sd, "bn1")), 3, 2, 1)
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for bi in (0, 1):
            p = f"layer{li}.{bi}"
            s = stride if bi == 0 else 1
            y = binary_conv2d(x, sd[p + ".conv1.weight"], None, s, 1)
            y = F.relu(_bn(y, sd, p + ".bn1"))
            y = binary_conv2d(y, sd[p + ".conv2.weight"], None, 1, 1)
            y = _bn(y, sd, p + ".bn2")
            if (p + ".downsample.1.weight") in sd:
                idn = F.avg_pool2d(x, s, s, 0, ceil_mode=True, count_include_pad=False)
                idn = binary_conv2d(idn, sd[p + ".downsample.1.weight"], None, 1, 0)
                idn = _bn(idn, sd, p + ".downsample.2")
            else:
                idn = x
            x = F.relu(y + idn)
    x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
    return F.linear(x, sd["fc.weight"], sd["fc.bias"])


RESNET18_SHAPES = None


def resnet18_state_shapes() -> Dict[str, tuple]:
    """state_dict() shapes of the cifar10.py model, spelled out so no model class is needed."""
    shapes: Dict[str, tuple] = {"conv1.weight": (64, 3, 7, 7)}

    def bn(p, c):
        shapes.update({p + ".weight": (c,), p + ".bias": (c,), p + ".running_mean": (c,),
                       p + ".running_var": (c,), p + ".num_batches_tracked": ()})
    bn("bn1", 64)
    cin = 64
    for li, c in ((1, 64), (2, 128), (3, 256), (4, 512)):
        for bi in (0, 1):
            p = f"layer{li}.{bi}"
            shapes[p + ".conv1.weight"] = (c, cin if bi == 0 else c, 3, 3)
            bn(p + ".bn1", c)
            shapes[p + ".conv2.weight"] = (c, c, 3, 3)
            bn(p + ".bn2", c)
            if bi == 0 and li > 1:
                shapes[p + ".downsample.1.weight"] = (c, cin, 1, 1)
                bn(p + ".downsample.2", c)
        cin = c
    shapes["fc.weight"] = (1000, 512)
    shapes["fc.bias"] = (1000,)
    return shapes
