"""The per-layer path's one-launch tails (``eval_tail`` / ``eval_stem`` / ``eval_head``) and the switches of the
per-layer tier (``per_layer_forward``, ``library_tails``)."""
from __future__ import annotations

import contextlib
import os
import weakref
from typing import Optional

import torch
import torch.nn as nn

from . import fastpath, hipops
from .executor import _is_float_layer, _is_float_layer_linear, fold_bn


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


def drop_derived(module: nn.Module) -> None:
    """Forget the folded BatchNorm constants and transposed head weights of every module under ``module`` (mode switches,
    ``fastpath.invalidate``): they are keyed on version counters, which writes through ``.data`` do not move."""
    for m in module.modules():
        _FOLDS.pop(m, None)
        _HEAD_WEIGHTS.pop(m, None)


def cached_fold(bn: nn.BatchNorm2d):
    """``fold_bn(bn)`` on ``bn``'s device, kept until one of the module's four tensors is written or replaced."""
    ts = (bn.running_mean, bn.running_var, bn.weight, bn.bias)
    key = tuple((id(t), t._version, t.data_ptr()) if t is not None else None for t in ts) + (bn.eps,)
    c = _FOLDS.get(bn)
    if c is None or c[0] != key or fastpath.strict_weights():
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
    if c is None or c[0] != key or fastpath.strict_weights():
        c = _HEAD_WEIGHTS[fc] = (key, w.detach().t().contiguous())
    return hipops.avgpool_fc(x, c[1], None if fc.bias is None else fc.bias.detach())


def _pair2(v):
    return (v, v) if isinstance(v, int) else tuple(v)
