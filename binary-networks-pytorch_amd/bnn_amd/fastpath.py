"""Dispatch of binary layers onto the HIP XNOR/popcount path.

A layer takes the HIP path when ALL of the following hold (otherwise the torch composition in
``layers/*.py`` — the reference's own formulation — runs):

* hooks are exactly ``BasicInputBinarizer`` (or, for inference, ``AdvancedInputBinarizer`` with its default
  ``tanh``, whose value is also ``sign(x)``) / ``XNORWeightBinarizer`` / (``Identity`` or a
  per-output-channel ``BasicScaleBinarizer``)      (reference: ``examples/cifar10.py:65-69``,
  ``test/test_layers.py:17-21``)
* input and weight are both float32 or both float16 (a ``.half()`` model: bit planes straight from the fp16
  tensor, fp32 arithmetic inside, output rounded to fp16 once) on the same HIP device
* autograd is not recording (``torch.no_grad()`` / inference); when it IS recording, ``Conv2d`` takes the
  training variant (HIP forward, fp32 library backward with the straight-through estimator:
  ``bnn_amd/training.py``), ``Conv1d``/``Linear`` fall back to the composition
* ``groups == 1``, ``padding_mode == 'zeros'``, numeric padding

When those hold and ``libbnn_hip.so`` cannot be loaded the call raises ``NativeError``: there is
no CPU or eager stand-in for the GPU path.

Packed weights (1 bit + 1 mask bit per weight, alpha per channel) are derived data: they are
cached on the layer, keyed on the weight's storage pointer and version counter, and rebuilt after
``load_state_dict`` / ``.to()`` / optimiser steps.  The fp32 ``weight`` Parameter stays the source
of truth (reference: ``bnn/layers/conv.py:111-112`` shares it with the float module).
"""
from __future__ import annotations

import os
import threading
import warnings
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import hipops, native

_stats_lock = threading.Lock()
_stats = {"conv2d": 0, "conv1d": 0, "linear": 0, "weight_packs": 0, "conv2d_train": 0}


def stats() -> dict:
    """Counters of HIP-path invocations (tests use them to prove which path ran)."""
    with _stats_lock:
        return dict(_stats)


def _bump(key: str) -> None:
    with _stats_lock:
        _stats[key] += 1


def strict_weights() -> bool:
    """``BNN_AMD_STRICT_WEIGHTS=1``: derived weight data is never cached — every forward of every binary layer re-derives
    sign bits, zero mask and alpha from the CURRENT values of ``weight`` and reads the zero-weight flag before it
    launches, exactly as the reference re-binarises on every forward (bnn/layers/conv.py:92, bnn/ops.py:129-140).
    Writes through ``.data`` need no ``invalidate()`` then, and a training step never runs on an unverified "no exact
    zero" assumption.  The price is one packing launch and one host round trip per layer and call, and ``model(x)``
    stays on the per-layer tier (the fused executors and their HIP graphs hold derived data by construction):
    README.md has the measured cost.  Read at every call, so tests can flip it."""
    return os.environ.get("BNN_AMD_STRICT_WEIGHTS", "0") == "1"


@dataclass
class Plan:
    center: bool
    compute_alpha: bool
    scale: Optional[torch.Tensor]  # BasicScaleBinarizer.alpha or None
    ste: bool = True               # the input hook's backward is the hard-tanh STE (training fast path allowed)


_HOOK_MODULES = ("bnn_amd.ops", "bnn_amd.bconfig", "bnn.ops", "bnn.bconfig")


def _is(obj, name: str) -> bool:
    """Exact hook class, by name, from this package or from the reference package (so that a
    model converted with the reference's own `bnn.ops` classes is accelerated as well)."""
    t = type(obj)
    return t.__name__ == name and t.__module__ in _HOOK_MODULES


def _recognise(layer: nn.Module, out_channels: int) -> Optional[Plan]:
    pre = layer.activation_pre_process
    wpre = layer.weight_pre_process
    post = layer.activation_post_process
    basic = _is(pre, "BasicInputBinarizer")
    # AdvancedInputBinarizer (bnn/ops.py:167-177) returns sign(f(t*x)); with the default f = tanh and t > 0
    # that IS sign(x), only its gradient differs — so inference may take the same kernels, training may not
    advanced = (_is(pre, "AdvancedInputBinarizer") and getattr(pre, "derivative_funct", None) is torch.tanh
                and isinstance(getattr(pre, "t", None), (int, float)) and pre.t > 0)
    if not (basic or advanced) or not _is(wpre, "XNORWeightBinarizer"):
        return None
    if _is(post, "Identity"):
        scale = None
    elif _is(post, "BasicScaleBinarizer") and post.alpha.numel() == out_channels \
            and post.alpha.dim() >= 2 and post.alpha.shape[1] == out_channels:
        scale = post.alpha
    else:
        return None
    return Plan(bool(wpre.center_weights), bool(wpre.compute_alpha), scale, ste=basic)


def _eligible(layer: nn.Module, x: torch.Tensor, plan: Plan) -> bool:
    w = layer.weight
    if not (x.is_cuda and w.is_cuda and x.device == w.device):
        return False
    if x.dtype != w.dtype or x.dtype not in (torch.float32, torch.float16):
        return False
    if torch.is_grad_enabled():
        needs = x.requires_grad or w.requires_grad
        needs = needs or (layer.bias is not None and layer.bias.requires_grad)
        needs = needs or (plan.scale is not None and plan.scale.requires_grad)
        if needs:
            return False
    return True


def _eligible_train(layer: nn.Module, x: torch.Tensor) -> bool:
    """Autograd is recording and something on the path needs a gradient: HIP forward + library
    backward (``bnn_amd/training.py``) instead of the pure torch composition."""
    from . import training
    w = layer.weight
    return (training.ENABLED and torch.is_grad_enabled() and x.is_cuda and w.is_cuda and x.device == w.device
            and x.dtype == torch.float32 and w.dtype == torch.float32)


def _numeric_padding(layer) -> bool:
    return not isinstance(layer.padding, str) and layer.padding_mode == "zeros" and layer.groups == 1


def plan_conv2d(layer, x: torch.Tensor) -> Optional[Plan]:
    if x.dim() != 4 or not _numeric_padding(layer):
        return None
    plan = _recognise(layer, layer.out_channels)
    return plan if plan is not None and _eligible(layer, x, plan) else None


def plan_conv2d_train(layer, x: torch.Tensor) -> Optional[Plan]:
    """Plan for a training-mode forward (only consulted when ``plan_conv2d`` declined)."""
    if x.dim() != 4 or not _numeric_padding(layer):
        return None
    plan = _recognise(layer, layer.out_channels)
    return plan if plan is not None and plan.ste and _eligible_train(layer, x) else None


def plan_conv1d(layer, x: torch.Tensor) -> Optional[Plan]:
    if x.dim() != 3 or not _numeric_padding(layer):
        return None
    plan = _recognise(layer, layer.out_channels)
    return plan if plan is not None and _eligible(layer, x, plan) else None


def plan_linear(layer, x: torch.Tensor) -> Optional[Plan]:
    if x.dim() < 1:
        return None
    plan = _recognise(layer, layer.out_features)
    return plan if plan is not None and _eligible(layer, x, plan) else None


def packed_weight(layer, plan: Plan, sync: bool = True, fresh: bool = False) -> hipops.PackedWeight:
    """Cached ``XNORWeightBinarizer`` output for ``layer.weight`` (rebuilt when it changes).

    The cache key is the weight's storage pointer + version counter (+ recipe).  Autograd's version counter
    does NOT see writes through ``weight.data`` (``p.data.clamp_()``, ``p.data.copy_(ema)``): after such a write
    call ``invalidate(model)`` (inference) — the training forward never trusts the cache (``fresh=True``: it
    re-derives sign bits and alpha from the current values on every call, like the reference does,
    bnn/ops.py:129-140).

    ``sync=False`` (training step: the weight changed, and will change again) skips the host round trip that
    reads the zero-weight flag: the pack is made under the assumption "no weight is exactly 0" and the flag
    travels to pinned host memory asynchronously.  It is resolved by the NEXT call for this layer — a training
    call checks the previous step's flag (long since on the host), an inference call (``sync=True``) waits for
    it — and a layer that ever showed a zero is packed synchronously (mask-aware kernel) from then on.  The FIRST
    pack of a layer is always synchronous: zeros that are there from the start (pruned or zero-initialised weights, a
    loaded sparse checkpoint) take the zero-aware kernels from the first step on; what the optimistic path can still
    meet is a weight that an update lands on exactly 0.0 — one forward treats it as -1, the next call warns."""
    if strict_weights():
        sync, fresh = True, True
    master = layer.__dict__.get("_bnn_master")
    if master is not None:          # a DataParallel replica: cached on the layer it was replicated from
        return _replica_packed_weight(layer, master, plan, fresh)
    w = layer.weight
    key = (w.data_ptr(), w._version, str(w.device), tuple(w.shape), plan.center, plan.compute_alpha)
    cached = layer.__dict__.get("_bnn_packed")
    sync = sync or cached is None
    if cached is not None and cached[1].zero_probe is not None and (sync or fresh or cached[0] != key):
        # a pack made without reading its zero flag: resolve it before it is trusted / replaced
        pw = cached[1]
        if pw.zero_found_later():
            layer.__dict__["_bnn_zero_seen"] = True
            warnings.warn("bnn_amd: a binary weight is exactly 0 (sign(0) == 0); a forward that ran before this "
                          "was known treated it as -1.  This layer now takes the zero-aware kernels.",
                          RuntimeWarning)
            cached = None      # the unmasked pack is wrong for this weight: rebuild below
        else:
            pw.zero_probe = None
    if cached is not None and cached[0] == key and not fresh:
        return cached[1]
    sync = sync or layer.__dict__.get("_bnn_zero_seen", False)
    pw = hipops.pack_weight(w, plan.center, plan.compute_alpha, sync=sync)
    if pw.has_zero:  # once a layer has shown an exact zero it is always packed synchronously (mask-aware)
        layer.__dict__["_bnn_zero_seen"] = True
    layer.__dict__["_bnn_packed"] = (key, pw)
    _bump("weight_packs")
    return pw


def _replica_packed_weight(layer, master, plan: Plan, fresh: bool = False) -> hipops.PackedWeight:
    """Packed weights of a ``DataParallel`` replica.  Its ``weight`` is a broadcast copy that is new on every forward
    (new storage, version 0), so the replica's own pointer / version say nothing; the values are those of the layer it
    was replicated from.  The pack is therefore keyed on the MASTER's weight (pointer, version, recipe) and kept on the
    master per replica device: one pack — and one host round trip for the zero-weight flag — per device and weight
    version instead of one per forward."""
    mw = master.weight
    w = layer.weight
    key = (mw.data_ptr(), mw._version, tuple(mw.shape), plan.center, plan.compute_alpha)
    cache = master.__dict__.setdefault("_bnn_packed_replicas", {})
    dev = str(w.device)
    hit = cache.get(dev)
    if hit is not None and hit[0] == key and not fresh:
        return hit[1]
    # ``fresh`` (the training forward): never trust the cache — a ``.data`` write on the master does not move its
    # version counter, and the fused backward re-derives What from the current values (forward and backward must agree)
    pw = hipops.pack_weight(w, plan.center, plan.compute_alpha, sync=True)
    cache[dev] = (key, pw)
    _bump("weight_packs")
    return pw


def invalidate(module: nn.Module, executors: bool = True) -> int:
    """Drop every cached packed weight under ``module`` (returns how many).  Needed after writing weights
    through ``.data`` (weight clipping ``p.data.clamp_(-1, 1)``, EMA swaps ``p.data.copy_(ema)``, hand-written
    SGD on ``.data``): those writes do not bump the Parameter's version counter, so nothing else can notice
    them.  ``FusedResNet.refresh()`` calls this for the model it wraps."""
    n = 0
    for m in module.modules():
        if m.__dict__.pop("_bnn_packed", None) is not None:
            n += 1
        m.__dict__.pop("_bnn_packed_replicas", None)
        if executors:       # a residual block's own fused executor (dispatch.BlockFusion); an executor that re-derives
            m.__dict__.pop("_bnn_auto_block", None)     # ITSELF (refresh) passes False: it may be that very object
    from .tails import drop_derived                 # folded BatchNorms / transposed head weights of the per-layer tails
    drop_derived(module)
    return n


def _f32(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Per-channel constants of a ``.half()`` model, widened (exactly) for the fp32 epilogue."""
    return t if t is None or t.dtype == torch.float32 else t.detach().float()


def conv2d(layer, x: torch.Tensor, plan: Plan) -> torch.Tensor:
    """HIP evaluation of ``bnn.layers.Conv2d.forward`` (bnn/layers/conv.py:90-97)."""
    native.require()
    pw = packed_weight(layer, plan)
    # one launch: sign(x) is computed inside the convolution kernel (csrc/bconv_fly.hip)
    out = hipops.bconv2d_direct(x, pw, _f32(layer.bias), _f32(plan.scale), layer.stride, layer.padding,
                                layer.dilation)
    _bump("conv2d")
    return out.to(x.dtype)


def conv2d_train(layer, x: torch.Tensor, plan: Plan) -> torch.Tensor:
    """Same forward under autograd: HIP kernels forward, fp32 library convolutions backward."""
    from . import training
    native.require()
    out = training.conv2d_train(layer, x, plan, packed_weight(layer, plan, sync=False, fresh=True))
    _bump("conv2d_train")
    return out


def conv1d(layer, x: torch.Tensor, plan: Plan) -> torch.Tensor:
    """``Conv1d`` as an ``H == 1`` 2-D convolution (bnn/layers/conv.py:36-43)."""
    native.require()
    pw = packed_weight(layer, plan)
    out = hipops.bconv2d_direct(x.unsqueeze(2), pw, _f32(layer.bias), _f32(plan.scale), (1, layer.stride[0]),
                                (0, layer.padding[0]), (1, layer.dilation[0]))
    _bump("conv1d")
    return out.squeeze(2).to(x.dtype)


def linear(layer, x: torch.Tensor, plan: Plan) -> torch.Tensor:
    """``Linear`` as a 1x1 convolution over 1x1 images (bnn/layers/linear.py:22-27)."""
    native.require()
    pw = packed_weight(layer, plan)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    out = hipops.bconv2d_direct(x2[:, :, None, None], pw, _f32(layer.bias), _f32(plan.scale))
    _bump("linear")
    return out.reshape(*lead, layer.out_features).to(x.dtype)
