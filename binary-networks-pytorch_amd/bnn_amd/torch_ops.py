"""``torch.ops.bnn_amd.*`` — the hot path as PyTorch custom operators.

The reference has no operator boundary of its own (its layers call ``Tensor.sign`` / ``F.conv2d``
directly: ``bnn/layers/conv.py:90-97``, ``bnn/ops.py:129-140``); SURVEY §8(b) asks the replacement to be
reachable as custom ops so that graph-level tooling (``torch.export``, ``torch.compile`` with opaque
ops, FX passes, profilers) sees ONE node per binary layer.  Each operator is a thin shim: tensors in,
raw device pointers into the C-ABI of ``include/bnn_hip.h`` (``bnn_amd/hipops.py``), tensors out.  There
is no CPU or composite implementation registered — calling them with CPU tensors raises, as the product
path must (the layers' torch composition is what runs on CPU, not these ops).

    y  = torch.ops.bnn_amd.binary_conv2d(x, weight, bias, post_scale, stride, padding, dilation,
                                         center_weights, compute_alpha)
    y  = torch.ops.bnn_amd.binary_linear(x, weight, bias, post_scale, center_weights, compute_alpha)
    P, M = torch.ops.bnn_amd.pack_sign(x)                      # BasicInputBinarizer as bit planes

``binary_conv2d`` has an autograd formula (``bnn_amd/training.py``: straight-through estimator for the
input, the reference's XNOR weight binarizer differentiated by torch for the weight), so it can also
stand in a training graph.
"""
from __future__ import annotations

import collections
import threading
import weakref
from typing import List, Optional, Tuple

import torch

from . import hipops, native

_NS = "bnn_amd"


_PACK_CACHE: "collections.OrderedDict" = collections.OrderedDict()
_PACK_CACHE_MAX = 256
# nn.DataParallel calls the ops from one thread per GPU (examples/cifar10.py:76) and the weak-reference callbacks run on
# whichever thread drops a tensor: every access to the dictionary is under this lock (re-entrant: a callback may fire
# while the owner of the lock is inside the dictionary)
_PACK_LOCK = threading.RLock()


def _packed(weight: torch.Tensor, center: bool, compute_alpha: bool) -> hipops.PackedWeight:
    """Packed form of ``weight``, cached on (storage pointer, version counter, shape, recipe): the pack reads its
    zero-weight flag with a blocking ``.item()``, which would make the ops unusable under HIP-graph capture and slow
    in a training graph if it ran on every call.  (Writes through ``weight.data`` bypass the version counter: call
    ``torch_ops.clear_cache()`` after them.)

    An entry keeps a weak reference to the tensor it was made from and is only trusted while that very tensor is
    alive: the caching allocator hands the address of a freed weight to the next tensor of the same shape, whose
    version counter is 0 again — pointer + version alone would return the previous tensor's signs and alpha."""
    key = (weight.data_ptr(), weight._version, str(weight.device), tuple(weight.shape), center, compute_alpha)
    strict = fastpath_strict()
    with _PACK_LOCK:
        hit = None if strict else _PACK_CACHE.get(key)
        if hit is not None and hit[0]() is weight:
            _PACK_CACHE.move_to_end(key)
            return hit[1]
    pw = hipops.pack_weight(weight, center, compute_alpha)      # (outside the lock: it synchronises with the device)
    if strict:
        return pw

    def _drop(_r, k=key):
        with _PACK_LOCK:
            _PACK_CACHE.pop(k, None)
    try:
        ref = weakref.ref(weight, _drop)   # dropped with its tensor
    except TypeError:       # (fake / functional tensor wrappers): not cacheable
        return pw
    with _PACK_LOCK:
        _PACK_CACHE[key] = (ref, pw)
        _PACK_CACHE.move_to_end(key)
        while len(_PACK_CACHE) > _PACK_CACHE_MAX:
            _PACK_CACHE.popitem(last=False)
    return pw


def fastpath_strict() -> bool:
    from . import fastpath
    return fastpath.strict_weights()


def clear_cache() -> None:
    with _PACK_LOCK:
        _PACK_CACHE.clear()


def _need_gpu(*ts: Optional[torch.Tensor]) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise native.NativeError(f"torch.ops.{_NS}: tensors must live on a HIP device (there is no CPU kernel; "
                                     "on CPU the layers run their torch composition)")


@torch.library.custom_op(f"{_NS}::pack_sign", mutates_args=(), device_types="cuda")
def pack_sign(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """sign(x) as two bit planes ``P`` (x > 0) and ``M`` (x < 0), int64 ``[N, ceil(C/64), H, W]``."""
    _need_gpu(x)
    a = hipops.pack_act(x)
    return a.P, a.M


@pack_sign.register_fake
def _(x):
    n, c, h, w = x.shape
    shape = (n, (c + 63) // 64, h, w)
    return x.new_empty(shape, dtype=torch.int64), x.new_empty(shape, dtype=torch.int64)


@torch.library.custom_op(f"{_NS}::binary_conv2d", mutates_args=(), device_types="cuda")
def binary_conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                  post_scale: Optional[torch.Tensor], stride: List[int], padding: List[int],
                  dilation: List[int], center_weights: bool, compute_alpha: bool) -> torch.Tensor:
    """``post_scale * (conv2d(sign(x), sign(Wc) * alpha) + bias)`` — ``bnn.layers.Conv2d.forward`` with
    ``BasicInputBinarizer`` / ``XNORWeightBinarizer`` / optional ``BasicScaleBinarizer`` hooks."""
    _need_gpu(x, weight, bias, post_scale)
    pw = _packed(weight, center_weights, compute_alpha)
    return hipops.bconv2d_direct(x, pw, bias, post_scale, tuple(stride), tuple(padding), tuple(dilation))


@binary_conv2d.register_fake
def _(x, weight, bias, post_scale, stride, padding, dilation, center_weights, compute_alpha):
    ho, wo = hipops.conv_out_hw(x.shape[2], x.shape[3], weight.shape[2], weight.shape[3], tuple(stride),
                                tuple(padding), tuple(dilation))
    return x.new_empty((x.shape[0], weight.shape[0], ho, wo), dtype=torch.float32)


def _conv_setup(ctx, inputs, output):
    x, weight, bias, post_scale, stride, padding, dilation, center, compute_alpha = inputs
    ctx.save_for_backward(x, weight, post_scale, bias)
    ctx.conf = (list(stride), list(padding), list(dilation), center, compute_alpha, bias is not None)


def _conv_backward(ctx, g):
    from .ops import XNORWeightBinarizer
    x, weight, post_scale, bias = ctx.saved_tensors
    stride, padding, dilation, center, compute_alpha, has_bias = ctx.conf
    gs = None
    if post_scale is not None:
        s = post_scale.reshape(1, -1, 1, 1)
        if ctx.needs_input_grad[3]:
            # d out / d post_scale = the PRE-scale output, recomputed by the forward kernels (dividing the saved
            # output by the scale would give NaN/inf for a zero entry of post_scale, where the gradient is finite)
            pre = hipops.bconv2d_direct(x, _packed(weight, center, compute_alpha), bias, None,
                                        tuple(stride), tuple(padding), tuple(dilation))
            gs = (g * pre).sum(dim=(0, 2, 3)).reshape(post_scale.shape)
        g = g * s
    with torch.enable_grad():
        w = weight.detach().requires_grad_(True)
        w_hat = XNORWeightBinarizer(compute_alpha=compute_alpha, center_weights=center)(w)
    gx, gwh, gb = torch.ops.aten.convolution_backward(
        g.contiguous(), torch.sign(x), w_hat.detach(), [weight.shape[0]] if has_bias else None, stride, padding,
        dilation, False, [0, 0], 1, [ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]])
    if gx is not None:
        gx = gx.masked_fill(x.abs() >= 1, 0)       # hard-tanh straight-through estimator (bnn/ops.py:68-73)
    gw = torch.autograd.grad(w_hat, w, gwh)[0] if gwh is not None else None
    return gx, gw, gb, gs, None, None, None, None, None


binary_conv2d.register_autograd(_conv_backward, setup_context=_conv_setup)


@torch.library.custom_op(f"{_NS}::binary_linear", mutates_args=(), device_types="cuda")
def binary_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                  post_scale: Optional[torch.Tensor], center_weights: bool, compute_alpha: bool) -> torch.Tensor:
    """``bnn.layers.Linear.forward`` (bnn/layers/linear.py:22-27) as a 1x1 convolution over 1x1 images."""
    _need_gpu(x, weight, bias, post_scale)
    pw = _packed(weight, center_weights, compute_alpha)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    out = hipops.bconv2d_direct(x2[:, :, None, None], pw, bias, post_scale)
    return out.reshape(*lead, weight.shape[0])


@binary_linear.register_fake
def _(x, weight, bias, post_scale, center_weights, compute_alpha):
    return x.new_empty((*x.shape[:-1], weight.shape[0]), dtype=torch.float32)


OPS = ("pack_sign", "binary_conv2d", "binary_linear")
