"""Binarizer hook modules (host-side mirror of the reference's ``bnn/ops.py``).

These modules define the *semantics* of a binary layer and are what ``BConfig`` refers to.  On a
HIP device, in inference mode, ``bnn_amd.layers`` recognises the combination
``BasicInputBinarizer`` + ``XNORWeightBinarizer`` (+ ``Identity`` | ``BasicScaleBinarizer``) and
evaluates it with the bit-packed XNOR/popcount kernels (see ``fastpath.py``); the ``forward``
methods below are the autograd-capable composition used for training, for CPU tensors and for
hook combinations outside the accelerated path.
"""
from __future__ import annotations

from functools import partial
from typing import Any, List, Optional

import torch
import torch.nn as nn

__all__ = [
    "BinarizerBase", "SignActivation", "SignActivationStochastic", "XNORWeightBinarizer",
    "BasicInputBinarizer", "StochasticInputBinarizer", "AdvancedInputBinarizer",
    "BasicScaleBinarizer",
]


class _Factory:
    """Callable returned by ``X.with_args(**kw)``; chainable, repr of the underlying partial.

    Same contract as the helper in ``bnn/ops.py:10-35`` (itself borrowed from
    ``torch.quantization.observer``): calling it builds a *new* module each time.
    """

    def __init__(self, target: Any, **kwargs: Any) -> None:
        self.p = partial(target, **kwargs)

    def __call__(self, *args: Any, **kwargs: Any) -> Any:
        return self.p(*args, **kwargs)

    def with_args(self, **kwargs: Any) -> "_Factory":
        return _Factory(self.p, **kwargs)

    def __repr__(self) -> str:
        return repr(self.p)


class BinarizerBase(nn.Module):
    """Base class of every hook; provides ``with_args`` currying (``bnn/ops.py:40-48``)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # pragma: no cover - abstract
        raise NotImplementedError

    @classmethod
    def with_args(cls, **kwargs: Any) -> _Factory:
        return _Factory(cls, **kwargs)


class SignActivation(torch.autograd.Function):
    """``sign(x)`` forward ({-1, 0, +1}: ``sign(0) == 0``), hard-tanh straight-through backward.

    Reference: ``bnn/ops.py:63-73``.
    """

    @staticmethod
    def forward(ctx, input: torch.Tensor) -> torch.Tensor:  # noqa: A002
        ctx.save_for_backward(input)
        return torch.sign(input)

    @staticmethod
    def backward(ctx, grad_output: torch.Tensor) -> torch.Tensor:
        (x,) = ctx.saved_tensors
        return grad_output.masked_fill(x.abs() >= 1, 0)


class SignActivationStochastic(SignActivation):
    """Stochastic binarisation (``bnn/ops.py:76-92``); training-only, never on the fast path.

    Unlike the reference this does not modify its input in place.
    """

    @staticmethod
    def forward(ctx, input: torch.Tensor) -> torch.Tensor:  # noqa: A002
        ctx.save_for_backward(input)
        noise = torch.rand_like(input) - 0.5
        return ((input + 1) / 2 + noise).clamp_(0, 1).round_().mul_(2).sub_(1)


class XNORWeightBinarizer(BinarizerBase):
    """XNOR-Net weight binarisation: ``sign(W) * mean|W|`` per output channel.

    Reference: ``bnn/ops.py:95-140``.  ``center_weights`` subtracts the mean over the input
    channel dimension first; ``compute_alpha=False`` returns plain ``sign(W)``.
    """

    def __init__(self, compute_alpha: bool = True, center_weights: bool = False) -> None:
        super().__init__()
        self.compute_alpha = compute_alpha
        self.center_weights = center_weights

    @staticmethod
    def _compute_alpha(w: torch.Tensor) -> torch.Tensor:
        if w.dim() not in (2, 3, 4):
            raise ValueError(f"Expected ndims equal with 2 or 4, but found {w.dim()}")
        reduce_dims = list(range(1, w.dim()))
        return w.abs().sum(dim=reduce_dims, keepdim=True) / w[0].numel()

    def forward(self, w: torch.Tensor) -> torch.Tensor:
        if self.center_weights:
            w = w - w.mean(dim=1, keepdim=True)
        signed = SignActivation.apply(w)
        if self.compute_alpha:
            signed = signed * self._compute_alpha(w)
        return signed

    def extra_repr(self) -> str:
        return f"compute_alpha={self.compute_alpha}, center_weights={self.center_weights}"


class BasicInputBinarizer(BinarizerBase):
    """Module form of :class:`SignActivation` (``bnn/ops.py:143-152``)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return SignActivation.apply(x)


class StochasticInputBinarizer(BinarizerBase):
    """Module form of :class:`SignActivationStochastic` (``bnn/ops.py:155-164``)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return SignActivationStochastic.apply(x)


class AdvancedInputBinarizer(BinarizerBase):
    """``sign(f(t*x))`` (``bnn/ops.py:167-177``).  As upstream, the sign is taken under ``torch.no_grad()``: the result
    carries NO gradient path (the reference returns a tensor that does not require grad, so a layer using this hook
    trains its weights but passes nothing back through its input).  ``soft_gradient=True`` (an extension, off by
    default) gives the value of ``sign(f(t*x))`` with the gradient of ``f(t*x)`` — what the class name suggests."""

    def __init__(self, derivative_funct=torch.tanh, t: int = 5, soft_gradient: bool = False) -> None:
        super().__init__()
        self.derivative_funct = derivative_funct
        self.t = t
        self.soft_gradient = soft_gradient

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        soft = self.derivative_funct(x * self.t)
        if self.soft_gradient:
            return torch.sign(soft).detach() + (soft - soft.detach())
        with torch.no_grad():
            return torch.sign(soft)


class BasicScaleBinarizer(BinarizerBase):
    """Learned per-output-channel scale applied to the layer output (``bnn/ops.py:180-205``)."""

    def __init__(self, module: nn.Module, shape: Optional[List[int]] = None) -> None:
        super().__init__()
        if isinstance(module, nn.Linear):
            channels = module.out_features
        elif hasattr(module, "out_channels"):
            channels = module.out_channels
        else:
            raise Exception(f"Unknown layer of type {type(module)} missing out_channels")
        if shape is None:
            shape = [1, channels] + [1] * (module.weight.dim() - 2)
        self.alpha = nn.Parameter(torch.ones(*shape))

    def forward(self, layer_out: torch.Tensor, layer_in: torch.Tensor) -> torch.Tensor:
        return layer_out.mul_(self.alpha)

    def extra_repr(self) -> str:
        return str(list(self.alpha.size()))
