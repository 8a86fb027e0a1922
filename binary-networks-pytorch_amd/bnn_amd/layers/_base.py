"""Behaviour shared by the binary Conv1d / Conv2d / Linear layers.

Reference template (identical in ``bnn/layers/conv.py:10-117`` and ``bnn/layers/linear.py:9-44``):

    forward(x) = post( F.op( pre(x), wpre(weight), bias ), x )

with ``pre``/``post``/``wpre`` instantiated from the layer's ``BConfig`` at construction and
registered as sub-modules (so they appear in ``state_dict()`` and ``repr``).
"""
from __future__ import annotations

from typing import Optional

import torch.nn as nn

from ..bconfig import BConfig
from .helpers import copy_paramters


class BinaryLayerMixin:
    """Hook construction + ``from_module`` for classes that also derive from a float layer."""

    _FLOAT_MODULE: type = nn.Module

    def _init_hooks(self, bconfig: Optional[BConfig]) -> None:
        assert bconfig, "bconfig is required for a binarized module"
        self.bconfig = bconfig
        self.activation_pre_process = bconfig.activation_pre_process()
        self.activation_post_process = bconfig.activation_post_process(self)
        self.weight_pre_process = bconfig.weight_pre_process()

    def train(self, mode: bool = True):
        """Switching between training and evaluation drops the derived data (packed weights): the packed-weight cache is
        keyed on the weight's storage pointer + autograd version counter, which writes through ``.data`` —
        ``p.data.clamp_(-1, 1)`` after the optimizer step, EMA swaps — do not move.  Whatever was written while training
        is therefore seen by the first forward after ``model.eval()`` (and the other way round); the reference has no such
        state to go stale (it re-binarises on every forward, bnn/layers/conv.py:92).  A ``.data`` write BETWEEN two
        forwards of the same mode still needs ``fastpath.invalidate(model)``."""
        if bool(mode) != self.training:
            self.__dict__.pop("_bnn_packed", None)
            self.__dict__.pop("_bnn_packed_replicas", None)
        return super().train(mode)

    def _replicate_for_data_parallel(self):
        """``nn.DataParallel`` (examples/cifar10.py:74-77) makes its per-device replicas with this on EVERY forward:
        ``__dict__`` is copied shallowly and the parameters are replaced by freshly broadcast copies.  The replica
        remembers the layer it was made from, so that the packed weights it needs are cached there per device and
        weight version (``fastpath.packed_weight``) instead of being re-derived — with a blocking read of the
        zero-weight flag — on every forward of every replica."""
        replica = super()._replicate_for_data_parallel()
        replica.__dict__["_bnn_master"] = self.__dict__.get("_bnn_master", self)
        replica.__dict__.pop("_bnn_packed", None)       # (device 0's entry, copied with __dict__)
        return replica

    @classmethod
    def _ctor_kwargs(cls, mod: nn.Module) -> dict:  # pragma: no cover - overridden
        raise NotImplementedError

    @classmethod
    def from_module(cls, mod: nn.Module, bconfig: Optional[BConfig] = None, update: bool = False):
        """Build a binary twin of ``mod`` that SHARES its ``weight``/``bias`` Parameters.

        Accepts the float class or an already-binary instance of ``cls`` (re-binarisation with
        a new recipe), exactly as ``bnn/layers/conv.py:99-117``.
        """
        assert type(mod) == cls._FLOAT_MODULE or type(mod) == cls, (
            "bnn." + cls.__name__ + ".from_float only works for " + cls._FLOAT_MODULE.__name__)
        if not bconfig:
            assert hasattr(mod, "bconfig"), "The input modele requires a predifined bconfig"
            assert mod.bconfig, "The input modele bconfig is invalid"
            bconfig = mod.bconfig
        twin = cls(**cls._ctor_kwargs(mod), bconfig=bconfig)
        twin.weight = mod.weight
        twin.bias = mod.bias
        if update:
            copy_paramters(mod, twin, bconfig)
        return twin
