"""Fused inference for binary ResNets — facade.

The implementation lives in four modules (round 5: this file used to hold all of it):

    executor.py   fold_bn, tap_binary_inputs, FusedResNet / FusedBlocks     the executors: 19 launches per ResNet-18
                                                                            forward, eager or as HIP graphs
    pipeline.py   concurrent_streams, PipelinedInference, TwoHalves         several batches in flight (across calls /
                                                                            inside one call)
    tails.py      per_layer_forward, library_tails, cached_fold,            the per-layer path's one-launch tails and
                  eval_tail / eval_stem / eval_head                         the switches of that tier
    dispatch.py   no_model_fusion, BlockFusion, AutoFusion, auto_fusion,    what ``block(x)`` / ``model(x)`` dispatch to
                  auto_forward, install_auto_fusion, optimize_for_inference by themselves (the drop-in tiers)

``from bnn_amd.inference import FusedResNet`` etc. keep working; module-level state (``_LIBRARY_TAILS`` ...) is read
through to the module that owns it.
"""
from __future__ import annotations

from . import dispatch as _dispatch
from . import executor as _executor
from . import pipeline as _pipeline
from . import tails as _tails
from .dispatch import (AutoFusion, BlockFusion, auto_block_forward, auto_forward, auto_fusion,  # noqa: F401
                       install_auto_fusion, no_model_fusion, optimize_for_inference, uninstall_auto_fusion)
from .executor import (FusedBlocks, FusedResNet, FusionError, fold_bn, is_native_model, resnet_shaped,  # noqa: F401
                       tap_binary_inputs)
from .pipeline import STREAM_PROBE_LOG, PipelinedInference, TwoHalves, concurrent_streams  # noqa: F401
from .tails import (cached_fold, eval_head, eval_stem, eval_tail, library_tails, per_layer_forward)  # noqa: F401

_PARTS = (_executor, _pipeline, _tails, _dispatch)


def __getattr__(name: str):
    """Everything else (private helpers, counters such as ``_LIBRARY_TAILS`` whose value changes at run time) is looked
    up in the owning module at the time of the access."""
    for mod in _PARTS:
        try:
            return getattr(mod, name)
        except AttributeError:
            continue
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
