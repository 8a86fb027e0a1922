"""bnn_amd — MI355X-native drop-in for the binary Conv2d/Linear inference path of ``bnn``.

    import bnn_amd as bnn
    from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer
    model = bnn.prepare_binary_model(model, bnn.BConfig(
        activation_pre_process=BasicInputBinarizer,
        activation_post_process=bnn.Identity,
        weight_pre_process=XNORWeightBinarizer))

Same API as ``1adrianb/binary-networks-pytorch`` (``bnn/__init__.py``); on a gfx950 device the
converted layers run hand-written HIP XNOR/popcount kernels through the C-ABI in
``include/bnn_hip.h``.
"""
from .version import __version__
from .bconfig import BConfig, Identity
from .binarize import (DEFAULT_MODULE_MAPPING, get_modules_to_binarize, get_unique_devices_,
                       prepare_binary_model, swap_modules_by_name)
from . import layers, ops  # noqa: F401
from . import torch_ops  # noqa: F401  (registers torch.ops.bnn_amd.*)

__all__ = [
    "__version__", "BConfig", "Identity", "DEFAULT_MODULE_MAPPING", "get_modules_to_binarize",
    "get_unique_devices_", "prepare_binary_model", "swap_modules_by_name", "layers", "ops",
]
