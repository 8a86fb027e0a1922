"""ctypes binding of the C-ABI declared in ``include/bnn_hip.h`` (``libbnn_hip.so``).

The library is built in-tree by ``csrc/Makefile`` / ``__graft_entry__.build()`` into
``bnn_amd/_lib/``.  ``require()`` raises ``RuntimeError`` if it cannot be loaded: the HIP path
must never silently turn into something else on a GPU box.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_BASENAME = "libbnn_hip.so"
DEFAULT_LIB_PATH = os.path.join(_HERE, "_lib", LIB_BASENAME)

OCB = 32
FLAG_FORCE_GENERIC = 1
FLAG_WEIGHT_ZEROS = 2
FLAG_WEIGHTS_SGPR = 4
FLAG_ACT_NONNEG = 32
FLAG_THROUGHPUT = 64
HBLOCK_CHANNEL_LANES = 128
STEM_EXACT_FP32 = 1
STEM_FP16 = 4
ABI_VERSION = 15
DTYPE_F32 = 0
DTYPE_F16 = 1

# every symbol include/bnn_hip.h declares (tests assert the .so exports all of them)
EXPORTED_SYMBOLS = (
    "bnn_hip_abi_version", "bnn_hip_status_string", "bnn_hip_launch_count", "bnn_hip_device_info",
    "bnn_hip_act_words", "bnn_hip_weight_layout", "bnn_hip_pack_act_f32", "bnn_hip_bn_act_pack_f32",
    "bnn_hip_avgpool_pack_f32", "bnn_hip_bn_relu_maxpool_pack_f32", "bnn_hip_stem7x7_bn_relu_pool_pack_f32",
    "bnn_hip_pack_weight_f32", "bnn_hip_bconv2d",
    "bnn_hip_bconv2d_fused", "bnn_hip_bconv2d_dot", "bnn_hip_blinear",
    "bnn_hip_conv_workspace_bytes", "bnn_hip_bconv2d_f32", "bnn_hip_probe_int_alu", "bnn_hip_avgpool_fc_f32", "bnn_hip_sign_thresholds_f32", "bnn_hip_pack_act_f16", "bnn_hip_orpool_packed",
    "bnn_hip_grad_weight_pack_bytes", "bnn_hip_grad_pack_weight_f32", "bnn_hip_bconv_grad_input_f32",
    "bnn_hip_bconv_grad_weight_splits", "bnn_hip_bconv_grad_weight_f32",
    "bnn_hip_bconv2d_direct", "bnn_hip_bconv2d_direct_plan", "bnn_hip_shortcut_fold_supported",
    "bnn_hip_probe_clock", "bnn_hip_pack_act_ste_f32", "bnn_hip_bconv_grad_input_packed_f32",
    "bnn_hip_bconv_grad_weight_packed_f32", "bnn_hip_bn_train_workspace_bytes", "bnn_hip_bn_train_forward_f32",
    "bnn_hip_bn_train_backward_f32", "bnn_hip_bn_relu_maxpool_train_forward_f32",
    "bnn_hip_bn_relu_maxpool_train_backward_f32", "bnn_hip_xnor_weight_forward_f32", "bnn_hip_xnor_weight_backward_f32",
    "bnn_hip_bn_act_f32", "bnn_hip_avgpool_fc_workspace_bytes", "bnn_hip_avgpool_fc_ws_f32",
    "bnn_hip_stem7x7_conv_f32", "bnn_hip_stem7x7_wgrad_workspace_bytes", "bnn_hip_stem7x7_wgrad_f32",
    "bnn_hip_avgpool2x2_backward_f32", "bnn_hip_xnor_grad_pack_weight_f32",
    "bnn_hip_hblock_supported", "bnn_hip_hblock_layout_of", "bnn_hip_hblock_pack_weights", "bnn_hip_hblock_forward",
    "bnn_hip_avgpool2_bn_pack2_f32", "bnn_hip_hblock_pack_weights_cl", "bnn_hip_stem7x7_bn_relu_pool_pack_affine_f32",
    "bnn_hip_hblock_pool_supported", "bnn_hip_hblock_pool_forward",
    "bnn_hip_hblock_shortcut_supported", "bnn_hip_hblock_pack_shortcut_weights", "bnn_hip_hblock_shortcut_forward",
)


class ConvDesc(ctypes.Structure):
    """``bnn_hip_conv_desc``"""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "N", "C", "H", "W", "O", "KH", "KW", "stride_h", "stride_w", "pad_h", "pad_w",
        "dil_h", "dil_w", "flags")]


class WLayout(ctypes.Structure):
    """``bnn_hip_wlayout``"""
    _fields_ = [("cw32", ctypes.c_int32), ("cwc", ctypes.c_int32), ("nchunk", ctypes.c_int32),
                ("taps", ctypes.c_int32), ("o_pad", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("n_words", ctypes.c_int64)]


class FlyPlan(ctypes.Structure):
    """``bnn_hip_fly_plan``"""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "images_per_band", "rows_per_band", "waves", "blocks_per_unit", "pack_ahead", "fine_head", "fine_tail", "producers",
        "lds_bytes", "n_bands")]


class HBlockDesc(ctypes.Structure):
    """``bnn_hip_hblock_desc``"""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "N", "C_in", "H", "W", "planes", "flags", "rows_per_band", "images_per_band", "waves", "reserved")]


class HBlockLayout(ctypes.Structure):
    """``bnn_hip_hblock_layout``"""
    _fields_ = [("weight_words", ctypes.c_int64), ("w_off", ctypes.c_int64 * 3), ("const_floats", ctypes.c_int64),
                ("alpha_off", ctypes.c_int64 * 3), ("pack_a_off", ctypes.c_int64 * 2), ("pack_b_off", ctypes.c_int64 * 2),
                ("next_a_off", ctypes.c_int64), ("next_b_off", ctypes.c_int64)]


class DevInfo(ctypes.Structure):
    """``bnn_hip_devinfo``"""
    _fields_ = [("name", ctypes.c_char * 64), ("arch", ctypes.c_char * 32),
                ("compute_units", ctypes.c_int32), ("clock_khz", ctypes.c_int32),
                ("mem_clock_khz", ctypes.c_int32), ("mem_bus_bits", ctypes.c_int32),
                ("wavefront", ctypes.c_int32), ("lds_bytes_per_block", ctypes.c_int32),
                ("total_mem_bytes", ctypes.c_int64), ("l2_bytes", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class Epilogue(ctypes.Structure):
    """``bnn_hip_epilogue``"""
    _fields_ = [("alpha", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("post_scale", ctypes.c_void_p),
                ("bn_scale", ctypes.c_void_p), ("bn_shift", ctypes.c_void_p),
                ("residual", ctypes.c_void_p), ("prelu", ctypes.c_void_p),
                ("relu", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("out_f32", ctypes.c_void_p), ("out_P", ctypes.c_void_p), ("out_M", ctypes.c_void_p),
                ("pack_scale", ctypes.c_void_p), ("pack_shift", ctypes.c_void_p),
                ("out_c_offset", ctypes.c_int32), ("out_c_total", ctypes.c_int32),
                ("sign_thresholds", ctypes.c_void_p),
                # ABI 11: the block's shortcut convolution folded into this one (include/bnn_hip.h)
                ("sc_P", ctypes.c_void_p), ("sc_wbits", ctypes.c_void_p), ("sc_alpha", ctypes.c_void_p),
                ("sc_bn_scale", ctypes.c_void_p), ("sc_bn_shift", ctypes.c_void_p),
                ("sc_C", ctypes.c_int32), ("sc_in_hw", ctypes.c_int32)]


EPI_RES_AFTER_ACT = 1
EPI_PACK_BEFORE_RES = 2
EPI_PACK_RELU = 4


class NativeError(RuntimeError):
    pass


_lock = threading.Lock()
_lib: Optional[ctypes.CDLL] = None
_load_error: Optional[str] = None

_vp = ctypes.c_void_p
_i = ctypes.c_int


def lib_path() -> str:
    return os.environ.get("BNN_AMD_LIB", DEFAULT_LIB_PATH)


def _declare(lib: ctypes.CDLL) -> None:
    lib.bnn_hip_abi_version.restype = _i
    lib.bnn_hip_status_string.restype = ctypes.c_char_p
    lib.bnn_hip_status_string.argtypes = [_i]
    lib.bnn_hip_launch_count.restype = ctypes.c_uint64
    lib.bnn_hip_device_info.argtypes = [_i, ctypes.POINTER(DevInfo)]
    lib.bnn_hip_act_words.argtypes = [_i]
    lib.bnn_hip_weight_layout.argtypes = [_i, _i, _i, _i, ctypes.POINTER(WLayout)]
    lib.bnn_hip_pack_act_f32.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp, _vp]
    lib.bnn_hip_orpool_packed.argtypes = [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]
    lib.bnn_hip_pack_act_f16.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp, _vp]
    lib.bnn_hip_avgpool_pack_f32.argtypes = [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]
    lib.bnn_hip_bn_act_pack_f32.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]
    lib.bnn_hip_stem7x7_bn_relu_pool_pack_f32.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]
    lib.bnn_hip_bn_relu_maxpool_pack_f32.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i,
                                                     _vp, _vp, _vp, _vp]
    lib.bnn_hip_pack_weight_f32.argtypes = [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]
    lib.bnn_hip_bconv2d.argtypes = [ctypes.POINTER(ConvDesc)] + [_vp] * 9
    lib.bnn_hip_shortcut_fold_supported.argtypes = [ctypes.POINTER(ConvDesc), _i]
    lib.bnn_hip_shortcut_fold_supported.restype = _i
    lib.bnn_hip_bconv2d_fused.argtypes = [ctypes.POINTER(ConvDesc)] + [_vp] * 4 + \
        [ctypes.POINTER(Epilogue), _vp]
    lib.bnn_hip_bconv2d_dot.argtypes = [ctypes.POINTER(ConvDesc)] + [_vp] * 6
    lib.bnn_hip_blinear.argtypes = [_i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]
    lib.bnn_hip_conv_workspace_bytes.restype = ctypes.c_size_t
    lib.bnn_hip_conv_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
    lib.bnn_hip_bconv2d_f32.argtypes = [ctypes.POINTER(ConvDesc)] + [_vp] * 9
    lib.bnn_hip_bconv2d_direct_plan.argtypes = [ctypes.POINTER(ConvDesc), ctypes.POINTER(FlyPlan)]
    lib.bnn_hip_bconv2d_direct.argtypes = [ctypes.POINTER(ConvDesc), _vp, _i] + [_vp] * 6 + \
        [ctypes.POINTER(FlyPlan), _vp]
    lib.bnn_hip_stem7x7_bn_relu_pool_pack_affine_f32.argtypes = [_vp] * 6 + [_i, _i, _i, _i, _vp, _vp, _vp, _vp]
    lib.bnn_hip_stem7x7_conv_f32.argtypes = [_vp, _vp, _i, _i, _i, _i, _vp, _vp]
    lib.bnn_hip_avgpool_fc_f32.argtypes = [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp]
    lib.bnn_hip_stem7x7_wgrad_workspace_bytes.argtypes = [_i, _i, _i]
    lib.bnn_hip_stem7x7_wgrad_workspace_bytes.restype = ctypes.c_size_t
    lib.bnn_hip_stem7x7_wgrad_f32.argtypes = [_vp, _vp, _i, _i, _i, _vp, ctypes.c_size_t, _vp, _vp]
    lib.bnn_hip_avgpool2x2_backward_f32.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp]
    lib.bnn_hip_xnor_grad_pack_weight_f32.argtypes = [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]
    lib.bnn_hip_avgpool_fc_workspace_bytes.argtypes = [_i, _i]
    lib.bnn_hip_avgpool_fc_workspace_bytes.restype = ctypes.c_size_t
    lib.bnn_hip_avgpool_fc_ws_f32.argtypes = [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, ctypes.c_size_t, _vp]
    lib.bnn_hip_sign_thresholds_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]
    lib.bnn_hip_grad_weight_pack_bytes.restype = ctypes.c_size_t
    lib.bnn_hip_grad_weight_pack_bytes.argtypes = [_i, _i, _i]
    lib.bnn_hip_grad_pack_weight_f32.argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp]
    lib.bnn_hip_bconv_grad_input_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]
    lib.bnn_hip_bconv_grad_weight_splits.argtypes = [_i, _i, _i, _i]
    lib.bnn_hip_bconv_grad_weight_f32.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]
    lib.bnn_hip_pack_act_ste_f32.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]
    lib.bnn_hip_bconv_grad_input_packed_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]
    lib.bnn_hip_bconv_grad_weight_packed_f32.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]
    lib.bnn_hip_bn_train_workspace_bytes.restype = ctypes.c_size_t
    lib.bnn_hip_bn_train_workspace_bytes.argtypes = [_i, _i, _i]
    _f = ctypes.c_float
    lib.bnn_hip_bn_act_f32.argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]
    lib.bnn_hip_bn_act_f32.restype = _i
    lib.bnn_hip_bn_train_forward_f32.argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.bnn_hip_bn_train_backward_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.bnn_hip_bn_relu_maxpool_train_forward_f32.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp, _f, _f] + [_vp] * 8
    lib.bnn_hip_bn_relu_maxpool_train_backward_f32.argtypes = [_vp] * 7 + [_i, _i, _i, _i] + [_vp] * 5
    lib.bnn_hip_xnor_weight_forward_f32.argtypes = [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]
    lib.bnn_hip_xnor_weight_backward_f32.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]
    lib.bnn_hip_probe_int_alu.argtypes = [_i, _i, ctypes.POINTER(ctypes.c_double),
                                          ctypes.POINTER(ctypes.c_double), _vp]
    lib.bnn_hip_probe_clock.argtypes = [_i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _vp]
    lib.bnn_hip_avgpool2_bn_pack2_f32.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]
    lib.bnn_hip_hblock_supported.argtypes = [ctypes.POINTER(HBlockDesc)]
    lib.bnn_hip_hblock_layout_of.argtypes = [_i, _i, ctypes.POINTER(HBlockLayout)]
    lib.bnn_hip_hblock_pack_weights.argtypes = [_i, _i, _vp, _vp, _vp, _vp, _vp]
    lib.bnn_hip_hblock_pack_weights_cl.argtypes = [_i, _i, _vp, _vp, _vp, _vp, _vp]
    lib.bnn_hip_hblock_forward.argtypes = [ctypes.POINTER(HBlockDesc)] + [_vp] * 7
    lib.bnn_hip_hblock_pool_supported.argtypes = [ctypes.POINTER(HBlockDesc)]
    lib.bnn_hip_hblock_pool_forward.argtypes = [ctypes.POINTER(HBlockDesc)] + [_vp] * 9
    lib.bnn_hip_hblock_shortcut_supported.argtypes = [ctypes.POINTER(HBlockDesc)]
    lib.bnn_hip_hblock_pack_shortcut_weights.argtypes = [_i, _i, _vp, _vp, _vp]
    lib.bnn_hip_hblock_shortcut_forward.argtypes = [ctypes.POINTER(HBlockDesc)] + [_vp] * 10


def load() -> Optional[ctypes.CDLL]:
    """Load the library once; returns ``None`` (and remembers why) when that fails."""
    global _lib, _load_error
    with _lock:
        if _lib is not None or _load_error is not None:
            return _lib
        path = lib_path()
        try:
            lib = ctypes.CDLL(path)
            for name in EXPORTED_SYMBOLS:
                getattr(lib, name)
            _declare(lib)
            if lib.bnn_hip_abi_version() != ABI_VERSION:
                raise OSError(f"ABI version mismatch: {lib.bnn_hip_abi_version()} != {ABI_VERSION}")
            _lib = lib
        except (OSError, AttributeError) as exc:  # missing file, missing libamdhip64, missing symbol
            _load_error = f"{path}: {exc}"
        return _lib


def available() -> bool:
    return load() is not None


def require() -> ctypes.CDLL:
    lib = load()
    if lib is None:
        raise NativeError(
            "bnn_amd: the HIP library could not be loaded (" + str(_load_error) + "). "
            "Build it with `make -C binary-networks-pytorch_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. The GPU path does not fall "
            "back to another implementation.")
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = require().bnn_hip_status_string(status).decode()
        raise NativeError(f"bnn_amd: {what} failed: {msg} (status {status})")


def weight_layout(O: int, C: int, KH: int, KW: int) -> WLayout:
    L = WLayout()
    check(require().bnn_hip_weight_layout(O, C, KH, KW, ctypes.byref(L)), "bnn_hip_weight_layout")
    return L


def device_info(device: int = 0) -> dict:
    info = DevInfo()
    check(require().bnn_hip_device_info(device, ctypes.byref(info)), "bnn_hip_device_info")
    return {f: (getattr(info, f).decode() if isinstance(getattr(info, f), bytes) else getattr(info, f))
            for f, _ in DevInfo._fields_ if f != "reserved"}


def launch_count() -> int:
    return int(require().bnn_hip_launch_count())
