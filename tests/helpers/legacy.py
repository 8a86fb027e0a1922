"""ctypes binding of libbnn_hip_legacy.so (csrc/legacy/legacy_api.h): TEST-ONLY kernels kept as independent
implementations the product kernels are compared with — the round-2 stem (conv tile staged through LDS) and the
LDS-staged weight tile north_star describes.  Until ABI 11 they rode in libbnn_hip.so behind flags; nothing under
bnn_amd/ loads this library."""
import ctypes
import os

import torch

from bnn_amd import hipops, native

_PATH = os.path.join(os.path.dirname(native.DEFAULT_LIB_PATH), "libbnn_hip_legacy.so")
_lib = None
_vp, _i = ctypes.c_void_p, ctypes.c_int


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_PATH)
        _lib.bnn_hip_legacy_stem_staged.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]
        _lib.bnn_hip_legacy_bconv2d_lds.argtypes = [ctypes.POINTER(native.ConvDesc)] + [_vp] * 8
    return _lib


def stem_staged(x, w, bn_scale, bn_shift, fp16=False, flags=None):
    """The round-2 stem kernel: same signature / results as ``hipops.stem7x7`` -> (fp32 NCHW, PackedAct)."""
    N, _, H, W = x.shape
    hc, wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    hp, wp = (hc - 1) // 2 + 1, (wc - 1) // 2 + 1
    with torch.cuda.device(x.device):
        y = torch.empty((N, 64, hp, wp), dtype=torch.float32, device=x.device)
        pk = hipops.empty_packed(N, 64, hp, wp, x.device)
        st = lib().bnn_hip_legacy_stem_staged(
            x.data_ptr(), w.data_ptr(), bn_scale.data_ptr(), bn_shift.data_ptr(), N, H, W,
            (native.STEM_FP16 if fp16 else 0) if flags is None else flags, y.data_ptr(), pk.P.data_ptr(),
            pk.M.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
    if flags is not None:
        return st
    native.check(st, "bnn_hip_legacy_stem_staged")
    pk.nonneg = True
    return y, pk


def bconv2d_lds(act, pw, stride=1, padding=0):
    """3x3 binary convolution with the weight tile staged in LDS -> fp32 NCHW (alpha applied, no bias)."""
    d = hipops._desc(act.shape, pw.shape, stride, padding, 1, 0)
    ho, wo = hipops.conv_out_hw(d.H, d.W, d.KH, d.KW, stride, padding, 1)
    dev = act.P.device
    with torch.cuda.device(dev):
        out = torch.empty((d.N, d.O, ho, wo), dtype=torch.float32, device=dev)
        native.check(lib().bnn_hip_legacy_bconv2d_lds(
            ctypes.byref(d), act.P.data_ptr(), act.M.data_ptr(), pw.wbits.data_ptr(), pw.alpha.data_ptr(), None, None,
            out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "bnn_hip_legacy_bconv2d_lds")
    return out
