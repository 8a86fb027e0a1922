"""Tensor-level wrappers over the C-ABI (``include/bnn_hip.h``).

torch is used here for what the tier calls plumbing: device memory (caching allocator), the
current HIP stream and device guards.  All arithmetic happens inside ``libbnn_hip.so``.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import native

_MAX_ELEMS = (1 << 31) - 1


def _pair(v) -> Tuple[int, int]:
    if isinstance(v, int):
        return v, v
    if len(v) == 1:
        return int(v[0]), int(v[0])
    return int(v[0]), int(v[1])


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_cuda_f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise native.NativeError(f"bnn_amd: {what} must live on a HIP device, got {t.device}")
    if t.dtype != torch.float32:
        raise native.NativeError(f"bnn_amd: {what} must be float32, got {t.dtype}")
    return t.contiguous()


@dataclass
class PackedAct:
    """sign(x) as bit planes: ``P``/``M`` int64 ``[N,H,W,cw64]``, ``nzc`` int16 ``[N,H,W]``."""
    P: torch.Tensor
    M: torch.Tensor
    nzc: torch.Tensor
    shape: Tuple[int, int, int, int]  # logical (N, C, H, W)


@dataclass
class PackedWeight:
    """sign(W) in the kernel-facing layout + per-channel alpha (padded to ``o_pad``)."""
    wbits: torch.Tensor   # int32 [n_words]
    wnz: torch.Tensor     # int32 [n_words]
    alpha: torch.Tensor   # float32 [o_pad]
    has_zero: bool
    shape: Tuple[int, int, int, int]  # logical (O, C, KH, KW)


def pack_act(x: torch.Tensor) -> PackedAct:
    """``BasicInputBinarizer`` on device: fp32 NCHW -> bit planes (bnn/ops.py:151-152)."""
    x = _require_cuda_f32(x, "activation")
    if x.dim() != 4:
        raise native.NativeError(f"bnn_amd: pack_act expects NCHW, got shape {tuple(x.shape)}")
    lib = native.require()
    N, C, H, W = x.shape
    cw64 = (C + 63) // 64
    with torch.cuda.device(x.device):
        P = torch.empty((N, H, W, cw64), dtype=torch.int64, device=x.device)
        M = torch.empty((N, H, W, cw64), dtype=torch.int64, device=x.device)
        nzc = torch.empty((N, H, W), dtype=torch.int16, device=x.device)
        native.check(lib.bnn_hip_pack_act_f32(x.data_ptr(), N, C, H, W, P.data_ptr(), M.data_ptr(),
                                              nzc.data_ptr(), _stream(x.device)),
                     "bnn_hip_pack_act_f32")
    return PackedAct(P, M, nzc, (N, C, H, W))


def pack_weight(w: torch.Tensor, center: bool = False, compute_alpha: bool = True) -> PackedWeight:
    """``XNORWeightBinarizer`` on device (bnn/ops.py:116-140).  Synchronises once to read the
    zero-weight flag — call it when the weight changes, not per forward."""
    w = _require_cuda_f32(w.detach(), "weight")
    if w.dim() == 2:
        w = w[:, :, None, None]
    elif w.dim() == 3:
        w = w[:, :, None, :]
    if w.dim() != 4:
        raise native.NativeError(f"bnn_amd: unsupported weight rank {w.dim()}")
    w = w.contiguous()
    lib = native.require()
    O, C, KH, KW = w.shape
    L = native.weight_layout(O, C, KH, KW)
    with torch.cuda.device(w.device):
        wbits = torch.empty(L.n_words, dtype=torch.int32, device=w.device)
        wnz = torch.empty(L.n_words, dtype=torch.int32, device=w.device)
        alpha = torch.empty(L.o_pad, dtype=torch.float32, device=w.device)
        flag = torch.zeros(1, dtype=torch.int32, device=w.device)
        native.check(lib.bnn_hip_pack_weight_f32(w.data_ptr(), O, C, KH, KW, int(center),
                                                 int(compute_alpha), wbits.data_ptr(),
                                                 wnz.data_ptr(), alpha.data_ptr(), flag.data_ptr(),
                                                 _stream(w.device)),
                     "bnn_hip_pack_weight_f32")
        has_zero = bool(flag.item())
    return PackedWeight(wbits, wnz, alpha, has_zero, (O, C, KH, KW))


def conv_out_hw(H, W, KH, KW, stride, padding, dilation) -> Tuple[int, int]:
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    return ((H + 2 * ph - dh * (KH - 1) - 1) // sh + 1, (W + 2 * pw - dw * (KW - 1) - 1) // sw + 1)


def _desc(act_shape, w_shape, stride, padding, dilation, flags) -> native.ConvDesc:
    N, C, H, W = act_shape
    O, C2, KH, KW = w_shape
    if C != C2:
        raise native.NativeError(f"bnn_amd: channel mismatch {C} vs {C2} (groups != 1 unsupported)")
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    return native.ConvDesc(N, C, H, W, O, KH, KW, sh, sw, ph, pw, dh, dw, flags)


def bconv2d(a: PackedAct, w: PackedWeight, bias: Optional[torch.Tensor] = None,
            post_scale: Optional[torch.Tensor] = None, stride=1, padding=0, dilation=1,
            force_generic: bool = False, raw_dot: bool = False) -> torch.Tensor:
    """Binary convolution on packed operands -> fp32 NCHW (or int32 dot when ``raw_dot``)."""
    lib = native.require()
    flags = (native.FLAG_FORCE_GENERIC if force_generic else 0) | \
            (native.FLAG_WEIGHT_ZEROS if w.has_zero else 0)
    d = _desc(a.shape, w.shape, stride, padding, dilation, flags)
    ho, wo = conv_out_hw(d.H, d.W, d.KH, d.KW, stride, padding, dilation)
    dev = a.P.device
    if bias is not None:
        bias = _require_cuda_f32(bias.detach(), "bias")
    if post_scale is not None:
        post_scale = _require_cuda_f32(post_scale.detach(), "post_scale").reshape(-1)
        if post_scale.numel() != d.O:
            raise native.NativeError("bnn_amd: post_scale must have one entry per output channel")
    with torch.cuda.device(dev):
        out = torch.empty((d.N, d.O, ho, wo), dtype=torch.int32 if raw_dot else torch.float32,
                          device=dev)
        # one launch addresses < 2^31 elements: split the batch when a tensor is larger
        per_img = max(d.O * ho * wo, d.H * d.W)
        step = max(1, min(d.N, _MAX_ELEMS // max(per_img, 1)))
        for n0 in range(0, d.N, step):
            n1 = min(d.N, n0 + step)
            dd = native.ConvDesc.from_buffer_copy(d)
            dd.N = n1 - n0
            args = (ctypes.byref(dd), a.P[n0:n1].data_ptr(), a.M[n0:n1].data_ptr(),
                    a.nzc[n0:n1].data_ptr(), w.wbits.data_ptr(), w.wnz.data_ptr())
            if raw_dot:
                st = lib.bnn_hip_bconv2d_dot(*args, out[n0:n1].data_ptr(), _stream(dev))
            else:
                st = lib.bnn_hip_bconv2d(*args, w.alpha.data_ptr(), _ptr(bias), _ptr(post_scale),
                                         out[n0:n1].data_ptr(), _stream(dev))
            native.check(st, "bnn_hip_bconv2d")
    return out


def probe_int_alu(iters: int = 4096, device: Optional[torch.device] = None) -> dict:
    """Sustained v_bitop3+v_bcnt lane-ops/s of the current device (roofline calibration)."""
    lib = native.require()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    rate = ctypes.c_double()
    ms = ctypes.c_double()
    with torch.cuda.device(dev):
        native.check(lib.bnn_hip_probe_int_alu(iters, ctypes.byref(rate), ctypes.byref(ms),
                                               _stream(dev)), "bnn_hip_probe_int_alu")
    return {"lane_ops_per_s": rate.value, "elapsed_ms": ms.value}
