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

_MAX_ELEMS = (1 << 30) - 1  # fp32 elements one conv launch addresses (32-bit byte offsets inside the kernels)
# the tiled conv kernels multiply indices with 24-bit multiplies (csrc/bconv.hip: small_indices()); a launch whose
# images x channels factor reaches 2^23 would silently take the slow shape-generic kernel — split the batch first
_MAX_INDEX_FACTOR = (1 << 23) - 1
_DIRECT_MIN_PIXELS = 16           # images up to this many pixels may take pack_act + bconv2d (_prefers_two_launches)
_MAX_DESC_BYTES = 0xFFFFFE00      # tensors addressed through a sized 32-bit buffer descriptor (capi.hip: kMaxDescBytes)


def _batch_step(n: int, per_img_elems: int, chan_factor: int) -> int:
    """Images per launch: as many as keep every tensor of the launch below ``_MAX_ELEMS`` elements and
    ``images * chan_factor`` (channels of the widest fp32 tensor, 64-channel groups of the planes) below the
    tiled kernels' index range."""
    step = min(n, _MAX_ELEMS // max(per_img_elems, 1), _MAX_INDEX_FACTOR // max(chan_factor, 1))
    return max(1, step)


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


def _per_channel(t: Optional[torch.Tensor], n: int, what: str) -> Optional[torch.Tensor]:
    if t is None:
        return None
    t = _require_cuda_f32(t.detach(), what).reshape(-1)
    if t.numel() != n:
        raise native.NativeError(f"bnn_amd: {what} must have one entry per output channel ({n})")
    return t


@dataclass
class PackedAct:
    """sign(x) as bit planes: ``P``/``M`` int64 ``[N,cw64,H,W]`` (format: include/bnn_hip.h)."""
    P: torch.Tensor
    M: torch.Tensor
    shape: Tuple[int, int, int, int]  # logical (N, C, H, W)
    # True when the producer guarantees M == 0 everywhere (values out of a ReLU are {0,+1}): the
    # 3x3 kernels then keep only the P plane in registers (BNN_HIP_FLAG_ACT_NONNEG).
    nonneg: bool = False

    def batch_slice(self, n0: int, n1: int) -> "PackedAct":
        return PackedAct(self.P[n0:n1], self.M[n0:n1], (n1 - n0,) + tuple(self.shape[1:]), self.nonneg)


@dataclass
class PackedWeight:
    """sign(W) in the kernel-facing layout + per-channel alpha (padded to ``o_pad``)."""
    wbits: torch.Tensor   # int32 [n_words]
    wnz: torch.Tensor     # int32 [n_words]
    alpha: torch.Tensor   # float32 [o_pad]
    has_zero: bool
    shape: Tuple[int, int, int, int]  # logical (O, C, KH, KW)
    # deferred zero check (pack_weight(..., sync=False)): pinned host copy of the flag + the event after it
    zero_probe: Optional[tuple] = None

    def zero_found_later(self) -> bool:
        """True if a pack made with ``sync=False`` turned out to contain a zero weight (waits for the copy)."""
        if self.zero_probe is None:
            return False
        host, ev = self.zero_probe
        ev.synchronize()
        return bool(host.item())


def empty_packed(N: int, C: int, H: int, W: int, device) -> PackedAct:
    cw64 = (C + 63) // 64
    return PackedAct(torch.empty((N, cw64, H, W), dtype=torch.int64, device=device),
                     torch.empty((N, cw64, H, W), dtype=torch.int64, device=device), (N, C, H, W))


def pack_act(x: torch.Tensor) -> PackedAct:
    """``BasicInputBinarizer`` on device: fp32 (or fp16) NCHW -> bit planes (bnn/ops.py:151-152)."""
    if x.dtype == torch.float16:
        if not x.is_cuda:
            raise native.NativeError(f"bnn_amd: activation must live on a HIP device, got {x.device}")
        x = x.contiguous()
    else:
        x = _require_cuda_f32(x, "activation")
    if x.dim() != 4:
        raise native.NativeError(f"bnn_amd: pack_act expects NCHW, got shape {tuple(x.shape)}")
    lib = native.require()
    N, C, H, W = x.shape
    if N == 0:  # empty batch: nothing to launch (the reference's conv2d returns an empty tensor too)
        return empty_packed(0, C, H, W, x.device)
    with torch.cuda.device(x.device):
        a = empty_packed(N, C, H, W, x.device)
        fn = lib.bnn_hip_pack_act_f16 if x.dtype == torch.float16 else lib.bnn_hip_pack_act_f32
        native.check(fn(x.data_ptr(), N, C, H, W, a.P.data_ptr(), a.M.data_ptr(), _stream(x.device)),
                     "bnn_hip_pack_act")
    return a


def bn_act_pack(x: torch.Tensor, bn_scale=None, bn_shift=None, relu: bool = False) -> PackedAct:
    """``sign(act(bn(x)))`` of an fp32 NCHW tensor in one pass: the input binarisation of a pre-activation
    block (bnn/models/layers/res_block.py:148, hierarchical_block.py:39)."""
    x = _require_cuda_f32(x, "activation")
    if x.dim() != 4:
        raise native.NativeError(f"bnn_amd: bn_act_pack expects NCHW, got shape {tuple(x.shape)}")
    lib = native.require()
    N, C, H, W = x.shape
    bn_scale = _per_channel(bn_scale, C, "bn_scale")
    bn_shift = _per_channel(bn_shift, C, "bn_shift")
    if N == 0:
        a = empty_packed(0, C, H, W, x.device)
        a.nonneg = bool(relu)
        return a
    with torch.cuda.device(x.device):
        a = empty_packed(N, C, H, W, x.device)
        native.check(lib.bnn_hip_bn_act_pack_f32(x.data_ptr(), N, C, H, W, _ptr(bn_scale), _ptr(bn_shift),
                                                 int(bool(relu)), a.P.data_ptr(), a.M.data_ptr(),
                                                 _stream(x.device)), "bnn_hip_bn_act_pack_f32")
    a.nonneg = bool(relu)
    return a


def avgpool_pack(x: torch.Tensor, k: int, nonneg: bool = False) -> PackedAct:
    """``AvgPool2d(k, k, ceil_mode=True, count_include_pad=False)`` + sign, fused
    (shortcut branch of bnn/models/resnet.py:128-133).  ``nonneg``: the caller knows ``x >= 0``
    (it is a ReLU output), so the averages are too."""
    x = _require_cuda_f32(x, "activation")
    lib = native.require()
    N, C, H, W = x.shape
    ho, wo = (H + k - 1) // k, (W + k - 1) // k
    with torch.cuda.device(x.device):
        a = empty_packed(N, C, ho, wo, x.device)
        native.check(lib.bnn_hip_avgpool_pack_f32(x.data_ptr(), N, C, H, W, k, a.P.data_ptr(),
                                                  a.M.data_ptr(), _stream(x.device)),
                     "bnn_hip_avgpool_pack_f32")
    a.nonneg = bool(nonneg)
    return a


def orpool_packed(a: PackedAct, k: int) -> PackedAct:
    """``sign(AvgPool2d(k, k, ceil_mode=True, count_include_pad=False)(x))`` from the sign planes of a
    NON-NEGATIVE ``x`` (``a.nonneg``): the OR of the P plane over each window (include/bnn_hip.h)."""
    if not a.nonneg:
        raise native.NativeError("bnn_amd: orpool_packed needs planes of a non-negative tensor (nonneg=True)")
    lib = native.require()
    N, C, H, W = a.shape
    ho, wo = (H + k - 1) // k, (W + k - 1) // k
    dev = a.P.device
    with torch.cuda.device(dev):
        out = empty_packed(N, C, ho, wo, dev)
        if N:
            native.check(lib.bnn_hip_orpool_packed(a.P.data_ptr(), N, C, H, W, k, out.P.data_ptr(), out.M.data_ptr(),
                                                   _stream(dev)), "bnn_hip_orpool_packed")
    out.nonneg = True
    return out


def stem7x7(x: torch.Tensor, w: torch.Tensor, bn_scale: torch.Tensor, bn_shift: torch.Tensor,
            out_f32: bool = True, out_packed: bool = True, exact_fp32: bool = False, fp16: bool = False,
            out: Optional[tuple] = None, pack_affine: Optional[tuple] = None):
    """conv 7x7/2/3 (3->64, no bias) -> folded BN -> ReLU -> MaxPool 3/2/1 in one MFMA kernel
    (bnn/models/resnet.py:93-96,150-153).  Returns (fp32 NCHW | None, PackedAct | None).
    ``out = (y, PackedAct)``: write into these preallocated results of an earlier call with the same input shape
    (the static buffers a captured HIP graph of the rest of the network reads: ``FusedResNet.forward_fresh``).
    ``pack_affine = (scale, shift)``: the planes are those of ``fmaf(y, scale[c], shift[c]) > 0`` — the input of a
    pre-activation block's first binary layer (its bn1 + ReLU) — instead of ``y > 0``; the fp32 output is unchanged."""
    x = _require_cuda_f32(x, "stem input")
    w = _require_cuda_f32(w.detach(), "stem weight")
    if x.dim() != 4 or x.shape[1] != 3 or tuple(w.shape) != (64, 3, 7, 7):
        raise native.NativeError("bnn_amd: stem7x7 expects x [N,3,H,W] and w [64,3,7,7]")
    lib = native.require()
    N, _, H, W = x.shape
    hc, wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    hp, wp = (hc - 1) // 2 + 1, (wc - 1) // 2 + 1
    bn_scale = _per_channel(bn_scale, 64, "bn_scale")
    bn_shift = _per_channel(bn_shift, 64, "bn_shift")
    with torch.cuda.device(x.device):
        if out is not None:
            y, pk = out
            if (y is not None and (tuple(y.shape) != (N, 64, hp, wp) or y.dtype != torch.float32 or y.device != x.device
                                   or not y.is_contiguous())) or \
                    (pk is not None and (tuple(pk.shape) != (N, 64, hp, wp) or pk.P.device != x.device)):
                raise native.NativeError("bnn_amd: stem7x7: `out` does not belong to this input shape / device")
        else:
            y = torch.empty((N, 64, hp, wp), dtype=torch.float32, device=x.device) if out_f32 else None
            pk = empty_packed(N, 64, hp, wp, x.device) if out_packed else None
        # the kernel addresses x and out through 32-bit buffer descriptors: a launch stays below 2^32 - 512 bytes per
        # tensor (include/bnn_hip.h) — larger batches go in several launches
        step = max(1, min(N, _MAX_DESC_BYTES // max(12 * H * W, 256 * hp * wp)))
        flags = native.STEM_EXACT_FP32 if exact_fp32 else (native.STEM_FP16 if fp16 else 0)
        if pack_affine is not None:
            if exact_fp32 or pk is None:
                raise native.NativeError("bnn_amd: stem7x7(pack_affine=...) needs packed output and is not built for the exact-fp32 stem")
            pa, pb = _per_channel(pack_affine[0], 64, "pack scale"), _per_channel(pack_affine[1], 64, "pack shift")
        for n0 in range(0, N, step):
            n1 = min(N, n0 + step)
            if pack_affine is not None:
                native.check(lib.bnn_hip_stem7x7_bn_relu_pool_pack_affine_f32(
                    x[n0:n1].data_ptr(), w.data_ptr(), bn_scale.data_ptr(), bn_shift.data_ptr(), pa.data_ptr(), pb.data_ptr(),
                    n1 - n0, H, W, flags, None if y is None else y[n0:n1].data_ptr(), pk.P[n0:n1].data_ptr(),
                    pk.M[n0:n1].data_ptr(), _stream(x.device)), "bnn_hip_stem7x7_bn_relu_pool_pack_affine_f32")
                continue
            native.check(lib.bnn_hip_stem7x7_bn_relu_pool_pack_f32(
                x[n0:n1].data_ptr(), w.data_ptr(), bn_scale.data_ptr(), bn_shift.data_ptr(), n1 - n0, H, W, flags,
                None if y is None else y[n0:n1].data_ptr(),
                None if pk is None else pk.P[n0:n1].data_ptr(), None if pk is None else pk.M[n0:n1].data_ptr(),
                _stream(x.device)), "bnn_hip_stem7x7_bn_relu_pool_pack_f32")
    if pk is not None:
        pk.nonneg = True  # ReLU output
    return y, pk


def stem7x7_conv(x: torch.Tensor, w: torch.Tensor, fp16: bool = False) -> torch.Tensor:
    """The stem's convolution alone — conv 7x7/2/3, 3 -> 64, no bias (bnn/models/resnet.py:93,150) — on the matrix cores
    (``bnn_hip_stem7x7_conv_f32``: the MFMA stream of ``stem7x7``, fp32-class accuracy): what a TRAINING step needs in
    front of its batch-statistics BatchNorm.  Returns fp32 ``[N, 64, Hc, Wc]``."""
    x = _require_cuda_f32(x, "stem input")
    w = _require_cuda_f32(w.detach(), "stem weight")
    if x.dim() != 4 or x.shape[1] != 3 or tuple(w.shape) != (64, 3, 7, 7):
        raise native.NativeError("bnn_amd: stem7x7_conv expects x [N,3,H,W] and w [64,3,7,7]")
    lib = native.require()
    N, _, H, W = x.shape
    hc, wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    with torch.cuda.device(x.device):
        y = torch.empty((N, 64, hc, wc), dtype=torch.float32, device=x.device)
        step = max(1, min(N, _MAX_DESC_BYTES // max(12 * H * W, 256 * hc * wc)))   # 32-bit descriptors (see stem7x7)
        for n0 in range(0, N, step):
            n1 = min(N, n0 + step)
            native.check(lib.bnn_hip_stem7x7_conv_f32(x[n0:n1].data_ptr(), w.data_ptr(), n1 - n0, H, W,
                                                      native.STEM_FP16 if fp16 else 0, y[n0:n1].data_ptr(),
                                                      _stream(x.device)), "bnn_hip_stem7x7_conv_f32")
    return y


def stem7x7_wgrad_supported(x: torch.Tensor) -> bool:
    """``stem7x7_wgrad`` implements this input size (image rows up to ~950 pixels: the kernel's LDS patch)."""
    lib = native.require()
    return x.dim() == 4 and x.shape[1] == 3 and lib.bnn_hip_stem7x7_wgrad_workspace_bytes(
        int(x.shape[0]), int(x.shape[2]), int(x.shape[3])) > 0


def stem7x7_wgrad(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """Weight gradient of the stem's convolution (``bnn_hip_stem7x7_wgrad_f32``; what
    ``aten::convolution_backward(dy, x, w, ...)[1]`` returns for conv 7x7/2/3, 3 -> 64): fp32 ``[64, 3, 7, 7]``.
    fp32 products on the matrix cores, partial sums added in index order — the same bits on every run."""
    x = _require_cuda_f32(x, "stem input")
    dy = _require_cuda_f32(dy, "stem output gradient")
    if x.dim() != 4 or x.shape[1] != 3:
        raise native.NativeError(f"bnn_amd: stem7x7_wgrad expects x [N,3,H,W], got {tuple(x.shape)}")
    N, _, H, W = x.shape
    hc, wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if tuple(dy.shape) != (N, 64, hc, wc):
        raise native.NativeError("bnn_amd: stem7x7_wgrad expects x [N,3,H,W] and dy [N,64,Hc,Wc]")
    lib = native.require()
    with torch.cuda.device(x.device):
        need = lib.bnn_hip_stem7x7_wgrad_workspace_bytes(N, H, W)
        if need == 0:
            raise native.NativeError(f"bnn_amd: stem7x7_wgrad does not implement input size {tuple(x.shape)}")
        work = torch.empty(need // 4, dtype=torch.float32, device=x.device)
        dw = torch.empty((64, 3, 7, 7), dtype=torch.float32, device=x.device)
        native.check(lib.bnn_hip_stem7x7_wgrad_f32(x.data_ptr(), dy.data_ptr(), N, H, W, work.data_ptr(), need,
                                                   dw.data_ptr(), _stream(x.device)), "bnn_hip_stem7x7_wgrad_f32")
    return dw


def avgpool2x2_backward(gy: torch.Tensor) -> torch.Tensor:
    """Backward of ``AvgPool2d(2, 2)`` on an even-sized map: ``gx[..., 2y + a, 2x + b] = gy[..., y, x] / 4``
    (``bnn_hip_avgpool2x2_backward_f32``)."""
    gy = _require_cuda_f32(gy, "pooled gradient")
    if gy.dim() != 4:
        raise native.NativeError("bnn_amd: avgpool2x2_backward expects [N,C,Ho,Wo]")
    lib = native.require()
    N, C, Ho, Wo = gy.shape
    with torch.cuda.device(gy.device):
        gx = torch.empty((N, C, 2 * Ho, 2 * Wo), dtype=torch.float32, device=gy.device)
        native.check(lib.bnn_hip_avgpool2x2_backward_f32(gy.data_ptr(), N, C, Ho, Wo, gx.data_ptr(), _stream(gy.device)),
                     "bnn_hip_avgpool2x2_backward_f32")
    return gx


def avgpool2_bn_pack2(x: torch.Tensor, bn1, relu1: bool, bn2=None, relu2: bool = False, out_f32: bool = False):
    """``AvgPool2d(2, 2)`` of an fp32 NCHW tensor (even H, W) + ``sign(act(bn(.)))`` of the pooled tensor for up to two
    BatchNorm branches in one pass (``bnn_hip_avgpool2_bn_pack2_f32``).  ``bn1`` / ``bn2``: (scale, shift).
    Returns ``(PackedAct 1, PackedAct 2 | None, pooled fp32 | None)``."""
    x = _require_cuda_f32(x, "activation")
    lib = native.require()
    N, C, H, W = x.shape
    if H % 2 or W % 2:
        raise native.NativeError("bnn_amd: avgpool2_bn_pack2 needs even H and W")
    a1, b1 = _per_channel(bn1[0], C, "bn scale"), _per_channel(bn1[1], C, "bn shift")
    a2 = b2 = None
    if bn2 is not None:
        a2, b2 = _per_channel(bn2[0], C, "bn scale"), _per_channel(bn2[1], C, "bn shift")
    with torch.cuda.device(x.device):
        p1 = empty_packed(N, C, H // 2, W // 2, x.device)
        p2 = empty_packed(N, C, H // 2, W // 2, x.device) if bn2 is not None else None
        t = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=x.device) if out_f32 else None
        if N:
            native.check(lib.bnn_hip_avgpool2_bn_pack2_f32(
                x.data_ptr(), N, C, H, W, a1.data_ptr(), b1.data_ptr(), int(bool(relu1)), p1.P.data_ptr(), p1.M.data_ptr(),
                _ptr(a2), _ptr(b2), int(bool(relu2)), None if p2 is None else p2.P.data_ptr(),
                None if p2 is None else p2.M.data_ptr(), _ptr(t), _stream(x.device)), "bnn_hip_avgpool2_bn_pack2_f32")
    p1.nonneg = bool(relu1)
    if p2 is not None:
        p2.nonneg = bool(relu2)
    return p1, p2, t


def sign_thresholds(w: PackedWeight, bn_scale: torch.Tensor, bn_shift: torch.Tensor, bias=None, post_scale=None):
    """Integer form of ``sign(relu(bn(alpha * dot + bias)))`` for ``bconv2d_fused(..., sign_thresholds=...)``:
    int32 ``[O, 4]`` = (bound T, flip word of the channel's 32-channel block, the two comparands of the kernels'
    two-instruction form of the test), ``bit = (dot >= T) ^ flip``, derived on
    the device with the epilogue's own float operations
    (include/bnn_hip.h: bnn_hip_sign_thresholds_f32).  Valid for THIS weight pack and THESE BatchNorm constants."""
    lib = native.require()
    O, C, KH, KW = w.shape
    dev = w.alpha.device
    bn_scale = _per_channel(bn_scale, O, "bn_scale")
    bn_shift = _per_channel(bn_shift, O, "bn_shift")
    bias = _per_channel(bias, O, "bias")
    post_scale = _per_channel(post_scale, O, "post_scale")
    with torch.cuda.device(dev):
        thr = torch.empty(((O + 31) // 32 * 32, 4), dtype=torch.int32, device=dev)   # whole 32-channel blocks
        native.check(lib.bnn_hip_sign_thresholds_f32(w.alpha.data_ptr(), _ptr(bias), _ptr(post_scale), bn_scale.data_ptr(),
                                                     bn_shift.data_ptr(), O, C * KH * KW, thr.data_ptr(), _stream(dev)),
                     "bnn_hip_sign_thresholds_f32")
    return thr


def avgpool_fc(x: torch.Tensor, w_t: torch.Tensor, bias: Optional[torch.Tensor], one_kernel: bool = False) -> torch.Tensor:
    """``fc(flatten(avgpool(x)))`` of bnn/models/resnet.py:160-164.  ``x``: fp32 ``[N,C,H,W]``,
    ``w_t``: the Linear weight transposed to ``[C,O]`` (contiguous), ``bias``: ``[O]`` or None.
    Two streaming launches through a workspace (``bnn_hip_avgpool_fc_ws_f32``); ``one_kernel``: the round-2 single
    kernel (``bnn_hip_avgpool_fc_f32``: same means, another summation order of the product)."""
    x = _require_cuda_f32(x, "head input")
    w_t = _require_cuda_f32(w_t.detach(), "head weight")
    if x.dim() != 4 or w_t.dim() != 2 or w_t.shape[0] != x.shape[1]:
        raise native.NativeError("bnn_amd: avgpool_fc expects x [N,C,H,W] and w_t [C,O]")
    lib = native.require()
    N, C, H, W = x.shape
    O = w_t.shape[1]
    bias = _per_channel(bias, O, "head bias")
    with torch.cuda.device(x.device):
        out = torch.empty((N, O), dtype=torch.float32, device=x.device)
        if N and one_kernel:
            native.check(lib.bnn_hip_avgpool_fc_f32(x.data_ptr(), N, C, H * W, w_t.data_ptr(), _ptr(bias), O,
                                                    out.data_ptr(), _stream(x.device)), "bnn_hip_avgpool_fc_f32")
        elif N:
            nbytes = int(lib.bnn_hip_avgpool_fc_workspace_bytes(N, C))
            ws = torch.empty((max(nbytes, 16) + 3) // 4, dtype=torch.float32, device=x.device)
            native.check(lib.bnn_hip_avgpool_fc_ws_f32(x.data_ptr(), N, C, H * W, w_t.data_ptr(), _ptr(bias), O,
                                                       out.data_ptr(), ws.data_ptr(), nbytes, _stream(x.device)),
                         "bnn_hip_avgpool_fc_ws_f32")
    return out


def bn_relu_maxpool_pack(x: torch.Tensor, bn_scale=None, bn_shift=None, relu: bool = True, k: int = 3,
                         stride: int = 2, pad: int = 1, out_f32: bool = True, out_packed: bool = True):
    """Stem tail in one pass: folded BN -> MaxPool2d(k, stride, pad) -> ReLU, written as fp32
    and/or sign planes (bnn/models/resnet.py:150-153 + the first binary conv's binarizer)."""
    x = _require_cuda_f32(x, "activation")
    lib = native.require()
    N, C, H, W = x.shape
    ho, wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    bn_scale = _per_channel(bn_scale, C, "bn_scale")
    bn_shift = _per_channel(bn_shift, C, "bn_shift")
    with torch.cuda.device(x.device):
        y = torch.empty((N, C, ho, wo), dtype=torch.float32, device=x.device) if out_f32 else None
        pk = empty_packed(N, C, ho, wo, x.device) if out_packed else None
        native.check(lib.bnn_hip_bn_relu_maxpool_pack_f32(
            x.data_ptr(), N, C, H, W, _ptr(bn_scale), _ptr(bn_shift), int(bool(relu)), k, stride, pad,
            _ptr(y), None if pk is None else pk.P.data_ptr(), None if pk is None else pk.M.data_ptr(),
            _stream(x.device)), "bnn_hip_bn_relu_maxpool_pack_f32")
    if pk is not None:
        pk.nonneg = bool(relu)
    return y, pk


def pack_weight(w: torch.Tensor, center: bool = False, compute_alpha: bool = True,
                sync: bool = True) -> PackedWeight:
    """``XNORWeightBinarizer`` on device (bnn/ops.py:116-140).  Synchronises once to read the
    zero-weight flag — call it when the weight changes, not per forward.

    ``sync=False`` (training: the weight changes every step) assumes "no exact zero" and returns at once;
    the flag travels to pinned host memory asynchronously and ``PackedWeight.zero_found_later()`` tells
    afterwards whether the assumption held (the caller then re-packs with ``sync=True``)."""
    half = w.dtype == torch.float16
    if half:   # a `.half()` model: the sign bits are those of the exact fp32 widening; alpha is rounded to fp16 below
        w = w.detach().float()
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
        if half:  # the reference computes alpha in the weight's own dtype (bnn/ops.py:116-127)
            alpha = alpha.half().float()
        if not sync:
            host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(w.device))
            return PackedWeight(wbits, wnz, alpha, False, (O, C, KH, KW), (host, ev))
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


def _flags(w: PackedWeight, force_generic: bool, weights: Optional[str], a: Optional[PackedAct] = None,
           throughput: bool = False) -> int:
    f = (native.FLAG_FORCE_GENERIC if force_generic else 0) | \
        (native.FLAG_WEIGHT_ZEROS if w.has_zero else 0) | (native.FLAG_THROUGHPUT if throughput else 0)
    if a is not None and a.nonneg:
        f |= native.FLAG_ACT_NONNEG
    if weights == "sgpr":
        f |= native.FLAG_WEIGHTS_SGPR
    elif weights is not None:
        raise ValueError("weights must be None or 'sgpr' (the LDS-staged tile is test-only since ABI 12: "
                         "tests/helpers/legacy.py)")
    return f


def bconv2d(a: PackedAct, w: PackedWeight, bias: Optional[torch.Tensor] = None,
            post_scale: Optional[torch.Tensor] = None, stride=1, padding=0, dilation=1,
            force_generic: bool = False, raw_dot: bool = False,
            weights: Optional[str] = None) -> torch.Tensor:
    """Binary convolution on packed operands -> fp32 NCHW (or int32 dot when ``raw_dot``)."""
    lib = native.require()
    d = _desc(a.shape, w.shape, stride, padding, dilation, _flags(w, force_generic, weights, a))
    ho, wo = conv_out_hw(d.H, d.W, d.KH, d.KW, stride, padding, dilation)
    dev = a.P.device
    bias = _per_channel(bias, d.O, "bias")
    post_scale = _per_channel(post_scale, d.O, "post_scale")
    with torch.cuda.device(dev):
        out = torch.empty((d.N, d.O, ho, wo), dtype=torch.int32 if raw_dot else torch.float32,
                          device=dev)
        if d.N == 0:
            return out
        # one launch addresses < 2^31 elements: split the batch when a tensor is larger
        # (the planes are limited to 2^29 uint64 words per launch, half the fp32 element budget: capi.hip check_desc)
        per_img = max(d.O * ho * wo, 2 * d.H * d.W * ((d.C + 63) // 64), 2 * ho * wo * ((d.O + 63) // 64))
        step = _batch_step(d.N, per_img, max(d.O, (d.C + 63) // 64))
        for n0 in range(0, d.N, step):
            n1 = min(d.N, n0 + step)
            dd = native.ConvDesc.from_buffer_copy(d)
            dd.N = n1 - n0
            args = (ctypes.byref(dd), a.P[n0:n1].data_ptr(), a.M[n0:n1].data_ptr(),
                    w.wbits.data_ptr(), w.wnz.data_ptr())
            if raw_dot:
                st = lib.bnn_hip_bconv2d_dot(*args, out[n0:n1].data_ptr(), _stream(dev))
            else:
                st = lib.bnn_hip_bconv2d(*args, w.alpha.data_ptr(), _ptr(bias), _ptr(post_scale),
                                         out[n0:n1].data_ptr(), _stream(dev))
            native.check(st, "bnn_hip_bconv2d")
    return out


def direct_plan(x_shape, w: PackedWeight, stride=1, padding=0, dilation=1) -> Optional[native.FlyPlan]:
    """The band plan ``bconv2d_direct`` would use for this geometry, or None where the one-launch layer does not
    apply (include/bnn_hip.h: bnn_hip_bconv2d_direct_plan)."""
    lib = native.require()
    d = _desc(tuple(x_shape), w.shape, stride, padding, dilation, native.FLAG_WEIGHT_ZEROS if w.has_zero else 0)
    plan = native.FlyPlan()
    st = lib.bnn_hip_bconv2d_direct_plan(ctypes.byref(d), ctypes.byref(plan))
    if st == -2:      # BNN_HIP_ERR_UNSUPPORTED
        return None
    native.check(st, "bnn_hip_bconv2d_direct_plan")
    return plan


def _prefers_two_launches(d) -> bool:
    """Shapes where ``pack_act`` + ``bconv2d`` beats the one-launch kernel.  That kernel packs 64 consecutive pixels of
    a band per item: with 1 x 1 images (``Linear`` layers) an item holds ONE valid lane and its loads stride by a
    whole image — 1/64 lane utilisation.  Measured on MI355X (tools/bench_linear.py, us one-launch vs
    two launches): 1x1 images N=1024 C=512 49 vs 30, N=256 C=4096 450 vs 48, N=64 C=2048 69 vs 29, N=32768 C=256 105 vs
    85; 3x3 on 2x2 / 4x4 images 76 vs 30 / 77 vs 37 — but N=256 C=512 1x1 26 vs 31, 1x1 conv on 2x2 images 18 vs 29,
    3x3 on 7x7 30 vs 33: tiny images go the two-launch way when the kernel is larger than 1x1 or the input is wide or
    the batch large."""
    return d.H * d.W <= _DIRECT_MIN_PIXELS and (d.KH * d.KW > 1 or d.C >= 1024 or d.N * d.C >= (1 << 19))


def bconv2d_direct(x: torch.Tensor, w: PackedWeight, bias: Optional[torch.Tensor] = None,
                   post_scale: Optional[torch.Tensor] = None, stride=1, padding=0, dilation=1,
                   plan: Optional[native.FlyPlan] = None, force_generic: bool = False,
                   route: Optional[str] = None) -> torch.Tensor:
    """``Conv2d.forward`` of the reference (bnn/layers/conv.py:90-97) in ONE launch: fp32 or fp16 NCHW activations in,
    fp32 NCHW out, ``sign(x)`` computed on the fly inside the convolution kernel (csrc/bconv_fly.hip) — no packed
    copy of the activations in HBM.  Geometries the one-launch kernel does not cover (an output row with its halo
    larger than a CU's LDS) take ``pack_act`` + ``bconv2d``; the results are bit-identical either way.

    ``route``: ``None`` picks per shape (``_prefers_two_launches``); ``"direct"`` / ``"packed"`` force one (tools)."""
    if route not in (None, "direct", "packed"):
        raise ValueError("route must be None, 'direct' or 'packed'")
    if x.dtype == torch.float16:
        if not x.is_cuda:
            raise native.NativeError(f"bnn_amd: activation must live on a HIP device, got {x.device}")
        x = x.contiguous()
    else:
        x = _require_cuda_f32(x, "activation")
    if x.dim() != 4:
        raise native.NativeError(f"bnn_amd: bconv2d_direct expects NCHW, got shape {tuple(x.shape)}")
    lib = native.require()
    d = _desc(tuple(x.shape), w.shape, stride, padding, dilation, _flags(w, force_generic, None))
    ho, wo = conv_out_hw(d.H, d.W, d.KH, d.KW, stride, padding, dilation)
    dev = x.device
    bias = _per_channel(bias, d.O, "bias")
    post_scale = _per_channel(post_scale, d.O, "post_scale")
    if d.N and plan is None and (route == "packed" or (route is None and _prefers_two_launches(d))):
        return bconv2d(pack_act(x), w, bias, post_scale, stride, padding, dilation, force_generic=force_generic)
    with torch.cuda.device(dev):
        out = torch.empty((d.N, d.O, ho, wo), dtype=torch.float32, device=dev)
        if d.N == 0:
            return out
        per_img = max(d.O * ho * wo, d.C * d.H * d.W, 2 * ho * wo * ((d.O + 63) // 64))
        step = _batch_step(d.N, per_img, max(d.O, (d.C + 63) // 64))
        dtype = native.DTYPE_F16 if x.dtype == torch.float16 else native.DTYPE_F32
        for n0 in range(0, d.N, step):
            n1 = min(d.N, n0 + step)
            dd = native.ConvDesc.from_buffer_copy(d)
            dd.N = n1 - n0
            st = lib.bnn_hip_bconv2d_direct(ctypes.byref(dd), x[n0:n1].data_ptr(), dtype, w.wbits.data_ptr(),
                                            w.wnz.data_ptr(), w.alpha.data_ptr(), _ptr(bias), _ptr(post_scale),
                                            out[n0:n1].data_ptr(), None if plan is None else ctypes.byref(plan),
                                            _stream(dev))
            if st == -2 and plan is None:   # BNN_HIP_ERR_UNSUPPORTED: the two-launch form of the same layer
                out[n0:n1] = bconv2d(pack_act(x[n0:n1]), w, bias, post_scale, stride, padding, dilation,
                                     force_generic=force_generic)
                continue
            native.check(st, "bnn_hip_bconv2d_direct")
    return out


def bconv2d_fused(a: PackedAct, w: PackedWeight, *, bias=None, post_scale=None, bn_scale=None,
                  bn_shift=None, residual=None, prelu=None, relu=False, out_f32=True,
                  out_packed=False, stride=1, padding=0, dilation=1, force_generic=False,
                  weights: Optional[str] = None, residual_after_act: bool = False,
                  pack_before_residual: bool = False, pack_scale=None, pack_shift=None,
                  pack_relu: bool = False, out: Optional[torch.Tensor] = None, out_c_offset: int = 0,
                  throughput: bool = False, sign_thresholds: Optional[torch.Tensor] = None,
                  shortcut: Optional[tuple] = None):
    """Binary convolution + fused epilogue (see ``bnn_hip_epilogue``): returns
    ``(y_fp32 | None, PackedAct(sign(p)) | None)``.

    ``shortcut = (PackedAct, PackedWeight, bn_scale, bn_shift)``: the down-sampling block's shortcut branch folded into
    this launch (``bnn_hip_epilogue.sc_*``): the residual is ``BN(conv1x1(packed))`` computed in the kernel instead of
    an fp32 tensor; see ``shortcut_fold_supported``.

    ``out``: write the fp32 result into channels ``[out_c_offset, out_c_offset + O)`` of this
    preallocated ``[N, C_total, Ho, Wo]`` tensor (``torch.cat`` in place); ``residual`` then has
    ``C_total`` channels too.  The remaining keyword arguments are the pre-activation switches of
    ``BNN_HIP_EPI_*`` (include/bnn_hip.h); ``throughput``: ``BNN_HIP_FLAG_THROUGHPUT`` (several batches in flight)."""
    lib = native.require()
    d = _desc(a.shape, w.shape, stride, padding, dilation, _flags(w, force_generic, weights, a, throughput))
    ho, wo = conv_out_hw(d.H, d.W, d.KH, d.KW, stride, padding, dilation)
    dev = a.P.device
    bias = _per_channel(bias, d.O, "bias")
    post_scale = _per_channel(post_scale, d.O, "post_scale")
    bn_scale = _per_channel(bn_scale, d.O, "bn_scale")
    bn_shift = _per_channel(bn_shift, d.O, "bn_shift")
    prelu = _per_channel(prelu, d.O, "prelu")
    pack_scale = _per_channel(pack_scale, d.O, "pack_scale")
    pack_shift = _per_channel(pack_shift, d.O, "pack_shift")
    c_total = d.O
    if out is not None:
        if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.dim() == 4
                and out.shape[0] == d.N and tuple(out.shape[2:]) == (ho, wo)
                and 0 <= out_c_offset and out_c_offset + d.O <= out.shape[1]):
            raise native.NativeError("bnn_amd: `out` must be a contiguous fp32 [N, C_total, Ho, Wo] tensor "
                                     "with room for O channels at out_c_offset")
        c_total = out.shape[1]
    if residual is not None:
        residual = _require_cuda_f32(residual, "residual")
        if tuple(residual.shape) != (d.N, c_total, ho, wo):
            raise native.NativeError(f"bnn_amd: residual shape {tuple(residual.shape)} != output")
    sc_in_hw = 0
    if shortcut is not None:
        sa, sw, sbn_a, sbn_b = shortcut
        # the shortcut planes at the output resolution, or UN-POOLED at twice it (the kernel ORs the 2 x 2 windows)
        pooled = tuple(sa.shape) == (d.N, sw.shape[1], ho, wo)
        unpooled = (sa.shape[0] == d.N and sa.shape[1] == sw.shape[1] and (sa.shape[2] + 1) // 2 == ho
                    and (sa.shape[3] + 1) // 2 == wo and max(sa.shape[2], sa.shape[3]) < 65536)
        if residual is not None or not sa.nonneg or not (pooled or unpooled) \
                or tuple(sw.shape) != (d.O, sa.shape[1], 1, 1) or sw.has_zero:
            raise native.NativeError("bnn_amd: folded shortcut: a non-negative packed [N, C, Ho, Wo] (or un-pooled "
                                     "[N, C, ~2 Ho, ~2 Wo]) input, a 1x1 [O, C, 1, 1] weight without zeros, and no "
                                     "separate residual")
        if not pooled:
            sc_in_hw = (sa.shape[2] << 16) | sa.shape[3]
        sbn_a = _per_channel(sbn_a, d.O, "shortcut bn_scale")
        sbn_b = _per_channel(sbn_b, d.O, "shortcut bn_shift")
    eflags = (native.EPI_RES_AFTER_ACT if residual_after_act else 0) | \
        (native.EPI_PACK_BEFORE_RES if pack_before_residual else 0) | (native.EPI_PACK_RELU if pack_relu else 0)
    with torch.cuda.device(dev):
        y = out if out is not None else (
            torch.empty((d.N, d.O, ho, wo), dtype=torch.float32, device=dev) if out_f32 else None)
        pk = empty_packed(d.N, d.O, ho, wo, dev) if out_packed else None
        # one launch addresses < 2^31 elements: split the batch when a tensor is larger (like bconv2d)
        step = fused_launch_images(d.N, d.C, d.H, d.W, d.O, (d.KH, d.KW), stride, padding, dilation, c_total)
        for n0 in range(0, d.N, step):
            n1 = min(d.N, n0 + step)
            dd = native.ConvDesc.from_buffer_copy(d)
            dd.N = n1 - n0
            e = native.Epilogue(w.alpha.data_ptr(), _ptr(bias), _ptr(post_scale), _ptr(bn_scale),
                                _ptr(bn_shift), None if residual is None else residual[n0:n1].data_ptr(),
                                _ptr(prelu), int(bool(relu)), eflags,
                                None if y is None else y[n0:n1].data_ptr(),
                                None if pk is None else pk.P[n0:n1].data_ptr(),
                                None if pk is None else pk.M[n0:n1].data_ptr(), _ptr(pack_scale), _ptr(pack_shift),
                                out_c_offset if out is not None else 0, c_total if out is not None else 0,
                                _ptr(sign_thresholds),
                                *((sa.P[n0:n1].data_ptr(), sw.wbits.data_ptr(), sw.alpha.data_ptr(), sbn_a.data_ptr(),
                                   sbn_b.data_ptr(), sa.shape[1], sc_in_hw) if shortcut is not None else ()))
            native.check(lib.bnn_hip_bconv2d_fused(ctypes.byref(dd), a.P[n0:n1].data_ptr(), a.M[n0:n1].data_ptr(),
                                                   w.wbits.data_ptr(), w.wnz.data_ptr(),
                                                   ctypes.byref(e), _stream(dev)),
                         "bnn_hip_bconv2d_fused")
    if pk is not None:  # mirrors the kernel's rule for leaving the M plane at zero
        late = residual is not None and residual_after_act
        pk.nonneg = bool(pack_relu) or (bool(relu) and prelu is None and pack_scale is None
                                        and (not late or pack_before_residual))
    return y, pk


@dataclass
class HBlockPack:
    """Derived data of one hierarchical block for ``hblock_forward`` (include/bnn_hip.h: bnn_hip_hblock_forward): the
    three weight packs in the kernel's dense layout and ONE constants buffer (alphas, folded bn2 / bn3, the next
    block's folded bn1)."""
    weights: torch.Tensor     # int32 [weight_words]
    consts: torch.Tensor      # float32 [const_floats]
    c_in: int
    planes: int
    has_next: bool
    packs: tuple = ()                              # the three standard packs (kept for the other weight order)
    weights_cl: Optional[torch.Tensor] = None      # the same weights in the channel-lane kernel's order (made on first use)

    def channel_lane_weights(self) -> torch.Tensor:
        if self.weights_cl is None:
            lib = native.require()
            w1, w2, w3 = self.packs
            dev = self.weights.device
            with torch.cuda.device(dev):
                buf = torch.empty_like(self.weights)
                native.check(lib.bnn_hip_hblock_pack_weights_cl(self.c_in, self.planes, w1.wbits.data_ptr(), w2.wbits.data_ptr(),
                                                                w3.wbits.data_ptr(), buf.data_ptr(), _stream(dev)),
                             "bnn_hip_hblock_pack_weights_cl")
            self.weights_cl = buf
        return self.weights_cl


def hblock_pack(w1: PackedWeight, w2: PackedWeight, w3: PackedWeight, bn2, bn3, next_bn=None) -> HBlockPack:
    """``w1..w3``: standard packs of the block's three 3x3 convolutions (no zero weights); ``bn2`` / ``bn3`` /
    ``next_bn``: (scale, shift) of the folded BatchNorms in front of conv2 / conv3 / the NEXT block's conv1."""
    lib = native.require()
    planes, c_in = 2 * w1.shape[0], w1.shape[1]
    if (w1.has_zero or w2.has_zero or w3.has_zero or tuple(w1.shape[2:]) != (3, 3)
            or tuple(w2.shape) != (planes // 4, planes // 2, 3, 3) or tuple(w3.shape) != (planes // 4, planes // 4, 3, 3)):
        raise native.NativeError("bnn_amd: hblock_pack expects the three 3x3 packs of an HBlock without zero weights")
    L = native.HBlockLayout()
    native.check(lib.bnn_hip_hblock_layout_of(c_in, planes, ctypes.byref(L)), "bnn_hip_hblock_layout_of")
    dev = w1.wbits.device
    with torch.cuda.device(dev):
        wbuf = torch.empty(int(L.weight_words), dtype=torch.int32, device=dev)
        native.check(lib.bnn_hip_hblock_pack_weights(c_in, planes, w1.wbits.data_ptr(), w2.wbits.data_ptr(),
                                                     w3.wbits.data_ptr(), wbuf.data_ptr(), _stream(dev)),
                     "bnn_hip_hblock_pack_weights")
        consts = torch.zeros(int(L.const_floats), dtype=torch.float32, device=dev)
        for k, w in enumerate((w1, w2, w3)):
            consts[L.alpha_off[k]:L.alpha_off[k] + w.shape[0]] = w.alpha[:w.shape[0]]
        for k, bn in enumerate((bn2, bn3)):
            n = (planes // 2, planes // 4)[k]
            consts[L.pack_a_off[k]:L.pack_a_off[k] + n] = _per_channel(bn[0], n, "bn scale")
            consts[L.pack_b_off[k]:L.pack_b_off[k] + n] = _per_channel(bn[1], n, "bn shift")
        if next_bn is not None:
            consts[L.next_a_off:L.next_a_off + planes] = _per_channel(next_bn[0], planes, "next bn scale")
            consts[L.next_b_off:L.next_b_off + planes] = _per_channel(next_bn[1], planes, "next bn shift")
    return HBlockPack(wbuf, consts, c_in, planes, next_bn is not None, (w1, w2, w3))


def _hblock_desc(N, c_in, H, W, planes, throughput=False, rows_per_band=0, images_per_band=0, waves=0, channel_lanes=False):
    flags = (native.FLAG_THROUGHPUT if throughput else 0) | (native.HBLOCK_CHANNEL_LANES if channel_lanes else 0)
    return native.HBlockDesc(N, c_in, H, W, planes, flags, rows_per_band, images_per_band, waves, 0)


def hblock_supported(N: int, c_in: int, H: int, W: int, planes: int, throughput: bool = False, rows_per_band: int = 0,
                     images_per_band: int = 0, waves: int = 0, channel_lanes: bool = False) -> bool:
    """Whether ``hblock_forward`` covers this geometry (with this plan) on the current device.  ``channel_lanes``: the
    small-image form of the kernel (csrc/hblock_cl.hip: 14 x 14 / 7 x 7, lanes = output channels)."""
    lib = native.require()
    d = _hblock_desc(N, c_in, H, W, planes, throughput, rows_per_band, images_per_band, waves, channel_lanes)
    return bool(lib.bnn_hip_hblock_supported(ctypes.byref(d)))


def hblock_forward(a: PackedAct, pack: HBlockPack, residual: torch.Tensor, out_packed: bool = True,
                   throughput: bool = False, rows_per_band: int = 0, images_per_band: int = 0, waves: int = 0,
                   channel_lanes: bool = False):
    """``HBlock.forward`` behind its first BatchNorm + ReLU (bnn/models/layers/hierarchical_block.py:38-60) in ONE
    launch: ``a`` = planes of ``sign(relu(bn1(x)))`` (non-negative), ``residual`` = the shortcut (fp32 NCHW).
    Returns ``(y, PackedAct(sign(relu(next_bn(y)))) | None)``."""
    lib = native.require()
    N, c_in, H, W = a.shape
    if not a.nonneg or c_in != pack.c_in:
        raise native.NativeError("bnn_amd: hblock_forward needs non-negative input planes of the block's width")
    residual = _require_cuda_f32(residual, "residual")
    if tuple(residual.shape) != (N, pack.planes, H, W):
        raise native.NativeError(f"bnn_amd: residual shape {tuple(residual.shape)} != block output")
    if out_packed and not pack.has_next:
        raise native.NativeError("bnn_amd: this HBlockPack holds no next-block BatchNorm")
    dev = a.P.device
    with torch.cuda.device(dev):
        y = torch.empty_like(residual)
        pk = None
        if out_packed:
            pk = PackedAct(torch.empty((N, pack.planes // 64, H, W), dtype=torch.int64, device=dev),
                           _zero_plane((N, pack.planes // 64, H, W), dev), (N, pack.planes, H, W), nonneg=True)
        if N == 0:
            return y, pk
        d = _hblock_desc(N, c_in, H, W, pack.planes, throughput, rows_per_band, images_per_band, waves, channel_lanes)
        wbuf = pack.channel_lane_weights() if channel_lanes else pack.weights
        native.check(lib.bnn_hip_hblock_forward(ctypes.byref(d), a.P.data_ptr(), wbuf.data_ptr(),
                                                pack.consts.data_ptr(), residual.data_ptr(), y.data_ptr(),
                                                None if pk is None else pk.P.data_ptr(), _stream(dev)),
                     "bnn_hip_hblock_forward")
    return y, pk


def hblock_shortcut_supported(N: int, c_in: int, H: int, W: int, planes: int, throughput: bool = False, rows_per_band: int = 0,
                              images_per_band: int = 0, waves: int = 0, channel_lanes: bool = False) -> bool:
    """Whether ``hblock_shortcut_forward`` covers this geometry (with this plan) on the current device."""
    lib = native.require()
    d = _hblock_desc(N, c_in, H, W, planes, throughput, rows_per_band, images_per_band, waves, channel_lanes)
    return bool(lib.bnn_hip_hblock_shortcut_supported(ctypes.byref(d)))


def hblock_shortcut_pack(w: PackedWeight):
    """``(weights [planes][C_in / 32] int32, alpha [planes])`` of a block's 1x1 shortcut convolution for
    ``hblock_shortcut_forward`` (no zero weights)."""
    lib = native.require()
    planes, c_in = w.shape[0], w.shape[1]
    if w.has_zero or tuple(w.shape[2:]) != (1, 1) or c_in % 64:
        raise native.NativeError("bnn_amd: hblock_shortcut_pack expects the pack of a 1x1 convolution without zero weights")
    dev = w.wbits.device
    with torch.cuda.device(dev):
        buf = torch.empty(planes * (c_in // 32), dtype=torch.int32, device=dev)
        native.check(lib.bnn_hip_hblock_pack_shortcut_weights(c_in, planes, w.wbits.data_ptr(), buf.data_ptr(), _stream(dev)),
                     "bnn_hip_hblock_pack_shortcut_weights")
        alpha = w.alpha[:planes].contiguous().clone()
    return buf, alpha


def hblock_shortcut_forward(a: PackedAct, pack: HBlockPack, sc: PackedAct, sc_pack, throughput: bool = False,
                            rows_per_band: int = 0, images_per_band: int = 0, waves: int = 0, channel_lanes: bool = False):
    """The first hierarchical block of a stage with its shortcut (``BatchNorm -> sign -> conv1x1``,
    hierarchical_block.py:30-36) computed inside the launch from the sign planes ``sc`` of that binarisation:
    no shortcut launch, no fp32 shortcut tensor.  Returns ``(y, PackedAct of the next block's input)``."""
    lib = native.require()
    N, c_in, H, W = a.shape
    if not a.nonneg or c_in != pack.c_in or tuple(sc.shape) != tuple(a.shape) or not pack.has_next:
        raise native.NativeError("bnn_amd: hblock_shortcut_forward needs the block's input planes, the shortcut's planes of "
                                 "the same shape, and a pack with the next block's BatchNorm")
    wsc, asc = sc_pack
    if wsc.numel() != pack.planes * (c_in // 32) or asc.numel() != pack.planes:
        raise native.NativeError("bnn_amd: shortcut pack of another width")
    dev = a.P.device
    with torch.cuda.device(dev):
        y = torch.empty((N, pack.planes, H, W), dtype=torch.float32, device=dev)
        shp = (N, pack.planes // 64, H, W)
        pk = PackedAct(torch.empty(shp, dtype=torch.int64, device=dev), _zero_plane(shp, dev), (N, pack.planes, H, W),
                       nonneg=True)
        if N == 0:
            return y, pk
        d = _hblock_desc(N, c_in, H, W, pack.planes, throughput, rows_per_band, images_per_band, waves, channel_lanes)
        wbuf = pack.channel_lane_weights() if channel_lanes else pack.weights
        native.check(lib.bnn_hip_hblock_shortcut_forward(ctypes.byref(d), a.P.data_ptr(), wbuf.data_ptr(),
                                                         pack.consts.data_ptr(), sc.P.data_ptr(), sc.M.data_ptr(),
                                                         wsc.data_ptr(), asc.data_ptr(), y.data_ptr(), pk.P.data_ptr(),
                                                         _stream(dev)), "bnn_hip_hblock_shortcut_forward")
    return y, pk


def hblock_pool_supported(N: int, c_in: int, H: int, W: int, planes: int, throughput: bool = False, rows_per_band: int = 0,
                          images_per_band: int = 0, waves: int = 0) -> bool:
    """Whether ``hblock_pool_forward`` covers this geometry (with this plan) on the current device."""
    lib = native.require()
    d = _hblock_desc(N, c_in, H, W, planes, throughput, rows_per_band, images_per_band, waves)
    return bool(lib.bnn_hip_hblock_pool_supported(ctypes.byref(d)))


def hblock_pool_consts(bn1, bn_ds, planes: int) -> torch.Tensor:
    """The constants of ``hblock_pool_forward`` (include/bnn_hip.h: ``[a1/4 | b1 | a2/4 | b2 | -a2/4 | -b2 | 0 | 0]``) from
    the folded bn1 of the next stage's first block and the folded BatchNorm of its shortcut (hierarchical_block.py:30-39).
    The average's division by 4 moves into the scales; raises when that is not exact (a scale next to the smallest normal)."""
    a1, b1, a2, b2 = (_per_channel(t, planes, "pool bn") for t in (bn1[0], bn1[1], bn_ds[0], bn_ds[1]))
    q1, q2 = a1 * 0.25, a2 * 0.25
    if not (bool((q1 * 4.0 == a1).all()) and bool((q2 * 4.0 == a2).all())):
        raise native.NativeError("bnn_amd: hblock_pool_consts: a BatchNorm scale too small to be divided by 4 exactly")
    z = torch.zeros_like(a1)
    return torch.cat((q1, b1, q2, b2, -q2, -b2, z, z)).contiguous()


def hblock_pool_forward(a: PackedAct, pack: HBlockPack, residual: torch.Tensor, pool_consts: torch.Tensor,
                        throughput: bool = False, rows_per_band: int = 0, images_per_band: int = 0, waves: int = 0):
    """The LAST hierarchical block of a stage + ``AvgPool2d(2, 2)`` + the two binarisations of the pooled tensor that the
    next stage's first block performs (its bn1 -> ReLU -> sign; its shortcut's BatchNorm -> sign), ONE launch, no fp32
    output (bnn/models/resnet.py: ``nn.Sequential(AvgPool2d(2, 2), HBlock(...))`` stages).  Returns
    ``(PackedAct of sign(relu(bn1(t))), PackedAct of sign(bn_ds(t)))`` with ``t = AvgPool2d(2, 2)(HBlock(x))``."""
    lib = native.require()
    N, c_in, H, W = a.shape
    if not a.nonneg or c_in != pack.c_in or c_in != pack.planes:
        raise native.NativeError("bnn_amd: hblock_pool_forward needs non-negative input planes of a width-preserving block")
    residual = _require_cuda_f32(residual, "residual")
    if tuple(residual.shape) != (N, pack.planes, H, W) or H % 2 or W % 2:
        raise native.NativeError(f"bnn_amd: residual shape {tuple(residual.shape)} != block output (even height and width)")
    if (pool_consts.dtype != torch.float32 or pool_consts.numel() != 8 * pack.planes or not pool_consts.is_contiguous()
            or pool_consts.device != a.P.device):
        raise native.NativeError("bnn_amd: pool_consts must be the buffer of hblock_pool_consts")
    dev = a.P.device
    with torch.cuda.device(dev):
        shp = (N, pack.planes // 64, H // 2, W // 2)
        p1 = PackedAct(torch.empty(shp, dtype=torch.int64, device=dev), _zero_plane(shp, dev),
                       (N, pack.planes, H // 2, W // 2), nonneg=True)
        p2 = PackedAct(torch.empty(shp, dtype=torch.int64, device=dev), torch.empty(shp, dtype=torch.int64, device=dev),
                       (N, pack.planes, H // 2, W // 2))
        if N == 0:
            return p1, p2
        d = _hblock_desc(N, c_in, H, W, pack.planes, throughput, rows_per_band, images_per_band, waves)
        native.check(lib.bnn_hip_hblock_pool_forward(ctypes.byref(d), a.P.data_ptr(), pack.weights.data_ptr(),
                                                     pack.consts.data_ptr(), pool_consts.data_ptr(), residual.data_ptr(),
                                                     p1.P.data_ptr(), p2.P.data_ptr(), p2.M.data_ptr(), _stream(dev)),
                     "bnn_hip_hblock_pool_forward")
    return p1, p2


_ZERO_PLANES = {}


def _zero_plane(shape, dev) -> torch.Tensor:
    """An all-zero M plane (read-only by convention: planes of non-negative activations), shared per shape and device."""
    key = (tuple(shape), dev.index)
    z = _ZERO_PLANES.get(key)
    if z is None:
        z = _ZERO_PLANES[key] = torch.zeros(shape, dtype=torch.int64, device=dev)
    return z


def fused_launch_images(N: int, C: int, H: int, W: int, O: int, kernel_size, stride=1, padding=0, dilation=1,
                        c_total: Optional[int] = None) -> int:
    """Images ONE launch of ``bconv2d_fused`` covers for this geometry (a batch whose tensors exceed the kernels'
    32-bit addressing is split into several launches): what shape-dependent decisions have to be taken on."""
    kh, kw = _pair(kernel_size)
    ho, wo = conv_out_hw(H, W, kh, kw, stride, padding, dilation)
    c_total = O if c_total is None else c_total
    per_img = max(c_total * ho * wo, 2 * H * W * ((C + 63) // 64), 2 * ho * wo * ((O + 63) // 64), 1)
    return _batch_step(N, per_img, max(c_total, O, (C + 63) // 64))


def shortcut_fold_supported(a: PackedAct, w: PackedWeight, sc_channels: int, stride=1, padding=0, dilation=1,
                            throughput: bool = False) -> bool:
    """Whether ``bconv2d_fused(a, w, ..., shortcut=...)`` exists for this convolution and a shortcut 1x1 convolution of
    ``sc_channels`` input channels (``bnn_hip_shortcut_fold_supported``)."""
    lib = native.require()
    d = _desc(a.shape, w.shape, stride, padding, dilation, _flags(w, False, None, a, throughput))
    return bool(lib.bnn_hip_shortcut_fold_supported(ctypes.byref(d), int(sc_channels)))


def grad_supported(x_shape, w_shape, stride, padding, dilation) -> bool:
    """Shapes the binary-aware gradient kernels cover: 3x3 / stride 1 or 2 / padding 1 and 1x1 / stride 1 / padding 0,
    dilation 1, width <= 64."""
    k = tuple(w_shape[2:])
    if _pair(dilation) != (1, 1) or x_shape[3] > 64 or x_shape[0] <= 0:
        return False
    if k == (3, 3):
        return _pair(stride) in ((1, 1), (2, 2)) and _pair(padding) == (1, 1)
    return k == (1, 1) and _pair(stride) == (1, 1) and _pair(padding) == (0, 0)


def grad_pack_weight(w_hat: torch.Tensor):
    """``What = sign(Wc) * alpha`` ([O,C,k,k] fp32, k = 3 or 1) -> (sign fragments for the input-gradient kernel,
    alpha[O])."""
    w_hat = _require_cuda_f32(w_hat.detach(), "w_hat")
    lib = native.require()
    O, C, k = w_hat.shape[0], w_hat.shape[1], w_hat.shape[2]
    with torch.cuda.device(w_hat.device):
        packed = torch.empty(int(lib.bnn_hip_grad_weight_pack_bytes(O, C, k)), dtype=torch.uint8, device=w_hat.device)
        alpha = torch.empty(O, dtype=torch.float32, device=w_hat.device)
        native.check(lib.bnn_hip_grad_pack_weight_f32(w_hat.data_ptr(), O, C, k, packed.data_ptr(), alpha.data_ptr(),
                                                      _stream(w_hat.device)), "bnn_hip_grad_pack_weight_f32")
    return packed, alpha


def xnor_grad_pack_weight(w: torch.Tensor, center: bool, compute_alpha: bool):
    """``grad_pack_weight(xnor_what(w, center, compute_alpha))`` in one launch (``bnn_hip_xnor_grad_pack_weight_f32``): the
    sign fragments and ``alpha[O]`` the input-gradient kernel reads, straight from the raw weight — the same bytes."""
    w = _require_cuda_f32(w.detach(), "weight")
    lib = native.require()
    O, C, k = w.shape[0], w.shape[1], w.shape[2]
    with torch.cuda.device(w.device):
        packed = torch.empty(int(lib.bnn_hip_grad_weight_pack_bytes(O, C, k)), dtype=torch.uint8, device=w.device)
        alpha = torch.empty(O, dtype=torch.float32, device=w.device)
        native.check(lib.bnn_hip_xnor_grad_pack_weight_f32(w.data_ptr(), O, C, k, int(center), int(compute_alpha),
                                                           packed.data_ptr(), alpha.data_ptr(), _stream(w.device)),
                     "bnn_hip_xnor_grad_pack_weight_f32")
    return packed, alpha


@dataclass
class SavedAct:
    """What the backward of a binary convolution needs from its fp32 input, in 3 bits per element: the sign planes
    (``sign``: P = x > 0, M = x < 0) and the straight-through mask ``T`` = |x| < 1, int64 ``[N, ceil(C/64), H, W]`` each
    (csrc/pack_ste.hip).  The reference's autograd keeps the fp32 ``x`` and an fp32 ``sign(x)`` instead."""
    sign: "PackedAct"
    T: torch.Tensor
    shape: Tuple[int, int, int, int]

    def nbytes(self) -> int:
        return 3 * self.T.numel() * 8


def pack_act_ste(x: torch.Tensor) -> SavedAct:
    """``bnn_hip_pack_act_ste_f32``: one pass over ``x`` -> the three bit planes of ``SavedAct``."""
    x = _require_cuda_f32(x, "activation")
    if x.dim() != 4:
        raise native.NativeError(f"bnn_amd: pack_act_ste expects NCHW, got shape {tuple(x.shape)}")
    lib = native.require()
    N, C, H, W = x.shape
    with torch.cuda.device(x.device):
        a = empty_packed(N, C, H, W, x.device)
        T = torch.empty_like(a.P)
        if N:
            native.check(lib.bnn_hip_pack_act_ste_f32(x.data_ptr(), N, C, H, W, a.P.data_ptr(), a.M.data_ptr(),
                                                      T.data_ptr(), _stream(x.device)), "bnn_hip_pack_act_ste_f32")
    return SavedAct(a, T, (N, C, H, W))


def _grad_shapes(g: torch.Tensor, x, stride: int):
    N, C, H, W = x.shape
    O = g.shape[1]
    if tuple(g.shape) != (N, O, (H - 1) // stride + 1, (W - 1) // stride + 1):
        raise native.NativeError(f"bnn_amd: grad_output {tuple(g.shape)} does not belong to input {tuple(x.shape)} "
                                 f"at stride {stride}")
    return N, O, C, H, W


def bconv_grad_input(g: torch.Tensor, x, packed: torch.Tensor, alpha: torch.Tensor,
                     ksize: int = 3, stride: int = 1) -> torch.Tensor:
    """dL/dx of the binary 3x3/p1 or 1x1/p0 conv incl. the hard-tanh STE mask (bnn/ops.py:68-73).  ``x``: the fp32
    input, or its ``SavedAct`` (the mask plane is all this kernel needs of it) — same bits either way."""
    g = _require_cuda_f32(g, "grad_output")
    lib = native.require()
    if isinstance(x, SavedAct):
        N, O, C, H, W = _grad_shapes(g, x, stride)
        with torch.cuda.device(g.device):
            gx = torch.empty((N, C, H, W), dtype=torch.float32, device=g.device)
            native.check(lib.bnn_hip_bconv_grad_input_packed_f32(
                g.data_ptr(), alpha.data_ptr(), packed.data_ptr(), x.T.data_ptr(), gx.data_ptr(), N, O, C, H, W, ksize,
                stride, _stream(g.device)), "bnn_hip_bconv_grad_input_packed_f32")
        return gx
    x = _require_cuda_f32(x, "input")
    N, O, C, H, W = _grad_shapes(g, x, stride)
    with torch.cuda.device(g.device):
        gx = torch.empty_like(x)
        native.check(lib.bnn_hip_bconv_grad_input_f32(g.data_ptr(), alpha.data_ptr(), packed.data_ptr(),
                                                      x.data_ptr(), gx.data_ptr(), N, O, C, H, W, ksize, stride,
                                                      _stream(g.device)), "bnn_hip_bconv_grad_input_f32")
    return gx


def bconv_grad_weight(g: torch.Tensor, x, ksize: int = 3, stride: int = 1, reduce: bool = True) -> torch.Tensor:
    """dL/dWhat [O,C,k,k] of the binary 3x3/p1 or 1x1/p0 conv: correlation of g with sign(x).  ``x``: the fp32 input, or
    its ``SavedAct`` (the sign planes are all this kernel needs of it) — same bits either way.  ``reduce=False``: the
    kernel's split-K partial slabs ``[splits, O, C, k, k]`` as they are (``xnor_weight_backward`` adds them itself)."""
    g = _require_cuda_f32(g, "grad_output")
    if not isinstance(x, SavedAct):
        x = _require_cuda_f32(x, "input")
    lib = native.require()
    N, O, C, H, W = _grad_shapes(g, x, stride)
    splits = int(lib.bnn_hip_bconv_grad_weight_splits(N, O, C, ksize))
    with torch.cuda.device(g.device):
        part = torch.empty((splits, O, C, ksize, ksize), dtype=torch.float32, device=g.device)
        if isinstance(x, SavedAct):
            native.check(lib.bnn_hip_bconv_grad_weight_packed_f32(
                g.data_ptr(), x.sign.P.data_ptr(), x.sign.M.data_ptr(), part.data_ptr(), splits, N, O, C, H, W, ksize,
                stride, _stream(g.device)), "bnn_hip_bconv_grad_weight_packed_f32")
        else:
            native.check(lib.bnn_hip_bconv_grad_weight_f32(g.data_ptr(), x.data_ptr(), part.data_ptr(), splits,
                                                           N, O, C, H, W, ksize, stride, _stream(g.device)),
                         "bnn_hip_bconv_grad_weight_f32")
        if not reduce:
            return part
        return part[0] if splits == 1 else part.sum(0)


def xnor_what(w: torch.Tensor, center: bool, compute_alpha: bool) -> torch.Tensor:
    """``XNORWeightBinarizer.forward`` value (bnn/ops.py:129-140) in one kernel: ``sign(Wc) * alpha`` as an fp32 tensor
    (no autograd graph).  Same centring / alpha reductions as ``pack_weight``."""
    w = _require_cuda_f32(w.detach(), "weight")
    lib = native.require()
    O, C = w.shape[0], w.shape[1]
    kh, kw = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
    with torch.cuda.device(w.device):
        what = torch.empty_like(w)
        native.check(lib.bnn_hip_xnor_weight_forward_f32(w.data_ptr(), O, C, kh, kw, int(center), int(compute_alpha),
                                                         what.data_ptr(), None, _stream(w.device)),
                     "bnn_hip_xnor_weight_forward_f32")
    return what


def xnor_weight_backward(w: torch.Tensor, dwhat: torch.Tensor, center: bool, compute_alpha: bool) -> torch.Tensor:
    """dL/dW from dL/dWhat through ``XNORWeightBinarizer`` (sign STE, alpha = mean|Wc|, centring) in one kernel.
    ``dwhat``: the gradient (``w``'s shape), or the split-K partial slabs ``[splits, *w.shape]`` of
    ``bconv_grad_weight(..., reduce=False)`` — added inside the kernel, in slab order."""
    w = _require_cuda_f32(w.detach(), "weight")
    dwhat = _require_cuda_f32(dwhat, "weight gradient")
    if dwhat.dim() == w.dim():
        dwhat = dwhat.unsqueeze(0)
    if tuple(dwhat.shape[1:]) != tuple(w.shape):
        raise native.NativeError(f"bnn_amd: weight gradient {tuple(dwhat.shape)} does not belong to weight {tuple(w.shape)}")
    lib = native.require()
    O, C = w.shape[0], w.shape[1]
    kh, kw = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
    with torch.cuda.device(w.device):
        dw = torch.empty_like(w)
        native.check(lib.bnn_hip_xnor_weight_backward_f32(w.data_ptr(), dwhat.data_ptr(), int(dwhat.shape[0]), O, C, kh, kw,
                                                          int(center), int(compute_alpha), dw.data_ptr(),
                                                          _stream(w.device)), "bnn_hip_xnor_weight_backward_f32")
    return dw


def bn_act(x: torch.Tensor, bn_scale: torch.Tensor, bn_shift: torch.Tensor, relu: bool = False,
           residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``relu?(fma(x, scale[c], shift[c]) (+ residual))``: eval-mode BatchNorm with folded constants + the residual add +
    ReLU of a residual block as ONE launch (include/bnn_hip.h: bnn_hip_bn_act_f32) — the tail of the per-layer path; the
    same float operations as the fused convolution epilogue."""
    x = _require_cuda_f32(x, "BatchNorm input")
    if x.dim() != 4:
        raise native.NativeError(f"bnn_amd: bn_act expects NCHW, got shape {tuple(x.shape)}")
    lib = native.require()
    N, C, H, W = x.shape
    bn_scale = _per_channel(bn_scale, C, "bn_scale")
    bn_shift = _per_channel(bn_shift, C, "bn_shift")
    if residual is not None:
        residual = _require_cuda_f32(residual, "residual")
        if residual.shape != x.shape:
            raise native.NativeError("bnn_amd: residual shape differs from the BatchNorm input's")
    with torch.cuda.device(x.device):
        y = torch.empty_like(x)
        if x.numel():
            native.check(lib.bnn_hip_bn_act_f32(x.data_ptr(), N, C, H * W, bn_scale.data_ptr(), bn_shift.data_ptr(),
                                                _ptr(residual), int(bool(relu)), y.data_ptr(), _stream(x.device)),
                         "bnn_hip_bn_act_f32")
    return y


def bn_train_forward(x: torch.Tensor, gamma, beta, running_mean, running_var, momentum: float, eps: float,
                     relu: bool = False, residual: Optional[torch.Tensor] = None):
    """``relu?(batch_norm(x, training=True) (+ residual))`` in three launches (csrc/bn_train.hip).  Updates the running
    statistics in place (unbiased variance, momentum) like ``torch.nn.BatchNorm2d`` in training mode.  Returns
    ``(y, save_mean, save_invstd)``."""
    x = _require_cuda_f32(x, "BatchNorm input")
    if x.dim() != 4:
        raise native.NativeError(f"bnn_amd: bn_train_forward expects NCHW, got shape {tuple(x.shape)}")
    lib = native.require()
    N, C, H, W = x.shape
    if residual is not None:
        residual = _require_cuda_f32(residual, "residual")
        if residual.shape != x.shape:
            raise native.NativeError("bnn_amd: residual shape differs from the BatchNorm input's")
    with torch.cuda.device(x.device):
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = torch.empty(int(lib.bnn_hip_bn_train_workspace_bytes(N, C, H * W)), dtype=torch.uint8, device=x.device)
        native.check(lib.bnn_hip_bn_train_forward_f32(
            x.data_ptr(), N, C, H * W, _ptr(gamma), _ptr(beta), _ptr(residual), int(bool(relu)), float(eps),
            float(momentum), _ptr(running_mean), _ptr(running_var), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
            ws.data_ptr(), _stream(x.device)), "bnn_hip_bn_train_forward_f32")
    return y, mean, invstd


def bn_train_backward(gy: torch.Tensor, y: Optional[torch.Tensor], x: torch.Tensor, mean: torch.Tensor,
                      invstd: torch.Tensor, gamma, want_dres: bool = False):
    """Backward of ``bn_train_forward``: ``(dx, dgamma, dbeta, dres | None)``.  ``y``: the forward's output when a ReLU
    was fused (its mask), else None."""
    gy = _require_cuda_f32(gy, "grad_output")
    lib = native.require()
    N, C, H, W = x.shape
    with torch.cuda.device(x.device):
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if want_dres else None
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = torch.empty(int(lib.bnn_hip_bn_train_workspace_bytes(N, C, H * W)), dtype=torch.uint8, device=x.device)
        native.check(lib.bnn_hip_bn_train_backward_f32(
            gy.data_ptr(), _ptr(y), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gamma), N, C, H * W,
            dx.data_ptr(), _ptr(dres), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), _stream(x.device)),
            "bnn_hip_bn_train_backward_f32")
    return dx, dgamma, dbeta, dres


def bn_relu_maxpool_train_forward(x: torch.Tensor, gamma, beta, running_mean, running_var, momentum: float, eps: float):
    """``maxpool3x3/2/1(relu(batch_norm(x, training=True)))`` without writing the normalised tensor (csrc/bn_train.hip):
    returns ``(pooled, code uint8, save_mean, save_invstd)``; ``code`` = which of the 9 window positions won."""
    x = _require_cuda_f32(x, "BatchNorm input")
    lib = native.require()
    N, C, H, W = x.shape
    hp, wp = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    with torch.cuda.device(x.device):
        p = torch.empty((N, C, hp, wp), dtype=torch.float32, device=x.device)
        code = torch.empty((N, C, hp, wp), dtype=torch.uint8, device=x.device)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = torch.empty(int(lib.bnn_hip_bn_train_workspace_bytes(N, C, H * W)), dtype=torch.uint8, device=x.device)
        native.check(lib.bnn_hip_bn_relu_maxpool_train_forward_f32(
            x.data_ptr(), N, C, H, W, _ptr(gamma), _ptr(beta), float(eps), float(momentum), _ptr(running_mean),
            _ptr(running_var), p.data_ptr(), code.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(),
            _stream(x.device)), "bnn_hip_bn_relu_maxpool_train_forward_f32")
    return p, code, mean, invstd


def bn_relu_maxpool_train_backward(gy: torch.Tensor, p: torch.Tensor, code: torch.Tensor, x: torch.Tensor,
                                   mean: torch.Tensor, invstd: torch.Tensor, gamma):
    """Backward of ``bn_relu_maxpool_train_forward``: ``(dx, dgamma, dbeta)``."""
    gy = _require_cuda_f32(gy, "grad_output")
    lib = native.require()
    N, C, H, W = x.shape
    with torch.cuda.device(x.device):
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = torch.empty(int(lib.bnn_hip_bn_train_workspace_bytes(N, C, H * W)), dtype=torch.uint8, device=x.device)
        native.check(lib.bnn_hip_bn_relu_maxpool_train_backward_f32(
            gy.data_ptr(), p.data_ptr(), code.data_ptr(), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gamma),
            N, C, H, W, dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), _stream(x.device)),
            "bnn_hip_bn_relu_maxpool_train_backward_f32")
    return dx, dgamma, dbeta


PROBE_MODES = {0: "bitop3+bcnt", 1: "xor+bcnt", 2: "bcnt", 3: "bitop3", 4: "xor", 5: "fma_f32",
               6: "add_u32", 7: "and_vgpr", 8: "and_vgpr+bcnt", 9: "xor_vgpr", 10: "and_sgpr+bcnt"}


def probe_clock(device: Optional[torch.device] = None, spin_iters: int = 10000) -> float:
    """Engine clock right now in MHz, sampled on the current stream of ``device`` (bnn_hip_probe_clock)."""
    lib = native.require()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    mhz = ctypes.c_double()
    us = ctypes.c_double()
    with torch.cuda.device(dev):
        native.check(lib.bnn_hip_probe_clock(spin_iters, ctypes.byref(mhz), ctypes.byref(us), _stream(dev)),
                     "bnn_hip_probe_clock")
    return mhz.value


def probe_int_alu(iters: int = 4096, device: Optional[torch.device] = None, mode: int = 0) -> dict:
    """Sustained lane-ops/s of a register-only instruction stream (roofline calibration)."""
    lib = native.require()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    rate = ctypes.c_double()
    ms = ctypes.c_double()
    with torch.cuda.device(dev):
        native.check(lib.bnn_hip_probe_int_alu(mode, iters, ctypes.byref(rate), ctypes.byref(ms),
                                               _stream(dev)), "bnn_hip_probe_int_alu")
    return {"lane_ops_per_s": rate.value, "elapsed_ms": ms.value}
