"""Training step of a binary ``Conv2d`` with the forward on the HIP XNOR/popcount path
(reference: ``bnn/ops.py:63-73`` straight-through estimator,
``bnn/layers/conv.py:90-97`` forward, ``examples/imagenet.py:337-384`` training loop).

    forward :  y = bconv2d( pack(sign(x)), pack(sign(Wc)), alpha ) [+ bias]        HIP XNOR/popcount kernels
    backward:  g_xhat[n,c] = sum_o,k g_y * W_hat      g_x = g_xhat * 1[|x| < 1]       STE of ops.py:68-73
               g_what[o,c] = sum_n,p g_y * sign(x)
               g_W through the weight hook's own autograd graph (sign STE + alpha = mean|W|)

Each gradient GEMM has ONE real operand (``g_y``) and one that is exactly ternary (``sign(Wc)``, ``sign(x)``).
For the 3x3 / padding 1 layers of stride 1 or 2 and the 1x1 / stride 1 (shortcut) layers — all 19 binary convs of a
ResNet-18 — both run on hand-written MFMA kernels (``csrc/grad.hip``: g split into THREE bf16 terms hi + mid + lo — 24
mantissa bits and the exponent range of fp32, so gradients of a mean-reduced loss at batch 256 (1e-5 .. 1e-9) keep
fp32 accuracy — the ternary side exact in bf16, three ``v_mfma_f32_16x16x32_bf16`` per product, fp32 accumulation), with
the STE mask fused into the input-gradient store; any other geometry uses ``aten::convolution_backward``.
``W_hat = weight_pre_process(W)`` is computed by the hook under autograd, which makes the weight gradient flow exactly
as in the reference composition; the forward kernel re-derives the packed form of the same weights on every training
forward (``fastpath.packed_weight(..., fresh=True)``).  The forward is ONE launch (``bnn_hip_bconv2d_direct``:
``sign(x)`` on the fly in LDS, no packed copy of the activations in HBM).  What it keeps for the backward is the fp32
``x`` by default, or — ``PACKED_STATE`` — THREE BITS per input element: the sign planes and the mask ``|x| < 1``
(``bnn_hip_pack_act_ste_f32``, one extra pass over x) instead of the fp32 ``x`` and fp32 ``sign(x)`` the reference's
autograd keeps alive; the gradient kernels read the planes (``bnn_hip_bconv_grad_*_packed_f32``) and return the same
bits as from the fp32 tensor.

Data-parallel training is ordinary ``DistributedDataParallel`` over RCCL (backend ``"nccl"``), one
process per GPU: the binary layers are ``nn.Module``s with ordinary fp32 Parameters, so gradient
bucketing / all-reduce needs nothing special (``make_ddp``).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import hipops

ENABLED = True  # set False to force the torch composition in training (tests compare the two)
BINARY_GRADS = True  # False: library fp32 gradient convolutions for every layer (tests / A-B timing)
# Keep 3 bits per input element for the backward (sign planes + STE mask) instead of the fp32 input: 10.7x less saved
# state per binary convolution, same gradients bit for bit, +8 % step time on ResNet-18 at batch 256 (32.0 vs 29.5 ms:
# the weight-gradient kernel's fill extracts bits and issues two plane loads per element).  Off by default (speed);
# BNN_AMD_TRAIN_PACKED_STATE=1 or `training.PACKED_STATE = True` turns it on (memory).
PACKED_STATE = os.environ.get("BNN_AMD_TRAIN_PACKED_STATE", "0") == "1"


_saved_bytes = 0     # bytes of layer INPUT state the binary convolutions have kept for their backward since the last reset


def saved_input_bytes(reset: bool = False) -> int:
    """Bytes of input state (fp32 ``x``, or its three bit planes) the training forwards of the binary convolutions have
    saved for the backward since the last reset — what ``PACKED_STATE`` shrinks (tests, tools/bench_train.py)."""
    global _saved_bytes
    n = _saved_bytes
    if reset:
        _saved_bytes = 0
    return n


class BinaryConv2dTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_hat, bias, layer, plan, packed):
        out = hipops.bconv2d_direct(x, packed, bias, None, layer.stride, layer.padding, layer.dilation)   # one launch
        stride, padding, dilation = tuple(layer.stride), tuple(layer.padding), tuple(layer.dilation)
        ctx.x_shape = None
        if PACKED_STATE and BINARY_GRADS and hipops.grad_supported(x.shape, w_hat.shape, stride, padding, dilation):
            # the gradient kernels need sign(x) and the mask |x| < 1: three bit planes (3/32 of the fp32 tensor the
            # reference's autograd keeps alive until the backward)
            sv = hipops.pack_act_ste(x)
            ctx.save_for_backward(sv.sign.P, sv.sign.M, sv.T, w_hat)
            ctx.x_shape = tuple(x.shape)
            kept = sv.nbytes()
        else:
            ctx.save_for_backward(x, w_hat)
            kept = x.numel() * x.element_size()
        global _saved_bytes
        _saved_bytes += kept
        ctx.conf = (stride, padding, dilation, bias is not None, None if bias is None else tuple(bias.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        stride, padding, dilation, has_bias, bias_shape = ctx.conf
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]
        if ctx.x_shape is not None:
            P, M, T, w_hat = ctx.saved_tensors
            x = hipops.SavedAct(hipops.PackedAct(P, M, ctx.x_shape), T, ctx.x_shape)
        else:
            x, w_hat = ctx.saved_tensors
        if ctx.x_shape is not None or (
                BINARY_GRADS and hipops.grad_supported(x.shape, w_hat.shape, stride, padding, dilation)):
            g = g.contiguous()
            gx = gw = gb = None
            if need_x:
                packed, alpha = hipops.grad_pack_weight(w_hat)
                gx = hipops.bconv_grad_input(g, x, packed, alpha, w_hat.shape[2], stride[0])   # STE mask fused
            if need_w:
                gw = hipops.bconv_grad_weight(g, x, w_hat.shape[2], stride[0])
            if need_b:
                gb = g.sum(dim=(0, 2, 3))
            return gx, gw, gb, None, None, None
        xh = torch.sign(x)
        gx, gw, gb = torch.ops.aten.convolution_backward(
            g.contiguous(), xh, w_hat, list(bias_shape) if has_bias else None, list(stride), list(padding),
            list(dilation), False, [0, 0], 1, [bool(need_x), bool(need_w), bool(need_b)])
        if need_x:
            gx = gx.masked_fill(x.abs() >= 1, 0)   # hard-tanh STE (bnn/ops.py:68-73)
        return (gx if need_x else None, gw if need_w else None, gb if need_b else None, None, None, None)


# ---- training-mode BatchNorm (+ residual) (+ ReLU) as one fused op ---------------------------------------------------
# Of the 29.5 ms ResNet-18 step at batch 256, 9.1 ms were the library's BatchNorm kernels and 6 ms its ReLU / add /
# their backward passes over the same fp32 tensors.  `bn_act` evaluates  act(bn(x) (+ identity))  of the reference's
# blocks (bnn/models/layers/res_block.py:40-56) in three launches forward and three backward (csrc/bn_train.hip).
FUSED_BN = os.environ.get("BNN_AMD_TRAIN_FUSED_BN", "1") == "1"


class BNActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, bn, relu):
        # the kernels address NCHW: what is SAVED must be the contiguous tensor they ran on (a channels_last / strided
        # input would otherwise reach the backward kernels with its own strides: wrong dx, dgamma, dbeta)
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        y, mean, invstd = hipops.bn_train_forward(x, weight, bias, bn.running_mean, bn.running_var,
                                                  _momentum(bn), bn.eps, relu, residual)
        _stats_written(bn)
        ctx.relu, ctx.has_res = relu, residual is not None
        ctx.save_for_backward(x, y if relu else None, mean, invstd, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, mean, invstd, weight = ctx.saved_tensors
        dx, dgamma, dbeta, dres = hipops.bn_train_backward(gy.contiguous(), y, x, mean, invstd, weight,
                                                           want_dres=ctx.has_res and ctx.relu and ctx.needs_input_grad[3])
        if ctx.has_res and ctx.needs_input_grad[3] and dres is None:
            dres = gy            # no ReLU in between: the residual's gradient IS the incoming one
        return (dx if ctx.needs_input_grad[0] else None, dgamma if ctx.needs_input_grad[1] else None,
                dbeta if ctx.needs_input_grad[2] else None, dres, None, None)


def _stats_written(bn: nn.BatchNorm2d) -> None:
    """The kernel has updated ``running_mean`` / ``running_var`` through raw pointers: bump their version counters (what an
    in-place torch op would have done — caches keyed on ``_version``, e.g. ``tails.cached_fold``, must see the
    change) and count the batch."""
    for t in (bn.running_mean, bn.running_var):
        if t is not None:
            torch.autograd.graph.increment_version(t)
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)


def _momentum(bn: nn.BatchNorm2d) -> float:
    if bn.momentum is None:     # cumulative moving average (torch: 1 / num_batches_tracked, counted before the update)
        return 1.0 / float(int(bn.num_batches_tracked) + 1)
    return float(bn.momentum)


def bn_act_applies(bn: nn.Module, act, x: torch.Tensor) -> bool:
    """The fused op stands in for ``act(bn(x) [+ identity])`` iff: training mode with batch statistics, a stock
    ``nn.BatchNorm2d`` with running statistics, ``act`` None or a stock ``nn.ReLU``, fp32 NCHW on a HIP device, autograd
    recording, and no forward hooks on either module (they would not fire)."""
    return (FUSED_BN and ENABLED and type(bn) is nn.BatchNorm2d and bn.training and bn.track_running_stats
            and bn.running_mean is not None and (act is None or type(act) is nn.ReLU)
            and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and torch.is_grad_enabled()
            and (bn.weight is None or bn.weight.dtype == torch.float32)
            and not bn._forward_hooks and not bn._forward_pre_hooks
            and (act is None or (not act._forward_hooks and not act._forward_pre_hooks)))


def bn_act(x: torch.Tensor, bn: nn.BatchNorm2d, act=None, residual=None) -> torch.Tensor:
    """``act(bn(x) (+ residual))`` — fused when ``bn_act_applies``, else the modules themselves."""
    if bn_act_applies(bn, act, x):
        return BNActFn.apply(x, bn.weight, bn.bias, residual, bn, act is not None)
    y = bn(x)
    if residual is not None:
        y = y + residual
    return y if act is None else act(y)


class StemTailFn(torch.autograd.Function):
    """``maxpool(relu(bn1(x)))`` of the stem (bnn/models/resnet.py:150-153), training mode: the normalised tensor is
    never written; one byte per pooled output routes the gradient back (csrc/bn_train.hip, "stem tail")."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn):
        x = x.contiguous()   # (see BNActFn.forward)
        p, code, mean, invstd = hipops.bn_relu_maxpool_train_forward(x, weight, bias, bn.running_mean, bn.running_var,
                                                                    _momentum(bn), bn.eps)
        _stats_written(bn)
        ctx.save_for_backward(x, p, code, mean, invstd, weight)
        ctx.mark_non_differentiable(code)
        return p, code

    @staticmethod
    def backward(ctx, gp, _gcode):
        x, p, code, mean, invstd, weight = ctx.saved_tensors
        dx, dgamma, dbeta = hipops.bn_relu_maxpool_train_backward(gp.contiguous(), p, code, x, mean, invstd, weight)
        return (dx if ctx.needs_input_grad[0] else None, dgamma if ctx.needs_input_grad[1] else None,
                dbeta if ctx.needs_input_grad[2] else None, None)


# The stem's convolution in a training step (bnn/models/resnet.py:150): the library's 7x7 forward is 1.2 ms of a 19.6 ms
# step at batch 256; the inference stem's MFMA core with a raw-conv epilogue (csrc/stem_rows.hip, RAW) is ~0.4 ms.  The
# weight gradient stays the library's (the input is data: no input gradient).
FUSED_STEM_CONV = os.environ.get("BNN_AMD_TRAIN_STEM_CONV", "1") == "1"
FUSED_STEM_WGRAD = os.environ.get("BNN_AMD_TRAIN_STEM_WGRAD", "1") == "1"


class StemConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return hipops.stem7x7_conv(x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        if not ctx.needs_input_grad[0] and FUSED_STEM_WGRAD and hipops.stem7x7_wgrad_supported(x):
            # the input is data: the weight gradient is the whole backward (csrc/stem_wgrad.hip: fp32 MFMA, 0.6 ms
            # where the library's implicit GEMM with its two NHWC transposes of the 822 MB gradient takes 1.46)
            return None, (hipops.stem7x7_wgrad(x, g.contiguous()) if ctx.needs_input_grad[1] else None)
        gx, gw, _ = torch.ops.aten.convolution_backward(
            g.contiguous(), x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
            [bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1]), False])
        return (gx if ctx.needs_input_grad[0] else None, gw if ctx.needs_input_grad[1] else None)


def stem_conv_applies(conv: nn.Module, x: torch.Tensor) -> bool:
    """``conv(x)`` is the canonical real-valued stem convolution in a training step on a HIP device: a stock
    ``nn.Conv2d`` (or a binary-class layer whose recipe is all-Identity, examples/cifar10.py:71) 3 -> 64, 7x7, stride 2,
    padding 3, no bias, fp32 NCHW, no hooks."""
    from .executor import _is_float_layer
    return (FUSED_STEM_CONV and ENABLED and isinstance(conv, nn.Conv2d) and _is_float_layer(conv)
            and conv.in_channels == 3 and conv.out_channels == 64 and tuple(conv.kernel_size) == (7, 7)
            and tuple(conv.stride) == (2, 2) and tuple(conv.padding) == (3, 3) and tuple(conv.dilation) == (1, 1)
            and conv.groups == 1 and conv.bias is None and conv.padding_mode == "zeros"
            and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and conv.weight.dtype == torch.float32 and torch.is_grad_enabled()
            and not conv._forward_hooks and not conv._forward_pre_hooks and not conv._backward_hooks
            and not conv._backward_pre_hooks          # (they would not fire on the fused path)
            and not torch.is_autocast_enabled())      # (under AMP the reference's conv returns fp16: the module itself)


def stem_conv(x: torch.Tensor, conv: nn.Module) -> torch.Tensor:
    """``conv(x)`` — the MFMA kernel when ``stem_conv_applies``, else the module itself."""
    if stem_conv_applies(conv, x):
        return StemConvFn.apply(x, conv.weight)
    return conv(x)


def stem_tail(x: torch.Tensor, bn: nn.Module, act: nn.Module, pool: nn.Module) -> torch.Tensor:
    """``pool(act(bn(x)))`` — one fused op when it is BatchNorm2d (training) -> ReLU -> MaxPool2d(3, 2, 1) on a HIP
    device, else the modules themselves."""
    if (bn_act_applies(bn, act, x) and act is not None and type(pool) is nn.MaxPool2d
            and pool.kernel_size in (3, (3, 3)) and pool.stride in (2, (2, 2)) and pool.padding in (1, (1, 1))
            and pool.dilation in (1, (1, 1)) and not pool.ceil_mode and not pool.return_indices
            and not pool._forward_hooks and not pool._forward_pre_hooks):
        return StemTailFn.apply(x, bn.weight, bn.bias, bn)[0]
    return pool(bn_act(x, bn, act))


# The shortcut's AvgPool2d(2, 2) in a training step (bnn/models/resnet.py:128-133): the library's forward, and a streaming
# kernel for its backward (the library's generic avg_pool2d backward: 0.53 ms per ResNet-18 step at batch 256, this 0.09).
FUSED_SHORTCUT_POOL = os.environ.get("BNN_AMD_TRAIN_SHORTCUT_POOL", "1") == "1"


class AvgPool2x2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return torch.nn.functional.avg_pool2d(x, 2)

    @staticmethod
    def backward(ctx, g):
        return hipops.avgpool2x2_backward(g.contiguous())


def shortcut_pool(x: torch.Tensor, pool: nn.Module) -> torch.Tensor:
    """``pool(x)`` — with the streaming backward when it is ``AvgPool2d(2, 2)`` without padding on an even-sized fp32 map of
    a HIP device under autograd, else the module itself."""
    if (FUSED_SHORTCUT_POOL and ENABLED and type(pool) is nn.AvgPool2d and pool.kernel_size in (2, (2, 2))
            and pool.stride in (2, (2, 2)) and pool.padding in (0, (0, 0)) and pool.divisor_override is None
            and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
            and torch.is_grad_enabled() and x.requires_grad
            and not pool._forward_hooks and not pool._forward_pre_hooks and not pool._backward_hooks):
        return AvgPool2x2Fn.apply(x)
    return pool(x)


# The weight hook of a training step as two kernels instead of torch's ~14 per layer (csrc/xnor_train.hip): the forward
# needs no fp32 What at all (the conv reads the packed weights), the backward derives What for the input-gradient
# kernel and maps dL/dWhat to dL/dW (sign STE, alpha = mean|Wc|, centring) in one kernel each.
FUSED_WEIGHT_HOOK = os.environ.get("BNN_AMD_TRAIN_FUSED_WEIGHT_HOOK", "1") == "1"


class BinaryConv2dTrainFusedFn(torch.autograd.Function):
    """``BinaryConv2dTrainFn`` with the XNORWeightBinarizer inside: inputs are x and the LATENT weight W."""

    @staticmethod
    def forward(ctx, x, w, bias, layer, plan, packed):
        out = hipops.bconv2d_direct(x, packed, bias, None, layer.stride, layer.padding, layer.dilation)   # one launch
        stride, padding, dilation = tuple(layer.stride), tuple(layer.padding), tuple(layer.dilation)
        ctx.x_shape = None
        if PACKED_STATE and BINARY_GRADS and hipops.grad_supported(x.shape, w.shape, stride, padding, dilation):
            sv = hipops.pack_act_ste(x)
            ctx.save_for_backward(sv.sign.P, sv.sign.M, sv.T, w)
            ctx.x_shape = tuple(x.shape)
            kept = sv.nbytes()
        else:
            ctx.save_for_backward(x, w)
            kept = x.numel() * x.element_size()
        global _saved_bytes
        _saved_bytes += kept
        ctx.conf = (stride, padding, dilation, bias is not None, None if bias is None else tuple(bias.shape),
                    bool(plan.center), bool(plan.compute_alpha))
        return out

    @staticmethod
    def backward(ctx, g):
        stride, padding, dilation, has_bias, bias_shape, center, compute_alpha = ctx.conf
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]
        if ctx.x_shape is not None:
            P, M, T, w = ctx.saved_tensors
            x = hipops.SavedAct(hipops.PackedAct(P, M, ctx.x_shape), T, ctx.x_shape)
        else:
            x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = gwhat = gb = None
        if ctx.x_shape is not None or (
                BINARY_GRADS and hipops.grad_supported(x.shape, w.shape, stride, padding, dilation)):
            if need_x:
                packed, alpha = hipops.xnor_grad_pack_weight(w, center, compute_alpha)   # (one launch, the same bytes)
                gx = hipops.bconv_grad_input(g, x, packed, alpha, w.shape[2], stride[0])       # STE mask fused
            if need_w:
                # the split-K slabs: up to 16 are added inside the hook's kernel (one workgroup per output channel walks them),
                # the hundreds of a 64-channel layer by the library's parallel reduction
                gwhat = hipops.bconv_grad_weight(g, x, w.shape[2], stride[0], reduce=False)
                if gwhat.shape[0] > 16:
                    gwhat = gwhat.sum(0)
            if need_b:
                gb = g.sum(dim=(0, 2, 3))
        else:
            w_hat = hipops.xnor_what(w, center, compute_alpha)
            gx, gwhat, gb = torch.ops.aten.convolution_backward(
                g, torch.sign(x), w_hat, list(bias_shape) if has_bias else None, list(stride), list(padding),
                list(dilation), False, [0, 0], 1, [bool(need_x), bool(need_w), bool(need_b)])
            if need_x:
                gx = gx.masked_fill(x.abs() >= 1, 0)   # hard-tanh STE (bnn/ops.py:68-73)
        gw = hipops.xnor_weight_backward(w, gwhat, center, compute_alpha) if need_w else None
        return (gx if need_x else None, gw, gb if need_b else None, None, None, None)


def conv2d_train(layer: nn.Module, x: torch.Tensor, plan, packed) -> torch.Tensor:
    """``bnn.layers.Conv2d.forward`` with autograd recording: HIP forward, library backward."""
    w = layer.weight
    if (FUSED_WEIGHT_HOOK and w.dim() == 4 and w.is_contiguous() and w.shape[2] * w.shape[3] <= 1024
            and not layer.weight_pre_process._forward_hooks and not layer.weight_pre_process._forward_pre_hooks):
        out = BinaryConv2dTrainFusedFn.apply(x, w, layer.bias, layer, plan, packed)
        if plan.scale is not None:                          # BasicScaleBinarizer (bnn/ops.py:200-202)
            out = out * plan.scale
        return out
    w_hat = layer.weight_pre_process(layer.weight)          # autograd edge to W (sign STE, alpha)
    out = BinaryConv2dTrainFn.apply(x, w_hat, layer.bias, layer, plan, packed)
    if plan.scale is not None:                              # BasicScaleBinarizer (bnn/ops.py:200-202)
        out = out * plan.scale
    return out


def make_ddp(model: nn.Module, device: torch.device | None = None, **kw) -> nn.Module:
    """Wrap ``model`` for one-process-per-GPU data-parallel training (RCCL when on GPU, gloo on CPU).
    ``torch.distributed`` must be initialised.  Gradient buckets default to 64 MB: xGMI rings are
    per-link bound, so fewer, larger all-reduces beat the 25 MB default tuned for NVSwitch."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    kw.setdefault("bucket_cap_mb", 64)
    if device is not None and device.type == "cuda":
        return DDP(model.to(device), device_ids=[device.index], output_device=device.index, **kw)
    return DDP(model, **kw)
