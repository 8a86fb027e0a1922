"""Training path on the GPU: HIP XNOR/popcount forward under autograd, fp32 library
backward with the straight-through estimator (bnn/ops.py:63-73), compared with the torch composition
(the reference's formulation) on the same device."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import bnn_amd as bnn
from bnn_amd import fastpath, hipops, native, training
from bnn_amd.models import resnet18
from bnn_amd.ops import BasicInputBinarizer, BasicScaleBinarizer, XNORWeightBinarizer
from tests.golden import gen

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _net_call_means_per_layer():
    """In this module `net(x)` means the per-layer drop-in path (one launch per binary layer + torch BN / ReLU / add);
    the fused executor is built explicitly (`FusedResNet(net)`).  What `net(x)` does by default — the fused executor,
    bnn_amd/inference.py: AutoFusion — is tested in tests/test_gpu_dropin.py."""
    from bnn_amd.inference import per_layer_forward
    with per_layer_forward():
        yield
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _layer(C, O, k, stride, pad, bias, scale, center, seed):
    conv = nn.Conv2d(C, O, k, stride=stride, padding=pad, bias=bias)
    conv.weight.data.copy_(torch.from_numpy(gen.conv_weight("kaiming", seed, (O, C, k, k))))
    if bias:
        conv.bias.data.copy_(torch.from_numpy(0.1 * gen.normal(seed + 1, (O,))))
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer,
                      activation_post_process=BasicScaleBinarizer if scale else bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer.with_args(center_weights=center))
    layer = bnn.prepare_binary_model(conv, cfg)
    if scale:
        layer.activation_post_process.alpha.data.copy_(
            torch.from_numpy((0.5 + gen.uniform(seed + 2, (O,))).astype(np.float32)).view(1, -1, 1, 1))
    return layer.to(DEV).train()


def _grads(layer, x_np, g_np, enabled):
    training.ENABLED = enabled
    try:
        for p in layer.parameters():
            p.grad = None
        x = dev(x_np).requires_grad_(True)
        y = layer(x)
        y.backward(dev(g_np))
        return y.detach(), x.grad.clone(), {n: p.grad.clone() for n, p in layer.named_parameters()}
    finally:
        training.ENABLED = True


@pytest.mark.parametrize("C,O,k,stride,pad,bias,scale,center", [
    (64, 64, 3, 1, 1, False, False, False),
    (128, 96, 3, 2, 1, True, True, False),
    (70, 40, 1, 1, 0, True, False, True),
    (256, 64, 3, 1, 1, False, True, True),
])
def test_training_forward_and_gradients_match_composition(C, O, k, stride, pad, bias, scale, center):
    layer = _layer(C, O, k, stride, pad, bias, scale, center, seed=C + O)
    x = (gen.normal(5, (3, C, 12, 10)) * 0.8).astype(np.float32)      # |x| straddles the STE threshold 1
    ho, wo = (12 + 2 * pad - k) // stride + 1, (10 + 2 * pad - k) // stride + 1
    g = gen.normal(6, (3, O, ho, wo))
    before = fastpath.stats()["conv2d_train"]
    y1, gx1, gp1 = _grads(layer, x, g, enabled=True)
    assert fastpath.stats()["conv2d_train"] == before + 1               # the HIP forward ran
    y0, gx0, gp0 = _grads(layer, x, g, enabled=False)
    assert fastpath.stats()["conv2d_train"] == before + 1               # ... and not in the control run
    tol = dict(rtol=1e-4, atol=1e-5 * float(y0.abs().max()))
    assert torch.allclose(y1, y0, **tol)
    # input gradient: same library conv on the same operands, same STE mask
    assert torch.allclose(gx1, gx0, rtol=1e-4, atol=1e-5 * float(gx0.abs().max()))
    assert ((gx1 == 0) == (gx0 == 0)).all()
    assert gp1.keys() == gp0.keys()
    for n in gp0:
        assert torch.allclose(gp1[n], gp0[n], rtol=1e-3, atol=1e-4 * float(gp0[n].abs().max()) + 1e-7), n


def test_inference_and_training_forwards_agree_and_repack_after_optimizer_step():
    layer = _layer(64, 64, 3, 1, 1, False, False, False, seed=9)
    x = dev(gen.normal(3, (2, 64, 8, 8)))
    with torch.no_grad():
        y_inf = layer(x)
    y_tr = layer(x.clone().requires_grad_(True))
    assert torch.equal(y_inf, y_tr.detach())                             # same kernels, same packed weights
    opt = torch.optim.SGD(layer.parameters(), lr=0.5)
    packs = fastpath.stats()["weight_packs"]
    y_tr.square().mean().backward()
    opt.step()
    y2 = layer(x.clone().requires_grad_(True))
    assert fastpath.stats()["weight_packs"] == packs + 1                 # version counter changed -> re-pack
    assert not torch.equal(y2.detach(), y_inf)


def test_training_pack_is_asynchronous_and_a_zero_weight_is_caught_one_step_later():
    layer = _layer(64, 32, 3, 1, 1, False, False, False, seed=21)
    x = dev(gen.normal(4, (2, 64, 6, 6)))
    y = layer(x.clone().requires_grad_(True))
    pw = layer.__dict__["_bnn_packed"][1]
    assert pw.zero_probe is None and not pw.has_zero   # the first pack of a layer reads its zero flag (zeros from the start)
    with torch.no_grad():
        layer.weight.mul_(1.0)                         # an optimizer step: new version
    y = layer(x.clone().requires_grad_(True))
    pw = layer.__dict__["_bnn_packed"][1]
    assert pw.zero_probe is not None and not pw.has_zero and not pw.zero_found_later()     # from then on: optimistic
    with torch.no_grad():
        layer.weight[3, 5] = 0.0                       # an exact zero appears between two steps
    layer(x.clone().requires_grad_(True))              # packed optimistically: the zeros are only flagged
    assert layer.__dict__["_bnn_packed"][1].zero_found_later()
    with torch.no_grad():
        layer.weight.mul_(1.0)                         # new version -> re-pack; the late flag is noticed now
    with pytest.warns(RuntimeWarning, match="exactly 0"):
        y2 = layer(x.clone().requires_grad_(True))
    assert layer.__dict__["_bnn_packed"][1].has_zero   # synchronous, zero-aware pack from here on
    training.ENABLED = False
    try:
        ref = layer(x.clone().requires_grad_(True))
    finally:
        training.ENABLED = True
    y2, ref = y2.detach(), ref.detach()
    assert torch.allclose(y2, ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))


def test_small_resnet_trains_on_the_hip_forward():
    """A few SGD steps of a binary ResNet-18: the loss trajectory with the HIP forward follows the
    composition's (identical math up to fp rounding in alpha and the conv sums)."""
    def run(enabled):
        training.ENABLED = enabled
        try:
            torch.manual_seed(0)
            net = resnet18(num_classes=10)
            cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                              weight_pre_process=XNORWeightBinarizer)
            net = bnn.prepare_binary_model(net, cfg, ignore_layers_name=["conv1", "fc"]).to(DEV).train()
            opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
            x = dev(gen.normal(21, (16, 3, 64, 64)))
            t = torch.arange(16, device=DEV) % 10
            losses = []
            for _ in range(4):
                opt.zero_grad(set_to_none=True)
                loss = nn.functional.cross_entropy(net(x), t)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
            return losses
        finally:
            training.ENABLED = True
    before = fastpath.stats()["conv2d_train"]
    fast = run(True)
    assert fastpath.stats()["conv2d_train"] == before + 4 * 19
    slow = run(False)
    assert min(fast[1:]) < fast[0]                                       # it learns
    # same first forward up to sign flips of exact-zero sums (the float conv leaves +-1e-9 residues where
    # the integer path gives 0.0: DESIGN.md "exact zero"), which batch-statistics BN amplifies slightly
    assert abs(fast[0] - slow[0]) < 2e-3 * max(1.0, abs(slow[0]))
    # later steps: binarised nets are chaotic under SGD (one flipped sign moves the whole trajectory, and the
    # library's backward uses atomics), so only the scale of the losses is compared
    assert abs(np.log(fast[-1] / slow[-1])) < 1.0


def test_torch_custom_ops_forward_and_autograd():
    """torch.ops.bnn_amd.binary_conv2d / binary_linear / pack_sign on the device: same numbers as the
    layer, gradients equal the composition's."""
    import oracle
    layer = _layer(96, 40, 3, 2, 1, True, True, False, seed=77).eval()
    x_np = (gen.normal(8, (2, 96, 9, 11)) * 0.9).astype(np.float32)
    x = dev(x_np)
    scale = layer.activation_post_process.alpha
    with torch.no_grad():
        ref = layer(x)
        y = torch.ops.bnn_amd.binary_conv2d(x, layer.weight, layer.bias, scale, [2, 2], [1, 1], [1, 1], False, True)
    assert torch.equal(y, ref)
    P, M = torch.ops.bnn_amd.pack_sign(x)
    Pr, Mr = oracle.pack_act(x_np)
    assert np.array_equal(P.cpu().numpy().view(np.uint64), Pr) and np.array_equal(M.cpu().numpy().view(np.uint64), Mr)
    # autograd through the op vs through the torch composition of the same layer
    g = dev(gen.normal(9, tuple(ref.shape)))
    xa = x.clone().requires_grad_(True)
    w, b, s = (t.detach().clone().requires_grad_(True) for t in (layer.weight, layer.bias, scale))
    torch.ops.bnn_amd.binary_conv2d(xa, w, b, s, [2, 2], [1, 1], [1, 1], False, True).backward(g)
    _, gx0, gp0 = _grads(layer.train(), x_np, g.cpu().numpy(), enabled=False)
    assert torch.allclose(xa.grad, gx0, rtol=1e-4, atol=1e-5 * float(gx0.abs().max()))
    assert torch.allclose(w.grad, gp0["weight"], rtol=1e-3, atol=1e-4 * float(gp0["weight"].abs().max()))
    assert torch.allclose(b.grad, gp0["bias"], rtol=1e-3, atol=1e-4 * float(gp0["bias"].abs().max()))
    assert torch.allclose(s.grad, gp0["activation_post_process.alpha"], rtol=1e-3,
                          atol=1e-4 * float(gp0["activation_post_process.alpha"].abs().max()))
    # linear
    lw = dev(gen.conv_weight("kaiming", 5, (11, 70, 1, 1)).reshape(11, 70))
    lx = dev(gen.normal(6, (3, 5, 70)))
    z = torch.ops.bnn_amd.binary_linear(lx, lw, None, None, False, True)
    zr = torch.nn.functional.linear(torch.sign(lx), torch.sign(lw) * lw.abs().mean(dim=1, keepdim=True))
    assert z.shape == (3, 5, 11) and torch.allclose(z, zr, rtol=1e-5, atol=1e-5)


GRAD_SHAPES = [  # (N, O, C, H, W): every slot width (8/16/32/64), channel tails, multi-chunk images, both NSUB variants
    (3, 64, 64, 12, 10), (2, 40, 70, 7, 7), (2, 128, 96, 14, 14), (2, 64, 64, 56, 56), (2, 512, 512, 7, 7),
    (1, 32, 16, 5, 64), (5, 130, 200, 28, 28), (2, 16, 8, 1, 1), (3, 96, 33, 9, 17), (1, 256, 256, 14, 14),
]


@pytest.mark.parametrize("ks,stride", [(3, 1), (3, 2), (1, 1)], ids=["3x3s1", "3x3s2", "1x1"])
@pytest.mark.parametrize("shape", GRAD_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_binary_gradient_kernels_match_library_backward(shape, ks, stride):
    """csrc/grad.hip (MFMA: g split into three bf16 terms, ternary operand exact) against aten::convolution_backward on
    the same operands: dL/dx (with the STE mask) and dL/dWhat, fp32-convolution rounding class.  3x3 / pad 1 at
    stride 1 and 2, and the 1x1 / pad 0 layer of the shortcut branches."""
    from bnn_amd import hipops
    N, O, C, H, W = shape
    pad = ks // 2
    x = dev((gen.normal(gen.seed_of("gx", shape), (N, C, H, W)) * 0.9).astype(np.float32))
    x.view(-1)[::7] = 0.0                                            # exact zeros: sign(0) == 0
    g = dev(gen.normal(gen.seed_of("gg", shape), (N, O, (H - 1) // stride + 1, (W - 1) // stride + 1)))
    w = dev(gen.conv_weight("kaiming", gen.seed_of("gw", shape), (O, C, ks, ks)))
    w.view(-1)[::11] = 0.0                                           # and zero weights
    w_hat = torch.sign(w) * w.abs().flatten(1).mean(1).view(-1, 1, 1, 1)
    assert hipops.grad_supported(x.shape, w_hat.shape, stride, pad, 1)
    packed, alpha = hipops.grad_pack_weight(w_hat)
    assert torch.equal(alpha, w_hat.abs().flatten(1).amax(1))
    gx = hipops.bconv_grad_input(g, x, packed, alpha, ks, stride)
    gw = hipops.bconv_grad_weight(g, x, ks, stride)
    conf = ([stride, stride], [pad, pad], [1, 1], False, [0, 0], 1, [True, True, False])
    rx, rw, _ = torch.ops.aten.convolution_backward(g, torch.sign(x), w_hat, None, *conf)
    rx = rx.masked_fill(x.abs() >= 1, 0)
    assert gx.shape == rx.shape and gw.shape == rw.shape
    assert ((gx == 0) | (x.abs() < 1)).all() and ((x.abs() >= 1) <= (gx == 0)).all()
    assert torch.allclose(gx, rx, rtol=1e-4, atol=2e-5 * float(rx.abs().max()))
    assert torch.allclose(gw, rw, rtol=1e-4, atol=2e-5 * float(rw.abs().max()))
    # fp64 reference: the error is that of an fp32 convolution, not of 16-bit operands
    rx64, rw64, _ = torch.ops.aten.convolution_backward(g.double(), torch.sign(x).double(), w_hat.double(), None, *conf)
    rx64 = rx64.masked_fill(x.abs() >= 1, 0)
    assert float((gx.double() - rx64).abs().max()) <= 4e-6 * float(rx64.abs().max())
    assert float((gw.double() - rw64).abs().max()) <= 4e-6 * float(rw64.abs().max())


@pytest.mark.parametrize("gscale", [1e-6, 1e-9, 3e-13, 1e4], ids=lambda s: f"g*{s:g}")
@pytest.mark.parametrize("ks,stride,shape", [(3, 1, (2, 128, 96, 14, 14)), (3, 2, (3, 64, 64, 12, 10)),
                                             (1, 1, (2, 40, 70, 7, 7))], ids=["3x3s1", "3x3s2", "1x1"])
def test_binary_gradient_kernels_keep_fp32_accuracy_at_realistic_gradient_magnitudes(ks, stride, shape, gscale):
    """The incoming gradient of a mean-reduced loss at batch 256 is 1e-5 .. 1e-9 (times alpha ~ 0.01 inside dgrad),
    not N(0,1).  The real operand is split into THREE bf16 terms (fp32 exponent range, 24 mantissa bits), so the
    error relative to an fp64 reference is the same 4e-6 at every magnitude — round 2's fp16 hi+lo split had an
    absolute 2^-24 quantum and returned exact zeros below ~3e-8."""
    from bnn_amd import hipops
    N, O, C, H, W = shape
    pad = ks // 2
    x = dev((gen.normal(gen.seed_of("gx", shape), (N, C, H, W)) * 0.9).astype(np.float32))
    g = dev(gen.normal(gen.seed_of("gg", shape), (N, O, (H - 1) // stride + 1, (W - 1) // stride + 1))) * gscale
    w = dev(gen.conv_weight("kaiming", gen.seed_of("gw", shape), (O, C, ks, ks)))
    w_hat = torch.sign(w) * w.abs().flatten(1).mean(1).view(-1, 1, 1, 1)
    packed, alpha = hipops.grad_pack_weight(w_hat)
    gx = hipops.bconv_grad_input(g, x, packed, alpha, ks, stride)
    gw = hipops.bconv_grad_weight(g, x, ks, stride)
    conf = ([stride, stride], [pad, pad], [1, 1], False, [0, 0], 1, [True, True, False])
    rx64, rw64, _ = torch.ops.aten.convolution_backward(g.double(), torch.sign(x).double(), w_hat.double(), None, *conf)
    rx64 = rx64.masked_fill(x.abs() >= 1, 0)
    assert float(rx64.abs().max()) > 0 and float(rw64.abs().max()) > 0
    assert float((gx.double() - rx64).abs().max()) <= 4e-6 * float(rx64.abs().max())
    assert float((gw.double() - rw64).abs().max()) <= 4e-6 * float(rw64.abs().max())
    # element-wise too: no value is flushed (the fp16 split returned exact zeros here)
    big = rx64.abs() > 1e-3 * rx64.abs().max()
    assert float(((gx.double() - rx64).abs() / rx64.abs().clamp_min(1e-300))[big].max()) < 1e-3


def test_unsupported_gradient_shapes_fall_back_to_the_library():
    from bnn_amd import hipops
    assert not hipops.grad_supported((2, 8, 8, 8), (8, 8, 1, 1), 2, 0, 1)     # strided 1x1
    assert not hipops.grad_supported((2, 8, 8, 8), (8, 8, 3, 3), 1, 0, 1)     # 3x3 without padding
    assert not hipops.grad_supported((2, 8, 8, 8), (8, 8, 5, 5), 1, 2, 1)
    assert not hipops.grad_supported((2, 8, 8, 65), (8, 8, 3, 3), 1, 1, 1)    # rows wider than a 64-slot chunk
    layer = _layer(8, 8, 1, 2, 0, False, False, False, seed=5)
    x = (gen.normal(3, (2, 8, 9, 9)) * 0.8).astype(np.float32)
    g = gen.normal(4, (2, 8, 5, 5))
    y, gx, gp = _grads(layer, x, g, enabled=True)
    y0, gx0, gp0 = _grads(layer, x, g, enabled=False)
    assert torch.allclose(y, y0, rtol=1e-5, atol=1e-5) and torch.allclose(gx, gx0, rtol=1e-4, atol=1e-5)


def test_binary_gradient_kernels_are_used_and_can_be_switched_off():
    layer = _layer(64, 64, 3, 1, 1, False, False, False, seed=33)
    x = (gen.normal(8, (2, 64, 14, 14)) * 0.8).astype(np.float32)
    g = gen.normal(9, (2, 64, 14, 14))
    from bnn_amd import native
    n0 = native.launch_count()
    y1, gx1, gp1 = _grads(layer, x, g, enabled=True)
    launches_binary = native.launch_count() - n0
    training.BINARY_GRADS = False
    try:
        n0 = native.launch_count()
        y0, gx0, gp0 = _grads(layer, x, g, enabled=True)
        launches_library = native.launch_count() - n0
    finally:
        training.BINARY_GRADS = True
    # the binary path: weight prep (one launch from W, round 5) + dgrad + wgrad; the library path: the fp32 What (one launch)
    assert launches_binary == launches_library + 2
    assert torch.equal(y1, y0)
    assert torch.allclose(gx1, gx0, rtol=1e-4, atol=2e-5 * float(gx0.abs().max()))
    for n in gp0:
        assert torch.allclose(gp1[n], gp0[n], rtol=1e-3, atol=1e-4 * float(gp0[n].abs().max()) + 1e-7), n


@pytest.mark.parametrize("ks,stride", [(3, 1), (3, 2), (1, 1)], ids=["3x3s1", "3x3s2", "1x1"])
@pytest.mark.parametrize("shape", GRAD_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_gradients_from_three_bits_per_element_equal_those_from_the_fp32_input(shape, ks, stride):
    """Round 4: what a training step keeps of x for the backward is the sign planes + the mask |x| < 1 (3 bits per
    element, csrc/pack_ste.hip) instead of the fp32 tensor; the gradient kernels read the planes
    (bnn_hip_bconv_grad_*_packed_f32) and must return THE SAME BITS as from the fp32 input — special values included
    (the reference's STE: grad * 1[|x| < 1] with NaN -> 0, bnn/ops.py:68-73; sign(+-0) = sign(NaN) = 0)."""
    import oracle
    from bnn_amd import hipops
    N, O, C, H, W = shape
    pad = ks // 2
    xn = (gen.normal(gen.seed_of("px", shape), (N, C, H, W)) * 0.9).astype(np.float32)
    flat = xn.reshape(-1)
    flat[::7] = 0.0
    flat[3::11] = -0.0
    flat[5::13] = np.nan
    flat[1::17] = 1.0          # |x| == 1 is outside the mask
    flat[2::19] = -1.0
    flat[4::23] = np.float32(1e-45)
    flat[6::29] = np.inf
    x = dev(xn)
    g = dev(gen.normal(gen.seed_of("pg", shape), (N, O, (H - 1) // stride + 1, (W - 1) // stride + 1)))
    w_hat = dev(np.sign(gen.normal(gen.seed_of("pw", shape), (O, C, ks, ks))).astype(np.float32) *
                (0.5 + gen.uniform(gen.seed_of("pa", shape), (O, 1, 1, 1))).astype(np.float32))
    sv = hipops.pack_act_ste(x)
    # the planes themselves: P / M are pack_act's (== the oracle's), T is |x| < 1
    ref = hipops.pack_act(x)
    assert torch.equal(sv.sign.P, ref.P) and torch.equal(sv.sign.M, ref.M)
    Pr, Mr = oracle.pack_act(xn)
    assert np.array_equal(sv.sign.P.cpu().numpy().view(np.uint64), Pr)
    t_bits = np.zeros((N, (C + 63) // 64, H, W), np.uint64)
    with np.errstate(invalid="ignore"):
        inside = np.abs(xn) < 1.0
    for c in range(C):
        t_bits[:, c // 64] |= inside[:, c].astype(np.uint64) << np.uint64(c % 64)
    assert np.array_equal(sv.T.cpu().numpy().view(np.uint64), t_bits)
    assert sv.nbytes() * 32 == 3 * x.numel() * 4 * ((C + 63) // 64 * 64) // C      # 3 bits per (padded) element
    packed, alpha = hipops.grad_pack_weight(w_hat)
    gx_f = hipops.bconv_grad_input(g, x, packed, alpha, ks, stride)
    gx_p = hipops.bconv_grad_input(g, sv, packed, alpha, ks, stride)
    assert torch.equal(gx_f.nan_to_num(0.0), gx_p.nan_to_num(0.0)) and torch.equal(gx_f.isnan(), gx_p.isnan())
    gw_f = hipops.bconv_grad_weight(g, x, ks, stride)
    gw_p = hipops.bconv_grad_weight(g, sv, ks, stride)
    assert torch.equal(gw_f, gw_p)
    assert pad in (0, 1)


def _r18_train():
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    net = bnn.prepare_binary_model(resnet18(), cfg, custom_config_layers_name={"conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 1).items()})
    return net.to(DEV).train()


def test_training_forward_keeps_three_bits_per_element_for_the_backward():
    """Input state the binary convolutions of one ResNet-18 training forward keep for the backward
    (training.saved_input_bytes): packed state (training.PACKED_STATE / BNN_AMD_TRAIN_PACKED_STATE=1) against the fp32
    input the reference's autograd keeps — at least 5x less (32/3 = 10.7x for channel counts
    that are multiples of 64) — with the same loss and gradients."""
    net = _r18_train()
    x = dev(gen.normal(91, (4, 3, 64, 64)))
    t = torch.tensor([1, 5, 9, 13], device=DEV)

    def step(packed):
        training.PACKED_STATE = packed
        try:
            net.zero_grad()
            training.saved_input_bytes(reset=True)
            loss = torch.nn.functional.cross_entropy(net(x), t)
            kept = training.saved_input_bytes(reset=True)
            loss.backward()
        finally:
            training.PACKED_STATE = False
        return kept, [p.grad.clone() for p in net.parameters()], float(loss.detach())
    k_fp32, g_fp32, l_fp32 = step(False)
    k_packed, g_packed, l_packed = step(True)
    assert abs(l_packed - l_fp32) <= 1e-6 * abs(l_fp32)
    for a, b in zip(g_packed, g_fp32):      # (library BatchNorm backward may reduce with atomics: tight tolerance)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 * float(b.abs().max()) + 1e-12)
    # 19 binary convolutions: fp32 inputs vs three uint64 planes per 64 channels
    want_fp32 = 4 * 4 * (4 * 64 * 16 * 16 + 64 * 16 * 16 + 3 * 128 * 8 * 8 + 64 * 8 * 8 + 128 * 8 * 8 + 3 * 256 * 4 * 4 +
                         128 * 4 * 4 + 256 * 4 * 4 + 3 * 512 * 2 * 2 + 256 * 2 * 2)
    assert k_fp32 == want_fp32 and k_packed * 32 == k_fp32 * 3
    assert k_fp32 >= 5 * k_packed


# ---- round 4: training-mode BatchNorm (+ residual) (+ ReLU) as one fused op (csrc/bn_train.hip) ----------------------

@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False), (False, True)],
                         ids=["bn_relu", "bn_add_relu", "bn", "bn_add"])
@pytest.mark.parametrize("shape", [(8, 64, 56, 56), (4, 512, 7, 7), (3, 96, 9, 17), (2, 40, 3, 3), (16, 128, 28, 28)],
                         ids=lambda s: "x".join(map(str, s)))
def test_fused_batchnorm_training_op_matches_the_library(shape, relu, res):
    """training.bn_act == act(bn(x) (+ identity)) of torch.nn.BatchNorm2d in training mode (the reference's blocks,
    bnn/models/layers/res_block.py:40-56): output, saved statistics, running statistics (unbiased variance, momentum,
    num_batches_tracked) and all four gradients, to fp32 rounding (the statistics here are accumulated in fp64)."""
    N, C, H, W = shape
    x0 = dev((gen.normal(gen.seed_of("bnx", shape), shape) * 1.7 + 0.3).astype(np.float32))
    r0 = dev(gen.normal(gen.seed_of("bnr", shape), shape)) if res else None
    gy = dev(gen.normal(gen.seed_of("bng", shape), shape))

    def make():
        bn = nn.BatchNorm2d(C).to(DEV).train()
        with torch.no_grad():
            bn.weight.copy_(dev((0.5 + gen.uniform(1, (C,))).astype(np.float32)))
            bn.bias.copy_(dev((0.3 * gen.normal(2, (C,))).astype(np.float32)))
            bn.running_mean.copy_(dev((0.5 * gen.normal(3, (C,))).astype(np.float32)))
            bn.running_var.copy_(dev((0.5 + gen.uniform(4, (C,))).astype(np.float32)))
        return bn

    def run(fused):
        bn, act = make(), (nn.ReLU(inplace=True) if relu else None)
        x = x0.clone().requires_grad_(True)
        r = None if r0 is None else r0.clone().requires_grad_(True)
        training.FUSED_BN = fused
        try:
            assert training.bn_act_applies(bn, act, x) == fused
            y = training.bn_act(x, bn, act, r)
        finally:
            training.FUSED_BN = True
        y.backward(gy)
        return (y.detach(), x.grad, bn.weight.grad, bn.bias.grad, None if r is None else r.grad,
                bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked))
    got, want = run(True), run(False)
    assert got[7] == want[7] == 1
    for name, a, b, tol in (("y", got[0], want[0], 2e-5), ("dx", got[1], want[1], 2e-4), ("dgamma", got[2], want[2], 2e-4),
                            ("dbeta", got[3], want[3], 2e-4), ("running_mean", got[5], want[5], 1e-5),
                            ("running_var", got[6], want[6], 1e-5)):
        assert torch.allclose(a, b, rtol=tol, atol=tol * float(b.abs().max())), (name, float((a - b).abs().max()))
    if res:
        assert torch.allclose(got[4], want[4], rtol=1e-6, atol=0)


def test_residual_blocks_train_with_the_fused_batchnorm_and_match_the_unfused_step():
    """A BasicBlock with a down-sampling shortcut and a whole ResNet-18: one training step with the fused BN ops
    (default) against the same step with the library's BatchNorm / ReLU / add — loss, gradients, running statistics."""
    def step(fused, what):
        training.FUSED_BN = fused
        try:
            if what == "block":
                from bnn_amd.models import BasicBlock
                from bnn_amd.models.blocks import conv1x1
                ds = nn.Sequential(nn.AvgPool2d(2, 2, ceil_mode=True, count_include_pad=False), conv1x1(64, 128), nn.BatchNorm2d(128))
                cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                                  weight_pre_process=XNORWeightBinarizer)
                net = bnn.prepare_binary_model(BasicBlock(64, 128, 2, ds), cfg)
                shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
                net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 3).items()})
                net = net.to(DEV).train()
                x = dev(gen.activation("relu", 5, (4, 64, 28, 28))).requires_grad_(True)
                loss = (net(x) * dev(gen.normal(6, (4, 128, 14, 14)))).sum()
            elif what in ("pre", "h"):
                from bnn_amd.models import HBlock, PreBasicBlock
                cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                                  weight_pre_process=XNORWeightBinarizer)
                blk = PreBasicBlock(64, 64, activation=nn.PReLU) if what == "pre" else HBlock(64, 64)
                net = bnn.prepare_binary_model(blk, cfg)
                shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
                net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 3).items()})
                net = net.to(DEV).train()
                x = dev(gen.normal(5, (4, 64, 14, 14))).requires_grad_(True)
                loss = (net(x) * dev(gen.normal(6, (4, 64, 14, 14)))).sum()
            else:
                net = _r18_train()
                x = dev(gen.normal(91, (8, 3, 64, 64))).requires_grad_(True)
                loss = torch.nn.functional.cross_entropy(net(x), torch.arange(8, device=DEV) * 7)
            loss.backward()
            return (float(loss.detach()), x.grad, [p.grad for p in net.parameters()],
                    [b.clone() for n, b in net.named_buffers() if "running" in n])
        finally:
            training.FUSED_BN = True
    for what in ("block", "pre", "h", "r18"):
        l1, gx1, gp1, rb1 = step(True, what)
        l0, gx0, gp0, rb0 = step(False, what)
        assert abs(l1 - l0) <= 1e-4 * abs(l0), what
        for a, b in zip(rb1, rb0):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)
        # binarised nets are discontinuous: a batch-statistics difference of 1e-7 can flip a sign(); compare in norm
        def close(a, b, tol):
            return float((a - b).norm()) <= tol * float(b.norm()) + 1e-12
        assert close(gx1, gx0, 2e-2 if what == "r18" else 1e-3), what
        for a, b in zip(gp1, gp0):
            assert close(a, b, 5e-2 if what == "r18" else 2e-3), what


@pytest.mark.parametrize("op", ["bn_add_relu", "stem_tail"])
def test_fused_batchnorm_ops_on_channels_last_inputs(op):
    """A channels_last (or otherwise strided) input of the fused BatchNorm ops — `model.to(memory_format=channels_last)`
    makes the library stem conv emit one: the kernels address NCHW, so the forward runs on a contiguous copy and the
    BACKWARD must run on that same copy (round-4 advisor finding: the original strided tensor was saved, dx / dgamma /
    dbeta were silently wrong).  Gradients against torch's own modules on the same strided tensors; the statistics'
    version counters move (caches keyed on `_version` see the update)."""
    shape = (4, 32, 18, 22)
    N, C, H, W = shape
    x0 = dev((gen.normal(gen.seed_of("clx", shape), shape) * 1.4 + 0.2).astype(np.float32)).contiguous(
        memory_format=torch.channels_last)
    r0 = dev(gen.normal(gen.seed_of("clr", shape), shape)).contiguous(memory_format=torch.channels_last)
    assert not x0.is_contiguous()

    def run(fused):
        bn = nn.BatchNorm2d(C).to(DEV).train()
        with torch.no_grad():
            bn.weight.copy_(dev((0.5 + gen.uniform(1, (C,))).astype(np.float32)))
            bn.bias.copy_(dev((0.3 * gen.normal(2, (C,))).astype(np.float32)))
        versions = (bn.running_mean._version, bn.running_var._version)
        x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
        r = r0.clone(memory_format=torch.preserve_format).requires_grad_(True)
        assert not x.is_contiguous()
        training.FUSED_BN = fused
        try:
            if op == "stem_tail":
                y = training.stem_tail(x, bn, nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1))
            else:
                y = training.bn_act(x, bn, nn.ReLU(inplace=True), r)
        finally:
            training.FUSED_BN = True
        if fused:   # (the library's own MIOpen BatchNorm updates the statistics without moving their version counters)
            assert bn.running_mean._version > versions[0] and bn.running_var._version > versions[1]
        gy = dev(gen.normal(gen.seed_of("clg", tuple(y.shape)), tuple(y.shape)))
        y.backward(gy)
        return (y.detach(), x.grad, bn.weight.grad, bn.bias.grad, r.grad if op != "stem_tail" else None,
                bn.running_mean.clone(), bn.running_var.clone())
    got, want = run(True), run(False)
    for name, a, b, tol in (("y", got[0], want[0], 2e-5), ("dx", got[1], want[1], 3e-4), ("dgamma", got[2], want[2], 3e-4),
                            ("dbeta", got[3], want[3], 3e-4), ("running_mean", got[5], want[5], 1e-5),
                            ("running_var", got[6], want[6], 1e-5)):
        assert a.shape == b.shape
        assert torch.allclose(a, b, rtol=tol, atol=tol * float(b.abs().max()) + 1e-7), (name, float((a - b).abs().max()))
    if op != "stem_tail":
        assert torch.allclose(got[4], want[4], rtol=1e-6, atol=0)


def test_fused_stem_tail_keeps_nan_like_relu_and_maxpool():
    """A NaN in the stem's conv output (a diverged run): torch's relu + max_pool2d propagate it, and so does the fused
    stem tail — it used to clamp NaN to 0 (`fmaxf`), which hides the divergence behind all-zero activations."""
    x = dev(gen.normal(5, (2, 4, 9, 9))).requires_grad_(True)
    with torch.no_grad():
        x[1, 2, 4, 4] = float("nan")
    bn = nn.BatchNorm2d(4).to(DEV).train()
    y = training.stem_tail(x, bn, nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1))
    ref = nn.MaxPool2d(3, 2, 1)(torch.relu(nn.BatchNorm2d(4).to(DEV).train()(x.detach())))
    # (the NaN poisons the batch statistics of ITS channel: that channel is NaN in every image, the others are clean)
    assert torch.equal(torch.isnan(y), torch.isnan(ref)) and bool(torch.isnan(y).any())
    assert bool(torch.isnan(y[:, 2]).all()) and not bool(torch.isnan(y[:, [0, 1, 3]]).any())


@pytest.mark.parametrize("shape", [(4, 64, 112, 112), (3, 16, 17, 23), (2, 8, 7, 7), (5, 64, 32, 32), (2, 3, 1, 1)],
                         ids=lambda s: "x".join(map(str, s)))
def test_fused_stem_tail_matches_batchnorm_relu_maxpool_of_the_library(shape):
    """training.stem_tail == maxpool3x3/2/1(relu(bn1(x))) in training mode (bnn/models/resnet.py:150-153): pooled output,
    running statistics, and dx / dgamma / dbeta — without writing the normalised tensor (one byte per pooled output
    routes the gradient).  Odd sizes: windows cut by the border; ReLU zeros: ties whose winner passes nothing."""
    N, C, H, W = shape
    x0 = dev((gen.normal(gen.seed_of("stx", shape), shape) * 1.3 - 0.2).astype(np.float32))
    hp, wp = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    gy = dev(gen.normal(gen.seed_of("stg", shape), (N, C, hp, wp)))

    def run(fused):
        bn = nn.BatchNorm2d(C).to(DEV).train()
        with torch.no_grad():
            bn.weight.copy_(dev(((0.5 + gen.uniform(1, (C,))) * np.where(np.arange(C) % 5 == 0, -1, 1)).astype(np.float32)))
            bn.bias.copy_(dev((0.3 * gen.normal(2, (C,))).astype(np.float32)))
        act, pool = nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1)
        x = x0.clone().requires_grad_(True)
        training.FUSED_BN = fused
        try:
            y = training.stem_tail(x, bn, act, pool)
        finally:
            training.FUSED_BN = True
        y.backward(gy)
        return y.detach(), x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()
    got, want = run(True), run(False)
    assert got[0].shape == (N, C, hp, wp)
    for name, a, b, tol in (("y", got[0], want[0], 2e-5), ("dx", got[1], want[1], 3e-4), ("dgamma", got[2], want[2], 3e-4),
                            ("dbeta", got[3], want[3], 3e-4), ("running_mean", got[4], want[4], 1e-5),
                            ("running_var", got[5], want[5], 1e-5)):
        assert torch.allclose(a, b, rtol=tol, atol=tol * float(b.abs().max()) + 1e-7), (name, float((a - b).abs().max()))


@pytest.mark.parametrize("center,compute_alpha", [(False, True), (True, True), (False, False), (True, False)],
                         ids=["xnor", "xnor_centered", "sign_only", "sign_centered"])
@pytest.mark.parametrize("shape", [(64, 64, 3, 3), (40, 70, 3, 3), (128, 96, 1, 1), (8, 200, 5, 5), (512, 512, 3, 3)],
                         ids=lambda s: "x".join(map(str, s)))
def test_fused_weight_hook_matches_the_binarizer_under_autograd(shape, center, compute_alpha):
    """csrc/xnor_train.hip against XNORWeightBinarizer evaluated by torch autograd (bnn/ops.py:129-140 with the STE of
    bnn/ops.py:68-73): What and dL/dW from an arbitrary dL/dWhat, incl. weights outside (-1, 1), exact zeros, centring."""
    from bnn_amd import hipops
    wn = gen.normal(gen.seed_of("wh", shape), shape).astype(np.float32) * 0.8
    wn.reshape(-1)[::9] = 0.0
    wn.reshape(-1)[4::13] *= 2.5          # |w| >= 1: the STE blocks these
    w = dev(wn).requires_grad_(True)
    g = dev(gen.normal(gen.seed_of("wg", shape), shape))
    hook = XNORWeightBinarizer(compute_alpha=compute_alpha, center_weights=center)
    want = hook(w)
    want.backward(g)
    what = hipops.xnor_what(w, center, compute_alpha)
    assert torch.allclose(what, want.detach(), rtol=1e-6, atol=1e-7)
    dw = hipops.xnor_weight_backward(w, g, center, compute_alpha)
    assert torch.allclose(dw, w.grad, rtol=1e-4, atol=1e-5 * float(w.grad.abs().max())), float((dw - w.grad).abs().max())


def test_training_step_with_and_without_the_fused_weight_hook():
    net_x = dev(gen.normal(91, (8, 3, 64, 64)))
    t = torch.arange(8, device=DEV) * 7

    def step(fused):
        training.FUSED_WEIGHT_HOOK = fused
        try:
            net = _r18_train()
            loss = torch.nn.functional.cross_entropy(net(net_x), t)
            loss.backward()
            return float(loss.detach()), [p.grad for p in net.parameters()]
        finally:
            training.FUSED_WEIGHT_HOOK = True
    l1, g1 = step(True)
    l0, g0 = step(False)
    assert l1 == l0                                         # the forward does not depend on the hook's form
    for a, b in zip(g1, g0):
        assert float((a - b).norm()) <= 1e-3 * float(b.norm()) + 1e-12


@pytest.mark.parametrize("shape", [(4, 3, 224, 224), (3, 3, 64, 64), (2, 3, 50, 38), (1, 3, 97, 131), (5, 3, 33, 65), (1, 3, 7, 9)],
                         ids=lambda s: "x".join(map(str, s)))
def test_stem_convolution_alone_matches_fp64_and_feeds_the_fused_stem_its_values(shape):
    """bnn_hip_stem7x7_conv_f32 (round 5: conv1 of a TRAINING step on the matrix cores, bnn/models/resnet.py:150): the raw
    convolution against fp64, and — BatchNorm, ReLU and MaxPool applied to it by torch — the fused inference stem's
    output: the same MFMA stream, so the same values."""
    import torch.nn.functional as F
    x = dev(gen.normal(gen.seed_of("stemraw", shape), shape))
    w = dev(gen.conv_weight("kaiming", 3, (64, 3, 7, 7)))
    y = hipops.stem7x7_conv(x, w)
    ref = F.conv2d(x.double(), w.double(), None, 2, 3)
    assert y.shape == ref.shape
    assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    a = dev((0.5 + gen.uniform(1, (64,))).astype(np.float32) * np.where(np.arange(64) % 7 == 0, -1, 1).astype(np.float32))
    b = dev((0.3 * gen.normal(2, (64,))).astype(np.float32))
    fused, _ = hipops.stem7x7(x, w, a, b, out_packed=False)
    # fma(y, a, b) in fp64 and rounded once (the product of two fp32 values is exact in fp64)
    z = (y.double() * a.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)).float()
    mine = F.max_pool2d(torch.relu(z), 3, 2, 1)
    assert mine.shape == fused.shape
    assert float((mine != fused).float().mean()) <= 1e-6          # (double rounding of the fp64 emulation of fma: ~never)
    assert torch.allclose(mine, fused, rtol=1e-6, atol=1e-6 * float(fused.abs().max()))
    y16 = hipops.stem7x7_conv(x, w, fp16=True)
    assert float((y16.double() - ref).abs().max()) <= 3e-3 * float(ref.abs().max())


def test_training_step_with_the_stem_convolution_on_the_matrix_cores():
    """One ResNet-18 training step with conv1 as the MFMA kernel (default) against the same step with the library's
    convolution: loss, the weight gradient of conv1 (the library's backward either way) and the input gradient."""
    def step(on):
        training.FUSED_STEM_CONV = on
        try:
            net = _r18_train()
            assert training.stem_conv_applies(net.conv1, dev(gen.normal(1, (2, 3, 32, 32)))) == on
            x = dev(gen.normal(91, (8, 3, 64, 64))).requires_grad_(True)
            loss = torch.nn.functional.cross_entropy(net(x), torch.arange(8, device=DEV) * 7)
            loss.backward()
            return float(loss.detach()), net.conv1.weight.grad.clone(), x.grad.clone()
        finally:
            training.FUSED_STEM_CONV = True
    l1, gw1, gx1 = step(True)
    l0, gw0, gx0 = step(False)
    assert abs(l1 - l0) <= 1e-4 * abs(l0)

    def close(a, b, tol):      # (binarised nets are discontinuous: compare in norm, as the fused-BatchNorm test does)
        return float((a - b).norm()) <= tol * float(b.norm()) + 1e-12
    assert close(gw1, gw0, 5e-2) and close(gx1, gx0, 2e-2)


@pytest.mark.parametrize("shape", [(4, 3, 224, 224), (3, 3, 64, 64), (2, 3, 50, 38), (1, 3, 97, 131), (5, 3, 33, 65), (1, 3, 7, 9),
                                   (2, 3, 8, 256)], ids=lambda s: "x".join(map(str, s)))
def test_stem_weight_gradient_kernel_matches_fp64_autograd(shape):
    """bnn_hip_stem7x7_wgrad_f32 (round 5: the backward of conv1 in a training step, bnn/models/resnet.py:150 — the input is
    data, so the weight gradient is all of it) against fp64 autograd of the same convolution; the same bits on every
    run (partials are added in index order); the library's backward for comparison of the error."""
    import torch.nn.functional as F
    n, _, h, w_ = shape
    x = dev(gen.normal(gen.seed_of("stemwg", shape), shape))
    dy = dev(gen.normal(gen.seed_of("stemwg_dy", shape), (n, 64, (h - 1) // 2 + 1, (w_ - 1) // 2 + 1)))
    assert hipops.stem7x7_wgrad_supported(x)
    dw = hipops.stem7x7_wgrad(x, dy)
    wd = torch.zeros(64, 3, 7, 7, device=DEV, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wd, None, 2, 3).backward(dy.double())
    ref = wd.grad
    scale = float(ref.abs().max())
    err = float((dw.double() - ref).abs().max())
    lib = torch.ops.aten.convolution_backward(dy, x, wd.detach().float(), None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                              [False, True, False])[1]
    lib_err = float((lib.double() - ref).abs().max())
    assert err <= 2e-5 * scale, (err, lib_err, scale)
    assert err <= 4 * lib_err + 1e-6 * scale, (err, lib_err)         # fp32 accumulation: the library's error class
    assert torch.equal(dw, hipops.stem7x7_wgrad(x, dy))


def test_stem_weight_gradient_kernel_declines_rows_beyond_its_patch():
    x = torch.zeros(1, 3, 8, 4096, device=DEV)
    assert not hipops.stem7x7_wgrad_supported(x)
    with pytest.raises(native.NativeError):
        hipops.stem7x7_wgrad(x, torch.zeros(1, 64, 4, 2048, device=DEV))


def test_training_step_with_the_stem_weight_gradient_kernel():
    """One ResNet-18 training step with conv1's backward as the fp32 MFMA kernel (default) against the same step with the
    library's convolution backward: the same loss, conv1's weight gradient to fp32 accumulation accuracy."""
    def step(on):
        training.FUSED_STEM_WGRAD = on
        try:
            net = _r18_train()
            x = dev(gen.normal(92, (8, 3, 64, 64)))
            loss = torch.nn.functional.cross_entropy(net(x), torch.arange(8, device=DEV) * 7)
            loss.backward()
            return float(loss.detach()), net.conv1.weight.grad.clone(), net.layer1[0].conv1.weight.grad.clone()
        finally:
            training.FUSED_STEM_WGRAD = True
    l1, gw1, gl1 = step(True)
    l0, gw0, gl0 = step(False)
    assert l1 == l0 and torch.equal(gl1, gl0)                    # nothing in front of conv1's backward changed
    assert float((gw1 - gw0).norm()) <= 1e-5 * float(gw0.norm())


@pytest.mark.parametrize("shape", [(4, 64, 56, 56), (3, 16, 6, 10), (2, 5, 2, 2), (1, 7, 4, 6), (2, 128, 28, 28)],
                         ids=lambda s: "x".join(map(str, s)))
def test_shortcut_pool_backward_kernel_is_the_avgpool_backward(shape):
    """bnn_hip_avgpool2x2_backward_f32 (the shortcut's AvgPool2d of bnn/models/resnet.py:128-133 in a training step): the
    same forward, and gx = gy / 4 at the four pixels of every window — the bits of torch's own backward."""
    pool = nn.AvgPool2d(kernel_size=2, stride=2, ceil_mode=True, count_include_pad=False)
    x0 = dev(gen.normal(gen.seed_of("sp", shape), shape))
    gy = dev(gen.normal(gen.seed_of("spg", shape), (shape[0], shape[1], shape[2] // 2, shape[3] // 2)))
    x1 = x0.clone().requires_grad_(True)
    y1 = training.shortcut_pool(x1, pool)
    assert y1.grad_fn is not None and "AvgPool2x2Fn" in type(y1.grad_fn).__name__
    y1.backward(gy)
    x2 = x0.clone().requires_grad_(True)
    y2 = pool(x2)
    y2.backward(gy)
    assert torch.equal(y1, y2) and torch.equal(x1.grad, x2.grad)
    # odd sizes, padding, other windows: the module itself
    xo = dev(gen.normal(3, (1, 2, 5, 7))).requires_grad_(True)
    assert "AvgPool2x2Fn" not in type(training.shortcut_pool(xo, pool).grad_fn).__name__
    assert "AvgPool2x2Fn" not in type(training.shortcut_pool(x1, nn.AvgPool2d(3, 2, 1)).grad_fn).__name__


@pytest.mark.parametrize("center,compute_alpha", [(False, True), (True, True), (False, False), (True, False)],
                         ids=["xnor", "xnor_centered", "sign_only", "sign_centered"])
@pytest.mark.parametrize("shape", [(64, 64, 3, 3), (40, 70, 3, 3), (128, 96, 1, 1), (33, 17, 3, 3), (512, 512, 3, 3)],
                         ids=lambda s: "x".join(map(str, s)))
def test_one_launch_weight_prep_of_the_input_gradient_equals_the_three_launch_form(shape, center, compute_alpha):
    """bnn_hip_xnor_grad_pack_weight_f32 (round 5): the fragments and alpha the input-gradient kernel reads, straight from
    W — byte for byte what xnor_what -> grad_pack_weight produce (padding channels included: the buffer starts as garbage),
    with exact zeros, weights outside (-1, 1) and an all-zero output channel."""
    wn = gen.normal(gen.seed_of("wp", shape), shape).astype(np.float32) * 0.8
    wn.reshape(-1)[::9] = 0.0
    wn[1] = 0.0                              # a channel whose signs are all 0 (after centring too: its mean is 0)
    w = dev(wn)
    want_p, want_a = hipops.grad_pack_weight(hipops.xnor_what(w, center, compute_alpha))
    torch.empty(want_p.numel() * 4, dtype=torch.uint8, device=DEV).fill_(0xAB)     # (dirty the allocator's next block)
    got_p, got_a = hipops.xnor_grad_pack_weight(w, center, compute_alpha)
    assert torch.equal(got_p, want_p)
    assert torch.equal(got_a, want_a)
    if not center:
        assert float(got_a[1]) == 0.0
