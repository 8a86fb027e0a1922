"""Seeded random sweep of the binary-convolution entry points against the CPU oracle: shapes, strides,
paddings, ragged channel counts, activation kinds, the non-negative-input promise, every epilogue switch and
channel-slice outputs — bit-exact integers, bit-exact float epilogue, bit-exact re-packed planes.
(The hand-picked cases of test_gpu_parity / test_gpu_fused pin the ResNet shapes; this sweep is for the
kernel-variant dispatch in between: chunk widths, split blocks, tails.)"""
import numpy as np
import pytest
import torch

import oracle
from bnn_amd import hipops
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


def u64(t):
    return t.cpu().numpy().view(np.uint64)


def _case(seed):
    r = np.random.RandomState(seed)
    k = int(r.choice([1, 3, 3, 3]))
    C = int(r.choice([3, 16, 40, 64, 96, 128, 130, 192, 256, 320, 384, 512, 640]))
    O = int(r.choice([1, 5, 16, 31, 32, 33, 40, 64, 72, 100, 128]))
    H, W = int(r.randint(k, 21)), int(r.randint(k, 21))
    stride = int(r.choice([1, 1, 2]))
    pad = int(r.choice([0, 1])) if k == 3 else 0
    N = int(r.randint(1, 6))
    act = str(r.choice(["normal", "relu", "sparse", "negrelu"]))
    return dict(k=k, C=C, O=O, H=H, W=W, stride=stride, pad=pad, N=N, act=act, seed=seed)


CASES = [_case(s) for s in range(128)]


@pytest.mark.parametrize("c", CASES, ids=lambda c: "s{seed}_k{k}_c{C}_o{O}_{H}x{W}_s{stride}p{pad}_{act}".format(**c))
def test_random_conv_bit_exact(c):
    r = np.random.RandomState(1000 + c["seed"])
    x = gen.activation(c["act"], 5 * c["seed"] + 1, (c["N"], c["C"], c["H"], c["W"]))
    w = gen.conv_weight("kaiming", 5 * c["seed"] + 2, (c["O"], c["C"], c["k"], c["k"]))
    act = hipops.pack_act(dev(x))
    act.nonneg = c["act"] == "relu" and bool(r.randint(2))          # promise only what is true
    pw = hipops.pack_weight(dev(w))
    kw = dict(stride=c["stride"], padding=c["pad"])
    ref_out, ref_dot = oracle.binary_conv2d_int(x, w, None, None, c["stride"], c["pad"], 1, False, True)
    dot = hipops.bconv2d(act, pw, raw_dot=True, **kw).cpu().numpy()
    assert np.array_equal(dot, ref_dot)
    assert np.array_equal(hipops.bconv2d(act, pw, **kw).cpu().numpy(), ref_out)

    # a random epilogue (ABI 3 switches included), optionally into a channel slice of a wider tensor
    O = c["O"]
    s = 7 * c["seed"]
    has_bn, has_res, relu, has_prelu = (bool(r.randint(2)) for _ in range(4))
    late = has_res and bool(r.randint(2))
    pre = late and bool(r.randint(2))
    aff, prelu_pack = bool(r.randint(2)), bool(r.randint(2))
    sliced = bool(r.randint(2))
    c_off = int(r.randint(0, 9)) if sliced else 0
    c_tot = O + c_off + (int(r.randint(0, 9)) if sliced else 0)
    full = (c["N"], c_tot) + ref_dot.shape[2:]
    bn_a = (0.5 + gen.uniform(s + 1, (O,))).astype(np.float32) if has_bn else None
    bn_b = (0.3 * gen.normal(s + 2, (O,))).astype(np.float32) if has_bn else None
    res = gen.normal(s + 3, full) if has_res else None
    prelu = (0.25 * gen.uniform(s + 4, (O,))).astype(np.float32) if has_prelu else None
    pa = ((0.5 + gen.uniform(s + 5, (O,))) * np.where(np.arange(O) % 3 == 0, -1, 1)).astype(np.float32) if aff else None
    pb = (0.3 * gen.normal(s + 6, (O,))).astype(np.float32) if aff else None
    canvas = gen.normal(s + 7, full)
    alpha = pw.alpha.cpu().numpy()[:O]
    want, pv = oracle.fused_epilogue2(ref_dot, alpha, None, None, bn_a, bn_b, res, prelu, relu, res_late=late,
                                      pack_pre=pre, pack_a=pa, pack_b=pb, pack_relu=prelu_pack,
                                      out=canvas.copy(), c_off=c_off)
    opt = lambda a: None if a is None else dev(a)  # noqa: E731
    y, pk = hipops.bconv2d_fused(act, pw, bn_scale=opt(bn_a), bn_shift=opt(bn_b), residual=opt(res),
                                 prelu=opt(prelu), relu=relu, residual_after_act=late, pack_before_residual=pre,
                                 pack_scale=opt(pa), pack_shift=opt(pb), pack_relu=prelu_pack,
                                 out=dev(canvas) if sliced else None, out_c_offset=c_off, out_f32=True,
                                 out_packed=True, **kw)
    got = y.cpu().numpy()
    if not sliced:
        want = want[:, :O]
    assert np.array_equal(got, want)
    P, M = oracle.pack_act(pv)
    assert np.array_equal(u64(pk.P), P) and np.array_equal(u64(pk.M), M)
    if pk.nonneg:
        assert not M.any()
