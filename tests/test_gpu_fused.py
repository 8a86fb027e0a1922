"""GPU parity of the fused pieces (epilogue, avgpool+pack, weight-source variants): bit-exact
against the CPU oracle's op-for-op restatement, through the C-ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from bnn_amd import hipops, native
from tests.golden import gen
from tests.golden.cases import LAYER_CASES, LAYER_CASES_BY_NAME, LayerCase
from tests.helpers import legacy

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


TILED = [c for c in LAYER_CASES if c.k in (1, 3) and c.dilation == 1 and c.winit != "withzeros"]


@pytest.mark.parametrize("case", TILED, ids=lambda c: c.name)
def test_weight_source_variants_bit_exact(case):
    """Scalar-cache weight stream (the product kernel) and, for the 3x3 shapes, the LDS-staged weight tile (test-only
    since ABI 12: csrc/legacy/bconv_lds.hip through tests/helpers/legacy.py) compute the same integers."""
    x, w, b, sc = case.tensors()
    act = hipops.pack_act(dev(x))
    pw = hipops.pack_weight(dev(w), case.center, case.compute_alpha)
    dot = hipops.bconv2d(act, pw, stride=case.stride, padding=case.pad, raw_dot=True,
                         weights="sgpr").cpu().numpy()
    _, ref_dot = oracle.binary_conv2d_int(x, w, None, None, case.stride, case.pad, case.dilation,
                                          case.center, case.compute_alpha)
    assert np.array_equal(dot, ref_dot)
    if case.k == 3:
        got = legacy.bconv2d_lds(act, pw, stride=case.stride, padding=case.pad)
        want = hipops.bconv2d(act, pw, stride=case.stride, padding=case.pad)
        assert torch.equal(got, want)       # alpha * dot: same float bits


EPI_CASES = [
    # name of layer case, dict of epilogue switches
    ("c2_relu", dict(bn=True, relu=True, res=False, prelu=False)),
    ("c2_relu", dict(bn=True, relu=True, res=True, prelu=False)),
    ("l4_512x7", dict(bn=True, relu=True, res=True, prelu=False)),
    ("l2_0_c1_s2", dict(bn=True, relu=True, res=False, prelu=False)),
    ("l3_ds_1x1", dict(bn=True, relu=False, res=False, prelu=False)),
    ("tail_c96", dict(bn=True, relu=False, res=True, prelu=True)),       # O=40: ragged channel block
    ("tail_c200_o5", dict(bn=False, relu=True, res=True, prelu=False)),  # O=5
    ("center_bias_scale", dict(bn=True, relu=True, res=True, prelu=False)),
    ("k5_generic", dict(bn=True, relu=True, res=True, prelu=False)),     # generic kernel epilogue
    ("special_vals", dict(bn=True, relu=False, res=False, prelu=True)),
]


# multi-chunk shapes with ragged channel blocks: exercise the split-block waves (16-bit packed stores)
ADHOC = {c.name: c for c in [
    LayerCase("adhoc_c256_o40", 2, 256, 9, 7, 40, 3, 1, 1, act="relu"),
    LayerCase("adhoc_c512_o72_s2", 1, 512, 9, 9, 72, 3, 2, 1, act="relu"),
    LayerCase("adhoc_c384_o100", 1, 384, 6, 5, 100, 3, 1, 1, act="normal"),
]}
EPI_CASES += [
    ("adhoc_c256_o40", dict(bn=True, relu=True, res=True, prelu=False)),
    ("adhoc_c512_o72_s2", dict(bn=True, relu=True, res=False, prelu=False)),
    ("adhoc_c384_o100", dict(bn=True, relu=False, res=True, prelu=True)),
]


@pytest.mark.parametrize("nonneg", [False, True], ids=["two-plane", "nonneg"])
@pytest.mark.parametrize("name,sw", EPI_CASES, ids=[f"{n}-{i}" for i, (n, _) in enumerate(EPI_CASES)])
def test_fused_epilogue_bit_exact(name, sw, nonneg):
    case = LAYER_CASES_BY_NAME.get(name) or ADHOC[name]
    if nonneg and not (case.act == "relu" and case.k == 3):
        pytest.skip("P-plane-only kernels apply to 3x3 convs of non-negative inputs")
    x, w, b, sc = case.tensors()
    act = hipops.pack_act(dev(x))
    act.nonneg = nonneg
    pw = hipops.pack_weight(dev(w), case.center, case.compute_alpha)
    _, dot = oracle.binary_conv2d_int(x, w, None, None, case.stride, case.pad, case.dilation,
                                      case.center, case.compute_alpha)
    O = case.O
    s = gen.seed_of("epi", name)
    bn_a = (0.5 + gen.uniform(s, (O,))).astype(np.float32) if sw["bn"] else None
    bn_b = (0.3 * gen.normal(s + 1, (O,))).astype(np.float32) if sw["bn"] else None
    res = gen.normal(s + 2, dot.shape) if sw["res"] else None
    pre = (0.25 * gen.uniform(s + 3, (O,))).astype(np.float32) if sw["prelu"] else None
    alpha = pw.alpha.cpu().numpy()[:O]
    ref = oracle.fused_epilogue(dot, alpha, b, sc, bn_a, bn_b, res, pre, sw["relu"])
    opt = lambda a: None if a is None else dev(a)  # noqa: E731
    y, pk = hipops.bconv2d_fused(act, pw, bias=opt(b), post_scale=opt(sc), bn_scale=opt(bn_a),
                                 bn_shift=opt(bn_b), residual=opt(res), prelu=opt(pre),
                                 relu=sw["relu"], out_f32=True, out_packed=True,
                                 stride=case.stride, padding=case.pad, dilation=case.dilation)
    assert np.array_equal(y.cpu().numpy(), ref)
    P, M = oracle.pack_act(ref)                       # sign(y) re-packed for the next layer
    assert np.array_equal(u64(pk.P), P) and np.array_equal(u64(pk.M), M)
    # packed-only output (no fp32 store) produces the same planes
    _, pk2 = hipops.bconv2d_fused(act, pw, bias=opt(b), post_scale=opt(sc), bn_scale=opt(bn_a),
                                  bn_shift=opt(bn_b), residual=opt(res), prelu=opt(pre),
                                  relu=sw["relu"], out_f32=False, out_packed=True,
                                  stride=case.stride, padding=case.pad, dilation=case.dilation)
    assert torch.equal(pk2.P, pk.P) and torch.equal(pk2.M, pk.M)


PRE_CASES = [
    # (layer case, switches) — the pre-activation orders of PreBasicBlock / HBlock (ABI 3 epilogue)
    ("c2_relu", dict(act="relu", res=False, late=False, pre=False, aff=True, prelu_pack=False, slice_=None)),
    ("c2_relu", dict(act="prelu", res=True, late=True, pre=False, aff=True, prelu_pack=False, slice_=None)),
    ("l2_0_c1_s2", dict(act=None, res=True, late=True, pre=True, aff=True, prelu_pack=True, slice_=(32, 200))),
    ("adhoc_c256_o40", dict(act=None, res=True, late=True, pre=True, aff=True, prelu_pack=True, slice_=(3, 50))),
    ("adhoc_c512_o72_s2", dict(act="prelu", res=True, late=True, pre=False, aff=False, prelu_pack=False, slice_=(0, 72))),
    ("l3_ds_1x1", dict(act="relu", res=True, late=True, pre=True, aff=False, prelu_pack=False, slice_=(64, 320))),
    ("k5_generic", dict(act="relu", res=True, late=True, pre=False, aff=True, prelu_pack=True, slice_=(1, 20))),
]


@pytest.mark.parametrize("name,sw", PRE_CASES, ids=[f"{n}-{i}" for i, (n, _) in enumerate(PRE_CASES)])
def test_preactivation_epilogue_bit_exact(name, sw):
    case = LAYER_CASES_BY_NAME.get(name) or ADHOC[name]
    x, w, b, sc = case.tensors()
    act = hipops.pack_act(dev(x))
    pw = hipops.pack_weight(dev(w), case.center, case.compute_alpha)
    _, dot = oracle.binary_conv2d_int(x, w, None, None, case.stride, case.pad, case.dilation,
                                      case.center, case.compute_alpha)
    N, O = dot.shape[:2]
    s = gen.seed_of("pre", name)
    c_off, c_tot = sw["slice_"] if sw["slice_"] else (0, O)
    full_shape = (N, c_tot) + dot.shape[2:]
    res = gen.normal(s + 2, full_shape) if sw["res"] else None
    pre = (0.25 * gen.uniform(s + 3, (O,))).astype(np.float32) if sw["act"] == "prelu" else None
    pa = ((0.5 + gen.uniform(s + 4, (O,))) * np.where(np.arange(O) % 3 == 0, -1, 1)).astype(np.float32) if sw["aff"] else None
    pb = (0.3 * gen.normal(s + 5, (O,))).astype(np.float32) if sw["aff"] else None
    alpha = pw.alpha.cpu().numpy()[:O]
    canvas = gen.normal(s + 6, full_shape)                 # pre-existing content of the wider tensor
    ref_out, pv = oracle.fused_epilogue2(dot, alpha, b, sc, None, None, res, pre, sw["act"] == "relu",
                                         res_late=sw["late"], pack_pre=sw["pre"], pack_a=pa, pack_b=pb,
                                         pack_relu=sw["prelu_pack"], out=canvas.copy(), c_off=c_off)
    opt = lambda a: None if a is None else dev(a)  # noqa: E731
    out = dev(canvas) if sw["slice_"] else None
    y, pk = hipops.bconv2d_fused(act, pw, bias=opt(b), post_scale=opt(sc), residual=opt(res), prelu=opt(pre),
                                 relu=sw["act"] == "relu", residual_after_act=sw["late"],
                                 pack_before_residual=sw["pre"], pack_scale=opt(pa), pack_shift=opt(pb),
                                 pack_relu=sw["prelu_pack"], out=out, out_c_offset=c_off, out_f32=True,
                                 out_packed=True, stride=case.stride, padding=case.pad, dilation=case.dilation)
    assert np.array_equal(y.cpu().numpy(), ref_out)       # the slice is written, the rest of the canvas kept
    P, M = oracle.pack_act(pv)
    assert np.array_equal(u64(pk.P), P) and np.array_equal(u64(pk.M), M)
    if pk.nonneg:
        assert not M.any()


@pytest.mark.parametrize("shape,relu,bn", [((2, 64, 14, 14), True, True), ((1, 70, 9, 7), False, True),
                                            ((3, 200, 5, 5), True, False), ((2, 128, 8, 6), False, False)])
def test_bn_act_pack_matches_sign_of_affine(shape, relu, bn):
    """sign(act(bn(x))) in one pass == the torch sequence bn -> act -> sign the reference runs."""
    C = shape[1]
    x = gen.normal(gen.seed_of("bnpack", shape), shape)
    a = ((0.5 + gen.uniform(1, (C,))) * np.where(np.arange(C) % 4 == 0, -1, 1)).astype(np.float32) if bn else None
    b = (0.4 * gen.normal(2, (C,))).astype(np.float32) if bn else None
    pk = hipops.bn_act_pack(dev(x), None if a is None else dev(a), None if b is None else dev(b), relu)
    v = x.astype(np.float64)
    if bn:
        v = v * a.astype(np.float64).reshape(1, -1, 1, 1) + b.astype(np.float64).reshape(1, -1, 1, 1)
    if relu:
        v = np.maximum(v, 0)
    P, M = oracle.pack_act(np.sign(v).astype(np.float32))
    assert np.array_equal(u64(pk.P), P) and np.array_equal(u64(pk.M), M)
    assert pk.nonneg == relu


def test_two_fused_layers_equal_unfused_chain():
    """conv -> BN -> ReLU -> (packed) -> conv: packed hand-over == fp32 round trip."""
    c1, c2 = LAYER_CASES_BY_NAME["l2_128x28"], LAYER_CASES_BY_NAME["c2_relu"]
    x, w1, _, _ = c1.tensors()
    _, w2, _, _ = c2.tensors()
    pw1, pw2 = hipops.pack_weight(dev(w1)), hipops.pack_weight(dev(w2))
    bn_a = dev((0.5 + gen.uniform(5, (128,))).astype(np.float32))
    bn_b = dev((0.3 * gen.normal(6, (128,))).astype(np.float32))
    act = hipops.pack_act(dev(x))
    y, pk = hipops.bconv2d_fused(act, pw1, bn_scale=bn_a, bn_shift=bn_b, relu=True, out_f32=True,
                                 out_packed=True, stride=1, padding=1)
    a = hipops.bconv2d(pk, pw2, stride=1, padding=1)
    b = hipops.bconv2d(hipops.pack_act(y), pw2, stride=1, padding=1)
    assert torch.equal(a, b)


@pytest.mark.parametrize("shape,k", [((2, 64, 56, 56), 2), ((3, 128, 7, 7), 2), ((2, 70, 9, 5), 2),
                                     ((1, 256, 14, 14), 2), ((2, 32, 6, 6), 3)])
def test_avgpool_pack_matches_torch_pool_then_sign(shape, k):
    x = gen.activation("relu", gen.seed_of("ap", shape), shape)
    x[0, 0] = -x[0, 0]  # some negative windows too
    pk = hipops.avgpool_pack(dev(x), k)
    pooled = oracle.avgpool_ceil(x, k)
    t = F.avg_pool2d(torch.from_numpy(x), k, k, ceil_mode=True, count_include_pad=False).numpy()
    assert np.allclose(pooled, t, rtol=1e-6, atol=1e-7)     # the oracle restates the reference op
    P, M = oracle.pack_act(pooled)
    assert np.array_equal(u64(pk.P), P) and np.array_equal(u64(pk.M), M)
    P2, M2 = oracle.pack_act(t)                              # and sign(torch's pool) is identical
    assert np.array_equal(P, P2) and np.array_equal(M, M2)


# ----------------------------------------------------------------------- whole-network executor
import os  # noqa: E402

import torch.nn as nn  # noqa: E402

import bnn_amd as bnn  # noqa: E402
from bnn_amd import fastpath  # noqa: E402
from bnn_amd.inference import FusedResNet, FusionError, optimize_for_inference  # noqa: E402
from bnn_amd.models import resnet18, resnet50  # noqa: E402
from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer  # noqa: E402


@pytest.mark.parametrize("shape,k", [((2, 64, 56, 56), 2), ((3, 128, 7, 7), 2), ((2, 70, 9, 5), 2), ((1, 200, 13, 10), 3),
                                     ((4, 512, 14, 14), 2), ((1, 64, 1, 1), 2)])
def test_orpool_of_sign_planes_equals_avgpool_then_sign_for_nonnegative_inputs(shape, k):
    """bnn_hip_orpool_packed: for x >= 0, sign(AvgPool_k(x)) (ceil mode, clipped windows) == OR of the P plane — from
    the planes alone, bit-exact against torch's pool + sign AND against the fp32-reading avgpool_pack kernel."""
    x = dev(gen.activation("relu", gen.seed_of("orpool", shape), shape))
    x.view(-1)[::5] = 0.0
    act = hipops.pack_act(x)
    act.nonneg = True
    got = hipops.orpool_packed(act, k)
    ref = torch.sign(F.avg_pool2d(x, k, k, 0, ceil_mode=True, count_include_pad=False))
    P, M = oracle.pack_act(ref.cpu().numpy())
    assert np.array_equal(u64(got.P), P) and np.array_equal(u64(got.M), M) and got.nonneg
    via_f32 = hipops.avgpool_pack(x, k, nonneg=True)
    assert torch.equal(via_f32.P, got.P) and torch.equal(via_f32.M, got.M)
    act.nonneg = False
    with pytest.raises(native.NativeError):
        hipops.orpool_packed(act, k)


def _r18(activation=None):
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    kw = {} if activation is None else {"activation": activation}
    net = bnn.prepare_binary_model(resnet18(**kw), cfg, custom_config_layers_name={
        "conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 1).items()})
    return net.to(DEV).eval()


@pytest.mark.parametrize("tag,shape", [("32", (4, 3, 32, 32)), ("64", (2, 3, 64, 64)), ("224", (2, 3, 224, 224))])
def test_fused_resnet18_matches_reference_logits(tag, shape):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "resnet18.npz"))
    net = _r18()
    fused = FusedResNet(net)
    x = dev(gen.normal(gen.seed_of("r18", tag), shape))
    before = fastpath.stats()["conv2d"]
    y = fused(x).cpu().numpy()
    assert fastpath.stats()["conv2d"] == before       # the fused path bypasses the per-layer path
    ref = g["logits_" + tag]
    assert np.allclose(y, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
    assert (y.argmax(1) == ref.argmax(1)).all()
    with torch.no_grad():
        y_layerwise = net(x).cpu().numpy()
    assert np.allclose(y, y_layerwise, rtol=1e-3, atol=1e-3 * np.abs(ref).max())


@pytest.mark.parametrize("activation", [None, nn.PReLU], ids=["relu", "prelu"])
def test_fused_resnet50_bottleneck_matches_layerwise_and_cpu(activation):
    """Bottleneck stages (1x1 -> 3x3(stride) -> 1x1): activations stay packed across the three
    binary convs; only the block output is written in fp32."""
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    kw = {} if activation is None else {"activation": activation}
    net = bnn.prepare_binary_model(resnet50(num_classes=64, **kw), cfg,
                                   custom_config_layers_name={"conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 2).items()})
    net.eval()
    x = torch.from_numpy(gen.normal(gen.seed_of("r50"), (2, 3, 96, 96)))
    with torch.no_grad():
        ref = net(x).numpy()                                   # CPU: torch composition
    net = net.to(DEV)
    fused = FusedResNet(net)
    assert len(fused._blocks) == 16 and all(len(b["convs"]) == 3 for b in fused._blocks)
    before = fastpath.stats()["conv2d"]
    y = fused(x.to(DEV)).cpu().numpy()
    assert fastpath.stats()["conv2d"] == before
    with torch.no_grad():
        lw = net(x.to(DEV)).cpu().numpy()                      # per-layer HIP path
    assert np.allclose(y, lw, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
    assert np.allclose(y, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())


# (HBlock + PReLU cannot run in the reference either: act1 has planes/2 parameters but sees inplanes
#  channels — hierarchical_block.py:33,41 — and the block here keeps that definition.)
@pytest.mark.parametrize("kind", ["pre_relu", "pre_prelu", "hblock_relu"])
def test_fused_preactivation_and_hierarchical_nets(kind):
    """PreBasicBlock (examples/imagenet.py dataflow) and HBlock nets in the fused executor: BN -> sign of the
    next layer folded into the producing conv's epilogue, torch.cat written in place."""
    from bnn_amd.models import HBlock, PreBasicBlock, ResNet
    from bnn_amd.ops import BasicScaleBinarizer
    act = nn.PReLU if kind.endswith("prelu") else nn.ReLU
    if kind.startswith("pre"):
        net = resnet18(block_type=PreBasicBlock, activation=act, num_classes=50)
    else:
        net = ResNet(HBlock, [2, 2, 1, 1], activation=act, num_classes=50)
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=BasicScaleBinarizer,
                      weight_pre_process=XNORWeightBinarizer)
    net = bnn.prepare_binary_model(net, cfg, custom_config_layers_name={"conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 4).items()})
    net.eval()
    x = torch.from_numpy(gen.normal(gen.seed_of("prenet", kind), (2, 3, 96, 96)))
    with torch.no_grad():
        ref = net(x).numpy()                                   # CPU: torch composition
    net = net.to(DEV)
    fused = FusedResNet(net)
    kinds = {b["kind"] for b in fused._blocks}
    assert kinds == ({"pre"} if kind.startswith("pre") else {"h", "pool"})
    before = fastpath.stats()["conv2d"]
    y = fused(x.to(DEV)).cpu().numpy()
    assert fastpath.stats()["conv2d"] == before                # nothing went through the per-layer path
    with torch.no_grad():
        lw = net(x.to(DEV)).cpu().numpy()
    assert np.allclose(y, lw, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
    if kind != "pre_relu":
        assert np.allclose(y, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
    # pre_relu: t = relu(alpha*dot) + identity feeds the shortcut's sign(avgpool(t)) un-normalised; where the
    # integer dot is 0 this path has t == 0 exactly while any float conv leaves +-1e-9 there (DESIGN.md
    # "exact zeros"): the float composition is not reproducible on such a net, so only the two HIP paths
    # are compared.
    fused.capture(x.to(DEV))                                   # and the whole thing replays as a HIP graph
    assert np.array_equal(fused(x.to(DEV)).cpu().numpy(), y)


def test_fused_resnet18_graph_replay_is_bit_identical():
    net = _r18()
    fused = FusedResNet(net)
    x = dev(gen.normal(5, (8, 3, 64, 64)))
    y0 = fused(x).clone()
    fused.capture(x)
    y1 = fused(x).clone()
    y2 = fused(dev(gen.normal(6, (8, 3, 64, 64)))).clone()
    assert torch.equal(y0, y1) and not torch.equal(y1, y2)
    assert torch.equal(fused(x), y0)
    # zero-copy replay: the caller fills the graph's own input buffer in place
    buf = fused.static_input
    assert buf is not None and buf.shape == x.shape and buf.data_ptr() != x.data_ptr()
    x3 = dev(gen.normal(7, (8, 3, 64, 64)))
    buf.copy_(x3)
    y3 = fused(buf).clone()
    fresh = FusedResNet(net)
    assert torch.equal(y3, fresh(x3)) and not torch.equal(y3, y0)


def test_pipelined_inference_two_streams_equals_single_stream():
    from bnn_amd.inference import PipelinedInference
    net = _r18()
    xs = [dev(gen.normal(40 + i, (4, 3, 64, 64))) for i in range(5)]
    single = FusedResNet(net)
    want = [single(x).clone() for x in xs]
    pipe = PipelinedInference(net, xs[0], n_streams=2)
    assert len(pipe) == 2 and pipe.input(0).data_ptr() != pipe.input(1).data_ptr()
    got = []
    for i, x in enumerate(xs):
        with torch.cuda.stream(pipe.stream(i)):
            pipe.input(i).copy_(x)
        y = pipe.launch(i)
        with torch.cuda.stream(pipe.stream(i)):
            got.append(y.clone())                      # the slot's buffer is reused two launches later
    pipe.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_fused_executor_follows_weight_changes():
    """Packed weights / folded BN are derived data: an in-place update or load_state_dict of the wrapped model is
    noticed (parameter version counters) and the executor — and its captured graph — are rebuilt."""
    net = _r18()
    x = dev(gen.normal(12, (2, 3, 64, 64)))
    fused = FusedResNet(net)
    y0 = fused(x).clone()
    with torch.no_grad():
        net.layer2[0].conv1.weight.mul_(-1.0)          # flips every sign of one binary layer
        net.layer1[0].bn1.running_mean.add_(0.25)
    y1 = fused(x).clone()
    assert not torch.equal(y0, y1) and torch.equal(y1, FusedResNet(net)(x))
    fused.capture(x)
    assert torch.equal(fused(x), y1)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    sd["layer3.1.conv2.weight"] = -sd["layer3.1.conv2.weight"]
    net.load_state_dict(sd)
    y2 = fused(x).clone()                              # graph re-captured behind the scenes
    assert fused._graph is not None and not torch.equal(y2, y1) and torch.equal(y2, FusedResNet(net)(x))


def test_fused_resnet18_prelu_variant():
    net = _r18(activation=nn.PReLU)
    fused = FusedResNet(net)
    x = dev(gen.normal(9, (2, 3, 64, 64)))
    with torch.no_grad():
        ref = net(x)
    assert torch.allclose(fused(x), ref, rtol=1e-3, atol=1e-3 * float(ref.abs().max()))


def test_unsupported_models_are_left_alone():
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    dab = bnn.prepare_binary_model(resnet18(stem_type="dabnn"), cfg).to(DEV).eval()
    assert isinstance(optimize_for_inference(dab), FusedResNet)   # round 4: the stem runs as modules, the blocks fused
    inorm = bnn.prepare_binary_model(resnet18(norm_layer=lambda c: torch.nn.InstanceNorm2d(c, affine=True)), cfg)
    inorm = inorm.to(DEV).eval()
    assert optimize_for_inference(inorm) is inorm      # no BatchNorm to fold: per-layer path only
    with pytest.raises(FusionError):
        FusedResNet(_r18().train())
    with pytest.raises(FusionError):
        FusedResNet(resnet18().to(DEV).eval())          # not binarised


@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (1, 70, 11, 9), (3, 32, 7, 7)])
def test_stem_tail_matches_torch_sequence(shape):
    """BN(eval) -> ReLU -> MaxPool(3,2,1) -> sign, fused, vs the torch ops the reference runs."""
    C = shape[1]
    x = dev(gen.normal(gen.seed_of("stem", shape), shape) * 3)
    a = dev((0.5 + gen.uniform(1, (C,))).astype(np.float32) * np.where(np.arange(C) % 5 == 0, -1, 1).astype(np.float32))
    b = dev((0.3 * gen.normal(2, (C,))).astype(np.float32))
    y, pk = hipops.bn_relu_maxpool_pack(x, a, b, True, 3, 2, 1)
    ref = F.max_pool2d(F.relu(x * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)), 3, 2, 1)
    assert torch.allclose(y, ref, rtol=1e-6, atol=1e-6)
    P, M = oracle.pack_act(y.cpu().numpy())
    assert np.array_equal(u64(pk.P), P) and np.array_equal(u64(pk.M), M)
    y2, _ = hipops.bn_relu_maxpool_pack(x, None, None, False, 2, 2, 0, out_packed=False)
    assert torch.equal(y2, F.max_pool2d(x, 2, 2, 0))


def test_stem_fp16_option_is_half_precision_accurate():
    """BNN_HIP_STEM_FP16 (BASELINE config 5's "fp16 MFMA stem"): opt-in, one MFMA per product."""
    shape = (2, 3, 96, 96)
    x = dev(gen.normal(gen.seed_of("stem16", shape), shape))
    w = dev(gen.conv_weight("kaiming", 3, (64, 3, 7, 7)))
    a = dev((0.5 + gen.uniform(1, (64,))).astype(np.float32))
    b = dev((0.3 * gen.normal(2, (64,))).astype(np.float32))
    y16, _ = hipops.stem7x7(x, w, a, b, fp16=True)
    y, _ = hipops.stem7x7(x, w, a, b)
    err = float((y16 - y).abs().max() / y.abs().max())
    assert 1e-6 < err < 3e-3          # genuinely the cheaper arithmetic, and within fp16's class


@pytest.mark.parametrize("exact", [False, True], ids=["f16x3", "fp32"])
@pytest.mark.parametrize("shape", [(2, 3, 224, 224), (3, 3, 64, 64), (1, 3, 32, 32), (2, 3, 50, 38), (1, 3, 97, 131),
                                   (5, 3, 33, 65), (1, 3, 7, 9)])
def test_mfma_stem_matches_torch_fp32_sequence(shape, exact):
    """conv7x7/2 -> BN -> ReLU -> MaxPool(3,2,1) on the matrix cores vs the torch ops.
    Default arithmetic: fp16 hi/lo split (3 MFMAs, fp32 accumulate); exact: the fp32 MFMA."""
    x = dev(gen.normal(gen.seed_of("stemx", shape), shape))
    w = dev(gen.conv_weight("kaiming", 3, (64, 3, 7, 7)))
    a = dev((0.5 + gen.uniform(1, (64,))).astype(np.float32) * np.where(np.arange(64) % 7 == 0, -1, 1).astype(np.float32))
    b = dev((0.3 * gen.normal(2, (64,))).astype(np.float32))
    y, pk = hipops.stem7x7(x, w, a, b, exact_fp32=exact)
    conv = F.conv2d(x.double(), w.double(), None, 2, 3)        # fp64 reference of the fp32 conv
    ref = F.max_pool2d(F.relu(conv * a.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)), 3, 2, 1)
    assert y.shape == ref.shape
    assert torch.allclose(y.double(), ref, rtol=1e-5, atol=2e-6 * float(ref.abs().max()))
    P, M = oracle.pack_act(y.cpu().numpy())
    assert np.array_equal(u64(pk.P), P) and np.array_equal(u64(pk.M), M)
    # and it agrees with the library fp32 conv + the fused tail kernel to rounding
    y2, _ = hipops.bn_relu_maxpool_pack(F.conv2d(x, w, None, 2, 3), a, b, True, 3, 2, 1, out_packed=False)
    assert torch.allclose(y, y2, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))


@pytest.mark.parametrize("fp16", [False, True], ids=["f16x3", "f16"])
@pytest.mark.parametrize("shape", [(2, 3, 224, 224), (3, 3, 64, 64), (1, 3, 32, 32), (2, 3, 50, 38), (1, 3, 97, 131),
                                   (5, 3, 33, 65), (1, 3, 7, 9), (1, 3, 225, 223), (300, 3, 64, 64), (129, 3, 224, 224)])
def test_the_two_stem_kernels_agree_bit_for_bit(shape, fp16):
    """The default kernel (transposed GEMM, max-pool in the accumulators, strips walked down the image with the last
    conv row kept in registers; stem_rows.hip) against the round-2 kernel (conv tile staged through LDS; test-only
    since ABI 12: csrc/legacy/stem_split.hip): two independent tilings of the same arithmetic, every fp32 value and
    sign bit equal.
    The last two shapes have more strips than workgroups (whole strips round-robin, a partial last round), the
    small ones fewer tiles than workgroups (one chunk each, chunks starting inside a strip)."""
    x = dev(gen.normal(gen.seed_of("stem2", shape), (min(shape[0], 6),) + shape[1:]))
    x = x.repeat((shape[0] + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:shape[0]].contiguous()
    w = dev(gen.conv_weight("kaiming", 3, (64, 3, 7, 7)))
    a = dev((0.5 + gen.uniform(1, (64,))).astype(np.float32) * np.where(np.arange(64) % 7 == 0, -1, 1).astype(np.float32))
    b = dev((0.3 * gen.normal(2, (64,))).astype(np.float32))
    y0, p0 = legacy.stem_staged(x, w, a, b, fp16=fp16)
    y1, p1 = hipops.stem7x7(x, w, a, b, fp16=fp16)
    assert torch.equal(y0, y1) and torch.equal(p0.P, p1.P) and torch.equal(p0.M, p1.M)
    y2, _ = hipops.stem7x7(x, w, a, b, fp16=fp16, out_packed=False)
    _, p3 = hipops.stem7x7(x, w, a, b, fp16=fp16, out_f32=False)
    assert torch.equal(y2, y1) and torch.equal(p3.P, p1.P)
    lib = native.require()
    for flags in (8, native.STEM_EXACT_FP32 | native.STEM_FP16):     # 8: BNN_HIP_STEM_STAGED of ABI <= 11, gone
        bad = lib.bnn_hip_stem7x7_bn_relu_pool_pack_f32(x.data_ptr(), w.data_ptr(), a.data_ptr(), b.data_ptr(), 1, 7,
                                                        9, flags, y1.data_ptr(), None, None, None)
        assert bad == -1      # BNN_HIP_ERR_INVALID_ARG
    assert legacy.stem_staged(x[:1], w, a, b, flags=native.STEM_EXACT_FP32) == -1   # no exact-fp32 mode there


def test_fused_resnet_with_and_without_mfma_stem_agree():
    net = _r18()
    x = dev(gen.normal(gen.seed_of("r18", "64"), (2, 3, 64, 64)))
    a = FusedResNet(net, use_mfma_stem=True)
    b = FusedResNet(net, use_mfma_stem=False)
    assert a._stem_mfma and not b._stem_mfma
    ya, yb = a(x), b(x)
    assert torch.allclose(ya, yb, rtol=1e-3, atol=1e-3 * float(yb.abs().max()))


@pytest.mark.parametrize("shape,O", [((256, 512, 7, 7), 1000), ((3, 512, 7, 7), 1000), ((9, 70, 3, 5), 13),
                                     ((1, 2048, 7, 7), 1000), ((17, 64, 1, 1), 300), ((33, 6, 7, 7), 65),
                                     ((2, 3000, 2, 2), 10)])
def test_head_kernel_matches_fp64_avgpool_fc(shape, O):
    """bnn_hip_avgpool_fc_f32 (avgpool -> flatten -> fc of bnn/models/resnet.py:160-164 in one kernel) against the
    same computation in fp64: fp32 accumulation over C terms -> 2e-6 of the largest logit."""
    N, C, H, W = shape
    x = dev(np.maximum(gen.normal(gen.seed_of("head", shape), shape), 0))
    w = dev(gen.conv_weight("default", 5, (O, C)))
    b = dev((0.1 * gen.normal(6, (O,))).astype(np.float32))
    y = hipops.avgpool_fc(x, w.t().contiguous(), b)
    ref = (x.double().mean((2, 3)) @ w.double().t() + b.double())
    assert y.shape == (N, O)
    assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    y2 = hipops.avgpool_fc(x, w.t().contiguous(), None)
    assert torch.allclose(y2.double(), ref - b.double(), rtol=0, atol=2e-6 * float(ref.abs().max()))
    lib_y = F.linear(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1), w, b)          # what the reference runs
    assert torch.allclose(y, lib_y, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    # round 5: `y` came from the two-launch head (bnn_hip_avgpool_fc_ws_f32); the one-kernel form computes the same
    # means and sums the product in another order
    y1 = hipops.avgpool_fc(x, w.t().contiguous(), b, one_kernel=True)
    assert float((y1.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    assert torch.allclose(y, y1, rtol=0, atol=2e-6 * float(ref.abs().max()))
    # an image's logits depend neither on the batch size nor on its position in the batch (16-image groups inside)
    for lo, hi in ((0, 1), (N // 2, N), (max(N - 3, 0), N)):
        if hi > lo:
            assert torch.equal(hipops.avgpool_fc(x[lo:hi].contiguous(), w.t().contiguous(), b), y[lo:hi])
    if N > 1:
        perm = torch.arange(N - 1, -1, -1, device=x.device)
        assert torch.equal(hipops.avgpool_fc(x[perm].contiguous(), w.t().contiguous(), b), y[perm])


@pytest.mark.parametrize("shape", [(3, 64, 14, 14, 64, 1), (2, 64, 12, 10, 128, 2), (2, 128, 9, 9, 96, 1),
                                   (2, 256, 7, 7, 256, 1), (1, 512, 7, 7, 64, 1), (2, 70, 6, 5, 40, 1)],
                         ids=lambda s: "x".join(map(str, s)))
def test_integer_sign_thresholds_give_the_float_epilogue_bits(shape):
    """bnn_hip_sign_thresholds_f32: BN + ReLU + sign of a conv1-type layer as an integer interval test on the dot.
    The planes must equal the float epilogue's bit for bit — including channels with a negative, zero, huge, tiny
    or NaN BatchNorm constant and all-zero weights (alpha = 0) — and the table must reproduce the float predicate on
    every dot the layer produced."""
    N, C, H, W, O, stride = shape
    x = dev(np.maximum(gen.normal(gen.seed_of("thrx", shape), (N, C, H, W)), 0))
    w = gen.conv_weight("kaiming", gen.seed_of("thrw", shape), (O, C, 3, 3))
    w[3] = 0.0                                                    # a channel whose alpha is 0
    bn_a = (0.5 + gen.uniform(1, (O,))).astype(np.float32) * np.where(np.arange(O) % 3 == 0, -1, 1).astype(np.float32)
    bn_b = (2.0 * gen.normal(2, (O,))).astype(np.float32)
    bn_a[1], bn_b[1] = 0.0, 0.25                                  # constant positive
    bn_a[2], bn_b[2] = 0.0, -0.25                                 # constant negative
    bn_a[5], bn_b[5] = 1e30, 1.0                                  # overflow to +-inf
    bn_a[6], bn_b[6] = 1e-30, -1e-38                              # denormal products
    bn_b[7] = np.nan                                              # never positive
    bn_a[8], bn_b[8] = 1.0, 0.0                                   # the threshold sits exactly at dot = 0
    pw = hipops.pack_weight(dev(w))
    act = hipops.pack_act(x)
    act.nonneg = True
    a_t, b_t = dev(bn_a), dev(bn_b)
    thr = hipops.sign_thresholds(pw, a_t, b_t)
    kw = dict(bn_scale=a_t, bn_shift=b_t, relu=True, out_f32=False, out_packed=True, stride=stride, padding=1)
    _, ref = hipops.bconv2d_fused(act, pw, **kw)
    _, got = hipops.bconv2d_fused(act, pw, sign_thresholds=thr, **kw)
    assert torch.equal(got.P, ref.P) and torch.equal(got.M, ref.M) and not bool(got.M.any())
    # the table against the float predicate on the dots themselves
    dot = hipops.bconv2d(act, pw, stride=stride, padding=1, raw_dot=True)                      # int32 [N,O,Ho,Wo]
    o_pad = (O + 31) // 32 * 32
    assert thr.shape == (o_pad, 4)                              # whole 32-channel blocks (include/bnn_hip.h)
    T = thr[:O, 0].view(1, -1, 1, 1).long()                     # bit = (dot >= T) XOR flip
    ch = torch.arange(O, device=thr.device)
    fword = thr[ch - ch % 32, 1].long() & 0xFFFFFFFF            # word 1 of the block's first (even) channel: the flips
    flip = (fword >> (ch % 32)) & 1
    # word 1: even channels repeat the block's flip word, odd channels its parity word (bit k = T of channel k is odd)
    allc = torch.arange(o_pad, device=thr.device)
    pword = thr[allc - allc % 32 + 1, 1].long() & 0xFFFFFFFF
    assert torch.equal((pword >> (allc % 32)) & 1, thr[:, 0].long() & 1)
    assert all(int(thr[o, 1]) == int(thr[o - o % 32 + (o & 1), 1]) for o in range(o_pad))
    assert bool((thr[O:, 0] == 0x40000000).all())               # pad channels: "never"
    # words 2, 3: the comparands of the two-instruction form (csrc/bconv_core.h midt2_shift_in)
    Tl = thr[:, 0].long()
    assert torch.equal(thr[:, 2].long(), (Tl + 1 + (2 << 20)) >> 1)
    assert torch.equal(thr[:, 3].long(), torch.clamp((2 + (2 << 20) - Tl) >> 1, min=0))
    bit = (dot.long() >= T) ^ flip.view(1, -1, 1, 1).bool()
    P, _ = oracle.pack_act(np.where(bit.cpu().numpy(), 1.0, 0.0).astype(np.float32))
    assert np.array_equal(u64(ref.P), P)
    # the disagreement form of the test: the same layer on TWO planes (an input with negative values)
    x2 = dev(gen.normal(gen.seed_of("thrx2", shape), (N, C, H, W)) * (gen.uniform(9, (N, C, H, W)) > 0.2))
    act2 = hipops.pack_act(x2)
    assert bool(act2.M.any()) and not getattr(act2, "nonneg", False)
    _, ref2 = hipops.bconv2d_fused(act2, pw, **kw)
    _, got2 = hipops.bconv2d_fused(act2, pw, sign_thresholds=thr, **kw)
    assert torch.equal(got2.P, ref2.P) and torch.equal(got2.M, ref2.M)


def test_fused_resnet_with_integer_thresholds_is_bit_identical():
    net = _r18()
    x = dev(gen.normal(43, (4, 3, 96, 96)))
    a, b = FusedResNet(net), FusedResNet(net, int_thresholds=False)
    assert a._blocks[0]["convs"][0].thr is not None and b._blocks[0]["convs"][0].thr is None
    assert torch.equal(a(x), b(x))


def test_throughput_mode_changes_the_kernel_choice_not_the_result():
    """BNN_HIP_FLAG_THROUGHPUT (PipelinedInference with several batches in flight): the multi-chunk 3x3 kernels keep a
    32-channel block in one wave instead of splitting it — bit-identical logits."""
    net = _r18()
    x = dev(gen.normal(41, (8, 3, 96, 96)))
    a, b = FusedResNet(net), FusedResNet(net, throughput_mode=True)
    assert b._blocks[0]["convs"][0].throughput and not a._blocks[0]["convs"][0].throughput
    assert torch.equal(a(x), b(x))


def test_fused_executor_uses_the_head_kernel_and_no_library_gemm():
    net = _r18()
    fused = FusedResNet(net)
    assert fused._head is not None and fused._head[0].shape == (512, 1000)
    x = dev(gen.normal(31, (4, 3, 64, 64)))
    before = native.launch_count()
    y = fused(x)
    # stem + 16 convs (the 3 shortcut convs AND, since ABI 12, the OR-pools of their inputs folded into the last conv of
    # their block) + head
    assert native.launch_count() - before == 1 + 16 + 2      # stem, convs, head (average pool + product)
    unfolded = FusedResNet(net, fold_shortcut=False)    # (building an executor packs the weights: launches too)
    before = native.launch_count()
    y2 = unfolded(x)
    assert native.launch_count() - before == 1 + 16 + 3 + 3 + 2    # + 3 shortcut convs as launches of their own
    assert torch.equal(y, y2)
    with torch.no_grad():
        assert torch.allclose(y, net(x), rtol=1e-3, atol=1e-3 * float(y.abs().max()))


@pytest.mark.parametrize("throughput", [False, True], ids=["latency", "throughput"])
@pytest.mark.parametrize("shape", [(3, 64, 128, 29, 27), (2, 128, 256, 14, 14), (2, 256, 512, 7, 7), (1, 64, 96, 9, 11),
                                   (70, 128, 256, 14, 14)],
                         ids=lambda s: "x".join(map(str, s)))
def test_folded_shortcut_conv_equals_the_two_launch_form(shape, throughput):
    """bnn_hip_epilogue.sc_* (ABI 11): the residual of a down-sampling block's last conv computed in the kernel from the
    OR-pooled sign planes and the packed 1x1 weight — against the stand-alone 1x1 conv + BN launch whose fp32 output
    is then read as `residual`.  Same float operations: every fp32 value and every sign bit equal.  Shapes: the three
    ResNet-18 stages (single-chunk, two and four chunks; 64 / 128 / 256 shortcut channels), an output-channel tail
    (96 = 3 blocks), a batch with more waves than the block split takes."""
    N, Cs, O, H, W = shape
    c_mid = O                                   # conv2 of a BasicBlock: O -> O
    a2 = hipops.pack_act(dev(gen.activation("relu", gen.seed_of("fold", shape), (N, c_mid, H, W)))); a2.nonneg = True
    sc = hipops.pack_act(dev(gen.activation("relu", gen.seed_of("fold", shape) + 1, (N, Cs, H, W)))); sc.nonneg = True
    w2 = hipops.pack_weight(dev(gen.conv_weight("kaiming", 21, (O, c_mid, 3, 3))))
    ws = hipops.pack_weight(dev(gen.conv_weight("kaiming", 22, (O, Cs, 1, 1))))
    bn = lambda s: (dev((0.5 + gen.uniform(s, (O,))).astype(np.float32) * np.where(np.arange(O) % 5 == 0, -1, 1).astype(np.float32)),
                    dev((0.3 * gen.normal(s + 1, (O,))).astype(np.float32)))
    (a2s, b2s), (ass, bss) = bn(31), bn(41)
    assert hipops.shortcut_fold_supported(a2, w2, Cs, 1, 1, 1, throughput=throughput)
    idn, _ = hipops.bconv2d_fused(sc, ws, bn_scale=ass, bn_shift=bss, out_f32=True, out_packed=False)
    kw = dict(bn_scale=a2s, bn_shift=b2s, relu=True, stride=1, padding=1, out_f32=True, out_packed=True,
              throughput=throughput)
    y0, p0 = hipops.bconv2d_fused(a2, w2, residual=idn, **kw)
    y1, p1 = hipops.bconv2d_fused(a2, w2, shortcut=(sc, ws, ass, bss), **kw)
    assert torch.equal(y0, y1) and torch.equal(p0.P, p1.P) and torch.equal(p0.M, p1.M) and p1.nonneg
    # ABI 12 (sc_in_hw): the shortcut planes given UN-POOLED at twice the resolution (even and odd sizes: ceil-mode
    # windows cut by the border) — the kernel ORs the 2 x 2 windows itself, no bnn_hip_orpool_packed launch
    for hh, ww in ((2 * H, 2 * W), (2 * H - 1, 2 * W), (2 * H, 2 * W - 1), (2 * H - 1, 2 * W - 1)):
        big = hipops.pack_act(dev(gen.activation("relu", gen.seed_of("fold", shape) + 2, (N, Cs, hh, ww)))); big.nonneg = True
        pooled = hipops.orpool_packed(big, 2)
        assert tuple(pooled.shape) == (N, Cs, H, W)
        ya, pa = hipops.bconv2d_fused(a2, w2, shortcut=(pooled, ws, ass, bss), **kw)
        launches = native.launch_count()
        yb, pb = hipops.bconv2d_fused(a2, w2, shortcut=(big, ws, ass, bss), **kw)
        assert native.launch_count() == launches + 1
        assert torch.equal(ya, yb) and torch.equal(pa.P, pb.P)


def test_folded_shortcut_is_refused_where_no_kernel_takes_it():
    a = hipops.pack_act(dev(gen.activation("normal", 3, (2, 64, 12, 12))))          # signed activations: two planes
    sc = hipops.pack_act(dev(gen.activation("relu", 4, (2, 64, 12, 12)))); sc.nonneg = True
    w2 = hipops.pack_weight(dev(gen.conv_weight("kaiming", 21, (64, 64, 3, 3))))
    ws = hipops.pack_weight(dev(gen.conv_weight("kaiming", 22, (64, 64, 1, 1))))
    one = dev(np.ones(64, np.float32))
    assert not hipops.shortcut_fold_supported(a, w2, 64, 1, 1, 1)
    with pytest.raises(native.NativeError):
        hipops.bconv2d_fused(a, w2, bn_scale=one, bn_shift=one, relu=True, stride=1, padding=1, out_f32=True,
                             out_packed=True, shortcut=(sc, ws, one, one))
    a.nonneg = True                                                                    # (a promise; relu-free data)
    w1 = hipops.pack_weight(dev(gen.conv_weight("kaiming", 23, (64, 64, 1, 1))))
    assert not hipops.shortcut_fold_supported(a, w1, 64, 1, 0, 1)                      # a 1x1 last conv (Bottleneck)
    assert not hipops.shortcut_fold_supported(a, w2, 96, 1, 1, 1)                      # 96 shortcut channels


def test_dead_fp32_outputs_are_not_written_and_nothing_changes():
    """The last conv of layer1/2/3 feeds a block that reads sign planes only (convs AND the AvgPool -> binary 1x1
    shortcut): its fp32 tensor is skipped (compiled profile BN + residual + ReLU -> planes).  Same logits, bit for bit,
    for the ResNet-18 / 34 block pattern, odd spatial sizes (ceil-mode pooling) and a ragged batch."""
    for net, shape in ((_r18(), (4, 3, 96, 96)), (_r18(), (3, 3, 75, 61))):
        x = dev(gen.normal(47, shape))
        a, b = FusedResNet(net), FusedResNet(net, skip_dead_f32=False)
        assert torch.equal(a(x), b(x))
    # the epilogue itself, against the same call with the fp32 tensor switched on
    act = hipops.pack_act(dev(gen.activation("relu", 5, (2, 64, 20, 20)))); act.nonneg = True
    pw = hipops.pack_weight(dev(gen.conv_weight("kaiming", 6, (64, 64, 3, 3))))
    res = dev(gen.normal(7, (2, 64, 20, 20)))
    kw = dict(bn_scale=dev(gen.normal(8, (64,))) * 0.2 + 1.0, bn_shift=dev(gen.normal(9, (64,))) * 0.3, relu=True,
              residual=res, stride=1, padding=1, out_packed=True)
    y, full = hipops.bconv2d_fused(act, pw, out_f32=True, **kw)
    none, only = hipops.bconv2d_fused(act, pw, out_f32=False, **kw)
    assert none is None and torch.equal(only.P, full.P) and torch.equal(only.M, full.M) and only.nonneg
    assert torch.equal(only.P, hipops.pack_act(y).P)


def test_images_beyond_the_24_bit_index_range_take_the_generic_kernel():
    """The tiled kernels multiply indices with 24-bit multiplies; launch_bconv() sends anything with an index factor
    of 2^23 or more (here Ho*Wo = 4100^2 > 2^24, second image: n*Ho*Wo needs the full multiply) to the shape-generic
    kernel.  Size-independent check: a 3x3 convolution is local, so strips of the big result must equal the result
    of the same layer on the strips alone (which the tiled kernels compute)."""
    H = W = 4100
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(2, 64, H, W, device="cuda", generator=g)
    pw = hipops.pack_weight(dev(gen.conv_weight("kaiming", 3, (16, 64, 3, 3))))   # N*O*Ho*Wo stays below 2^30
    kw = dict(bn_scale=dev(gen.normal(4, (16,))) * 0.2 + 1.0, bn_shift=dev(gen.normal(5, (16,))) * 2.0, relu=True,
              stride=1, padding=1, out_f32=False, out_packed=True)
    _, big = hipops.bconv2d_fused(hipops.pack_act(x), pw, **kw)
    for n, rows in ((0, slice(0, 40)), (1, slice(H - 40, H)), (1, slice(2000, 2040))):
        strip = x[n:n + 1, :, rows].contiguous()
        _, small = hipops.bconv2d_fused(hipops.pack_act(strip), pw, **kw)
        r0 = rows.start
        lo, hi = (0 if r0 == 0 else 1), (40 if rows.stop == H else 39)       # rows whose 3x3 window lies inside the strip
        assert torch.equal(big.P[n:n + 1, :, r0 + lo:r0 + hi], small.P[:, :, lo:hi])
        assert torch.equal(big.M[n:n + 1, :, r0 + lo:r0 + hi], small.M[:, :, lo:hi])
    assert bool(big.P.any()) and not bool(big.M.any())


@pytest.mark.parametrize("fp16", [False, True], ids=["f16x3", "f16"])
@pytest.mark.parametrize("shape", [(2, 3, 224, 224), (3, 3, 64, 64), (1, 3, 32, 32), (2, 3, 50, 38), (1, 3, 97, 131),
                                   (1, 3, 7, 9), (129, 3, 224, 224)])
def test_stem_planes_behind_the_first_blocks_batchnorm(shape, fp16):
    """bnn_hip_stem7x7_bn_relu_pool_pack_affine_f32 (ABI 15): the stem's planes taken behind a per-channel affine of its
    output — what a pre-activation block's first binary layer reads (hierarchical_block.py:39) — are the planes the
    packing pass computes from the stem's fp32 output (bnn_hip_bn_act_pack_f32, relu = 1), bit for bit; the fp32 output is
    the plain stem's."""
    x = dev(gen.normal(gen.seed_of("stemaff", shape), (min(shape[0], 6),) + shape[1:]))
    x = x.repeat((shape[0] + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:shape[0]].contiguous()
    w = dev(gen.conv_weight("kaiming", 3, (64, 3, 7, 7)))
    a = dev((0.5 + gen.uniform(1, (64,))).astype(np.float32) * np.where(np.arange(64) % 7 == 0, -1, 1).astype(np.float32))
    b = dev((0.3 * gen.normal(2, (64,))).astype(np.float32))
    pa = dev((0.4 + gen.uniform(3, (64,))).astype(np.float32) * np.where(np.arange(64) % 5 == 0, -1, 1).astype(np.float32))
    pb = dev((-0.6 + 0.5 * gen.normal(4, (64,))).astype(np.float32))
    y0, _ = hipops.stem7x7(x, w, a, b, fp16=fp16)
    ref = hipops.bn_act_pack(y0, pa, pb, relu=True)
    y1, p1 = hipops.stem7x7(x, w, a, b, fp16=fp16, pack_affine=(pa, pb))
    assert torch.equal(y0, y1) and torch.equal(p1.P, ref.P) and torch.equal(p1.M, ref.M)
    frac = float((p1.P != hipops.stem7x7(x, w, a, b, fp16=fp16)[1].P).float().mean())
    assert frac > 0.5                        # (the affine really moves signs: not the plain planes again)
    _, p2 = hipops.stem7x7(x, w, a, b, fp16=fp16, pack_affine=(pa, pb), out_f32=False)
    assert torch.equal(p2.P, ref.P)
    with pytest.raises(native.NativeError):
        hipops.stem7x7(x[:1], w, a, b, exact_fp32=True, pack_affine=(pa, pb))
    lib = native.require()
    args = (x.data_ptr(), w.data_ptr(), a.data_ptr(), b.data_ptr(), pa.data_ptr(), pb.data_ptr(), 1, shape[2], shape[3])
    assert lib.bnn_hip_stem7x7_bn_relu_pool_pack_affine_f32(*args, native.STEM_EXACT_FP32, None, p1.P.data_ptr(),
                                                            p1.M.data_ptr(), None) == -1
    assert lib.bnn_hip_stem7x7_bn_relu_pool_pack_affine_f32(*args, 0, None, None, None, None) == -1
