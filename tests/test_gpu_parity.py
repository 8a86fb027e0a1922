"""GPU parity tests (run with `-m gpu` on an MI355X): every check goes through the C-ABI
(libbnn_hip.so via ctypes) and compares with the CPU oracle and with the fixtures generated from
the reference.

Bars (BASELINE.json north_star):
  * bit planes, weight bits, alpha, integer dot, float epilogue  ->  BIT-EXACT vs oracle route I
  * float output vs the reference's fp32 forward                 ->  rtol 1e-3 (+ atol 1e-5*max|ref|)
"""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import bnn_amd as bnn
import oracle
from bnn_amd import fastpath, hipops, native
from bnn_amd.models import Bottleneck, HBlock, PreBasicBlock, resnet18
from bnn_amd.ops import BasicInputBinarizer, BasicScaleBinarizer, XNORWeightBinarizer
from tests.golden import gen
from tests.golden.cases import LAYER_CASES, LAYER_CASES_BY_NAME, LINEAR_CASES

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
RTOL, ATOL_FRAC = 1e-3, 1e-5


def close(a, b):
    scale = max(1.0, float(np.nanmax(np.abs(b))))
    return np.allclose(a, b, rtol=RTOL, atol=ATOL_FRAC * scale, equal_nan=True)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def u64(t):
    return t.cpu().numpy().view(np.uint64)


# --------------------------------------------------------------------------------- pack_act
@pytest.mark.parametrize("shape,kind", [
    ((2, 64, 8, 8), "normal"), ((2, 128, 12, 12), "relu"), ((3, 70, 7, 7), "sparse"),   # HW odd -> VP=1
    ((2, 3, 11, 13), "normal"), ((1, 200, 7, 6), "negrelu"),                              # HW even -> VP=2
    ((2, 64, 8, 8), "special"), ((1, 1, 1, 1), "normal"), ((5, 33, 1, 1), "special"),
    ((1, 512, 14, 14), "relu"), ((2, 1024, 4, 4), "sparse"),
])
def test_pack_act_bit_exact(shape, kind):
    x = gen.activation(kind, gen.seed_of("pack", shape, kind), shape)
    act = hipops.pack_act(dev(x))
    P, M = oracle.pack_act(x)
    assert np.array_equal(u64(act.P), P) and np.array_equal(u64(act.M), M)


def test_pack_act_sign_semantics_on_device():
    x = np.array([0.0, -0.0, np.nan, 1e-45, -1e-45, np.inf, -np.inf, 3.0, -2.0, 1e-39, -1e-39],
                 np.float32).reshape(1, 11, 1, 1)
    act = hipops.pack_act(dev(x))
    assert int(u64(act.P)[0, 0, 0, 0]) == 0b01010101000
    assert int(u64(act.M)[0, 0, 0, 0]) == 0b10101010000


# ------------------------------------------------------------------------------ pack_weight
@pytest.mark.parametrize("case", LAYER_CASES, ids=lambda c: c.name)
def test_pack_weight_bit_exact(case):
    _, w, _, _ = case.tensors()
    pw = hipops.pack_weight(dev(w), case.center, case.compute_alpha)
    wbits, wnz, alpha, anyz = oracle.pack_weight(w, case.center, case.compute_alpha)
    assert np.array_equal(pw.wbits.cpu().numpy().view(np.uint32), wbits)
    assert np.array_equal(pw.wnz.cpu().numpy().view(np.uint32), wnz)
    assert np.array_equal(pw.alpha.cpu().numpy(), alpha)      # same reduction tree -> same bits
    assert pw.has_zero == anyz


# ------------------------------------------------------------------- integer dot + epilogue
@pytest.mark.parametrize("force_generic", [False, True], ids=["tiled", "generic"])
@pytest.mark.parametrize("case", LAYER_CASES, ids=lambda c: c.name)
def test_conv_bit_exact_vs_oracle_and_close_to_reference(case, force_generic, golden_layers):
    x, w, b, sc = case.tensors()
    act = hipops.pack_act(dev(x))
    pw = hipops.pack_weight(dev(w), case.center, case.compute_alpha)
    kw = dict(stride=case.stride, padding=case.pad, dilation=case.dilation, force_generic=force_generic)
    dot = hipops.bconv2d(act, pw, raw_dot=True, **kw).cpu().numpy()
    out = hipops.bconv2d(act, pw, None if b is None else dev(b), None if sc is None else dev(sc),
                         **kw).cpu().numpy()
    ref_out, ref_dot = oracle.binary_conv2d_int(x, w, b, sc, case.stride, case.pad, case.dilation,
                                                case.center, case.compute_alpha)
    assert np.array_equal(dot, ref_dot)          # popcount path == emulated integer path
    assert np.array_equal(out, ref_out)          # same fmaf epilogue -> same float bits
    assert close(out, golden_layers[case.name + "/out"])   # the reference's fp32 forward
    if not case.center:
        assert np.array_equal(dot, golden_layers[case.name + "/dot"].astype(np.int32))


@pytest.mark.parametrize("case", [c for c in LAYER_CASES if c.k == 3], ids=lambda c: c.name)
def test_nonneg_activation_kernels_bit_exact(case):
    """BNN_HIP_FLAG_ACT_NONNEG (P-plane-only kernels for ReLU outputs, M == 0) on max(x, 0): same
    integers and same float bits as the oracle, and as the general two-plane kernel."""
    x, w, b, sc = case.tensors()
    x = np.where(np.isnan(x), x, np.maximum(x, 0)).astype(np.float32)   # keep NaN probes: sign(NaN) = 0
    act = hipops.pack_act(dev(x))
    assert not act.nonneg and not u64(act.M).any()
    pw = hipops.pack_weight(dev(w), case.center, case.compute_alpha)
    kw = dict(stride=case.stride, padding=case.pad, dilation=case.dilation)
    general = hipops.bconv2d(act, pw, raw_dot=True, **kw).cpu().numpy()
    act.nonneg = True
    dot = hipops.bconv2d(act, pw, raw_dot=True, **kw).cpu().numpy()
    out = hipops.bconv2d(act, pw, None if b is None else dev(b), None if sc is None else dev(sc),
                         **kw).cpu().numpy()
    ref_out, ref_dot = oracle.binary_conv2d_int(x, w, b, sc, case.stride, case.pad, case.dilation,
                                                case.center, case.compute_alpha)
    assert np.array_equal(dot, ref_dot) and np.array_equal(dot, general)
    assert np.array_equal(out, ref_out)


@pytest.mark.parametrize("case", LINEAR_CASES, ids=lambda c: c.name)
def test_linear_through_c_abi(case, golden_layers):
    import ctypes
    x, w, b, sc = case.tensors()
    act = hipops.pack_act(dev(x[:, :, None, None]))
    pw = hipops.pack_weight(dev(w), case.center, True)
    out = torch.empty((case.B, case.O), device=DEV)
    bt = None if b is None else dev(b)
    st = None if sc is None else dev(sc)
    lib = native.require()
    native.check(lib.bnn_hip_blinear(case.B, case.F, case.O, act.P.data_ptr(), act.M.data_ptr(),
                                     pw.wbits.data_ptr(), pw.wnz.data_ptr(),
                                     int(pw.has_zero), pw.alpha.data_ptr(),
                                     None if bt is None else bt.data_ptr(),
                                     None if st is None else st.data_ptr(), out.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream), "blinear")
    ref, _ = oracle.binary_conv2d_int(x[:, :, None, None], w[:, :, None, None], b, sc, center=case.center)
    assert np.array_equal(out.cpu().numpy(), ref[:, :, 0, 0])
    assert close(out.cpu().numpy(), golden_layers["linear/" + case.name + "/out"])


def test_fused_f32_entry_point_matches_two_step():
    import ctypes
    case = LAYER_CASES_BY_NAME["c2_relu"]
    x, w, _, _ = case.tensors()
    xd, pw = dev(x), hipops.pack_weight(dev(w))
    d = native.ConvDesc(case.N, case.C, case.H, case.W, case.O, 3, 3, 1, 1, 1, 1, 1, 1, 0)
    lib = native.require()
    ws = torch.empty(lib.bnn_hip_conv_workspace_bytes(ctypes.byref(d)), dtype=torch.uint8, device=DEV)
    out = torch.empty((case.N, case.O, case.H, case.W), device=DEV)
    native.check(lib.bnn_hip_bconv2d_f32(ctypes.byref(d), xd.data_ptr(), pw.wbits.data_ptr(),
                                         pw.wnz.data_ptr(), pw.alpha.data_ptr(), None, None,
                                         out.data_ptr(), ws.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "bconv2d_f32")
    two = hipops.bconv2d(hipops.pack_act(xd), pw, stride=1, padding=1)
    assert torch.equal(out, two)


# --------------------------------------------------------------------- the drop-in API path
def make_layer(case):
    x, w, b, sc = case.tensors()
    conv = nn.Conv2d(case.C, case.O, case.k, stride=case.stride, padding=case.pad,
                     dilation=case.dilation, bias=case.bias)
    conv.weight.data.copy_(torch.from_numpy(w))
    if b is not None:
        conv.bias.data.copy_(torch.from_numpy(b))
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer,
                      activation_post_process=BasicScaleBinarizer if case.post == "scale" else bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer.with_args(compute_alpha=case.compute_alpha,
                                                                       center_weights=case.center))
    layer = bnn.prepare_binary_model(conv, cfg)
    if sc is not None:
        layer.activation_post_process.alpha.data.copy_(torch.from_numpy(sc).view(1, -1, 1, 1))
    return layer.to(DEV).eval(), x


@pytest.mark.parametrize("name", ["l1_64x56", "l2_0_c1_s2", "l4_512x7", "tail_c16_1x1",
                                  "center_bias_scale", "no_alpha", "zero_weights", "special_vals"])
def test_layer_forward_runs_hip_and_matches_reference(name, golden_layers):
    layer, x = make_layer(LAYER_CASES_BY_NAME[name])
    before, launches = fastpath.stats()["conv2d"], native.launch_count()
    with torch.no_grad():
        out = layer(dev(x)).cpu().numpy()
    assert fastpath.stats()["conv2d"] == before + 1 and native.launch_count() >= launches + 2
    assert close(out, golden_layers[name + "/out"])
    # same layer, autograd recording -> torch composition; both formulations agree
    out_eager = layer(dev(x)).detach().cpu().numpy()
    assert fastpath.stats()["conv2d"] == before + 1
    assert close(out, out_eager)


def test_known_answer_vectors_on_device(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_test_layers.npz"))
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer,
                      activation_post_process=BasicScaleBinarizer, weight_pre_process=XNORWeightBinarizer)
    w = torch.from_numpy(g["weights"])
    data = torch.from_numpy(g["data"])
    lin = nn.Linear(3, 3, bias=False); lin.weight.data.copy_(w.view(3, 3))
    c1 = nn.Conv1d(3, 3, 1, bias=False); c1.weight.data.copy_(w.view(3, 3, 1))
    c2 = nn.Conv2d(3, 3, 1, bias=False); c2.weight.data.copy_(w.view(3, 3, 1, 1))
    s0 = fastpath.stats()
    with torch.no_grad():
        o_lin = bnn.prepare_binary_model(lin, cfg).to(DEV)(data[:, :, 0, 0].reshape(1, 3).to(DEV)).cpu().numpy()
        o_c1 = bnn.prepare_binary_model(c1, cfg).to(DEV)(data[:, :, :, 0].reshape(1, 3, 2).to(DEV)).cpu().numpy()
        o_c2 = bnn.prepare_binary_model(c2, cfg).to(DEV)(data.to(DEV)).cpu().numpy()
    s1 = fastpath.stats()
    assert (s1["linear"], s1["conv1d"], s1["conv2d"]) == (s0["linear"] + 1, s0["conv1d"] + 1, s0["conv2d"] + 1)
    assert np.allclose(o_lin, g["lit_linear"], atol=1e-4) and np.allclose(o_lin, g["linear"], atol=1e-6)
    assert np.allclose(o_c1, g["lit_conv1d"], atol=1e-4) and np.allclose(o_c1, g["conv1d"], atol=1e-6)
    assert np.allclose(o_c2, g["lit_conv2d"], atol=1e-4) and np.allclose(o_c2, g["conv2d"], atol=1e-6)


def test_packed_weight_cache_tracks_weight_changes():
    layer, x = make_layer(LAYER_CASES_BY_NAME["c2_relu"])
    xd = dev(x)
    with torch.no_grad():
        y0 = layer(xd)
        packs = fastpath.stats()["weight_packs"]
        y1 = layer(xd)
        assert fastpath.stats()["weight_packs"] == packs and torch.equal(y0, y1)   # cache hit
        layer.weight.mul_(-1.0)                                                    # in-place update
        y2 = layer(xd)
        assert fastpath.stats()["weight_packs"] == packs + 1 and torch.equal(y2, -y0)
        sd = {k: v.clone() for k, v in layer.state_dict().items()}
        sd["weight"] = -sd["weight"]
        layer.load_state_dict(sd)
        assert torch.equal(layer(xd), y0)


def test_missing_library_raises_on_gpu_tensors(monkeypatch):
    layer, x = make_layer(LAYER_CASES_BY_NAME["c2_relu"])

    def boom():
        raise native.NativeError("library missing")
    monkeypatch.setattr(native, "require", boom)
    with torch.no_grad(), pytest.raises(native.NativeError):
        layer(dev(x))


# ------------------------------------------------------------------------ networks / blocks
def _load_state(model, seed):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, seed).items()})
    return model


def xnor_cfg():
    return bnn.BConfig(activation_pre_process=BasicInputBinarizer,
                       activation_post_process=bnn.Identity, weight_pre_process=XNORWeightBinarizer)


@pytest.fixture(scope="module")
def r18():
    net = bnn.prepare_binary_model(resnet18(), xnor_cfg(),
                                   custom_config_layers_name={"conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    return _load_state(net, 1).to(DEV).eval()


@pytest.mark.parametrize("tag,shape", [("32", (4, 3, 32, 32)), ("64", (2, 3, 64, 64)), ("224", (2, 3, 224, 224))])
def test_resnet18_logits_match_reference(r18, tag, shape, golden_dir):
    """Configs 1 and 3 of BASELINE.json at reduced batch: logits of the reference's own forward."""
    g = np.load(os.path.join(golden_dir, "resnet18.npz"))
    x = dev(gen.normal(gen.seed_of("r18", tag), shape))
    before = fastpath.stats()["conv2d"]
    with torch.no_grad():
        y = r18(x).cpu().numpy()
    assert fastpath.stats()["conv2d"] == before + 19        # all binary convs incl. 3 downsamples
    ref = g["logits_" + tag]
    assert np.allclose(y, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
    assert (y.argmax(1) == ref.argmax(1)).all()


@pytest.mark.parametrize("name,ctor,shape", [
    ("hblock_256", lambda: HBlock(256, 256), (2, 256, 8, 8)),
    ("bottleneck_256_64", lambda: Bottleneck(256, 64), (2, 256, 8, 8)),
    ("prebasic_64", lambda: PreBasicBlock(64, 64), (2, 64, 10, 10)),
    ("prebasic_64_prelu", lambda: PreBasicBlock(64, 64, activation=nn.PReLU), (2, 64, 10, 10)),
])
def test_blocks_match_reference_on_device(name, ctor, shape, golden_dir):
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    blk = _load_state(bnn.prepare_binary_model(ctor(), xnor_cfg()), gen.seed_of("block", name)).to(DEV).eval()
    with torch.no_grad():
        y = blk(dev(gen.normal(gen.seed_of("blockx", name), shape))).cpu().numpy()
    assert np.allclose(y, g[name], rtol=1e-3, atol=1e-4 * np.abs(g[name]).max())


# -------------------------------------------------------- full-size, size-independent properties
@pytest.fixture(scope="module")
def c2_full():
    """BASELINE config 2 at full size: Conv2d 128->128 3x3 p1, 56x56, batch 256 (post-ReLU input)."""
    N, C, H, W, O = 256, 128, 56, 56, 128
    base = dev(gen.activation("relu", 7, (8, C, H, W)))
    x = base.repeat(N // 8, 1, 1, 1).clone()
    x[8:] *= torch.linspace(0.5, 1.5, N - 8, device=DEV).view(-1, 1, 1, 1)   # sign-preserving scaling
    x[100, :, 3:9] = 0.0                                                      # make some images distinct
    x[200] = -x[200]
    w = dev(gen.conv_weight("kaiming", 8, (O, C, 3, 3)))
    return x, w


def test_c2_full_size_against_torch_gpu_float(c2_full):
    """Size-independent cross-check: the float formulation on the GPU (sign + F.conv2d, MIOpen)."""
    x, w = c2_full
    pw = hipops.pack_weight(w)
    out = hipops.bconv2d(hipops.pack_act(x), pw, stride=1, padding=1)
    alpha = w.abs().flatten(1).mean(1).view(-1, 1, 1, 1)
    ref = F.conv2d(torch.sign(x), torch.sign(w) * alpha, None, 1, 1)
    assert torch.allclose(out, ref, rtol=1e-3, atol=1e-5 * float(ref.abs().max()))
    # the integer dot recovered from both sides is identical
    dot = hipops.bconv2d(hipops.pack_act(x), pw, stride=1, padding=1, raw_dot=True)
    ref_dot = F.conv2d(torch.sign(x[:16]).double(), torch.sign(w).double(), None, 1, 1)
    assert torch.equal(dot[:16].double(), ref_dot)


def test_c2_full_size_properties(c2_full):
    x, w = c2_full
    act = hipops.pack_act(x)
    pw = hipops.pack_weight(w)
    out = hipops.bconv2d(act, pw, stride=1, padding=1)
    # determinism: integer kernel -> run-to-run bit-exact
    assert torch.equal(out, hipops.bconv2d(act, pw, stride=1, padding=1))
    # tiled kernel == shape-generic kernel on a batch slice
    sl = act.batch_slice(0, 32)
    assert torch.equal(out[:32], hipops.bconv2d(sl, pw, stride=1, padding=1, force_generic=True))
    # images are independent units: a batch of one gives the same image
    for n in (0, 100, 255):
        one = act.batch_slice(n, n + 1)
        assert torch.equal(out[n:n + 1], hipops.bconv2d(one, pw, stride=1, padding=1))
    # antisymmetry: negating the weights (or the input) negates every output exactly
    assert torch.equal(hipops.bconv2d(act, hipops.pack_weight(-w), stride=1, padding=1), -out)
    assert torch.equal(hipops.bconv2d(hipops.pack_act(-x), pw, stride=1, padding=1), -out)
    # sign() is scale-invariant: images 0..7 repeat with positive scales
    assert torch.equal(out[0:8], out[16:24]) and torch.equal(out[200], -out[200 % 8])
    # output-channel permutation commutes with the convolution
    perm = torch.randperm(128, device=DEV)
    assert torch.equal(hipops.bconv2d(act, hipops.pack_weight(w[perm]), stride=1, padding=1), out[:, perm])
    # checksum: sum over output channels of dot == dot against the summed sign-weights (linearity)
    dot = hipops.bconv2d(act, pw, stride=1, padding=1, raw_dot=True)
    wsum = torch.sign(w).sum(0, keepdim=True)
    lin = F.conv2d(torch.sign(x[:8]).double(), wsum.double(), None, 1, 1)
    assert torch.equal(dot[:8].sum(1, keepdim=True).double(), lin)


def test_probe_reports_a_plausible_rate():
    info = native.device_info(0)
    assert info["arch"].startswith("gfx950")
    rate = hipops.probe_int_alu(512)["lane_ops_per_s"]
    peak = info["compute_units"] * 4 * 32 * info["clock_khz"] * 1e3
    assert 0.05 * peak < rate < 1.2 * peak


# ------------------------------------------------------- BASELINE config 5 building blocks at net level
@pytest.mark.parametrize("name", ["resnet50_bottleneck", "resnet_hblock", "resnet18_preact_prelu"])
def test_config5_style_networks_hip_equals_composition(name):
    """Mixed 1x1 / 3x3 binary convs (Bottleneck), hierarchical blocks and the imagenet.py
    pre-activation/PReLU dataflow: the per-layer HIP path against the torch composition (the
    reference's own formulation, itself pinned to the reference by module-level fixtures)."""
    from bnn_amd.models import ResNet, resnet50
    torch.manual_seed(0)
    if name == "resnet50_bottleneck":
        net = resnet50(num_classes=100)
    elif name == "resnet_hblock":
        net = ResNet(HBlock, [1, 2, 1, 1], num_classes=100)
    else:
        net = resnet18(block_type=PreBasicBlock, activation=nn.PReLU, num_classes=100)
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=BasicScaleBinarizer,
                      weight_pre_process=XNORWeightBinarizer.with_args(center_weights=True))
    net = bnn.prepare_binary_model(net, cfg, ignore_layers_name=["_first_", "_last_"])
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 5).items()})
    net.eval()
    x = torch.from_numpy(gen.normal(gen.seed_of("c5", name), (2, 3, 64, 64)))
    with torch.no_grad():
        ref = net(x).numpy()                       # CPU: torch composition
        before = fastpath.stats()["conv2d"]
        got = net.to(DEV)(x.to(DEV)).cpu().numpy()  # GPU: XNOR/popcount kernels per layer
    n_bin = sum(isinstance(m, bnn.layers.Conv2d) for m in net.modules())
    assert fastpath.stats()["conv2d"] == before + n_bin and n_bin >= 8
    assert np.allclose(got, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())


# ------------------------------------------------------------------ edge cases of the host side
def test_empty_batch_and_noncontiguous_inputs(golden_layers):
    layer, x = make_layer(LAYER_CASES_BY_NAME["c2_relu"])
    with torch.no_grad():
        assert layer(torch.empty((0, 128, 12, 12), device=DEV)).shape == (0, 128, 12, 12)
        xd = dev(x)
        ref = layer(xd)
        cl = xd.contiguous(memory_format=torch.channels_last)        # same values, NHWC strides
        assert torch.equal(layer(cl), ref)
        wide = torch.zeros((2, 128, 12, 20), device=DEV)
        wide[..., 4:16] = xd
        assert torch.equal(layer(wide[..., 4:16]), ref)              # strided view


def test_batch_splitting_path_matches_single_launch(monkeypatch):
    case = LAYER_CASES_BY_NAME["l2_ds_1x1"]
    x, w, _, _ = case.tensors()
    act, pw = hipops.pack_act(dev(np.concatenate([x] * 4))), hipops.pack_weight(dev(w))
    whole = hipops.bconv2d(act, pw)
    monkeypatch.setattr(hipops, "_MAX_ELEMS", 3 * 128 * 14 * 14)    # forces launches of <= 3 images
    assert torch.equal(hipops.bconv2d(act, pw), whole)
    assert torch.equal(hipops.bconv2d(act, pw, raw_dot=True), hipops.bconv2d(act, pw, raw_dot=True))


def test_fused_batch_splitting_matches_single_launch(monkeypatch):
    """bconv2d_fused splits the batch like bconv2d when a tensor exceeds one launch's 2^31-element addressing."""
    case = LAYER_CASES_BY_NAME["l2_ds_1x1"]
    x, w, _, _ = case.tensors()
    act, pw = hipops.pack_act(dev(np.concatenate([x] * 4))), hipops.pack_weight(dev(w))
    n, o, hw = 8, case.O, 14
    res = dev(gen.normal(77, (n, o, hw, hw)))
    a_, b_ = dev((0.5 + gen.uniform(1, (o,))).astype(np.float32)), dev((0.3 * gen.normal(2, (o,))).astype(np.float32))
    kw = dict(bn_scale=a_, bn_shift=b_, residual=res, relu=True, out_f32=True, out_packed=True)
    y0, p0 = hipops.bconv2d_fused(act, pw, **kw)
    monkeypatch.setattr(hipops, "_MAX_ELEMS", 3 * o * hw * hw)       # forces launches of <= 3 images
    y1, p1 = hipops.bconv2d_fused(act, pw, **kw)
    assert torch.equal(y0, y1) and torch.equal(p0.P, p1.P) and torch.equal(p0.M, p1.M) and p0.nonneg == p1.nonneg
    big = torch.zeros((n, 2 * o, hw, hw), device=DEV)
    hipops.bconv2d_fused(act, pw, bn_scale=a_, bn_shift=b_, out=big, out_c_offset=o, out_f32=True)
    monkeypatch.setattr(hipops, "_MAX_ELEMS", (1 << 31) - 1)
    ref = torch.zeros_like(big)
    hipops.bconv2d_fused(act, pw, bn_scale=a_, bn_shift=b_, out=ref, out_c_offset=o, out_f32=True)
    assert torch.equal(big, ref)


ZERO_W_SHAPES = [  # (N, C, H, W, O, k, stride, pad): single-chunk 3x3, both chunk widths, multi-chunk, 1x1 of every cwc
    (2, 64, 9, 9, 40, 3, 1, 1), (2, 128, 8, 8, 32, 3, 2, 1), (1, 256, 7, 7, 64, 3, 1, 1), (1, 96, 6, 5, 33, 3, 1, 0),
    (2, 64, 7, 7, 32, 1, 1, 0), (2, 128, 5, 5, 70, 1, 1, 0), (1, 256, 4, 4, 32, 1, 2, 0), (1, 512, 3, 3, 32, 1, 1, 0),
]


@pytest.mark.parametrize("shape", ZERO_W_SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("act_kind", ["relu", "sparse"])
def test_tiled_zero_weight_kernels_bit_exact(shape, act_kind):
    """Pruned weights (sign(0) == 0, bnn/ops.py:66,136) on the TILED kernels (BNN_HIP_FLAG_WEIGHT_ZEROS with a
    second scalar stream for the mask): integer dot == oracle == the shape-generic kernel, fused epilogue too."""
    N, C, H, W, O, k, st, pad = shape
    x = gen.activation(act_kind, gen.seed_of("wz", shape), (N, C, H, W))
    w = gen.conv_weight("withzeros", gen.seed_of("wzw", shape), (O, C, k, k))
    act, pw = hipops.pack_act(dev(x)), hipops.pack_weight(dev(w))
    assert pw.has_zero
    dot = hipops.bconv2d(act, pw, stride=st, padding=pad, raw_dot=True)
    gen_dot = hipops.bconv2d(act, pw, stride=st, padding=pad, raw_dot=True, force_generic=True)
    _, ref_dot = oracle.binary_conv2d_int(x, w, None, None, st, pad)
    assert np.array_equal(dot.cpu().numpy(), ref_dot) and torch.equal(dot, gen_dot)
    out = hipops.bconv2d(act, pw, stride=st, padding=pad)
    ref_out, _ = oracle.binary_conv2d_int(x, w, None, None, st, pad)
    assert np.array_equal(out.cpu().numpy(), ref_out)
    a_, b_ = dev((0.5 + gen.uniform(1, (O,))).astype(np.float32)), dev((0.3 * gen.normal(2, (O,))).astype(np.float32))
    y0, p0 = hipops.bconv2d_fused(act, pw, bn_scale=a_, bn_shift=b_, relu=True, out_f32=True, out_packed=True,
                                  stride=st, padding=pad)
    y1, p1 = hipops.bconv2d_fused(act, pw, bn_scale=a_, bn_shift=b_, relu=True, out_f32=True, out_packed=True,
                                  stride=st, padding=pad, force_generic=True)
    assert torch.equal(y0, y1) and torch.equal(p0.P, p1.P) and torch.equal(p0.M, p1.M)
    if act_kind == "relu":   # the non-negative promise is ignored by the zero-weight variant (two-plane maths)
        act.nonneg = True
        assert torch.equal(hipops.bconv2d(act, pw, stride=st, padding=pad, raw_dot=True), dot)


def test_runs_on_the_current_side_stream():
    case = LAYER_CASES_BY_NAME["c2_relu"]
    x, w, _, _ = case.tensors()
    xd, wd = dev(x), dev(w)
    ref = hipops.bconv2d(hipops.pack_act(xd), hipops.pack_weight(wd), stride=1, padding=1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = hipops.bconv2d(hipops.pack_act(xd), hipops.pack_weight(wd), stride=1, padding=1)
    side.synchronize()
    assert torch.equal(out, ref)


def test_grad_mode_takes_the_training_variant_not_the_inference_path():
    layer, x = make_layer(LAYER_CASES_BY_NAME["c2_relu"])
    before = fastpath.stats()["conv2d"]
    y = layer(dev(x).requires_grad_(True))          # autograd recording
    y.sum().backward()
    assert fastpath.stats()["conv2d"] == before


@pytest.mark.parametrize("name", ["c2_relu", "c2_normal", "special_vals", "tail_c96", "tail_c16_1x1"])
def test_fp16_pack_is_bit_exact_and_half_models_stay_on_the_hip_path(name):
    """`.half()` models (SURVEY §8b "dtype fp32/fp16"): planes straight from the fp16 tensor — sign() is exact in
    any precision, so they equal the oracle's planes of the widened tensor bit for bit — fp32 arithmetic inside,
    one rounding to fp16 at the end; compared with the reference formulation run by torch in fp16."""
    case = LAYER_CASES_BY_NAME[name]
    layer, x = make_layer(case)
    xh = dev(x).half()
    act = hipops.pack_act(xh)
    P, M = oracle.pack_act(xh.float().cpu().numpy())
    assert np.array_equal(u64(act.P), P) and np.array_equal(u64(act.M), M)
    layer = layer.half()
    before = fastpath.stats()["conv2d"]
    with torch.no_grad():
        y = layer(xh)
        assert fastpath.stats()["conv2d"] == before + 1 and y.dtype == torch.float16
        # the reference's own op sequence in fp16 (torch composition on the GPU)
        xs = layer.activation_pre_process(xh)
        ref = layer.activation_post_process(
            F.conv2d(xs, layer.weight_pre_process(layer.weight), layer.bias, layer.stride, layer.padding,
                     layer.dilation), xh)
    ok = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(y), ok)
    # fp16 has 11 significand bits: one rounding of the output (5e-4) + alpha rounded to fp16 (5e-4)
    assert torch.allclose(y[ok].float(), ref[ok].float(), rtol=4e-3, atol=4e-3 * float(ref[ok].float().abs().max()))


def test_writes_through_dot_data_need_invalidate_and_training_never_trusts_the_cache():
    """`p.data.clamp_()` / `p.data.copy_(ema)` do not bump the version counter the packed-weight cache is keyed
    on: the inference path needs `fastpath.invalidate`, the training forward re-derives the bits every call."""
    layer, x = make_layer(LAYER_CASES_BY_NAME["c2_relu"])
    xd = dev(x)
    with torch.no_grad():
        y0 = layer(xd)
        v = layer.weight._version
        layer.weight.data.mul_(-1.0)                  # invisible to autograd's version counter
        assert layer.weight._version == v
        assert torch.equal(layer(xd), y0)             # stale: documented behaviour
        assert fastpath.invalidate(layer) == 1
        assert torch.equal(layer(xd), -y0)
        layer.weight.data.mul_(-1.0)
    y_tr = layer(xd.clone().requires_grad_(True))     # training forward: always from the current values
    assert torch.equal(y_tr.detach(), y0)
    # round 5: train() <-> eval() drops the derived data, so what was written through .data while training (weight
    # clipping after the optimizer step) is seen by the first evaluation forward without invalidate()
    layer.eval()
    with torch.no_grad():
        assert torch.equal(layer(xd), y0)
        layer.train()
        layer.weight.data.mul_(-1.0)                  # "p.data.clamp_()" of a training loop
        layer.eval()
        assert torch.equal(layer(xd), -y0)
        layer.weight.data.mul_(-1.0)                  # between two forwards of the SAME mode: still the documented case
        assert torch.equal(layer(xd), -y0)
        fastpath.invalidate(layer)
        assert torch.equal(layer(xd), y0)


def test_zero_weights_first_seen_by_a_grad_mode_forward_are_honoured_at_once():
    """Weights that are exactly 0 from the start (pruned, zero-initialised): the first pack of a layer reads its zero
    flag even in grad mode, so the training forward and the inference forward behind it are both exact (sign(0) == 0,
    never -1) and nothing has to be warned about afterwards."""
    import warnings
    case = LAYER_CASES_BY_NAME["zero_weights"]
    layer, x = make_layer(case)
    xd = dev(x)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        yt = layer(xd.clone().requires_grad_(True))   # grad mode, cold cache: synchronous, zero-aware pack
        with torch.no_grad():
            y = layer(xd)
    assert layer.__dict__["_bnn_packed"][1].has_zero and torch.equal(yt.detach(), y)
    ref, _ = oracle.binary_conv2d_int(x, case.tensors()[1], None, None, case.stride, case.pad, case.dilation,
                                      case.center, case.compute_alpha)
    assert np.array_equal(y.cpu().numpy(), ref)


def test_packed_checkpoint_round_trip_on_device(tmp_path):
    """save_packed -> load_packed -> load_state_dict: the HIP forward is bit-identical (non-centred XNOR:
    same sign bits, alpha recomputed exactly by the double-precision reduction of pack_weight)."""
    from bnn_amd import checkpoint
    from bnn_amd.inference import FusedResNet
    net = _r18_for_ckpt()
    x = dev(gen.normal(gen.seed_of("ckpt"), (4, 3, 64, 64)))
    with torch.no_grad():
        y0 = net(x).clone()
        f0 = FusedResNet(net)(x).clone()
    path = str(tmp_path / "r18.bnnpack")
    stats = checkpoint.save_packed(net, path)
    assert stats["binary_weights_packed"] * 30 < stats["binary_weights_fp32"]
    net.load_state_dict(checkpoint.load_packed(path, map_location=DEV))
    with torch.no_grad():
        assert torch.equal(net(x), y0)
        assert torch.equal(FusedResNet(net)(x), f0)


def _r18_for_ckpt():
    torch.manual_seed(0)
    net = resnet18(num_classes=10)
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    net = bnn.prepare_binary_model(net, cfg, ignore_layers_name=["conv1", "fc"])
    g = torch.Generator().manual_seed(1)
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data = torch.rand(m.num_features, generator=g) + 0.5
            m.bias.data = torch.randn(m.num_features, generator=g) * 0.3
            m.running_mean = torch.randn(m.num_features, generator=g) * 0.5
            m.running_var = torch.rand(m.num_features, generator=g) + 0.5
    return net.to(DEV).eval()


@pytest.mark.parametrize("C,O,k,stride,pad,L", [(64, 32, 3, 1, 1, 50), (70, 40, 5, 2, 2, 33), (128, 64, 1, 1, 0, 17)])
def test_conv1d_hip_matches_composition(C, O, k, stride, pad, L):
    """bnn.layers.Conv1d (bnn/layers/conv.py:10-62) on the device: an H == 1 convolution through the same
    kernels (generic kernel for 1 x k windows), against the layer's own torch composition on CPU."""
    torch.manual_seed(C + O + k)
    conv = nn.Conv1d(C, O, k, stride=stride, padding=pad, bias=True)
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    layer = bnn.prepare_binary_model(conv, cfg).eval()
    x = torch.from_numpy(gen.normal(gen.seed_of("c1d", (C, O, k)), (3, C, L)))
    with torch.no_grad():
        ref = layer(x)
        before = fastpath.stats()["conv1d"]
        got = layer.to(DEV)(x.to(DEV)).cpu()
    assert fastpath.stats()["conv1d"] == before + 1
    assert got.shape == ref.shape and torch.allclose(got, ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))


def test_advanced_input_binarizer_takes_the_hip_path_in_inference_only():
    """AdvancedInputBinarizer (bnn/ops.py:167-177, SURVEY (f)2): value sign(tanh(t*x)) == sign(x) -> same kernels
    for inference; its gradient is not the STE, so a training forward keeps the torch composition."""
    from bnn_amd.ops import AdvancedInputBinarizer
    case = LAYER_CASES_BY_NAME["c2_relu"]
    x, w, _, _ = case.tensors()
    conv = nn.Conv2d(case.C, case.O, case.k, stride=case.stride, padding=case.pad, bias=False)
    conv.weight.data.copy_(torch.from_numpy(w))
    mk = lambda pre: bnn.prepare_binary_model(  # noqa: E731
        nn.Sequential(conv), bnn.BConfig(activation_pre_process=pre, activation_post_process=bnn.Identity,
                                         weight_pre_process=XNORWeightBinarizer))[0].to(DEV).eval()
    adv, basic = mk(AdvancedInputBinarizer), mk(BasicInputBinarizer)
    s0 = fastpath.stats()
    with torch.no_grad():
        ya, yb = adv(dev(x)), basic(dev(x))
    s1 = fastpath.stats()
    assert s1["conv2d"] == s0["conv2d"] + 2 and torch.equal(ya, yb)
    adv(dev(x).requires_grad_(True)).sum().backward()          # autograd: composition, not the STE fast path
    assert fastpath.stats()["conv2d_train"] == s1["conv2d_train"]
    odd = mk(AdvancedInputBinarizer.with_args(derivative_funct=torch.sigmoid))   # sign(sigmoid(.)) != sign(x)
    with torch.no_grad():
        odd(dev(x))
    assert fastpath.stats()["conv2d"] == s1["conv2d"]
