"""GPU parity tests of the ONE-LAUNCH layer (`bnn_hip_bconv2d_direct`, csrc/bconv_fly.hip): float NCHW in -> fp32
NCHW out with sign(x) computed on the fly in LDS.  Reference: bnn/layers/conv.py:90-97 + bnn/ops.py:63-66.

Bars: bit-exact against the CPU oracle (integer route: same fmaf epilogue) and bit-identical to the two-launch
form pack_act + bconv2d, for every band plan; within 1e-3 of the reference's fp32 forward (fixtures).
"""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from bnn_amd import fastpath, hipops, native
from tests.golden import gen
from tests.golden.cases import LAYER_CASES, LAYER_CASES_BY_NAME
from tests.test_gpu_parity import close, dev, make_layer

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


def _args(case):
    x, w, b, sc = case.tensors()
    pw = hipops.pack_weight(dev(w), case.center, case.compute_alpha)
    kw = dict(stride=case.stride, padding=case.pad, dilation=case.dilation)
    return x, w, b, sc, pw, kw


def _plan(images=1, rows=0, waves=8, obw=1, ahead=-1, head=-1, tail=-1, prod=-1):
    p = native.FlyPlan()
    p.images_per_band, p.rows_per_band, p.waves, p.blocks_per_unit, p.pack_ahead = images, rows, waves, obw, ahead
    p.fine_head, p.fine_tail, p.producers = head, tail, prod
    return p


@pytest.mark.parametrize("case", LAYER_CASES, ids=lambda c: c.name)
def test_direct_layer_bit_exact_vs_oracle_and_two_launch_form(case, golden_layers):
    x, w, b, sc, pw, kw = _args(case)
    bt, st = (None if b is None else dev(b)), (None if sc is None else dev(sc))
    assert hipops.direct_plan(x.shape, pw, **kw) is not None          # every golden case runs as one launch
    before = native.launch_count()
    out = hipops.bconv2d_direct(dev(x), pw, bt, st, route="direct", **kw)
    assert native.launch_count() == before + 1
    # (the default route sends tiny images with wide inputs / large kernels through pack_act + conv: same bits)
    assert torch.equal(hipops.bconv2d_direct(dev(x), pw, bt, st, **kw), out)
    ref_out, _ = oracle.binary_conv2d_int(x, w, b, sc, case.stride, case.pad, case.dilation, case.center,
                                          case.compute_alpha)
    assert np.array_equal(out.cpu().numpy(), ref_out)                 # same integers, same fmaf epilogue
    two = hipops.bconv2d(hipops.pack_act(dev(x)), pw, bt, st, **kw)
    assert torch.equal(out, two)
    assert close(out.cpu().numpy(), golden_layers[case.name + "/out"])  # the reference's fp32 forward
    # the shape-generic unit on the same LDS tile
    gen_out = hipops.bconv2d_direct(dev(x), pw, bt, st, force_generic=True, **kw)
    assert torch.equal(gen_out, two)


@pytest.mark.parametrize("name", ["l1_64x56", "l2_0_c1_s2", "l2_128x28", "l3_256x14", "l4_512x7", "l2_ds_1x1",
                                  "c1024_1x1", "c2_normal", "special_vals", "tail_c3", "tail_c96", "tail_c200_o5",
                                  "k5_generic", "k3_dil2_generic", "zero_weights", "center_bias_scale"])
def test_direct_layer_fp16_input(name):
    """A `.half()` model: planes of the exactly widened values, fp32 arithmetic, fp32 out."""
    case = LAYER_CASES_BY_NAME[name]
    x, w, b, sc, pw, kw = _args(case)
    with np.errstate(over="ignore"):
        x16 = x.astype(np.float16)
    bt, st = (None if b is None else dev(b)), (None if sc is None else dev(sc))
    out = hipops.bconv2d_direct(dev(x16), pw, bt, st, **kw)
    ref_out, _ = oracle.binary_conv2d_int(x16.astype(np.float32), w, b, sc, case.stride, case.pad, case.dilation,
                                          case.center, case.compute_alpha)
    assert np.array_equal(out.cpu().numpy(), ref_out)
    assert torch.equal(out, hipops.bconv2d(hipops.pack_act(dev(x16)), pw, bt, st, **kw))


def test_direct_layer_sign_semantics_on_device():
    """NaN / +-0 -> 0, denormals and infinities keep their sign (torch.sign; bnn/ops.py:66) — in every channel
    position of a word, through the hand-written class test of the fly kernel, fp32 and fp16."""
    vals = np.array([0.0, -0.0, np.nan, 1e-45, -1e-45, np.inf, -np.inf, 3.0, -2.0, 1e-39, -1e-39, -np.nan],
                    np.float32)
    rng = np.random.default_rng(5)
    x = vals[rng.integers(0, len(vals), size=(3, 70, 6, 7))]
    w = gen.conv_weight("kaiming", 91, (40, 70, 3, 3))
    pw = hipops.pack_weight(dev(w))
    ref_out, _ = oracle.binary_conv2d_int(x, w, None, None, 1, 1, 1)
    assert np.array_equal(hipops.bconv2d_direct(dev(x), pw, padding=1).cpu().numpy(), ref_out)
    v16 = np.array([0.0, -0.0, np.nan, 6e-8, -6e-8, np.inf, -np.inf, 3.0, -2.0, 6e-5, -6e-5], np.float16)
    x16 = v16[rng.integers(0, len(v16), size=(3, 70, 6, 7))]
    ref16, _ = oracle.binary_conv2d_int(x16.astype(np.float32), w, None, None, 1, 1, 1)
    assert np.array_equal(hipops.bconv2d_direct(dev(x16), pw, padding=1).cpu().numpy(), ref16)


PLAN_CASES = ["l1_64x56", "l2_0_c1_s2", "l2_128x28", "l3_256x14", "l4_512x7", "l3_ds_1x1", "c2_relu", "tail_c96",
              "tail_c200_o5", "pad0_3x3", "k5_generic", "k3_dil2_generic", "k1_s2", "zero_weights", "tail_c3"]


@pytest.mark.parametrize("name", PLAN_CASES)
def test_every_band_plan_gives_the_same_bits(name):
    """Row-split bands (halo rows packed by two workgroups), several images per band, 1..16 waves, 1/2/4 blocks per
    unit: the plan only moves work around."""
    case = LAYER_CASES_BY_NAME[name]
    x, w, b, sc, pw, kw = _args(case)
    # a few more images than the fixture has, so that multi-image bands and ragged last bands exist
    x = np.concatenate([x, x[::-1] * -1.0, x[:1]], axis=0) if case.N < 5 else x
    xd = dev(x)
    bt, st = (None if b is None else dev(b)), (None if sc is None else dev(sc))
    base = hipops.bconv2d(hipops.pack_act(xd), pw, bt, st, **kw)
    ho = base.shape[2]
    plans = [_plan(1, ho, 16, 1), _plan(1, ho, 1, 2, head=0, tail=0), _plan(1, ho, 3, 4, head=1, tail=1),
             _plan(2, ho, 4, 1), _plan(3, ho, 8, 2, ahead=0, head=5, tail=0), _plan(x.shape[0], ho, 16, 4, tail=100),
             _plan(1, 1, 4, 1), _plan(1, 2, 2, 2, head=100), _plan(1, max(1, ho // 2), 8, 1, ahead=3),
             _plan(1, max(1, ho - 1), 5, 2, head=0, tail=2), _plan(1, 3, 16, 4, ahead=50)]
    cw32 = 2 * ((case.C + 63) // 64)
    ran = 0
    for p in plans:
        p.rows_per_band = min(p.rows_per_band, ho)
        what = (p.images_per_band, p.rows_per_band, p.waves, p.blocks_per_unit)
        slab_rows = (p.rows_per_band - 1) * case.stride + (case.k - 1) * case.dilation + 1
        tile = min(p.images_per_band, x.shape[0]) * slab_rows * (case.W + 2 * case.pad) * 2 * cw32 * 4
        try:
            out = hipops.bconv2d_direct(xd, pw, bt, st, plan=p, **kw)
        except native.NativeError:
            assert tile > 150 * 1024, what        # only a band that cannot fit a CU's LDS may be refused
            continue
        assert torch.equal(out, base), what
        ran += 1
    assert ran >= 8


def test_invalid_plans_are_rejected():
    case = LAYER_CASES_BY_NAME["c2_relu"]
    x, w, b, sc, pw, kw = _args(case)
    for p in (_plan(0, 12, 8, 1), _plan(1, 0, 8, 1), _plan(1, 13, 8, 1), _plan(2, 6, 8, 1), _plan(1, 12, 17, 1),
              _plan(1, 12, 8, 3)):
        with pytest.raises(native.NativeError):
            hipops.bconv2d_direct(dev(x), pw, plan=p, **kw)


@pytest.mark.parametrize("shape", [
    # (N, C, H, W, O, k, stride, pad): the binary convs of ResNet-18 @224 at their real spatial size
    (12, 64, 56, 56, 64, 3, 1, 1), (12, 64, 56, 56, 128, 3, 2, 1), (16, 128, 28, 28, 128, 3, 1, 1),
    (16, 64, 28, 28, 128, 1, 1, 0), (16, 128, 28, 28, 256, 3, 2, 1), (24, 256, 14, 14, 256, 3, 1, 1),
    (24, 128, 14, 14, 256, 1, 1, 0), (24, 256, 14, 14, 512, 3, 2, 1), (40, 512, 7, 7, 512, 3, 1, 1),
    (40, 256, 7, 7, 512, 1, 1, 0),
    # config 2 (BASELINE.json) at reduced batch, a wide image (row-split bands), a tall thin one
    (20, 128, 56, 56, 128, 3, 1, 1), (2, 128, 40, 300, 64, 3, 1, 1), (3, 32, 200, 9, 48, 3, 2, 1),
], ids=str)
def test_direct_layer_matches_two_launch_form_at_real_sizes(shape):
    N, C, H, W, O, k, s, p = shape
    x = gen.activation("normal" if C % 64 else "relu", gen.seed_of("fly", shape), (N, C, H, W))
    x[0, :, 0, :3] = 0.0
    w = gen.conv_weight("kaiming", gen.seed_of("flyw", shape), (O, C, k, k))
    pw = hipops.pack_weight(dev(w))
    xd = dev(x)
    one = hipops.bconv2d_direct(xd, pw, stride=s, padding=p)
    two = hipops.bconv2d(hipops.pack_act(xd), pw, stride=s, padding=p)
    assert torch.equal(one, two)


@pytest.mark.parametrize("shape", [(3, 128, 40, 24, 96, 3, 1, 1), (2, 64, 33, 20, 64, 3, 2, 1), (2, 256, 20, 20, 64, 3, 1, 1),
                                   (3, 200, 19, 19, 48, 1, 1, 0), (2, 32, 30, 30, 16, 5, 1, 2)], ids=str)
def test_units_with_and_without_negative_inputs_in_one_launch(shape):
    """A unit whose receptive fields hold no negative value takes the P-plane-only loop (v_and + v_bcnt, agreements);
    the others the two-plane loop.  Here the sign structure changes inside a launch — whole images, the upper half of
    an image, single rows and single pixels are non-negative — and every band plan must still give the bits of the
    two-launch form."""
    N, C, H, W, O, k, s, p = shape
    x = gen.activation("normal", gen.seed_of("mixed", shape), (N, C, H, W))
    x[0] = np.maximum(x[0], 0)                       # image 0: a ReLU output
    x[1, :, : H // 2] = np.abs(x[1, :, : H // 2])    # image 1: upper half non-negative
    x[1, :, H // 2 + 3, :] = np.maximum(x[1, :, H // 2 + 3, :], 0)
    if N > 2:
        x[2] = np.maximum(x[2], 0)
        x[2, C // 2, H - 1, W - 1] = -1.0            # image 2: ONE negative value, in the last pixel
    w = gen.conv_weight("kaiming", gen.seed_of("mixedw", shape), (O, C, k, k))
    pw = hipops.pack_weight(dev(w))
    xd = dev(x)
    two = hipops.bconv2d(hipops.pack_act(xd), pw, stride=s, padding=p)
    ho = two.shape[2]
    for pl in (None, _plan(1, ho, 16, 2), _plan(1, ho, 4, 1, prod=0), _plan(N, ho, 8, 4, prod=1),
               _plan(1, max(1, ho // 3), 8, 2, head=0, tail=0), _plan(1, 1, 2, 1, prod=1), _plan(2, ho, 16, 1, head=9)):
        assert torch.equal(hipops.bconv2d_direct(xd, pw, stride=s, padding=p, plan=pl), two), \
            None if pl is None else (pl.images_per_band, pl.rows_per_band, pl.waves, pl.blocks_per_unit)


def test_direct_layer_fuzz_against_two_launch_form():
    """Seeded shape fuzz: kernel sizes 1/3/5 (+ a 2x3), strides, paddings, dilations, ragged channel counts,
    zero weights, bias / post-scale — one launch == two launches, bit for bit."""
    rng = np.random.default_rng(20260927)
    n_done = 0
    for it in range(96):
        kh, kw = [(1, 1), (3, 3), (3, 3), (3, 3), (5, 5), (2, 3)][rng.integers(0, 6)]
        C = int(rng.choice([1, 3, 17, 32, 33, 64, 65, 96, 128, 129, 200, 256, 300, 520]))
        O = int(rng.choice([1, 5, 16, 31, 32, 33, 64, 70, 128]))
        H, W = int(rng.integers(1, 23)), int(rng.integers(1, 23))
        s = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        p = (int(rng.integers(0, 3)), int(rng.integers(0, 3)))
        d = (int(rng.integers(1, 3)), int(rng.integers(1, 3)))
        N = int(rng.integers(1, 6))
        ho, wo = hipops.conv_out_hw(H, W, kh, kw, s, p, d)
        if ho <= 0 or wo <= 0:
            continue
        x = gen.activation(["normal", "relu", "sparse", "special"][it % 4], 1000 + it, (N, C, H, W))
        w = gen.conv_weight("withzeros" if it % 7 == 0 else "kaiming", 2000 + it, (O, C, kh, kw))
        b = dev(np.float32(gen.normal(3000 + it, (O,)))) if it % 3 == 0 else None
        sc = dev(np.float32(0.5 + gen.uniform(4000 + it, (O,)))) if it % 5 == 0 else None
        pw = hipops.pack_weight(dev(w))
        xd = dev(x)
        one = hipops.bconv2d_direct(xd, pw, b, sc, s, p, d)
        two = hipops.bconv2d(hipops.pack_act(xd), pw, b, sc, s, p, d)
        assert torch.equal(one, two), (it, (N, C, H, W), (O, kh, kw), s, p, d)
        n_done += 1
    assert n_done >= 80


def test_workspace_query_is_zero_and_f32_entry_point_is_one_launch():
    case = LAYER_CASES_BY_NAME["c2_relu"]
    x, w, _, _, pw, kw = _args(case)
    lib = native.require()
    for c in LAYER_CASES:
        d = native.ConvDesc(c.N, c.C, c.H, c.W, c.O, c.k, c.k, c.stride, c.stride, c.pad, c.pad, c.dilation,
                            c.dilation, 0)
        assert lib.bnn_hip_conv_workspace_bytes(ctypes.byref(d)) == 0, c.name
    d = native.ConvDesc(case.N, case.C, case.H, case.W, case.O, 3, 3, 1, 1, 1, 1, 1, 1, 0)
    out = torch.empty((case.N, case.O, case.H, case.W), device=DEV)
    xd = dev(x)
    before = native.launch_count()
    native.check(lib.bnn_hip_bconv2d_f32(ctypes.byref(d), xd.data_ptr(), pw.wbits.data_ptr(), pw.wnz.data_ptr(),
                                         pw.alpha.data_ptr(), None, None, out.data_ptr(), None,
                                         torch.cuda.current_stream().cuda_stream), "bconv2d_f32")
    assert native.launch_count() == before + 1
    assert torch.equal(out, hipops.bconv2d(hipops.pack_act(xd), pw, **kw))
    # an image row wider than a CU's LDS can hold: the workspace form is still there
    big = native.ConvDesc(1, 512, 3, 2000, 32, 3, 3, 1, 1, 1, 1, 1, 1, 0)
    assert lib.bnn_hip_conv_workspace_bytes(ctypes.byref(big)) > 0
    xb = dev(gen.activation("normal", 77, (1, 512, 3, 2000)))
    pwb = hipops.pack_weight(dev(gen.conv_weight("kaiming", 78, (32, 512, 3, 3))))
    assert hipops.direct_plan(xb.shape, pwb, padding=1) is None
    assert torch.equal(hipops.bconv2d_direct(xb, pwb, padding=1),
                       hipops.bconv2d(hipops.pack_act(xb), pwb, padding=1))


def test_conv2d_forward_is_one_launch(golden_layers):
    layer, x = make_layer(LAYER_CASES_BY_NAME["c2_relu"])
    xd = dev(x)
    with torch.no_grad():
        layer(xd)                                   # packs the weight
        before, n = native.launch_count(), fastpath.stats()["conv2d"]
        out = layer(xd)
    assert native.launch_count() == before + 1 and fastpath.stats()["conv2d"] == n + 1
    assert close(out.cpu().numpy(), golden_layers["c2_relu/out"])


def test_config2_full_size_properties():
    """BASELINE config 2 at its stated size (256 x 128 x 56 x 56 -> 128): one launch == two launches on a
    64-image slice; any sub-batch gives the same bits (bands are independent); a second run is identical."""
    N = 256
    x = torch.from_numpy(gen.activation("relu", 7, (8, 128, 56, 56))).to(DEV).repeat(N // 8, 1, 1, 1)
    x = x * (1.0 + torch.arange(N, device=DEV, dtype=torch.float32).view(N, 1, 1, 1) / N) - 0.3
    pw = hipops.pack_weight(torch.from_numpy(gen.conv_weight("kaiming", 8, (128, 128, 3, 3))).to(DEV))
    out = hipops.bconv2d_direct(x, pw, padding=1)
    assert torch.equal(out[64:128], hipops.bconv2d(hipops.pack_act(x[64:128]), pw, padding=1))
    assert torch.equal(out[200:203], hipops.bconv2d_direct(x[200:203], pw, padding=1))
    assert torch.equal(out, hipops.bconv2d_direct(x, pw, padding=1))
    half = _plan(1, 28, 8, 2)
    assert torch.equal(out, hipops.bconv2d_direct(x, pw, padding=1, plan=half))
