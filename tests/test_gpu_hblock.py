"""The hierarchical block in one launch (csrc/hblock.hip, bnn_hip_hblock_forward): bit for bit the three
bnn_hip_bconv2d_fused launches it replaces (fp32 output AND the next block's sign planes), on every plan (whole images,
several images per workgroup, bands of rows with their halos), and within tolerance of the reference's formulation
(bnn/models/layers/hierarchical_block.py:38-60) out of torch float ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from bnn_amd import hipops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _block(seed, c_in, planes, N, H, W):
    g = torch.Generator().manual_seed(seed)
    half, quarter = planes // 2, planes // 4

    def rnd(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).to(DEV)

    def bn(c):
        return (torch.rand(c, generator=g) + 0.5).to(DEV), rnd(c, scale=0.3)

    ws = [rnd(half, c_in, 3, 3, scale=0.05), rnd(quarter, half, 3, 3, scale=0.05), rnd(quarter, quarter, 3, 3, scale=0.05)]
    x = rnd(N, c_in, H, W)
    res = rnd(N, planes, H, W)
    return x, res, ws, bn(c_in), bn(half), bn(quarter), bn(planes)


def _launch_by_launch(p_in, pws, bn2, bn3, nbn, res):
    """bnn_amd/executor.py: the three slice-writing launches + the packing pass of the next block."""
    N, _, H, W = res.shape
    planes = res.shape[1]
    half, quarter = planes // 2, planes // 4
    y = torch.empty_like(res)
    late = dict(residual=res, residual_after_act=True, pack_before_residual=True, out=y, out_f32=True, padding=1)
    _, p = hipops.bconv2d_fused(p_in, pws[0], out_packed=True, out_c_offset=0, pack_scale=bn2[0], pack_shift=bn2[1],
                                pack_relu=True, **late)
    _, p = hipops.bconv2d_fused(p, pws[1], out_packed=True, out_c_offset=half, pack_scale=bn3[0], pack_shift=bn3[1],
                                pack_relu=True, **late)
    hipops.bconv2d_fused(p, pws[2], out_packed=False, out_c_offset=half + quarter, **late)
    return y, hipops.bn_act_pack(y, nbn[0], nbn[1], relu=True)


SHAPES = [  # c_in, planes, N, H, W
    (64, 64, 3, 56, 56), (64, 128, 3, 28, 28), (128, 128, 3, 28, 28), (128, 256, 5, 14, 14), (256, 256, 5, 14, 14),
    (256, 512, 9, 7, 7), (512, 512, 9, 7, 7),
    (64, 64, 2, 13, 9), (128, 128, 2, 5, 7), (256, 256, 3, 3, 3), (512, 512, 2, 1, 1),
]
PLANS = [dict(), dict(rows_per_band=4), dict(rows_per_band=5), dict(images_per_band=2), dict(images_per_band=4),
         dict(throughput=True), dict(waves=4), dict(rows_per_band=1)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_one_launch_equals_three_launches(shape):
    c_in, planes, N, H, W = shape
    x, res, ws, bn1, bn2, bn3, nbn = _block(hash(shape) % 1000, c_in, planes, N, H, W)
    p_in = hipops.bn_act_pack(x, bn1[0], bn1[1], relu=True)
    pws = [hipops.pack_weight(w) for w in ws]
    want_y, want_p = _launch_by_launch(p_in, pws, bn2, bn3, nbn, res)
    pack = hipops.hblock_pack(*pws, bn2, bn3, nbn)
    pack_last = hipops.hblock_pack(*pws, bn2, bn3, None)
    for plan in PLANS:
        if plan.get("rows_per_band", 0) > H or plan.get("images_per_band", 0) > N:
            continue
        if not hipops.hblock_supported(N, c_in, H, W, planes, **plan):     # (several large images exceed the LDS)
            assert plan.get("images_per_band", 0) > 1 and H * W > 1000
            continue
        y, p = hipops.hblock_forward(p_in, pack, res, **plan)
        assert torch.equal(y, want_y), plan
        assert torch.equal(p.P, want_p.P), plan
        assert p.nonneg and not bool(p.M.any())
        y2, p2 = hipops.hblock_forward(p_in, pack_last, res, out_packed=False, **plan)
        assert p2 is None and torch.equal(y2, want_y), plan
    # the small-image form (lanes = output channels): 14 x 14 / 7 x 7, widths from 256 on
    cl = hipops.hblock_supported(N, c_in, H, W, planes, channel_lanes=True)
    assert cl == ((H, W) in ((14, 14), (7, 7)) and planes % 256 == 0)
    if cl:
        for waves in (0, 4, 3):
            y, p = hipops.hblock_forward(p_in, pack, res, channel_lanes=True, waves=waves)
            assert torch.equal(y, want_y) and torch.equal(p.P, want_p.P), waves
        y2, p2 = hipops.hblock_forward(p_in, pack_last, res, out_packed=False, channel_lanes=True)
        assert p2 is None and torch.equal(y2, want_y)


def test_against_the_float_formulation():
    """The reference's op sequence in torch float ops (sign -> conv2d with sign(W) * alpha -> cat -> + residual)."""
    c_in, planes, N, H, W = 128, 128, 4, 28, 28
    x, res, ws, bn1, bn2, bn3, nbn = _block(7, c_in, planes, N, H, W)

    def binconv(t, w):
        alpha = w.abs().mean(dim=(1, 2, 3), keepdim=True)
        return F.conv2d(torch.sign(t), torch.sign(w) * alpha, padding=1)

    def bnrelu(t, bn):
        return torch.relu(t * bn[0][None, :, None, None] + bn[1][None, :, None, None])

    o1 = binconv(bnrelu(x, bn1), ws[0])
    o2 = binconv(bnrelu(o1, bn2), ws[1])
    o3 = binconv(bnrelu(o2, bn3), ws[2])
    want = torch.cat((o1, o2, o3), 1) + res
    p_in = hipops.bn_act_pack(x, bn1[0], bn1[1], relu=True)
    pack = hipops.hblock_pack(*[hipops.pack_weight(w) for w in ws], bn2, bn3, nbn)
    y, _ = hipops.hblock_forward(p_in, pack, res)
    # a sign() in front of conv2 / conv3 flips where the float conv's rounding crosses zero: compare where it did not
    close = torch.isclose(y, want, rtol=1e-3, atol=1e-4)
    assert close.float().mean().item() > 0.999
    assert torch.allclose(y[:, :planes // 2], want[:, :planes // 2], rtol=1e-3, atol=1e-4)   # conv1: no sign() between


def test_argument_checks():
    c_in, planes, N, H, W = 64, 64, 2, 8, 8
    x, res, ws, bn1, bn2, bn3, nbn = _block(3, c_in, planes, N, H, W)
    p_in = hipops.bn_act_pack(x, bn1[0], bn1[1], relu=True)
    pws = [hipops.pack_weight(w) for w in ws]
    pack = hipops.hblock_pack(*pws, bn2, bn3, None)
    from bnn_amd import native
    with pytest.raises(native.NativeError):
        hipops.hblock_forward(p_in, pack, res, out_packed=True)          # no next-block constants in this pack
    with pytest.raises(native.NativeError):
        hipops.hblock_forward(p_in, pack, res[:, :32], out_packed=False)  # residual of the wrong width
    assert hipops.hblock_supported(128, 64, 56, 56, 64)
    assert not hipops.hblock_supported(128, 64, 56, 56, 96)              # planes % 64
    assert not hipops.hblock_supported(2, 96, 8, 8, 64)                  # three-word input cells
    wz = ws[0].clone()
    wz[0, 0, 0, 0] = 0.0
    with pytest.raises(native.NativeError):
        hipops.hblock_pack(hipops.pack_weight(wz), pws[1], pws[2], bn2, bn3, None)


@pytest.mark.parametrize("shape", [(3, 64, 56, 56), (2, 128, 28, 28), (5, 256, 14, 14), (2, 96, 6, 10)],
                         ids=lambda s: "x".join(map(str, s)))
def test_pool_and_both_packs_in_one_pass(shape):
    """bnn_hip_avgpool2_bn_pack2_f32 == ATen's avg_pool2d followed by the two packing passes, bit for bit."""
    N, C, H, W = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, C, H, W, generator=g).to(DEV)
    x[0, 0, :2, :2] = 0.0                                # an exactly-zero window: neither plane
    x[0, 1, 0, 0] = float("nan")
    bn1 = ((torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.3).to(DEV))
    bn2 = ((torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.3).to(DEV))
    t = F.avg_pool2d(x, 2, 2, ceil_mode=True, count_include_pad=False)
    w1 = hipops.bn_act_pack(t, *bn1, relu=True)
    w2 = hipops.bn_act_pack(t, *bn2, relu=False)
    p1, p2, tp = hipops.avgpool2_bn_pack2(x, bn1, True, bn2, False, out_f32=True)
    assert torch.equal(tp[0, 2:], t[0, 2:]) and torch.equal(tp[1:], t[1:])     # (NaN != NaN in channel 1 of image 0)
    assert torch.equal(p1.P, w1.P) and torch.equal(p1.M, w1.M) and p1.nonneg
    assert torch.equal(p2.P, w2.P) and torch.equal(p2.M, w2.M) and not p2.nonneg
    q1, q2, tq = hipops.avgpool2_bn_pack2(x, bn1, True)
    assert q2 is None and tq is None and torch.equal(q1.P, w1.P)


POOL_SHAPES = [  # planes, N, H, W
    (64, 3, 56, 56), (128, 3, 28, 28), (256, 5, 14, 14), (64, 2, 12, 10), (128, 2, 6, 6), (256, 3, 2, 2), (64, 5, 8, 24),
    (128, 130, 28, 28),
]
POOL_PLANS = [dict(), dict(rows_per_band=4), dict(rows_per_band=6), dict(rows_per_band=2), dict(images_per_band=2),
              dict(images_per_band=4), dict(throughput=True), dict(waves=4), dict(waves=3, rows_per_band=8)]


@pytest.mark.parametrize("shape", POOL_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_last_block_of_a_stage_with_the_pool_and_both_binarisations(shape):
    """bnn_hip_hblock_pool_forward: block + AvgPool2d(2, 2) + sign(relu(bn1(.))) + sign(bn_ds(.)) in one launch, no fp32
    output — the planes of bnn_hip_hblock_forward's y through bnn_hip_avgpool2_bn_pack2_f32, bit for bit, on every plan
    (whole images, several per workgroup, bands of whole windows with their halos)."""
    planes, N, H, W = shape
    x, res, ws, bn1, bn2, bn3, nbn = _block(hash(shape) % 1000, planes, planes, N, H, W)
    g = torch.Generator().manual_seed(7)
    ds = ((torch.rand(planes, generator=g) + 0.5).to(DEV) * torch.where(torch.arange(planes) % 5 == 0, -1.0, 1.0).to(DEV),
          (torch.randn(planes, generator=g) * 0.3).to(DEV))
    p_in = hipops.bn_act_pack(x, bn1[0], bn1[1], relu=True)
    pws = [hipops.pack_weight(w) for w in ws]
    pack = hipops.hblock_pack(*pws, bn2, bn3, None)
    y, _ = hipops.hblock_forward(p_in, pack, res, out_packed=False)
    w1, w2, _ = hipops.avgpool2_bn_pack2(y, nbn, True, ds, False, out_f32=False)
    assert bool(w2.M.any()) and bool(w2.P.any())
    kp = hipops.hblock_pool_consts(nbn, ds, planes)
    ran = 0
    for plan in POOL_PLANS:
        if plan.get("rows_per_band", 0) > H or plan.get("images_per_band", 0) > N:
            continue
        if not hipops.hblock_pool_supported(N, planes, H, W, planes, **plan):
            assert plan.get("images_per_band", 0) > 1 and H * W > 1000
            continue
        p1, p2 = hipops.hblock_pool_forward(p_in, pack, res, kp, **plan)
        assert torch.equal(p1.P, w1.P), plan
        assert torch.equal(p2.P, w2.P) and torch.equal(p2.M, w2.M), plan
        assert p1.nonneg and not bool(p1.M.any()) and not p2.nonneg
        ran += 1
    assert ran >= 4


def test_pool_form_argument_checks():
    from bnn_amd import native
    assert not hipops.hblock_pool_supported(2, 64, 13, 9, 64)              # odd image
    assert not hipops.hblock_pool_supported(2, 64, 28, 28, 128)            # the first block of a stage: other widths
    assert not hipops.hblock_pool_supported(2, 512, 14, 14, 512)           # no instance
    assert not hipops.hblock_pool_supported(2, 64, 28, 28, 64, rows_per_band=5)   # bands of whole windows
    x, res, ws, bn1, bn2, bn3, nbn = _block(3, 64, 64, 2, 8, 8)
    p_in = hipops.bn_act_pack(x, bn1[0], bn1[1], relu=True)
    pack = hipops.hblock_pack(*[hipops.pack_weight(w) for w in ws], bn2, bn3, None)
    kp = hipops.hblock_pool_consts(nbn, nbn, 64)
    with pytest.raises(native.NativeError):
        hipops.hblock_pool_forward(p_in, pack, res, kp[:-1])
    with pytest.raises(native.NativeError):
        hipops.hblock_pool_forward(p_in, pack, res[:, :, :7], kp)
    with pytest.raises(native.NativeError):
        hipops.hblock_pool_forward(p_in, pack, res, kp, rows_per_band=3)


SC_SHAPES = [(64, 3, 28, 28), (128, 5, 14, 14), (64, 2, 9, 13), (128, 2, 5, 5), (64, 130, 28, 28), (256, 9, 7, 7), (128, 3, 7, 7),
             (256, 3, 14, 14)]   # C_in (planes = 2 C_in), N, H, W


@pytest.mark.parametrize("shape", SC_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_first_block_of_a_stage_with_its_shortcut_convolution_inside(shape):
    """bnn_hip_hblock_shortcut_forward: the block whose shortcut is BatchNorm -> sign -> binary conv1x1 of its input
    (hierarchical_block.py:30-36) computes that convolution per pass from the two sign planes — bit for bit
    bnn_hip_bconv2d (the 1x1 launch, fp32 out) + bnn_hip_hblock_forward on its output, fp32 tensor and next planes alike."""
    c_in, N, H, W = shape
    planes = 2 * c_in
    x, res, ws, bn1, bn2, bn3, nbn = _block(hash(shape) % 1000, c_in, planes, N, H, W)
    g = torch.Generator().manual_seed(11)
    ds_bn = ((torch.rand(c_in, generator=g) + 0.5).to(DEV) * torch.where(torch.arange(c_in) % 3 == 0, -1.0, 1.0).to(DEV),
             (torch.randn(c_in, generator=g) * 0.3).to(DEV))
    w_sc = (torch.randn(planes, c_in, 1, 1, generator=g) * 0.05).to(DEV)
    x[:, :, 0, 0] = -ds_bn[1] / ds_bn[0]                       # some exact zeros of the shortcut's BatchNorm: sign() == 0
    p_in = hipops.bn_act_pack(x, bn1[0], bn1[1], relu=True)
    p_sc = hipops.bn_act_pack(x, ds_bn[0], ds_bn[1], relu=False)
    assert bool(p_sc.M.any()) and bool(p_sc.P.any())
    pw_sc = hipops.pack_weight(w_sc)
    idn = hipops.bconv2d(p_sc, pw_sc)
    pack = hipops.hblock_pack(*[hipops.pack_weight(w) for w in ws], bn2, bn3, nbn)
    want_y, want_p = hipops.hblock_forward(p_in, pack, idn)
    sc_pack = hipops.hblock_shortcut_pack(pw_sc)
    ran = 0
    pixel_lanes = hipops.hblock_shortcut_supported(N, c_in, H, W, planes)
    assert pixel_lanes == (c_in in (64, 128))              # (256 -> 512: the small-image form only)
    for plan in PLANS if pixel_lanes else ():
        if plan.get("rows_per_band", 0) > H or plan.get("images_per_band", 0) > N:
            continue
        if not hipops.hblock_shortcut_supported(N, c_in, H, W, planes, **plan):
            assert plan.get("images_per_band", 0) > 1 and H * W > 1000
            continue
        y, p = hipops.hblock_shortcut_forward(p_in, pack, p_sc, sc_pack, **plan)
        assert torch.equal(y, want_y), plan
        assert torch.equal(p.P, want_p.P) and p.nonneg, plan
        ran += 1
    assert ran >= 4 or not pixel_lanes
    # the small-image form (lanes = output channels) has the same variant from 128 input channels on
    cl = hipops.hblock_shortcut_supported(N, c_in, H, W, planes, channel_lanes=True)
    assert cl == ((H, W) in ((14, 14), (7, 7)) and c_in in (128, 256))
    if cl:
        for waves in (0, 4, 3):
            y, p = hipops.hblock_shortcut_forward(p_in, pack, p_sc, sc_pack, channel_lanes=True, waves=waves)
            assert torch.equal(y, want_y) and torch.equal(p.P, want_p.P), waves


def test_shortcut_form_argument_checks():
    from bnn_amd import native
    assert not hipops.hblock_shortcut_supported(2, 64, 28, 28, 64)           # not a width-changing block
    assert not hipops.hblock_shortcut_supported(2, 256, 14, 14, 512)         # no instance
    x, res, ws, bn1, bn2, bn3, nbn = _block(5, 64, 128, 2, 8, 8)
    p_in = hipops.bn_act_pack(x, bn1[0], bn1[1], relu=True)
    p_sc = hipops.bn_act_pack(x, bn1[0], bn1[1], relu=False)
    pws = [hipops.pack_weight(w) for w in ws]
    sc_pack = hipops.hblock_shortcut_pack(hipops.pack_weight(torch.randn(128, 64, 1, 1, device=DEV)))
    with pytest.raises(native.NativeError):                                  # a pack without the next block's BatchNorm
        hipops.hblock_shortcut_forward(p_in, hipops.hblock_pack(*pws, bn2, bn3, None), p_sc, sc_pack)
    with pytest.raises(native.NativeError):
        hipops.hblock_shortcut_pack(pws[0])                                  # a 3x3 pack
    wz = torch.randn(128, 64, 1, 1, device=DEV)
    wz[3, 5] = 0.0
    with pytest.raises(native.NativeError):
        hipops.hblock_shortcut_pack(hipops.pack_weight(wz))                  # zero weights take the separate launch
