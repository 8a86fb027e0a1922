"""BASELINE config 3 at the size the headline metric is quoted on: binary ResNet-18, 224x224, batch 256.

Parity against the REFERENCE's own forward on 256 distinct images (fixture tests/golden/resnet18_b256.npz,
written by tests/golden/make_golden.py from /root/reference: fp32 logits, the same model evaluated in fp64,
and a position-weighted checksum of the sign() result in front of each of the 19 binary convolutions).

A binarised network is discontinuous: an activation within rounding distance of 0 in front of a sign()
lands on either side depending on summation order.  The reference itself does that — the fixture records
that switching only its convolution backend (oneDNN -> ATen native) flips a sign somewhere in 42 of these
256 images and moves logits by up to 0.15, while wherever no sign flips the logits agree to 3e-6.  So the
contract ("within 1e-3 of the reference forward") is tested in two parts:

  (1) STRICT: every image whose 19 sign checksums all equal the reference's is within 1e-3 (in fact 1e-4),
  (2) COUNTED: the number of images with a flipped sign is bounded by the measured value (deterministic
      kernels), reported per path together with the first layer that diverges, and compared with the
      reference's distance to its own fp64 evaluation.
"""
import contextlib
import json
import os

import numpy as np
import pytest
import torch

import bnn_amd as bnn
from bnn_amd import inference
from bnn_amd.inference import FusedResNet, PipelinedInference
from bnn_amd.models import HBlock, ResNet, resnet18
from bnn_amd.ops import BasicInputBinarizer, XNORWeightBinarizer
from tests.golden import gen, sighash

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
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

# images (of 256) allowed to differ in at least one sign() from (the reference's fp32 forward, the reference evaluated
# in fp64): exactly the values measured on MI355X in rounds 2, 3 and 4 (profiles/r0N_c3_b256_parity_*.json: 2/3, 6/3,
# 6/3, 3/2 — the kernels are deterministic and the stem has been bit-stable for three rounds: no allowance on top).
# A regression from 2 flipped images to 3 must not pass.
# "layerwise" (round 4: the stem kernel and the one-launch BatchNorm tails of the per-layer path) computes the fused
# executor's integers; "layerwise_library" (torch's own stem / BatchNorm / ReLU / add around the binary layers) is the
# independent composition the 2/3 were measured on.
MAX_FLIPPED = {"layerwise_library": (2, 3), "layerwise": (6, 3), "fused": (6, 3), "fused_exact_stem": (3, 2)}
# every flip starts where the reference's own fp32 rounding decides: behind the real-valued stem / first residual sums
EARLY_LAYERS = ("layer1.", "layer2.0.conv1", "layer2.0.downsample")


def _r18(ctor=resnet18):
    cfg = bnn.BConfig(activation_pre_process=BasicInputBinarizer, activation_post_process=bnn.Identity,
                      weight_pre_process=XNORWeightBinarizer)
    net = bnn.prepare_binary_model(ctor(), cfg, custom_config_layers_name={
        "conv1": bnn.BConfig(), "fc": bnn.BConfig()})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gen.model_state(shapes, 1).items()})
    return net.to(DEV).eval()


@pytest.fixture(scope="module")
def fixture():
    g = np.load(os.path.join(HERE, "golden", "resnet18_b256.npz"))
    return {k: g[k] for k in g.files}


@pytest.fixture(scope="module")
def images():
    return torch.from_numpy(gen.normal(gen.seed_of("r18", "b256"), (256, 3, 224, 224))).to(DEV)


def _run_layerwise(net, x, names, library=False):
    hashes = {n: None for n in names}
    hooks = []
    mods = dict(net.named_modules())
    for n in names:
        def pre(m, inp, n=n):
            hashes[n] = sighash.sign_hash_torch(inp[0])
        hooks.append(mods[n].register_forward_pre_hook(pre))
    with torch.no_grad(), (inference.library_tails() if library else contextlib.nullcontext()):
        y = net(x)
    for h in hooks:
        h.remove()
    return y, torch.stack([hashes[n] for n in names], 1)


def _run_fused(net, x, names, **kw):
    fused = FusedResNet(net, **kw)
    hashes = {}

    def tap(name, act):
        hashes[name] = sighash.sign_hash_planes(act.P, act.M, act.shape[1])
    with inference.tap_binary_inputs(tap):
        y = fused(x)
    assert set(hashes) == set(names), sorted(set(names) ^ set(hashes))
    return y, torch.stack([hashes[n] for n in names], 1)


def _compare(y, h, ref, href, names):
    tol = 1e-3 * np.abs(ref).max() + 1e-3 * np.abs(ref)
    dev_ = np.abs(y - ref)
    ok = np.all(dev_ <= tol, 1)
    flipped = np.any(h != href, 1)
    first = [int(np.argmax(h[i] != href[i])) for i in np.nonzero(flipped)[0]]
    return {"images": int(len(ok)), "within_tol": int(ok.sum()), "images_with_a_sign_flip": int(flipped.sum()),
            "max_abs_logit_dev": float(dev_.max()),
            "max_dev_without_flip": float(dev_[~flipped].max()) if (~flipped).any() else 0.0,
            "argmax_agree": int((y.argmax(1) == ref.argmax(1)).sum()),
            "first_diverging_layer_histogram": {names[k]: first.count(k) for k in sorted(set(first))}}, ok, flipped


@pytest.mark.parametrize("path", ["layerwise_library", "layerwise", "fused", "fused_exact_stem"])
def test_c3_batch256_against_reference_forward(path, fixture, images):
    names = [str(n) for n in fixture["layers"]]
    net = _r18()
    if path.startswith("layerwise"):
        y, h = _run_layerwise(net, images, names, library=path == "layerwise_library")
        if path == "layerwise":      # same stem kernel, same float operations in the tails: the fused executor's integers
            assert torch.equal(h, _run_fused(net, images, names)[1])
    else:
        y, h = _run_fused(net, images, names, stem_exact_fp32=(path == "fused_exact_stem"))
    y, h = y.cpu().numpy(), h.cpu().numpy()
    ref, href = fixture["logits"], fixture["sign_hash"]
    rep, ok, flipped = _compare(y, h, ref, href, names)
    rep64, ok64, flipped64 = _compare(y, h, fixture["logits_f64"], fixture["sign_hash_f64"], names)
    report = {"path": path, "vs_reference_fp32": rep, "vs_reference_evaluated_in_fp64": rep64,
              "reference_self_check": json.loads(str(fixture["self_check"]))}
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"c3_b256_parity_{path}.json"), "w") as fh:
            json.dump(report, fh, indent=1)
    print(json.dumps({k: report[k] for k in ("path", "vs_reference_fp32", "vs_reference_evaluated_in_fp64")}))
    # (1) same integers everywhere  =>  same logits up to fp32 rounding of the real-valued layers
    assert ok[~flipped].all() and ok64[~flipped64].all()
    assert rep["max_dev_without_flip"] <= 1e-4 * np.abs(ref).max()
    # (2) how many images saw a sign() decided differently (the reference vs its own fp64 evaluation: 3)
    assert rep["images_with_a_sign_flip"] <= MAX_FLIPPED[path][0], rep
    assert rep64["images_with_a_sign_flip"] <= MAX_FLIPPED[path][1], rep64
    assert rep["within_tol"] >= 256 - MAX_FLIPPED[path][0]
    for r in (rep, rep64):
        assert all(n.startswith(EARLY_LAYERS) for n in r["first_diverging_layer_histogram"]), r


def test_c3_batch256_properties(images):
    """Size-independent properties of the benched configuration (fused executor, batch 256, 224x224):
    images are independent (any sub-batch gives the same bits), graph replay == eager launches,
    two batches in flight == one at a time."""
    net = _r18()
    fused = FusedResNet(net)
    y = fused(images).clone()
    assert y.shape == (256, 1000) and torch.isfinite(y).all()
    for lo, hi in ((0, 8), (100, 116), (251, 256)):
        assert torch.equal(fused(images[lo:hi].contiguous()), y[lo:hi])
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(3)).to(DEV)
    assert torch.equal(fused(images[perm].contiguous()), y[perm])
    fused.capture(images)
    assert torch.equal(fused(images), y)
    pipe = PipelinedInference(net, images, n_streams=2)
    other = torch.roll(images, 1, 0)
    with torch.cuda.stream(pipe.stream(1)):
        pipe.input(1).copy_(other)
    outs = [pipe.launch(i) for i in range(2)]
    pipe.synchronize()
    assert torch.equal(outs[0], y) and torch.equal(outs[1], torch.roll(y, 1, 0))


def _binary_conv_names(net):
    return [n for n, m in net.named_modules()
            if isinstance(m, bnn.layers.Conv2d) and isinstance(m.activation_pre_process, BasicInputBinarizer)]


C5_FLIPPED = 0          # measured (round 6, MI355X): no image of the 32 differs from the reference in any sign()
C5_FP16_ARGMAX = 32     # measured: the plain-fp16 stem keeps the class of all 32 images


def test_c5_hblock_3463_at_its_stated_size():
    """BASELINE config 5 at full size on one GPU's share: the build-defined ResNet(HBlock,[3,4,6,3]) (the
    reference cannot construct it), 224x224, 128 images, fp16 MFMA stem: properties at full size.
    Parity (round 5; round 6: 32 images, measured counts): the first 32 images against the REFERENCE — tests/golden/hblock_net_b32.npz, the same [3,4,6,3]
    stack assembled from the reference's own HBlock / BatchNorm / shortcut modules by tests/golden/make_golden.py
    (`hblock_net`): fused executor and per-layer path both against its fp32 logits and the sign checksums in front of
    all 51 binary convolutions, with the strict / counted split of config 3."""
    net = _r18(lambda: ResNet(HBlock, [3, 4, 6, 3]))
    x = torch.from_numpy(gen.normal(gen.seed_of("c5", "b128"), (128, 3, 224, 224))).to(DEV)
    fused16 = FusedResNet(net, stem_fp16=True)
    y = fused16(x).clone()
    assert y.shape == (128, 1000) and torch.isfinite(y).all()
    for lo, hi in ((0, 8), (60, 70), (123, 128)):
        assert torch.equal(fused16(x[lo:hi].contiguous()), y[lo:hi])
    fused16.capture(x)
    assert torch.equal(fused16(x), y)
    g = np.load(os.path.join(HERE, "golden", "hblock_net_b32.npz"))
    names = [str(n) for n in g["layers"]]
    assert names == _binary_conv_names(net) or sorted(names) == sorted(_binary_conv_names(net))
    assert len(names) == 3 * 16 + 3                     # 16 HBlocks x 3 convs + 3 binary 1x1 shortcuts
    assert [str(k) for k in g["state_keys"]] == list(net.state_dict().keys())      # the same model, key by key
    ref, href = g["logits"], g["sign_hash"]
    xs = x[:32].contiguous()
    report = {"reference_self_check": json.loads(str(g["self_check"]))}
    for path, (yy, hh) in (("layerwise", _run_layerwise(net, xs, names)), ("fused", _run_fused(net, xs, names))):
        rep, ok, flipped = _compare(yy.cpu().numpy(), hh.cpu().numpy(), ref, href, names)
        report[path] = rep
        # (1) strict: same integers everywhere => logits within 1e-3 (measured: 1e-5 of the largest logit)
        assert ok[~flipped].all(), rep
        assert rep["max_dev_without_flip"] <= 1e-4 * np.abs(ref).max(), rep
        # (2) counted: 32 images x 51 layers.  Measured on MI355X (round 6), fused and per-layer alike: C5_FLIPPED images
        # with a sign decided differently from the reference's fp32 forward (the reference against its own fp64
        # evaluation: 1 of 32; against its other conv backend: 0) — pinned at the measured count, as config 3 is
        assert rep["images_with_a_sign_flip"] <= C5_FLIPPED, rep
    # the tapped run above is launch by launch (the taps want the planes in front of every convolution); what runs without
    # a tap is one launch per hierarchical block (csrc/hblock.hip): the same logits bit for bit
    from bnn_amd import native
    eng = FusedResNet(net)
    assert torch.equal(eng(xs), yy)
    n0 = native.launch_count()
    eng(xs)
    # stem (writes block 1's planes) + 16 blocks + the 14x14 -> 7x7 pool + 2 head launches: the stage ends at 56x56 and
    # 28x28 pool and binarise their own output (bnn_hip_hblock_pool_forward), the first block of the 28x28, 14x14 and 7x7
    # stages computes its shortcut convolution itself (bnn_hip_hblock_shortcut_forward)
    assert native.launch_count() - n0 == 20
    assert torch.equal(FusedResNet(net, fuse_hblock=False)(xs), yy)
    # the plan of several batches in flight (whole images per workgroup, lanes = channels on 14x14 too): the same bits
    assert torch.equal(FusedResNet(net, throughput_mode=True)(xs), yy)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "c5_b32_parity.json"), "w") as fh:
            json.dump(report, fh, indent=1)
    print(json.dumps({k: report[k] for k in ("layerwise", "fused")}))
    # the fp16 stem is a precision trade (5e-4 relative in the stem): same classes for almost every image
    agree = int((y[:32].argmax(1).cpu().numpy() == ref.argmax(1)).sum())
    print(json.dumps({"fp16_stem_argmax_agree_of_32": agree}))
    assert agree >= C5_FP16_ARGMAX
