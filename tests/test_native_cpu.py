"""The C-ABI library loads on a GPU-less machine and exports every symbol include/bnn_hip.h
declares.  No kernel is launched here (host-only entry points are exercised)."""
import ctypes
import os
import re

import pytest

import oracle
from bnn_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "bnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bnn_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(native.lib_path()), "build with __graft_entry__.build() first"
    lib = ctypes.CDLL(native.lib_path())
    syms = header_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/bnn_hip.h but not exported"
    assert set(native.EXPORTED_SYMBOLS) == set(syms)


def test_legacy_kernels_live_in_their_own_test_only_library():
    """ABI 12: the round-2 stem kernel and the LDS-staged weight tile are cross-check material, not product code —
    libbnn_hip.so does not contain them, libbnn_hip_legacy.so (loaded only by tests/helpers/legacy.py) does."""
    import subprocess
    legacy = os.path.join(os.path.dirname(native.lib_path()), "libbnn_hip_legacy.so")
    assert os.path.exists(legacy), "build with __graft_entry__.build() first"
    ll = ctypes.CDLL(legacy)
    assert hasattr(ll, "bnn_hip_legacy_stem_staged") and hasattr(ll, "bnn_hip_legacy_bconv2d_lds")
    product = ctypes.CDLL(native.lib_path())
    assert not hasattr(product, "bnn_hip_legacy_stem_staged")
    names = subprocess.run(["strings", "-n", "12", native.lib_path()], capture_output=True, text=True).stdout
    assert "stem_split_kernel" not in names and "bconv_lds_kernel" not in names
    assert "stem_rows_kernel" in names and "bconv_sgpr_kernel" in names
    src = os.path.join(ROOT, "binary-networks-pytorch_amd", "bnn_amd")
    for fn in os.listdir(src):
        if fn.endswith(".py"):
            assert "libbnn_hip_legacy" not in open(os.path.join(src, fn)).read(), fn


def test_capi_validators_survive_an_argument_fuzz_under_asan_and_ubsan():
    """tools/fuzz/capi_fuzz.hip: csrc/capi.hip compiled HOST-ONLY with AddressSanitizer + UBSan against stub launchers
    that check the contract the kernels rely on (tensor sizes below the 32-bit addressing limits, alignments, a
    consistent weight layout).  Edge-value integers and null / misaligned pointers at every entry point: no crash, no
    signed overflow in the size arithmetic, no launcher reached with arguments that break the contract.  (Round 4:
    this found three overflowing limit checks — N*H*W*words in check_desc, o_pad*taps in bnn_hip_weight_layout,
    2*pad in the max-pool entry — and an unbounded bnn_hip_conv_workspace_bytes.)"""
    import subprocess
    csrc = os.path.join(ROOT, "binary-networks-pytorch_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "fuzz"], check=True, capture_output=True)
    exe = os.path.join(ROOT, "build", "fuzz", "capi_fuzz")
    for seed in ("0", "1", "2"):
        out = subprocess.run([exe, "150000", seed], capture_output=True, text=True,
                             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert out.returncode == 0 and "CAPI_FUZZ_OK" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_require_loads_and_reports_abi():
    lib = native.require()
    assert lib.bnn_hip_abi_version() == native.ABI_VERSION == 15
    assert lib.bnn_hip_status_string(0) == b"ok"
    assert b"invalid" in lib.bnn_hip_status_string(-1)
    assert isinstance(native.launch_count(), int)


def test_host_only_entry_points():
    lib = native.require()
    assert lib.bnn_hip_act_words(1) == 1 and lib.bnn_hip_act_words(64) == 1
    assert lib.bnn_hip_act_words(65) == 2 and lib.bnn_hip_act_words(0) < 0
    for (O, C, k) in [(64, 64, 3), (128, 128, 3), (5, 200, 3), (128, 64, 1), (64, 1024, 1),
                      (8, 32, 5), (1000, 512, 1), (3, 3, 1), (512, 512, 3)]:
        L = native.weight_layout(O, C, k, k)
        ref = oracle.weight_layout(O, C, k, k)
        got = {f: getattr(L, f) for f in ("cw32", "cwc", "nchunk", "taps", "o_pad", "n_words")}
        assert got == ref
    bad = native.WLayout()
    assert lib.bnn_hip_weight_layout(0, 3, 1, 1, ctypes.byref(bad)) == -1
    assert lib.bnn_hip_weight_layout(3, 3, 1, 1, None) == -1


def test_ctypes_structs_have_the_layout_of_the_c_header(tmp_path):
    """The structs bnn_amd/native.py mirrors (bnn_hip_conv_desc, bnn_hip_epilogue, bnn_hip_wlayout, bnn_hip_fly_plan)
    against sizeof / offsetof of include/bnn_hip.h as gcc sees it: a field added on one side only (the epilogue grew
    by the folded-shortcut fields in ABI 11) would otherwise shift every pointer behind it silently."""
    import subprocess
    structs = {"bnn_hip_conv_desc": native.ConvDesc, "bnn_hip_epilogue": native.Epilogue,
               "bnn_hip_wlayout": native.WLayout, "bnn_hip_fly_plan": native.FlyPlan}
    lines = []
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    src = tmp_path / "layout.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "bnn_hip.h"\nint main(void) {\n%s\nreturn 0; }\n'
                   % "\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


def test_shortcut_fold_query_is_a_host_function():
    """bnn_hip_shortcut_fold_supported (ABI 11): 3x3 tiled kernels on non-negative activations, 64 / 128 / 256 shortcut
    channels — answered without a GPU."""
    lib = native.require()
    nn = native.FLAG_ACT_NONNEG
    d = native.ConvDesc(256, 128, 28, 28, 128, 3, 3, 1, 1, 1, 1, 1, 1, nn)
    assert lib.bnn_hip_shortcut_fold_supported(ctypes.byref(d), 64) == 1
    assert lib.bnn_hip_shortcut_fold_supported(ctypes.byref(d), 96) == 0          # not a supported channel count
    d.flags = 0
    assert lib.bnn_hip_shortcut_fold_supported(ctypes.byref(d), 64) == 0          # two-plane activations
    d.flags = nn | native.FLAG_WEIGHT_ZEROS
    assert lib.bnn_hip_shortcut_fold_supported(ctypes.byref(d), 64) == 0
    d1 = native.ConvDesc(256, 128, 28, 28, 128, 1, 1, 1, 1, 0, 0, 1, 1, nn)
    assert lib.bnn_hip_shortcut_fold_supported(ctypes.byref(d1), 64) == 0         # a 1x1 last conv (Bottleneck)
    d5 = native.ConvDesc(2, 512, 7, 7, 512, 3, 3, 1, 1, 1, 1, 1, 1, nn)
    assert lib.bnn_hip_shortcut_fold_supported(ctypes.byref(d5), 256) == 1        # multi-chunk layer
    assert lib.bnn_hip_shortcut_fold_supported(None, 64) == 0
    # the epilogue's shortcut fields are all-or-nothing, and exclusive with `residual`
    e = native.Epilogue()
    e.alpha = 16; e.out_f32 = 16; e.sc_wbits = 16
    assert lib.bnn_hip_bconv2d_fused(ctypes.byref(d), 16, 16, 16, 16, ctypes.byref(e), None) == -1


def test_argument_validation_without_touching_the_gpu():
    lib = native.require()
    d = native.ConvDesc(1, 64, 8, 8, 32, 3, 3, 1, 1, 1, 1, 1, 1, 0)
    # null pointers are rejected before any launch
    assert lib.bnn_hip_bconv2d(ctypes.byref(d), None, None, None, None, None, None, None, None, None) == -1
    assert lib.bnn_hip_bconv2d_fused(ctypes.byref(d), 16, 16, 16, 16, None, None) == -1
    assert lib.bnn_hip_pack_act_f32(None, 1, 1, 1, 1, None, None, None) == -1
    assert lib.bnn_hip_avgpool_pack_f32(None, 1, 1, 1, 1, 2, None, None, None) == -1
    assert lib.bnn_hip_pack_weight_f32(None, 1, 1, 1, 1, 0, 1, None, None, None, None, None) == -1
    assert lib.bnn_hip_pack_act_f16(None, 1, 1, 1, 1, None, None, None) == -1
    assert lib.bnn_hip_orpool_packed(None, 1, 64, 8, 8, 2, None, None, None) == -1
    assert lib.bnn_hip_orpool_packed(16, 1, 64, 8, 8, 0, 16, 16, None) == -1
    assert lib.bnn_hip_avgpool_fc_f32(None, 1, 1, 1, None, None, 1, None, None) == -1
    assert lib.bnn_hip_avgpool_fc_f32(16, 0, 512, 49, 16, None, 1000, 16, None) == -1
    # gradient kernels: host-side geometry helpers and argument checks
    assert lib.bnn_hip_grad_weight_pack_bytes(64, 64, 3) == 2 * 9 * 4 * 64 * 16
    assert lib.bnn_hip_grad_weight_pack_bytes(64, 64, 1) == 2 * 4 * 64 * 16
    assert lib.bnn_hip_grad_weight_pack_bytes(0, 64, 3) == 0 and lib.bnn_hip_grad_weight_pack_bytes(64, 64, 5) == 0
    assert 1 <= lib.bnn_hip_bconv_grad_weight_splits(256, 64, 64, 3) <= 256
    assert lib.bnn_hip_bconv_grad_weight_splits(1, 512, 512, 3) == 1
    assert 1 <= lib.bnn_hip_bconv_grad_weight_splits(256, 128, 64, 1) <= 256
    assert lib.bnn_hip_grad_pack_weight_f32(None, 64, 64, 3, None, None, None) == -1
    assert lib.bnn_hip_grad_pack_weight_f32(16, 64, 64, 5, 16, 16, None) == -2          # 5x5: unsupported
    assert lib.bnn_hip_bconv_grad_input_f32(16, 16, 16, 16, 16, 2, 64, 64, 8, 65, 3, 1, None) == -2  # width > 64
    assert lib.bnn_hip_bconv_grad_input_f32(16, 16, 16, 16, 16, 2, 64, 64, 8, 8, 3, 3, None) == -2   # stride 3
    assert lib.bnn_hip_bconv_grad_input_f32(16, 16, 16, 16, 16, 2, 64, 64, 8, 8, 1, 2, None) == -2   # strided 1x1
    assert lib.bnn_hip_bconv_grad_weight_f32(None, 16, 16, 1, 2, 64, 64, 8, 8, 3, 1, None) == -1
    # stem convolution of a training step: forward argument checks; backward workspace arithmetic and support rule
    assert lib.bnn_hip_stem7x7_conv_f32(None, 16, 1, 8, 8, 0, 16, None) == -1
    assert lib.bnn_hip_stem7x7_conv_f32(16, 16, 1, 8, 8, 2, 16, None) == -1                 # unknown flag
    ws = lib.bnn_hip_stem7x7_wgrad_workspace_bytes(256, 224, 224)
    assert ws > 0 and ws % (64 * 176 * 4) == 0 and ws // (64 * 176 * 4) <= 256 * 28      # one slab per workgroup, <= bands
    assert lib.bnn_hip_stem7x7_wgrad_workspace_bytes(1, 7, 9) == 64 * 176 * 4                # one band
    assert lib.bnn_hip_stem7x7_wgrad_workspace_bytes(1, 8, 4096) == 0                         # rows beyond the LDS patch
    assert lib.bnn_hip_stem7x7_wgrad_workspace_bytes(0, 224, 224) == 0
    assert lib.bnn_hip_stem7x7_wgrad_f32(None, 16, 1, 8, 8, 16, 1 << 20, 16, None) == -1
    assert lib.bnn_hip_stem7x7_wgrad_f32(16, 16, 1, 8, 4096, 16, 1 << 20, 16, None) == -2    # unsupported width
    assert lib.bnn_hip_stem7x7_wgrad_f32(16, 16, 1, 8, 8, 16, 16, 16, None) == -1             # workspace too small
    assert lib.bnn_hip_xnor_grad_pack_weight_f32(None, 64, 64, 3, 0, 1, 16, 16, None) == -1
    assert lib.bnn_hip_xnor_grad_pack_weight_f32(16, 64, 64, 5, 0, 1, 16, 16, None) == -2      # 5x5: unsupported
    assert lib.bnn_hip_xnor_grad_pack_weight_f32(16, 64, 64, 3, 0, 1, 8, 16, None) == -1       # packed: 16-byte aligned
    assert lib.bnn_hip_avgpool2x2_backward_f32(None, 1, 1, 1, 1, 16, None) == -1
    assert lib.bnn_hip_avgpool2x2_backward_f32(16, 1, 1, 0, 1, 16, None) == -1
    assert lib.bnn_hip_avgpool2x2_backward_f32(16, 1 << 15, 1 << 10, 1 << 5, 1 << 5, 16, None) == -4
    assert lib.bnn_hip_conv_workspace_bytes(ctypes.byref(d)) == 0       # the layer is one launch: no workspace
    assert lib.bnn_hip_bconv2d_direct(ctypes.byref(d), None, 0, None, None, None, None, None, None, None, None) == -1
    assert lib.bnn_hip_bconv2d_direct(ctypes.byref(d), 16, 7, 16, 16, 16, None, None, 16, None, None) == -1   # dtype
    assert lib.bnn_hip_bconv2d_direct_plan(ctypes.byref(d), None) == -1
    d.N = 0
    assert lib.bnn_hip_bconv2d(ctypes.byref(d), 16, 16, 16, 16, 16, None, None, 16, None) == -1


def test_direct_layer_plans_fit_the_lds_and_cover_the_output():
    """Host-side planner of the one-launch layer (bnn_hip_bconv2d_direct_plan): the band of a workgroup fits a CU's
    160 KiB of LDS, the bands tile the output, and only geometries whose single output row cannot fit are refused."""
    lib = native.require()

    def plan(N, C, H, W, O, k, s=1, p=0, dil=1):
        d = native.ConvDesc(N, C, H, W, O, k, k, s, s, p, p, dil, dil, 0)
        pl = native.FlyPlan()
        st = lib.bnn_hip_bconv2d_direct_plan(ctypes.byref(d), ctypes.byref(pl))
        return st, pl, d

    # BASELINE config 2: one whole image per workgroup (58 x 58 cells x (32 B of planes + a 4 B counter) = 121 KB),
    # 16 waves, 256 bands
    st, pl, _ = plan(256, 128, 56, 56, 128, 3, 1, 1)
    assert st == 0 and (pl.images_per_band, pl.rows_per_band, pl.waves, pl.n_bands) == (1, 56, 16, 256)
    assert 58 * 58 * 36 <= pl.lds_bytes <= 58 * 58 * 36 + 1024 and pl.blocks_per_unit == 2
    shapes = [(256, 64, 56, 56, 64, 3, 1, 1), (256, 64, 56, 56, 128, 3, 2, 1), (256, 128, 28, 28, 128, 3, 1, 1),
              (256, 256, 14, 14, 256, 3, 1, 1), (256, 512, 7, 7, 512, 3, 1, 1), (256, 256, 7, 7, 512, 1, 1, 0),
              (1, 3, 11, 13, 16, 3, 1, 1), (2, 128, 40, 300, 64, 3, 1, 1), (1, 64, 12, 12, 16, 3, 1, 2, 2),
              (4, 512, 224, 224, 64, 3, 1, 1), (7, 1024, 4, 4, 64, 1, 1, 0), (3, 32, 9, 9, 8, 5, 1, 2)]
    for sh in shapes:
        st, pl, d = plan(*sh)
        assert st == 0, sh
        assert 0 < pl.lds_bytes <= 160 * 1024 and 1 <= pl.waves <= 16 and pl.blocks_per_unit in (1, 2, 4)
        assert pl.pack_ahead == pl.fine_head == pl.fine_tail == pl.producers == -1    # the planner's defaults
        ho = (d.H + 2 * d.pad_h - d.dil_h * (d.KH - 1) - 1) // d.stride_h + 1
        assert 1 <= pl.rows_per_band <= ho and (pl.images_per_band == 1 or pl.rows_per_band == ho)
        assert pl.n_bands == -(-d.N // pl.images_per_band) * -(-ho // pl.rows_per_band)
        assert lib.bnn_hip_conv_workspace_bytes(ctypes.byref(d)) == 0
    # several small images share a band; a 2000-pixel-wide 512-channel row does not fit at all
    st, pl, _ = plan(4096, 512, 7, 7, 512, 3, 1, 1)
    assert st == 0 and pl.images_per_band > 1
    st, pl, d = plan(1, 512, 3, 2000, 32, 3, 1, 1)
    assert st == -2 and lib.bnn_hip_conv_workspace_bytes(ctypes.byref(d)) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import importlib
    monkeypatch.setenv("BNN_AMD_LIB", str(tmp_path / "nope.so"))
    fresh = importlib.reload(native)
    try:
        assert not fresh.available()
        with pytest.raises(fresh.NativeError, match="does not fall back"):
            fresh.require()
    finally:
        monkeypatch.delenv("BNN_AMD_LIB")
        importlib.reload(native)


def test_torch_custom_ops_are_registered_with_shape_inference_and_no_cpu_kernel():
    """torch.ops.bnn_amd.* (SURVEY §8(b)): schema + meta kernels exist everywhere, compute only on HIP."""
    import pytest
    import torch

    import bnn_amd  # noqa: F401
    from bnn_amd import torch_ops
    for name in torch_ops.OPS:
        assert hasattr(torch.ops.bnn_amd, name)
    x = torch.empty(2, 70, 9, 7, device="meta")
    w = torch.empty(40, 70, 3, 3, device="meta")
    y = torch.ops.bnn_amd.binary_conv2d(x, w, None, None, [2, 2], [1, 1], [1, 1], False, True)
    assert y.shape == (2, 40, 5, 4) and y.dtype == torch.float32 and y.device.type == "meta"
    P, M = torch.ops.bnn_amd.pack_sign(x)
    assert P.shape == (2, 2, 9, 7) and P.dtype == torch.int64 and M.shape == P.shape
    z = torch.ops.bnn_amd.binary_linear(torch.empty(3, 5, 70, device="meta"), torch.empty(11, 70, device="meta"),
                                        None, None, False, True)
    assert z.shape == (3, 5, 11)
    with pytest.raises((NotImplementedError, RuntimeError)):   # product path: no CPU stand-in
        torch.ops.bnn_amd.binary_conv2d(torch.zeros(1, 8, 4, 4), torch.zeros(4, 8, 3, 3), None, None,
                                        [1, 1], [1, 1], [1, 1], False, True)


def test_batch_step_respects_both_launch_limits():
    """hipops splits a batch so that one launch stays inside the kernels' addressing (2^30 elements per tensor) AND
    inside the tiled kernels' 24-bit index factors (images x channels < 2^23: beyond that the C side would fall back
    to the slow shape-generic kernel)."""
    from bnn_amd import hipops
    assert hipops._batch_step(256, 512 * 7 * 7, 512) == 256                      # ResNet-18 layer4: one launch
    assert hipops._batch_step(8192, 2048 * 7 * 7, 2048) == 4095                  # 8192 x 2048 channels: index factor
    assert 4095 * 2048 < (1 << 23) <= 4096 * 2048
    assert hipops._batch_step(4096, 64 * 112 * 112, 64) == ((1 << 30) - 1) // (64 * 112 * 112)   # addressing
    assert hipops._batch_step(5, 1 << 40, 1) == 1 and hipops._batch_step(3, 0, 0) == 3           # degenerate inputs


def test_integration_doc_names_every_exported_entry_point():
    """INTEGRATION.md's binding table covers the whole C-ABI (a few rows use the `_backward_f32` / `_workspace_bytes`
    shorthand next to their forward entry point)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = [s for s in native.EXPORTED_SYMBOLS
               if s not in text and s.replace("_backward_f32", "") not in text and s.replace("_workspace_bytes", "") not in text]
    assert not missing, missing
