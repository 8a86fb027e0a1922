"""Compiler-output check that needs no GPU: the assembly of every source with matrix-core instructions holds no back-to-back pair of different matrix-core
opcodes chained through SrcC (tools/mfma_pairs.py; tools/experiments/README.md 54 — gfx950 returned two of four result
registers wrong for it and hipcc 7.2 schedules no wait states there)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mfma_pairs  # noqa: E402

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def test_scanner_sees_the_pair_and_nothing_else():
    bad = """_ZN3bnn1kEv:
	v_mfma_f32_16x16x32_f16 v[98:101], v[34:37], v[94:97], v[98:101]
	s_waitcnt lgkmcnt(2)
	v_mfma_f32_16x16x16_f16 v[94:97], v[158:159], v[106:107], v[98:101]
"""
    assert len(mfma_pairs.dependent_pairs(bad)) == 1
    same_opcode = bad.replace("16x16x16_f16 v[94:97], v[158:159], v[106:107]", "16x16x32_f16 v[94:97], v[158:161], v[106:109]")
    assert mfma_pairs.dependent_pairs(same_opcode) == []
    apart = bad.replace("\ts_waitcnt lgkmcnt(2)\n", "\ts_nop 15\n")
    assert mfma_pairs.dependent_pairs(apart) == []
    independent = bad.replace("v[106:107], v[98:101]", "v[106:107], v[82:85]")
    assert mfma_pairs.dependent_pairs(independent) == []
    accvgpr = bad.replace("v[98:101]", "a[8:11]")       # accumulators in AccVGPRs: the same pair
    assert len(mfma_pairs.dependent_pairs(accvgpr)) == 1


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src", ["stem_rows.hip", "stem.hip", "stem_wgrad.hip", "grad.hip"])
def test_mfma_kernels_hold_no_back_to_back_mixed_mfma_chain(tmp_path, src):
    out = tmp_path / "k.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                    os.path.join(ROOT, "binary-networks-pytorch_amd", "csrc", src), "-o", str(out)],
                   check=True, capture_output=True, timeout=600)
    text = out.read_text()
    assert "v_mfma" in text
    assert mfma_pairs.dependent_pairs(text) == []


def test_no_source_uses_the_wide_buffer_load_builtins_hipcc_7_2_miscompiles():
    """`__builtin_amdgcn_raw_buffer_load_b64 / _b96 / _b128`: hipcc 7.2 narrows them to one dword when the elements of the
    result are extracted (tools/experiments/README.md 55) — `csrc/stem_rows.hip: rows_ld2` shows the form that works."""
    import glob
    import re
    csrc = os.path.join(ROOT, "binary-networks-pytorch_amd", "csrc")
    bad = []
    for path in glob.glob(os.path.join(csrc, "**", "*.h*"), recursive=True):
        for no, line in enumerate(open(path), 1):
            code = line.split("//")[0]
            if re.search(r"__builtin_amdgcn_raw_buffer_load_b(64|96|128)\b", code):
                bad.append(f"{os.path.relpath(path, ROOT)}:{no}")
    assert not bad, bad
