"""Scan gfx950 assembly for a matrix-core pair the hardware was seen to get wrong (tools/experiments/README.md 54):

    v_mfma_f32_16x16x32_f16 vA, ..., ...            (8 passes)
    v_mfma_f32_16x16x16_f16 vB, ..., ..., vA        (4 passes; SrcC = the result above, vDst another quad)

issued in consecutive slots: hipcc 7.2 puts no wait states between them and two of the four result registers come out
wrong.  More generally: any MFMA immediately followed (only s_waitcnt between) by an MFMA of a DIFFERENT opcode that
reads its result as SrcC.  ``python tools/mfma_pairs.py file.s`` prints the pairs and exits 1 if there are any;
tests/test_isa_cpu.py compiles csrc/stem_rows.hip and calls :func:`dependent_pairs` on the result.
"""
import re
import sys

_REG = re.compile(r"^([av])(?:\[(\d+):(\d+)\]|(\d+))$")


def _reg(tok):
    """(file, lo, hi) of a VGPR / AccVGPR operand, None for anything else (constants, modifiers)."""
    m = _REG.match(tok.strip())
    if not m:
        return None
    if m.group(4) is not None:
        return m.group(1), int(m.group(4)), int(m.group(4))
    return m.group(1), int(m.group(2)), int(m.group(3))


def dependent_pairs(asm_text):
    """[(kernel, line number, first, second)] of back-to-back different-opcode MFMAs chained through SrcC."""
    out, kernel, prev = [], "", None
    for no, line in enumerate(asm_text.split("\n"), 1):
        if line.startswith("_Z") and ":" in line:
            kernel, prev = line.split(":")[0], None
            continue
        t = line.split(";")[0].strip()
        if not t or t.startswith(".") or t.startswith("s_waitcnt"):
            continue
        if not t.startswith("v_mfma"):
            prev = None
            continue
        op = t.split()[0]
        ops = [_reg(x) for x in t[len(op):].split(",")[:4]]
        dst = ops[0]
        srcc = ops[3] if len(ops) == 4 else None
        if (prev and prev[0] != op and dst is not None and srcc is not None and prev[1] is not None
                and srcc[0] == prev[1][0] and srcc[1] <= prev[1][2] and prev[1][1] <= srcc[2]):
            out.append((kernel, no, prev[2], t))
        prev = (op, dst, t)
    return out


if __name__ == "__main__":
    found = dependent_pairs(open(sys.argv[1]).read())
    for k, no, a, b in found:
        print(f"{k}:{no}\n    {a}\n    {b}")
    sys.exit(1 if found else 0)
