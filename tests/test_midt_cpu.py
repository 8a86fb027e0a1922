"""Host-side check of the two-instruction sign test of the conv1-type epilogue (csrc/bconv_core.h: midt2_shift_in).

The kernel replaces  bit = (+-2*cnt -+ nz >= T)  (v_lshl_add + v_cmp + v_addc per channel and pixel) with
``v_subbrev_co_u32`` + ``v_addc_co_u32`` on a popcount chain that starts from ``bias - nz // 2``.  This file restates
the instruction pair in Python integers (32-bit unsigned borrow, the scalar-unit arithmetic that derives the
comparand and the carry mask from T) and enumerates every case a ResNet layer can produce against the plain integer
test; the device-side check is tests/test_gpu_fused.py::test_integer_sign_thresholds_give_the_float_epilogue_bits.
"""
import itertools

import pytest

BIAS = 1 << 20
M32 = (1 << 32) - 1


def _s32(v):
    v &= M32
    return v - (1 << 32) if v & (1 << 31) else v


def _borrow(cnt_biased, a, carry):
    """vcc out of ``v_subbrev_co_u32 tmp, vcc, a, cnt, vcc``: cnt - a - carry borrows (operands as 32-bit unsigned)."""
    return int((cnt_biased & M32) < (a & M32) + carry)


def bit_two_instr(nn, cnt, nz, T):
    """The bit the kernel stores for one lane and channel (after the flip-word complement of the agreement form)."""
    q, p = nz >> 1, nz & 1
    chain = BIAS - q + cnt                       # the popcount chain's value: seeded with bias - q
    if nn:                                       # agreements: 2 cnt - nz >= T
        carry = p if (T & 1) == 0 else 0         # s_bitcmp0 + s_cselect
        a = _s32(_s32(T + 1 + 2 * BIAS) >> 1)    # s_add_i32 + s_ashr_i32
        return 1 - _borrow(chain, a, carry)      # complement: joins the flip word
    carry = p if (T & 1) == 1 else 0             # disagreements: nz - 2 cnt >= T
    a = max(_s32(_s32(2 + 2 * BIAS - T) >> 1), 0)  # s_sub_i32 + s_ashr_i32 + s_max_i32
    return _borrow(chain, a, carry)


def bit_reference(nn, cnt, nz, T):
    return int((2 * cnt - nz if nn else nz - 2 * cnt) >= T)


@pytest.mark.parametrize("nn", [True, False], ids=["agreements", "disagreements"])
def test_two_instruction_threshold_equals_the_integer_test_by_enumeration(nn):
    checked = 0
    for nz in range(0, 70):
        for cnt, T in itertools.product(range(0, nz + 1), range(-82, 83)):
            assert bit_two_instr(nn, cnt, nz, T) == bit_reference(nn, cnt, nz, T), (nn, cnt, nz, T)
            checked += 1
    assert checked > 400_000


@pytest.mark.parametrize("nn", [True, False], ids=["agreements", "disagreements"])
@pytest.mark.parametrize("K", [576, 1152, 2304, 4608, (1 << 19) - 1])
def test_two_instruction_threshold_at_layer_sizes_and_sentinels(nn, K):
    """Thresholds the device derives: any T in [-K, K], "always" (T = -K) and "never" (T = 2^30); counts at the ends
    and around the threshold."""
    for T in (-K, -K + 1, -3, -2, -1, 0, 1, 2, 3, K - 1, K, 1 << 30):
        for nz in (0, 1, 2, 3, K // 2, K // 2 + 1, K - 1, K):
            around = {0, 1, nz // 2, nz}
            for d in range(-3, 4):
                around.add(min(max((nz + T) // 2 + d, 0), nz))
                around.add(min(max((nz - T) // 2 + d, 0), nz))
            for cnt in around:
                assert bit_two_instr(nn, cnt, nz, T) == bit_reference(nn, cnt, nz, T), (nn, cnt, nz, T, K)
