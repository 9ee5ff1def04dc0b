"""The short forms of 1 / d, x / d and sqrt(x) the kernels use inside a guarded operand range (restir_amd/csrc/rs_exact.h) against the
compiler's correctly rounded operators -- bit for bit, on every operand for the two unary forms and on 2^13 denominators x all 2^23
numerators (7e10 pairs) plus pairs at the edges of the exponent range for the quotient.  tools/verify_exact_division.py is the same
check over ALL 2^23 denominators (every pair of significands; its log is profiles/r03_exact_division_all_pairs.log)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(hip, op, first=0, count=1, ex=0, ed=0):
    out = (C.c_ulonglong * 3)()
    hip.check(hip.lib().rs_debug_exact_ops_mismatches(op, first, count, ex, ed, out))
    return int(out[0]), int(out[1]), int(out[2])


def test_reciprocal_of_every_float_in_range(hip):
    bad, n, _ = run(hip, 0)
    assert n == 2 * (0x5D800000 - 0x21800000) and bad == 0    # both signs; including the all-ones significands, the refinement's textbook exception


def test_square_root_of_every_float_in_range(hip):
    bad, n, _ = run(hip, 1)
    assert n == 0x5D800000 - 0x21800000 and bad == 0


@pytest.mark.parametrize("ex,ed", [(0, 0), (-60, 59), (59, -60), (-60, -60), (59, 59), (17, -3)])
def test_quotient_over_all_numerators(hip, ex, ed):
    """Denominator significands: the first and last 512, 512 around one half, and 6656 spread over the rest by a stride coprime to 2^23."""
    total_bad, total_n = 0, 0
    blocks = [(0, 512), ((1 << 23) - 512, 512), ((1 << 22) - 256, 512)]
    rng = np.random.default_rng((ex + 60) * 131 + ed + 60)
    blocks += [(int(s), 256) for s in rng.integers(1024, (1 << 23) - 2048, 26 if (ex, ed) == (0, 0) else 4)]
    for first, count in blocks:
        bad, n, _ = run(hip, 2, first, count, ex, ed)
        total_bad += bad; total_n += n
        assert bad == 0, (first, count, bad)
    assert total_n >= 2 ** 23 * 2000


@pytest.mark.parametrize("op", [3, 4, 5], ids=["-x / d", "x / -d", "-x / -d"])
def test_quotient_of_negative_operands(hip, op):
    for first in (0, (1 << 23) - 256, 1234567, 4194304):
        bad, n, _ = run(hip, op, first, 256, 3, -5)
        assert bad == 0 and n == 256 << 23


def test_the_guarded_functions_fall_back_outside_the_range(hip):
    """rs_debug_exact_ops_mismatches refuses exponents outside the guarded range: the kernels take the compiler's operator there."""
    out = (C.c_ulonglong * 3)()
    assert hip.lib().rs_debug_exact_ops_mismatches(2, 0, 16, 60, 0, out) != 0
    assert hip.lib().rs_debug_exact_ops_mismatches(2, 0, 16, 0, -61, out) != 0
