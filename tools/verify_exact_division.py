#!/usr/bin/env python3
"""tools/verify_exact_division.py [first_chunk [chunks]] -- rs_exact.h's quotient (correctly rounded reciprocal, product, exact
residual, one correction) against the compiler's correctly rounded x / d on EVERY pair of significands: 2^23 denominators x 2^23
numerators = 7.04e13 pairs at exponent 0.  Every operation of the short form commutes with scaling by powers of two as long as no
intermediate leaves the normal range, which the guard [2^-60, 2^60) ensures, so this is the statement for the whole guarded range
(tests/test_gpu_exact_ops.py adds pairs at the edges of the range).  512 launches of 16 384 denominators each; prints progress."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from restir_amd import capi

capi.init(0)
CH = 1 << 14
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 23) // CH - first
out = (C.c_ulonglong * 3)()
bad = n = excluded = 0
t0 = time.time(); last = t0
for c in range(first, first + chunks):
    capi.check(capi.lib().rs_debug_exact_ops_mismatches(2, c * CH, CH, 0, 0, out))
    bad += out[0]; n += out[1]; excluded += out[2]
    if time.time() - last > 20 or c == first + chunks - 1:
        last = time.time()
        print("denominators [%d, %d): %.4g pairs compared, %d differ, %.0f s" % (first * CH, (c + 1) * CH, n, bad, time.time() - t0), flush=True)
print("RESULT: %d pairs, %d differ" % (n, bad))
sys.exit(1 if bad else 0)
