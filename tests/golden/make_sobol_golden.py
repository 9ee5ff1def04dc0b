#!/usr/bin/env python3
"""Generates tests/golden/sobol_oracle.npz -- inputs and EXPECTED OUTPUTS of the Sobol sampler branch (src/sampler.h:9-36).

The reference's table file (sobol_10k_200.bin) is not in its repository, so nothing here is reference-derived: the table is the
build's own (restir_amd/sobol.py: scipy's Joe-Kuo direction numbers), pinned by corner samples and its digest; the sampler
streams come from an independent numpy restatement of Sampler::sample written next to the source text; the frames are the oracle's
own output in Sobol mode (a regression pin, like frames_oracle.npz).      python tests/golden/make_sobol_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import binding as ob  # noqa: E402
from restir_amd import sobol  # noqa: E402
from tests.common import OracleRenderer, get_scene  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def utilhash_np(a):
    """Math::utilhash (src/mathUtil.h:190-198) on uint32 arrays."""
    a = a.astype(np.uint64)
    M = np.uint64(0xffffffff)
    a = ((a + np.uint64(0x7ed55d16)) + (a << np.uint64(12))) & M
    a = ((a ^ np.uint64(0xc761c23c)) ^ (a >> np.uint64(19))) & M
    a = ((a + np.uint64(0x165667b1)) + (a << np.uint64(5))) & M
    a = ((a + np.uint64(0xd3a2646c)) ^ (a << np.uint64(9))) & M
    a = ((a + np.uint64(0xfd7046c5)) + (a << np.uint64(3))) & M
    a = ((a ^ np.uint64(0xb55a4f09)) ^ (a >> np.uint64(16))) & M
    return a.astype(np.uint32)


def sobol_stream_np(table, looper, index, dim, m):
    """m draws of Sampler(looper * 200 + dim, utilhash(index), data) (src/sampler.h:19-32)."""
    flat = table.reshape(-1)
    ptr = looper.astype(np.int64) * 200 + dim
    scramble = utilhash_np(index.astype(np.uint32))
    out = np.zeros((len(looper), m), np.float32)
    for k in range(m):
        r = flat[ptr + k] ^ scramble
        scramble = utilhash_np(scramble)
        out[:, k] = r.astype(np.float32) * np.float32(2.0 ** -32)      # uint32 -> float rounds to nearest even, the scaling is exact
    return out


def main():
    t = sobol.sobol_table()
    assert sobol.table_digest(t) == sobol.TABLE_SHA256, sobol.table_digest(t)
    g = {"table_head": t[:16, :16].copy(), "table_tail": t[-4:, -8:].copy(), "table_col_sums": t.astype(np.uint64).sum(0)}
    rng = np.random.default_rng(7)
    n = 256
    looper = rng.integers(0, 9999, n).astype(np.int32); looper[:4] = [0, 1, 9998, 42]
    index = rng.integers(0, 1920 * 1080, n).astype(np.int32); index[:4] = [0, 1, 2073599, 12345]
    dim = rng.integers(0, 8, n).astype(np.int32); dim[:8] = 0
    g.update(stream_looper=looper, stream_index=index, stream_dim=dim, stream_out=sobol_stream_np(t, looper, index, dim, 181))
    sd = get_scene("cornell")
    for reuse in (0, 1, 2, 3):
        o = OracleRenderer(sd, 64, 64, sobol=t)
        o.looper = 9998                                  # frames 9998, 9999, 0: the wrap of `(looper + 1) % SobolSampleNum`
        for _ in range(3):
            img = o.frame(reuse)
        assert o.looper == 1
        g[f"cornell64_reuse{reuse}_frame2"] = img.copy()
        g[f"cornell64_reuse{reuse}_M"] = o.restir.last["numSamples"].copy()
    o = OracleRenderer(sd, 64, 64, sobol=t)
    g["cornell64_ptdirect"] = o.frame(0, use_reservoir=False).copy()
    np.savez_compressed(os.path.join(OUT, "sobol_oracle.npz"), **g)
    print("wrote sobol_oracle.npz", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
