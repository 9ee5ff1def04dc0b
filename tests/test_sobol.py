"""The Sobol sampler branch (src/sampler.h:9-36, SAMPLER_USE_SOBOL; table upload src/scene.cpp:500-506; looper wrap
src/restir.cu:441-445) on the CPU: the build's own table, the oracle's sampler against an independent numpy restatement, and a
regression pin of the oracle's frames in that mode (tests/golden/sobol_oracle.npz, tests/golden/make_sobol_golden.py).

The reference's table file is absent from its repository: Sobol parity is pinned to the oracle over the build's table only."""
import os

import numpy as np
import pytest

from oracle import binding as ob
from restir_amd import sobol
from tests.common import OracleRenderer, bits_equal, get_scene

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sobol_oracle.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def table():
    return sobol.sobol_table()


def test_table_is_the_pinned_one(g, table):
    assert table.shape == (sobol.SOBOL_SAMPLE_NUM, sobol.SOBOL_SAMPLE_DIM) == (10000, 200) and table.dtype == np.uint32
    assert np.array_equal(table[:16, :16], g["table_head"]) and np.array_equal(table[-4:, -8:], g["table_tail"])
    assert np.array_equal(table.astype(np.uint64).sum(0), g["table_col_sums"])
    assert sobol.table_digest(table) == sobol.TABLE_SHA256
    # what a Sobol sequence is: point 0 the origin, point 1 the centre, every coordinate of the first 8192 points a permutation
    # of the multiples of 2^-13
    assert (table[0] == 0).all() and (table[1] == 1 << 31).all()
    first = np.sort(table[:8192] >> 19, axis=0)
    assert (first == np.arange(8192, dtype=np.uint32)[:, None]).all()


def test_table_file_round_trip(tmp_path, table):
    p = tmp_path / "sobol_10k_200.bin"
    sobol.write_table(p, table)
    assert os.path.getsize(p) == 10000 * 200 * 4                       # what DevScene::create reads (src/scene.cpp:502)
    assert np.array_equal(sobol.read_table(p), table)


def test_oracle_sampler_equals_numpy_restatement(g, table):
    """Sampler(looper * 200 + dim, utilhash(index), data): r = data[ptr++] ^ scramble; scramble = utilhash(scramble); r * 2^-32."""
    n, m = g["stream_out"].shape
    out = np.zeros((n, m), np.float32)
    flat = np.ascontiguousarray(table.reshape(-1))
    ob.lib().orc_sobol_stream(flat, n, g["stream_looper"], g["stream_index"], g["stream_dim"], m, out.reshape(-1))
    assert bits_equal(out, g["stream_out"])
    assert out.min() >= 0.0 and out.max() <= 1.0                       # 1.0f itself can come out (r >= 0xffffff80), as with the default engine
    # and by hand for one pixel: looper 0 reads the table's first row, the origin, so the draws are the scramble chain alone
    k = int(np.nonzero((g["stream_looper"] == 0) & (g["stream_dim"] == 0))[0][0])
    a = int(g["stream_index"][k])

    def utilhash(a):
        a = (a + 0x7ed55d16 + (a << 12)) & 0xffffffff
        a = (a ^ 0xc761c23c ^ (a >> 19)) & 0xffffffff
        a = (a + 0x165667b1 + (a << 5)) & 0xffffffff
        a = ((a + 0xd3a2646c) ^ (a << 9)) & 0xffffffff
        a = (a + 0xfd7046c5 + (a << 3)) & 0xffffffff
        a = (a ^ 0xb55a4f09 ^ (a >> 16)) & 0xffffffff
        return a
    s = utilhash(a)
    for j in range(4):
        assert out[k, j] == np.float32(s) * np.float32(2.0 ** -32)
        s = utilhash(s)


def test_frames_regression_pin_with_looper_wrap(g, table):
    """Oracle frames in Sobol mode, loopers 9998, 9999, 0 -- `State::looper = (State::looper + 1) % SobolSampleNum`."""
    sd = get_scene("cornell")
    for reuse in (0, 1, 2, 3):
        o = OracleRenderer(sd, 64, 64, sobol=table)
        o.looper = 9998
        for _ in range(3):
            img = o.frame(reuse)
        assert o.looper == 1
        assert bits_equal(img, g[f"cornell64_reuse{reuse}_frame2"]), reuse
        assert np.array_equal(o.restir.last["numSamples"], g[f"cornell64_reuse{reuse}_M"])
    o = OracleRenderer(sd, 64, 64, sobol=table)
    assert bits_equal(o.frame(0, use_reservoir=False), g["cornell64_ptdirect"])


def test_sobol_frames_estimate_the_same_image(table):
    """Both samplers estimate the same integrals: averaged over frames the two modes' images agree to Monte-Carlo accuracy, and
    switching the table off gives the default engine's bits again."""
    sd = get_scene("cornell")
    W = H = 48
    acc = []
    for tab in (None, table):
        o = OracleRenderer(sd, W, H, sobol=tab)
        s = np.zeros((W * H, 3), np.float64)
        for _ in range(24):
            s += o.frame(0)
        acc.append(s / 24)
    lit = acc[0].sum(1) > 0.05
    assert lit.sum() > 500
    assert abs(acc[0][lit].mean() / acc[1][lit].mean() - 1.0) < 0.03
    assert not np.array_equal(acc[0], acc[1])
    o = OracleRenderer(sd, W, H, sobol=table)
    a = o.frame(3).copy()
    o2 = OracleRenderer(sd, W, H, scene=o.scene, sobol=None)
    o2.scene.set_sample_sequence(None)
    b = o2.frame(3)
    ref = OracleRenderer(sd, W, H).frame(3)
    assert bits_equal(b, ref) and not bits_equal(a, ref)
