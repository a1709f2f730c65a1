"""Pins the oracle's NumPy-compatible bit streams against
(1) fixtures produced by the NumPy of the build container (tests/golden/numpy_streams.npz,
manifest records the version) and (2) the NumPy importable at test time."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import load


@pytest.fixture(scope="module")
def g():
    return load("numpy_streams.npz")


def test_seedsequence_pcg64_state(g):
    seeds = g["seeds"]
    _, st = O.rng_fill(0, seeds, 0)
    np.testing.assert_array_equal(st.T, g["pcg_state"])
    for j in range(3):
        _, st = O.rng_fill(0, seeds, 0, spawn_key=j)
        np.testing.assert_array_equal(st.T, g["child_state"][:, j])


def test_survey_known_answers():
    # PCG64(SeedSequence(42)) state / inc / first outputs
    out, st = O.rng_fill(0, [42], 2)
    assert (int(st[0, 0]) << 64 | int(st[1, 0])) == 0xCEA44F6798798F2AACBC7C9D68860AC8
    assert (int(st[2, 0]) << 64 | int(st[3, 0])) == 0xFA505436C9A8416E66CAF2E28D25ABFF
    assert [int(x) for x in out[:, 0]] == [0xC621FBCD16D92688, 0x705A5661A791FFC1]
    rnd, _ = O.rng_fill(1, [42], 3)
    assert list(rnd[:, 0]) == [0.7739560485559633, 0.4388784397520523, 0.8585979199113825]
    n1, _ = O.rng_fill(2, [42], 3, spawn_key=1)
    np.testing.assert_allclose(n1[:, 0], [1.2544943667397455, 0.6062894401319061, -1.3401775973994274], rtol=1e-15)


def test_raw_random_normal(g):
    seeds = g["seeds"]
    raw, _ = O.rng_fill(0, seeds, 8)
    np.testing.assert_array_equal(raw.T, g["raw"])
    rnd, _ = O.rng_fill(1, seeds, 8)
    np.testing.assert_array_equal(rnd.T, g["random"])
    nrm, _ = O.rng_fill(2, seeds, 2000)
    np.testing.assert_array_equal(nrm.T, g["normal"])


def test_child_normals(g):
    seeds = g["seeds"]
    for j in range(3):
        z, _ = O.rng_fill(2, seeds, 16, spawn_key=j)
        np.testing.assert_allclose((0.3 + 2.0 * z).T, g["child_normal"][:, j], rtol=1e-15)


def test_normal_long_run_hits_wedge_and_tail(g):
    z, _ = O.rng_fill(2, [2024], 2_000_000)
    z = z[:, 0]
    idx = g["normal_long_tail_idx"]
    assert idx.size > 10  # the tail branch (|z| > r) is exercised
    np.testing.assert_allclose(z[idx], g["normal_long_tail_val"], rtol=1e-14)
    s = g["normal_long_sum"]
    assert abs(z.sum() - s[0]) < 1e-6 and np.abs(z).max() == pytest.approx(s[1], rel=1e-14) and z[-1] == s[2]


def test_against_live_numpy():
    rng = np.random.default_rng(5)
    seeds = np.concatenate([rng.integers(0, 2**63, size=64, dtype=np.uint64), np.arange(64, dtype=np.uint64)])
    z, st = O.rng_fill(2, seeds, 64)
    u, _ = O.rng_fill(1, seeds, 8)
    for i, s in enumerate(seeds[:128:7]):
        i = i * 7
        gen = np.random.default_rng(int(s))
        np.testing.assert_array_equal(z[:, i], gen.standard_normal(64))
        np.testing.assert_array_equal(u[:, i], np.random.default_rng(int(s)).random(8))
        for j in (0, 2):
            c = np.random.SeedSequence(int(s)).spawn(3)[j]
            zz, _ = O.rng_fill(2, [s], 8, spawn_key=j)
            np.testing.assert_array_equal(zz[:, 0], np.random.default_rng(c).standard_normal(8))


def test_categorical_seed0(g):
    u, _ = O.rng_fill(1, [0], 64)
    cs = np.cumsum(np.array([0.6, 0.2, 0.2]))
    idx = [int(np.argmax(cs > r)) for r in u[:, 0]]
    np.testing.assert_array_equal(idx, g["categorical_seed0"])
    assert idx[:10] == [1, 0, 0, 0, 2, 2, 1, 1, 0, 2]


def test_exponential_geometric_dirichlet(g):
    """Streams behind MemorylessScheduler (rng.geometric, schedulers.py:108,112) and RandomCategorical
    (rng.dirichlet, distribution.py:38): bit-exact with NumPy incl. the ziggurat wedge/tail paths."""
    import ctypes as C

    L = O.lib()
    for j, s in enumerate((0, 7, 99)):
        out = np.zeros(400)
        L.orc_exponential(C.c_uint64(s), 400, out.ctypes.data_as(C.c_void_p))
        np.testing.assert_array_equal(out, g["exponential"][j])
    out = np.zeros(1_000_000)
    L.orc_exponential(C.c_uint64(2025), out.size, out.ctypes.data_as(C.c_void_p))
    assert out.sum() == g["exponential_long"][0] and out.max() == g["exponential_long"][1] and out[-1] == g["exponential_long"][2]
    for name, p in (("geometric_p5", 0.5), ("geometric_p1", 0.1), ("geometric_p001", 0.001)):
        for j, s in enumerate((0, 7)):
            o = np.zeros(200, dtype=np.int64)
            L.orc_geometric(C.c_uint64(s), C.c_double(p), 200, o.ctypes.data_as(C.c_void_p))
            np.testing.assert_array_equal(o, g[name][j])
    for n, seed, key, sl in ((3, 12, "dirichlet3", slice(1, None)), (4, 13, "dirichlet4", slice(None))):
        o = np.zeros((50, n))
        L.orc_dirichlet_ones(C.c_uint64(seed), n, 50, o.ctypes.data_as(C.c_void_p))
        np.testing.assert_array_equal(o, g[key][sl])
