"""The PCG64 jump-ahead table the kernels use to re-derive an env's np_random stream at an arbitrary draw
(ns_gym_amd/csrc/nsg_rng.hip.h, pcg_at; built on the host by nsg_pcg64_jump_table) against plain big-integer arithmetic, and
the jump itself against NumPy's own generator: state after n draws == A_n * S_0 + inc * G_n (mod 2^128)."""
import ctypes as C

import numpy as np

M = 0x2360ED051FC65DA44385DF649FCCF645      # PCG_DEFAULT_MULTIPLIER_128 (numpy/random/src/pcg64/pcg64.h [UPSTREAM])
MASK = (1 << 128) - 1


def _table():
    from ns_gym_amd import _lib

    lib = _lib.load()
    out = (C.c_uint64 * (5 * 256 * 4))()
    lib.nsg_pcg64_jump_table(out)
    w = np.frombuffer(out, dtype=np.uint64).reshape(5, 256, 4)
    return [[((int(e[0]) << 64) | int(e[1]), (int(e[2]) << 64) | int(e[3])) for e in row] for row in w]


def _geom(n):      # 1 + M + ... + M^(n-1) mod 2^128, by doubling
    a, g, res_a, res_g = M, 1, 1, 0
    while n:
        if n & 1:
            res_g = (res_g * a + g) & MASK
            res_a = (res_a * a) & MASK
        g = (g * (a + 1)) & MASK
        a = (a * a) & MASK
        n >>= 1
    return res_a, res_g


def test_table_entries_are_powers_and_geometric_sums():
    t = _table()
    for d in range(5):
        for v in (0, 1, 2, 17, 128, 255):
            e = v * 256 ** d
            assert t[d][v] == (pow(M, e, 1 << 128), _geom(e)[1]), (d, v)
    assert t[0][0] == (1, 0) and t[3][0] == (1, 0)


def test_composed_jump_equals_numpys_generator():
    """Compose the table the way the kernel does (one entry per non-zero 8-bit digit of n) and compare the jumped state with
    NumPy's PCG64 advanced by n raw draws."""
    t = _table()
    rng = np.random.default_rng(12345)
    for seed in (0, 42, 2**40 + 7):
        for n in [0, 1, 2, 255, 256, 257, 4 * 31337, 65535, 65536, 10**7 + 3] + [int(x) for x in rng.integers(0, 2**39, size=6)]:
            bg = np.random.PCG64(np.random.SeedSequence(seed))
            st = bg.state["state"]
            s0, inc = st["state"], st["inc"]
            A, G, rest, d = 1, 0, n, 0
            while rest:
                v = rest & 255
                rest >>= 8
                if v:
                    Ad, Gd = t[d][v]
                    G = (G * Ad + Gd) & MASK
                    A = (A * Ad) & MASK
                d += 1
            jumped = (A * s0 + inc * G) & MASK
            want = bg.advance(n).state["state"]["state"] if n else s0
            assert jumped == want, (seed, n)
