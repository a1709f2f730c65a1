"""The PCG64 jump-ahead table the kernels use to re-derive an env's np_random stream at an arbitrary draw
(ns_gym_amd/csrc/nsg_rng.hip.h, pcg_at; built on the host by nsg_pcg64_jump_table) against plain big-integer arithmetic, and
the jump itself against NumPy's own generator: state after n draws == A_n * S_0 + inc * G_n (mod 2^128); and the kernels'
actual form, which starts one step BEFORE the seeded state (T0 = inc + initstate, S_0 = step(T0)) and reads its digit-0 entry
at (n mod 256) + LEAD."""
import ctypes as C

import numpy as np

M = 0x2360ED051FC65DA44385DF649FCCF645      # PCG_DEFAULT_MULTIPLIER_128 (numpy/random/src/pcg64/pcg64.h [UPSTREAM])
MASK = (1 << 128) - 1
LOW = 264                                    # NSG_JUMP_LOW (include/nsgym_hip.h)


def _table():
    from ns_gym_amd import _lib

    lib = _lib.load()
    out = (C.c_uint64 * ((LOW + 4 * 256) * 4))()
    lib.nsg_pcg64_jump_table(out)
    w = np.frombuffer(out, dtype=np.uint64).reshape(-1, 4)
    e = [((int(x[0]) << 64) | int(x[1]), (int(x[2]) << 64) | int(x[3])) for x in w]
    # [0]: the digit-0 block (exponent = index, LOW entries); [d]: exponent v * 256^d
    return [e[:LOW]] + [e[LOW + (d - 1) * 256: LOW + d * 256] for d in range(1, 5)]


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
    assert len(t[0]) == LOW
    for e in (256, 257, LOW - 1):      # the digit-0 block runs past 255: exponent = index
        assert t[0][e] == (pow(M, e, 1 << 128), _geom(e)[1]), e
    assert t[0][256] == t[1][1]


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


def test_kernel_form_with_the_seeding_step_folded_in():
    """pcg_at<LEAD> (nsg_rng.hip.h): state = A * T0 + inc * G with the digit-0 entry taken at (n & 255) + LEAD and the further
    digits from n >> 8.  LEAD = 1 is the generator's state after n draws; LEAD = 2 is one step further - the state whose
    output IS draw n."""
    t = _table()
    rng = np.random.default_rng(7)
    for seed in (0, 3, 2**63 + 11):
        for n in [0, 1, 254, 255, 256, 510, 511, 65535, 4 * 99999 + 3] + [int(x) for x in rng.integers(0, 2**39, size=6)]:
            bg = np.random.PCG64(np.random.SeedSequence(seed))
            st = bg.state["state"]
            s0, inc = st["state"], st["inc"]
            t0 = ((s0 - inc) * pow(M, -1, 1 << 128)) & MASK       # S_0 = T0 * M + inc
            for lead in (1, 2):
                A, G = t[0][(n & 255) + lead]
                rest, d = n >> 8, 1
                while rest:
                    v = rest & 255
                    rest >>= 8
                    if v:
                        Ad, Gd = t[d][v]
                        G = (G * Ad + Gd) & MASK
                        A = (A * Ad) & MASK
                    d += 1
                got = (A * t0 + inc * G) & MASK
                b2 = np.random.PCG64(np.random.SeedSequence(seed))
                m = n + lead - 1
                want = b2.advance(m).state["state"]["state"] if m else s0
                assert got == want, (seed, n, lead)
