"""Accuracy of the kernels' own float64 sincos / exp / log1p (ns_gym_amd/csrc/nsg_math.hip.h).
The header is plain C++ under NSG_HD, so the very same source is compiled for the host here and
compared with long-double libm: <= 1 ulp everywhere the integrators / θ-engine evaluate it."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "%s/ns_gym_amd/csrc/nsg_math.hip.h"
extern "C" {
void t_sincos(const double* x, double* s, double* c, long n) { for (long i = 0; i < n; i++) nsg::nsg_sincos(x[i], s + i, c + i); }
void t_sincos_poly0(const double* x, double* s, double* c, long n) { for (long i = 0; i < n; i++) nsg::nsg_sincos_t<0>(x[i], s + i, c + i); }
void t_sincos_poly1(const double* x, double* s, double* c, long n) { for (long i = 0; i < n; i++) nsg::nsg_sincos_t<1>(x[i], s + i, c + i); }
void t_sincos_poly2(const double* x, double* s, double* c, long n) { for (long i = 0; i < n; i++) nsg::nsg_sincos_t<2>(x[i], s + i, c + i); }
void t_pymod(const double* x, double m, double* y, long n) { for (long i = 0; i < n; i++) y[i] = nsg::nsg_pymod_pos(x[i], m); }
void t_exp(const double* x, double* y, long n) { for (long i = 0; i < n; i++) y[i] = nsg::nsg_exp(x[i]); }
void t_exp_libm(const double* x, double* y, long n) { for (long i = 0; i < n; i++) y[i] = nsg::nsg_exp_libm(x[i]); }
void t_log1p(const double* x, double* y, long n) { for (long i = 0; i < n; i++) y[i] = nsg::nsg_log1p(x[i]); }
void t_log1p_libm(const double* x, double* y, long n) { for (long i = 0; i < n; i++) y[i] = nsg::nsg_log1p_libm(x[i]); }
void t_wrap(const double* x, double* y, long n) { for (long i = 0; i < n; i++) y[i] = nsg::nsg_wrap_pi(x[i]); }
// gymnasium's wrap(x, -pi, pi) as written: the definition the closed form is measured against
void t_wrap_loop(const double* x, double* y, long n) {
  const double M = 3.141592653589793, m = -M, diff = M - m;
  for (long i = 0; i < n; i++) { double v = x[i]; while (v > M) v = v - diff; while (v < m) v = v + diff; y[i] = v; }
}
}
''' % ROOT


@pytest.fixture(scope="module")
def m():
    d = tempfile.mkdtemp()
    src, so = os.path.join(d, "m.cpp"), os.path.join(d, "m.so")
    open(src, "w").write(SRC)
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-mfma", "-I", os.path.join(ROOT, "include"), "-fPIC", "-shared", "-o", so, src])
    return C.CDLL(so)


def _ulps(got, want_ld):
    want = want_ld.astype(np.float64)
    ulp = np.spacing(np.abs(want))
    return np.max(np.abs(got.astype(np.longdouble) - want_ld) / ulp)


def _call(fn, x, nout=1):
    x = np.ascontiguousarray(x, dtype=np.float64)
    outs = [np.empty_like(x) for _ in range(nout)]
    fn(x.ctypes.data_as(C.c_void_p), *[o.ctypes.data_as(C.c_void_p) for o in outs], C.c_long(x.size))
    return outs


def test_sincos_within_one_ulp(m):
    rng = np.random.default_rng(0)
    xs = [rng.uniform(-r, r, 400_000) for r in (0.3, 4.0, 30.0, 1e3, 1e6)]
    k = np.arange(-2000, 2001)[:, None] * (np.pi / 2)
    near = np.concatenate([np.nextafter(k, np.inf), k, np.nextafter(k, -np.inf)]).ravel()
    # either side of the first-round / compensated-reduction switch (|y0| vs |x| * 2^-15)
    kk = np.arange(-400, 401)[:, None, None] * (np.pi / 2)
    rel = 2.0 ** -np.arange(8, 30)[None, :, None] * np.array([1.0, -1.0, 0.7, -1.3])[None, None, :]
    edge = (kk * (1.0 + rel)).ravel()
    for x in xs + [near, edge]:
        xl = x.astype(np.longdouble)
        got = {}
        for fn in ("t_sincos", "t_sincos_poly0", "t_sincos_poly1", "t_sincos_poly2"):   # the default and every polynomial form
            s, c = got[fn] = _call(getattr(m, fn), x, 2)
            assert _ulps(s, np.sin(xl)) <= 1.0 and _ulps(c, np.cos(xl)) <= 1.0, fn
        # compiler-fused and SGPR-addend forms are the same arithmetic (on the host both are __builtin_fma)
        assert np.array_equal(got["t_sincos_poly1"][0], got["t_sincos_poly2"][0]) and np.array_equal(got["t_sincos_poly1"][1], got["t_sincos_poly2"][1])


def test_exp_and_log1p_within_one_ulp(m):
    rng = np.random.default_rng(1)
    for r in (1e-3, 1.0, 8.0, 50.0, 700.0):
        x = rng.uniform(-r, r, 300_000)
        (y,) = _call(m.t_exp, x)
        assert _ulps(y, np.exp(x.astype(np.longdouble))) <= 1.0
    x = -rng.uniform(0.0, 1.0, 500_000)
    x = x[x > -1.0]
    (y,) = _call(m.t_log1p, x)
    assert _ulps(y, np.log1p(x.astype(np.longdouble))) <= 1.0
    for r in (1e-10, 0.5, 10.0, 1e6):
        x = rng.uniform(-min(r, 0.999), r, 200_000)
        (y,) = _call(m.t_log1p, x)
        assert _ulps(y, np.log1p(x.astype(np.longdouble))) <= 1.0
    (y,) = _call(m.t_exp, np.array([710.0, -800.0, np.nan, 0.0]))
    assert np.isinf(y[0]) and y[1] == 0.0 and np.isnan(y[2]) and y[3] == 1.0
    (y,) = _call(m.t_log1p, np.array([-1.0, -2.0, 0.0, np.inf]))
    assert y[0] == -np.inf and np.isnan(y[1]) and y[2] == 0.0 and np.isinf(y[3])


def _fma_libm():
    flags = open("/proc/cpuinfo").read()
    return " fma " in flags and " avx2 " in flags


@pytest.mark.skipif(not _fma_libm(), reason="libm dispatches exp to its FMA build only on CPUs with FMA + AVX2; nsg_exp restates that build")
def test_exp_and_log1p_equal_libm_bit_for_bit(m):
    """The samplers NumPy's Generator runs in C call libm's exp and log1p; nsg_exp_libm / nsg_log1p_libm (nsg_exp / nsg_log1p
    in a NSG_LIBM_EXACT unit) are glibc 2.35's own algorithms with every rounding where libm has it: equal on every argument tried, subnormal results and special values included.
    (np.exp / np.log1p on ARRAYS or scalars are NumPy's own SIMD kernels on an AVX-512 host and differ from libm - and so from what
    NumPy's own random module computes - in the last bit; math.exp / math.log1p are libm's.)"""
    import math

    rng = np.random.default_rng(7)
    for r in (1e-18, 1e-16, 1e-9, 1e-3, 0.5, 1.0, 5.0, 20.0, 100.0, 500.0, 700.0, 720.0, 746.0, 800.0, 1100.0):
        x = rng.uniform(-r, r, 12_000)
        (y,) = _call(m.t_exp_libm, x)
        want = np.array([math.exp(v) if v < 709.78 else np.inf for v in x])
        assert np.array_equal(y.view(np.uint64), want.view(np.uint64)), r
    for r in (1e-17, 1e-12, 1e-9, 1e-6, 1e-3, 0.3, 0.5, 1.0, 3.0, 100.0, 1e6, 1e15, 1e17, 1e300):
        x = rng.uniform(-min(r, 0.9999999), r, 12_000)
        (y,) = _call(m.t_log1p_libm, x)
        want = np.array([math.log1p(v) for v in x])
        assert np.array_equal(y.view(np.uint64), want.view(np.uint64)), r
    # either side of every branch test of s_log1p.c (they are made on the high word) and of e_exp.c
    e = np.array([-0.2928932188134524, -0.29289340972900390625, 0.41421356237309515, 0.41421365737915039, 2.0 ** -29, 2.0 ** -54, 2.0 ** 53, 2.0 ** -20, 1.0, 3.0])
    x = (e[:, None] + np.arange(-2000, 2001)[None, :] * np.spacing(e)[:, None]).ravel()
    (y,) = _call(m.t_log1p_libm, x)
    assert np.array_equal(y.view(np.uint64), np.array([math.log1p(v) for v in x]).view(np.uint64))
    e = np.array([512.0, -512.0, 709.782712893384, -745.1332191019411, -708.3964185322641, 2.0 ** -54, -2.0 ** -54, 1.0, -1.0])
    x = (e[:, None] + np.arange(-2000, 2001)[None, :] * np.spacing(np.abs(e))[:, None]).ravel()
    (y,) = _call(m.t_exp_libm, x)
    want = np.array([math.exp(v) if v < 709.782712893384 else np.inf for v in x])
    ok = (y.view(np.uint64) == want.view(np.uint64)) | (x > 709.78)     # (math.exp raises OverflowError beyond; checked below)
    assert ok.all()
    (y,) = _call(m.t_exp_libm, np.array([710.0, 1e5, -1e5, -np.inf, np.inf]))
    assert np.isinf(y[0]) and np.isinf(y[1]) and y[2] == 0.0 and y[3] == 0.0 and np.isinf(y[4])


def test_exp_literals_are_the_tables_words():
    import re
    import struct

    words = re.findall(r"0x([0-9a-f]{16})ULL", open(os.path.join(ROOT, "include", "nsg_pow_tab.inc")).read())
    src = open(os.path.join(ROOT, "ns_gym_amd", "csrc", "nsg_math.hip.h")).read()
    body = src[src.index("NSG_HD double nsg_exp_libm"):]
    lits = re.findall(r"(?:InvLn2N|Shift|NegLn2hiN|NegLn2loN|C2|C3|C4|C5) = (-?0x[0-9a-f.]+p[+-]\d+)", body)[:8]
    assert len(lits) == 8
    for lit, w in zip(lits, words[9:17]):
        assert struct.pack("<d", float.fromhex(lit)) == struct.pack("<Q", int(w, 16)), (lit, w)


def test_pymod_equals_python_float_mod_bit_for_bit(m):
    """Pendulum's angle_normalize uses Python's float %: nsg_pymod_pos must reproduce it exactly (incl. negative
    arguments, exact multiples, values next to multiples, tiny and large arguments)."""
    rng = np.random.default_rng(2)
    two_pi = 2 * np.pi
    k = np.arange(-3000, 3001, dtype=np.float64) * two_pi
    xs = np.concatenate([rng.uniform(-r, r, 300_000) for r in (1.0, 10.0, 1e3, 1e6, 1e9)] +
                        [k, np.nextafter(k, np.inf), np.nextafter(k, -np.inf), np.array([0.0, -0.0, 5e-324, -5e-324, 1e-300, -1e-300])])
    for mod in (two_pi, 1.0, 0.3, 7.5):
        y = np.empty_like(xs)
        m.t_pymod.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_long]
        m.t_pymod(xs.ctypes.data_as(C.c_void_p), C.c_double(mod), y.ctypes.data_as(C.c_void_p), C.c_long(xs.size))
        want = np.array([float(v) % mod for v in xs[:200_000]] )      # Python's own operator
        assert np.array_equal(y[:200_000], want)
        want_np = np.mod(xs, mod)                                     # NumPy's floor-mod agrees with Python's for floats
        assert np.array_equal(y, want_np) or np.array_equal(y[want_np != mod], want_np[want_np != mod])


def _wrap_cases():
    rng = np.random.default_rng(5)
    parts = [rng.uniform(-r, r, c) for r, c in ((3.2, 200_000), (7, 200_000), (70, 200_000), (130, 200_000), (600, 200_000), (5000, 200_000),
                                                 (1e5, 20_000), (1e6, 4000), (3e7, 300), (5e9, 6))]
    # both sides of every binade edge, and arguments a few turns above one (where the bulk hands over to single turns)
    e = np.arange(2, 25)
    edges = (2.0 ** e)[:, None] + np.arange(-300, 301)[None, :] * (2.0 ** (e - 53))[:, None]
    above = (2.0 ** np.arange(7, 23))[:, None] + np.arange(0, 400)[None, :] * (2 * np.pi) * (1 + 1e-13 * np.arange(0, 400)[None, :])
    parts += [edges.ravel(), -edges.ravel(), above.ravel(), -above.ravel(), np.array([0.0, -0.0, np.pi, -np.pi, np.nextafter(np.pi, 4), -np.nextafter(np.pi, 4)])]
    return np.concatenate(parts)


def test_wrap_pi_equals_the_reference_loop_turn_for_turn(m):
    """`nsg_wrap_pi` takes the turns of a whole binade at once; the result is the loop's, in every bit, up to 5e9 rad (8e8 turns)."""
    x = _wrap_cases()
    (got,), (want,) = _call(m.t_wrap, x), _call(m.t_wrap_loop, x)
    bad = np.flatnonzero(got.view(np.uint64) != want.view(np.uint64))
    assert bad.size == 0, (bad.size, x[bad[:3]], got[bad[:3]], want[bad[:3]])
    # where the reference's loop never returns (x - diff == x) the argument comes back as it came; NaN fails the loop's test
    for v in (2.0 ** 56, -2.0 ** 60, 1e300, np.inf, -np.inf):
        assert _call(m.t_wrap, np.array([v]))[0][0] == v
    assert np.isnan(_call(m.t_wrap, np.array([np.nan]))[0][0])


def test_oracle_wrap_equals_the_reference_loop(m):
    """The oracle runs the loop itself below 2^22 and brings larger values down binade by binade first: same results as the loop."""
    import ctypes

    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libnsgym_oracle.so"))
    lib.orc_wrap_pi.restype = ctypes.c_double
    lib.orc_wrap_pi.argtypes = [ctypes.c_double]
    rng = np.random.default_rng(6)
    x = np.concatenate([rng.uniform(-1e7, 1e7, 400), rng.uniform(-3e8, 3e8, 40), rng.uniform(-5e9, 5e9, 4), [4194304.0, -4194304.0, 4194303.9999, 8388608.0]])
    (want,) = _call(m.t_wrap_loop, x)
    got = np.array([lib.orc_wrap_pi(float(v)) for v in x])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
