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
void t_log1p(const double* x, double* y, long n) { for (long i = 0; i < n; i++) y[i] = nsg::nsg_log1p(x[i]); }
}
''' % ROOT


@pytest.fixture(scope="module")
def m():
    d = tempfile.mkdtemp()
    src, so = os.path.join(d, "m.cpp"), os.path.join(d, "m.so")
    open(src, "w").write(SRC)
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src])
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
