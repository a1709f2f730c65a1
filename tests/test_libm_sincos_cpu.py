"""ns_gym_amd/csrc/nsg_libm.hip.h restates libm's sin / cos (glibc 2.35, the FMA build its x86-64 entry points dispatch to) - the
functions np.sin / np.cos resolve to in the reference - with every rounding where libm has it.  The header is plain C++ under
NSG_HD: compiled for the host here and compared with libm itself, bit for bit, over every argument range of the algorithm.
(`sin` and `cos` are called through volatile pointers: gcc merges sin(x) + cos(x) into sincos(x), and glibc's sincos is NOT the
FMA build - its results differ from sin / cos in the last bit for 0.06 % of arguments, which is also why the oracle is built with
-fno-builtin-sin -fno-builtin-cos.)"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cmath>
#include <cstring>
#include "%(root)s/ns_gym_amd/csrc/nsg_libm.hip.h"
static const unsigned long long TABW[NSG_SINCOS_TAB_WORDS] = {
#include "%(root)s/include/nsg_sincos_tab.inc"
};
static const unsigned long long POWFW[NSG_POWF_TAB_WORDS] = {
#include "%(root)s/include/nsg_powf_tab.inc"
};
static double (*volatile ppow)(double, double) = pow;
static float (*volatile ppowf)(float, float) = powf;
static double (*volatile psin)(double) = sin;
static double (*volatile pcos)(double) = cos;
extern "C" {
// returns how many of the n arguments give a sin or a cos that differs from libm's in any bit
long t_compare(const double* x, long n, double* first_bad) {
  double tab[NSG_SINCOS_TAB_WORDS];
  memcpy(tab, TABW, sizeof(tab));
  const nsg::LibmTab tb{tab};
  long bad = 0;
  for (long i = 0; i < n; i++) {
    const double s = nsg::nsg_sin_libm(tb, x[i]), c = nsg::nsg_cos_libm(tb, x[i]), ls = psin(x[i]), lc = pcos(x[i]);
    const double s2 = nsg::nsg_sin_libm_merged<false>(tb, x[i]), c2 = nsg::nsg_cos_libm_merged<false>(tb, x[i]);      // the merged body for every range
    const double s3 = nsg::nsg_sin_libm_merged<true>(tb, x[i]), c3 = nsg::nsg_cos_libm_merged<true>(tb, x[i]);
    if (memcmp(&s, &ls, 8) || memcmp(&c, &lc, 8) || memcmp(&s2, &ls, 8) || memcmp(&c2, &lc, 8) || memcmp(&s3, &ls, 8) || memcmp(&c3, &lc, 8)) { if (!bad) *first_bad = x[i]; bad++; }
  }
  return bad;
}
// x ** 2 on a float64 scalar: how many differ from libm's pow(x, 2.0); *not_product counts where pow itself is not x * x
long t_compare_sq(const double* x, long n, double* first_bad, long* not_product) {
  const nsg::PowTab tb{nsg::kNsgPowTab};
  long bad = 0;
  for (long i = 0; i < n; i++) {
    const double a = nsg::nsg_sq_libm(tb, x[i]), w = ppow(x[i], 2.0);
    if (memcmp(&a, &w, 8) && !(a != a && w != w)) { if (!bad) *first_bad = x[i]; bad++; }
    if (w != x[i] * x[i]) ++*not_product;
  }
  return bad;
}
long t_compare_sqf(const float* x, long n, float* first_bad, long* not_product) {
  const nsg::PowTab tb{POWFW};
  long bad = 0;
  for (long i = 0; i < n; i++) {
    const float a = nsg::nsg_sqf_libm(tb, x[i]), w = ppowf(x[i], 2.0f);
    if (memcmp(&a, &w, 4) && !(a != a && w != w)) { if (!bad) *first_bad = x[i]; bad++; }
    if (w != x[i] * x[i]) ++*not_product;
  }
  return bad;
}
// beyond the restated range (|x| >= 105414336, inf, NaN) every form answers with the kernels' own sincos - and none lets such bits reach its
// table index: returns how many of the n arguments give different answers in the three forms (NaN == NaN here)
long t_extremes_differ(const double* x, long n) {
  double tab[NSG_SINCOS_TAB_WORDS];
  memcpy(tab, TABW, sizeof(tab));
  const nsg::LibmTab tb{tab};
  long bad = 0;
  for (long i = 0; i < n; i++) {
    const double s1 = nsg::nsg_sin_libm(tb, x[i]), s2 = nsg::nsg_sin_libm_merged<false>(tb, x[i]), s3 = nsg::nsg_sin_libm_merged<true>(tb, x[i]);
    const double c1 = nsg::nsg_cos_libm(tb, x[i]), c2 = nsg::nsg_cos_libm_merged<false>(tb, x[i]), c3 = nsg::nsg_cos_libm_merged<true>(tb, x[i]);
    if (memcmp(&s1, &s2, 8) || memcmp(&s1, &s3, 8) || memcmp(&c1, &c2, 8) || memcmp(&c1, &c3, 8)) bad++;
  }
  return bad;
}
long t_merged_sincos_differs(const double* x, long n) {   // what gcc's sincos() merge would have compared against
  long bad = 0;
  for (long i = 0; i < n; i++) { double s, c; sincos(x[i], &s, &c); const double ls = psin(x[i]), lc = pcos(x[i]); if (s != ls || c != lc) bad++; }
  return bad;
}
}
''' % {"root": ROOT}


@pytest.fixture(scope="module")
def m():
    d = tempfile.mkdtemp()
    src, so = os.path.join(d, "m.cpp"), os.path.join(d, "m.so")
    open(src, "w").write(SRC)
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-mfma", "-I", os.path.join(ROOT, "include"), "-fPIC", "-shared", "-o", so, src])
    lib = C.CDLL(so)
    lib.t_compare.restype = C.c_long
    lib.t_merged_sincos_differs.restype = C.c_long
    lib.t_extremes_differ.restype = C.c_long
    lib.t_compare_sq.restype = C.c_long
    lib.t_compare_sqf.restype = C.c_long
    return lib


def _bad(m, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    first = C.c_double(0.0)
    n = m.t_compare(x.ctypes.data_as(C.c_void_p), C.c_long(x.size), C.byref(first))
    return n, first.value


def _fma_libm():
    flags = open("/proc/cpuinfo").read()
    return " fma " in flags and " avx2 " in flags


@pytest.mark.skipif(not _fma_libm(), reason="libm dispatches to its FMA build only on CPUs with FMA + AVX2; the restatement is of that build")
def test_sin_and_cos_equal_libm_bit_for_bit(m):
    rng = np.random.default_rng(0)
    total = 0
    for r in (1e-9, 1e-6, 0.13, 0.3, 0.9, 2.5, 8.0, 100.0, 1e4, 1e6, 1.05e8):      # every branch of __sin / __cos below 105414336
        x = rng.uniform(-r, r, 1_500_000)
        n, first = _bad(m, x)
        assert n == 0, (r, n, first)
        total += x.size
    # around the multiples of pi/2 and around every threshold of the algorithm
    k = np.arange(-3000, 3001)[:, None] * (np.pi / 2)
    near = k + np.arange(-40, 41)[None, :] * np.spacing(np.abs(k) + 1e-300)
    edges = np.array([0.126, 0.855469, 2.426265, 2.0 ** -26, 2.0 ** -27, np.pi / 4, 1 / 128, 0.859375, 105414330.0])      # (the last range ends at 105414336.0 = high word 0x419921fb)
    around = (edges[:, None] * np.array([1.0, -1.0])[None, :]).ravel()[:, None] + np.arange(-3000, 3001)[None, :] * np.spacing(np.abs(edges).repeat(2))[:, None]
    for x in (near.ravel(), around.ravel()):
        n, first = _bad(m, x)
        assert n == 0, (n, first)
        total += x.size
    assert total > 1.6e7


def test_arguments_beyond_the_restated_range_never_reach_a_table_index(m):
    """|x| >= 105414336, inf, NaN: the merged form evaluates its shared body for every lane, so such lanes must be handed a harmless
    argument first (their bits as a table index would be a wild read); all three forms give the fallback's answer."""
    x = np.array([105414336.0, 105414337.0, 1e9, -1e10, 3.3e12, 2.0 ** 45, 2.0 ** 45 * 1.5, -2.0 ** 52, 1e100, -1e300, np.inf, -np.inf, np.nan,
                  np.nextafter(105414336.0, 0)] + list(np.random.default_rng(8).uniform(-1e15, 1e15, 2000)))
    assert m.t_extremes_differ(x.ctypes.data_as(C.c_void_p), C.c_long(x.size)) == 0


@pytest.mark.skipif(not _fma_libm(), reason="needs the FMA build")
def test_gccs_sincos_merge_is_not_libms_sin_and_cos(m):
    """Documents why the oracle is compiled with -fno-builtin-sin -fno-builtin-cos: sincos() differs from sin() / cos() in the last bit."""
    x = np.random.default_rng(1).uniform(-3.0, 3.0, 2_000_000)
    n = m.t_merged_sincos_differs(x.ctypes.data_as(C.c_void_p), C.c_long(x.size))
    assert 100 < n < 20_000, n


def test_table_is_libms_and_numpy_resolves_to_libm():
    """The shipped table is the installed libm's own; NumPy's float64 sin / cos ARE libm's on this image (what the restatement rests on)."""
    import math
    import struct
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import extract_libm_sincos_table as E

    out = os.path.join(tempfile.mkdtemp(), "tab.inc")
    E.main(out)
    assert open(out).read() == open(os.path.join(ROOT, "include", "nsg_sincos_tab.inc")).read()
    x = np.random.default_rng(2).uniform(-40.0, 40.0, 100_000)
    assert np.array_equal(np.sin(x), np.array([math.sin(v) for v in x])) and np.array_equal(np.cos(x), np.array([math.cos(v) for v in x]))
    assert np.array_equal(np.array([float(np.cos(float(v))) for v in x[:20_000]]), np.array([math.cos(v) for v in x[:20_000]]))   # the scalar path gymnasium takes


@pytest.mark.skipif(not _fma_libm(), reason="needs the FMA build")
def test_scalar_square_equals_libm_pow_bit_for_bit(m):
    """`x ** 2` on a float64 scalar is libm's pow(x, 2.0) in the reference (checked below), not the product; nsg_sq_libm restates it."""
    rng = np.random.default_rng(3)
    total = nots = 0
    for r in (1e-30, 1e-12, 1e-6, 0.1, 1.0, 1.0001, 2.0, 30.0, 1e3, 1e6, 1e30, 1e100):
        x = rng.uniform(-r, r, 1_000_000)
        first, notp = C.c_double(0.0), C.c_long(0)
        n = m.t_compare_sq(x.ctypes.data_as(C.c_void_p), C.c_long(x.size), C.byref(first), C.byref(notp))
        assert n == 0, (r, n, first.value)
        total += x.size
        nots += notp.value
    assert 2e-4 < nots / total < 3e-3, nots          # pow(x, 2) is NOT x * x for ~0.08 % of arguments: why this function exists
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-200, 1e200, 5e-324, 2.0 ** -368, 2.0 ** 368, np.nextafter(2.0 ** 368, 0), np.nextafter(2.0 ** -368, 0)])
    first, notp = C.c_double(0.0), C.c_long(0)
    assert m.t_compare_sq(sp.ctypes.data_as(C.c_void_p), C.c_long(sp.size), C.byref(first), C.byref(notp)) == 0, first.value


@pytest.mark.skipif(not _fma_libm(), reason="needs the FMA build")
def test_float32_scalar_square_equals_libm_powf_bit_for_bit(m):
    """Pendulum's `u ** 2` (a float32 scalar): every float32 of [-2, 2] down to 2^-20 in magnitude, and a sample of the rest."""
    e = np.arange(107, 129, dtype=np.uint32)[:, None] << 23
    pos = (e | np.arange(0, 1 << 23, dtype=np.uint32)[None, :]).ravel()
    rest = ((np.arange(65, 189, dtype=np.uint32)[:, None] << 23) | np.arange(0, 1 << 23, 97, dtype=np.uint32)[None, :]).ravel()
    nots = total = 0
    for bits in (pos, pos | np.uint32(1 << 31), rest, rest | np.uint32(1 << 31)):
        x = np.ascontiguousarray(bits).view(np.float32)
        first, notp = C.c_float(0.0), C.c_long(0)
        n = m.t_compare_sqf(x.ctypes.data_as(C.c_void_p), C.c_long(x.size), C.byref(first), C.byref(notp))
        assert n == 0, (n, first.value)
        total += x.size
        nots += notp.value
    assert 1e-4 < nots / total < 1e-2, nots
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 1e-30, 1e30, 1e-45], dtype=np.float32)
    first, notp = C.c_float(0.0), C.c_long(0)
    assert m.t_compare_sqf(sp.ctypes.data_as(C.c_void_p), C.c_long(sp.size), C.byref(first), C.byref(notp)) == 0, first.value


def test_pow_tables_are_libms_and_scalar_power_resolves_to_libm():
    import ctypes
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import extract_libm_pow_tables as E

    d = tempfile.mkdtemp()
    E.main(os.path.join(d, "nsg_pow_tab.inc"))
    for f in ("nsg_pow_tab.inc", "nsg_powf_tab.inc"):
        assert open(os.path.join(d, f)).read() == open(os.path.join(ROOT, "include", f)).read(), f
    libm = ctypes.CDLL("libm.so.6")
    libm.pow.restype, libm.pow.argtypes = ctypes.c_double, [ctypes.c_double] * 2
    libm.powf.restype, libm.powf.argtypes = ctypes.c_float, [ctypes.c_float] * 2
    x = np.random.default_rng(4).uniform(-30.0, 30.0, 60_000)
    want = np.array([libm.pow(float(v), 2.0) for v in x])
    assert np.array_equal(np.array([float(np.float64(v) ** 2) for v in x]), want)      # NumPy's scalar power (Acrobot's dtheta ** 2)
    assert np.array_equal(np.array([float(v) ** 2 for v in x]), want)                  # CPython's float power (lc1 ** 2)
    assert (want != x * x).sum() > 5                                                   # ... and neither is the product
    assert np.array_equal(np.square(x), x * x) and np.array_equal(x ** 2, x * x)       # np.square / array power ARE (CartPole)
    xf = x.astype(np.float32)[:30_000]
    assert np.array_equal(np.array([np.float32(v) ** 2 for v in xf], dtype=np.float32), np.array([libm.powf(float(v), 2.0) for v in xf], dtype=np.float32))
