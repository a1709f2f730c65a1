// nsg_libm.hip.h — float64 sin / cos that return what the reference's sin / cos return, bit for bit.
//
// The base MDPs call np.sin / np.cos on float64 scalars; NumPy 2.2 resolves those to libm's sin / cos (on this image: glibc 2.35,
// whose x86-64 entry points dispatch to the FMA build of sysdeps/ieee754/dbl-64/s_sin.c on every CPU with FMA + AVX2).  An
// integrator that is to reproduce the reference's float64 STATE - not merely stay within a tolerance until an unstable plant or a
// chaotic one has amplified the last ulp (Acrobot, a balanced CartPole: profiles/NOTEBOOK.md) - has to evaluate the same
// algorithm with the same roundings.  This is a restatement of that algorithm [UPSTREAM glibc 2.35, IBM Accurate Mathematical
// Library: s_sin.c __sin / __cos, do_sin, do_cos, reduce_sincos, TAYLOR_SIN; usncs.h constants], with every fused multiply-add
// exactly where the image's libm.so.6 has one (read off the disassembly of its FMA variant; tools/extract_libm_sincos_table.py
// documents the table), for |x| < 105414336 (high word below 0x419921fb; beyond that the reference takes __branred; no episode gets there, and nsg_sincos
// answers).  tests/test_libm_sincos_cpu.py compiles this header for the host and compares it with libm over 4e8 arguments: equal.
//
// Everything outside __builtin_fma is a single IEEE operation (the build has -ffp-contract=off).
#pragma once
#include "nsg_math.hip.h"

namespace nsg {

// sin(k/128), cos(k/128) as (high, low) pairs, k = 0 .. 109: 440 float64 bit patterns (3.5 KB)
#define NSG_SINCOS_TAB_WORDS 440

struct LibmTab {
  const double* x;   // NSG_SINCOS_TAB_WORDS doubles
};

NSG_HD double lm_abs(double v) { return __builtin_fabs(v); }
NSG_HD long long lm_bits(double v) { return __builtin_bit_cast(long long, v); }
NSG_HD double lm_from_bits(unsigned long long b) { return __builtin_bit_cast(double, b); }
NSG_HD double lm_copysign(double mag, double sgn) { return __builtin_copysign(mag, sgn); }

// TAYLOR_SIN(xx, x, dx), |x| < 0.126
NSG_HD double lm_taylor_sin(double x, double dx) {
  const double s1 = -0.16666666666666666, s2 = 0.008333333333332329, s3 = -0.00019841269834414642, s4 = 2.755729806860771e-06,
               s5 = -2.5022014848318398e-08;
  const double xx = x * x;
  double p = __builtin_fma(xx, s5, s4);
  p = __builtin_fma(xx, p, s3);
  p = __builtin_fma(xx, p, s2);
  p = __builtin_fma(xx, p, s1);
  const double t = __builtin_fma(p, x, -(0.5 * dx));
  return x + __builtin_fma(xx, t, dx);
}

// do_sin(x, dx): sin(x + dx) for |x| < 0.855469 (after reduction: |x| <= pi/4)
NSG_HD double lm_do_sin(const LibmTab tb, double x, double dx) {
  const double big = 52776558133248.0, sn3 = -0.16666666666666488, sn5 = 0.008333332142857223, cs2 = 0.5,
               cs4 = -0.04166666666666644, cs6 = 0.001388888740079376;
  const double ax = lm_abs(x);
  if (ax < 0.126) return lm_taylor_sin(x, dx);
  if (!(x > 0.0)) dx = -dx;
  const double u = big + ax;
  const double xr = ax - (u - big);
  const int k = (int)((unsigned)lm_bits(u) << 2);
  const double xx = xr * xr;
  const double p = __builtin_fma(xx, sn5, sn3);
  const double s = xr + __builtin_fma(xr * xx, p, dx);
  const double q = __builtin_fma(xx, __builtin_fma(xx, cs6, cs4), cs2);
  const double c = __builtin_fma(xr, dx, xx * q);
  const double sn = tb.x[k], ssn = tb.x[k + 1], cs = tb.x[k + 2], ccs = tb.x[k + 3];
  const double cor = __builtin_fma(s, cs, __builtin_fma(-c, sn, __builtin_fma(s, ccs, ssn)));
  return lm_copysign(lm_abs(sn + cor), x);
}

// do_cos(x, dx): cos(x + dx)
NSG_HD double lm_do_cos(const LibmTab tb, double x, double dx) {
  const double big = 52776558133248.0, sn3 = -0.16666666666666488, sn5 = 0.008333332142857223, cs2 = 0.5,
               cs4 = -0.04166666666666644, cs6 = 0.001388888740079376;
  if (x < 0.0) dx = -dx;
  const double ax = lm_abs(x);
  const double u = big + ax;
  const double xr = (ax - (u - big)) + dx;
  const int k = (int)((unsigned)lm_bits(u) << 2);
  const double xx = xr * xr;
  const double p = __builtin_fma(xx, sn5, sn3);
  const double s = __builtin_fma(xr * xx, p, xr);
  const double c = xx * __builtin_fma(xx, __builtin_fma(xx, cs6, cs4), cs2);
  const double sn = tb.x[k], ssn = tb.x[k + 1], cs = tb.x[k + 2], ccs = tb.x[k + 3];
  const double cor = __builtin_fma(-s, sn, __builtin_fma(-c, cs, __builtin_fma(-s, ssn, ccs)));
  return cs + cor;
}

// reduce_sincos: x = n * pi/2 + (a + da), |a| <= pi/4, for |x| < 105414350; returns n mod 4
NSG_HD int lm_reduce(double x, double& a, double& da) {
  const double hpinv = 0.6366197723675814, toint = 6755399441055744.0, mp1 = 1.5707963407039642, mp2 = -1.3909067564377153e-08,
               pp3 = -4.97899623147991e-17, pp4 = -1.9034889620193266e-25;
  const double t = __builtin_fma(x, hpinv, toint);
  const double xn = t - toint;
  const int n = (int)((unsigned)lm_bits(t) & 3u);
  const double y = __builtin_fma(-xn, mp2, __builtin_fma(-xn, mp1, x));
  const double t2 = __builtin_fma(-xn, pp3, y);
  double db = __builtin_fma(-pp3, xn, y - t2);
  const double b = __builtin_fma(-xn, pp4, t2);
  db = db + __builtin_fma(-xn, pp4, t2 - b);
  a = b;
  da = db;
  return n;
}

NSG_HD double nsg_sin_libm(const LibmTab tb, double x) {
  const double hp0 = 1.5707963267948966, hp1 = 6.123233995736766e-17;
  const unsigned k = (unsigned)((unsigned long long)lm_bits(x) >> 32) & 0x7fffffffu;
  if (k < 0x3e500000u) return x;                                   // |x| < 2^-26
  if (k < 0x3feb6000u) return lm_do_sin(tb, x, 0.0);               // |x| < 0.855469
  if (k < 0x400368fdu) {                                           // |x| < 2.426265
    const double t = hp0 - lm_abs(x);
    return lm_copysign(lm_abs(lm_do_cos(tb, t, hp1)), x);
  }
  if (k < 0x419921fbu) {                                           // |x| < 105414336
    double a, da;
    const int n = lm_reduce(x, a, da);
    const double r = (n & 1) ? lm_do_cos(tb, a, da) : lm_do_sin(tb, a, da);
    return (n & 2) ? -r : r;
  }
  double s, c;                                                     // (the reference: __branred; never reached by an episode)
  nsg_sincos(x, &s, &c);
  return s;
}

NSG_HD double nsg_cos_libm(const LibmTab tb, double x) {
  const double hp0 = 1.5707963267948966, hp1 = 6.123233995736766e-17;
  const unsigned k = (unsigned)((unsigned long long)lm_bits(x) >> 32) & 0x7fffffffu;
  if (k < 0x3e400000u) return 1.0;                                 // |x| < 2^-27
  if (k < 0x3feb6000u) return lm_do_cos(tb, x, 0.0);
  if (k < 0x400368fdu) {
    const double y = hp0 - lm_abs(x);
    const double a = y + hp1;
    const double da = (y - a) + hp1;
    return lm_do_sin(tb, a, da);
  }
  if (k < 0x419921fbu) {
    double a, da;
    const int n = lm_reduce(x, a, da) + 1;
    const double r = (n & 1) ? lm_do_cos(tb, a, da) : lm_do_sin(tb, a, da);
    return (n & 2) ? -r : r;
  }
  double s, c;
  nsg_sincos(x, &s, &c);
  return c;
}

}  // namespace nsg
