// nsg_libm.hip.h — the transcendentals of the integrators as the reference's host evaluates them, bit for bit: float64 sin / cos,
// `x ** 2` on a float64 / float32 scalar (libm's pow / powf), and the switch (NSG_LIBM_EXACT) that routes a unit through them.
// (exp and log1p: nsg_math.hip.h.)
//
// sin / cos.  The base MDPs call np.sin / np.cos on float64 scalars; NumPy 2.2 resolves those to libm's sin / cos (on this image:
// glibc 2.35, whose x86-64 entry points dispatch to the FMA build of sysdeps/ieee754/dbl-64/s_sin.c on every CPU with FMA + AVX2).
// An integrator that is to reproduce the reference's float64 STATE - not merely stay within a tolerance until an unstable plant or
// a chaotic one has amplified the last ulp (Acrobot, a balanced CartPole: profiles/NOTEBOOK.md) - has to evaluate the same
// algorithm with the same roundings.  This is a restatement of that algorithm [UPSTREAM glibc 2.35, IBM Accurate Mathematical
// Library: s_sin.c __sin / __cos, do_sin, do_cos, reduce_sincos, TAYLOR_SIN; usncs.h constants], with every fused multiply-add
// exactly where the image's libm.so.6 has one (read off the disassembly of its FMA variant; tools/extract_libm_sincos_table.py
// documents the table), for |x| < 105414336 (high word below 0x419921fb; beyond that the reference takes __branred; no episode
// gets there, and nsg_sincos answers).  tests/test_libm_sincos_cpu.py compiles this header for the host and compares every form
// below with libm over 1.6e7 arguments in every range and around every threshold: equal.
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
  double p = nsg_fma_coef(xx, s5, s4);     // (constant addends through nsg_fma_coef: an SGPR operand of ONE v_fma_f64, not a coefficient parked
  p = nsg_fma_coef(xx, p, s3);            //  in a VGPR pair for the whole kernel - nsg_math.hip.h; the same single rounding)
  p = nsg_fma_coef(xx, p, s2);
  p = nsg_fma_coef(xx, p, s1);
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
  const double p = nsg_fma_coef(xx, sn5, sn3);
  const double s = xr + __builtin_fma(xr * xx, p, dx);
  const double q = nsg_fma_coef(xx, nsg_fma_coef(xx, cs6, cs4), cs2);
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
  const double p = nsg_fma_coef(xx, sn5, sn3);
  const double s = __builtin_fma(xr * xx, p, xr);
  const double c = xx * nsg_fma_coef(xx, nsg_fma_coef(xx, cs6, cs4), cs2);
  const double sn = tb.x[k], ssn = tb.x[k + 1], cs = tb.x[k + 2], ccs = tb.x[k + 3];
  const double cor = __builtin_fma(-s, sn, __builtin_fma(-c, cs, __builtin_fma(-s, ssn, ccs)));
  return cs + cor;
}

// reduce_sincos: x = n * pi/2 + (a + da), |a| <= pi/4, for |x| < 105414350; returns n mod 4
NSG_HD int lm_reduce(double x, double& a, double& da) {
  const double hpinv = 0.6366197723675814, toint = 6755399441055744.0, mp1 = 1.5707963407039642, mp2 = -1.3909067564377153e-08,
               pp3 = -4.97899623147991e-17, pp4 = -1.9034889620193266e-25;
  const double t = nsg_fma_coef(x, hpinv, toint);
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

// do_sin and do_cos in ONE pass, the lane choosing: cosm ? cos(x + dx) : sin(x + dx).  The two share the table index, the table
// reads and both polynomials, and differ in where dx enters and in how the table values combine; written with selects, a wavefront
// whose lanes sit in different argument ranges (Pendulum's angle, Acrobot's) evaluates ONE body instead of do_sin, do_cos and
// their callers' variants one after the other.  Every lane's operation sequence is do_sin's or do_cos's own: same bits.
// |x| < 0.8555 (the callers' ranges, as for do_sin / do_cos): the table index is the low word of 2^45 * 1.5 + |x|, 0 .. 4 * 109.
NSG_HD double lm_do_sincos(const LibmTab tb, double x, double dx, bool cosm) {
  const double big = 52776558133248.0, sn3 = -0.16666666666666488, sn5 = 0.008333332142857223, cs2 = 0.5,
               cs4 = -0.04166666666666644, cs6 = 0.001388888740079376;
  const double ax = lm_abs(x);
  const double d = x < 0.0 ? -dx : dx;            // (do_sin tests !(x > 0): the same but at x = 0, where TAYLOR_SIN answers)
  const double u = big + ax;
  const double base = ax - (u - big);
  const double xr = cosm ? base + d : base;
  const int k = (int)((unsigned)lm_bits(u) << 2);
  const double xx = xr * xr;
  const double p = nsg_fma_coef(xx, sn5, sn3);
  const double q = nsg_fma_coef(xx, nsg_fma_coef(xx, cs6, cs4), cs2);
  const double t3 = xr * xx, xq = xx * q;
  const double s = cosm ? __builtin_fma(t3, p, xr) : xr + __builtin_fma(t3, p, d);
  const double c = cosm ? xq : __builtin_fma(xr, d, xq);
  const double sn = tb.x[k], ssn = tb.x[k + 1], cs = tb.x[k + 2], ccs = tb.x[k + 3];
  const double s1 = cosm ? -s : s;
  const double lead = cosm ? cs : sn;
  const double cor = __builtin_fma(s1, cosm ? sn : cs, __builtin_fma(-c, lead, __builtin_fma(s1, cosm ? ssn : ccs, cosm ? ccs : ssn)));
  const double r = lead + cor;
  if (!cosm && ax < 0.126) return lm_taylor_sin(x, dx);
  return cosm ? r : lm_copysign(lm_abs(r), x);
}

#if defined(__HIP_DEVICE_COMPILE__)
#define NSG_LM_ALL(c) __all(c)
#else
#define NSG_LM_ALL(c) (c)
#endif

// __sin / __cos with the ranges merged: each lane prepares its own (argument, tail, which function, sign) and one lm_do_sincos serves
// them all.  FAST: when every lane of the wavefront is in the no-reduction range the plain do_sin / do_cos runs - no selects.
// tests/test_libm_sincos_cpu.py runs both instantiations, and the branchy form below, against libm.
template <bool FAST>
NSG_HD double nsg_sin_libm_merged(const LibmTab tb, double x) {
  const double hp0 = 1.5707963267948966, hp1 = 6.123233995736766e-17;
  const unsigned k = (unsigned)((unsigned long long)lm_bits(x) >> 32) & 0x7fffffffu;
  if (FAST && NSG_LM_ALL(k >= 0x3e500000u && k < 0x3feb6000u)) return lm_do_sin(tb, x, 0.0);   // 2^-26 <= |x| < 0.855469
  double a = x, da = 0.0;
  bool cosm = false;
  int n = 0;
  if (k >= 0x3feb6000u) {
    if (k < 0x400368fdu) {                                           // |x| < 2.426265: sin x = +-cos(pi/2 - |x|)
      a = hp0 - lm_abs(x);
      da = hp1;
      cosm = true;
    } else if (k < 0x419921fbu) {                                    // |x| < 105414336
      n = lm_reduce(x, a, da);
      cosm = (n & 1) != 0;
    }
  }
  if (k >= 0x419921fbu) a = 0.0;        // (answered below; its bits must not reach the table index of the shared evaluation)
  double r = lm_do_sincos(tb, a, da, cosm);
  if (k >= 0x3feb6000u && k < 0x400368fdu) r = lm_copysign(lm_abs(r), x);
  if (n & 2) r = -r;
  if (k < 0x3e500000u) r = x;                                        // |x| < 2^-26
  if (k >= 0x419921fbu) {                                            // (the reference: __branred; never reached by an episode)
    double sn, cs;
    nsg_sincos(x, &sn, &cs);
    r = sn;
  }
  return r;
}

template <bool FAST>
NSG_HD double nsg_cos_libm_merged(const LibmTab tb, double x) {
  const double hp0 = 1.5707963267948966, hp1 = 6.123233995736766e-17;
  const unsigned k = (unsigned)((unsigned long long)lm_bits(x) >> 32) & 0x7fffffffu;
  if (FAST && NSG_LM_ALL(k >= 0x3e400000u && k < 0x3feb6000u)) return lm_do_cos(tb, x, 0.0);   // 2^-27 <= |x| < 0.855469
  double a = x, da = 0.0;
  bool cosm = true;
  int n = 0;
  if (k >= 0x3feb6000u) {
    if (k < 0x400368fdu) {                                           // cos x = sin(pi/2 - |x|)
      const double y = hp0 - lm_abs(x);
      a = y + hp1;
      da = (y - a) + hp1;
      cosm = false;
    } else if (k < 0x419921fbu) {
      n = lm_reduce(x, a, da) + 1;
      cosm = (n & 1) != 0;
    }
  }
  if (k >= 0x419921fbu) a = 0.0;        // (answered below; its bits must not reach the table index of the shared evaluation)
  double r = lm_do_sincos(tb, a, da, cosm);
  if (n & 2) r = -r;
  if (k < 0x3e400000u) r = 1.0;                                      // |x| < 2^-27
  if (k >= 0x419921fbu) {
    double sn, cs;
    nsg_sincos(x, &sn, &cs);
    r = cs;
  }
  return r;
}

// __sin / __cos as written: one branch per argument range.  What the exact units of CartPole (one range per wavefront anyway), Acrobot
// and the MountainCars call: there the merged form buys nothing and costs registers (MountainCar's fused rollout 71 -> 81 VGPRs, seven
// wavefronts per SIMD -> five: +8 % per step; Acrobot +5 %), while Pendulum - whose angle is anywhere - gains 15 % per step from it
// (12.9 -> 11.0 us at 2^18 envs; profiles/NOTEBOOK.md).
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

// ---- x ** 2 ----------------------------------------------------------------------------------------------------------------------
// gymnasium writes Acrobot's and Pendulum's squares as `x ** 2` on float64 SCALARS; NumPy's scalar power and CPython's float power
// both call libm's pow(x, 2.0) [UPSTREAM glibc 2.35 sysdeps/ieee754/dbl-64/e_pow.c, from ARM's optimized-routines: log_inline,
// exp_inline], which is within 0.52 ulp but not always the correctly rounded product: 0.08 % of arguments differ from x * x in the
// last bit (np.square and array ** 2 ARE the product - CartPole uses those).  This is pow's main path specialised to y = 2, with
// every fused multiply-add where the image's libm.so.6 (__pow_fma) has one; tools/extract_libm_pow_tables.py documents the tables.
// Valid for 2^-368 <= |x| < 2^368 (there |2 log|x|| < 512 and pow takes no special case but x = +-1); outside it answers x * x,
// which IS pow's answer for 0, inf and NaN, and which no integrator's state reaches otherwise.
struct PowTab {
  const unsigned long long* w;   // NSG_POW_TAB_WORDS bit patterns, layout in include/nsg_pow_tab.inc
};

NSG_HD double nsg_sq_libm(const PowTab tb, double x) {
  const unsigned long long* W = tb.w;
  const unsigned long long ix = (unsigned long long)lm_bits(x) & 0x7fffffffffffffffULL;
  const unsigned top = (unsigned)(ix >> 52);
  if (top - 655u >= 736u) return x * x;
  // log_inline(ix): |x| = 2^k z, z in [OFF, 2 OFF), subinterval i of 128; log|x| = lhi + llo
  const unsigned long long tmp = ix - 0x3fe6955500000000ULL;
  const int i = (int)((tmp >> 45) & 127u);
  const double kd = (double)(int)((long long)tmp >> 52);
  const double z = lm_from_bits(ix - (tmp & 0xfff0000000000000ULL));
  const double ln2hi = lm_from_bits(W[0]), ln2lo = lm_from_bits(W[1]);
  const double invc = lm_from_bits(W[17 + 3 * i]), logc = lm_from_bits(W[18 + 3 * i]), logctail = lm_from_bits(W[19 + 3 * i]);
  const double t1 = __builtin_fma(kd, ln2hi, logc);
  const double r = __builtin_fma(z, invc, -1.0);
  const double ar = r * lm_from_bits(W[2]);
  const double lo1 = __builtin_fma(kd, ln2lo, logctail);
  const double p12 = __builtin_fma(r, lm_from_bits(W[4]), lm_from_bits(W[3]));
  const double p34 = __builtin_fma(r, lm_from_bits(W[6]), lm_from_bits(W[5]));
  const double t2 = r + t1;
  const double ar2 = r * ar;
  const double ar3 = r * ar2;
  const double lo3 = __builtin_fma(ar, r, -ar2);
  const double lo2 = (t1 - t2) + r;
  const double p56 = __builtin_fma(r, lm_from_bits(W[8]), lm_from_bits(W[7]));
  const double hi = t2 + ar2;
  const double lo4 = (t2 - hi) + ar2;
  const double pp = __builtin_fma(ar2, __builtin_fma(p56, ar2, p34), p12);
  const double lo = __builtin_fma(pp, ar3, ((lo1 + lo2) + lo3) + lo4);
  const double lhi = hi + lo;
  const double llo = (hi - lhi) + lo;
  // y log|x| = ehi + elo, y = 2
  const double ehi = 2.0 * lhi;
  const double elo = __builtin_fma(2.0, llo, __builtin_fma(lhi, 2.0, -ehi));
  // exp_inline(ehi, elo)
  const unsigned abstop = (unsigned)((unsigned long long)lm_bits(ehi) >> 52) & 0x7ffu;
  if (abstop - 0x3c9u >= 0x3fu) return 1.0 + ehi;   // |ehi| < 2^-54 (x = +-1; the range check above excludes |ehi| >= 512)
  const double shift = lm_from_bits(W[10]);
  const double kk = __builtin_fma(ehi, lm_from_bits(W[9]), shift);
  const unsigned long long ki = (unsigned long long)lm_bits(kk);
  const double kn = kk - shift;
  double rr = __builtin_fma(kn, lm_from_bits(W[12]), __builtin_fma(kn, lm_from_bits(W[11]), ehi));
  rr = elo + rr;
  const int j = 2 * (int)(ki & 127u);
  const double tail = lm_from_bits(W[401 + j]);
  const double scale = lm_from_bits(W[402 + j] + (ki << 45));
  const double c23 = __builtin_fma(rr, lm_from_bits(W[14]), lm_from_bits(W[13]));
  const double tr = rr + tail;
  const double r2 = rr * rr;
  const double c45 = __builtin_fma(rr, lm_from_bits(W[16]), lm_from_bits(W[15]));
  const double u = __builtin_fma(c23, r2, tr);
  const double v = __builtin_fma(c45, r2 * r2, u);
  return __builtin_fma(v, scale, scale);
}

// float32 x ** 2: powf(x, 2.0f) [UPSTREAM glibc 2.35 sysdeps/ieee754/flt-32/e_powf.c, __powf_fma] - Pendulum's `u ** 2`, a float32
// scalar power in the reference.  log2 by a 16-entry table and a degree-4 polynomial in float64, exp2 by a 32-entry table and a
// degree-3 polynomial, ONE rounding to float32 at the end (0.09 % of arguments differ from the float32 product).  Valid for
// 2^-62 <= |x| < 2^62; outside it answers x * x (pow's answer for 0, inf, NaN; the clipped torque is within [-2, 2]).
#define NSG_POWF_TAB_WORDS 73

NSG_HD float nsg_sqf_libm(const PowTab tb, float x) {
  const unsigned long long* W = tb.w;
  const unsigned ix = __builtin_bit_cast(unsigned, x) & 0x7fffffffu;
  if ((ix >> 23) - 65u >= 124u) return x * x;
  const unsigned tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) & 15u);
  const unsigned top = tmp & 0xff800000u;
  const double z = (double)__builtin_bit_cast(float, ix - top);
  const double k = (double)((int)top >> 23);
  const double r = __builtin_fma(z, lm_from_bits(W[36 + 2 * i]), -1.0);
  const double y0 = k + lm_from_bits(W[37 + 2 * i]);
  const double p01 = __builtin_fma(r, lm_from_bits(W[68]), lm_from_bits(W[69]));
  const double p23 = __builtin_fma(r, lm_from_bits(W[70]), lm_from_bits(W[71]));
  const double r2 = r * r;
  const double q = __builtin_fma(r, lm_from_bits(W[72]), y0);
  const double logx = __builtin_fma(p01, r2 * r2, __builtin_fma(r2, p23, q));
  const double ylogx = 2.0 * logx;
  const double shift = lm_from_bits(W[32]);
  const double kd = ylogx + shift;
  const unsigned long long ki = (unsigned long long)lm_bits(kd);
  const double rr = ylogx - (kd - shift);
  const double s = lm_from_bits(W[ki & 31u] + (ki << 47));
  const double zz = __builtin_fma(rr, lm_from_bits(W[33]), lm_from_bits(W[34]));
  const double yy = __builtin_fma(rr, lm_from_bits(W[35]), 1.0);
  return (float)(__builtin_fma(zz, rr * rr, yy) * s);
}

// ---- the units of a config with NSG_F_LIBM_EXACT -----------------------------------------------------------------------------------
// NSG_LIBM_EXACT 0 (the precompiled kernels, and every unit of a config without the flag): the kernels' own fdlibm-derived sincos
// (< 1 ulp) and plain products - float32 state within 1e-5 of the reference's until an unstable or chaotic plant has amplified the last
// ulp (Acrobot, a balanced CartPole after ~270 steps: profiles/NOTEBOOK.md).  1: libm's own algorithms with libm's own roundings for
// every sin / cos / scalar power of the integrators and of the θ-engine - what the reference's host evaluates, bit for bit, so that
// the float64 STATE of every classic-control env equals the reference's for as long as one cares to step it.  The price is in
// DESIGN.md section 4 (C1 +4 % per step ... Acrobot x 1.9).  (exp and log1p: nsg_math.hip.h.)
#ifndef NSG_LIBM_EXACT
#define NSG_LIBM_EXACT 0
#endif
#if NSG_LIBM_EXACT
__device__ static const unsigned long long kLibmSincosTab[NSG_SINCOS_TAB_WORDS] = {
#include "nsg_sincos_tab.inc"
};
__device__ static const unsigned long long kLibmPowfTab[NSG_POWF_TAB_WORDS] = {
#include "nsg_powf_tab.inc"
};
__device__ __forceinline__ double env_sin(double x) { return nsg_sin_libm(LibmTab{reinterpret_cast<const double*>(kLibmSincosTab)}, x); }
__device__ __forceinline__ double env_cos(double x) { return nsg_cos_libm(LibmTab{reinterpret_cast<const double*>(kLibmSincosTab)}, x); }
// (Pendulum: lanes in every argument range at once)
__device__ __forceinline__ double env_sin_any(double x) { return nsg_sin_libm_merged<true>(LibmTab{reinterpret_cast<const double*>(kLibmSincosTab)}, x); }
__device__ __forceinline__ double env_cos_any(double x) { return nsg_cos_libm_merged<true>(LibmTab{reinterpret_cast<const double*>(kLibmSincosTab)}, x); }
// `x ** 2` on a float64 / float32 SCALAR is libm's pow / powf in the reference (Acrobot's _dsdt, Pendulum's step), not the product
__device__ __forceinline__ double env_sq(double x) { return nsg_sq_libm(PowTab{kNsgPowTab}, x); }
__device__ __forceinline__ double env_sqf(double x) { return (double)nsg_sqf_libm(PowTab{kLibmPowfTab}, (float)x); }
#else
NSG_HD double env_sq(double x) { return x * x; }
NSG_HD double env_sqf(double x) { return x * x; }
#endif

}  // namespace nsg
