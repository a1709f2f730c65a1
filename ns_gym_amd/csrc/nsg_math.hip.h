// nsg_math.hip.h — float64 sin/cos for the integrators.
//
// The base MDPs call np.sin/np.cos (float64) on bounded arguments: CartPole's pole angle is
// inside ±0.21 rad whenever a step is taken, Acrobot's joint angles are wrapped to [-π, π]
// (RK4 stages stray a little beyond), MountainCar evaluates cos(3x) on x in [-1.2, 0.6],
// Pendulum's angle drifts by at most 8 rad/s * dt per step.  The general-purpose ocml
// sincos carries a Payne-Hanek path for |x| up to 1e308 whose register footprint caps the
// kernel's occupancy; this version keeps fdlibm's kernels (|error| < 1 ulp) with a 3-stage
// Cody-Waite reduction that is exact for |x| < 2^20 * π/2 (≈ 1.6e6 rad) — far beyond anything
// an episode can reach.  Larger arguments still return values in [-1, 1] (reduced accuracy).
#pragma once
#if defined(__HIPCC__)
#define NSG_HD __host__ __device__ __forceinline__
#else
#define NSG_HD inline
#endif
#ifndef __HIPCC_RTC__
#include <math.h>
#endif

#ifndef NSG_SINCOS_FMA
#define NSG_SINCOS_FMA 0           // polynomial form of plain nsg_sincos (nsg_sincos_t<POLY>): 0 fdlibm as written, 1 compiler-fused, 2 SGPR-addend fma
#endif
#ifndef NSG_SINCOS_SHORTCUT
#define NSG_SINCOS_SHORTCUT 1      // 1: wave-uniform test for "no lane needs a reduction" (CartPole's pole angle) ahead of everything else
#endif
#ifndef NSG_SINCOS_FIRST_ROUND
#define NSG_SINCOS_FIRST_ROUND 1   // 1: fdlibm's cheap first reduction round where it is accurate enough (per lane), else always the long one
#endif
#ifndef NSG_SINCOS_STAGES
#define NSG_SINCOS_STAGES 3   // pieces of pi/2 the argument reduction subtracts (2 or 3)
#endif

namespace nsg {

// fma(a, b, C) with C a literal coefficient.  Device, mode 2: ONE VOP3 v_fma_f64 whose addend is an SGPR pair.  (Left to itself the
// compiler turns a Horner step with a constant addend into the two-address v_fmac_f64, which wants the coefficient in the
// destination VGPR pair: it parks all eleven coefficients in VGPRs - +18-29 VGPRs, a wavefront of occupancy - or re-copies them.)
NSG_HD double nsg_fma_coef(double a, double b, double coef) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(coef));
  return r;
#else
  return __builtin_fma(a, b, coef);
#endif
}

// POLY: 0 = fdlibm's kernels as written (a rounding after every multiply and every add), 1 = Horner steps fused by the compiler
// (__builtin_fma), 2 = fused through nsg_fma_coef.  1 and 2 give the same values; all three are < 1 ulp (tests/test_math_cpu.py).
template <int POLY> NSG_HD void nsg_sincos_t(double x, double* sn, double* cs) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
               pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
               pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
               pio2_3t = 8.47842766036889956997e-32;
  // argument reduction: x = n*(π/2) + (y0 + y1), |y0| <= π/4.  π/2 = pio2_1 + pio2_2 + pio2_3 + pio2_3t
  // in 33-bit pieces, so fn*pio2_k is exact for |fn| < 2^20; the two rounded subtractions are
  // compensated with TwoSum error terms (branch-free equivalent of fdlibm's 3-iteration scheme).
  const double fn = rint(x * invpio2);
  // fdlibm's first branch (|x| <= π/4: no reduction), taken when it holds for EVERY lane of the
  // wavefront so the branch is uniform.  CartPole's pole angle never leaves it while an episode runs.
  // With fn == 0 the general path below yields y0 = x, y1 = 0, n = 0 exactly, so both paths agree bit for bit.
#if !NSG_SINCOS_SHORTCUT
  const bool no_reduction = false;
#elif defined(__HIP_DEVICE_COMPILE__)
  const bool no_reduction = __all(fn == 0.0);
#else
  const bool no_reduction = fn == 0.0;
#endif
  double y0, y1;
  int n;
  if (no_reduction) {
    y0 = x;
    y1 = 0.0;
    n = 0;
  } else {
    const double r1 = x - fn * pio2_1;  // exact (Sterbenz)
#if NSG_SINCOS_FIRST_ROUND
    // fdlibm's first round: pi/2 ~ pio2_1 + pio2_1t (86 bits).  (y0, y1) is r1 - fl(fn * pio2_1t) as an exact double-double; what
    // it leaves out - the rounding of the product and of pio2_1t itself - is below |fn| * 1.4e-26, i.e. below 2^-70 of the
    // reduced argument as long as no more than 16 leading bits cancelled (|y0| >= 2^-16 |x|): 2^-17 ulp on top of the kernels'
    // own error.  Lanes closer than that to a multiple of pi/2 (one evaluation in ~2^15) take the compensated three-piece
    // reduction below; the choice is PER LANE (a lane's result never depends on its neighbours), the branch around the long
    // path is wave-uniform.  11 instructions instead of 25 on the path every Acrobot / Pendulum / MountainCar evaluation takes.
    const double w1 = fn * pio2_1t;
    y0 = r1 - w1;
    y1 = (r1 - y0) - w1;
    const bool first_round_ok = fabs(y0) * 65536.0 >= fabs(x);
#if defined(__HIP_DEVICE_COMPILE__)
    if (!__all(first_round_ok))
#else
    if (!first_round_ok)
#endif
#endif
    {
      const double c2 = -(fn * pio2_2);
      const double r2 = r1 + c2;
      const double b2 = r2 - r1;
      const double e2 = (r1 - (r2 - b2)) + (c2 - b2);
#if NSG_SINCOS_STAGES == 2
      // pi/2 = pio2_1 + pio2_2 + pio2_2t to 119 bits: the reduced argument is off by |x| * 2^-119, i.e. by 2^(c-119) of itself
      // when c leading bits cancel; no double below 2^20 * pi/2 cancels more than ~62 bits against a multiple of pi/2, so the
      // third piece (fdlibm's third iteration) could only ever move the result by < 2^-57 of itself - 1/16 ulp.
      const double tail = e2 - fn * pio2_2t;
      const double z0 = r2 + tail;
      const double z1 = (r2 - z0) + tail;
#else
      const double c3 = -(fn * pio2_3);
      const double r3 = r2 + c3;
      const double b3 = r3 - r2;
      const double e3 = (r2 - (r3 - b3)) + (c3 - b3);
      const double tail = (e2 + e3) - fn * pio2_3t;
      const double z0 = r3 + tail;
      const double z1 = (r3 - z0) + tail;
#endif
#if NSG_SINCOS_FIRST_ROUND
      y0 = first_round_ok ? y0 : z0;
      y1 = first_round_ok ? y1 : z1;
#else
      y0 = z0;
      y1 = z1;
#endif
    }
    n = (int)fn;
  }
  (void)pio2_1t; (void)pio2_2t; (void)pio2_3; (void)pio2_3t;
  // fdlibm __kernel_sin / __kernel_cos on (y0, y1)
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double z = y0 * y0;
  const double v = z * y0;
  double s, c;
  if constexpr (POLY == 1) {
    // the same polynomials, Horner steps fused (one rounding per step instead of two: never less accurate); the two
    // kernels together are 17 instructions instead of 35
    const double rs = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, S5), S4), S3), S2);
    s = y0 - __builtin_fma(-v, S1, __builtin_fma(z, __builtin_fma(-v, rs, 0.5 * y1), -y1));
    const double rc = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z;
    const double wc = 1.0 - hz;
    c = wc + (((1.0 - wc) - hz) + __builtin_fma(z, rc, -(y0 * y1)));
  } else if constexpr (POLY == 2) {
    const double rs = nsg_fma_coef(z, nsg_fma_coef(z, nsg_fma_coef(z, nsg_fma_coef(z, S6, S5), S4), S3), S2);
    s = y0 - __builtin_fma(-v, S1, __builtin_fma(z, __builtin_fma(-v, rs, 0.5 * y1), -y1));
    const double rc = z * nsg_fma_coef(z, nsg_fma_coef(z, nsg_fma_coef(z, nsg_fma_coef(z, nsg_fma_coef(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z;
    const double wc = 1.0 - hz;
    c = wc + (((1.0 - wc) - hz) + __builtin_fma(z, rc, -(y0 * y1)));
  } else {
    const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    s = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z;
    const double wc = 1.0 - hz;
    c = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));
  }
  switch (n & 3) {
    case 0: *sn = s; *cs = c; break;
    case 1: *sn = c; *cs = -s; break;
    case 2: *sn = -s; *cs = -c; break;
    default: *sn = -c; *cs = s; break;
  }
}

// Python's float `x % m` for m > 0 (floor-mod: fmod, then + m when the remainder is negative), exactly, without
// fmod's long-division loop.  q = floor(x / m) is the true floored quotient or one too large (rounding to nearest is
// monotonic, so it never falls below an integer the true quotient reaches); x - q*m then lies in (-m, m), is a multiple
// of the operands' finer ulp and so fits a double: the fused multiply-add returns it EXACTLY, and the one conditional
// + m (exact too: the sum is the true remainder) finishes.  Valid for |x / m| < 2^52.  tests/test_math_cpu.py checks it
// against Python's own % bit for bit.
NSG_HD double nsg_pymod_pos(double x, double m) {
  const double q = floor(x / m);
  double r = __builtin_fma(-q, m, x);
  if (r < 0.0) r += m;
  return r;
}

// gymnasium's wrap(x, -pi, pi) [UPSTREAM acrobot.py wrap]: `while x > M: x = x - diff`, `while x < m: x = x + diff` with diff = 2 pi - ONE
// ROUNDED subtraction per turn, so the result is not fmod's and depends on the turns taken.  A healthy Acrobot step needs at most
// two; a step whose RK4 stages blew up (C4's LINK_MASS_2 growing: one env in 262 144 at step 114 came out at 5131 rad) needs as
// many as the reference takes - 817 there, 1.6e8 at 1e9 rad.  Taking them one by one would let one lane hold its wavefront for
// milliseconds, so the turns of a whole binade are taken at once, EXACTLY:
//   in [2^e, 2^(e+1)) every double is a multiple of u = 2^(e-52); for a on that grid with a - D >= 2^e the rounded difference is
//   a - RN_u(D) (round-to-nearest is translation invariant on the grid; a tie needs D mod u = u/2, which for D = 0x401921FB54442D18
//   - lowest set bit 2^-47 - happens in [64, 128) only), so n turns inside one binade subtract n * RN_u(D): a product and a
//   difference of grid multiples below 2^(e+1), both exact.  The last three turns of a binade, the crossing into the next and
//   everything below 128 are taken as the reference takes them.
// From 2^56 on, a - D == a: the reference's loop never returns; this one returns x as it came (so does it for inf; NaN fails the
// loop's test in both).  tests/test_math_cpu.py compares it with the loop itself.
NSG_HD double nsg_wrap_pi(double x) {
  const double PI = 3.141592653589793, D = PI - -PI;
  double a = __builtin_fabs(x);
  if (!(a > PI)) return x;
  if (!(a < 72057594037927936.0)) return x;   // 2^56
  while (a >= 128.0) {     // one binade per pass, at most 49 passes
    const double m = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, a) & 0xfff0000000000000ULL);   // 2^e
    const double step = (m + D) - m;          // RN_u(D)
    const double n = floor((a - m) / step) - 3.0;
    if (n > 0.0) a = a - n * step;
    while (a >= m) a = a - D;                 // <= 5 turns: leaves the binade the way the reference does
  }
  while (a > PI) a = a - D;                   // <= 21 turns
  return x < 0.0 ? -a : a;
}

NSG_HD void nsg_sincos(double x, double* sn, double* cs) { nsg_sincos_t<NSG_SINCOS_FMA>(x, sn, cs); }
NSG_HD double nsg_sin(double x) { double s, c; nsg_sincos(x, &s, &c); return s; }
NSG_HD double nsg_cos(double x) { double s, c; nsg_sincos(x, &s, &c); return c; }
template <int POLY> NSG_HD double nsg_cos_t(double x) { double s, c; nsg_sincos_t<POLY>(x, &s, &c); return c; }
template <int POLY> NSG_HD double nsg_sin_t(double x) { double s, c; nsg_sincos_t<POLY>(x, &s, &c); return s; }

}  // namespace nsg

namespace nsg {

// ---- float64 exp / log1p ------------------------------------------------------------------------------------------------------
// Used by the "full" θ-engine: ExponentialDecay / SigmoidTransition (np.exp, single_param.py:286, 384) and by NumPy's samplers,
// which run in C and call libm (the ziggurat's wedge tests: exp; its tails and geometric: log1p).  The device library's versions cost
// ~60 VGPRs more, which costs the RandomWalk kernels a wave of occupancy.
//   nsg_log1p_libm glibc 2.35's log1p [UPSTREAM sysdeps/ieee754/dbl-64/s_log1p.c] operation for operation: the normal variates of a
//                  ziggurat tail and the geometric waits equal NumPy's in every bit.  What nsg_log1p IS in a unit built with
//                  NSG_LIBM_EXACT; elsewhere it is fdlibm's original (why: at the function).
//   nsg_exp_libm   glibc 2.35's exp [UPSTREAM e_exp.c, __exp_fma, from ARM's optimized-routines: 128-entry table, degree-5
//                  polynomial] with every fused multiply-add where the image's libm.so.6 has one.  The table is __exp_data, extracted
//                  from the image's libm by tools/extract_libm_pow_tables.py (words 401..656 of include/nsg_pow_tab.inc; the log
//                  side of the same file serves nsg_sq_libm).  What nsg_exp IS in a unit built with NSG_LIBM_EXACT.
//   nsg_exp_fdlibm fdlibm's exp (< 1 ulp), what nsg_exp is everywhere else (why: below).
// tests/test_math_cpu.py holds nsg_log1p_libm and nsg_exp_libm equal to libm itself, subnormal results and special values included.
// (np.exp on a scalar or an array is NumPy's own SIMD kernel on an AVX-512 host, 1 ulp off libm's for 4.6 % of arguments: an
// ExponentialDecay theta is host-dependent in the reference itself.  libm's is what the oracle - plain C - calls, and what a host
// without AVX-512 gets.)
#define NSG_POW_TAB_WORDS 657
#if defined(__HIP_DEVICE_COMPILE__)
__device__ static const unsigned long long kNsgPowTab[NSG_POW_TAB_WORDS] = {
#else
static const unsigned long long kNsgPowTab[NSG_POW_TAB_WORDS] = {
#endif
#include "nsg_pow_tab.inc"
};

NSG_HD double nsg_exp_libm(double x) {
  // __exp_data's eight scalars as literals (words 9..16 of the table file; tests/test_math_cpu.py asserts they ARE those words): in
  // registers they are scalar operands, as loads they were vector registers the RandomWalk kernels do not have to spare
  const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p+52, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47,
               C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
  const unsigned long long bx = __builtin_bit_cast(unsigned long long, x);
  unsigned abstop = (unsigned)(bx >> 52) & 0x7ffu;
  if (abstop - 0x3c9u >= 0x3fu) {
    if ((int)(abstop - 0x3c9u) < 0) return 1.0 + x;                       // |x| < 2^-54
    if (abstop >= 0x409u) {                                                 // |x| >= 1024, inf, NaN
      if (bx == 0xfff0000000000000ULL) return 0.0;
      if (abstop >= 0x7ffu) return 1.0 + x;
      return (bx >> 63) ? 0.0 : __builtin_inf();
    }
    abstop = 0;                                                             // 512 <= |x| < 1024: the scale may leave the normal range
  }
  const double kk = __builtin_fma(x, InvLn2N, Shift);
  const unsigned ki = (unsigned)__builtin_bit_cast(unsigned long long, kk);   // (the low word holds every bit that is used)
  const double kn = kk - Shift;
  const double r = __builtin_fma(kn, NegLn2loN, __builtin_fma(kn, NegLn2hiN, x));
  const unsigned long long* T = kNsgPowTab + 401 + 2 * (ki & 127u);
  const unsigned long long sbits = T[1] + ((unsigned long long)ki << 45);
  const double tr = r + __builtin_bit_cast(double, T[0]);
  const double r2 = r * r;
  const double tmp = __builtin_fma(r2 * r2, __builtin_fma(r, C5, C4), __builtin_fma(__builtin_fma(r, C3, C2), r2, tr));
  if (abstop == 0) {                                                        // specialcase()
    if ((ki & 0x80000000u) == 0) {                                          // k > 0: the exponent of scale may have overflowed
      const double scale = __builtin_bit_cast(double, sbits - (1009ULL << 52));
      return 0x1p1009 * __builtin_fma(scale, tmp, scale);
    }
    const double scale = __builtin_bit_cast(double, sbits + (1022ULL << 52));   // k < 0: round once, in the subnormal range
    const double st = scale * tmp;
    double y = scale + st;
    if (y < 1.0) {
      double lo = (scale - y) + st;
      const double hi = 1.0 + y;
      lo = ((1.0 - hi) + y) + lo;
      y = (hi + lo) - 1.0;
      if (y == 0.0) y = 0.0;
    }
    return 0x1p-1022 * y;
  }
  const double scale = __builtin_bit_cast(double, sbits);
  return __builtin_fma(scale, tmp, scale);
}

// fdlibm's exp (< 1 ulp; no table, 4-8 vector registers less than the one above inside the RandomWalk kernels - a wavefront of
// occupancy for C2's fused rollout: 90 -> 98 VGPRs measured).  What every build WITHOUT NSG_LIBM_EXACT evaluates: there exp decides the
// ziggurat's wedge test (an ulp of exp changes an accept / reject once in ~1e16 draws) and feeds ExponentialDecay / SigmoidTransition
// thetas, compared at 1e-12.  A unit built with NSG_LIBM_EXACT evaluates libm's.
NSG_HD double nsg_ldexp_norm(double y, int k) {  // y * 2^k for the k range exp() produces
  // two-step scaling keeps subnormal results correctly rounded once
  union { double d; unsigned long long u; } a;
  if (k > 1023) {
    a.u = 0x7fe0000000000000ULL;  // 2^1023
    y *= a.d;
    k -= 1023;
    if (k > 1023) k = 1023;
  } else if (k < -1022) {
    a.u = 0x0360000000000000ULL;  // 2^-969
    y *= a.d;
    k += 969;
    if (k < -1022) k = -1022;
  }
  a.u = (unsigned long long)(k + 1023) << 52;
  return y * a.d;
}

NSG_HD double nsg_exp_fdlibm(double x) {
  const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
               P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 7.09782712893383973096e+02) return __builtin_inf();
  if (x < -7.45133219101941108420e+02) return 0.0;
  const double kf = rint(x * invln2);
  const int k = (int)kf;
  const double hi = x - kf * ln2HI;
  const double lo = kf * ln2LO;
  const double r = hi - lo;
  const double t = r * r;
  const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  return k == 0 ? y : nsg_ldexp_norm(y, k);
}

#ifndef NSG_LIBM_EXACT
#define NSG_LIBM_EXACT 0
#endif
NSG_HD double nsg_exp(double x) {
#if NSG_LIBM_EXACT
  return nsg_exp_libm(x);
#else
  return nsg_exp_fdlibm(x);
#endif
}

NSG_HD double nsg_log1p_libm(double x) {
  // glibc 2.35's log1p [UPSTREAM sysdeps/ieee754/dbl-64/s_log1p.c: fdlibm's algorithm with the polynomial split in four] operation for
  // operation - the function NumPy's generators call (ziggurat tails, geometric) - so that an exact unit's streams equal NumPy's in every bit:
  // branch tests on the HIGH WORD as upstream has them (they are not the double comparisons they approximate), the |x| < 2^-29 and
  // |f| < 2^-20 shortcuts, R = ((z Lp1 + z^2 (Lp2 + z Lp3)) + z^4 (Lp4 + z Lp5)) + z^6 (Lp6 + z Lp7).  The x86-64 build has no FMA
  // variant of this function.  tests/test_math_cpu.py: equal to libm on 6 M arguments over every branch.
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
               Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  const unsigned long long bx = __builtin_bit_cast(unsigned long long, x);
  const int hx = (int)(bx >> 32), ax = hx & 0x7fffffff;
  int k = 1, hu = 0;
  double f = x, c = 0.0;
  if (hx < 0x3FDA827A) {                       // x < 0.41422
    if (ax >= 0x3ff00000) return x == -1.0 ? -__builtin_inf() : __builtin_nan("");
    if (ax < 0x3e200000) return ax < 0x3c900000 ? x : x - x * x * 0.5;      // |x| < 2^-29 (2^-54)
    if (hx > 0 || hx < (int)0xbfd2bec4) { k = 0; hu = 1; }                    // sqrt(2)/2 - 1 < x < sqrt(2) - 1, by high word
  } else if (hx >= 0x7ff00000) {
    return x + x;
  }
  if (k != 0) {
    double u;
    if (hx < 0x43400000) {
      u = 1.0 + x;
      hu = (int)(__builtin_bit_cast(unsigned long long, u) >> 32);
      k = (hu >> 20) - 1023;
      c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);   // correction term for the rounding of 1 + x
      c /= u;
    } else {
      u = x;
      hu = hx;
      k = (hu >> 20) - 1023;
    }
    hu &= 0x000fffff;
    const unsigned long long lo = __builtin_bit_cast(unsigned long long, u) & 0xffffffffULL;
    if (hu < 0x6a09e) {
      u = __builtin_bit_cast(double, ((unsigned long long)(unsigned)(hu | 0x3ff00000) << 32) | lo);   // normalise u
    } else {
      k += 1;
      u = __builtin_bit_cast(double, ((unsigned long long)(unsigned)(hu | 0x3fe00000) << 32) | lo);   // normalise u/2
      hu = (0x00100000 - hu) >> 2;
    }
    f = u - 1.0;
  }
  const double hfsq = 0.5 * f * f;
  const double kd = (double)k;
  if (hu == 0) {                               // |f| < 2^-20
    if (f == 0.0) return k == 0 ? 0.0 : kd * ln2_hi + (c + kd * ln2_lo);
    const double R = hfsq * (1.0 - 0.66666666666666666 * f);
    return k == 0 ? f - R : kd * ln2_hi - ((R - (kd * ln2_lo + c)) - f);
  }
  const double s = f / (2.0 + f);
  const double z = s * s, z2 = z * z, z4 = z2 * z2, z6 = z4 * z2;
  const double R = ((z * Lp1 + z2 * (Lp2 + z * Lp3)) + z4 * (Lp4 + z * Lp5)) + z6 * (Lp6 + z * Lp7);
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  return kd * ln2_hi - ((hfsq - (s * (hfsq + R) + (kd * ln2_lo + c))) - f);
}

// fdlibm's log1p as published (< 1 ulp; equal to glibc's for all but 0.08 % of arguments - glibc tests the high word where this
// compares doubles, has two more shortcuts and splits the polynomial).  What every build WITHOUT NSG_LIBM_EXACT evaluates: log1p only
// runs in a ziggurat tail (2.6e-4 of the normal draws) and in geometric, and glibc's form costs C2's step kernel 14 more spilled SGPRs
// - 4 % at BASELINE's 65 536 envs, where a launch is latency-bound (6.77 -> 7.04 us, profiles/r04_ab_log1p.txt).
NSG_HD double nsg_log1p_fdlibm(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
               Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  if (x != x) return x;
  if (x < -1.0) return __builtin_nan("");
  if (x == -1.0) return -__builtin_inf();
  if (x == __builtin_inf()) return x;
  const double ax = fabs(x);
  if (ax < 5.55111512312578270212e-17) return x;  // |x| < 2^-54
  int k = 1;
  double f = x, c = 0.0;
  if (x > -0.2928932188134524 && x < 0.41421356237309515) {
    k = 0;  // sqrt(2)/2 - 1 < x < sqrt(2) - 1: no reduction
  } else {
    union { double d; unsigned long long u; } uu;
    const double u1 = 1.0 + x;
    uu.d = u1;
    k = (int)((uu.u >> 52) & 0x7ff) - 1023;
    // correction term for the rounding of 1 + x
    c = (k > 0) ? 1.0 - (u1 - x) : x - (u1 - 1.0);
    c /= u1;
    unsigned long long m = uu.u & 0x000fffffffffffffULL;
    if (m < 0x6a09e667f3bcdULL) {
      uu.u = m | 0x3ff0000000000000ULL;  // normalise u
    } else {
      k += 1;
      uu.u = m | 0x3fe0000000000000ULL;  // normalise u/2
    }
    f = uu.d - 1.0;
  }
  const double hfsq = 0.5 * f * f;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  const double kd = (double)k;
  return kd * ln2_hi - ((hfsq - (s * (hfsq + R) + (kd * ln2_lo + c))) - f);
}

NSG_HD double nsg_log1p(double x) {
#if NSG_LIBM_EXACT
  return nsg_log1p_libm(x);
#else
  return nsg_log1p_fdlibm(x);
#endif
}

}  // namespace nsg
