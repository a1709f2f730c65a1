// nsg_envs.hip.h — per-env-type transition dynamics, reset draws, observations and the
// wrapper's constraint checker, in device code.
//
// Base MDPs: gymnasium 1.2.1 [UPSTREAM, uv.lock:958-959 — not in the reference tree]; the
// reference calls them at ns_gym/base.py:313 (step) and :377 (reset).  The CartPole block is
// corroborated in-tree by rats-experiments/code/envs/nscartpole_v0.py:24-36,92-108.
// Constraint checker and dependency resolver: ns_gym/wrappers/classic_control.py:193-458.
//
// Everything is float64 in the reference's operation order (obs cast to float32 at the end,
// like gymnasium); θ arrives as a fixed-size register array indexed only by compile-time
// constants (slot order = ATTRIBUTE_MAP, ns_gym/base.py:611-635).
#pragma once
#include "nsgym_hip.h"
#include "nsg_math.hip.h"
#include "nsg_libm.hip.h"
#include "nsg_rng.hip.h"

namespace nsg {

#define NSG_PI 3.141592653589793238462643383279502884

// Polynomial form of each env type's sincos (nsg_math.hip.h, nsg_sincos_t<POLY>).  Acrobot is bound by float64 vector-ALU issue
// (15 sincos per step): the fused form with SGPR addends (2) takes its step from 59.6 to 55.0 us at 2^20 envs, its fused rollout
// from 50.4 to 44.6 us per step, C4's mixed launch from 23.4 to 21.7 us, at unchanged occupancy.  Pendulum and the MountainCars
// gain 1-3 %.  CartPole keeps fdlibm's form as written (0): its kernels sit on the 80-VGPR / 6-wavefront boundary the launch policy
// is built on, the fused form costs them 4 VGPRs (the generic fused rollout would spill) and buys 0.7 %.
#ifndef NSG_ACROBOT_SINCOS_POLY
#define NSG_ACROBOT_SINCOS_POLY 2
#endif
#ifndef NSG_LIGHT_SINCOS_POLY
#define NSG_LIGHT_SINCOS_POLY 2     // Pendulum, MountainCar, MountainCarContinuous
#endif
constexpr int kAcroPoly = NSG_ACROBOT_SINCOS_POLY;
constexpr int kLightPoly = NSG_LIGHT_SINCOS_POLY;


template <int ENV> struct EnvTraits;
// RETURN_PER_STEP: env types whose reward is the same constant on EVERY step (CartPole-v1: 1.0 incl. the terminating
// step; MountainCar-v0: -1.0) - their episode return is RETURN_PER_STEP * t exactly (float32 sums of +-1 are exact far
// beyond any TimeLimit), so the kernels derive last_return from t instead of round-tripping a running-return row.
#ifndef NSG_CARTPOLE_INLANE   // measurement knob: 1 = CartPole resets in-lane too (no hand-over, no barriers)
#define NSG_CARTPOLE_INLANE 0
#endif
template <> struct EnvTraits<NSG_ENV_CARTPOLE> { static constexpr bool RESET_IN_LANE = NSG_CARTPOLE_INLANE != 0; static constexpr int RESET_DRAWS = 4, PHYS = 4, OBS = 4, NTHETA = 6, NDERIVED = 2; static constexpr bool FLOAT_ACT = false; static constexpr float RETURN_PER_STEP = 1.f; static constexpr bool NEVER_TERMINATES = false; };
template <> struct EnvTraits<NSG_ENV_PENDULUM> { static constexpr bool RESET_IN_LANE = true; static constexpr int RESET_DRAWS = 2, NDERIVED = 0, PHYS = 2, OBS = 3, NTHETA = 4; static constexpr bool FLOAT_ACT = true; static constexpr float RETURN_PER_STEP = 0.f; static constexpr bool NEVER_TERMINATES = true; };
template <> struct EnvTraits<NSG_ENV_ACROBOT> { static constexpr bool RESET_IN_LANE = true; static constexpr int RESET_DRAWS = 4, NDERIVED = 0, PHYS = 4, OBS = 6, NTHETA = 8; static constexpr bool FLOAT_ACT = false; static constexpr float RETURN_PER_STEP = 0.f; static constexpr bool NEVER_TERMINATES = false; };
template <> struct EnvTraits<NSG_ENV_MOUNTAINCAR> { static constexpr bool RESET_IN_LANE = true; static constexpr int RESET_DRAWS = 1, NDERIVED = 0, PHYS = 2, OBS = 2, NTHETA = 2; static constexpr bool FLOAT_ACT = false; static constexpr float RETURN_PER_STEP = -1.f; static constexpr bool NEVER_TERMINATES = false; };
template <> struct EnvTraits<NSG_ENV_MOUNTAINCAR_CONT> { static constexpr bool RESET_IN_LANE = true; static constexpr int RESET_DRAWS = 1, NDERIVED = 0, PHYS = 2, OBS = 2, NTHETA = 1; static constexpr bool FLOAT_ACT = true; static constexpr float RETURN_PER_STEP = 0.f; static constexpr bool NEVER_TERMINATES = false; };

// ---- reset draws: np_random.uniform(low, high) = low + (high - low) * next_double ----------
// EnvTraits::RESET_DRAWS = how many doubles a reset takes from the env's stream (the kernels jump to draw RESET_DRAWS * episode)
// EnvTraits::NEVER_TERMINATES: the base MDP's step never returns terminated (Pendulum): episodes end by TimeLimit only (step_chunk)
// EnvTraits::RESET_IN_LANE: how the step kernels run the reset path (seed -> jump-ahead -> draws, ~550 instructions).  CartPole
// under random actions ends ~5 % of its episodes per step, i.e. 96 % of the wavefronts hold a resetting lane: the resets are
// compacted per workgroup (two barriers, ONE wavefront runs the path).  The other env types end 0.5-1 % per step (TimeLimit
// 200-999, rare terminations): half or more of the wavefronts hold no resetting lane at all, so each lane runs its own reset and
// the step has no barrier - measured at 2^20 envs: Acrobot 58.5 -> see DESIGN.md, C4's mixed launch likewise.
// Component k of the initial state from the k-th uniform double of reset() [UPSTREAM]; k < RESET_DRAWS.
template <int ENV> __device__ __forceinline__ double env_reset_map(int k, double u) {
  if constexpr (ENV == NSG_ENV_CARTPOLE) {
    return -0.05 + (0.05 - -0.05) * u;
  } else if constexpr (ENV == NSG_ENV_PENDULUM) {
    return k == 0 ? -NSG_PI + (NSG_PI - -NSG_PI) * u : -1.0 + (1.0 - -1.0) * u;
  } else if constexpr (ENV == NSG_ENV_ACROBOT) {  // .astype(np.float32) upstream
    return (double)(float)(-0.1 + (0.1 - -0.1) * u);
  } else {  // MountainCar / MountainCarContinuous: [uniform(-0.6, -0.4), 0]
    return -0.6 + (-0.4 - -0.6) * u;
  }
}
// READY: `g` already stands AFTER the step of its next draw (pcg_at<2>): the first uniform is the output of the state as it is.
template <int ENV, bool READY = false> __device__ __forceinline__ void env_reset_draw(Pcg& g, double* s) {
  using T = EnvTraits<ENV>;
#pragma unroll
  for (int k = 0; k < T::PHYS; k++) {
    if (k < T::RESET_DRAWS) s[k] = env_reset_map<ENV>(k, (READY && k == 0) ? pcg_double_out(g) : pcg_double(g));
    else s[k] = 0.0;
  }
}

// ---- observation (float32, like gymnasium's _get_obs) -------------------------------------
template <int ENV> __device__ __forceinline__ void env_obs(const double* s, float* o) {
  if constexpr (ENV == NSG_ENV_CARTPOLE) {
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = (float)s[k];
  } else if constexpr (ENV == NSG_ENV_PENDULUM) {
    double sn, cs;
#if NSG_LIBM_EXACT
    cs = env_cos_any(s[0]); sn = env_sin_any(s[0]);
#else
    nsg_sincos_t<kLightPoly>(s[0], &sn, &cs);
#endif
    o[0] = (float)cs; o[1] = (float)sn; o[2] = (float)s[1];
  } else if constexpr (ENV == NSG_ENV_ACROBOT) {
    double s0, c0, s1, c1;
#if NSG_LIBM_EXACT
    c0 = env_cos(s[0]); s0 = env_sin(s[0]); c1 = env_cos(s[1]); s1 = env_sin(s[1]);
#else
    nsg_sincos_t<kAcroPoly>(s[0], &s0, &c0);
    nsg_sincos_t<kAcroPoly>(s[1], &s1, &c1);
#endif
    o[0] = (float)c0; o[1] = (float)s0; o[2] = (float)c1; o[3] = (float)s1; o[4] = (float)s[2]; o[5] = (float)s[3];
  } else {
    o[0] = (float)s[0]; o[1] = (float)s[1];
  }
}

// ---- constraint checker (classic_control.py:193-422) --------------------------------------
// nv: proposed value per θ slot, cur: pre-update attribute, tuned: bit mask of slots that are
// in tunable_params (wave-uniform).  Returns the bit mask of rejected slots.
template <int ENV> __device__ __forceinline__ unsigned constraint_mask(const double* nv, const double* cur, unsigned tuned) {
  unsigned v = 0;
  if constexpr (ENV == NSG_ENV_CARTPOLE) {  // gravity masscart masspole force_mag tau length
    if ((tuned & 1u) && nv[0] < 0) v |= 1u;
    if ((tuned & 2u) && nv[1] <= 0) v |= 2u;
    if ((tuned & 4u) && nv[2] <= 0) v |= 4u;
    if ((tuned & 32u) && nv[5] <= 0) v |= 32u;
  } else if constexpr (ENV == NSG_ENV_PENDULUM) {  // m l dt g
    if ((tuned & 1u) && nv[0] <= 0) v |= 1u;
    if ((tuned & 2u) && nv[1] <= 0) v |= 2u;
    if ((tuned & 4u) && nv[2] <= 0) v |= 4u;
    if ((tuned & 8u) && nv[3] < 0) v |= 8u;
  } else if constexpr (ENV == NSG_ENV_ACROBOT) {  // dt L1 L2 M1 M2 C1 C2 MOI
    if (tuned & 2u) {
      if (nv[1] <= 0) v |= 2u;
      else if ((tuned & 32u) && nv[5] > nv[1]) v |= 2u;
      else if (nv[1] < cur[5]) v |= 2u;
    }
    if ((tuned & 4u) && nv[2] <= 0) v |= 4u;  // the COM cross-checks of LINK_LENGTH_2 are dead code (:267)
    if ((tuned & 8u) && nv[3] <= 0) v |= 8u;
    if ((tuned & 16u) && nv[4] <= 0) v |= 16u;
    if (tuned & 32u) {
      if (nv[5] <= 0) v |= 32u;
      else if ((tuned & 2u) && nv[1] < nv[5]) v |= 32u;
      else if (nv[5] > cur[1]) v |= 32u;
    }
    if (tuned & 64u) {
      if (nv[6] <= 0) v |= 64u;
      else if ((tuned & 4u) && nv[2] < nv[6]) v |= 64u;
      else if (nv[6] > cur[2]) v |= 64u;
    }
  } else if constexpr (ENV == NSG_ENV_MOUNTAINCAR) {  // gravity force
    if ((tuned & 1u) && nv[0] <= 0) v |= 1u;
    if ((tuned & 2u) && nv[1] <= 0) v |= 2u;
  } else {  // power
    if ((tuned & 1u) && nv[0] <= 0) v |= 1u;
  }
  return v;
}

// ---- transitions ----------------------------------------------------------------------------
// psq: lc1 ** 2, l1 ** 2, lc2 ** 2 - the same three scalar powers in each of RK4's four stages, taken once per step
__device__ __forceinline__ void acrobot_dsdt(const double* th, const double* psq, const double* y, double a, double* d) {
  const double m1 = th[3], m2 = th[4], l1 = th[1], lc1 = th[5], lc2 = th[6], I1 = th[7], I2 = th[7], g = 9.8;
  const double theta1 = y[0], theta2 = y[1], dtheta1 = y[2], dtheta2 = y[3];
  double sin2, cos2;
#if NSG_LIBM_EXACT
  cos2 = env_cos(theta2); sin2 = env_sin(theta2);
  const double cos12 = env_cos(theta1 + theta2 - NSG_PI / 2.0), cos1 = env_cos(theta1 - NSG_PI / 2);
#else
  nsg_sincos_t<kAcroPoly>(theta2, &sin2, &cos2);
  const double cos12 = nsg_cos_t<kAcroPoly>(theta1 + theta2 - NSG_PI / 2.0), cos1 = nsg_cos_t<kAcroPoly>(theta1 - NSG_PI / 2);
#endif
  const double lc2sq = psq[2];
  double d1 = m1 * psq[0] + m2 * (psq[1] + lc2sq + 2 * l1 * lc2 * cos2) + I1 + I2;
  double d2 = m2 * (lc2sq + l1 * lc2 * cos2) + I2;
  double phi2 = m2 * lc2 * g * cos12;
  double phi1 = -m2 * l1 * lc2 * env_sq(dtheta2) * sin2 - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * sin2 +
                (m1 * lc1 + m2 * l1) * g * cos1 + phi2;
  double ddtheta2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * env_sq(dtheta1) * sin2 - phi2) /
                    (m2 * lc2sq + I2 - env_sq(d2) / d1);
  double ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;
  d[0] = dtheta1; d[1] = dtheta2; d[2] = ddtheta1; d[3] = ddtheta2;
}

// Advances s in place with parameters th (post-update θ), returns terminated, writes reward.
template <int ENV>
__device__ __forceinline__ bool env_step(const double* th, double* s, int ai, float af, double& reward) {
  if constexpr (ENV == NSG_ENV_CARTPOLE) {
    const double gravity = th[0], masspole = th[2], force_mag = th[3], tau = th[4], length = th[5];
    // th[6] = total_mass, th[7] = polemass_length: resolved by the caller (_dependency_resolver,
    // classic_control.py:424-444); a frozen planning copy carries the values resolved at fork time
    const double total_mass = th[6], polemass_length = th[7];
    double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
    double force = ai == 1 ? force_mag : -force_mag;
    double sintheta, costheta;
#if NSG_LIBM_EXACT
    costheta = env_cos(theta); sintheta = env_sin(theta);
#else
    nsg_sincos(theta, &sintheta, &costheta);
#endif
    double temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass;
    double thetaacc = (gravity * sintheta - costheta * temp) /
                      (length * (4.0 / 3.0 - masspole * (costheta * costheta) / total_mass));
    double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
    x = x + tau * x_dot;
    x_dot = x_dot + tau * xacc;
    theta = theta + tau * theta_dot;
    theta_dot = theta_dot + tau * thetaacc;
    s[0] = x; s[1] = x_dot; s[2] = theta; s[3] = theta_dot;
    const double thr = 12 * 2 * NSG_PI / 360;
    reward = 1.0;
    return x < -2.4 || x > 2.4 || theta < -thr || theta > thr;
  } else if constexpr (ENV == NSG_ENV_PENDULUM) {
    const double m = th[0], l = th[1], dt = th[2], g = th[3];
    double t0 = s[0], thdot = s[1];
    double u = (double)af;
    if (u < -2.0) u = -2.0;
    if (u > 2.0) u = 2.0;
    const double an = nsg_pymod_pos(t0 + NSG_PI, 2 * NSG_PI) - NSG_PI;  // angle_normalize: ((x + pi) % (2 pi)) - pi [UPSTREAM]
    double costs = env_sq(an) + 0.1 * env_sq(thdot) + 0.001 * env_sqf(u);   // `u ** 2`: a float32 scalar power upstream
#if NSG_LIBM_EXACT
    const double sin_t0 = env_sin_any(t0);
#else
    const double sin_t0 = nsg_sin_t<kLightPoly>(t0);
#endif
    double newthdot = thdot + (3 * g / (2 * l) * sin_t0 + 3.0 / (m * env_sq(l)) * u) * dt;
    if (newthdot < -8.0) newthdot = -8.0;
    if (newthdot > 8.0) newthdot = 8.0;
    double newth = t0 + newthdot * dt;
    s[0] = newth; s[1] = newthdot;
    reward = -costs;
    return false;
  } else if constexpr (ENV == NSG_ENV_ACROBOT) {
    const double a = (double)(ai - 1);  // AVAIL_TORQUE = [-1, 0, +1]
    const double dt = th[0] - 0.0, dt2 = dt / 2.0;
    double y0[4] = {s[0], s[1], s[2], s[3]}, k1[4], k2[4], k3[4], k4[4], y[4];
    const double psq[3] = {env_sq(th[5]), env_sq(th[1]), env_sq(th[6])};
    acrobot_dsdt(th, psq, y0, a, k1);
#pragma unroll
    for (int k = 0; k < 4; k++) y[k] = y0[k] + dt2 * k1[k];
    acrobot_dsdt(th, psq, y, a, k2);   // the torque component has derivative 0.0: a + dt2 * 0.0 == a
#pragma unroll
    for (int k = 0; k < 4; k++) y[k] = y0[k] + dt2 * k2[k];
    acrobot_dsdt(th, psq, y, a, k3);
#pragma unroll
    for (int k = 0; k < 4; k++) y[k] = y0[k] + dt * k3[k];
    acrobot_dsdt(th, psq, y, a, k4);
    double ns[4];
#pragma unroll
    for (int k = 0; k < 4; k++) ns[k] = y0[k] + dt / 6.0 * (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]);
#pragma unroll
    // wrap(x, -pi, pi) [UPSTREAM]: rounded turn by turn (nsg_math.hip.h).  Against round 3's two loops cut at 64 turns, same box: step
    // 50.7 -> 49.9 us at 2^20 envs, 18.5 -> 17.5 at 2^18, fused rollout 13.24 -> 13.07 per step (profiles/r04_ab_wrap.txt)
    for (int k = 0; k < 2; k++) ns[k] = nsg_wrap_pi(ns[k]);
    const double mv1 = 4 * NSG_PI, mv2 = 9 * NSG_PI;
    ns[2] = fmin(fmax(ns[2], -mv1), mv1);
    ns[3] = fmin(fmax(ns[3], -mv2), mv2);
#pragma unroll
    for (int k = 0; k < 4; k++) s[k] = ns[k];
#if NSG_LIBM_EXACT
    bool term = (-env_cos(s[0]) - env_cos(s[1] + s[0])) > 1.0;
#else
    bool term = (-nsg_cos_t<kAcroPoly>(s[0]) - nsg_cos_t<kAcroPoly>(s[1] + s[0])) > 1.0;
#endif
    reward = term ? 0.0 : -1.0;
    return term;
  } else if constexpr (ENV == NSG_ENV_MOUNTAINCAR) {
    double position = s[0], velocity = s[1];
#if NSG_LIBM_EXACT
    velocity += (double)(ai - 1) * th[1] + env_cos(3 * position) * (-th[0]);
#else
    velocity += (double)(ai - 1) * th[1] + nsg_cos_t<kLightPoly>(3 * position) * (-th[0]);
#endif
    if (velocity < -0.07) velocity = -0.07;
    if (velocity > 0.07) velocity = 0.07;
    position += velocity;
    if (position < -1.2) position = -1.2;
    if (position > 0.6) position = 0.6;
    if (position == -1.2 && velocity < 0) velocity = 0;
    s[0] = position; s[1] = velocity;
    reward = -1.0;
    return position >= 0.5 && velocity >= 0;
  } else {  // MountainCarContinuous
    double position = s[0], velocity = s[1];
    double a0 = (double)af;
    double force = fmin(fmax(a0, -1.0), 1.0);
#if NSG_LIBM_EXACT
    velocity += force * th[0] - 0.0025 * env_cos(3 * position);
#else
    velocity += force * th[0] - 0.0025 * nsg_cos_t<kLightPoly>(3 * position);
#endif
    if (velocity > 0.07) velocity = 0.07;
    if (velocity < -0.07) velocity = -0.07;
    position += velocity;
    if (position > 0.6) position = 0.6;
    if (position < -1.2) position = -1.2;
    if (position == -1.2 && velocity < 0) velocity = 0;
    bool term = position >= 0.45 && velocity >= 0;
    double r = 0;
    if (term) r = 100.0;
    r -= env_sq(a0) * 0.1;   // math.pow(action[0], 2)
    s[0] = (double)(float)position; s[1] = (double)(float)velocity;  // state stored as float32 upstream
    reward = r;
    return term;
  }
}

}  // namespace nsg
