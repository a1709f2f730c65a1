// nsg_theta.hip.h — the θ-schedule engine in device code: "when" (Scheduler) and "how"
// (UpdateFn) a parameter changes at wrapper time t.
//
// Replaces, per (env, param) and per step:
//   Scheduler.__call__ / _check         ns_gym/base.py:67-81, ns_gym/schedulers.py
//   UpdateFn.__call__ / _update          ns_gym/base.py:124-149, ns_gym/update_functions/single_param.py
//   UpdateDistributionFn + W1 delta      ns_gym/base.py:185-203, ns_gym/utils.py:55-94,
//                                        ns_gym/update_functions/distribution.py
//
// The (scheduler kind, update kind, constants) of a param are identical for all envs of a
// batch, so every switch below is wave-uniform (scalar branch, no divergence); only the
// fire predicate and θ are per-lane.  All arithmetic is float64 in the reference's operation
// order; the translation unit is compiled with -ffp-contract=off so a*b+c rounds twice like
// Python floats (Increment/Decrement/W1 results are bit-identical to the reference).
#pragma once
#include "nsgym_hip.h"
#include "nsg_rng.hip.h"
#include "nsg_libm.hip.h"

namespace nsg {

struct Tables {       // constant-table blob staged in LDS (bit tables, value tables, FrozenLake desc)
  const uint8_t* base;
  __device__ __forceinline__ const uint32_t* bits(int off_words) const { return (const uint32_t*)base + off_words; }
  __device__ __forceinline__ const double* vals(int off_doubles) const { return (const double*)base + off_doubles; }
};

// Scheduler.__call__: start <= t <= end and _check(t)
__device__ __forceinline__ bool sched_fire(const nsg_param_cfg& pc, const Tables& tb, int t) {
  double td = (double)t;
  if (!(pc.sched_start <= td && td <= pc.sched_end)) return false;
  switch (pc.sched_kind) {
    case NSG_SCHED_CONTINUOUS: return true;
    case NSG_SCHED_PERIODIC: return (t % (int)pc.sched_i0) == 0;
    case NSG_SCHED_BURST: return (t % (int)(pc.sched_i0 + pc.sched_i1)) < (int)pc.sched_i0;
    case NSG_SCHED_TABLE: {
      if (t < 0) return false;
      if (t >= pc.sched_tab_len) return pc.sched_i0 == 1;  // 2: a sampled CustomScheduler beyond its horizon (sched_overrun)
      return (tb.bits(pc.sched_tab_off)[t >> 5] >> (t & 31)) & 1u;
    }
    default: return false;
  }
}

// A CustomScheduler is a Python callable sampled into a bit table over a horizon; asked about a later t (inside its
// start / end gate) the table has no answer: the kernels count that (NSG_CNT_SCHED_OVERRUN) and the host raises.
__device__ __forceinline__ bool sched_overrun(const nsg_param_cfg& pc, int t) {
  const double td = (double)t;
  return pc.sched_kind == NSG_SCHED_TABLE && pc.sched_i0 == 2 && t >= pc.sched_tab_len && pc.sched_start <= td && td <= pc.sched_end;
}

__host__ __device__ inline bool sched_is_stochastic(int k) {
  return k == NSG_SCHED_RANDOM || k == NSG_SCHED_DECAYING || k == NSG_SCHED_MEMORYLESS;
}

// Construction-time state of a stochastic scheduler: rng = default_rng(seed) (+ the constructor's first
// geometric draw for Memoryless, schedulers.py:107-108); seed=None -> a fixed per-env stream.
__device__ inline void sched_construct(const nsg_param_cfg& pc, const ZigLds& zg, int p, int64_t i, Pcg& r, int& next) {
  if (pc.has_sched_seed) pcg_seed(r, pc.sched_seed, -1);
  else pcg_seed(r, (uint64_t)i, 2000 + p);
  next = 0;
  if (pc.sched_kind == NSG_SCHED_MEMORYLESS) {
    const int64_t g = pcg_geometric(r, zg, pc.sched_p0);
    next = g > 0x7fffffff ? 0x7fffffff : (int)g;
  }
}

// Scheduler.__call__ of the stochastic kinds: a draw happens only when start <= t <= end
__device__ inline bool sched_fire_stoch(const nsg_param_cfg& pc, const ZigLds& zg, int t, Pcg& r, int& next) {
  const double td = (double)t;
  if (!(pc.sched_start <= td && td <= pc.sched_end)) return false;
  switch (pc.sched_kind) {
    case NSG_SCHED_RANDOM: return pcg_double(r) < pc.sched_p0;                                   // schedulers.py:27-28
    case NSG_SCHED_DECAYING: return pcg_double(r) < pc.sched_p0 * nsg_exp(-pc.sched_p1 * td);   // :175-177
    case NSG_SCHED_MEMORYLESS:                                                                  // :110-116
      if (t == next) {
        const int64_t g = pcg_geometric(r, zg, pc.sched_p0) + t;
        next = g > 0x7fffffff ? 0x7fffffff : (int)g;
        return true;
      }
      return false;
    default: return false;
  }
}

// Which update kinds need the "full" θ-engine build (float64 exp/sin/log1p from the device
// library + the ziggurat sampler).  Batches whose update fns are all plain arithmetic /
// table look-ups run a kernel instantiated without those paths: its register footprint is
// much smaller, so more wavefronts stay resident to cover HBM latency.
__host__ __device__ inline bool upd_kind_is_simple(int k) {
  return k == NSG_UPD_INCREMENT || k == NSG_UPD_DECREMENT || k == NSG_UPD_TREND || k == NSG_UPD_POLY ||
         k == NSG_UPD_GEOMETRIC || k == NSG_UPD_LERP || k == NSG_UPD_STEPWISE || k == NSG_UPD_CYCLIC ||
         k == NSG_UPD_NOUPDATE || (k >= NSG_UPD_D_INCREMENT && k != NSG_UPD_D_RANDOMCAT && k != NSG_UPD_D_LCBOUNDED);
}

// UpdateFn._update for the scalar classes.  `rng` is touched only by the stochastic kinds.
template <bool FULL>
__device__ inline double upd_scalar(const nsg_param_cfg& pc, const Tables& tb, const ZigLds& zg, double th, int t,
                                    Pcg& rng, int& cursor) {
  const double* u = pc.u;
  const double td = (double)t;
  switch (pc.upd_kind) {
    case NSG_UPD_INCREMENT: return th + u[0];
    case NSG_UPD_DECREMENT: return th - u[0];
    case NSG_UPD_TREND: return th + u[0] * td;
    case NSG_UPD_POLY: {
      const double* c = tb.vals(pc.val_tab_off);
      double trend = 0.0, pw = 1.0;
      for (int i = 0; i < pc.val_tab_len; i++) {
        pw *= td;
        trend = trend + c[i] * pw;
      }
      return th + trend;
    }
    case NSG_UPD_GEOMETRIC: return th * u[0];
    case NSG_UPD_EXPDECAY: if constexpr (FULL) return th * nsg_exp(-u[0] * td); else return th;
#if NSG_LIBM_EXACT
    case NSG_UPD_OSCILLATING: if constexpr (FULL) return th + u[0] * env_sin(td); else return th;   // np.sin(t): libm's
#else
    case NSG_UPD_OSCILLATING: if constexpr (FULL) return th + u[0] * nsg_sin(td); else return th;
#endif
    case NSG_UPD_SIGMOID: {
      if constexpr (FULL) {
        double sg = 1.0 / (1.0 + nsg_exp(-u[2] * (td - u[3])));
        return u[0] + (u[1] - u[0]) * sg;
      } else return th;
    }
    case NSG_UPD_LERP: {
      double frac = td / u[2];
      if (!(frac < 1.0)) frac = 1.0;
      return u[0] + (u[1] - u[0]) * frac;
    }
    case NSG_UPD_STEPWISE: {
      if (cursor < pc.val_tab_len) return tb.vals(pc.val_tab_off)[cursor++];
      return th;
    }
    case NSG_UPD_CYCLIC: {
      double v = tb.vals(pc.val_tab_off)[cursor];
      cursor = cursor + 1 == pc.val_tab_len ? 0 : cursor + 1;
      return v;
    }
    case NSG_UPD_NOUPDATE: return th;
    default: break;
  }
  if constexpr (FULL) {
    switch (pc.upd_kind) {
      case NSG_UPD_RANDOMWALK: return th + pcg_normal(rng, zg, u[0], u[1]);
      case NSG_UPD_RW_DRIFT: {
        double w = pcg_normal(rng, zg, u[1], u[2]);
        return u[0] + th + w;
      }
      case NSG_UPD_RW_DRIFT_TREND: {
        double w = pcg_normal(rng, zg, u[1], u[2]);
        return u[0] + th + w + u[3] * td;
      }
      case NSG_UPD_OU: {
        double noise = u[2] > 0 ? pcg_normal(rng, zg, 0.0, u[2]) : 0.0;
        return th + u[0] * (u[1] - th) + noise;
      }
      case NSG_UPD_BOUNDED_RW: {
        double v = th + pcg_normal(rng, zg, u[0], u[1]);
        if (v < u[2]) v = u[2];
        if (v > u[3]) v = u[3];
        return v;
      }
      default: break;
    }
  }
  return th;
}

__host__ __device__ __forceinline__ bool upd_uses_cursor(int kind) {
  return kind == NSG_UPD_STEPWISE || kind == NSG_UPD_CYCLIC || kind == NSG_UPD_D_STEPWISE || kind == NSG_UPD_D_CYCLIC ||
         kind == NSG_UPD_D_LCBOUNDED;  // LCBounded keeps prev_time + 1 in its cursor row
}

// 1-Wasserstein distance between two pmfs on {0..ND-1}: SciPy's _cdf_distance(p=1) with
// values = arange(ND): np.sum(|cdf_u - cdf_v| * deltas), deltas = [0,1,0,1,...,0] over the merged
// support (fewer than 8 terms: summed in index order); weights normalised by their sums.
template <int ND> __device__ __forceinline__ double w1_n(const double* a, const double* b) {
  double ca[ND], cb[ND];
  ca[0] = a[0]; cb[0] = b[0];
#pragma unroll
  for (int k = 1; k < ND; k++) { ca[k] = ca[k - 1] + a[k]; cb[k] = cb[k - 1] + b[k]; }
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < ND - 1; k++) {
    acc = acc + 0.0;
    acc = acc + fabs(ca[k] / ca[ND - 1] - cb[k] / cb[ND - 1]);
  }
  return acc + 0.0;
}

// UpdateDistributionFn._update for the slip distributions of the grid wrappers (ND = 3 or 4).
template <int ND, bool FULL>
__device__ inline void upd_dist(const nsg_param_cfg& pc, const Tables& tb, const ZigLds& zg, const double* p, int t, int& cursor,
                                Pcg& rng, double* q, bool& exhausted) {
  exhausted = false;
  const double* u = pc.u;
  const double td = (double)t;
#pragma unroll
  for (int k = 0; k < ND; k++) q[k] = p[k];
  switch (pc.upd_kind) {
    case NSG_UPD_D_INCREMENT: {
      double v = p[0] + u[0];
      q[0] = v > 1.0 ? 1.0 : v;
#pragma unroll
      for (int k = 1; k < ND; k++) q[k] = (1.0 - q[0]) / (double)(ND - 1);
      break;
    }
    case NSG_UPD_D_DECREMENT: {
      double v = p[0] - u[0];
      q[0] = v < 0.0 ? 0.0 : v;
#pragma unroll
      for (int k = 1; k < ND; k++) q[k] = (1.0 - q[0]) / (double)(ND - 1);
      break;
    }
    case NSG_UPD_D_STEPWISE:
      if (cursor < pc.val_tab_len) {
        const double* v = tb.vals(pc.val_tab_off) + ND * cursor;
#pragma unroll
        for (int k = 0; k < ND; k++) q[k] = v[k];
        cursor++;
      }
      break;
    case NSG_UPD_D_CYCLIC: {
      const double* v = tb.vals(pc.val_tab_off) + ND * cursor;
#pragma unroll
      for (int k = 0; k < ND; k++) q[k] = v[k];
      cursor = cursor + 1 == pc.val_tab_len ? 0 : cursor + 1;
      break;
    }
    case NSG_UPD_D_NOUPDATE: break;
    case NSG_UPD_D_UNIFORMDRIFT: {
      const double un = 1.0 / ND;
#pragma unroll
      for (int k = 0; k < ND; k++) q[k] = (1 - u[0]) * p[k] + u[0] * un;
      break;
    }
    case NSG_UPD_D_TARGETREV:
#pragma unroll
      for (int k = 0; k < ND; k++) q[k] = p[k] + u[ND] * (u[k] - p[k]);
      break;
    case NSG_UPD_D_RANDOMCAT: {  // rng.dirichlet(ones(n)): n standard exponentials, then val *= 1/acc
      if constexpr (FULL) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < ND; k++) { q[k] = pcg_std_exponential(rng, zg); acc = acc + q[k]; }
        const double invacc = 1.0 / acc;
#pragma unroll
        for (int k = 0; k < ND; k++) q[k] = q[k] * invacc;
      }
      break;
    }
    case NSG_UPD_D_LCBOUNDED: {  // distribution.py:167-183; `cursor` holds prev_time + 1
      if constexpr (FULL) {
        const double d = u[0] * fabs(td - (double)(cursor - 1));
        if (u[1] != 0.0) break;  // inner DistributionNoUpdate: W1 = 0 <= d, accepted at once
        exhausted = true;
        for (int tries = 0; tries < 100000; tries++) {  // max_trys = int(1e5); bounded, so the wave always drains
          double cand[ND], acc = 0.0;
#pragma unroll
          for (int k = 0; k < ND; k++) { cand[k] = pcg_std_exponential(rng, zg); acc = acc + cand[k]; }
          const double invacc = 1.0 / acc;
#pragma unroll
          for (int k = 0; k < ND; k++) cand[k] = cand[k] * invacc;
          if (w1_n<ND>(p, cand) <= d) {
#pragma unroll
            for (int k = 0; k < ND; k++) q[k] = cand[k];
            exhausted = false;
            break;
          }
        }  // exhausted: the reference raises ValueError (:178-182); here the distribution stays unchanged and the caller counts it
      }
      break;
    }
    case NSG_UPD_D_LERP: {
      double frac = td / u[2 * ND];
      if (!(frac < 1.0)) frac = 1.0;
#pragma unroll
      for (int k = 0; k < ND; k++) q[k] = u[k] + (u[ND + k] - u[k]) * frac;
      break;
    }
    default: break;
  }
}

}  // namespace nsg
