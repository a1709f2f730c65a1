/* nsgym_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement (plain C, one env at a time, float64 like the reference) of the
 * hot path the HIP library accelerates:
 *
 *   NSClassicControlWrapper.step/reset   ns_gym/wrappers/classic_control.py:60-109,193-458
 *   NSFrozenLakeWrapper.step/reset       ns_gym/wrappers/toy_text.py:342-399,426-469
 *   NSWrapper.step/reset/_seed_update_fns ns_gym/base.py:296-431
 *   Scheduler.__call__ / UpdateFn.__call__ ns_gym/base.py:67-81,124-149,182,192-203
 *   schedulers / update functions        ns_gym/schedulers.py, ns_gym/update_functions/{single_param,distribution}.py
 *   wasserstein_distance                 ns_gym/utils.py:55-94 (SciPy _cdf_distance, p=1)
 *   base MDPs                            gymnasium 1.2.1 [UPSTREAM, uv.lock:958-959, absent]
 *   bit streams                          NumPy SeedSequence/PCG64/Generator [UPSTREAM, uv.lock:1868-1869]
 *
 * Pinning (tests/test_oracle_*.py): every schedule/update/wrapper golden vector under
 * tests/golden/ was produced by the reference's own classes; the NumPy stream vectors by
 * the NumPy installed in the build container.  Integrator arithmetic (a20) is restated
 * from the published gymnasium sources and is PARITY-UNPINNED by the reference's tests.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * It shares with the product ONLY the data-format header (include/nsgym_hip.h) and the
 * NumPy ziggurat table data; no code.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared  (see oracle/Makefile).  Contraction is off
 * so every a*b+c rounds twice exactly like the reference's Python floats.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/nsgym_hip.h"
#include "../include/nsg_zig_tables.inc"

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ NumPy SeedSequence */
/* numpy/random/bit_generator.pyx [UPSTREAM]: 32-bit hash-mix entropy pool, pool size 4. */
#define SS_INIT_A 0x43b0d7e5u
#define SS_MULT_A 0x931e8875u
#define SS_INIT_B 0x8b51f9ddu
#define SS_MULT_B 0x58f38dedu
#define SS_MIX_L 0xca01f9ddu
#define SS_MIX_R 0x4973f715u

static uint32_t ss_hashmix(uint32_t v, uint32_t* hc) {
  v ^= *hc;
  *hc *= SS_MULT_A;
  v *= *hc;
  v ^= v >> 16;
  return v;
}
static uint32_t ss_mix(uint32_t x, uint32_t y) {
  uint32_t r = SS_MIX_L * x - SS_MIX_R * y;
  r ^= r >> 16;
  return r;
}

/* SeedSequence(entropy=seed, spawn_key=(child,) if child >= 0 else ()).generate_state(4, uint64) */
static void ss_generate(uint64_t seed, int child, uint64_t out[4]) {
  uint32_t ent[8];
  int n = 0;
  ent[n++] = (uint32_t)seed;
  if (seed >> 32) ent[n++] = (uint32_t)(seed >> 32);
  if (child >= 0) {
    while (n < 4) ent[n++] = 0; /* entropy zero-padded to pool size before the spawn key */
    ent[n++] = (uint32_t)child;
  }
  uint32_t pool[4], hc = SS_INIT_A;
  for (int i = 0; i < 4; i++) pool[i] = ss_hashmix(i < n ? ent[i] : 0u, &hc);
  for (int s = 0; s < 4; s++)
    for (int d = 0; d < 4; d++)
      if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], &hc));
  for (int s = 4; s < n; s++)
    for (int d = 0; d < 4; d++) pool[d] = ss_mix(pool[d], ss_hashmix(ent[s], &hc));
  uint32_t hb = SS_INIT_B, w[8];
  for (int i = 0; i < 8; i++) {
    uint32_t v = pool[i & 3];
    v ^= hb;
    hb *= SS_MULT_B;
    v *= hb;
    v ^= v >> 16;
    w[i] = v;
  }
  for (int i = 0; i < 4; i++) out[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}

/* ------------------------------------------------------------------ NumPy PCG64 */
typedef struct { u128 state, inc; } pcg64;
#define PCG_MULT ((((u128)2549297995355413924ULL) << 64) | (u128)4865540595714422341ULL)

static void pcg_seed(pcg64* r, uint64_t seed, int child) {
  uint64_t v[4];
  ss_generate(seed, child, v);
  u128 initstate = ((u128)v[0] << 64) | v[1];
  u128 initseq = ((u128)v[2] << 64) | v[3];
  r->state = 0;
  r->inc = (initseq << 1) | 1;
  r->state = r->state * PCG_MULT + r->inc;
  r->state += initstate;
  r->state = r->state * PCG_MULT + r->inc;
}
static uint64_t pcg_next64(pcg64* r) {
  r->state = r->state * PCG_MULT + r->inc;
  uint64_t hi = (uint64_t)(r->state >> 64), lo = (uint64_t)r->state;
  uint64_t x = hi ^ lo;
  unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((-rot) & 63));
}
static double pcg_double(pcg64* r) { return (double)(pcg_next64(r) >> 11) * (1.0 / 9007199254740992.0); }

/* numpy/random/src/distributions/distributions.c random_standard_normal [UPSTREAM] */
static double u2d(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
#define ZIG_R 3.6541528853610087963519472518
#define ZIG_INV_R 0.27366123732975827203338247596
static double pcg_std_normal(pcg64* g) {
  for (;;) {
    uint64_t r = pcg_next64(g);
    int idx = (int)(r & 0xff);
    r >>= 8;
    int sign = (int)(r & 1);
    uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
    double x = (double)rabs * u2d(NSG_ZIG_WI_BITS[idx]);
    if (sign) x = -x;
    if (rabs < NSG_ZIG_KI[idx]) return x;
    if (idx == 0) {
      for (;;) {
        double xx = -ZIG_INV_R * log1p(-pcg_double(g));
        double yy = -log1p(-pcg_double(g));
        if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? -(ZIG_R + xx) : ZIG_R + xx;
      }
    } else {
      double f1 = u2d(NSG_ZIG_FI_BITS[idx - 1]), f0 = u2d(NSG_ZIG_FI_BITS[idx]);
      if ((f1 - f0) * pcg_double(g) + f0 < exp(-0.5 * x * x)) return x;
    }
  }
}
static double pcg_normal(pcg64* g, double loc, double scale) { return loc + scale * pcg_std_normal(g); }

/* numpy random_standard_exponential (256-layer ziggurat) [UPSTREAM distributions.c] */
#define ZIGE_R 7.69711747013104972
static double pcg_std_exponential(pcg64* g) {
  for (;;) {
    uint64_t ri = pcg_next64(g);
    ri >>= 3;
    int idx = (int)(ri & 0xFF);
    ri >>= 8;
    double x = (double)ri * u2d(NSG_ZIGE_WE_BITS[idx]);
    if (ri < NSG_ZIGE_KE[idx]) return x; /* 98.9 % */
    if (idx == 0) return ZIGE_R - log1p(-pcg_double(g));
    double f1 = u2d(NSG_ZIGE_FE_BITS[idx - 1]), f0 = u2d(NSG_ZIGE_FE_BITS[idx]);
    if ((f1 - f0) * pcg_double(g) + f0 < exp(-x)) return x;
  }
}
/* numpy random_geometric: search for p >= 1/3, inversion otherwise */
static int64_t pcg_geometric(pcg64* g, double p) {
  if (p >= 0.333333333333333333333333) {
    int64_t X = 1;
    double sum = p, prod = p, q = 1.0 - p;
    double U = pcg_double(g);
    while (U > sum) { prod *= q; sum += prod; X++; }
    return X;
  }
  double z = ceil(-pcg_std_exponential(g) / log1p(-p));
  if (z >= 9.223372036854776e+18) return INT64_MAX;
  return (int64_t)z;
}

/* streams are stored as 32-byte records [N][4]: state_hi, state_lo, inc_hi, inc_lo */
static void rng_load(const uint64_t* base, int64_t N, int64_t i, pcg64* r) {
  (void)N;
  r->state = ((u128)base[4 * i + 0] << 64) | base[4 * i + 1];
  r->inc = ((u128)base[4 * i + 2] << 64) | base[4 * i + 3];
}
static void rng_store(uint64_t* base, int64_t N, int64_t i, const pcg64* r) {
  (void)N;
  base[4 * i + 0] = (uint64_t)(r->state >> 64);
  base[4 * i + 1] = (uint64_t)r->state;
  base[4 * i + 2] = (uint64_t)(r->inc >> 64);
  base[4 * i + 3] = (uint64_t)r->inc;
}

/* ------------------------------------------------------------------ schedulers */
/* Scheduler.__call__ (ns_gym/base.py:67-81): start <= t <= end and _check(t). */
static int sched_fire(const nsg_param_cfg* pc, const uint8_t* tables, int t) {
  double td = (double)t;
  if (!(pc->sched_start <= td && td <= pc->sched_end)) return 0;
  switch (pc->sched_kind) {
    case NSG_SCHED_CONTINUOUS: return 1;                          /* schedulers.py:52-53 */
    case NSG_SCHED_PERIODIC: return (t % pc->sched_i0) == 0;      /* :88-89 */
    case NSG_SCHED_BURST: return (t % (pc->sched_i0 + pc->sched_i1)) < pc->sched_i0; /* :139-140 */
    case NSG_SCHED_TABLE: {                                       /* :73-74, :197-198, :42-43 */
      if (t < 0) return 0;
      if (t >= pc->sched_tab_len) return pc->sched_i0 == 1; /* 2 = a sampled callable beyond its horizon: the reference would call it; no answer here */
      const uint32_t* bits = (const uint32_t*)tables + pc->sched_tab_off;
      return (bits[t >> 5] >> (t & 31)) & 1;
    }
    default: return 0;
  }
}

static int sched_is_stochastic(int k) { return k == NSG_SCHED_RANDOM || k == NSG_SCHED_DECAYING || k == NSG_SCHED_MEMORYLESS; }

/* construction-time state of a stochastic scheduler: rng = default_rng(seed) (+ the ctor's first
 * geometric draw for Memoryless, schedulers.py:107-108).  seed=None -> a fixed per-env stream. */
static void sched_construct(const nsg_param_cfg* pc, int p, int64_t i, pcg64* r, int32_t* next) {
  if (pc->has_sched_seed) pcg_seed(r, pc->sched_seed, -1);
  else pcg_seed(r, (uint64_t)i, 2000 + p);
  *next = 0;
  if (pc->sched_kind == NSG_SCHED_MEMORYLESS) {
    int64_t g = pcg_geometric(r, pc->sched_p0);
    *next = g > 0x7fffffff ? 0x7fffffff : (int32_t)g;
  }
}

/* Scheduler.__call__ for the stochastic kinds: draws only when start <= t <= end */
static int sched_fire_stoch(const nsg_param_cfg* pc, int t, pcg64* r, int32_t* next) {
  double td = (double)t;
  if (!(pc->sched_start <= td && td <= pc->sched_end)) return 0;
  switch (pc->sched_kind) {
    case NSG_SCHED_RANDOM: return pcg_double(r) < pc->sched_p0;                               /* schedulers.py:27-28 */
    case NSG_SCHED_DECAYING: return pcg_double(r) < pc->sched_p0 * exp(-pc->sched_p1 * td);   /* :175-177 */
    case NSG_SCHED_MEMORYLESS:                                                               /* :110-116 */
      if (t == *next) {
        int64_t g = pcg_geometric(r, pc->sched_p0) + t;
        *next = g > 0x7fffffff ? 0x7fffffff : (int32_t)g;
        return 1;
      }
      return 0;
    default: return 0;
  }
}

/* fire predicate of param p of env i, advancing the scheduler's own state when it has one */
static int fire_param(const nsg_config* cfg, const uint8_t* tables, const nsg_buffers* b, int64_t N, int64_t i, int p, int t) {
  const nsg_param_cfg* pc = &cfg->params[p];
  if (!sched_is_stochastic(pc->sched_kind)) return sched_fire(pc, tables, t);
  pcg64 r;
  rng_load(b->rng_sched + (int64_t)pc->sched_slot * 4 * N, N, i, &r);
  int32_t next = b->sched_next[pc->sched_slot * N + i];
  int f = sched_fire_stoch(pc, t, &r, &next);
  rng_store(b->rng_sched + (int64_t)pc->sched_slot * 4 * N, N, i, &r);
  b->sched_next[pc->sched_slot * N + i] = next;
  return f;
}

/* ------------------------------------------------------------------ scalar update fns */
static const double* val_table(const nsg_param_cfg* pc, const uint8_t* tables) {
  return (const double*)tables + pc->val_tab_off;
}

/* UpdateFn._update for every scalar class (single_param.py); rng may be NULL if !uses_rng. */
static double upd_scalar(const nsg_param_cfg* pc, const uint8_t* tables, double th, int t, pcg64* rng,
                         int32_t* cursor) {
  const double* u = pc->u;
  double td = (double)t;
  switch (pc->upd_kind) {
    case NSG_UPD_INCREMENT: return th + u[0];                               /* :173-175 */
    case NSG_UPD_DECREMENT: return th - u[0];                               /* :197-199 */
    case NSG_UPD_TREND: return th + u[0] * td;                              /* :38-40 */
    case NSG_UPD_POLY: {                                                    /* :471-473 */
      const double* c = val_table(pc, tables);
      double trend = 0.0, pw = 1.0;
      for (int i = 0; i < pc->val_tab_len; i++) {
        pw *= td; /* t**(i+1), exact in fp64 for the horizons used */
        trend = trend + c[i] * pw;
      }
      return th + trend;
    }
    case NSG_UPD_GEOMETRIC: return th * u[0];                               /* :305-307 */
    case NSG_UPD_EXPDECAY: return th * exp(-u[0] * td);                     /* :285-287 */
    case NSG_UPD_OSCILLATING: return th + u[0] * sin(td);                   /* :262-264 */
    case NSG_UPD_SIGMOID: {                                                 /* :383-385 */
      double sg = 1.0 / (1.0 + exp(-u[2] * (td - u[3])));
      return u[0] + (u[1] - u[0]) * sg;
    }
    case NSG_UPD_LERP: {                                                    /* :506-508 */
      double frac = td / u[2];
      if (!(frac < 1.0)) frac = 1.0; /* min(t / T, 1.0) */
      return u[0] + (u[1] - u[0]) * frac;
    }
    case NSG_UPD_STEPWISE: {                                                /* :217-223 */
      if (*cursor < pc->val_tab_len) return val_table(pc, tables)[(*cursor)++];
      return th; /* exhausted list: value unchanged, still reported as fired */
    }
    case NSG_UPD_CYCLIC: {                                                  /* :405-408 */
      double v = val_table(pc, tables)[*cursor];
      *cursor = (*cursor + 1) % pc->val_tab_len;
      return v;
    }
    case NSG_UPD_NOUPDATE: return th;                                       /* :239-240 */
    case NSG_UPD_RANDOMWALK: return th + pcg_normal(rng, u[0], u[1]);       /* :110-113 */
    case NSG_UPD_RW_DRIFT: {                                                /* :148-151 */
      double w = pcg_normal(rng, u[1], u[2]);
      return u[0] + th + w;
    }
    case NSG_UPD_RW_DRIFT_TREND: {                                          /* :78-81 */
      double w = pcg_normal(rng, u[1], u[2]);
      return u[0] + th + w + u[3] * td;
    }
    case NSG_UPD_OU: {                                                      /* :344-346 */
      double noise = u[2] > 0 ? pcg_normal(rng, 0.0, u[2]) : 0.0;
      return th + u[0] * (u[1] - th) + noise;
    }
    case NSG_UPD_BOUNDED_RW: {                                              /* :446-448 */
      double v = th + pcg_normal(rng, u[0], u[1]);
      if (v < u[2]) v = u[2];
      if (v > u[3]) v = u[3];
      return v;
    }
    default: return th;
  }
}

/* ------------------------------------------------------------------ distribution update fns */
#define NSG_ND_MAX 4
static int is_grid_env(int env) { return env == NSG_ENV_FROZENLAKE || env == NSG_ENV_CLIFFWALKING || env == NSG_ENV_BRIDGE; }
static int n_dist(int env) { return env == NSG_ENV_CLIFFWALKING ? 4 : 3; }

/* ns_gym/utils.py:55-94 -> scipy.stats.wasserstein_distance(values=arange(n), weights) ->
 * _cdf_distance(p=1): np.sum(|U_cdf - V_cdf| * deltas), deltas = [0,1,0,1,...,0] over the merged
 * support, cdf_k = cumsum_k / cumsum_last; fewer than 8 terms -> summed in index order. */
static double w1_n(const double* a, const double* b, int n) {
  double ca[NSG_ND_MAX], cb[NSG_ND_MAX];
  ca[0] = a[0]; cb[0] = b[0];
  for (int k = 1; k < n; k++) { ca[k] = ca[k - 1] + a[k]; cb[k] = cb[k - 1] + b[k]; }
  double acc = 0.0;
  for (int k = 0; k < n - 1; k++) {
    acc = acc + 0.0;
    acc = acc + fabs(ca[k] / ca[n - 1] - cb[k] / cb[n - 1]);
  }
  return acc + 0.0;
}

static void upd_dist(const nsg_param_cfg* pc, const uint8_t* tables, const double* p, int n, int t, int32_t* cursor,
                     pcg64* rng, double* q) {
  const double* u = pc->u;
  double td = (double)t;
  for (int k = 0; k < n; k++) q[k] = p[k];
  switch (pc->upd_kind) {
    case NSG_UPD_D_INCREMENT: {                        /* distribution.py:61-67 */
      double v = p[0] + u[0];
      q[0] = v > 1.0 ? 1.0 : v;                        /* min(1, p0 + k) */
      for (int k = 1; k < n; k++) q[k] = (1.0 - q[0]) / (double)(n - 1);
      break;
    }
    case NSG_UPD_D_DECREMENT: {                        /* :88-97 */
      double v = p[0] - u[0];
      q[0] = v < 0.0 ? 0.0 : v;                        /* max(0, p0 - k) */
      for (int k = 1; k < n; k++) q[k] = (1.0 - q[0]) / (double)(n - 1);
      break;
    }
    case NSG_UPD_D_STEPWISE:                           /* :116-130 */
      if (*cursor < pc->val_tab_len) {
        const double* v = val_table(pc, tables) + n * (*cursor)++;
        for (int k = 0; k < n; k++) q[k] = v[k];
      }
      break;
    case NSG_UPD_D_CYCLIC: {                           /* :353-356 */
      const double* v = val_table(pc, tables) + n * (*cursor);
      for (int k = 0; k < n; k++) q[k] = v[k];
      *cursor = (*cursor + 1) % pc->val_tab_len;
      break;
    }
    case NSG_UPD_D_NOUPDATE: break;                    /* :230-231 */
    case NSG_UPD_D_UNIFORMDRIFT: {                     /* :256-261 */
      double un = 1.0 / n;
      for (int k = 0; k < n; k++) q[k] = (1 - u[0]) * p[k] + u[0] * un;
      break;
    }
    case NSG_UPD_D_TARGETREV:                          /* :289-293 */
      for (int k = 0; k < n; k++) q[k] = p[k] + u[n] * (u[k] - p[k]);
      break;
    case NSG_UPD_D_RANDOMCAT: {                        /* :38  rng.dirichlet(np.ones(n)): numpy draws n
         standard gammas of shape 1.0 (== standard exponentials), then val *= 1/acc [UPSTREAM _generator.pyx] */
      double acc = 0.0;
      for (int k = 0; k < n; k++) { q[k] = pcg_std_exponential(rng); acc = acc + q[k]; }
      double invacc = 1.0 / acc;
      for (int k = 0; k < n; k++) q[k] = q[k] * invacc;
      break;
    }
    case NSG_UPD_D_LCBOUNDED: {                        /* :167-183; *cursor holds prev_time + 1 */
      double d = u[0] * fabs((double)t - (double)(*cursor - 1));
      if (u[1] != 0.0) break; /* inner DistributionNoUpdate: W1 = 0 <= d, accepted at once */
      for (int tries = 0; tries < 100000; tries++) {
        double cand[NSG_ND_MAX], acc = 0.0;
        for (int k = 0; k < n; k++) { cand[k] = pcg_std_exponential(rng); acc = acc + cand[k]; }
        double invacc = 1.0 / acc;
        for (int k = 0; k < n; k++) cand[k] = cand[k] * invacc;
        if (w1_n(p, cand, n) <= d) {
          for (int k = 0; k < n; k++) q[k] = cand[k];
          break;
        }
      } /* exhausted: the reference raises ValueError; here the distribution stays unchanged */
      break;
    }
    case NSG_UPD_D_LERP: {                             /* :326-331 */
      double frac = td / u[2 * n];
      if (!(frac < 1.0)) frac = 1.0;
      for (int k = 0; k < n; k++) q[k] = u[k] + (u[n + k] - u[k]) * frac;
      break;
    }
    default: break;
  }
}

/* ------------------------------------------------------------------ env tables */
static const int PHYS_DIM[NSG_ENV_COUNT] = {4, 2, 4, 2, 2, 0, 0, 0};
static const int OBS_DIM[NSG_ENV_COUNT] = {4, 3, 6, 2, 2, 1, 1, 1};
static const int N_THETA[NSG_ENV_COUNT] = {6, 4, 8, 2, 1, 1, 1, 3};

int orc_phys_dim(int env) { return PHYS_DIM[env]; }
int orc_obs_dim(int env) { return OBS_DIM[env]; }

/* constraint checker (ns_gym/wrappers/classic_control.py:193-422).  nv = proposed values for
 * every θ slot (untuned slots carry their current value), cur = pre-update attributes,
 * tuned = bit mask of slots present in new_vals.  Returns bit mask of violated slots. */
static unsigned constraint_mask(int env, const double* nv, const double* cur, unsigned tuned) {
  unsigned v = 0;
  switch (env) {
    case NSG_ENV_CARTPOLE: /* :208-235  gravity masscart masspole force_mag tau length */
      if ((tuned & 1u) && nv[0] < 0) v |= 1u;
      if ((tuned & 2u) && nv[1] <= 0) v |= 2u;
      if ((tuned & 4u) && nv[2] <= 0) v |= 4u;
      if ((tuned & 32u) && nv[5] <= 0) v |= 32u;
      break;
    case NSG_ENV_PENDULUM: /* :389-420  m l dt g */
      if ((tuned & 1u) && nv[0] <= 0) v |= 1u;
      if ((tuned & 2u) && nv[1] <= 0) v |= 2u;
      if ((tuned & 4u) && nv[2] <= 0) v |= 4u;
      if ((tuned & 8u) && nv[3] < 0) v |= 8u;
      break;
    case NSG_ENV_ACROBOT: { /* :237-357  dt L1 L2 M1 M2 C1 C2 MOI */
      if (tuned & 2u) { /* LINK_LENGTH_1 :241-265 */
        if (nv[1] <= 0) v |= 2u;
        else if ((tuned & 32u) && nv[5] > nv[1]) v |= 2u;
        else if (nv[1] < cur[5]) v |= 2u;
      }
      if ((tuned & 4u) && nv[2] <= 0) v |= 4u;  /* LINK_LENGTH_2: only <= 0 is live (:267) */
      if ((tuned & 8u) && nv[3] <= 0) v |= 8u;  /* :293-298 */
      if ((tuned & 16u) && nv[4] <= 0) v |= 16u; /* :300-305 */
      if (tuned & 32u) { /* LINK_COM_POS_1 :307-331 */
        if (nv[5] <= 0) v |= 32u;
        else if ((tuned & 2u) && nv[1] < nv[5]) v |= 32u;
        else if (nv[5] > cur[1]) v |= 32u;
      }
      if (tuned & 64u) { /* LINK_COM_POS_2 :333-357 */
        if (nv[6] <= 0) v |= 64u;
        else if ((tuned & 4u) && nv[2] < nv[6]) v |= 64u;
        else if (nv[6] > cur[2]) v |= 64u;
      }
      break;
    }
    case NSG_ENV_MOUNTAINCAR: /* :359-376 gravity force */
      if ((tuned & 1u) && nv[0] <= 0) v |= 1u;
      if ((tuned & 2u) && nv[1] <= 0) v |= 2u;
      break;
    case NSG_ENV_MOUNTAINCAR_CONT: /* :378-387 power */
      if ((tuned & 1u) && nv[0] <= 0) v |= 1u;
      break;
    default: break;
  }
  return v;
}

/* ------------------------------------------------------------------ base MDP steps [UPSTREAM gymnasium 1.2.1] */
typedef struct { double s[4]; } phys4;

static void env_reset_draw(int env, pcg64* g, double* s) {
  switch (env) {
    case NSG_ENV_CARTPOLE: /* np_random.uniform(-0.05, 0.05, size=4): low + (high-low)*u */
      for (int k = 0; k < 4; k++) s[k] = -0.05 + (0.05 - -0.05) * pcg_double(g);
      break;
    case NSG_ENV_PENDULUM: { /* uniform(low=-[pi,1], high=[pi,1]) */
      s[0] = -M_PI + (M_PI - -M_PI) * pcg_double(g);
      s[1] = -1.0 + (1.0 - -1.0) * pcg_double(g);
      break;
    }
    case NSG_ENV_ACROBOT: /* uniform(-0.1, 0.1, size=4).astype(float32) */
      for (int k = 0; k < 4; k++) s[k] = (double)(float)(-0.1 + (0.1 - -0.1) * pcg_double(g));
      break;
    case NSG_ENV_MOUNTAINCAR:
    case NSG_ENV_MOUNTAINCAR_CONT: /* [uniform(-0.6, -0.4), 0] */
      s[0] = -0.6 + (-0.4 - -0.6) * pcg_double(g);
      s[1] = 0.0;
      break;
    default: break;
  }
}

static void env_obs(int env, const double* s, float* o) {
  switch (env) {
    case NSG_ENV_CARTPOLE: for (int k = 0; k < 4; k++) o[k] = (float)s[k]; break;
    case NSG_ENV_PENDULUM: o[0] = (float)cos(s[0]); o[1] = (float)sin(s[0]); o[2] = (float)s[1]; break;
    case NSG_ENV_ACROBOT:
      o[0] = (float)cos(s[0]); o[1] = (float)sin(s[0]); o[2] = (float)cos(s[1]); o[3] = (float)sin(s[1]);
      o[4] = (float)s[2]; o[5] = (float)s[3];
      break;
    case NSG_ENV_MOUNTAINCAR:
    case NSG_ENV_MOUNTAINCAR_CONT: o[0] = (float)s[0]; o[1] = (float)s[1]; break;
    default: break;
  }
}

static double py_fmod_pos(double x, double m) { /* Python float % for m > 0 */
  double r = fmod(x, m);
  if (r != 0 && r < 0) r += m;
  return r;
}

/* `x ** 2` on a float64 SCALAR [UPSTREAM gymnasium acrobot.py _dsdt, pendulum.py step; NumPy scalar power and CPython float_pow both
 * call libm's pow]: pow(x, 2.0) is within 0.52 ulp but not always the correctly rounded product - 0.08 % of arguments differ from
 * x * x in the last bit (checked against NumPy here).  The file is built with -fno-builtin-pow -fno-builtin-powf so that gcc does not
 * fold these calls into products.  (np.square and array ** 2 ARE products: CartPole.) */
static double sq_pow(double x) { return pow(x, 2.0); }

/* wrap(x, -pi, pi) [UPSTREAM gymnasium acrobot.py wrap]: `while x > M: x = x - diff; while x < m: x = x + diff` - the loop itself, one
 * rounded subtraction per turn, for every |x| below 2^22 (667 544 turns).  A step whose RK4 stages blew up can come out far
 * beyond that (the reference then loops for minutes, from 2^56 on for ever: x - diff == x).  So that a test run ends, larger
 * values are first brought down binade by binade with the SAME result: inside [2^e, 2^(e+1)), e >= 7, every turn subtracts
 * diff rounded to the binade's grid (exact argument in ns_gym_amd/csrc/nsg_math.hip.h, nsg_wrap_pi; tests/test_math_cpu.py
 * compares both with the plain loop up to 3e10).  From 2^56 on, and for inf, x is returned as it came. */
static double wrap_pi(double x) {
  const double M = M_PI, m = -M_PI, diff = M - m;
  double a = fabs(x);
  if (!(a < 72057594037927936.0)) return x;
  if (a >= 4194304.0) {
    while (a >= 4194304.0) {
      int e;
      (void)frexp(a, &e);
      const double floor2 = ldexp(1.0, e - 1); /* 2^e' <= a */
      volatile double up = floor2 + diff;
      const double step = up - floor2;
      const double n = floor((a - floor2) / step) - 3.0;
      if (n > 0.0) a = a - n * step;
      while (a >= floor2) a = a - diff;
    }
    x = x < 0 ? -a : a;
  }
  while (x > M) x = x - diff;
  while (x < m) x = x + diff;
  return x;
}

double orc_wrap_pi(double x) { return wrap_pi(x); } /* (for tests/test_math_cpu.py) */

static void acrobot_dsdt(const double* th, const double* y, double a, double* d) {
  /* AcrobotEnv._dsdt, "book" variant. th: dt L1 L2 M1 M2 C1 C2 MOI */
  double m1 = th[3], m2 = th[4], l1 = th[1], lc1 = th[5], lc2 = th[6], I1 = th[7], I2 = th[7], g = 9.8;
  double theta1 = y[0], theta2 = y[1], dtheta1 = y[2], dtheta2 = y[3];
  double d1 = m1 * sq_pow(lc1) + m2 * (sq_pow(l1) + sq_pow(lc2) + 2 * l1 * lc2 * cos(theta2)) + I1 + I2;
  double d2 = m2 * (sq_pow(lc2) + l1 * lc2 * cos(theta2)) + I2;
  double phi2 = m2 * lc2 * g * cos(theta1 + theta2 - M_PI / 2.0);
  double phi1 = -m2 * l1 * lc2 * sq_pow(dtheta2) * sin(theta2) - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * sin(theta2) +
                (m1 * lc1 + m2 * l1) * g * cos(theta1 - M_PI / 2) + phi2;
  double ddtheta2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * sq_pow(dtheta1) * sin(theta2) - phi2) /
                    (m2 * sq_pow(lc2) + I2 - sq_pow(d2) / d1);
  double ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;
  d[0] = dtheta1; d[1] = dtheta2; d[2] = ddtheta1; d[3] = ddtheta2; d[4] = 0.0;
}

/* returns terminated; writes reward */
static int env_step(int env, const double* th, double* s, int ai, float af, double* reward) {
  switch (env) {
    case NSG_ENV_CARTPOLE: { /* th: gravity masscart masspole force_mag tau length */
      double gravity = th[0], masspole = th[2], force_mag = th[3], tau = th[4], length = th[5];
      double total_mass = th[6], polemass_length = th[7]; /* resolved by the caller (classic_control.py:424-444) */
      double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
      double force = ai == 1 ? force_mag : -force_mag;
      double costheta = cos(theta), sintheta = sin(theta);
      double temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass;
      double thetaacc = (gravity * sintheta - costheta * temp) /
                        (length * (4.0 / 3.0 - masspole * (costheta * costheta) / total_mass));
      double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
      x = x + tau * x_dot;
      x_dot = x_dot + tau * xacc;
      theta = theta + tau * theta_dot;
      theta_dot = theta_dot + tau * thetaacc;
      s[0] = x; s[1] = x_dot; s[2] = theta; s[3] = theta_dot;
      double thr = 12 * 2 * M_PI / 360;
      *reward = 1.0;
      return x < -2.4 || x > 2.4 || theta < -thr || theta > thr;
    }
    case NSG_ENV_PENDULUM: { /* th: m l dt g */
      double m = th[0], l = th[1], dt = th[2], g = th[3];
      double t0 = s[0], thdot = s[1];
      double u = (double)af;
      if (u < -2.0) u = -2.0;
      if (u > 2.0) u = 2.0;
      double an = py_fmod_pos(t0 + M_PI, 2 * M_PI) - M_PI;
      /* `u` is a float32 scalar there (np.clip(u, ...)[0]): `u ** 2` is powf, rounded to float32, before the float64 `0.001 *` (NumPy
       * 1.26.4 scalar promotion, the reference's pin) */
      double costs = sq_pow(an) + 0.1 * sq_pow(thdot) + 0.001 * (double)powf((float)u, 2.0f);
      double newthdot = thdot + (3 * g / (2 * l) * sin(t0) + 3.0 / (m * sq_pow(l)) * u) * dt;
      if (newthdot < -8.0) newthdot = -8.0;
      if (newthdot > 8.0) newthdot = 8.0;
      double newth = t0 + newthdot * dt;
      s[0] = newth; s[1] = newthdot;
      *reward = -costs;
      return 0;
    }
    case NSG_ENV_ACROBOT: { /* rk4 over [0, dt] */
      double a = (double)(ai - 1); /* AVAIL_TORQUE = [-1, 0, +1] */
      double dt = th[0] - 0.0, dt2 = dt / 2.0;
      double y0[5] = {s[0], s[1], s[2], s[3], a}, k1[5], k2[5], k3[5], k4[5], y[5];
      acrobot_dsdt(th, y0, y0[4], k1);
      for (int k = 0; k < 5; k++) y[k] = y0[k] + dt2 * k1[k];
      acrobot_dsdt(th, y, y[4], k2);
      for (int k = 0; k < 5; k++) y[k] = y0[k] + dt2 * k2[k];
      acrobot_dsdt(th, y, y[4], k3);
      for (int k = 0; k < 5; k++) y[k] = y0[k] + dt * k3[k];
      acrobot_dsdt(th, y, y[4], k4);
      double ns[4];
      for (int k = 0; k < 4; k++) ns[k] = y0[k] + dt / 6.0 * (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]);
      for (int k = 0; k < 2; k++) ns[k] = wrap_pi(ns[k]);
      double mv1 = 4 * M_PI, mv2 = 9 * M_PI;
      ns[2] = fmin(fmax(ns[2], -mv1), mv1);
      ns[3] = fmin(fmax(ns[3], -mv2), mv2);
      for (int k = 0; k < 4; k++) s[k] = ns[k];
      int term = (-cos(s[0]) - cos(s[1] + s[0])) > 1.0;
      *reward = term ? 0.0 : -1.0;
      return term;
    }
    case NSG_ENV_MOUNTAINCAR: { /* th: gravity force */
      double position = s[0], velocity = s[1];
      velocity += (double)(ai - 1) * th[1] + cos(3 * position) * (-th[0]);
      if (velocity < -0.07) velocity = -0.07;
      if (velocity > 0.07) velocity = 0.07;
      position += velocity;
      if (position < -1.2) position = -1.2;
      if (position > 0.6) position = 0.6;
      if (position == -1.2 && velocity < 0) velocity = 0;
      s[0] = position; s[1] = velocity;
      *reward = -1.0;
      return position >= 0.5 && velocity >= 0;
    }
    case NSG_ENV_MOUNTAINCAR_CONT: { /* th: power */
      double position = s[0], velocity = s[1];
      double a0 = (double)af;
      double force = fmin(fmax(a0, -1.0), 1.0);
      velocity += force * th[0] - 0.0025 * cos(3 * position);
      if (velocity > 0.07) velocity = 0.07;
      if (velocity < -0.07) velocity = -0.07;
      position += velocity;
      if (position > 0.6) position = 0.6;
      if (position < -1.2) position = -1.2;
      if (position == -1.2 && velocity < 0) velocity = 0;
      int term = position >= 0.45 && velocity >= 0;
      double r = 0;
      if (term) r = 100.0;
      r -= pow(a0, 2.0) * 0.1; /* math.pow(action[0], 2) */
      s[0] = (double)(float)position; s[1] = (double)(float)velocity; /* state kept as float32 upstream */
      *reward = r;
      return term;
    }
    default: *reward = 0; return 0;
  }
}

/* ------------------------------------------------------------------ grid envs (FrozenLake, CliffWalking, Bridge) */
/* initial distribution of param p: Bridge's P_right side starts from initial_prob[1] (toy_text.py:573-591) */
static const double* grid_initial(const nsg_config* cfg, int p) {
  return (cfg->env_type == NSG_ENV_BRIDGE && cfg->params[p].theta_slot == 2) ? cfg->initial_prob[1] : cfg->initial_prob[0];
}
static int grid_start_state(const nsg_config* cfg, const uint8_t* tables) {
  if (cfg->env_type == NSG_ENV_CLIFFWALKING) return (cfg->nrow - 1) * cfg->ncol; /* start_state_index = (3, 0) */
  if (cfg->env_type == NSG_ENV_BRIDGE) return 2 * cfg->ncol + 4;                  /* envs/Bridge.py:110 */
  const uint8_t* desc = tables + cfg->desc_tab_off;
  for (int k = 0; k < cfg->nrow * cfg->ncol; k++)
    if (desc[k] == 'S') return k;
  return 0;
}

/* ------------------------------------------------------------------ wrapper-level reset */
static int letter_index(uint8_t c) { return c == 'S' ? 0 : c == 'F' ? 1 : c == 'H' ? 2 : 3; }

static void reset_one(const nsg_config* cfg, const uint8_t* tables, const nsg_buffers* b, int64_t N, int64_t i,
                      int has_seed, uint64_t seed) {
  int env = cfg->env_type, P = cfg->n_params;
  pcg64 g;
  if (has_seed) pcg_seed(&g, seed, -1); /* gymnasium Env.reset(seed) -> np_random(seed) */
  else rng_load(b->rng_env, N, i, &g);
  if (is_grid_env(env)) {
    /* FrozenLakeEnv / CliffWalkingEnv.reset: categorical_sample(initial_state_distrib) consumes one
       random(); the one-hot distribution always yields the start cell.  Bridge.reset draws nothing
       (envs/Bridge.py:103-111). */
    if (env != NSG_ENV_BRIDGE) {
      double r = pcg_double(&g);
      (void)r; /* argmax(cumsum(one-hot) > r) is the start cell for every r in [0, 1) */
    }
    b->cell[i] = grid_start_state(cfg, tables);
    if (b->prob) b->prob[i] = 1.0f;
  } else {
    double s[4] = {0, 0, 0, 0};
    env_reset_draw(env, &g, s);
    for (int k = 0; k < PHYS_DIM[env]; k++) b->phys[k * N + i] = s[k];
    env_obs(env, s, b->obs + i * OBS_DIM[env]);
  }
  rng_store(b->rng_env, N, i, &g);
  b->t[i] = 0; /* base.py:379 */
  if (b->t_fork) b->t_fork[i] = 0; /* a reset planning copy: TimeLimit.reset() zeroes its elapsed count too */
  int persistent = (cfg->flags & NSG_F_PERSISTENT_PARAMS) != 0;
  for (int p = 0; p < P; p++) {
    const nsg_param_cfg* pc = &cfg->params[p];
    if (!persistent) { /* base.py:381-384 deepcopy(init_initial_params); classic_control.py:105-107 */
      if (is_grid_env(env)) {
        const int nd = n_dist(env);
        const double* ini = grid_initial(cfg, p);
        for (int k = 0; k < nd; k++) b->theta[(p * nd + k) * N + i] = ini[k]; /* toy_text.py:207,396,659-664;
           table_prob is NOT restored: the wrapper's self.P survives reset (toy_text.py:187,365-367) */
      } else
        b->theta[p * N + i] = cfg->base_theta[pc->theta_slot];
      if (b->cursor) b->cursor[p * N + i] = 0;
      if (sched_is_stochastic(pc->sched_kind)) { /* the scheduler is rewound with the rest of init_initial_params */
        pcg64 sr;
        int32_t nx;
        sched_construct(pc, p, i, &sr, &nx);
        rng_store(b->rng_sched + (int64_t)p * 4 * N, N, i, &sr);
        b->sched_next[p * N + i] = nx;
      }
    }
    if (pc->upd_kind == NSG_UPD_D_LCBOUNDED) { /* the wrapper cannot see the inner fn's rng: rewound with the deepcopy,
         never re-seeded (base.py:151-158,381-391) */
      if (pc->uses_rng && !persistent) {
        pcg64 r;
        if (pc->has_fn_seed) pcg_seed(&r, pc->fn_seed, -1);
        else pcg_seed(&r, (uint64_t)i, 1000 + p);
        rng_store(b->rng_upd + (int64_t)p * 4 * N, N, i, &r);
      }
    } else
    if (pc->uses_rng && has_seed) { /* base.py:386-388,412-421: SeedSequence(seed).spawn(P)[j] */
      pcg64 r;
      pcg_seed(&r, seed, pc->rng_child);
      rng_store(b->rng_upd + (int64_t)p * 4 * N, N, i, &r);
    } /* no seed: streams continue (transplant, base.py:389-391) */
    b->env_change[p * N + i] = 0;
    b->delta_change[p * N + i] = 0.0f;
  }
  b->reward[i] = 0.0f;
  b->terminated[i] = 0;
  b->truncated[i] = 0;
  b->status[i] = 0;
  if (cfg->flags & NSG_F_TRACK_RETURNS) { b->ep_return[i] = 0.0f; b->ep_length[i] = 0; }
}

/* initial (pre-first-reset) streams of stochastic update fns: default_rng(fn_seed) */
int orc_init_streams(const nsg_config* cfg, const nsg_buffers* b, int64_t N, const uint64_t* entropy) {
  if (is_grid_env(cfg->env_type)) { /* __init__ builds P from initial_prob_dist (toy_text.py:82-84,337-340) */
    const int nd = n_dist(cfg->env_type);
    for (int64_t i = 0; i < N; i++) {
      for (int p = 0; p < cfg->n_params; p++)
        for (int k = 0; k < nd; k++) b->theta[(p * nd + k) * N + i] = grid_initial(cfg, p)[k];
      if (b->table_prob)
        for (int k = 0; k < nd; k++) b->table_prob[k * N + i] = cfg->initial_prob[0][k];
    }
  }
  /* construction-time θ (what persistent_params keeps across resets) and fresh list cursors */
  for (int p = 0; p < cfg->n_params; p++)
    for (int64_t i = 0; i < N; i++) {
      if (!is_grid_env(cfg->env_type)) b->theta[p * N + i] = cfg->base_theta[cfg->params[p].theta_slot];
      if (b->cursor) b->cursor[p * N + i] = 0;
    }
  for (int p = 0; p < cfg->n_params; p++) {
    const nsg_param_cfg* pc = &cfg->params[p];
    if (sched_is_stochastic(pc->sched_kind))
      for (int64_t i = 0; i < N; i++) {
        pcg64 sr;
        int32_t nx;
        sched_construct(pc, p, i, &sr, &nx);
        rng_store(b->rng_sched + (int64_t)p * 4 * N, N, i, &sr);
        b->sched_next[p * N + i] = nx;
      }
    if (!pc->uses_rng) continue;
    for (int64_t i = 0; i < N; i++) {
      pcg64 r;
      if (pc->has_fn_seed) pcg_seed(&r, pc->fn_seed, -1);
      else pcg_seed(&r, entropy ? entropy[i] : (uint64_t)i, 1000 + p);
      rng_store(b->rng_upd + (int64_t)p * 4 * N, N, i, &r);
    }
  }
  return 0;
}

int orc_reset(const nsg_config* cfg, const uint8_t* tables, const nsg_buffers* b, int64_t N,
              const uint64_t* seeds, const uint8_t* mask) {
  for (int64_t i = 0; i < N; i++) {
    if (mask && !mask[i]) continue;
    reset_one(cfg, tables, b, N, i, seeds != NULL, seeds ? seeds[i] : 0);
  }
  return 0;
}

/* ------------------------------------------------------------------ wrapper-level step */
static void frozenlake_move(const nsg_config* cfg, int row, int col, int a, int* nr, int* nc) {
  /* toy_text.py:449-458 inc() */
  if (a == 0) col = col - 1 > 0 ? col - 1 : 0;
  else if (a == 1) row = row + 1 < cfg->nrow - 1 ? row + 1 : cfg->nrow - 1;
  else if (a == 2) col = col + 1 < cfg->ncol - 1 ? col + 1 : cfg->ncol - 1;
  else if (a == 3) row = row - 1 > 0 ? row - 1 : 0;
  *nr = row; *nc = col;
}

/* what the last step_one of this thread paid, in the base MDP's own float64 (the reward row is float32), and whether it was a
   transition at all (a pending autoreset takes no action): the closed loops of orc_rollout_policy sum THIS, like the reference's
   Python loops sum the env's Python float */
static __thread double g_reward64;
static __thread int g_took_step;

static void step_one(const nsg_config* cfg, const uint8_t* tables, const nsg_buffers* b, int64_t N, int64_t i,
                     const void* actions, uint64_t* cnt) {
  int env = cfg->env_type, P = cfg->n_params;
  g_reward64 = 0.0;
  g_took_step = 0;
  /* NSG_F_NO_AUTORESET: the reference's single wrappers forward every step() to gymnasium whatever `done` said (base.py:313);
     status bit 0 then means "terminated at an earlier step of this episode" (CartPole's steps_beyond_terminated [UPSTREAM]) */
  const int noauto = (cfg->flags & NSG_F_NO_AUTORESET) != 0;
  if (!noauto && (b->status[i] & NSG_ST_NEEDS_RESET)) { /* next-step autoreset == env.reset() with no seed */
    reset_one(cfg, tables, b, N, i, 0, 0);
    if (b->done_bits) b->done_bits[i >> 6] &= ~(1ULL << (i & 63));
    return;
  }
  int t = b->t[i];
  double reward = 0;
  int term = 0;
  const int sim = (cfg->flags & NSG_F_SIM_ENV) != 0;
  const int theta_live = !(sim && !(cfg->flags & NSG_F_IN_SIM_CHANGE));
  if (is_grid_env(env)) {
    const int nd = n_dist(env);
    /* ---- θ: every distribution param (toy_text.py:178-185, 362-366, 605-631) ---- */
    for (int p = 0; p < P; p++) {
      const nsg_param_cfg* pc = &cfg->params[p];
      double pp[NSG_ND_MAX], q[NSG_ND_MAX];
      for (int k = 0; k < nd; k++) pp[k] = b->theta[(p * nd + k) * N + i];
      int fired = theta_live && fire_param(cfg, tables, b, N, i, p, t); /* frozen planning copy: toy_text.py:170-176,354-360,636-645 */
      double delta = 0.0;
      if (fired) {
        pcg64 ur;
        if (pc->uses_rng) rng_load(b->rng_upd + (int64_t)pc->fn_slot * 4 * N, N, i, &ur);
        upd_dist(pc, tables, pp, nd, t, b->cursor ? &b->cursor[pc->fn_slot * N + i] : NULL, pc->uses_rng ? &ur : NULL, q);
        if (pc->uses_rng) rng_store(b->rng_upd + (int64_t)pc->fn_slot * 4 * N, N, i, &ur);
        delta = w1_n(pp, q, nd); /* base.py:192-203 */
        for (int k = 0; k < nd; k++) {
          b->theta[(p * nd + k) * N + i] = q[k];
          if (b->table_prob) b->table_prob[k * N + i] = q[k]; /* P rebuilt / re-weighted only on a fire (:181-182,:365-366) */
        }
      }
      b->env_change[p * N + i] = (uint8_t)fired;
      b->delta_change[p * N + i] = (float)delta;
      if (fired) cnt[NSG_CNT_FIRED * NSG_CNT_SHARDS]++;
      if (pc->upd_kind == NSG_UPD_D_LCBOUNDED && theta_live) b->cursor[pc->fn_slot * N + i] = t + 1; /* prev_time = t, fired or not */
    }
    int a = ((const int32_t*)actions)[i];
    int s = b->cell[i];
    int row = s / cfg->ncol, col = s % cfg->ncol;
    pcg64 g;
    rng_load(b->rng_env, N, i, &g);
    double r = pcg_double(&g); /* one uniform per step from the env stream */
    rng_store(b->rng_env, N, i, &g);
    double prob = 1.0;
    if (env == NSG_ENV_FROZENLAKE) { /* gymnasium FrozenLakeEnv.step over the NS table (toy_text.py:426-444) */
      const uint8_t* desc = tables + cfg->desc_tab_off;
      double pt[3];
      for (int k = 0; k < 3; k++) pt[k] = b->table_prob[k * N + i];
      uint8_t letter = desc[s];
      if (letter == 'G' || letter == 'H') { /* :435-436 single self-loop entry (1.0, s, 0, True) */
        prob = 1.0; reward = 0; term = 1;
      } else {
        double c0 = pt[0], c1 = c0 + pt[1], c2 = c1 + pt[2];
        int idx = c0 > r ? 0 : c1 > r ? 1 : c2 > r ? 2 : 0; /* argmax of all-False is 0 */
        int dir = idx == 0 ? a : idx == 1 ? (a + 1) % 4 : (a + 3) % 4; /* [a, a+1, a-1] :438 */
        int nr, nc;
        frozenlake_move(cfg, row, col, dir, &nr, &nc);
        int ns = nr * cfg->ncol + nc;
        uint8_t nl = desc[ns];
        term = nl == 'G' || nl == 'H';
        reward = (cfg->flags & NSG_F_MODIFIED_REWARDS) ? cfg->letter_reward[letter_index(nl)] : (nl == 'G' ? 1.0 : 0.0);
        prob = pt[idx];
        s = ns;
      }
    } else if (env == NSG_ENV_CLIFFWALKING) { /* CliffWalkingEnv.step over the NS table (toy_text.py:86-148) */
      double pt[4], cs = 0.0;
      int idx = 0, found = 0;
      for (int k = 0; k < 4; k++) pt[k] = b->table_prob[k * N + i];
      for (int k = 0; k < 4; k++) { /* np.argmax(np.cumsum(p) > r); all-False -> 0 */
        cs = k == 0 ? pt[0] : cs + pt[k];
        if (!found && cs > r) { idx = k; found = 1; }
      }
      static const int off[4] = {0, 1, 3, 2}; /* b_actions = [a, a+1, a-1, a+2] (:96) */
      int dir = (a + off[idx]) % 4;
      static const int dr[4] = {-1, 0, 1, 0}, dc[4] = {0, 1, 0, -1}; /* UP RIGHT DOWN LEFT (:74-76) */
      int nr = row + dr[dir], nc = col + dc[dir];
      nr = nr < 0 ? 0 : nr > cfg->nrow - 1 ? cfg->nrow - 1 : nr;
      nc = nc < 0 ? 0 : nc > cfg->ncol - 1 ? cfg->ncol - 1 : nc;
      int cliff = nr == cfg->nrow - 1 && nc >= 1 && nc <= cfg->ncol - 2;
      int goal = nr == cfg->nrow - 1 && nc == cfg->ncol - 1;
      /* rewards: modified_rewards {S,F,H,G} -> letter_reward[0..3] (:117-125) */
      reward = cliff ? cfg->letter_reward[2] : goal ? cfg->letter_reward[3] : cfg->letter_reward[1];
      term = cliff ? ((cfg->flags & NSG_F_TERMINAL_CLIFF) != 0) : goal;
      s = cliff ? (cfg->nrow - 1) * cfg->ncol : nr * cfg->ncol + nc;
      prob = pt[idx];
    } else { /* Bridge.step (envs/Bridge.py:89-134): np.random.choice([a, a+1, a-1], p=P) */
      const uint8_t* desc = tables + cfg->desc_tab_off;
      double pt[3] = {cfg->initial_prob[0][0], cfg->initial_prob[0][1], cfg->initial_prob[0][2]};
      int want = 0; /* uniform mode: slot 0; split mode: P_left (col < ncol/2) or P_right (:149-158) */
      int split = 0;
      for (int p = 0; p < P; p++) split |= cfg->params[p].theta_slot != 0;
      if (split) {
        want = col < cfg->ncol / 2 ? 1 : 2;
        const double* ini = want == 2 ? cfg->initial_prob[1] : cfg->initial_prob[0];
        for (int k = 0; k < 3; k++) pt[k] = ini[k]; /* an omitted side stays at its initial value */
      }
      for (int p = 0; p < P; p++)
        if (cfg->params[p].theta_slot == want)
          for (int k = 0; k < 3; k++) pt[k] = b->theta[(p * 3 + k) * N + i];
      /* Generator/RandomState.choice(p=): cdf = cumsum(p); cdf /= cdf[-1]; idx = searchsorted(cdf, u, 'right') */
      double c0 = pt[0], c1 = c0 + pt[1], c2 = c1 + pt[2];
      int idx = (c0 / c2 <= r) + (c1 / c2 <= r) + (c2 / c2 <= r);
      if (idx > 2) idx = 2;
      int dir = idx == 0 ? a : idx == 1 ? (a + 1) % 4 : (a + 3) % 4;
      static const int dr[4] = {0, 1, 0, -1}, dc[4] = {-1, 0, 1, 0}; /* LEFT DOWN RIGHT UP (:13-16) */
      int nr = row + dr[dir], nc = col + dc[dir];
      if (nr < 0 || nr >= cfg->nrow || nc < 0 || nc >= cfg->ncol) { nr = row; nc = col; } /* out of bounds: stay (:127-128) */
      uint8_t nl = desc[nr * cfg->ncol + nc];
      if (nl == 'H') { reward = -1; term = 1; } else if (nl == 'G') { reward = 1; term = 1; } else { reward = 0; term = 0; }
      s = nr * cfg->ncol + nc;
      prob = pt[0];
    }
    b->cell[i] = s;
    if (b->prob) b->prob[i] = (float)prob;
  } else {
    int K = N_THETA[env];
    double cur[NSG_MAX_THETA], nv[NSG_MAX_THETA];
    for (int k = 0; k < K; k++) cur[k] = cfg->base_theta[k];
    unsigned tuned = 0, firedmask = 0;
    for (int p = 0; p < P; p++) cur[cfg->params[p].theta_slot] = b->theta[p * N + i];
    for (int k = 0; k < K; k++) nv[k] = cur[k];
    for (int p = 0; p < P; p++) { /* classic_control.py:80-85 */
      const nsg_param_cfg* pc = &cfg->params[p];
      int k = pc->theta_slot;
      tuned |= 1u << k;
      if (theta_live && fire_param(cfg, tables, b, N, i, p, t)) { /* frozen planning copy: classic_control.py:70-75 */
        pcg64 r;
        if (pc->uses_rng) rng_load(b->rng_upd + (int64_t)pc->fn_slot * 4 * N, N, i, &r);
        nv[k] = upd_scalar(pc, tables, cur[k], t, pc->uses_rng ? &r : NULL, b->cursor ? &b->cursor[pc->fn_slot * N + i] : NULL);
        if (pc->uses_rng) rng_store(b->rng_upd + (int64_t)pc->fn_slot * 4 * N, N, i, &r);
        firedmask |= 1u << p;
      }
    }
    unsigned viol = constraint_mask(env, nv, cur, tuned); /* :87 */
    double th[NSG_MAX_THETA + 2];
    for (int k = 0; k < K; k++) th[k] = cur[k];
    for (int p = 0; p < P; p++) { /* :87-92 */
      int k = cfg->params[p].theta_slot;
      int fired = (firedmask >> p) & 1;
      double delta = fired ? nv[k] - cur[k] : 0.0; /* base.py:182 */
      if ((viol >> k) & 1u) {
        if (fired) cnt[NSG_CNT_VIOLATION * NSG_CNT_SHARDS]++;
        fired = 0; delta = 0.0;
      } else {
        th[k] = nv[k];
      }
      if (b->violation) b->violation[p * N + i] = (uint8_t)((viol >> k) & 1u);
      b->theta[p * N + i] = th[k];
      b->env_change[p * N + i] = (uint8_t)fired;
      b->delta_change[p * N + i] = (float)delta;
      if (fired) cnt[NSG_CNT_FIRED * NSG_CNT_SHARDS]++;
    }
    if (env == NSG_ENV_CARTPOLE) {
      if (sim && !theta_live && b->derived) { /* frozen planning copy: resolver never runs again */
        th[6] = b->derived[0 * N + i];
        th[7] = b->derived[1 * N + i];
      } else { /* _dependency_resolver, classic_control.py:426-444 */
        th[6] = th[2] + th[1];
        th[7] = th[5] * th[2];
      }
    }
    double s[4];
    for (int k = 0; k < PHYS_DIM[env]; k++) s[k] = b->phys[k * N + i];
    int ai = 0; float af = 0;
    if (env == NSG_ENV_PENDULUM || env == NSG_ENV_MOUNTAINCAR_CONT) af = ((const float*)actions)[i];
    else ai = ((const int32_t*)actions)[i];
    term = env_step(env, th, s, ai, af, &reward);
    /* CartPoleEnv.step [UPSTREAM]: `elif self.steps_beyond_terminated is None: ... reward = 1.0  else: ... reward = 0.0` */
    if (noauto && env == NSG_ENV_CARTPOLE && term && (b->status[i] & NSG_ST_NEEDS_RESET)) reward = 0.0;
    for (int k = 0; k < PHYS_DIM[env]; k++) b->phys[k * N + i] = s[k];
    env_obs(env, s, b->obs + i * OBS_DIM[env]);
  }
  t += 1; /* base.py:314 */
  b->t[i] = t;
  /* TimeLimit [UPSTREAM] counts steps of ITS env: a planning copy restarts at the fork */
  int elapsed = t - ((sim && b->t_fork) ? b->t_fork[i] : 0);
  int trunc = cfg->max_episode_steps > 0 && elapsed >= cfg->max_episode_steps;
  b->reward[i] = (float)reward;
  g_reward64 = reward;
  g_took_step = 1;
  b->terminated[i] = (uint8_t)term;
  b->truncated[i] = (uint8_t)trunc;
  int done = term || trunc;
  b->status[i] = noauto ? (uint8_t)((b->status[i] & NSG_ST_NEEDS_RESET) | (term ? NSG_ST_NEEDS_RESET : 0)) : (done ? NSG_ST_NEEDS_RESET : 0);
  if (cfg->flags & NSG_F_TRACK_RETURNS) {
    float er = b->ep_return[i] + (float)reward;
    int el = b->ep_length[i] + 1;
    if (done) { b->last_return[i] = er; b->last_length[i] = el; er = 0.0f; el = 0; }
    b->ep_return[i] = er; b->ep_length[i] = el;
  }
  cnt[NSG_CNT_STEPS * NSG_CNT_SHARDS]++;
  if (b->done_bits) {
    if (done) b->done_bits[i >> 6] |= 1ULL << (i & 63);
    else b->done_bits[i >> 6] &= ~(1ULL << (i & 63));
  }
  if (done) cnt[NSG_CNT_DONE * NSG_CNT_SHARDS]++;
}

int orc_step(const nsg_config* cfg, const uint8_t* tables, const nsg_buffers* b, int64_t N, const void* actions) {
  uint64_t* cnt = b->counters; /* running totals, shard 0 */
  for (int64_t i = 0; i < N; i++) step_one(cfg, tables, b, N, i, actions, cnt);
  return 0;
}

/* range version for multi-threaded baselines (counters are per-call scratch of the caller) */
int orc_step_range(const nsg_config* cfg, const uint8_t* tables, const nsg_buffers* b, int64_t N, const void* actions,
                   int64_t lo, int64_t hi, uint64_t* cnt) {
  for (int64_t i = lo; i < hi; i++) step_one(cfg, tables, b, N, i, actions, cnt);
  return 0;
}

/* ---- closed loops: the reference's step consumers, restated --------------------------------------------------------------------
 *   MCTS._default_policy       benchmark_algorithms/MCTS.py:162-181   while not terminated and depth < d and not truncated:
 *                                                                      action = np.random.choice(actions); step; tot_reward += reward * gamma ** depth
 *   run_episode                evaluate/run_experiment.py:108-129     while not done and not truncated: act, step, total_reward += reward
 *   the tutorial's run_episode tutorial.ipynb cell 12                  action = policy[observation]
 * One env at a time, one step at a time: decide from the env's last observation (the float32 obs row / the cell), step, add.
 * Mirrors nsg_rollout_policy's contract (include/nsgym_hip.h): `discount[j]` is gamma ** j as the CALLER's pow computed it, and a
 * pending autoreset takes no action and changes no account.  reward64_out / actions_out: [K][N] or NULL. */
static uint64_t pol_mix64(uint64_t x) { /* splitmix64's finaliser */
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27; x *= 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
uint64_t orc_policy_bits(uint64_t seed, uint64_t env, uint64_t step) {
  return pol_mix64(pol_mix64(seed + 0x9E3779B97F4A7C15ULL * (env + 1)) + 0xD1B54A32D192ED03ULL * (step + 1));
}
int orc_rollout_policy(const nsg_config* cfg, const uint8_t* tables, const nsg_buffers* b, int64_t N, const nsg_policy* pol, int k_steps,
                       const nsg_episode_acc* acc, double* reward64_out, uint8_t* took_out) {
  static const float ACT_LO[NSG_ENV_COUNT] = {0, -2.f, 0, 0, -1.f, 0, 0, 0}, ACT_HI[NSG_ENV_COUNT] = {0, 2.f, 0, 0, 1.f, 0, 0, 0};
  static const int N_ACT[NSG_ENV_COUNT] = {2, 0, 3, 3, 0, 4, 4, 4};
  const int env = cfg->env_type, D = OBS_DIM[env];
  const int fa = env == NSG_ENV_PENDULUM || env == NSG_ENV_MOUNTAINCAR_CONT;
  int32_t* ai = (int32_t*)calloc((size_t)N, sizeof(int32_t));
  float* af = (float*)ai;
  if (!ai) return -1;
  uint64_t* cnt = b->counters;
  for (int k = 0; k < k_steps; k++) {
    for (int64_t i = 0; i < N; i++) {
      switch (pol->kind) {
        case NSG_POL_TABLE:
          if (fa) af[i] = ((const float*)pol->data)[(int64_t)k * N + i]; else ai[i] = ((const int32_t*)pol->data)[(int64_t)k * N + i];
          break;
        case NSG_POL_UNIFORM: {
          const uint64_t bits = orc_policy_bits(pol->seed, (uint64_t)(pol->index0 + i), (uint64_t)(uint32_t)(pol->step0 + k));
          if (fa) af[i] = ACT_LO[env] + (ACT_HI[env] - ACT_LO[env]) * ((float)(bits >> 40) * 5.9604644775390625e-08f);
          else ai[i] = (int32_t)(((bits >> 32) * (uint64_t)N_ACT[env]) >> 32);
        } break;
        case NSG_POL_BY_STATE: ai[i] = ((const int32_t*)pol->data)[b->cell[i]]; break;
        default: { /* NSG_POL_LINEAR: float32, bias first, then the terms in order */
          const float* W = (const float*)pol->data;
          const float* o = b->obs + i * D;
          if (fa) {
            float sc = W[D];
            for (int d = 0; d < D; d++) sc = sc + W[d] * o[d];
            af[i] = sc < ACT_LO[env] ? ACT_LO[env] : sc > ACT_HI[env] ? ACT_HI[env] : sc;
          } else {
            int best = 0; float top = 0.f;
            for (int j = 0; j < N_ACT[env]; j++) {
              float sc = W[j * (D + 1) + D];
              for (int d = 0; d < D; d++) sc = sc + W[j * (D + 1) + d] * o[d];
              if (j == 0 || sc > top) { top = sc; best = j; }
            }
            ai[i] = best;
          }
        }
      }
      if (pol->actions_out) {
        if (fa) ((float*)pol->actions_out)[(int64_t)k * N + i] = af[i]; else ((int32_t*)pol->actions_out)[(int64_t)k * N + i] = ai[i];
      }
      step_one(cfg, tables, b, N, i, ai, cnt);
      if (reward64_out) reward64_out[(int64_t)k * N + i] = g_reward64;
      if (took_out) took_out[(int64_t)k * N + i] = (uint8_t)g_took_step;
      if (acc && g_took_step) {
        const int alive = acc->alive ? acc->alive[i] != 0 : 1;
        if (alive) {
          const int len = acc->length ? acc->length[i] : 0;
          const double g = (acc->discount && len < acc->n_discount) ? acc->discount[len] : 1.0;
          if (acc->ret) acc->ret[i] = acc->ret[i] + g_reward64 * g; /* tot_reward += reward * gamma ** depth */
          if (acc->length) acc->length[i] = len + 1;
          if (acc->alive && (b->terminated[i] || b->truncated[i])) acc->alive[i] = 0;
        }
      }
    }
  }
  free(ai);
  return 0;
}

/* get_planning_env() / __deepcopy__ for all envs (classic_control.py:120-186, toy_text.py:471-511) */
int orc_fork(const nsg_config* scfg, const nsg_buffers* sb, const nsg_config* dcfg, const nsg_buffers* db, int64_t N,
             uint64_t entropy, int theta_mode) {
  int env = scfg->env_type, P = scfg->n_params;
  const int fl = is_grid_env(env);
  const int nd = n_dist(env);
  const int in_sim_change = (dcfg->flags & NSG_F_IN_SIM_CHANGE) != 0;
  for (int64_t i = 0; i < N; i++) {
    for (int k = 0; k < PHYS_DIM[env]; k++) db->phys[k * N + i] = sb->phys[k * N + i];
    if (fl) db->cell[i] = sb->cell[i];
    db->t[i] = sb->t[i];
    db->t_fork[i] = sb->t[i];
    /* NSG_F_NO_AUTORESET: the copy wraps a NEW base env that was reset (classic_control.py:168-178): not terminated yet */
    db->status[i] = (dcfg->flags & NSG_F_NO_AUTORESET) ? 0 : sb->status[i];
    for (int r = 0; r < (fl ? nd * P : P); r++) {
      double cur = sb->theta[r * N + i];
      double init = fl ? grid_initial(scfg, r / nd)[r % nd] : scfg->base_theta[scfg->params[r].theta_slot];
      db->theta[r * N + i] = theta_mode == 1 ? init : cur;
    }
    if (env == NSG_ENV_CARTPOLE && db->derived) { /* sim_env._dependency_resolver() at copy time, :183 */
      double cur[NSG_MAX_THETA];
      for (int k = 0; k < 6; k++) cur[k] = scfg->base_theta[k];
      for (int p = 0; p < P; p++) cur[scfg->params[p].theta_slot] = sb->theta[p * N + i];
      db->derived[0 * N + i] = cur[2] + cur[1];
      db->derived[1 * N + i] = cur[5] * cur[2];
    }
    if (fl && db->table_prob) { /* which P table the copy steps with (see include/nsgym_hip.h nsg_fork):
         FrozenLake   toy_text.py:479-480,505-508 + :365-367: an in_sim_change copy re-installs ITS OWN table,
                      built from initial_prob_dist by its constructor;
         CliffWalking toy_text.py:219-221,246-249 + :187: the copy's own table IS the copied current one, so an
                      in_sim_change planning copy steps with the current table although θ reads initial */
      /* A copy OF A COPY (MCTS.search deep-copies the planning env it is given, MCTS.py:131) takes the source copy's OWN table
         (`deepcopy(self.P)`, toy_text.py:508 / 246-249), not the one the source steps with (`unwrapped.P`):
         FrozenLake   a copy's own table is the constructor's, built from initial_prob_dist: the second copy steps with THAT;
         CliffWalking a copy's own table is the first source's current one, also after get_planning_env() without delta
                      notification overwrote the base env's table with the initial one: the second copy steps with the CURRENT
                      table again.  The own table of a frozen CliffWalking copy is kept in buffers.derived ([4][N]). */
      const int src_sim = (scfg->flags & NSG_F_SIM_ENV) != 0;
      if (env == NSG_ENV_FROZENLAKE) {
        int use_initial = in_sim_change || theta_mode == 1 || src_sim;
        for (int k = 0; k < nd; k++)
          db->table_prob[k * N + i] = use_initial ? scfg->initial_prob[0][k] : sb->table_prob[k * N + i];
      } else {
        const int own_in_derived = src_sim && !in_sim_change && sb->derived;
        const int use_initial = theta_mode == 1 && !in_sim_change;
        for (int k = 0; k < nd; k++) {
          const double own = own_in_derived ? sb->derived[k * N + i] : sb->table_prob[k * N + i];
          if (db->derived) db->derived[k * N + i] = own;
          db->table_prob[k * N + i] = use_initial ? scfg->initial_prob[0][k] : own;
        }
      }
    }
    for (int p = 0; p < P; p++) {
      if (db->cursor && sb->cursor) db->cursor[p * N + i] = sb->cursor[p * N + i]; /* deepcopy(tunable_params) */
      if (sched_is_stochastic(scfg->params[p].sched_kind)) { /* scheduler state is copied, not re-seeded */
        for (int k = 0; k < 4; k++) db->rng_sched[((int64_t)p * N + i) * 4 + k] = sb->rng_sched[((int64_t)p * N + i) * 4 + k];
        db->sched_next[p * N + i] = sb->sched_next[p * N + i];
      }
      db->env_change[p * N + i] = sb->env_change[p * N + i];
      db->delta_change[p * N + i] = sb->delta_change[p * N + i];
      if (scfg->params[p].uses_rng && scfg->params[p].upd_kind == NSG_UPD_D_LCBOUNDED) { /* inner rng: deep-copied */
        for (int k = 0; k < 4; k++) db->rng_upd[((int64_t)p * N + i) * 4 + k] = sb->rng_upd[((int64_t)p * N + i) * 4 + k];
      } else if (scfg->params[p].uses_rng) { /* _reseed_planning_env_rngs: fresh entropy */
        pcg64 r;
        pcg_seed(&r, entropy + (uint64_t)i, 7100 + p);
        rng_store(db->rng_upd + (int64_t)p * 4 * N, N, i, &r);
      }
    }
    pcg64 g; /* the copy's base env is a new gym.make(): unseeded np_random */
    pcg_seed(&g, entropy + (uint64_t)i, 7001);
    rng_store(db->rng_env, N, i, &g);
    if (!fl) for (int k = 0; k < OBS_DIM[env]; k++) db->obs[i * OBS_DIM[env] + k] = sb->obs[i * OBS_DIM[env] + k];
    db->reward[i] = sb->reward[i];
    db->terminated[i] = sb->terminated[i];
    db->truncated[i] = sb->truncated[i];
    if (db->prob && sb->prob) db->prob[i] = sb->prob[i];
    if (dcfg->flags & NSG_F_TRACK_RETURNS) {
      db->ep_return[i] = (scfg->flags & NSG_F_TRACK_RETURNS) ? sb->ep_return[i] : 0.0f;
      db->ep_length[i] = 0; db->last_return[i] = 0.0f; db->last_length[i] = 0;
    }
  }
  return 0;
}

int orc_seed_streams(const nsg_config* cfg, const nsg_buffers* b, int64_t N, const uint64_t* seeds, int which) {
  for (int64_t i = 0; i < N; i++) {
    if (which == 0) {
      pcg64 g;
      pcg_seed(&g, seeds[i], -1);
      rng_store(b->rng_env, N, i, &g);
    } else {
      for (int p = 0; p < cfg->n_params; p++)
        if (cfg->params[p].uses_rng) {
          pcg64 r;
          pcg_seed(&r, seeds[i], cfg->params[p].rng_child);
          rng_store(b->rng_upd + (int64_t)p * 4 * N, N, i, &r);
        }
    }
  }
  return 0;
}

/* multi-threaded variant for the CPU baseline leg of bench.py (OpenMP over env ranges) */
int orc_step_mt(const nsg_config* cfg, const uint8_t* tables, const nsg_buffers* b, int64_t N, const void* actions,
                int nthreads) {
  if (nthreads < 1) nthreads = 1;
  uint64_t* total = b->counters;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int th = 0; th < nthreads; th++) {
    uint64_t* cnt = (uint64_t*)calloc((size_t)NSG_CNT_COUNT * NSG_CNT_SHARDS, sizeof(uint64_t));
    int64_t per = ((N + nthreads - 1) / nthreads + 63) & ~63LL; /* 64-aligned: done_bits words stay thread-private */
    int64_t lo = th * per, hi = lo + per < N ? lo + per : N;
    for (int64_t i = lo; i < hi; i++) step_one(cfg, tables, b, N, i, actions, cnt);
    for (int c = 0; c < NSG_CNT_COUNT; c++)
      __atomic_fetch_add(&total[c * NSG_CNT_SHARDS], cnt[c * NSG_CNT_SHARDS], __ATOMIC_RELAXED);
    free(cnt);
  }
  return 0;
}

/* θ-engine alone: mirrors nsg_theta_trace */
int orc_theta_trace(const nsg_config* cfg, const uint8_t* tables, int p, int n, int t0, int T, const double* theta0,
                    uint64_t* rng_state, double* theta_out, uint8_t* fired_out, double* delta_out) {
  const nsg_param_cfg* pc = &cfg->params[p];
  int dist = pc->upd_kind >= NSG_UPD_D_INCREMENT;
  for (int i = 0; i < n; i++) {
    pcg64 r;
    if (pc->uses_rng && rng_state) rng_load(rng_state, n, i, &r);
    int32_t cursor = 0, snext = 0;
    pcg64 sr;
    if (sched_is_stochastic(pc->sched_kind)) sched_construct(pc, p, i, &sr, &snext);
    const int nd = n_dist(cfg->env_type);
    double th[NSG_ND_MAX] = {0, 0, 0, 0};
    if (dist) for (int c = 0; c < nd; c++) th[c] = theta0[nd * i + c];
    else th[0] = theta0[i];
    for (int k = 0; k < T; k++) {
      int t = t0 + k;
      int fired = sched_is_stochastic(pc->sched_kind) ? sched_fire_stoch(pc, t, &sr, &snext) : sched_fire(pc, tables, t);
      double delta = 0.0;
      if (fired) {
        if (dist) {
          double q[NSG_ND_MAX];
          upd_dist(pc, tables, th, nd, t, &cursor, pc->uses_rng ? &r : NULL, q);
          delta = w1_n(th, q, nd);
          for (int c = 0; c < nd; c++) th[c] = q[c];
        } else {
          double nvv = upd_scalar(pc, tables, th[0], t, pc->uses_rng ? &r : NULL, &cursor);
          delta = nvv - th[0];
          th[0] = nvv;
        }
      }
      if (pc->upd_kind == NSG_UPD_D_LCBOUNDED) cursor = t + 1;
      if (dist) for (int c = 0; c < nd; c++) theta_out[((int64_t)k * nd + c) * n + i] = th[c];
      else theta_out[(int64_t)k * n + i] = th[0];
      fired_out[(int64_t)k * n + i] = (uint8_t)fired;
      delta_out[(int64_t)k * n + i] = delta;
    }
    if (pc->uses_rng && rng_state) rng_store(rng_state, n, i, &r);
  }
  return 0;
}

/* NumPy-compatible streams: mirrors nsg_rng_fill */
int orc_rng_fill(int kind, const uint64_t* seeds, int n, int spawn_key, int count, void* out, uint64_t* state_out) {
  for (int i = 0; i < n; i++) {
    pcg64 r;
    pcg_seed(&r, seeds[i], spawn_key);
    if (state_out) rng_store(state_out, n, i, &r);
    for (int k = 0; k < count; k++) {
      if (kind == 0) ((uint64_t*)out)[(int64_t)k * n + i] = pcg_next64(&r);
      else if (kind == 1) ((double*)out)[(int64_t)k * n + i] = pcg_double(&r);
      else ((double*)out)[(int64_t)k * n + i] = pcg_std_normal(&r);
    }
  }
  return 0;
}

/* extra stream KATs: kind 3 = standard_exponential; orc_geometric / orc_dirichlet below */
int orc_exponential(uint64_t seed, int count, double* out) {
  pcg64 r;
  pcg_seed(&r, seed, -1);
  for (int k = 0; k < count; k++) out[k] = pcg_std_exponential(&r);
  return 0;
}
int orc_geometric(uint64_t seed, double p, int count, int64_t* out) {
  pcg64 r;
  pcg_seed(&r, seed, -1);
  for (int k = 0; k < count; k++) out[k] = pcg_geometric(&r, p);
  return 0;
}
int orc_dirichlet_ones(uint64_t seed, int n, int count, double* out) {
  pcg64 r;
  pcg_seed(&r, seed, -1);
  for (int c = 0; c < count; c++) {
    double acc = 0.0;
    for (int k = 0; k < n; k++) { out[c * n + k] = pcg_std_exponential(&r); acc = acc + out[c * n + k]; }
    double inv = 1.0 / acc;
    for (int k = 0; k < n; k++) out[c * n + k] = out[c * n + k] * inv;
  }
  return 0;
}

size_t orc_sizeof_config(void) { return sizeof(nsg_config); }
size_t orc_sizeof_buffers(void) { return sizeof(nsg_buffers); }
