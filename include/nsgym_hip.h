/* nsgym_hip.h — C-ABI of the MI355X-native vectorised non-stationary env stepper.
 *
 * The reference (scope-lab-vu/ns_gym) is 100 % Python and has NO FFI/plugin ABI: its
 * boundary is the gymnasium Python protocol (ns_gym/base.py:206,248).  This header is
 * therefore the boundary a maintainer would bind with ctypes (INTEGRATION.md shows the
 * stub); each entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every function returns 0 on success or a negative NSG_E* code; the message is in
 *     nsg_last_error() (thread-local).  Nothing throws across the ABI.
 *   - the CALLER owns every per-env device buffer (PyTorch tensors' data_ptr()); the
 *     library owns only the small constant tables it uploads at nsg_create().
 *   - all launches are enqueue-only on the caller's hipStream_t (passed as void*);
 *     no internal synchronisation, graph-capturable.
 *   - one host thread per handle.
 *
 * Data layout: struct-of-arrays over N parallel env instances.  Field f of env i
 * lives at base[f * N + i] (coalesced across a wavefront).
 */
#ifndef NSGYM_HIP_H
#define NSGYM_HIP_H

#ifndef __HIPCC_RTC__  /* hiprtc (config-specialised builds) predefines the fixed-width types */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define NSG_ABI_VERSION 4 /* 4: NSG_F_NO_AUTORESET; the grid status byte carries a table hint (NSG_ST_TABLE_*), Pendulum's episode
                             word no needs-reset bit (round 3, not re-versioned then); nsg_table_prob_dirty */
#define NSG_MAX_PARAMS 8 /* tunable params per env (Acrobot has 8: ns_gym/base.py:622-631) */
#define NSG_MAX_THETA 8
#define NSG_MAX_SEGMENTS 8 /* env-type segments of one heterogeneous launch */

/* error codes */
#define NSG_OK 0
#define NSG_EINVAL (-22)
#define NSG_ENOMEM (-12)
#define NSG_EHIP (-5)
#define NSG_ENOTBOUND (-77)
#define NSG_EUNSUPPORTED (-95) /* config specialisation unavailable (no hiprtc) or its compilation failed */

/* ---- env types; θ slot order = ATTRIBUTE_MAP order (ns_gym/base.py:611-635) ---------- */
enum {
  NSG_ENV_CARTPOLE = 0,          /* gravity masscart masspole force_mag tau length            */
  NSG_ENV_PENDULUM = 1,          /* m l dt g                                                  */
  NSG_ENV_ACROBOT = 2,           /* dt LINK_LENGTH_1 LINK_LENGTH_2 LINK_MASS_1 LINK_MASS_2
                                    LINK_COM_POS_1 LINK_COM_POS_2 LINK_MOI                    */
  NSG_ENV_MOUNTAINCAR = 3,       /* gravity force                                             */
  NSG_ENV_MOUNTAINCAR_CONT = 4,  /* power                                                     */
  NSG_ENV_FROZENLAKE = 5,        /* P  (3-way slip distribution, ns_gym/wrappers/toy_text.py:265-519) */
  NSG_ENV_CLIFFWALKING = 6,      /* P  (4-way slip [a,a+1,a-1,a+2], toy_text.py:14-262)        */
  NSG_ENV_BRIDGE = 7,            /* P | P_left, P_right (3-way slip, toy_text.py:524-715,
                                    ns_gym/envs/Bridge.py); θ slots 0 = P, 1 = P_left, 2 = P_right */
  NSG_ENV_COUNT = 8
};

/* ---- schedulers (ns_gym/schedulers.py; range gate ns_gym/base.py:67-81) -------------- */
enum {
  NSG_SCHED_CONTINUOUS = 0, /* schedulers.py:46-53                                          */
  NSG_SCHED_PERIODIC = 1,   /* :77-89   i0 = period                                         */
  NSG_SCHED_BURST = 2,      /* :119-140 i0 = on, i1 = off                                   */
  NSG_SCHED_TABLE = 3,      /* Discrete :56-74, Window :180-198, Custom :31-43 compiled to a
                               bit table over t (tab_off, tab_len bits; i0 = value beyond the table:
                               0 / 1, or 2 = unknown - a sampled callable: no fire, NSG_CNT_SCHED_OVERRUN) */
  NSG_SCHED_RANDOM = 4,     /* :9-28    p0 = probability, own PCG64 stream                  */
  NSG_SCHED_DECAYING = 5,   /* :143-177 p0 = initial probability, p1 = decay rate            */
  NSG_SCHED_MEMORYLESS = 6  /* :92-116  p0 = p (geometric gaps)                             */
};

/* ---- scalar update functions (ns_gym/update_functions/single_param.py) --------------- */
enum {
  NSG_UPD_INCREMENT = 0,      /* :154-175 u0=k                    */
  NSG_UPD_DECREMENT = 1,      /* :178-199 u0=k                    */
  NSG_UPD_TREND = 2,          /* :20-40   u0=slope                */
  NSG_UPD_POLY = 3,           /* :451-473 coeffs in value table   */
  NSG_UPD_GEOMETRIC = 4,      /* :290-307 u0=r                    */
  NSG_UPD_EXPDECAY = 5,       /* :266-287 u0=decay_rate           */
  NSG_UPD_OSCILLATING = 6,    /* :243-264 u0=delta                */
  NSG_UPD_SIGMOID = 7,        /* :349-385 u0=a u1=b u2=k u3=t0    */
  NSG_UPD_LERP = 8,           /* :476-508 u0=start u1=end u2=T    */
  NSG_UPD_STEPWISE = 9,       /* :202-223 values in value table   */
  NSG_UPD_CYCLIC = 10,        /* :388-408 values in value table   */
  NSG_UPD_NOUPDATE = 11,      /* :226-240                         */
  NSG_UPD_RANDOMWALK = 12,    /* :84-113  u0=mu u1=sigma          */
  NSG_UPD_RW_DRIFT = 13,      /* :116-151 u0=alpha u1=mu u2=sigma */
  NSG_UPD_RW_DRIFT_TREND = 14,/* :43-81   u0=alpha u1=mu u2=sigma u3=slope */
  NSG_UPD_OU = 15,            /* :310-346 u0=theta u1=mu u2=sigma */
  NSG_UPD_BOUNDED_RW = 16,    /* :411-448 u0=mu u1=sigma u2=lo u3=hi */
  /* distribution update functions (ns_gym/update_functions/distribution.py); n = 3 (FrozenLake,
     Bridge) or 4 (CliffWalking); vector constants are packed n at a time in u[] */
  NSG_UPD_D_INCREMENT = 32,   /* :41-67   u0=k                    */
  NSG_UPD_D_DECREMENT = 33,   /* :70-97   u0=k                    */
  NSG_UPD_D_STEPWISE = 34,    /* :100-130 triples in value table  */
  NSG_UPD_D_CYCLIC = 35,      /* :334-356 triples in value table  */
  NSG_UPD_D_NOUPDATE = 36,    /* :217-231                         */
  NSG_UPD_D_UNIFORMDRIFT = 37,/* :234-261 u0=rate                 */
  NSG_UPD_D_TARGETREV = 38,   /* :264-293 u[0..n)=target u[n]=theta */
  NSG_UPD_D_LERP = 39,        /* :296-331 u[0..n)=start u[n..2n)=end u[2n]=T */
  NSG_UPD_D_RANDOMCAT = 40,   /* :11-38   rng.dirichlet(ones(n)): n standard exponentials, normalised */
  NSG_UPD_D_LCBOUNDED = 41    /* :133-183 rejection-sample the inner update (u1: 0 = RandomCategorical, 1 = NoUpdate)
                                 until W1 <= u0 * |t - prev_time|, at most 1e5 tries; cursor row = prev_time + 1 */
};

/* flags of nsg_config.flags (constructor kwargs of NSWrapper, ns_gym/base.py:222-232) */
#define NSG_F_CHANGE_NOTIFICATION 0x1u
#define NSG_F_DELTA_NOTIFICATION 0x2u
#define NSG_F_PERSISTENT_PARAMS 0x4u
#define NSG_F_TRACK_RETURNS 0x8u   /* keep per-env episode return / length accumulators   */
#define NSG_F_MODIFIED_REWARDS 0x10u /* FrozenLake modified_rewards (toy_text.py:465-468)  */
#define NSG_F_VIOLATION_MASK 0x200u /* also write the per-(env,param) constraint-violation mask (buffers.violation) */
#define NSG_F_TERMINAL_CLIFF 0x100u /* CliffWalking terminal_cliff (toy_text.py:39,126-128)              */
#define NSG_F_SIM_ENV 0x40u        /* planning copy: is_sim_env (base.py:270, classic_control.py:184)   */
#define NSG_F_IN_SIM_CHANGE 0x80u  /* in_sim_change: θ keeps evolving inside planning copies            */
#define NSG_F_LIBM_EXACT 0x800u    /* classic-control envs: every sin / cos / scalar `** 2` of the integrators and every sin / exp of the θ-engine is
                                      evaluated with libm's own algorithm and roundings (glibc 2.35's FMA builds - what np.sin / np.cos / scalar
                                      power resolve to in the reference): float64 state, float32 observation, reward and θ then EQUAL the
                                      reference's bit for bit instead of tracking them within 1e-5.  Runs on the handle's specialised unit only
                                      (nsg_specialize; nsg_step / nsg_rollout / ... refuse an unspecialised handle with this flag).  A group
                                      launch (nsg_step_group / nsg_rollout_group) is exact when ALL its classic-control members carry the flag -
                                      it then needs the member list's own unit - and is refused when only some do.  Costs +4 % (CartPole) to x 1.9 (Acrobot) per step.  Added within ABI
                                      version 4: an older library rejects the unknown flag bit. */
#define NSG_F_KNOWN 0xfdfu         /* every flag bit this header defines; nsg_create refuses any other */
#define NSG_F_NO_AUTORESET 0x400u  /* a finished env is NOT reset by the next step: it keeps stepping exactly like the reference's single
                                      wrappers, which forward to gymnasium whatever `done` said (base.py:313): CartPole integrates on and pays
                                      0.0 from the second terminated step (steps_beyond_terminated), TimeLimit keeps reporting truncated,
                                      FrozenLake's terminal cell self-loops and still consumes its draw (toy_text.py:435-436), θ-engine and t
                                      run on.  Bit 0 of the episode word then means "terminated at an earlier step of this episode".  Only
                                      nsg_reset / nsg_reset_seeded start a new episode.  Not combinable with NSG_F_TRACK_RETURNS (episode
                                      accounting is defined by the autoreset) */

/* per-env status: grid envs keep a byte (buffers.status); classic-control envs keep an episode WORD (buffers.episode):
 * bit 0 = NSG_ST_NEEDS_RESET, bits 1-31 = how many resets the env has drawn from its np_random stream so far (= the index of
 * its next episode in that stream, see buffers.rng_env).  Pendulum never terminates: for a batch that is not a planning copy "needs
 * reset" IS t >= max_episode_steps, bit 0 of its word stays clear and a single step touches the word only in the lanes that reset */
#define NSG_ST_NEEDS_RESET 0x1u
#define NSG_EP_COUNT_SHIFT 1
/* grid envs, bits 1-7 of the status byte: WHICH distribution buffers.table_prob holds for this env, when the config already holds
 * it - a step then skips the table_prob rows (24-32 B per env-step).  The rows stay authoritative: every writer of table_prob
 * (construction, a fire, nsg_fork) also sets these bits, NSG_ST_TABLE_ROWS ("read the rows") is always valid, and a caller that
 * writes table_prob itself must clear them */
#define NSG_ST_TABLE_SHIFT 1
#define NSG_ST_TABLE_ROWS 0u      /* no statement: read buffers.table_prob                                       */
#define NSG_ST_TABLE_INITIAL 1u   /* == cfg.initial_prob[0] (initial_prob_dist)                                   */
#define NSG_ST_TABLE_LIST0 2u     /* 2 + j: == entry j of params[0]'s value list (DistributionStepWise / Cyclic)  */
#define NSG_ST_TABLE_MAX 127u
/* classic-control env streams: descriptor word 0 of buffers.rng_env (see there) */
#define NSG_STREAM_AFFINE (1ULL << 63)

/* device-side counters: uint64 running totals, one shard per wavefront slot of a launch.
 * buffers.counters is [NSG_CNT_COUNT][NSG_CNT_SHARDS]; a total is the sum over its shards.
 * Produced by wavefront ballots + popcounts; every wavefront adds its counts to the shard
 * (workgroup index within the handle's launch range x 4 + wavefront index) mod NSG_CNT_SHARDS with one
 * fire-and-forget atomic add: no workgroup barrier, no wait.  Up to NSG_CNT_SHARDS / 4 workgroups a shard
 * has exactly one owner per launch; the wider launches of very large batches share a shard between four. */
enum {
  NSG_CNT_DONE = 0,       /* episodes finished (terminated or truncated)                */
  NSG_CNT_FIRED = 1,      /* (env,param) updates applied (notification flags raised)    */
  NSG_CNT_VIOLATION = 2,  /* (env,param) updates rejected by the constraint checker     */
  NSG_CNT_STEPS = 3,      /* env transitions executed (autoreset lanes excluded)        */
  /* conditions for which the reference RAISES; a kernel cannot, so it counts them and the host turns a non-zero count into the
   * reference's exception when it polls (VecNSEnv.check_errors, the N = 1 adaptors after every step):                        */
  NSG_CNT_LC_EXHAUSTED = 4,   /* LCBoundedDistrubutionUpdate found no Lipschitz-continuous candidate in 1e5 tries: ValueError
                                 (ns_gym/update_functions/distribution.py:168-182); the distribution was left unchanged       */
  NSG_CNT_SCHED_OVERRUN = 5,  /* a CustomScheduler was asked about a t beyond the horizon its callable was sampled over
                                 (the reference calls event_function(t) for any t, ns_gym/schedulers.py:31-43); it did not fire */
  NSG_CNT_COUNT = 6
};
#define NSG_CNT_SHARDS 16384
#define NSG_MAX_ENVS (1LL << 27) /* envs per handle: rows are addressed with 32-bit byte offsets */

/* One tunable parameter = (Scheduler, UpdateFn) pair: UpdateFn.__call__ (ns_gym/base.py:124-149). */
typedef struct nsg_param_cfg {
  int32_t theta_slot;   /* which θ of the env type this entry drives                       */
  int32_t sched_kind;
  int32_t upd_kind;
  int32_t rng_child;    /* position in tunable_params dict order = SeedSequence child index
                           (ns_gym/base.py:418-421); also the index of its stream buffer    */
  double sched_start;   /* inclusive range gate, end may be +inf (ns_gym/base.py:56-81)     */
  double sched_end;
  int64_t sched_i0;
  int64_t sched_i1;
  double sched_p0;
  double sched_p1;
  int32_t sched_tab_off; /* bit table: offset in 32-bit words into the table blob          */
  int32_t sched_tab_len; /* bit table: number of valid bits                                 */
  int32_t val_tab_off;   /* value table: offset in doubles into the table blob              */
  int32_t val_tab_len;   /* value table: number of entries (triples for distributions)      */
  double u[10];
  uint64_t fn_seed;      /* constructor seed of a stochastic fn (valid if has_fn_seed)      */
  int32_t has_fn_seed;
  int32_t uses_rng;      /* 1 if the update fn owns a PCG64 stream                          */
  uint64_t sched_seed;   /* constructor seed of a stochastic scheduler (valid if has_sched_seed) */
  int32_t has_sched_seed;
  /* Shared objects.  The reference's tunable_params may map several parameter names to ONE UpdateFn object, and
   * several update fns may hold ONE Scheduler object (the reference's own fixtures do: tests/test_step_reset.py:34-44).
   * A shared stateful object is one stream / one list cursor / one transition_time consumed by its users in dict
   * order (classic_control.py:80-92).  fn_slot / sched_slot = index of the FIRST entry that uses the same UpdateFn /
   * Scheduler object (= the entry's own index when it is not shared): the rows of rng_upd + cursor / of
   * rng_sched + sched_next this entry reads and writes.  rng_child of every sharer is the index of the LAST one
   * (reset(seed) seeds in dict order: the last seed wins, base.py:412-421). */
  int32_t fn_slot;
  int32_t sched_slot;
  int32_t reserved0;
} nsg_param_cfg;

typedef struct nsg_config {
  int32_t abi_version;
  int32_t env_type;
  int32_t n_params;
  int32_t max_episode_steps;  /* gymnasium TimeLimit [UPSTREAM]; <= 0 means none            */
  uint32_t flags;
  int32_t nrow, ncol;         /* FrozenLake map (ns_gym/wrappers/toy_text.py:314-319)       */
  int32_t desc_tab_off;       /* FrozenLake desc bytes: offset in BYTES into the table blob */
  double base_theta[NSG_MAX_THETA]; /* construction-time θ (TUNABLE_PARAMS, base.py:1156)   */
  double initial_prob[2][4];  /* initial_prob_dist of the grid wrappers (toy_text.py:37,291,556);
                                 row 1 = Bridge's right-half initial distribution            */
  double letter_reward[4];    /* modified_rewards for S,F,H,G (FrozenLake, CliffWalking)     */
  nsg_param_cfg params[NSG_MAX_PARAMS];
} nsg_config;

/* Caller-owned device buffers.  N = envs, F = phys dims of the env type, P = n_params,
 * D = obs dim.  Pointers that a configuration does not need may be NULL (nsg_layout says
 * which are needed). */
typedef struct nsg_buffers {
  double* phys;          /* [ceil(N/256)][F][256] integrator state, fp64 like the reference: the F rows of a
                            256-env chunk (= one workgroup) are contiguous; element (row k, env i) sits at
                            ((i / 256) * F + k) * 256 + i % 256                                        */
  int32_t* cell;         /* [N]    grid envs: state index s                                */
  double* theta;         /* [P][N] tuned θ   (grid envs: [P][n][N] distributions, n = 3 or 4) */
  double* table_prob;    /* [ceil(N/256)][n][256] (chunk-blocked like phys) FrozenLake / CliffWalking: probabilities baked into the wrapper's P table.
                            Differs from theta after a reset: the reference's reset restores
                            transition_prob (toy_text.py:396) but its next step re-installs the
                            wrapper's un-reset self.P (toy_text.py:365-367), so the previous
                            episode's slip probabilities stay in force until the next fire.   */
  double* derived;       /* [2][N] CartPole planning copies only: total_mass, polemass_length as resolved
                            at fork time.  get_planning_env() re-installs the initial θ AFTER the
                            copy's _dependency_resolver() ran and a frozen copy never resolves again
                            (classic_control.py:70-75,133-136,183), so these can be stale w.r.t. θ.
                            [4][N] CliffWalking planning copies: the probabilities of the copy's OWN table (the wrapper's
                            `self.P`, toy_text.py:246-248), which a frozen copy never steps with (table_prob holds the base
                            env's) but hands to a copy taken of IT (nsg_fork)                                            */
  int32_t* t;            /* [N]    wrapper time t == obs["relative_time"] (base.py:314,347) */
  int32_t* t_fork;       /* [N]    planning copies only: t at fork time.  __deepcopy__ builds a fresh
                            gym.make() env, so TimeLimit's elapsed count restarts at 0 while the
                            wrapper's t is preserved (classic_control.py:168-180)              */
  uint8_t* status;       /* [N]    grid envs: NSG_ST_NEEDS_RESET | NSG_ST_TABLE_* << 1 (classic-control envs: NULL, see episode) */
  int32_t* episode;      /* [N]    classic-control envs: NSG_ST_NEEDS_RESET | resets drawn so far << 1.  A dense row that every
                            step reads and rewrites (Pendulum: resetting lanes only, see NSG_ST_NEEDS_RESET); with it the env's
                            np_random needs NO per-env stream state (rng_env)  */
  uint64_t* rng_env;     /* env np_random (gymnasium seeding [UPSTREAM]; PCG64(SeedSequence(seed))).
                            Grid envs draw one uniform per env per step: chunk-blocked state rows [ceil(N/256)][4][256]
                            (state_hi, state_lo, inc_hi, inc_lo).
                            Classic-control envs draw ONLY in reset() (D doubles: CartPole / Acrobot 4, Pendulum 2,
                            MountainCar 1), so episode e of env i starts from draws [D*e, D*e + D) of the stream seeded
                            with seed_i - a pure function of (seed_i, e) that the kernels re-derive by PCG64 jump-ahead
                            instead of round-tripping a 32-byte record through HBM at every reset (measured: those records
                            cost a 2^22-env CartPole step 24 %, nsg_rng.hip.h).  Layout [N + 1][2]:
                              [0]      stream descriptor: word 0 = NSG_STREAM_AFFINE | (uint32) spawn key, word 1 = base seed.
                                       AFFINE: env i is seeded base + i (gymnasium's vector-env convention; what
                                       nsg_reset_seeded, nsg_bind and nsg_fork install) and records 1.. are not read;
                              [1 + i]  (seed_i, spawn key_i) of env i, read by the few lanes that reset, when the streams are
                                       not affine (nsg_reset / nsg_seed_streams with an arbitrary seed array).
                            Spawn key -1 = none: SeedSequence(seed); k >= 0: SeedSequence(seed, spawn_key=(k,))              */
  uint64_t* rng_upd;     /* [P][N][4] update-fn PCG64 streams (only rows with uses_rng)   */
  uint64_t* rng_sched;   /* [P][N][4] PCG64 records of stochastic schedulers (Random, DecayingProbability,
                            Memoryless).  A scheduler lives inside the deep-copied init_initial_params, so a
                            non-persistent reset REWINDS its stream to the construction state
                            (base.py:381-384); it is never re-seeded by reset(seed) (base.py:151-158)   */
  int32_t* sched_next;   /* [P][N] MemorylessScheduler.transition_time (schedulers.py:108-113)   */
  int32_t* cursor;       /* [P][N] StepWise/Cyclic list cursor                            */
  float* obs;            /* [N][D] obs["state"] float32 (classic control)                 */
  float* reward;         /* [N]                                                           */
  uint8_t* terminated;   /* [N]                                                           */
  uint8_t* truncated;    /* [N]                                                           */
  uint8_t* env_change;   /* [P][N] ground-truth flags (info["Ground Truth Env Change"])   */
  float* delta_change;   /* [P][N] ground-truth deltas (info["Ground Truth Delta Change"]) */
  uint8_t* violation;    /* [P][N] 1 where this step's proposal was rejected by the constraint checker
                            (classic_control.py:87-92; the reference warns instead).  NSG_F_VIOLATION_MASK */
  float* prob;           /* [N]    FrozenLake info["prob"]                                */
  float* ep_return;      /* [N]    running episode return      (NSG_F_TRACK_RETURNS).  Not maintained for the env types
                            whose reward is the same constant on every step (CartPole +1, MountainCar -1: the running return
                            IS +-t) or paid only by the step that ends the episode (FrozenLake with default rewards,
                            Bridge: the return IS that reward); last_return is derived when an episode ends        */
  int32_t* ep_length;    /* [N]    running episode length                                 */
  float* last_return;    /* [N]    return of the last finished episode.  NULL for CartPole (+1 per step) and MountainCar (-1 per
                            step): their return is +-last_length                                              */
  int32_t* last_length;  /* [N]                                                           */
  uint64_t* counters;    /* [NSG_CNT_COUNT][NSG_CNT_SHARDS] running totals                */
  uint64_t* done_bits;   /* [ceil(N/64)] wavefront ballot of "episode ended this step":
                            bit l of word w = env 64*w + l   (input of nsg_compact_done)   */
} nsg_buffers;

/* element counts the caller must allocate for each nsg_buffers member (0 = not needed) */
typedef struct nsg_layout {
  int64_t n;
  int32_t phys_dim, obs_dim, n_params, n_theta_rows, action_is_float;
  int32_t n_actions;           /* discrete action count, 0 for continuous                  */
  int64_t phys, cell, theta, table_prob, derived, t, t_fork, status, episode, rng_env, rng_upd, rng_sched, sched_next, cursor, obs, reward, terminated,
      truncated, env_change, delta_change, violation, prob, ep_return, ep_length, last_return, last_length,
      counters, done_bits;
} nsg_layout;

typedef struct nsg_handle nsg_handle;

int nsg_abi_version(void);
const char* nsg_last_error(void);
size_t nsg_sizeof_config(void);
size_t nsg_sizeof_buffers(void);
size_t nsg_sizeof_layout(void);

/* Buffer sizes for (cfg, n).  Pure host arithmetic. */
int nsg_layout_query(const nsg_config* cfg, int64_t n, nsg_layout* out);

/* Replaces NSClassicControlWrapper.__init__ / NSFrozenLakeWrapper.__init__
 * (ns_gym/wrappers/classic_control.py:27-58, ns_gym/wrappers/toy_text.py:282-340) for N
 * env instances: validates cfg, uploads the constant tables (`tables`: the blob that
 * sched_tab_off / val_tab_off / desc_tab_off index). */
int nsg_create(const nsg_config* cfg, const void* tables, size_t table_bytes, int64_t n,
               nsg_handle** out);

/* Attach caller-owned device buffers (sizes from nsg_layout_query). */
int nsg_bind(nsg_handle* h, const nsg_buffers* bufs);

/* Replaces NSWrapper.reset + subclass tails (ns_gym/base.py:365-431,
 * classic_control.py:102-109, toy_text.py:382-399) for every env with mask[i] != 0
 * (mask NULL = all).  seeds_dev NULL = reset(seed=None): env and update-fn streams
 * continue; otherwise env i is seeded with seeds_dev[i] exactly like reset(seed=seeds[i]). */
int nsg_reset(nsg_handle* h, const uint64_t* seeds_dev, const uint8_t* mask_dev, void* stream);

/* reset(seed=s) for ALL envs with gymnasium's vector-env convention: env i is seeded base_seed + i, exactly like
 * nsg_reset with seeds_dev[i] = base_seed + i.  For the classic-control envs this keeps the streams in their affine form
 * (buffers.rng_env): no per-env seed record is read when an env resets. */
int nsg_reset_seeded(nsg_handle* h, uint64_t base_seed, void* stream);

/* Host-side construction of the PCG64 jump-ahead table the kernels use (nsg_rng.hip.h): NSG_JUMP_TABLE_WORDS uint64 words,
 * 4 per entry (A_hi, A_lo, G_hi, G_lo): NSG_JUMP_LOW entries for the exponents 0 .. NSG_JUMP_LOW-1, then 4 x 256 entries for
 * the exponents v * 256^d (d = 1 .. 4).  No GPU needed; exported so that the table can be checked against plain big-integer
 * arithmetic (tests/test_jump_table_cpu.py). */
#define NSG_JUMP_LOW 264
#define NSG_JUMP_TABLE_WORDS ((NSG_JUMP_LOW + 4 * 256) * 4)
void nsg_pcg64_jump_table(uint64_t* out_words);

/* Replaces NSClassicControlWrapper.step / NSFrozenLakeWrapper.step -> NSWrapper.step ->
 * gymnasium step (classic_control.py:60-100, toy_text.py:342-380, base.py:296-363) for all
 * N envs in one fused launch.  actions_dev: int32[N] (discrete) or float[N] (continuous).
 * Envs whose previous step ended an episode are reset (seed=None) instead of stepped
 * (gymnasium next-step autoreset): reward 0, terminated/truncated 0, relative_time 0. */
int nsg_step(nsg_handle* h, const void* actions_dev, void* stream);

/* Batched planning-env snapshot: get_planning_env() / __deepcopy__ of the reference wrappers
 * (classic_control.py:120-186, toy_text.py:471-511, base.py:433-441) for all N envs at once.
 * `dst` must have been created from the same config plus NSG_F_SIM_ENV and bound to its own
 * buffers.  Copies state, t (t_fork = t), list cursors and the last outputs; θ per theta_mode:
 *   0 = current θ          (deepcopy; get_planning_env when delta_change_notification or sim env)
 *   1 = construction-time θ (get_planning_env without delta_change_notification)
 * and re-seeds every stream from `entropy` (the reference uses fresh OS entropy:
 * _reseed_planning_env_rngs, base.py:433-441; env np_random of the new gym.make() env).
 * `src` may itself be a planning copy (MCTS.search deep-copies the planning env it is given for every simulation, MCTS.py:131).
 * Grid envs then follow the reference's two P tables (the wrapper's own and the base env's): a FrozenLake copy of a copy steps
 * with the table of initial_prob_dist (toy_text.py:491-508: the first copy's own table is its constructor's), a CliffWalking copy
 * of a copy with the first source's current table even where get_planning_env() had given the first copy the initial one
 * (toy_text.py:219-221,246-249).  tests/golden/policy_mcts_frozenlake_*.npz, policy_mcts_cliff_theta0.npz.
 * dst may hold k whole copies of src (dst N = k * src N; copy j <- env j mod src N, streams from
 * entropy + j): the simulations a planner runs for one decision (MCTS.py:131: one deepcopy per
 * simulation) as ONE batch that a single nsg_rollout launch then advances. */
int nsg_fork(nsg_handle* src, nsg_handle* dst, uint64_t entropy, int32_t theta_mode, void* stream);

/* Re-seed streams without touching env state: which = 0 env np_random (env.np_random =
 * default_rng(seed)), 1 = update-fn streams (SeedSequence(seed).spawn children, base.py:412-421). */
int nsg_seed_streams(nsg_handle* h, const uint64_t* seeds_dev, int32_t which, void* stream);

/* K fused steps with state held in registers between steps; actions [K][N]; per-step
 * outputs are written to the caller's [K][...] trajectory buffers (any may be NULL). */
typedef struct nsg_rollout_out {
  float* obs;          /* [K][N][D]  (FrozenLake: int32 [K][N] passed through this pointer) */
  float* reward;       /* [K][N] */
  uint8_t* terminated; /* [K][N] */
  uint8_t* truncated;  /* [K][N] */
  uint8_t* env_change; /* [K][P][N] */
  float* delta_change; /* [K][P][N] */
} nsg_rollout_out;
int nsg_rollout(nsg_handle* h, const void* actions_dev, int32_t k_steps, const nsg_rollout_out* out,
                void* stream);

/* ---- fused rollouts that choose their own actions and keep the episode accounts ------------------------------------------------
 * The reference's step consumers are loops of the shape `action = policy(observation); observation, reward, ... = env.step(action)`
 * that sum rewards until the episode ends: MCTS._default_policy (benchmark_algorithms/MCTS.py:162-181: uniformly random actions,
 * tot_reward += reward * gamma ** depth while not terminated and depth < d and not truncated), run_episode
 * (evaluate/run_experiment.py:108-129: total_reward += reward until done / truncated), the tutorial's run_episode with a tabular
 * policy (tutorial.ipynb cell 12: action = policy[observation]).  With nsg_rollout such a loop needs the actions of all K steps
 * up front, so only open loops fuse; nsg_rollout_policy evaluates the policy INSIDE the launch, per env, from the observation the
 * previous fused step produced, and keeps the accounts in registers: a closed loop of K steps is one launch with no per-step
 * traffic at all (no action row in, and - when `out` is NULL - nothing out but the accounts).
 *
 * Action sources (nsg_policy.kind):
 *   NSG_POL_TABLE     data = actions[K][N] (int32 / float): exactly nsg_rollout.
 *   NSG_POL_UNIFORM   counter-based uniform actions: env i, step k of the launch draws from bits = nsg_policy_bits(seed, index0 + i,
 *                     step0 + k) (below; a pure function - any sharding of the batch and any chunking of the K steps reproduce the
 *                     same actions).  Discrete: action = ((bits >> 32) * n_actions) >> 32.  Continuous: low + (high - low) *
 *                     ((float)(bits >> 40) * 2^-24f) in float32, [low, high] the env type's action bounds (Pendulum +-2, MountainCarContinuous +-1).
 *                     (The reference draws from NumPy's unseeded global generator - np.random.choice, MCTS.py:176 - or from the action
 *                     space's own unseeded one: a distribution, not a stream that could be reproduced.)
 *   NSG_POL_BY_STATE  grid envs: action = ((const int32_t*)data)[cell], n_data >= nrow * ncol entries (tutorial cell 12).
 *   NSG_POL_LINEAR    classic control: data = float W[rows][obs_dim + 1] (bias last), score_j = W[j][obs_dim] + sum_d W[j][d] * obs[d]
 *                     accumulated in float32 in that order on the float32 observation an agent would see.  Discrete: rows =
 *                     n_actions, action = first argmax_j.  Continuous: rows = 1, action = clip(score_0, low, high).
 * actions_out (may be NULL): the actions taken, [K][N] - the A of run_episode's SARNS records.
 *
 * Accounts (nsg_episode_acc, may be NULL; every pointer in it may be NULL): per env, in/out across launches,
 *   alive[i]   1 while the env's episode is running; cleared by the step that returns terminated or truncated
 *   ret[i]     += reward64 * discount[length[i]] on every step taken while alive - the float64 reward of the base MDP and Python's
 *              operation order (`tot_reward += reward * gamma ** depth`); discount NULL or length[i] >= n_discount: 1.0.  The caller
 *              fills discount[j] = gamma ** j with ITS pow, which is the reference's
 *   length[i]  += 1 on every step taken while alive
 * A step that performs a pending autoreset instead of a transition takes no action and changes no account. */
enum { NSG_POL_TABLE = 0, NSG_POL_UNIFORM = 1, NSG_POL_BY_STATE = 2, NSG_POL_LINEAR = 3 };
typedef struct nsg_policy {
  int32_t kind;
  int32_t step0;        /* NSG_POL_UNIFORM: counter of the launch's first step                                   */
  uint64_t seed;        /* NSG_POL_UNIFORM: key of the action streams                                             */
  int64_t index0;       /* NSG_POL_UNIFORM: global index of the handle's env 0 (shards of one job, copy batches)  */
  const void* data;     /* device memory, see above                                                               */
  int32_t n_data;       /* NSG_POL_BY_STATE: entries; NSG_POL_LINEAR: rows                                        */
  int32_t reserved0;
  void* actions_out;    /* [K][N] int32 / float, or NULL                                                          */
} nsg_policy;
typedef struct nsg_episode_acc {
  double* ret;            /* [N] */
  int32_t* length;        /* [N] */
  uint8_t* alive;         /* [N]; NULL: every env counts as alive on entry and nothing is carried over              */
  const double* discount; /* [n_discount] device memory, or NULL                                                     */
  int32_t n_discount;
  int32_t reserved0;
} nsg_episode_acc;
int nsg_rollout_policy(nsg_handle* h, const nsg_policy* pol, int32_t k_steps, const nsg_rollout_out* out,
                       const nsg_episode_acc* acc, void* stream);
/* Which kernel nsg_rollout_policy launches for this handle and action source `kind`: 0 the precompiled generic kernel, 1 the handle's
 * specialised unit of that kind (valid after the first such rollout of a specialised handle). */
int nsg_rollout_policy_kind(const nsg_handle* h, int32_t kind);
/* The bits behind NSG_POL_UNIFORM (host-callable; tests and callers that want the same actions elsewhere). */
uint64_t nsg_policy_bits(uint64_t seed, uint64_t env_index, uint64_t step);
/* The specialised unit nsg_rollout_policy builds, on first use, for a specialised handle and the action source `kind` (one kernel,
 * nsg_spec_rollout_policy, with the kind as a compile-time constant); no GPU needed, see nsg_spec_build. */
int nsg_spec_build_policy(const nsg_config* cfg, int32_t kind, const char* arch, void** code_out, size_t* size_out);

/* ---- resident stepper: closed loops in the launch-bound regime ---------------------------------------------------------------
 * nsg_step costs a dependent launch per step (4-5 us on this stack before the kernel does anything); a batch of <= 2^17 envs is
 * one wavefront per SIMD, whose step itself takes ~2.7 us.  Callers that can hand over K action rows at once use nsg_rollout; a
 * CLOSED loop - a policy that needs step k's observation to choose action k + 1 - cannot.  nsg_resident_start launches ONE kernel
 * that stays on the device, keeps the env state in registers / LDS like nsg_rollout, and takes a step whenever the producer of
 * the actions publishes the next row through the mailbox (caller-owned DEVICE memory, ZEROED by the caller before every start):
 *
 *   producer (kernels / a resident kernel on another stream), per 256-env CHUNK j:  write actions_dev[256 j .. 256 j + 256) for step
 *              k; if mb->stop is clear, release-store mb->act_seq[j] = k + 1 (agent scope)
 *   stepper:   the workgroup of chunk j polls act_seq[j] >= k + 1, steps its 256 envs, stores their OUTPUT rows, fences, release-stores
 *              mb->step_seq[j] = k + 1
 *   consumer:  polls step_seq[j] >= k + 1 (acquire), reads chunk j of the handle's output rows (obs, reward, terminated, ...)
 *
 * The hand-shake is per chunk on purpose: a counter that all workgroups of a step add themselves to costs tens of microseconds
 * per step on this chip (256 same-address device-scope atomics serialise at the memory side - measured, profiles/NOTEBOOK.md),
 * more than the launches the resident kernel is there to save.  A policy that works env by env (or chunk by chunk) pairs its
 * workgroup j with the stepper's; one that needs the whole batch waits for every step_seq[j] on its own side.
 *
 * The wait is BOUNDED.  A workgroup that has waited `wait_budget_us` for its next row raises mb->stop itself; from the moment a
 * workgroup sees `stop` raised (by a starved workgroup, or by anyone who wants the loop to end) it keeps looking for ONE more
 * row for a grace period (NSG_RESIDENT_GRACE_US) and then leaves.  Leaving, a workgroup writes its persistent rows back; the last
 * one out writes status, steps_done (the FEWEST steps any chunk has taken) and steps_max (the most):
 *   NSG_MB_FINISHED  every chunk has taken max_steps steps - the way a loop of K steps ends.  The handle's buffers describe the batch
 *                    exactly as after max_steps nsg_step calls, bit for bit.
 *   NSG_MB_STARVED   the producer went silent / NSG_MB_STOPPED  stop was raised from outside.  Every env is consistent at the step
 *                    count of ITS chunk (step_seq[j]).  With a producer that publishes for all chunks at once (nsg_resident_publish,
 *                    which does not publish once stop is raised) the chunks agree - steps_done == steps_max - and the batch is as
 *                    after that many nsg_step calls: such a producer publishes at most one more step in the shadow of a stop,
 *                    microseconds after it, when every workgroup is still polling.  Chunks whose producers run independently of
 *                    each other (the demo policy's workgroups) are wherever each got to.
 * Afterwards nsg_step / nsg_rollout / another nsg_resident_start carry on from there.  While the kernel is resident nothing else
 * may launch on the handle.  Batches of at most NSG_RESIDENT_MAX_ENVS envs (one workgroup per chunk, all of them resident at once). */
#define NSG_RESIDENT_MAX_ENVS (1 << 17)
#define NSG_RESIDENT_MAX_CHUNKS (NSG_RESIDENT_MAX_ENVS / 256)
#define NSG_RESIDENT_GRACE_US 200u
typedef struct nsg_mailbox {
  uint64_t stop;          /* non-zero = leave after the grace period (NSG_MB_STARVED when a starved workgroup raised it)       */
  uint64_t status;        /* 0 while resident / never started; NSG_MB_* once the launch has left                          */
  uint64_t steps_done;    /* valid with status: the fewest steps a chunk has taken since the start ...                      */
  uint64_t steps_max;     /* ... and the most (equal for NSG_MB_FINISHED and for producers that publish for all chunks at once) */
  uint64_t leave, taken_max, taken_min_inv, reserved[1];   /* internal */
  uint64_t act_seq[NSG_RESIDENT_MAX_CHUNKS];    /* producer -> stepper, per chunk: k + 1 once chunk j's actions of step k are in place */
  uint64_t step_seq[NSG_RESIDENT_MAX_CHUNKS];   /* stepper -> consumer, per chunk: k + 1 once chunk j's outputs of step k are in the rows */
} nsg_mailbox;
#define NSG_MB_FINISHED 1u
#define NSG_MB_STARVED 2u
#define NSG_MB_STOPPED 3u
int nsg_resident_start(nsg_handle* h, const void* actions_dev, nsg_mailbox* mb_dev, int32_t max_steps, uint32_t wait_budget_us,
                       void* stream);
/* The producer's publish for callers whose policy is ordinary kernels (or host copies) enqueued per step: call it on THEIR stream
 * right behind whatever wrote actions_dev for step `step` - it sets act_seq[j] = step + 1 for every chunk of the batch (coherently;
 * not at all once stop has been raised).  The consumer side of such a caller reads mb->step_seq[] with ordinary kernels. */
int nsg_resident_publish(nsg_handle* h, nsg_mailbox* mb_dev, int32_t step, void* stream);
/* A stand-in policy for measurements and tests (discrete-action classic-control envs): a resident kernel on the OTHER side of the
 * mailbox whose workgroup j, for each of max_steps steps, waits (bounded, same rules) for step_seq[j], writes
 * action[i] = ((obs[i][watch] > 0) + k) mod n_actions for its chunk and publishes act_seq[j].  Launch it on a stream of its own. */
int nsg_resident_demo_policy(nsg_handle* h, int32_t watch, int32_t* actions_dev, nsg_mailbox* mb_dev, int32_t max_steps,
                             uint32_t wait_budget_us, void* stream);

/* Heterogeneous batch: one launch over up to NSG_MAX_SEGMENTS handles of different env
 * types (per-env-type dispatch is wave-uniform because segments are block-aligned).
 * The first launch of a member list PLANS it (block ranges, a device-side segment table, the group's specialised unit when
 * every member is specialised): that synchronises the device once.  The plan is kept per member list and stays valid until
 * one of ITS members is re-bound, specialised or destroyed; other handles coming and going do not touch it.  A segment table
 * is never overwritten or freed while anything - a launch in flight, a captured HIP graph - may read it (a table that a re-plan, an
 * eviction or the destruction of a member replaced stays allocated for the next 1024 such events; it is then freed behind a
 * synchronisation of its own device, on a planning call, never inside nsg_destroy and never while the stream is capturing).  On a capturing
 * stream a launch that would have to plan first is refused (NSG_EINVAL): launch the group once before the capture. */
int nsg_step_group(nsg_handle* const* hs, int32_t n_handles, const void* const* actions_dev,
                   void* stream);
/* K fused steps of EVERY member in one launch: nsg_rollout's heterogeneous counterpart (same plan, same block ranges as
 * nsg_step_group; every member's persistent rows stay in registers / LDS for the K steps).  actions_dev[k]: member k's [K][N_k]
 * actions; outs: NULL or one nsg_rollout_out per member (any pointer may be NULL).  Bit-identical to K nsg_step_group calls. */
int nsg_rollout_group(nsg_handle* const* hs, int32_t n_handles, const void* const* actions_dev, int32_t k_steps,
                      const nsg_rollout_out* outs, void* stream);
/* Which kernel the current plan of this member list launches (>= 0), or a negative error code. */
#define NSG_GROUP_UNPLANNED 0        /* not launched yet, or a member changed since */
#define NSG_GROUP_GENERIC_SIMPLE 1   /* precompiled kernel, plain-arithmetic theta engine */
#define NSG_GROUP_GENERIC_FULL 2     /* precompiled kernel, full theta engine */
#define NSG_GROUP_SPECIALISED 3      /* the unit compiled for the ordered tuple of the members' configs (nsg_spec_group) */
#define NSG_GROUP_SPECIALISED_PREBUILT 4 /* the same, shipped with the library (NSG_SPEC_ORIGIN_PREBUILT) */
int nsg_step_group_kind(nsg_handle* const* hs, int32_t n_handles);

/* θ-schedule engine alone (Scheduler.__call__ + UpdateFn.__call__, base.py:67-81,124-149)
 * on param slot `p` of the handle's config: n lanes, each starting from theta0[i] (3 doubles
 * per lane for distributions) and its stream, iterated over t = t0 .. t0+T-1, θ fed back.
 * Outputs [T][n] (theta: [T][n] or [T][3][n]).  rng_state: [n][4] records or NULL. */
int nsg_theta_trace(nsg_handle* h, int32_t p, int32_t n, int32_t t0, int32_t T, const double* theta0,
                    uint64_t* rng_state, double* theta_out, uint8_t* fired_out, double* delta_out,
                    void* stream);

/* The same with the FULL per-object state carried across calls, so that a host-side Scheduler / UpdateFn
 * object called repeatedly behaves like the reference's stateful objects (rng draws continue, StepWise /
 * Cyclic lists advance, LCBounded remembers prev_time, stochastic schedulers keep their stream and
 * transition_time; base.py:67-81,124-149).  All arrays are device pointers, in/out, and may be NULL when
 * the config has no such state. */
typedef struct nsg_trace_state {
  uint64_t* rng;        /* [n][4] update-fn stream records                                         */
  int32_t* cursor;      /* [n]    list cursor (StepWise/Cyclic) or prev_time + 1 (LCBounded)       */
  uint64_t* sched_rng;  /* [n][4] stream of a stochastic scheduler                                 */
  int32_t* sched_next;  /* [n]    MemorylessScheduler.transition_time                              */
  int32_t resume;       /* 0: cursor / scheduler state start at construction values (the arrays
                           receive the final state); 1: continue from the arrays                   */
  int32_t reserved0;
} nsg_trace_state;
int nsg_theta_trace_stateful(nsg_handle* h, int32_t p, int32_t n, int32_t t0, int32_t T, const double* theta0,
                             const nsg_trace_state* state, double* theta_out, uint8_t* fired_out,
                             double* delta_out, void* stream);

/* NumPy-compatible bit streams on device (SeedSequence -> PCG64; numpy Generator.random /
 * normal).  kind 0: raw uint64, 1: random() double, 2: standard_normal double.
 * seeds[n], spawn_key < 0 = root stream, else SeedSequence(seed).spawn(..)[spawn_key].
 * out [count][n].  state_out [n][4] records, may be NULL. */
int nsg_rng_fill(int32_t kind, const uint64_t* seeds_dev, int32_t n, int32_t spawn_key, int32_t count,
                 void* out_dev, uint64_t* state_out_dev, void* stream);

/* A caller that writes buffers.table_prob itself (grid envs) calls this afterwards: it clears the table hint of every env's status
 * byte (NSG_ST_TABLE_ROWS), so that the next step reads the rows instead of the distribution the byte still named. */
int nsg_table_prob_dirty(nsg_handle* h, void* stream);

/* Done-mask compaction: expands the ballot words of the last step into a dense list of env
 * indices (order unspecified).  idx_out_dev: int32[N], count_out_dev: uint64[1] (zeroed here). */
int nsg_compact_done(nsg_handle* h, int32_t* idx_out_dev, uint64_t* count_out_dev, void* stream);

/* Timing helper: average device time (ms) of `iters` back-to-back nsg_step launches measured
 * with hipEvents on `stream` (synchronises; not for use inside graph capture). */
int nsg_time_steps(nsg_handle* h, const void* actions_dev, int32_t iters, void* stream, float* ms_avg);

/* Counter calibration helper (MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE must be calibrated
 * on a known byte count in the kernel's own access width): streams n float64 from src to dst
 * with one 8-byte access per lane, the access shape of the step kernels' state rows. */
int nsg_calib_copy_f64(const double* src_dev, double* dst_dev, int64_t n, void* stream);

/* Latency-bound read-back (the N = 1 adaptors' step outputs): one small launch copies `bytes` (a multiple of 16, at most
 * 1 MiB) from device memory into PINNED, device-mapped host memory (hipHostMalloc; same address on both sides) and then stores
 * `seq` into the uint64 that follows the copied bytes (system-scope release).  The host polls that word: no DMA copy, no
 * event.  `dst_host_mapped` must therefore have room for bytes + 8.  New: the reference reads Python attributes. */
int nsg_read_back(const void* src_dev, void* dst_host_mapped, int64_t bytes, uint64_t seq, void* stream);

/* Config-specialised kernels.  The generic kernels read nsg_config through scalar loads and branch
 * on it at run time — the device-side equivalent of the reference's per-step Python dispatch over
 * Scheduler / UpdateFn objects (ns_gym/base.py:124-149, classic_control.py:60-100).  nsg_specialize()
 * compiles the SAME kernel bodies with the handle's nsg_config as a compile-time constant (hiprtc,
 * sources embedded in the library, ~1 s once per distinct config; the code objects persist in
 * NSG_SPEC_CACHE=<dir>, by default $XDG_CACHE_HOME/ns_gym_amd or $HOME/.cache/ns_gym_amd, "off" for none,
 * keyed by config, kernel sources, compile options and HIP version; an unusable cached object is rebuilt)
 * and routes nsg_step / nsg_rollout of this handle through them.
 * Results are bit-identical to the generic kernels.  Returns NSG_EUNSUPPORTED (generic path stays in
 * force) when libhiprtc is missing (a unit already in the cache directory loads without it) or the compilation fails.
 * The unit also depends on the batch size where a launch policy does: CartPole batches of 49 152 - 163 840 envs reset in-lane,
 * classic-control batches of >= 2^24 envs store their persistent rows non-temporally, configs without a table blob stage
 * nothing into LDS.  A unit in which any kernel spills vector registers or owns scratch memory is never loaded (rebuilt
 * without the register bound, else refused: profiles/r03_case61_spill_evidence.md).
 * nsg_spec_build compiles only (no GPU needed; arch e.g. "gfx950"): *code_out is a malloc'ed code
 * object to be released with nsg_spec_free. */
int nsg_specialize(nsg_handle* h);
int nsg_is_specialized(const nsg_handle* h);
/* Where the handle's specialised unit came from (0 = not specialised). */
#define NSG_SPEC_ORIGIN_NONE 0
#define NSG_SPEC_ORIGIN_HIPRTC 1    /* compiled by hiprtc in this process                                        */
#define NSG_SPEC_ORIGIN_CACHE 2     /* read from the user's disk cache (NSG_SPEC_CACHE), compiled by an earlier process */
#define NSG_SPEC_ORIGIN_PREBUILT 3  /* shipped with the library: <directory of libnsgym_hip.so>/prebuilt (or NSG_PREBUILT_DIR), built and
                                       inspected at library build time (nsg_spec_prebuild; csrc/prebuilt_resource_usage.txt)  */
int nsg_spec_origin(const nsg_handle* h);
/* Build-time side of NSG_SPEC_ORIGIN_PREBUILT (no GPU needed): compiles the unit nsg_specialize() would compile for a handle of
 * (cfg, n envs) on a device whose gcnArchName is `arch` (e.g. "gfx950:sramecc+:xnack-": the name is part of the key, and so is n where
 * the launch policy depends on it) and writes it to <dir>/nsg_<key>.hsaco, the file name nsg_specialize looks for.  The same spill
 * rule applies: a unit that spills is not written.  nsg_spec_prebuild_group: the unit of nsg_step_group / nsg_rollout_group for
 * the ordered member list (cfgs[k], ns[k]). */
int nsg_spec_prebuild(const nsg_config* cfg, int64_t n, const char* arch, const char* dir);
int nsg_spec_prebuild_group(const nsg_config* const* cfgs, const int64_t* ns, int32_t count, const char* arch, const char* dir);
/* ... and the fused policy rollout's unit (nsg_rollout_policy of a specialised handle of (cfg, n)) for the action source `kind` */
int nsg_spec_prebuild_policy(const nsg_config* cfg, int64_t n, int32_t kind, const char* arch, const char* dir);
int nsg_spec_build(const nsg_config* cfg, const char* arch, void** code_out, size_t* size_out);
/* the same for the unit nsg_step_group uses when every member is specialised: one kernel (nsg_spec_group) for the ordered tuple
 * of the members' configs */
int nsg_spec_build_group(const nsg_config* const* cfgs, int32_t n, const char* arch, void** code_out, size_t* size_out);
/* ... and for the unit nsg_resident_start builds, on first use, for a specialised handle (one kernel: nsg_spec_resident) */
int nsg_spec_build_resident(const nsg_config* cfg, const char* arch, void** code_out, size_t* size_out);
void nsg_spec_free(void* code);

int nsg_destroy(nsg_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* NSGYM_HIP_H */
