// nsg_kernels.hip.h — the fused struct-of-arrays kernels (gfx950 / CDNA4, wave64).
//
// One launch = one wrapper step for every env instance:
//   θ-schedule (schedulers + update fns)  ->  constraint checker  ->  dependency resolver
//   ->  base-MDP transition  ->  t += 1 / TimeLimit  ->  notification ground truth  ->
//   autoreset bookkeeping  ->  wave-ballot done mask + per-wavefront counters,
// i.e. NSClassicControlWrapper.step (ns_gym/wrappers/classic_control.py:60-100) /
// NSFrozenLakeWrapper.step (ns_gym/wrappers/toy_text.py:342-380) -> NSWrapper.step
// (ns_gym/base.py:296-363) -> gymnasium step, for N envs at once.
//
// Mapping: one env per lane, 256-thread workgroups (4 wavefronts), grid-stride over 256-env
// chunks.  Every per-env array is SoA ([field][N]) so a wavefront's access to a field is one
// contiguous 256/512-byte run.  The path is HBM-bound elementwise work: no MFMA, no
// inter-workgroup communication; constant tables (ziggurat, bit/value tables, FrozenLake
// map) are staged in LDS once per workgroup.  Workgroups touch disjoint env ranges, so the
// blockIdx -> XCD mapping has no L2-sharing consequence and is left as dispatched.
#pragma once
#include "nsgym_hip.h"
#include "nsg_envs.hip.h"
#include "nsg_rng.hip.h"
#include "nsg_theta.hip.h"

namespace nsg {

// Workgroup size: NSG_BLOCK lanes = NSG_BLOCK / 256 consecutive 256-env layout chunks (the chunk-blocked rows are laid out in
// 256-env chunks whatever the workgroup size, nsg_rng.hip.h blk_off8).  Measured with the counter-based env streams (C1,
// specialised, 2^20 / 2^22 / 2^24 envs): 256 lanes 26.0 / 87.3 / 460 us, 512 lanes 25.8 / 87.2 / 447, 1024 lanes 26.4 / 87.4 / 447;
// Acrobot and the full theta-engine lose with the larger ones (63 -> 65 -> 81 us, 34.4 -> 35.3 -> 40.4); 128 lanes (measured again on
// round 2's final code): C1 23.9 -> 24.6 us at 2^20 envs, 80.8 -> 81.6 at 2^22, C2 32.9 -> 36.2, Pendulum 20.2 -> 20.9.  256 stays.
#ifndef NSG_BLOCK
#define NSG_BLOCK 256
#endif
constexpr int kBlock = NSG_BLOCK;
constexpr int kMaxTableBytes = 16384;
constexpr int kCntShards = NSG_CNT_SHARDS;
// NSG_UNCOND_LOADS 1: the state / action loads do not wait for the episode word (a resetting lane's are discarded).
#ifndef NSG_RESET_LANE_PER_DRAW
#define NSG_RESET_LANE_PER_DRAW 1   // single-step reset hand-over: D helper lanes per reset (one draw each) instead of one
#endif
#ifndef NSG_UNCOND_LOADS
#define NSG_UNCOND_LOADS 1
#endif
#ifndef NSG_MIN_WAVES
#define NSG_MIN_WAVES 1
#endif
#ifndef NSG_TABLES_DIRECT
#define NSG_TABLES_DIRECT 0   // 1: constant tables are read from their global copies, not staged in LDS (stage_tables)
#endif


// Device-resident description of one homogeneous env segment (read through scalar loads).
struct Segment {
  nsg_config cfg;
  nsg_buffers buf;
  int64_t N;
  const uint8_t* tables;   // constant-table blob (global copy)
  const uint64_t* zig;     // ki | wi | fi | ke | we | fe, 256 words each (global copy)
  int32_t table_bytes;
  int32_t uses_normal;     // some update fn draws normals -> stage the normal ziggurat tables
  int32_t uses_exp;        // Memoryless (p < 1/3) / RandomCategorical -> stage the exponential ziggurat tables
  int32_t simple_theta;    // every update fn is plain arithmetic / table look-up (upd_kind_is_simple)
  const uint64_t* jump;    // PCG64 jump-ahead table (nsg_rng.hip.h, pcg_at): kJumpWords words, global copy
  int32_t block_begin;     // heterogeneous launch (nsg_step_group) only: first block and block count of this member in the
  int32_t block_count;     // launch's segment table (a per-plan COPY of the members' segments, see nsgym_hip.hip)
};

struct ActionPtrs {
  const void* p[NSG_MAX_SEGMENTS];
};
// Dynamic LDS layout (sized per handle at launch: a batch with tiny tables must not pay 22 KB of
// LDS per workgroup, which would cap residency at 6-7 workgroups per CU):
//   [ pad | reset_n[2] | pad | reset_list[kBlock] (short) | reset_state[kBlock][4] (f64) |
//     table blob | ziggurat ki/wi/fi | streams[kBlock][4] (u64; fused rollouts of classic envs only) ]
struct LdsTables {
  int* reset_n;         // [2] workgroup-level compaction of the autoreset lanes (double-buffered)
  short* reset_list;    // [kBlock] lanes whose env resets in this chunk (the owner leaves its episode count in its reset_state slot)
  double* reset_state;  // [kBlock][4] initial states drawn by the helper lanes, read back by the owners
  uint64_t* blob;       // constant-table blob
  uint64_t* zig;        // 768 words (normal) + 768 words (exponential), each only when needed
  uint64_t* streams;    // [kBlock][4] fused rollouts of the classic envs: every lane's env stream, derived ONCE per launch (rollout_body)
  uint64_t* ustreams;   // [kMaxLdsUpd][kBlock][4] the chunk's update-fn streams (first stochastic fns) during a fused rollout
};
constexpr int kLdsStreamBytes = kBlock * 32;
constexpr int kMaxLdsUpd = 2;  // update-fn streams held in LDS during a fused rollout; further ones go through memory

// Which LDS stream block the update fn in slot `slot` uses (= how many stochastic fn objects precede it).
__host__ __device__ inline int upd_lds_index(const nsg_config& cfg, int slot) {
  int n = 0;
  for (int q = 0; q < slot; q++)
    if (cfg.params[q].uses_rng && cfg.params[q].fn_slot == q) n++;
  return n;
}
__host__ __device__ inline int upd_lds_count(const nsg_config& cfg) {
  const int n = upd_lds_index(cfg, cfg.n_params);
  return n < kMaxLdsUpd ? n : kMaxLdsUpd;
}
// (1.5 KB more here - a 64-bit queue entry instead of a lane index - once cost C2's fused rollout its fifth workgroup per CU:
// 31.4 -> 32.9 KB of LDS each, 17.1 -> 18.8 us per step)
constexpr int kLdsHeaderBytes = 32 + kBlock * 2 + kBlock * 4 * 8;

__host__ __device__ constexpr int lds_bytes_for(int table_bytes, int uses_normal, int uses_exp) {
  return kLdsHeaderBytes + ((table_bytes + 7) & ~7) + (uses_normal ? 768 * 8 : 0) + (uses_exp ? 768 * 8 : 0);
}

// the largest launch (full tables, both ziggurats, a fused rollout's env + update-fn streams) must fit the 64 KB of dynamic LDS
// a kernel gets without opting in to more
static_assert(kBlock != 256 || lds_bytes_for(kMaxTableBytes, 1, 1) + kLdsStreamBytes * (1 + kMaxLdsUpd) <= 65536, "dynamic LDS budget");

// ... and a fused rollout of the C2 shape (normal ziggurat, env streams + one update-fn stream in LDS) must keep FIVE workgroups per
// CU (160 KB of LDS; its 94 VGPRs allow five): at 32.9 KB each it was four, and 11 % slower
static_assert(kBlock != 256 || lds_bytes_for(256, 1, 0) + kLdsStreamBytes * 2 <= 160 * 1024 / 5, "LDS per workgroup of the C2-shaped rollout");

// Cooperative staging of the constant tables into LDS (once per workgroup).
// DIRECT (single-step launches of a specialised unit whose config has no table blob, NSG_TABLES_DIRECT): nothing is staged.  A
// launch of up to 1536 chunks runs one chunk per workgroup, i.e. it would stage the ziggurat tables - a global-load round trip,
// the LDS writes and a barrier ahead of everything else - for 256 env-steps each, of which a third (C2) or none (C1) draws a
// normal at all; the few lookups read the tables' global copies (6 KB, cache-resident after the first launch) instead.  Measured
// (profiles/r03_ab_runs.txt): C2 at 65 536 envs 7.23 -> 6.88 us, at 2^18 12.1 -> 11.4, at 2^20 33.05 -> 32.44; C1 at 65 536 envs 5.42 ->
// 5.20 (no barrier left in the kernel).  Not for the grid envs, whose map is looked up by every lane on every step (C3 +10-15 %),
// nor for configs with schedule / value tables, nor for fused rollouts (K steps of lookups per launch).
// KIND: what the caller knows about the segment at compile time - 0 nothing (nsg_theta_trace on an unbound handle: run-time tests),
// 1 a bound classic-control batch (its stream descriptor is fetched through the scalar cache, no dependent tests), 2 a grid env
// (no descriptor).
template <bool DIRECT = false, bool HANDOVER = true, int KIND = 0>
__device__ __forceinline__ void stage_tables(const Segment& sg, LdsTables& lds, Tables& tb, ZigLds& zg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char nsg_dyn_lds[];
  unsigned char* base = nsg_dyn_lds;
  lds.reset_n = (int*)(base + 16);
  lds.reset_list = (short*)(base + 32);
  lds.reset_state = (double*)(base + 32 + kBlock * 2);
  lds.blob = (uint64_t*)(base + kLdsHeaderBytes);
  lds.zig = (uint64_t*)(base + kLdsHeaderBytes + ((sg.table_bytes + 7) & ~7));
  lds.streams = (uint64_t*)(base + lds_bytes_for(sg.table_bytes, sg.uses_normal, sg.uses_exp));  // valid when the launch reserved it (fused rollouts)
  lds.ustreams = lds.streams + kBlock * 4;
  const int tid = threadIdx.x;
  if constexpr (DIRECT) {
    if constexpr (HANDOVER) {   // only the reset hand-over's queue counters still live in LDS
      if (tid == 0) lds.reset_n[0] = lds.reset_n[1] = 0;
      __syncthreads();
    }
    tb.base = sg.tables;
    zg.ki = sg.zig;
    zg.wi = (const double*)(sg.zig + 256);
    zg.fi = (const double*)(sg.zig + 512);
    zg.ke = sg.zig + 768;
    zg.we = (const double*)(sg.zig + 1024);
    zg.fe = (const double*)(sg.zig + 1280);
    zg.jump = sg.jump;
    // the batch's stream descriptor: two uniform words that no launch of this kernel writes - read through the scalar cache, and
    // without the env-type / bound-handle tests of the general path (DIRECT is a bound classic-control step launch by construction),
    // each of which would be one more dependent memory round trip ahead of the first row load
    typedef const __attribute__((address_space(4))) uint64_t* scalar_words;
    scalar_words d = (scalar_words)sg.buf.rng_env;
    zg.sd0 = d[0];
    zg.sd1 = d[1];
    return;
  }
  uint64_t* zexp = lds.zig + (sg.uses_normal ? 768 : 0);
  if (sg.uses_normal) {
    for (int k = tid; k < 768; k += kBlock) lds.zig[k] = sg.zig[k];
  }
  if (sg.uses_exp) {
    for (int k = tid; k < 768; k += kBlock) zexp[k] = sg.zig[768 + k];
  }
  const int words = (sg.table_bytes + 7) >> 3;
  const uint64_t* src = (const uint64_t*)sg.tables;
  for (int k = tid; k < words; k += kBlock) lds.blob[k] = src[k];
  if (tid == 0) lds.reset_n[0] = lds.reset_n[1] = 0;
  __syncthreads();
  tb.base = (const uint8_t*)lds.blob;
  zg.ki = lds.zig;
  zg.wi = (const double*)(lds.zig + 256);
  zg.fi = (const double*)(lds.zig + 512);
  zg.ke = zexp;
  zg.we = (const double*)(zexp + 256);
  zg.fe = (const double*)(zexp + 512);
  zg.jump = sg.jump;
  if constexpr (KIND == 1) {
    typedef const __attribute__((address_space(4))) uint64_t* scalar_words;
    scalar_words d = (scalar_words)sg.buf.rng_env;
    zg.sd0 = d[0];
    zg.sd1 = d[1];
  } else if constexpr (KIND == 0) {
    if (sg.cfg.env_type <= NSG_ENV_MOUNTAINCAR_CONT && sg.buf.rng_env) {  // classic-control env types; unbound handles (nsg_theta_trace) have no rows
      zg.sd0 = sg.buf.rng_env[0];
      zg.sd1 = sg.buf.rng_env[1];
    }
  }
}

// Where one step's per-env outputs go: the handle's own buffers (nsg_step) or the k-th slice of
// a trajectory (nsg_rollout).  Wave-uniform, lives in SGPRs.
struct StepOut {
  float* obs;
  float* reward;
  uint8_t* terminated;
  uint8_t* truncated;
  uint8_t* env_change;
  float* delta_change;
};
__device__ __forceinline__ StepOut default_out(const nsg_buffers& b) {
  return StepOut{b.obs, b.reward, b.terminated, b.truncated, b.env_change, b.delta_change};
}

// Per-wavefront running counts (lane-uniform), added to the wavefront's own counter shard when it retires.
struct WaveCounts {
  unsigned done = 0, fired = 0, viol = 0, steps = 0, lc_exhausted = 0, sched_overrun = 0;
};

// `block_rel` = workgroup index within the handle's launch range (beyond NSG_CNT_SHARDS / 4 workgroups the shards are shared).
__device__ __forceinline__ void flush_counts(uint64_t* counters, int block_rel, const WaveCounts& wc) {
  const int lane = threadIdx.x & 63;
  if (counters && lane < NSG_CNT_COUNT) {
    const unsigned v = lane == NSG_CNT_DONE ? wc.done : lane == NSG_CNT_FIRED ? wc.fired : lane == NSG_CNT_VIOLATION ? wc.viol
                     : lane == NSG_CNT_STEPS ? wc.steps : lane == NSG_CNT_LC_EXHAUSTED ? wc.lc_exhausted : wc.sched_overrun;
    // fire-and-forget add (no return value requested): the wavefront retires without waiting for a
    // read-modify-write round trip; up to 4096 workgroups the shard has a single owner per launch, so there is no contention
    if (v) atomicAdd((unsigned long long*)&counters[(int64_t)lane * kCntShards + ((block_rel * (kBlock / 64) + (threadIdx.x >> 6)) & (kCntShards - 1))],
                     (unsigned long long)v);
  }
}

template <int N_> struct Unroll {
  template <typename F> static __device__ __forceinline__ void run(F&& f) {
    Unroll<N_ - 1>::run(f);
    f(N_ - 1);
  }
};
template <> struct Unroll<0> {
  template <typename F> static __device__ __forceinline__ void run(F&&) {}
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// a step output: write-once (non-temporal) - or written through, when another running kernel is about to read it (IoMode::coh)
template <typename T> __device__ __forceinline__ void stg_o(bool coh, T* base, uint32_t byte_off, T v) {
  if (coh) {
    NSG_GLOBAL char* p = (NSG_GLOBAL char*)pin_sgpr((uint64_t)base);
    __hip_atomic_store((NSG_GLOBAL T*)(p + byte_off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    stg_out(base, byte_off, v);
  }
}
template <typename T> __device__ __forceinline__ T ldg_in(bool coh, const T* base, uint32_t byte_off) {
  if (coh) {
    const NSG_GLOBAL char* p = (const NSG_GLOBAL char*)pin_sgpr((uint64_t)base);
    return __hip_atomic_load((const NSG_GLOBAL T*)(p + byte_off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return ldg(base, byte_off);
}
template <int ENV> __device__ __forceinline__ void store_obs(float* obs, int64_t i, const float* o, bool coh = false) {
  const uint32_t u = (uint32_t)i;
  if (coh) {   // resident stepper: element by element through the L2 (IoMode::coh)
#pragma unroll
    for (int k = 0; k < EnvTraits<ENV>::OBS; k++) stg_o(true, obs, u * (uint32_t)(4 * EnvTraits<ENV>::OBS) + 4u * (uint32_t)k, o[k]);
    return;
  }
  if constexpr (ENV == NSG_ENV_CARTPOLE) {
    stg_out(reinterpret_cast<f32x4*>(obs), u * 16u, f32x4{o[0], o[1], o[2], o[3]});
  } else if constexpr (ENV == NSG_ENV_PENDULUM) {
    stg(obs, u * 12u, o[0]); stg(obs, u * 12u + 4u, o[1]); stg(obs, u * 12u + 8u, o[2]);
  } else if constexpr (ENV == NSG_ENV_ACROBOT) {
    f32x2* q = reinterpret_cast<f32x2*>(obs);
    stg(q, u * 24u, f32x2{o[0], o[1]}); stg(q, u * 24u + 8u, f32x2{o[2], o[3]});
    stg(q, u * 24u + 16u, f32x2{o[4], o[5]});
  } else {
    stg(reinterpret_cast<f32x2*>(obs), u * 8u, f32x2{o[0], o[1]});
  }
}

// Envs whose constraint checker looks at one proposal at a time (everything but Acrobot, whose
// link-length / centre-of-mass checks cross-reference other proposals, classic_control.py:241-357).
template <int ENV> __device__ __forceinline__ bool own_constraint_violated(int slot, double n) {
  if constexpr (ENV == NSG_ENV_CARTPOLE) {  // :208-235  gravity masscart masspole force_mag tau length
    return slot == 0 ? n < 0 : (slot == 1 || slot == 2 || slot == 5) ? n <= 0 : false;
  } else if constexpr (ENV == NSG_ENV_PENDULUM) {  // :389-420  m l dt g
    return slot == 3 ? n < 0 : n <= 0;
  } else {  // MountainCar :359-376 gravity force;  MountainCarContinuous :378-387 power
    return n <= 0;
  }
}

// Per-lane copy of an env's persistent rows.  nsg_step round-trips them through HBM on every launch
// (load = store = true); nsg_rollout keeps them in registers between the K fused steps of a launch
// and touches HBM only on the first / last step.
template <int ENV> struct LaneState {
  double s[EnvTraits<ENV>::PHYS] = {};
  double th0 = 0.0, th1 = 0.0;   // θ rows 0 and 1 (rows >= 2 always go through memory)
  int t = 0;
  unsigned st = 0;
  float er = 0.f;
  // planning copies only: TimeLimit origin (t_fork) and CartPole's fork-time total_mass / polemass_length (derived)
  int tf = 0;
  double d0 = 0.0, d1 = 0.0;
  // fused policy rollouts only (IoMode::act_lane; dead everywhere else): the action this lane takes next, and what its last step
  // produced - the float32 observation an agent sees, the base MDP's float64 reward, terminated | truncated << 1 | transition taken << 2
  int ai = 0;
  float af = 0.f;
  float o[EnvTraits<ENV>::OBS] = {};
  double rw = 0.0;
  unsigned fl = 0;
};
struct IoMode {  // wave-uniform
  bool load;   // fetch the persistent rows from memory (else: they are in the LaneState)
  bool store;  // write them back
  bool dirty;  // the LaneState's θ rows 0/1 may differ from memory (earlier fused steps did not store)
  bool lds_rng;  // fused rollouts: the chunk's env streams (derived once per launch) and its first stochastic update fns' PCG64
                 // records are in LDS: a reset inside the launch is four sequential draws, not a re-derivation
  bool wt = false;  // single-step launches (nsg_step): persistent rows of the grid envs / Pendulum state leave through
                    // agent-scope stores (stg_wt); fused rollouts keep plain stores (C3 rollout: 12.3 vs 14.6 us per step)
  bool coh = false; // resident stepper (nsg_resident_start): another kernel that is running NOW reads this step's outputs and wrote
                    // its actions.  They are accessed coherently one by one - outputs written through the L2 (agent-scope stores),
                    // actions read past it (agent-scope loads) - so that the hand-over needs no L2-wide writeback / invalidate:
                    // with 32 workgroups per XCD each issuing its own pair of cache-wide fences a resident step cost 20 us
                    // (profiles/NOTEBOOK.md, round 4)
  bool opt_out = false;   // fused rollouts, every step but the last: a step output whose pointer is NULL is not stored at all (the caller does
                          // not record that row; the handle's own row is written by the launch's last step)
  bool act_lane = false;  // fused policy rollouts (nsg_rollout_policy): the action comes from the lane's own registers (LaneState::ai / af,
                          // GridLane::ai), and the step leaves its observation, float64 reward and flags there for the next decision
};
template <typename T> __device__ __forceinline__ void stg_p(bool wt, T* base, uint32_t byte_off, T v) {
  if (wt) stg_wt(base, byte_off, v);
  else stg(base, byte_off, v);
}

// Fire predicate of param p of env i.  Deterministic schedulers are pure functions of t; the
// stochastic ones (FULL builds) advance their own PCG64 record and, for Memoryless, transition_time.
// On a non-persistent reset the scheduler is rewound to its construction state together with the
// rest of the deep-copied init_initial_params (base.py:381-384).
template <bool FULL>
__device__ __forceinline__ bool fire_param(const nsg_config& cfg, const nsg_buffers& b, const Tables& tb, const ZigLds& zg,
                                           int64_t N, int64_t i, int p, int t, bool eval, bool rewind, unsigned& overrun) {
  const nsg_param_cfg& pc = cfg.params[p];
  if (eval && sched_overrun(pc, t)) overrun = 1u;
  if constexpr (FULL) {
    if (sched_is_stochastic(pc.sched_kind)) {
      bool f = false;
      if (eval) {
        Pcg r;
        pcg_load(b.rng_sched + (int64_t)pc.sched_slot * 4 * N, N, i, r);
        int next = pc.sched_kind == NSG_SCHED_MEMORYLESS ? b.sched_next[(int64_t)pc.sched_slot * N + i] : 0;
        f = sched_fire_stoch(pc, zg, t, r, next);
        pcg_store_state(b.rng_sched + (int64_t)pc.sched_slot * 4 * N, N, i, r);
        if (pc.sched_kind == NSG_SCHED_MEMORYLESS) b.sched_next[(int64_t)pc.sched_slot * N + i] = next;
      }
      if (rewind) {
        Pcg r;
        int next;
        sched_construct(pc, zg, pc.sched_slot, i, r, next);  // the shared object's own construction stream
        pcg_store_all(b.rng_sched + (int64_t)pc.sched_slot * 4 * N, N, i, r);
        b.sched_next[(int64_t)pc.sched_slot * N + i] = next;
      }
      return f;
    }
  }
  return eval && sched_fire(pc, tb, t);
}

// Update-fn stream of the entry `pc`: memory record, or - during a fused rollout - the lane's LDS copy.
__device__ __forceinline__ void upd_stream_load(const nsg_config& cfg, const nsg_buffers& b, const LdsTables& lds, const nsg_param_cfg& pc,
                                                int64_t N, int64_t i, const IoMode io, Pcg& r) {
  const int li = io.lds_rng ? upd_lds_index(cfg, pc.fn_slot) : kMaxLdsUpd;
  if (li < kMaxLdsUpd) {
    const uint64_t* rec = lds.ustreams + ((int64_t)li * kBlock + threadIdx.x) * 4;
    r.sh = rec[0]; r.sl = rec[1]; r.ih = rec[2]; r.il = rec[3];
  } else {
    pcg_load(b.rng_upd + (int64_t)pc.fn_slot * 4 * N, N, i, r);
  }
}
__device__ __forceinline__ void upd_stream_store(const nsg_config& cfg, const nsg_buffers& b, const LdsTables& lds, const nsg_param_cfg& pc,
                                                 int64_t N, int64_t i, const IoMode io, const Pcg& r) {
  const int li = io.lds_rng ? upd_lds_index(cfg, pc.fn_slot) : kMaxLdsUpd;
  if (li < kMaxLdsUpd) {
    uint64_t* rec = lds.ustreams + ((int64_t)li * kBlock + threadIdx.x) * 4;
    rec[0] = r.sh; rec[1] = r.sl;
  } else {
    pcg_store_state(b.rng_upd + (int64_t)pc.fn_slot * 4 * N, N, i, r);
  }
}

// ============================================================================================
// classic-control step for one chunk of kBlock envs (the whole workgroup; one env per lane).
//   phase 1: every lane steps its env (θ-engine, constraints, integrator, outputs); an env that
//            ended an episode on the previous call takes no transition and is queued in LDS
//   phase 2: the first `reset_n` lanes of the workgroup perform the queued resets (PCG64 draws of
//            the initial state).  ~5 % of the envs reset per step but they sit in ~96 % of the
//            wavefronts; compacting them per workgroup lets ONE wavefront execute the draw path
//            instead of all of them.
// ============================================================================================
template <int ENV, bool FULL>
__device__ __forceinline__ void step_chunk(const nsg_config& cfg, const nsg_buffers& b, const int64_t N, const Tables& tb, const ZigLds& zg, const void* actions,
                                           const StepOut& out, int64_t base, int parity, LdsTables& lds, WaveCounts& wc,
                                           LaneState<ENV>& ls, const IoMode io) {
  using T = EnvTraits<ENV>;
  const int P = cfg.n_params;
  const bool persistent = (cfg.flags & NSG_F_PERSISTENT_PARAMS) != 0;
  // planning copies (get_planning_env / __deepcopy__): θ frozen unless in_sim_change
  // (classic_control.py:70-75); TimeLimit counts from the fork (t_fork)
  const bool sim = (cfg.flags & NSG_F_SIM_ENV) != 0;
  const bool theta_live = !(sim && !(cfg.flags & NSG_F_IN_SIM_CHANGE));
  const int tid = threadIdx.x;
  const int64_t i = base + tid;
  const bool active = i < N;

  const uint32_t o1 = (uint32_t)i, o4 = o1 * 4u, o8 = o1 * 8u;  // per-lane byte offsets (N <= 2^27)
  const bool track = (cfg.flags & NSG_F_TRACK_RETURNS) != 0;
  const int t = !active ? 0 : io.load ? ldg(b.t, o4) : ls.t;
  if (sim && io.load && active) ls.tf = ldg(b.t_fork, o4);
  // episode word: bit 0 = needs reset, bits 1-31 = resets this env has drawn from its np_random so far (nsgym_hip.h).
  // An env type that never terminates (Pendulum) ends its episodes by TimeLimit alone, so "needs reset" IS t >= max_episode_steps:
  // its word carries the count only, a single step reads it in the lanes that reset (all of them on the same launch when the
  // batch was started together, none on the other 199) and rewrites it there - 8 B per env-step less in dense rows.  Planning
  // copies (TimeLimit counted from the fork; a copy of a finished env must still reset first) keep the stored bit.
  // NSG_F_NO_AUTORESET (gymnasium after `done`, what the reference's single wrappers do: base.py:313): nothing resets inside a step;
  // bit 0 of the episode word then means "terminated at an earlier step of this episode" (CartPole's steps_beyond_terminated)
  const bool noauto = (cfg.flags & NSG_F_NO_AUTORESET) != 0;
  const bool reset_from_t = T::NEVER_TERMINATES && !sim && !noauto;
  unsigned st;
  bool do_reset;
  if (reset_from_t) {
    do_reset = active && cfg.max_episode_steps > 0 && t >= cfg.max_episode_steps;
    // fused rollouts carry the count in registers from their first step on (its dense load positions the lane's LDS stream)
    st = !active ? 0u : !io.load ? ls.st : (do_reset || io.lds_rng) ? (unsigned)ldg(b.episode, o4) & ~NSG_ST_NEEDS_RESET : 0u;
  } else {
    st = !active ? 0u : io.load ? (unsigned)ldg(b.episode, o4) : ls.st;
    do_reset = active && (st & NSG_ST_NEEDS_RESET) && !noauto;
  }
  const bool do_step = active && !do_reset;
  const bool ld_state = NSG_UNCOND_LOADS ? active : do_step;
  double s[T::PHYS];
#pragma unroll
  for (int k = 0; k < T::PHYS; k++) s[k] = !ld_state ? 0.0 : io.load ? ldg(b.phys, blk_off8(T::PHYS, k, i)) : ls.s[k];
  int ai = 0;
  float af = 0.f;
  if (ld_state) {
    if (io.act_lane) { ai = ls.ai; af = ls.af; }
    else if constexpr (T::FLOAT_ACT) af = ldg_in(io.coh, (const float*)actions, o4);
    else ai = ldg_in(io.coh, (const int32_t*)actions, o4);
  }
  constexpr bool kReturnFromT = T::RETURN_PER_STEP != 0.f;  // the return is a function of t: no running row (nsg_envs.hip.h)
  float er = 0.f;
  if (track && do_step && !kReturnFromT) er = io.load ? ldg(b.ep_return, o4) : ls.er;
  double pre0 = 0.0, pre1 = 0.0;
  if (active && P > 0) pre0 = io.load ? ldg(b.theta, o8) : ls.th0;
  if (active && P > 1) pre1 = io.load ? ldg(b.theta + N, o8) : ls.th1;

  double th[T::NTHETA + T::NDERIVED];
#pragma unroll
  for (int k = 0; k < T::NTHETA; k++) th[k] = cfg.base_theta[k];
  unsigned n_fired = 0, n_viol = 0, n_overrun = 0;

  if constexpr (ENV != NSG_ENV_ACROBOT) {
    // ---- single pass: propose, check, commit (classic_control.py:80-92) ----------------------
    for (int p = 0; p < P; p++) {
      const nsg_param_cfg& pc = cfg.params[p];
      const int slot = pc.theta_slot;
      const double c = !active ? cfg.base_theta[slot] : p == 0 ? pre0 : p == 1 ? pre1 : ldg(b.theta + (int64_t)p * N, o8);
      double n = c;
      bool fired = fire_param<FULL>(cfg, b, tb, zg, N, i, p, t, do_step && theta_live, do_reset && !persistent, n_overrun);
      if (fired) {
        Pcg r = {0, 0, 0, 0};
        int cursor = 0;
        const bool has_cur = upd_uses_cursor(pc.upd_kind);
        if (FULL && pc.uses_rng) upd_stream_load(cfg, b, lds, pc, N, i, io, r);
        if (has_cur) cursor = ldg(b.cursor + (int64_t)pc.fn_slot * N, o4);
        n = upd_scalar<FULL>(pc, tb, zg, c, t, r, cursor);
        if (FULL && pc.uses_rng) upd_stream_store(cfg, b, lds, pc, N, i, io, r);
        if (has_cur) stg(b.cursor + (int64_t)pc.fn_slot * N, o4, cursor);
      }
      const bool rejected = own_constraint_violated<ENV>(slot, n);
      double delta = fired ? n - c : 0.0;  // UpdateFn._get_delta_change, base.py:182
      double fin = rejected ? c : n;
      if (rejected) {
        n_viol += fired ? 1u : 0u;
        fired = false;
        delta = 0.0;
      }
      if (do_reset) {  // base.py:381-384 + classic_control.py:105-107; streams continue (base.py:389-391)
        fin = persistent ? c : cfg.base_theta[slot];
        if (!persistent && upd_uses_cursor(pc.upd_kind)) stg(b.cursor + (int64_t)pc.fn_slot * N, o4, 0);
      }
      Unroll<T::NTHETA>::run([&](int k) {
        if (k == slot) th[k] = fin;
      });
      if (active) {
        if (p == 0) ls.th0 = fin;
        if (p == 1) ls.th1 = fin;
        // rows 0/1 live in registers across fused steps: written when this step stores (always then,
        // because earlier fused steps may have changed them without a store)
        if (p > 1 ? fin != c : (io.store && (fin != c || io.dirty))) stg(b.theta + (int64_t)p * N, o8, fin);
        if (!io.opt_out || out.env_change) stg_o(io.coh, out.env_change + (int64_t)p * N, o1, (uint8_t)(fired ? 1 : 0));
        if (!io.opt_out || out.delta_change) stg_o(io.coh, out.delta_change + (int64_t)p * N, o4, (float)delta);
        if (cfg.flags & NSG_F_VIOLATION_MASK) stg(b.violation + (int64_t)p * N, o1, (uint8_t)(rejected && do_step ? 1 : 0));
      }
      n_fired += fired ? 1u : 0u;
    }
  } else {
    // ---- Acrobot: all proposals first, then the cross-referencing checker --------------------
    double cur[T::NTHETA], nv[T::NTHETA];
#pragma unroll
    for (int k = 0; k < T::NTHETA; k++) cur[k] = nv[k] = cfg.base_theta[k];
    unsigned tuned = 0, firedmask = 0;
    for (int p = 0; p < P; p++) {
      const nsg_param_cfg& pc = cfg.params[p];
      const int slot = pc.theta_slot;
      const double c = !active ? cfg.base_theta[slot] : p == 0 ? pre0 : p == 1 ? pre1 : ldg(b.theta + (int64_t)p * N, o8);
      double n = c;
      if (fire_param<FULL>(cfg, b, tb, zg, N, i, p, t, do_step && theta_live, do_reset && !persistent, n_overrun)) {
        Pcg r = {0, 0, 0, 0};
        int cursor = 0;
        const bool has_cur = upd_uses_cursor(pc.upd_kind);
        if (FULL && pc.uses_rng) upd_stream_load(cfg, b, lds, pc, N, i, io, r);
        if (has_cur) cursor = ldg(b.cursor + (int64_t)pc.fn_slot * N, o4);
        n = upd_scalar<FULL>(pc, tb, zg, c, t, r, cursor);
        if (FULL && pc.uses_rng) upd_stream_store(cfg, b, lds, pc, N, i, io, r);
        if (has_cur) stg(b.cursor + (int64_t)pc.fn_slot * N, o4, cursor);
        firedmask |= 1u << p;
      }
      Unroll<T::NTHETA>::run([&](int k) {
        if (k == slot) { cur[k] = c; nv[k] = n; }
      });
      tuned |= 1u << slot;
    }
    const unsigned viol = constraint_mask<ENV>(nv, cur, tuned);
#pragma unroll
    for (int k = 0; k < T::NTHETA; k++) th[k] = cur[k];
    for (int p = 0; p < P; p++) {
      const nsg_param_cfg& pc = cfg.params[p];
      const int slot = pc.theta_slot;
      double c = 0.0, n = 0.0;
      Unroll<T::NTHETA>::run([&](int k) {
        if (k == slot) { c = cur[k]; n = nv[k]; }
      });
      bool fired = (firedmask >> p) & 1u;
      const bool rejected = (viol >> slot) & 1u;
      double delta = fired ? n - c : 0.0;
      double fin = rejected ? c : n;
      if (rejected) {
        n_viol += fired ? 1u : 0u;
        fired = false;
        delta = 0.0;
      }
      if (do_reset) {
        fin = persistent ? c : cfg.base_theta[slot];
        if (!persistent && upd_uses_cursor(pc.upd_kind)) stg(b.cursor + (int64_t)pc.fn_slot * N, o4, 0);
      }
      Unroll<T::NTHETA>::run([&](int k) {
        if (k == slot) th[k] = fin;
      });
      if (active) {
        if (p == 0) ls.th0 = fin;
        if (p == 1) ls.th1 = fin;
        // rows 0/1 live in registers across fused steps: written when this step stores (always then,
        // because earlier fused steps may have changed them without a store)
        if (p > 1 ? fin != c : (io.store && (fin != c || io.dirty))) stg(b.theta + (int64_t)p * N, o8, fin);
        if (!io.opt_out || out.env_change) stg_o(io.coh, out.env_change + (int64_t)p * N, o1, (uint8_t)(fired ? 1 : 0));
        if (!io.opt_out || out.delta_change) stg_o(io.coh, out.delta_change + (int64_t)p * N, o4, (float)delta);
        if (cfg.flags & NSG_F_VIOLATION_MASK) stg(b.violation + (int64_t)p * N, o1, (uint8_t)(rejected && do_step ? 1 : 0));
      }
      n_fired += fired ? 1u : 0u;
    }
  }

  // ---- base MDP transition with the updated θ (base.py:313) ----------------------------------
  double reward = 0.0;
  bool term = false, trunc = false;
  int tnew = 0;
  if constexpr (ENV == NSG_ENV_CARTPOLE) {
    if (sim && !theta_live && b.derived) {  // frozen planning copy: the resolver never runs again
      if (io.load && active) {
        ls.d0 = ldg(b.derived, o8);
        ls.d1 = ldg(b.derived + N, o8);
      }
      th[6] = do_step ? ls.d0 : 0.0;
      th[7] = do_step ? ls.d1 : 0.0;
    } else {  // _dependency_resolver, classic_control.py:426-444
      th[6] = th[2] + th[1];
      th[7] = th[5] * th[2];
    }
  }
  if (do_step) {
    term = env_step<ENV>(th, s, ai, af, reward);
    if constexpr (ENV == NSG_ENV_CARTPOLE) {
      // CartPoleEnv.step [UPSTREAM]: a step that finds the pole down AGAIN (steps_beyond_terminated is not None) pays 0.0
      if (noauto && term && (st & NSG_ST_NEEDS_RESET)) reward = 0.0;
    }
    tnew = t + 1;  // base.py:314
    // TimeLimit [UPSTREAM] counts the steps of ITS env: a planning copy restarts at the fork
    const int elapsed = tnew - (sim ? ls.tf : 0);
    trunc = cfg.max_episode_steps > 0 && elapsed >= cfg.max_episode_steps;
  }
  if (sim && do_reset) {
    stg(b.t_fork, o4, 0);
    ls.tf = 0;
  }
  const bool done = term || trunc;

  // ---- resets: queue -> helper lanes re-derive the streams and draw -> owners read back -----------------------------
  // ~5 % of the envs reset per step but they sit in ~96 % of the wavefronts; compacting them per workgroup lets ONE wavefront
  // execute the seeding + jump-ahead + draw path.  No stream state is read from or written to memory (nsg_rng.hip.h, pcg_at):
  // the owner hands its episode count over through LDS, the helper rebuilds PCG64(SeedSequence(seed_i)) at draw D * count.
  if constexpr (T::RESET_IN_LANE) {
    // rare resets (nsg_envs.hip.h): every resetting lane re-derives its own stream and draws; no queue, no barrier
    if (do_reset) {
      Pcg g;
      if (io.lds_rng) {   // fused rollout: the lane's stream sits in LDS, positioned at its next episode
        uint64_t* rec = lds.streams + tid * 4;
        g.sh = rec[0]; g.sl = rec[1]; g.ih = rec[2]; g.il = rec[3];
        env_reset_draw<ENV>(g, s);
        rec[0] = g.sh; rec[1] = g.sl;
      } else {            // jump straight to the state whose output is the episode's first draw
        const u64x2 desc = {zg.sd0, zg.sd1};
        env_stream_at<2>(b.rng_env, i, (uint64_t)(st >> NSG_EP_COUNT_SHIFT) * (uint64_t)T::RESET_DRAWS, zg.jump, g, &desc);
        env_reset_draw<ENV, true>(g, s);
      }
    }
  } else
  {
    constexpr int D = T::RESET_DRAWS;
    static_assert((64 % D) == 0, "the D helper lanes of one reset must sit in one wavefront (see the count read below)");
    int* rn = lds.reset_n + (parity & 1);
    // the owner queues its lane and leaves its episode count in the first word of its own result slot - as the BIT PATTERN of a
    // double, so that every access to these words goes through one type (a uint64 view of the same words would be "no alias" to
    // the compiler, which could then move the count's load past the helper's store below)
    if (do_reset) {
      lds.reset_list[atomicAdd(rn, 1)] = (short)tid;
      lds.reset_state[tid * 4] = __longlong_as_double((long long)(st >> NSG_EP_COUNT_SHIFT));
    }
    __syncthreads();
    const int n_reset = *rn;
    if (tid == 0) lds.reset_n[(parity + 1) & 1] = 0;  // the other buffer is idle until the next chunk
    if (io.lds_rng) {
      // fused rollout: the owner's stream sits in LDS, positioned at its next episode - D sequential draws by one helper lane
      if (tid < n_reset) {
        const int owner = lds.reset_list[tid];
        uint64_t* rec = lds.streams + owner * 4;
        Pcg g = {rec[0], rec[1], rec[2], rec[3]};
        double r0[T::PHYS];
        env_reset_draw<ENV>(g, r0);
        rec[0] = g.sh; rec[1] = g.sl;
#pragma unroll
        for (int k = 0; k < T::PHYS; k++) lds.reset_state[owner * 4 + k] = r0[k];
      }
    } else {
      // single step: D helper lanes per reset, lane (q, j) produces draw j of reset q on its own - gymnasium reset() ->
      // np_random draws of the initial state [UPSTREAM]: seed -> T0, ONE jump to the state after draw D * count + j's step,
      // output.  No sequential PCG64 steps at all; a typical chunk's ~13 resets fill one wavefront.
#if NSG_RESET_LANE_PER_DRAW
      for (int h = tid; h < n_reset * D; h += kBlock) {
        const int owner = lds.reset_list[h / D], j = h % D;
        // the count shares its word with result 0: the D lanes of a reset are neighbours in ONE wavefront (64 % D == 0) and a
        // wavefront's LDS operations execute in issue order, so all of them have read the count before any of them stores a
        // result; the wave barrier (no instruction: a scheduling fence) keeps the compiler from moving the load below the store
        const uint64_t count = (uint64_t)__double_as_longlong(lds.reset_state[owner * 4]);
        __builtin_amdgcn_wave_barrier();
        Pcg g;
        const u64x2 desc = {zg.sd0, zg.sd1};
        env_stream_at<2>(b.rng_env, base + owner, count * (uint64_t)D + (uint64_t)j, zg.jump, g, &desc);
        lds.reset_state[owner * 4 + j] = env_reset_map<ENV>(j, pcg_double_out(g));
      }
#else   // one helper lane per reset: one jump, then D - 1 sequential PCG64 steps
      if (tid < n_reset) {
        const int owner = lds.reset_list[tid];
        const uint64_t count = (uint64_t)__double_as_longlong(lds.reset_state[owner * 4]);
        __builtin_amdgcn_wave_barrier();
        Pcg g;
        const u64x2 desc = {zg.sd0, zg.sd1};
        env_stream_at<2>(b.rng_env, base + owner, count * (uint64_t)D, zg.jump, g, &desc);
        double r0[T::PHYS];
        env_reset_draw<ENV, true>(g, r0);
#pragma unroll
        for (int k = 0; k < D; k++) lds.reset_state[owner * 4 + k] = r0[k];
      }
#endif
    }
    __syncthreads();
    if (do_reset) {
#pragma unroll
      for (int k = 0; k < T::PHYS; k++) s[k] = k < D || io.lds_rng ? lds.reset_state[tid * 4 + k] : 0.0;
    }
  }

  // a reset consumed one more episode of the env's stream
  const unsigned bit0 = noauto ? ((st & NSG_ST_NEEDS_RESET) | (term ? NSG_ST_NEEDS_RESET : 0u)) : (done && !reset_from_t ? NSG_ST_NEEDS_RESET : 0u);
  const unsigned stw = bit0 | (((st >> NSG_EP_COUNT_SHIFT) + (do_reset ? 1u : 0u)) << NSG_EP_COUNT_SHIFT);
#pragma unroll
  for (int k = 0; k < T::PHYS; k++) ls.s[k] = s[k];
  ls.t = tnew;
  ls.st = stw;
  if (active) {  // every row is written by its owner lane: fully coalesced stores
    if (io.store) {
#pragma unroll
      for (int k = 0; k < T::PHYS; k++) {
        // agent-scope (write-through) stores of the integrator state pay for Pendulum only (19.5 -> 18.8 us; CartPole and
        // its full engine: +1 %), like for the grid envs' rows (nsg_rng.hip.h, stg_wt)
        if constexpr (ENV == NSG_ENV_PENDULUM) stg_p(io.wt, b.phys, blk_off8(T::PHYS, k, i), s[k]);
        else stg(b.phys, blk_off8(T::PHYS, k, i), s[k]);
      }
    }
    float o[T::OBS];
    env_obs<ENV>(s, o);
    if (!io.opt_out || out.obs) store_obs<ENV>(out.obs, i, o, io.coh);
    if (io.act_lane) {
#pragma unroll
      for (int k = 0; k < T::OBS; k++) ls.o[k] = o[k];
      ls.rw = reward;
      ls.fl = (term ? 1u : 0u) | (trunc ? 2u : 0u) | (do_step ? 4u : 0u);
    }
    if (io.store) stg(b.t, o4, tnew);
    if (!io.opt_out || out.reward) stg_o(io.coh, out.reward, o4, (float)reward);
    if (!io.opt_out || out.terminated) stg_o(io.coh, out.terminated, o1, (uint8_t)(term ? 1 : 0));
    if (!io.opt_out || out.truncated) stg_o(io.coh, out.truncated, o1, (uint8_t)(trunc ? 1 : 0));
    if (io.store && (!reset_from_t || do_reset || io.lds_rng)) stg(b.episode, o4, (int32_t)stw);
    if (track) {  // the episode length is the wrapper time t: only the return needs a running row
      if constexpr (kReturnFromT) {
        // one scattered store per finished episode, not two: the return of these env types IS +-length (buffers.last_return is
        // not allocated for them; VecNSEnv.episode_returns derives it)
        if (done) stg(b.last_length, o4, tnew);
      } else {
        er += (float)reward;
        if (done) {
          stg(b.last_return, o4, er);
          stg(b.last_length, o4, tnew);
          er = 0.f;
        }
        ls.er = er;
        if (io.store) stg(b.ep_return, o4, er);
      }
    }
  }

  // ---- wavefront ballots: done-mask word + counters ------------------------------------------
  const unsigned long long done_mask = __ballot(done);
  if (b.done_bits && (tid & 63) == 0 && i < N) b.done_bits[i >> 6] = done_mask;
  wc.done += __popcll(done_mask);
  wc.steps += __popcll(__ballot(do_step));
  for (unsigned bit = 0; bit < 4; bit++) {  // per-lane small integers summed with one ballot per bit
    wc.fired += __popcll(__ballot((n_fired >> bit) & 1u)) << bit;
    wc.viol += __popcll(__ballot((n_viol >> bit) & 1u)) << bit;
  }
  wc.sched_overrun += __popcll(__ballot(n_overrun != 0));
}

// ============================================================================================
// Grid envs (FrozenLake, CliffWalking, Bridge): integer / categorical path, one env per lane.
// FrozenLake and CliffWalking are bit-exact with the reference incl. the env-stream draws.
// ============================================================================================
__host__ __device__ inline bool is_grid_env(int e) { return e == NSG_ENV_FROZENLAKE || e == NSG_ENV_CLIFFWALKING || e == NSG_ENV_BRIDGE; }

__device__ __forceinline__ const double* grid_initial(const nsg_config& cfg, int p) {
  return (cfg.env_type == NSG_ENV_BRIDGE && cfg.params[p].theta_slot == 2) ? cfg.initial_prob[1] : cfg.initial_prob[0];
}
__device__ __forceinline__ int grid_start_state(const nsg_config& cfg, const uint8_t* desc) {
  if (cfg.env_type == NSG_ENV_CLIFFWALKING) return (cfg.nrow - 1) * cfg.ncol;  // start_state_index = (3, 0)
  if (cfg.env_type == NSG_ENV_BRIDGE) return 2 * cfg.ncol + 4;                  // envs/Bridge.py:110
  const int nS = cfg.nrow * cfg.ncol;
  for (int k = 0; k < nS; k++)
    if (desc[k] == 'S') return k;
  return 0;
}

// Per-lane copy of a grid env's persistent rows for fused rollouts (the counterpart of LaneState): cell, t,
// status, episode return, the env PCG64 stream and - FrozenLake / CliffWalking - the probabilities baked into the
// wrapper's P table stay in registers between the K steps of a launch.  (θ rows and Bridge's per-side
// distributions are touched on fires / read through the cache: they stay in memory.)
template <int ND> struct GridLane {
  int cell = 0, t = 0;
  unsigned st = 0;
  float er = 0.f;
  Pcg g = {0, 0, 0, 0};
  double tp[ND] = {};
  // fused policy rollouts only (IoMode::act_lane): next action; last step's float64 reward and terminated | truncated << 1 | taken << 2
  int ai = 0;
  double rw = 0.0;
  unsigned fl = 0;
};

template <int ENV, bool FULL>
__device__ __forceinline__ void step_grid(const nsg_config& cfg, const nsg_buffers& b, const int64_t N, const Tables& tb, const ZigLds& zg, const void* actions,
                                          const StepOut& out, int64_t i, bool active, WaveCounts& wc,
                                          GridLane<ENV == NSG_ENV_CLIFFWALKING ? 4 : 3>& gl, const IoMode io) {
  constexpr int ND = ENV == NSG_ENV_CLIFFWALKING ? 4 : 3;
  const int P = cfg.n_params;
  const bool persistent = (cfg.flags & NSG_F_PERSISTENT_PARAMS) != 0;
  const bool sim = (cfg.flags & NSG_F_SIM_ENV) != 0;
  const bool theta_live = !(sim && !(cfg.flags & NSG_F_IN_SIM_CHANGE));  // toy_text.py:170-176,354-360,636-645
  const uint8_t* desc = tb.base + cfg.desc_tab_off;

  const uint32_t o1 = (uint32_t)i, o4 = o1 * 4u, o8 = o1 * 8u;  // per-lane byte offsets into [N] rows
  const bool track = (cfg.flags & NSG_F_TRACK_RETURNS) != 0;
  // FrozenLake with its default rewards and Bridge pay only on the step that ends the episode (toy_text.py:441,
  // envs/Bridge.py:99-101): the episode return IS that last reward, no running-return row is needed
  const bool return_is_last_reward = ENV == NSG_ENV_BRIDGE || (ENV == NSG_ENV_FROZENLAKE && !(cfg.flags & NSG_F_MODIFIED_REWARDS));
  if (io.load && active) {  // first step of a launch: fetch the persistent rows
    gl.st = ldg(b.status, o1);
    gl.t = ldg(b.t, o4);
    gl.cell = ldg(b.cell, o4);
    if (track && !return_is_last_reward) gl.er = ldg(b.ep_return, o4);
    pcg_load<true>(b.rng_env, N, i, gl.g);
    if constexpr (ENV != NSG_ENV_BRIDGE) {
      // the P table of most envs is a distribution the config already holds (NSG_ST_TABLE_*): only the lanes whose status byte
      // names none fetch their table_prob rows (C3: 24 of 114 B per env-step)
      const unsigned hint = gl.st >> NSG_ST_TABLE_SHIFT;
      const nsg_param_cfg& p0 = cfg.params[0];
      const bool listed = hint >= NSG_ST_TABLE_LIST0 && (p0.upd_kind == NSG_UPD_D_STEPWISE || p0.upd_kind == NSG_UPD_D_CYCLIC) &&
                          (int)(hint - NSG_ST_TABLE_LIST0) < p0.val_tab_len;   // (a byte that names no entry of THIS config: the rows)
      if (hint != NSG_ST_TABLE_INITIAL && !listed) {
#pragma unroll
        for (int k = 0; k < ND; k++) gl.tp[k] = ldg(b.table_prob, blk_off8(ND, k, i));
      } else if (hint == NSG_ST_TABLE_INITIAL) {
#pragma unroll
        for (int k = 0; k < ND; k++) gl.tp[k] = cfg.initial_prob[0][k];
      } else {
        const double* v = tb.vals(p0.val_tab_off) + ND * (int)(hint - NSG_ST_TABLE_LIST0);
#pragma unroll
        for (int k = 0; k < ND; k++) gl.tp[k] = v[k];
      }
    }
  }
  const unsigned st = active ? gl.st : 0u;
  unsigned table_hint = st >> NSG_ST_TABLE_SHIFT;
  const int t = active ? gl.t : 0;
  // NSG_F_NO_AUTORESET: a finished env keeps stepping like gymnasium's (FrozenLake: the terminal cell's one-entry self-loop row,
  // which still consumes its draw - toy_text.py:435-436; CliffWalking / Bridge: an ordinary move from the terminal cell)
  const bool do_reset = active && (st & NSG_ST_NEEDS_RESET) && !(cfg.flags & NSG_F_NO_AUTORESET);
  const bool do_step = active && !do_reset;

  int cell = do_step ? gl.cell : 0;
  const int a = !do_step ? 0 : io.act_lane ? gl.ai : ldg_in(io.coh, (const int32_t*)actions, o4);
  // one uniform per step from the env stream (categorical_sample / np.random.choice); FrozenLakeEnv
  // and CliffWalkingEnv.reset also draw one (categorical_sample over the one-hot start distribution),
  // Bridge.reset draws nothing (envs/Bridge.py:103-111)
  double r = 0.0;
  if (do_step || (do_reset && ENV != NSG_ENV_BRIDGE)) r = pcg_double(gl.g);

  // ---- θ: every distribution parameter (toy_text.py:178-185, 362-366, 605-631) ---------------
  unsigned n_fired = 0, n_overrun = 0, n_exhausted = 0;
  double pt[ND];  // the probabilities the transition samples from
#pragma unroll
  for (int k = 0; k < ND; k++) pt[k] = cfg.initial_prob[0][k];
  bool have_table = false;
  int want = 0;  // Bridge: which param's distribution applies to this cell (0 = P, 1 = P_left, 2 = P_right)
  if constexpr (ENV == NSG_ENV_BRIDGE) {
    bool split = false;
    for (int p = 0; p < P; p++) split |= cfg.params[p].theta_slot != 0;
    if (split) {
      const int col = cell % cfg.ncol;
      want = col < cfg.ncol / 2 ? 1 : 2;  // get_loc_based_prob, envs/Bridge.py:149-158
      const double* ini = want == 2 ? cfg.initial_prob[1] : cfg.initial_prob[0];
#pragma unroll
      for (int k = 0; k < ND; k++) pt[k] = ini[k];  // an omitted side stays at its initial value
    }
  }
  for (int p = 0; p < P; p++) {
    const nsg_param_cfg& pc = cfg.params[p];
    const bool fired = fire_param<FULL>(cfg, b, tb, zg, N, i, p, t, do_step && theta_live, do_reset && !persistent, n_overrun);
    double delta = 0.0;
    double q[ND];
    bool have_q = false;
    if (fired) {
      double pp[ND];
#pragma unroll
      for (int k = 0; k < ND; k++) pp[k] = ldg(b.theta + (int64_t)(p * ND + k) * N, o8);
      int cursor = 0;
      const bool has_cur = upd_uses_cursor(pc.upd_kind);
      if (has_cur) cursor = ldg(b.cursor + (int64_t)pc.fn_slot * N, o4);
      Pcg ur = {0, 0, 0, 0};
      if (FULL && pc.uses_rng) pcg_load(b.rng_upd + (int64_t)pc.fn_slot * 4 * N, N, i, ur);
      bool exhausted;
      const int list_entry = cursor;
      upd_dist<ND, FULL>(pc, tb, zg, pp, t, cursor, ur, q, exhausted);
      if constexpr (ENV != NSG_ENV_BRIDGE)  // q goes into the P table below: name it in the status byte when it is list entry `list_entry`
        table_hint = (p == 0 && (pc.upd_kind == NSG_UPD_D_STEPWISE || pc.upd_kind == NSG_UPD_D_CYCLIC) && list_entry < pc.val_tab_len &&
                      list_entry <= (int)(NSG_ST_TABLE_MAX - NSG_ST_TABLE_LIST0))
                         ? NSG_ST_TABLE_LIST0 + (unsigned)list_entry : NSG_ST_TABLE_ROWS;
      n_exhausted |= exhausted ? 1u : 0u;
      if (FULL && pc.uses_rng) pcg_store_state(b.rng_upd + (int64_t)pc.fn_slot * 4 * N, N, i, ur);
      if (has_cur && pc.upd_kind != NSG_UPD_D_LCBOUNDED) stg_p(io.wt, b.cursor + (int64_t)pc.fn_slot * N, o4, cursor);
      delta = w1_n<ND>(pp, q);  // base.py:192-203
#pragma unroll
      for (int k = 0; k < ND; k++) {
        stg_p(io.wt, b.theta + (int64_t)(p * ND + k) * N, o8, q[k]);
        if constexpr (ENV != NSG_ENV_BRIDGE) {  // P re-weighted on a fire only
          stg_p(io.wt, b.table_prob, blk_off8(ND, k, i), q[k]);
          gl.tp[k] = q[k];
        }
      }
      have_q = true;
    }
    if constexpr (ENV == NSG_ENV_BRIDGE) {
      if (do_step && pc.theta_slot == want) {  // Bridge reads the live attribute every step (toy_text.py:626-630)
#pragma unroll
        for (int k = 0; k < ND; k++) pt[k] = have_q ? q[k] : ldg(b.theta + (int64_t)(p * ND + k) * N, o8);
      }
    } else {
      if (have_q) {
#pragma unroll
        for (int k = 0; k < ND; k++) pt[k] = q[k];
        have_table = true;
      }
    }
    if (pc.upd_kind == NSG_UPD_D_LCBOUNDED && do_step && theta_live)
      stg_p(io.wt, b.cursor + (int64_t)pc.fn_slot * N, o4, t + 1);  // UpdateFn.__call__ records prev_time = t, fired or not (base.py:143-148)
    if (do_reset && !persistent) {  // toy_text.py:206-209, 394-399, 657-666 (the P TABLE is not restored)
      const double* ini = grid_initial(cfg, p);
#pragma unroll
      for (int k = 0; k < ND; k++) stg_p(io.wt, b.theta + (int64_t)(p * ND + k) * N, o8, ini[k]);
      if (upd_uses_cursor(pc.upd_kind)) stg_p(io.wt, b.cursor + (int64_t)pc.fn_slot * N, o4, 0);
      if (FULL && pc.upd_kind == NSG_UPD_D_LCBOUNDED && pc.uses_rng) {  // inner sampler rewound with the deepcopy
        Pcg r;
        if (pc.has_fn_seed) pcg_seed(r, pc.fn_seed, -1);
        else pcg_seed(r, (uint64_t)i, 1000 + pc.fn_slot);
        pcg_store_all(b.rng_upd + (int64_t)pc.fn_slot * 4 * N, N, i, r);
      }
    }
    if (active) {
      if (!io.opt_out || out.env_change) stg_o(io.coh, out.env_change + (int64_t)p * N, o1, (uint8_t)(fired ? 1 : 0));
      if (!io.opt_out || out.delta_change) stg_o(io.coh, out.delta_change + (int64_t)p * N, o4, (float)delta);
    }
    n_fired += fired ? 1u : 0u;
  }
  if constexpr (ENV != NSG_ENV_BRIDGE) {
    if (do_step && !have_table) {
#pragma unroll
      for (int k = 0; k < ND; k++) pt[k] = gl.tp[k];
    }
  }

  // ---- transition ---------------------------------------------------------------------------
  double reward = 0.0, prob = 1.0;
  bool term = false, trunc = false;
  int tnew = 0;
  if (do_step) {
    int row = cell / cfg.ncol, col = cell - row * cfg.ncol;
    if constexpr (ENV == NSG_ENV_FROZENLAKE) {  // gymnasium FrozenLakeEnv.step over the wrapper's P (toy_text.py:426-469)
      const int letter = desc[cell];
      if (letter == 'G' || letter == 'H') {  // single self-loop entry (1.0, s, 0, True), :435-436
        prob = 1.0; reward = 0.0; term = true;
      } else {
        const double c0 = pt[0], c1 = c0 + pt[1], c2 = c1 + pt[2];          // np.cumsum
        const int idx = c0 > r ? 0 : c1 > r ? 1 : c2 > r ? 2 : 0;          // np.argmax(cs > r); all-False -> 0
        const int dir = idx == 0 ? a : idx == 1 ? ((a + 1) & 3) : ((a + 3) & 3);  // [a, a+1, a-1], :438
        if (dir == 0) col = col - 1 > 0 ? col - 1 : 0;                      // inc(), :449-458
        else if (dir == 1) row = row + 1 < cfg.nrow - 1 ? row + 1 : cfg.nrow - 1;
        else if (dir == 2) col = col + 1 < cfg.ncol - 1 ? col + 1 : cfg.ncol - 1;
        else row = row - 1 > 0 ? row - 1 : 0;
        cell = row * cfg.ncol + col;
        const int nl = desc[cell];
        term = nl == 'G' || nl == 'H';
        if (cfg.flags & NSG_F_MODIFIED_REWARDS) reward = cfg.letter_reward[nl == 'S' ? 0 : nl == 'F' ? 1 : nl == 'H' ? 2 : 3];
        else reward = nl == 'G' ? 1.0 : 0.0;
        prob = idx == 0 ? pt[0] : idx == 1 ? pt[1] : pt[2];
      }
    } else if constexpr (ENV == NSG_ENV_CLIFFWALKING) {  // CliffWalkingEnv.step over the NS table (toy_text.py:86-148)
      const double c0 = pt[0], c1 = c0 + pt[1], c2 = c1 + pt[2], c3 = c2 + pt[3];
      const int idx = c0 > r ? 0 : c1 > r ? 1 : c2 > r ? 2 : c3 > r ? 3 : 0;
      const int off = idx == 0 ? 0 : idx == 1 ? 1 : idx == 2 ? 3 : 2;      // b_actions = [a, a+1, a-1, a+2], :96
      const int dir = (a + off) & 3;                                      // UP RIGHT DOWN LEFT, :74-76
      int nr = row + (dir == 0 ? -1 : dir == 2 ? 1 : 0), nc = col + (dir == 1 ? 1 : dir == 3 ? -1 : 0);
      nr = nr < 0 ? 0 : nr > cfg.nrow - 1 ? cfg.nrow - 1 : nr;
      nc = nc < 0 ? 0 : nc > cfg.ncol - 1 ? cfg.ncol - 1 : nc;
      const bool cliff = nr == cfg.nrow - 1 && nc >= 1 && nc <= cfg.ncol - 2;
      const bool goal = nr == cfg.nrow - 1 && nc == cfg.ncol - 1;
      reward = cliff ? cfg.letter_reward[2] : goal ? cfg.letter_reward[3] : cfg.letter_reward[1];  // :117-125
      term = cliff ? ((cfg.flags & NSG_F_TERMINAL_CLIFF) != 0) : goal;                              // :126-128
      cell = cliff ? (cfg.nrow - 1) * cfg.ncol : nr * cfg.ncol + nc;                                 // :129
      prob = idx == 0 ? pt[0] : idx == 1 ? pt[1] : idx == 2 ? pt[2] : pt[3];
    } else {  // Bridge.step (envs/Bridge.py:89-134): np.random.choice([a, a+1, a-1], p=P)
      // choice(p=): cdf = cumsum(p); cdf /= cdf[-1]; idx = searchsorted(cdf, u, side="right")
      const double c0 = pt[0], c1 = c0 + pt[1], c2 = c1 + pt[2];
      int idx = (c0 / c2 <= r ? 1 : 0) + (c1 / c2 <= r ? 1 : 0) + (c2 / c2 <= r ? 1 : 0);
      idx = idx > 2 ? 2 : idx;
      const int dir = idx == 0 ? a : idx == 1 ? ((a + 1) & 3) : ((a + 3) & 3);  // LEFT DOWN RIGHT UP, :13-16
      int nr = row + (dir == 1 ? 1 : dir == 3 ? -1 : 0), nc = col + (dir == 0 ? -1 : dir == 2 ? 1 : 0);
      if (nr < 0 || nr >= cfg.nrow || nc < 0 || nc >= cfg.ncol) { nr = row; nc = col; }  // out of bounds: stay, :127-128
      cell = nr * cfg.ncol + nc;
      const int nl = desc[cell];
      if (nl == 'H') { reward = -1.0; term = true; } else if (nl == 'G') { reward = 1.0; term = true; }
      prob = pt[0];
    }
    tnew = t + 1;
    trunc = cfg.max_episode_steps > 0 && (tnew - (sim ? ldg(b.t_fork, o4) : 0)) >= cfg.max_episode_steps;
  } else if (do_reset) {
    if (sim) stg_p(io.wt, b.t_fork, o4, 0);
    cell = grid_start_state(cfg, desc);
  }
  const bool done = term || trunc;
  if (active) {
    gl.cell = cell;
    gl.t = tnew;
    gl.st = (done ? NSG_ST_NEEDS_RESET : 0u) | (table_hint << NSG_ST_TABLE_SHIFT);
    if (io.act_lane) {
      gl.rw = reward;
      gl.fl = (term ? 1u : 0u) | (trunc ? 2u : 0u) | (do_step ? 4u : 0u);
    }
    if (out.obs) stg_o(io.coh, (int32_t*)out.obs, o4, cell);  // trajectory slice (rollout); NULL for nsg_step: cell[] is the obs
    if (!io.opt_out || out.reward) stg_o(io.coh, out.reward, o4, (float)reward);
    if (!io.opt_out || out.terminated) stg_o(io.coh, out.terminated, o1, (uint8_t)(term ? 1 : 0));
    if (!io.opt_out || out.truncated) stg_o(io.coh, out.truncated, o1, (uint8_t)(trunc ? 1 : 0));
    if (track) {
      float er = return_is_last_reward ? (float)reward : do_reset ? 0.f : gl.er + (float)reward;
      if (done) {
        stg_p(io.wt, b.last_return, o4, er);
        stg_p(io.wt, b.last_length, o4, tnew);
        er = 0.f;
      }
      gl.er = er;
    }
    if (io.store) {  // last step of a launch: the persistent rows go back to memory
      stg_p(io.wt, b.cell, o4, (int32_t)cell);
      stg_p(io.wt, b.t, o4, (int32_t)tnew);
      stg_p(io.wt, b.status, o1, (uint8_t)gl.st);
      if (b.prob) stg_p(io.wt, b.prob, o4, (float)prob);
      if (track && !return_is_last_reward) stg_p(io.wt, b.ep_return, o4, gl.er);
      if (io.wt) pcg_store_state<true, true>(b.rng_env, N, i, gl.g); else pcg_store_state<true>(b.rng_env, N, i, gl.g);
    }
  }
  const unsigned long long done_mask = __ballot(done);
  if (b.done_bits && (threadIdx.x & 63) == 0 && i < N) b.done_bits[i >> 6] = done_mask;
  wc.done += __popcll(done_mask);
  wc.steps += __popcll(__ballot(do_step));
  for (unsigned bit = 0; bit < 2; bit++) wc.fired += __popcll(__ballot((n_fired >> bit) & 1u)) << bit;
  wc.sched_overrun += __popcll(__ballot(n_overrun != 0));
  wc.lc_exhausted += __popcll(__ballot(n_exhausted != 0));
}

// One chunk of kBlock envs of any env type (block-level call: contains workgroup barriers).
template <int ENV, bool FULL>
__device__ __forceinline__ void step_block(const nsg_config& cfg, const nsg_buffers& b, const int64_t N, const Tables& tb, const ZigLds& zg, const void* actions,
                                           const StepOut& out, int64_t base, int parity, LdsTables& lds, WaveCounts& wc) {
  // single step: every persistent row round-trips through memory
  if constexpr (ENV == NSG_ENV_FROZENLAKE || ENV == NSG_ENV_CLIFFWALKING || ENV == NSG_ENV_BRIDGE) {
    const int64_t i = base + threadIdx.x;
    GridLane<ENV == NSG_ENV_CLIFFWALKING ? 4 : 3> gl;
    step_grid<ENV, FULL>(cfg, b, N, tb, zg, actions, out, i, i < N, wc, gl, IoMode{true, true, false, false, true});
  } else {
    LaneState<ENV> ls;
    step_chunk<ENV, FULL>(cfg, b, N, tb, zg, actions, out, base, parity, lds, wc, ls, IoMode{true, true, false, false, true});
  }
}

// Homogeneous launch: grid-stride over 256-env chunks.  `cfg` is the segment's own copy (generic
// kernels, scalar loads) or a compile-time constant of a config-specialised build (nsg_spec.hip.h).
template <int ENV, bool FULL>
__device__ __forceinline__ void step_body(const nsg_config& cfg, const Segment& sg, const void* __restrict__ actions,
                                          const int block_rel, const int block_count, const int reverse = 0) {
  LdsTables lds;
  Tables tb;
  ZigLds zg;
  const int64_t N = sg.N;   // asked for before the staging, so that it travels with the first batch of scalar loads
  constexpr bool kGrid = ENV == NSG_ENV_FROZENLAKE || ENV == NSG_ENV_CLIFFWALKING || ENV == NSG_ENV_BRIDGE;
  if constexpr (NSG_TABLES_DIRECT != 0 && !kGrid) stage_tables<true, !EnvTraits<kGrid ? NSG_ENV_CARTPOLE : ENV>::RESET_IN_LANE, 1>(sg, lds, tb, zg);
  else stage_tables<false, true, kGrid ? 2 : 1>(sg, lds, tb, zg);
  WaveCounts wc;
  const nsg_buffers& b = sg.buf;
  const StepOut out = default_out(b);
  const int64_t chunks = (N + kBlock - 1) / kBlock;
  int parity = 0;
  // XCD-aware traversal.  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2 and translation caches, so chunk c is
  // always stepped by XCD c mod 8 - and what a launch touched last is what that XCD has warmest when the next launch starts.
  // Every other launch (`reverse`, nsg_step alternates it per handle) therefore walks the groups of 8 chunks back to front,
  // each chunk STAYING ON ITS XCD: the state rows it reads first are the ones the previous launch wrote last.  Measured at
  // 2^20 envs, same box, three interleaved repetitions: C1 26.4 -> 25.0 us, C2 35.5 -> 33.4, C3 22.4 -> 20.1, Pendulum
  // 20.8 -> 20.1, Acrobot 61.5 -> 59.5 (generic kernels alike).  The XCD is what matters: a plain reversal (chunk c ->
  // chunks-1-c, which moves every chunk to another XCD) is SLOWER than no reversal at all - C1 27.3 vs 26.6 us, C3 22.5 vs
  // 22.4 - so the effect is local to the XCD, not the memory-side Infinity Cache (L2 hit / miss counters do not move: DESIGN.md section 4).  A ragged tail of fewer than 8 chunks keeps its
  // place.  Results do not depend on the order.
  const int64_t groups = chunks >> 3;
  auto chunk_of = [&](int64_t c) { return (reverse && (c >> 3) < groups) ? ((groups - 1 - (c >> 3)) << 3) + (c & 7) : c; };
  for (int64_t c = block_rel; c < chunks; c += block_count, parity ^= 1)
    step_block<ENV, FULL>(cfg, b, N, tb, zg, actions, out, chunk_of(c) * kBlock, parity, lds, wc);
  flush_counts(b.counters, block_rel, wc);
}

// The precompiled full-engine step kernels sit just above the 128-VGPR line (130 for CartPole and Pendulum with the two-level sincos
// reduction; 3 wavefronts per SIMD instead of 4: C2 on the generic kernels 44.5 -> 53.4 us at 2^20 envs): they are held at 4 wavefronts
// per SIMD, which the compiler meets without spilling (csrc/resource_usage.txt, tests/test_generic_kernel_resources.py).  Acrobot's
// full engine (154 VGPRs) cannot and keeps its natural count.
template <int ENV, bool FULL> constexpr int kStepMinWaves = (FULL && ENV != NSG_ENV_ACROBOT && NSG_MIN_WAVES < 4) ? 4 : NSG_MIN_WAVES;

template <int ENV, bool FULL>
__global__ __launch_bounds__(kBlock, (kStepMinWaves<ENV, FULL>)) void step_kernel(const Segment* __restrict__ seg, const void* __restrict__ actions, int reverse) {
  step_body<ENV, FULL>(seg->cfg, *seg, actions, (int)blockIdx.x, (int)gridDim.x, reverse);
}

// Heterogeneous launch: block ranges are assigned to env-type segments, so the env-type switch
// is uniform per workgroup (no intra-wave divergence between Pendulum and Acrobot lanes).
// The segment table of a heterogeneous launch reaches the kernel as a `const … __restrict__` kernel argument: that is what
// lets the compiler read the members' configs and row pointers through SCALAR loads.  (Round 2 first passed per-member
// segment pointers inside a by-value struct: `*args.seg[sidx]` has unknown provenance, every config access became a vector
// load, and C4's launch went from 24.0 to 28-29 us.)
__device__ __forceinline__ int group_segment_of_block(const Segment* __restrict__ segs, int nseg) {
  int sidx = 0;  // block ranges are disjoint but not ordered by member index (the host places long-running env types first)
  for (int k = 1; k < nseg; k++)
    if ((int)blockIdx.x >= segs[k].block_begin && (int)blockIdx.x < segs[k].block_begin + segs[k].block_count) sidx = k;
  return sidx;
}

template <bool FULL>
__global__ __launch_bounds__(kBlock) void step_group_kernel(const Segment* __restrict__ segs, int nseg, ActionPtrs acts, int reverse) {
  const int sidx = group_segment_of_block(segs, nseg);
  const Segment& sg = segs[sidx];
  const void* actions = acts.p[sidx];
  const int rel = (int)blockIdx.x - sg.block_begin;
  const int cnt = sg.block_count;
  switch (sg.cfg.env_type) {
    case NSG_ENV_CARTPOLE: step_body<NSG_ENV_CARTPOLE, FULL>(sg.cfg, sg, actions, rel, cnt, reverse); break;
    case NSG_ENV_PENDULUM: step_body<NSG_ENV_PENDULUM, FULL>(sg.cfg, sg, actions, rel, cnt, reverse); break;
    case NSG_ENV_ACROBOT: step_body<NSG_ENV_ACROBOT, FULL>(sg.cfg, sg, actions, rel, cnt, reverse); break;
    case NSG_ENV_MOUNTAINCAR: step_body<NSG_ENV_MOUNTAINCAR, FULL>(sg.cfg, sg, actions, rel, cnt, reverse); break;
    case NSG_ENV_MOUNTAINCAR_CONT: step_body<NSG_ENV_MOUNTAINCAR_CONT, FULL>(sg.cfg, sg, actions, rel, cnt, reverse); break;
    case NSG_ENV_FROZENLAKE: step_body<NSG_ENV_FROZENLAKE, FULL>(sg.cfg, sg, actions, rel, cnt, reverse); break;
    case NSG_ENV_CLIFFWALKING: step_body<NSG_ENV_CLIFFWALKING, FULL>(sg.cfg, sg, actions, rel, cnt, reverse); break;
    default: step_body<NSG_ENV_BRIDGE, FULL>(sg.cfg, sg, actions, rel, cnt, reverse); break;
  }
}

// ============================================================================================
// reset(seed) / reset(): NSWrapper.reset + subclass tails (base.py:365-431,
// classic_control.py:102-109, toy_text.py:382-399).
// ============================================================================================
// `restart`: reset(seed = base_seed + i) for every env (nsg_reset_seeded: the host has installed the affine descriptor), the
// update fns' streams included (SeedSequence(seed).spawn, base.py:412-421).
// With `seeds` the host has put the classic-control streams into their per-env form first (materialize_streams_kernel).
template <int ENV>
__global__ __launch_bounds__(kBlock) void reset_kernel(const Segment* __restrict__ seg, const uint64_t* __restrict__ seeds,
                                                       const uint8_t* __restrict__ mask, int restart, uint64_t base_seed) {
  const Segment& sg = *seg;
  const nsg_config& cfg = sg.cfg;
  const nsg_buffers& b = sg.buf;
  const int64_t N = sg.N;
  const int P = cfg.n_params;
  const bool persistent = (cfg.flags & NSG_F_PERSISTENT_PARAMS) != 0;
  // not a hot path: the ziggurat tables are read from their global copy
  const ZigLds zg = {sg.zig, (const double*)(sg.zig + 256), (const double*)(sg.zig + 512),
                     sg.zig + 768, (const double*)(sg.zig + 1024), (const double*)(sg.zig + 1280), sg.jump};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    if (mask && !mask[i]) continue;
    Pcg g;
    constexpr bool GRID = ENV == NSG_ENV_FROZENLAKE || ENV == NSG_ENV_CLIFFWALKING || ENV == NSG_ENV_BRIDGE;
    [[maybe_unused]] uint32_t count = 0;   // classic envs: resets drawn from the env's stream so far
    if constexpr (GRID) {
      if (seeds) pcg_seed(g, seeds[i], -1);  // gymnasium Env.reset(seed) -> np_random(seed) [UPSTREAM]
      else pcg_load<true>(b.rng_env, N, i, g);
    } else {
      if (seeds) {   // this env's own (seed, no spawn key) record; its stream starts over
        b.rng_env[2 * (i + 1)] = seeds[i];
        b.rng_env[2 * (i + 1) + 1] = (uint64_t)(uint32_t)-1;
      } else if (!restart) {
        count = (uint32_t)b.episode[i] >> NSG_EP_COUNT_SHIFT;   // reset(seed=None): the stream continues
      }
      if (seeds) pcg_seed(g, seeds[i], -1);
      else env_stream_at(b.rng_env, i, (uint64_t)count * EnvTraits<GRID ? NSG_ENV_CARTPOLE : ENV>::RESET_DRAWS, sg.jump, g);
    }
    if constexpr (ENV == NSG_ENV_FROZENLAKE || ENV == NSG_ENV_CLIFFWALKING || ENV == NSG_ENV_BRIDGE) {
      // FrozenLakeEnv / CliffWalkingEnv.reset consume one random() (categorical_sample over the one-hot
      // start distribution); Bridge.reset draws nothing
      if constexpr (ENV != NSG_ENV_BRIDGE) (void)pcg_double(g);
      b.cell[i] = grid_start_state(cfg, sg.tables + cfg.desc_tab_off);
      if (b.prob) b.prob[i] = 1.0f;
    } else {
      using T = EnvTraits<ENV>;
      double s[T::PHYS];
      env_reset_draw<ENV>(g, s);
#pragma unroll
      for (int k = 0; k < T::PHYS; k++) b.phys[blk_off8(T::PHYS, k, i) / 8] = s[k];
      float o[T::OBS];
      env_obs<ENV>(s, o);
      store_obs<ENV>(b.obs, i, o);
    }
    if constexpr (GRID) pcg_store_all<true>(b.rng_env, N, i, g);
    else b.episode[i] = (int32_t)((count + 1u) << NSG_EP_COUNT_SHIFT);   // one more episode drawn; needs-reset cleared
    b.t[i] = 0;
    if (b.t_fork) b.t_fork[i] = 0;
    for (int p = 0; p < P; p++) {
      const nsg_param_cfg& pc = cfg.params[p];
      if (!persistent) {
        if constexpr (ENV == NSG_ENV_FROZENLAKE || ENV == NSG_ENV_CLIFFWALKING || ENV == NSG_ENV_BRIDGE) {
          constexpr int ND = ENV == NSG_ENV_CLIFFWALKING ? 4 : 3;
          const double* ini = grid_initial(cfg, p);
#pragma unroll
          for (int k = 0; k < ND; k++) b.theta[(int64_t)(p * ND + k) * N + i] = ini[k];
        } else {
          b.theta[(int64_t)p * N + i] = cfg.base_theta[pc.theta_slot];
        }
        if (b.cursor) b.cursor[(int64_t)p * N + i] = 0;
        if (sched_is_stochastic(pc.sched_kind)) {  // rewound with the rest of init_initial_params
          Pcg sr;
          int nx;
          sched_construct(pc, zg, p, i, sr, nx);
          pcg_store_all(b.rng_sched + (int64_t)p * 4 * N, N, i, sr);
          b.sched_next[(int64_t)p * N + i] = nx;
        }
      }
      if (pc.upd_kind == NSG_UPD_D_LCBOUNDED) {  // inner sampler: rewound, never re-seeded by reset(seed)
        if (pc.uses_rng && !persistent) {
          Pcg r;
          if (pc.has_fn_seed) pcg_seed(r, pc.fn_seed, -1);
          else pcg_seed(r, (uint64_t)i, 1000 + p);
          pcg_store_all(b.rng_upd + (int64_t)p * 4 * N, N, i, r);
        }
      } else if (pc.uses_rng && (seeds || restart)) {  // SeedSequence(seed).spawn(P)[rng_child], base.py:412-421
        Pcg r;
        pcg_seed(r, seeds ? seeds[i] : base_seed + (uint64_t)i, pc.rng_child);
        pcg_store_all(b.rng_upd + (int64_t)p * 4 * N, N, i, r);
      }
      b.env_change[(int64_t)p * N + i] = 0;
      b.delta_change[(int64_t)p * N + i] = 0.f;
    }
    b.reward[i] = 0.f;
    b.terminated[i] = 0;
    b.truncated[i] = 0;
    if constexpr (GRID) b.status[i] &= (uint8_t)~NSG_ST_NEEDS_RESET;  // the table hint stays: reset does not restore the P table
    if (cfg.flags & NSG_F_TRACK_RETURNS) {
      if (b.ep_return) b.ep_return[i] = 0.f;
      if (b.ep_length) b.ep_length[i] = 0;
    }
    // clear THIS env's bit of the ballot word only: a masked reset must neither wipe the bits of envs that are done but were
    // not reset nor leave the bit of a reset env standing (nsg_compact_done reads these words)
    if (b.done_bits) atomicAnd((unsigned long long*)&b.done_bits[i >> 6], ~(1ULL << (i & 63)));
  }
}

#ifndef NSG_SPEC_BUILD  // a config-specialised unit carries only its step / rollout kernels
// Construction-time state (what the wrapper constructors set up): θ = construction values,
// fresh cursors, FrozenLake P table from initial_prob_dist, update-fn streams
// default_rng(fn.seed) (single_param.py:76,108,146,342,444), zeroed outputs / counters.
__global__ __launch_bounds__(kBlock) void init_kernel(const Segment* __restrict__ seg) {
  const Segment& sg = *seg;
  const nsg_config& cfg = sg.cfg;
  const nsg_buffers& b = sg.buf;
  const int64_t N = sg.N;
  const bool fl = is_grid_env(cfg.env_type);
  const int nd = cfg.env_type == NSG_ENV_CLIFFWALKING ? 4 : 3;
  const ZigLds zg = {sg.zig, (const double*)(sg.zig + 256), (const double*)(sg.zig + 512),
                     sg.zig + 768, (const double*)(sg.zig + 1024), (const double*)(sg.zig + 1280), sg.jump};
  if (blockIdx.x == 0 && b.counters)
    for (int k = threadIdx.x; k < NSG_CNT_COUNT * kCntShards; k += kBlock) b.counters[k] = 0;
  // a never-seeded classic-control batch: env i's np_random is PCG64(SeedSequence(i, spawn_key=(999,))) - a fixed stream where
  // the reference has OS entropy - in the affine form (nothing per env)
  if (!fl && blockIdx.x == 0 && threadIdx.x == 0) env_stream_set_affine(b.rng_env, 0, 999);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    if (fl) {
      for (int p = 0; p < cfg.n_params; p++)
        for (int k = 0; k < nd; k++) b.theta[(int64_t)(p * nd + k) * N + i] = grid_initial(cfg, p)[k];
      if (b.table_prob)
        for (int k = 0; k < nd; k++) b.table_prob[blk_off8(nd, k, i) / 8] = cfg.initial_prob[0][k];
      b.cell[i] = 0;
      if (b.prob) b.prob[i] = 1.f;
    }
    for (int p = 0; p < cfg.n_params; p++) {
      const nsg_param_cfg& pc = cfg.params[p];
      if (!fl) b.theta[(int64_t)p * N + i] = cfg.base_theta[pc.theta_slot];
      if (b.cursor) b.cursor[(int64_t)p * N + i] = 0;
      if (sched_is_stochastic(pc.sched_kind)) {
        Pcg sr;
        int nx;
        sched_construct(pc, zg, p, i, sr, nx);
        pcg_store_all(b.rng_sched + (int64_t)p * 4 * N, N, i, sr);
        b.sched_next[(int64_t)p * N + i] = nx;
      }
      if (pc.uses_rng) {
        Pcg r;
        if (pc.has_fn_seed) pcg_seed(r, pc.fn_seed, -1);
        else pcg_seed(r, (uint64_t)i, 1000 + p);  // reference: OS entropy; here a fixed per-env stream
        pcg_store_all(b.rng_upd + (int64_t)p * 4 * N, N, i, r);
      }
      b.env_change[(int64_t)p * N + i] = 0;
      b.delta_change[(int64_t)p * N + i] = 0.f;
    }
    if (fl) {
      Pcg g;
      pcg_seed(g, (uint64_t)i, 999);
      pcg_store_all<true>(b.rng_env, N, i, g);
      b.status[i] = (uint8_t)((b.table_prob ? NSG_ST_TABLE_INITIAL : NSG_ST_TABLE_ROWS) << NSG_ST_TABLE_SHIFT);
    } else {
      b.episode[i] = 0;
    }
    b.t[i] = 0;
    b.reward[i] = 0.f;
    b.terminated[i] = 0;
    b.truncated[i] = 0;
    if (cfg.flags & NSG_F_TRACK_RETURNS) {
      if (b.ep_return) b.ep_return[i] = 0.f;
      if (b.ep_length) b.ep_length[i] = 0;
      if (b.last_return) b.last_return[i] = 0.f;
      b.last_length[i] = 0;
    }
    if (b.done_bits && (i & 63) == 0) b.done_bits[i >> 6] = 0;
  }
}

// ============================================================================================
// done-mask compaction: ballot words -> dense list of env indices (wave prefix via mbcnt).
// ============================================================================================
__global__ __launch_bounds__(kBlock) void compact_done_kernel(const uint64_t* __restrict__ done_bits, int64_t N,
                                                              int32_t* __restrict__ out_idx,
                                                              unsigned long long* __restrict__ out_count) {
  const int lane = threadIdx.x & 63;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ((N + 63) & ~63LL); i += (int64_t)gridDim.x * kBlock) {
    const unsigned long long m = done_bits[i >> 6];  // wave-uniform load
    if (m == 0) continue;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(out_count, (unsigned long long)__popcll(m));
    base = __shfl(base, 0);
    if ((m >> lane) & 1ULL) {
      const int rank = __popcll(m & ((1ULL << lane) - 1ULL));
      out_idx[base + rank] = (int32_t)i;
    }
  }
}

// ============================================================================================
// θ-engine alone and raw NumPy-compatible streams (known-answer tests through the C-ABI).
// ============================================================================================
__global__ __launch_bounds__(kBlock) void theta_trace_kernel(const Segment* __restrict__ seg, int p, int n, int t0, int T,
                                                             const double* __restrict__ theta0, uint64_t* rng_state,
                                                             double* __restrict__ theta_out, uint8_t* __restrict__ fired_out,
                                                             double* __restrict__ delta_out, nsg_trace_state ts) {
  LdsTables lds;
  const Segment& sg = *seg;
  Tables tb;
  ZigLds zg;
  stage_tables(sg, lds, tb, zg);
  const nsg_param_cfg& pc = sg.cfg.params[p];
  const bool dist = pc.upd_kind >= NSG_UPD_D_INCREMENT;
  const int nd = sg.cfg.env_type == NSG_ENV_CLIFFWALKING ? 4 : 3;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  Pcg r = {0, 0, 0, 0};
  if (pc.uses_rng && rng_state) pcg_load(rng_state, n, i, r);
  int cursor = 0, snext = 0;
  Pcg sr = {0, 0, 0, 0};
  const bool stoch = sched_is_stochastic(pc.sched_kind);
  if (ts.resume) {  // a host-side object called again: continue from ITS state
    if (ts.cursor) cursor = ts.cursor[i];
    if (stoch && ts.sched_rng) pcg_load(ts.sched_rng, n, i, sr);
    if (stoch && ts.sched_next) snext = ts.sched_next[i];
  } else if (stoch) {
    sched_construct(pc, zg, p, i, sr, snext);
  }
  double th[4] = {0, 0, 0, 0};
  if (dist) {
    for (int c = 0; c < nd; c++) th[c] = theta0[nd * i + c];
  } else {
    th[0] = theta0[i];
  }
  for (int k = 0; k < T; k++) {
    const int t = t0 + k;
    const bool fired = stoch ? sched_fire_stoch(pc, zg, t, sr, snext) : sched_fire(pc, tb, t);
    bool raised = sched_overrun(pc, t);  // the reference would have raised / needed an answer the table does not hold
    double delta = 0.0;
    if (fired) {
      if (dist) {
        double q[4] = {0, 0, 0, 0};
        if (nd == 4) { upd_dist<4, true>(pc, tb, zg, th, t, cursor, r, q, raised); delta = w1_n<4>(th, q); }
        else { upd_dist<3, true>(pc, tb, zg, th, t, cursor, r, q, raised); delta = w1_n<3>(th, q); }
        for (int c = 0; c < nd; c++) th[c] = q[c];
      } else {
        double nvv = upd_scalar<true>(pc, tb, zg, th[0], t, r, cursor);
        delta = nvv - th[0];
        th[0] = nvv;
      }
    }
    if (pc.upd_kind == NSG_UPD_D_LCBOUNDED) cursor = t + 1;
    if (dist) {
      for (int c = 0; c < nd; c++) theta_out[((int64_t)k * nd + c) * n + i] = th[c];
    } else {
      theta_out[(int64_t)k * n + i] = th[0];
    }
    fired_out[(int64_t)k * n + i] = raised ? (pc.upd_kind == NSG_UPD_D_LCBOUNDED && !sched_overrun(pc, t) ? 0xFF : 0xFE) : fired ? 1 : 0;
    delta_out[(int64_t)k * n + i] = delta;
  }
  if (pc.uses_rng && rng_state) pcg_store_all(rng_state, n, i, r);
  if (ts.cursor) ts.cursor[i] = cursor;
  if (stoch && ts.sched_rng) pcg_store_all(ts.sched_rng, n, i, sr);
  if (stoch && ts.sched_next) ts.sched_next[i] = snext;
}

__global__ __launch_bounds__(kBlock) void rng_fill_kernel(int kind, const uint64_t* __restrict__ seeds, int n, int spawn_key,
                                                          int count, void* out, uint64_t* state_out,
                                                          const uint64_t* __restrict__ zig) {
  __shared__ uint64_t lz[768];
  for (int k = threadIdx.x; k < 768; k += kBlock) lz[k] = zig[k];
  __syncthreads();
  ZigLds zg = {lz, (const double*)(lz + 256), (const double*)(lz + 512)};
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  Pcg r;
  pcg_seed(r, seeds[i], spawn_key);
  if (state_out) pcg_store_all(state_out, n, i, r);
  for (int k = 0; k < count; k++) {
    if (kind == 0) ((uint64_t*)out)[(int64_t)k * n + i] = pcg_next64(r);
    else if (kind == 1) ((double*)out)[(int64_t)k * n + i] = pcg_double(r);
    else ((double*)out)[(int64_t)k * n + i] = pcg_std_normal(r, zg);
  }
}

// ============================================================================================
// Batched planning-env snapshot (get_planning_env / __deepcopy__, classic_control.py:120-186,
// toy_text.py:471-511, base.py:433-441): dst := copy of src with is_sim_env semantics.
// ============================================================================================
__global__ __launch_bounds__(kBlock) void fork_kernel(const Segment* __restrict__ sseg, const Segment* __restrict__ dseg,
                                                      uint64_t entropy, int theta_mode) {
  const Segment& ss = *sseg;
  const Segment& ds = *dseg;
  const nsg_config& cfg = ss.cfg;
  const nsg_buffers& sb = ss.buf;
  const nsg_buffers& db = ds.buf;
  // the copy may hold SEVERAL copies of every source env (dst N = k * src N, copy j <- env j mod src N): a planner's
  // simulations of one decision as ONE batch; every copy gets its own streams (entropy + j)
  const int64_t Ns = ss.N, N = ds.N;
  const int env = cfg.env_type, P = cfg.n_params;
  const bool fl = is_grid_env(env);
  const int nd = env == NSG_ENV_CLIFFWALKING ? 4 : 3;
  const bool in_sim_change = (ds.cfg.flags & NSG_F_IN_SIM_CHANGE) != 0;
  const int phys = fl ? 0 : (env == NSG_ENV_CARTPOLE || env == NSG_ENV_ACROBOT ? 4 : 2);
  const int obs = fl ? 0 : (env == NSG_ENV_CARTPOLE ? 4 : env == NSG_ENV_PENDULUM ? 3 : env == NSG_ENV_ACROBOT ? 6 : 2);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    const int64_t is = i % Ns;
    for (int k = 0; k < phys; k++) db.phys[blk_off8(phys, k, i) / 8] = sb.phys[blk_off8(phys, k, is) / 8];
    if (fl) db.cell[i] = sb.cell[is];
    const int t = sb.t[is];
    db.t[i] = t;
    db.t_fork[i] = t;
    // a non-sim Pendulum source keeps no needs-reset bit (step_chunk: derived from t); the copy, whose TimeLimit restarts, gets it
    const bool src_from_t = env == NSG_ENV_PENDULUM && !(cfg.flags & NSG_F_SIM_ENV);
    const int src_needs_reset = fl ? 0 : src_from_t ? (cfg.max_episode_steps > 0 && t >= cfg.max_episode_steps ? 1 : 0)
                                                    : (sb.episode[is] & (int32_t)NSG_ST_NEEDS_RESET);
    // NSG_F_NO_AUTORESET: the copy is a NEW base env that was reset (classic_control.py:168-178): it has not terminated yet
    const bool noauto = (ds.cfg.flags & NSG_F_NO_AUTORESET) != 0;
    if (fl) db.status[i] = noauto ? (uint8_t)(sb.status[is] & ~NSG_ST_NEEDS_RESET) : sb.status[is];  // (its table hint is settled with the table, below)
    else db.episode[i] = noauto ? 0 : src_needs_reset;   // the copy's own stream starts at its first episode
    for (int r = 0; r < (fl ? nd * P : P); r++) {
      const double cur = sb.theta[(int64_t)r * Ns + is];
      const double init = fl ? grid_initial(cfg, r / nd)[r % nd] : cfg.base_theta[cfg.params[r].theta_slot];
      db.theta[(int64_t)r * N + i] = theta_mode == 1 ? init : cur;
    }
    if (env == NSG_ENV_CARTPOLE && db.derived) {  // sim_env._dependency_resolver() at copy time (:183)
      double c[6];
      for (int k = 0; k < 6; k++) c[k] = cfg.base_theta[k];
      for (int p = 0; p < P; p++) {
        const int slot = cfg.params[p].theta_slot;
        const double v = sb.theta[(int64_t)p * Ns + is];
        for (int k = 0; k < 6; k++)
          if (k == slot) c[k] = v;
      }
      db.derived[i] = c[2] + c[1];
      db.derived[N + i] = c[5] * c[2];
    }
    if (fl) {
      if (db.table_prob) {
        // which P table the copy steps with.  The reference wrappers hold TWO tables - the wrapper's own `self.P` and the base env's
        // `unwrapped.P`; a frozen copy steps with the latter and never touches either, an in_sim_change copy (like a real env)
        // re-installs its own before every step.  buffers.table_prob is the one the env steps with.
        // FrozenLake (toy_text.py:479-480,505-508,365-367): a copy's OWN table is built from initial_prob_dist by its constructor and
        // its base env gets the SOURCE's own table - so an in_sim_change copy steps with the initial table, and so does a copy OF A
        // COPY (the source's own table is that initial one; MCTS.search deep-copies the planning env it was given, MCTS.py:131:
        // its simulations run on the initial distribution whatever the slipperiness has become).
        // CliffWalking (toy_text.py:219-221,246-249,187): a copy's own table AND its base env's are the source's own table; only
        // get_planning_env() without delta notification then overwrites the base env's with the initial one - which a copy of THAT
        // copy does not inherit (it takes the source's own, still current, table).  A frozen CliffWalking copy therefore keeps its
        // own table's probabilities in buffers.derived.
        const bool src_sim = (cfg.flags & NSG_F_SIM_ENV) != 0;
        if (env == NSG_ENV_FROZENLAKE) {
          const bool use_initial = in_sim_change || theta_mode == 1 || src_sim;
          for (int k = 0; k < nd; k++)
            db.table_prob[blk_off8(nd, k, i) / 8] = use_initial ? cfg.initial_prob[0][k] : sb.table_prob[blk_off8(nd, k, is) / 8];
          if (use_initial) db.status[i] = (uint8_t)((sb.status[is] & NSG_ST_NEEDS_RESET) | (NSG_ST_TABLE_INITIAL << NSG_ST_TABLE_SHIFT));
        } else {
          const bool own_in_derived = src_sim && !in_sim_change && sb.derived;   // the source is a frozen copy
          const bool use_initial = theta_mode == 1 && !in_sim_change;
          for (int k = 0; k < nd; k++) {
            const double own = own_in_derived ? sb.derived[(int64_t)k * Ns + is] : sb.table_prob[blk_off8(nd, k, is) / 8];
            if (db.derived) db.derived[(int64_t)k * N + i] = own;
            db.table_prob[blk_off8(nd, k, i) / 8] = use_initial ? cfg.initial_prob[0][k] : own;
          }
          if (use_initial) db.status[i] = (uint8_t)((sb.status[is] & NSG_ST_NEEDS_RESET) | (NSG_ST_TABLE_INITIAL << NSG_ST_TABLE_SHIFT));
          else if (own_in_derived) db.status[i] = (uint8_t)(sb.status[is] & NSG_ST_NEEDS_RESET);   // (the source's hint names ITS stepping table: read the rows)
        }
      }
      if (db.prob && sb.prob) db.prob[i] = sb.prob[is];
    }
    for (int p = 0; p < P; p++) {
      if (db.cursor && sb.cursor) db.cursor[(int64_t)p * N + i] = sb.cursor[(int64_t)p * Ns + is];  // deepcopy(tunable_params)
      if (sched_is_stochastic(cfg.params[p].sched_kind)) {  // scheduler state is copied, never re-seeded (base.py:433-441)
        for (int k = 0; k < 4; k++) db.rng_sched[((int64_t)p * N + i) * 4 + k] = sb.rng_sched[((int64_t)p * Ns + is) * 4 + k];
        db.sched_next[(int64_t)p * N + i] = sb.sched_next[(int64_t)p * Ns + is];
      }
      db.env_change[(int64_t)p * N + i] = sb.env_change[(int64_t)p * Ns + is];
      db.delta_change[(int64_t)p * N + i] = sb.delta_change[(int64_t)p * Ns + is];
      if (cfg.params[p].uses_rng && cfg.params[p].upd_kind == NSG_UPD_D_LCBOUNDED) {  // inner rng: deep-copied
        for (int k = 0; k < 4; k++) db.rng_upd[((int64_t)p * N + i) * 4 + k] = sb.rng_upd[((int64_t)p * Ns + is) * 4 + k];
      } else if (cfg.params[p].uses_rng) {  // _reseed_planning_env_rngs (base.py:433-441): fresh entropy
        Pcg r;
        pcg_seed(r, entropy + (uint64_t)i, 7100 + p);
        pcg_store_all(db.rng_upd + (int64_t)p * 4 * N, N, i, r);
      }
    }
    // the copy's base env is a new gym.make(): its np_random is unseeded -> stream (entropy + i, spawn key 7001)
    if (fl) {
      Pcg g;
      pcg_seed(g, entropy + (uint64_t)i, 7001);
      pcg_store_all<true>(db.rng_env, N, i, g);
    } else if (i == 0) {
      env_stream_set_affine(db.rng_env, entropy, 7001);
    }
    for (int k = 0; k < obs; k++) db.obs[i * obs + k] = sb.obs[is * obs + k];
    db.reward[i] = sb.reward[is];
    db.terminated[i] = sb.terminated[is];
    db.truncated[i] = sb.truncated[is];
    if (ds.cfg.flags & NSG_F_TRACK_RETURNS) {
      if (db.ep_return) db.ep_return[i] = ((cfg.flags & NSG_F_TRACK_RETURNS) && sb.ep_return) ? sb.ep_return[is] : 0.f;
      if (db.last_return) db.last_return[i] = 0.f;
      db.last_length[i] = 0;
    }
    const unsigned long long done_mask = __ballot(((fl ? (int)sb.status[is] : src_needs_reset) & (int)NSG_ST_NEEDS_RESET) != 0);  // == the source's ballot word when N == Ns
    if (db.done_bits && (i & 63) == 0) db.done_bits[i >> 6] = done_mask;
  }
}

// Classic-control env streams, affine -> per-env form (before a reset / re-seed with an ARBITRARY seed array): every env gets
// its own (seed, spawn key) record; the descriptor is cleared by a second, one-thread launch (stream_set_kernel) so that no
// block of this one can see it change.
__global__ __launch_bounds__(kBlock) void materialize_streams_kernel(const Segment* __restrict__ seg) {
  const Segment& sg = *seg;
  const int64_t N = sg.N;
  uint64_t* r = sg.buf.rng_env;
  const uint64_t d0 = r[0], d1 = r[1];
  if (!(d0 & NSG_STREAM_AFFINE)) return;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    r[2 * (i + 1)] = d1 + (uint64_t)i;
    r[2 * (i + 1) + 1] = d0 & 0xffffffffULL;
  }
}
__global__ void stream_set_kernel(const Segment* __restrict__ seg, uint64_t word0, uint64_t word1) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    seg->buf.rng_env[0] = word0;
    seg->buf.rng_env[1] = word1;
  }
}

// env.np_random = default_rng(seed) / fn.seed(child) without touching env state
__global__ __launch_bounds__(kBlock) void seed_streams_kernel(const Segment* __restrict__ seg, const uint64_t* __restrict__ seeds,
                                                              int which) {
  const Segment& sg = *seg;
  const int64_t N = sg.N;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    if (which == 0) {
      if (is_grid_env(sg.cfg.env_type)) {
        Pcg g;
        pcg_seed(g, seeds[i], -1);
        pcg_store_all<true>(sg.buf.rng_env, N, i, g);
      } else {   // per-env record (the host has put the streams into their per-env form); the stream starts over
        sg.buf.rng_env[2 * (i + 1)] = seeds[i];
        sg.buf.rng_env[2 * (i + 1) + 1] = (uint64_t)(uint32_t)-1;
        sg.buf.episode[i] &= (int32_t)NSG_ST_NEEDS_RESET;
      }
    } else {
      for (int p = 0; p < sg.cfg.n_params; p++)
        if (sg.cfg.params[p].uses_rng) {
          Pcg r;
          pcg_seed(r, seeds[i], sg.cfg.params[p].rng_child);
          pcg_store_all(sg.buf.rng_upd + (int64_t)p * 4 * N, N, i, r);
        }
    }
  }
}

__global__ __launch_bounds__(kBlock) void calib_copy_f64_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) dst[i] = src[i] + 1.0;
}

// Small read-back for latency-bound callers (the N = 1 adaptors): ONE workgroup copies `bytes` (a multiple of 16) from device
// memory into pinned, device-mapped HOST memory with plain stores and then publishes `seq` in the word that follows them
// (system-scope release after a workgroup barrier) - the host polls that word instead of paying for a DMA copy + event.
__global__ __launch_bounds__(kBlock) void read_back_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16, uint64_t seq) {
  for (int64_t k = threadIdx.x; k < n16; k += kBlock) dst[k] = src[k];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(reinterpret_cast<uint64_t*>(dst + n16), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

#endif  // NSG_SPEC_BUILD

}  // namespace nsg
