// nsg_rollout.hip.h — K wrapper steps per launch (caller supplies actions[K][N]).
//
// Every lane owns one env for the whole launch, so consecutive steps of an env are ordered by
// program order and no inter-workgroup synchronisation is needed.  Per-step outputs go to the
// k-th slice of the caller's trajectory buffers.  For the classic-control envs the persistent rows
// (integrator state, t, status, θ rows 0-1, episode return) stay in REGISTERS between the K steps
// (LaneState): HBM sees them once per launch, so a step costs the action (4 B) plus its outputs
// (~31 B for CartPole) instead of ~140 B.  Grid envs keep cell, t, status, episode return, the env stream and the
// table probabilities in registers the same way (GridLane): ~23 B per step instead of ~117 B.
#pragma once
#include "nsg_kernels.hip.h"

namespace nsg {

// ---- fused policy rollouts (nsg_rollout_policy): the action of step k is computed in the lane from what step k - 1 left there ----
// The loops this replaces: MCTS._default_policy (benchmark_algorithms/MCTS.py:162-181), run_episode
// (evaluate/run_experiment.py:108-129), the tutorial's tabular run_episode (tutorial.ipynb cell 12).
struct PolicyArgs {
  nsg_policy pol;
  nsg_episode_acc acc;
  float act_lo, act_hi;   // continuous env types: the action bounds (Box.sample / clip)
  int32_t n_actions;      // discrete env types
  int32_t reserved;
};

// NSG_POL_UNIFORM: two rounds of the splitmix64 finaliser over (key, env, step) - a pure function, so any sharding of the batch and any
// chunking of the steps draw the same actions (host-callable: nsg_policy_bits, the oracle and the tests mirror it)
__host__ __device__ __forceinline__ uint64_t pol_mix(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ uint64_t pol_bits(uint64_t seed, uint64_t env, uint64_t step) {
  return pol_mix(pol_mix(seed + 0x9E3779B97F4A7C15ull * (env + 1u)) + 0xD1B54A32D192ED03ull * (step + 1u));
}
__device__ __forceinline__ int pol_uniform_discrete(uint64_t bits, int n_actions) { return (int)(((bits >> 32) * (uint64_t)(uint32_t)n_actions) >> 32); }
__device__ __forceinline__ float pol_uniform_float(uint64_t bits, float lo, float hi) { return lo + (hi - lo) * ((float)(bits >> 40) * 5.9604644775390625e-08f); }

// NSG_POL_LINEAR on the float32 observation: score_j = W[j][D] + sum_d W[j][d] * o[d], float32, in that order (the weights are uniform:
// scalar loads).  Discrete: first argmax.  Continuous: clip(score_0).
template <int D, bool FLOAT_ACT>
__device__ __forceinline__ void pol_linear(const PolicyArgs& pa, const float* o, int& ai, float& af) {
  typedef const __attribute__((address_space(4))) float* scalar_f32;
  scalar_f32 W = (scalar_f32)(uint64_t)pa.pol.data;
  if constexpr (FLOAT_ACT) {
    float sc = W[D];
#pragma unroll
    for (int d = 0; d < D; d++) sc = sc + W[d] * o[d];
    af = sc < pa.act_lo ? pa.act_lo : sc > pa.act_hi ? pa.act_hi : sc;
  } else {
    int best = 0;
    float top = 0.f;
    for (int j = 0; j < pa.n_actions; j++) {
      float sc = W[j * (D + 1) + D];
#pragma unroll
      for (int d = 0; d < D; d++) sc = sc + W[j * (D + 1) + d] * o[d];
      if (j == 0 || sc > top) { top = sc; best = j; }
    }
    ai = best;
  }
}

// `block_rel` / `block_count`: the workgroup's index within, and the size of, the range of workgroups that walks this segment
// (the whole launch for nsg_rollout; a member's block range for nsg_rollout_group).
// POL: nsg_rollout_policy - `pa` says where the actions come from and where the episode accounts live; everything else is the same
// launch (POL = false compiles to exactly the table-driven rollout).  KIND: the action source as a compile-time constant (the
// specialised units: one small unit per (config, kind), so that a uniform rollout does not carry the linear policy's observation
// registers across its steps), or -1: read from `pa` (the precompiled kernels).
template <int ENV, bool FULL, bool POL = false, int KIND = -1>
__device__ __forceinline__ void rollout_body(const nsg_config& cfg, const Segment& sg, const void* __restrict__ actions,
                                             int k_steps, const nsg_rollout_out& ro, const int block_rel, const int block_count,
                                             const PolicyArgs* __restrict__ pa = nullptr) {
  LdsTables lds;
  Tables tb;
  ZigLds zg;
  constexpr bool GRID_ENV = ENV == NSG_ENV_FROZENLAKE || ENV == NSG_ENV_CLIFFWALKING || ENV == NSG_ENV_BRIDGE;
  stage_tables<false, true, GRID_ENV ? 2 : 1>(sg, lds, tb, zg);
  WaveCounts wc;
  const nsg_buffers& b = sg.buf;
  const int64_t N = sg.N;
  const int P = cfg.n_params > 0 ? cfg.n_params : 1;
  const StepOut dflt = default_out(b);
  constexpr bool GRID = ENV == NSG_ENV_FROZENLAKE || ENV == NSG_ENV_CLIFFWALKING || ENV == NSG_ENV_BRIDGE;
  constexpr int D = GRID ? 1 : EnvTraits<GRID ? NSG_ENV_CARTPOLE : ENV>::OBS;
  constexpr bool FA = ENV == NSG_ENV_PENDULUM || ENV == NSG_ENV_MOUNTAINCAR_CONT;
  const int64_t chunks = (N + kBlock - 1) / kBlock;
  int parity = 0;
  for (int64_t c = block_rel; c < chunks; c += block_count) {
    [[maybe_unused]] LaneState<GRID ? NSG_ENV_CARTPOLE : ENV> ls;
    [[maybe_unused]] GridLane<ENV == NSG_ENV_CLIFFWALKING ? 4 : 3> gl;
    // classic envs: the first stochastic update fns' streams live in LDS for the K steps (each lane touches only its own
    // record); the env's own np_random needs no state at all (episode word + jump-ahead, nsg_rng.hip.h)
    [[maybe_unused]] const int64_t ir = c * kBlock + threadIdx.x;
    // policy rollouts: the lane's episode accounts, and what its first decision looks at - the handle's own observation row (the last
    // reset / step left it there: nsg_step's and nsg_rollout's contract), for the grid envs the cell row
    [[maybe_unused]] double acc_ret = 0.0;
    [[maybe_unused]] int acc_len = 0;
    [[maybe_unused]] bool acc_alive = false;
    [[maybe_unused]] int cell0 = 0;
    if constexpr (POL) {
      if (ir < N) {
        const nsg_episode_acc& ac = pa->acc;
        acc_alive = ac.alive ? ldg(ac.alive, (uint32_t)ir) != 0 : true;
        if (ac.ret) acc_ret = ldg(ac.ret, (uint32_t)ir * 8u);
        if (ac.length) acc_len = ldg(ac.length, (uint32_t)ir * 4u);
        const int kind0 = KIND >= 0 ? KIND : pa->pol.kind;
        if (kind0 == NSG_POL_LINEAR) {
          if constexpr (!GRID) {
#pragma unroll
            for (int q = 0; q < D; q++) ls.o[q] = ldg(b.obs, (uint32_t)ir * (uint32_t)(4 * D) + 4u * (uint32_t)q);
          }
        }
        if (kind0 == NSG_POL_BY_STATE) {
          if constexpr (GRID) cell0 = ldg(b.cell, (uint32_t)ir * 4u);
        }
      }
    }
    if constexpr (!GRID) {
      // every lane derives its env stream ONCE per launch (seed -> jump to draw D * episodes so far: ~500 instructions, all lanes
      // busy) and parks it in LDS: the resets of the K fused steps then draw sequentially from there (4 PCG64 steps each)
      // instead of re-deriving the stream at every reset like nsg_step's hand-over has to
      if (ir < N) {
        Pcg g;
        const uint64_t count = (uint64_t)((uint32_t)ldg(b.episode, (uint32_t)ir * 4u) >> NSG_EP_COUNT_SHIFT);
        const u64x2 desc = {zg.sd0, zg.sd1};
        env_stream_at(b.rng_env, ir, count * (uint64_t)EnvTraits<GRID ? NSG_ENV_CARTPOLE : ENV>::RESET_DRAWS, zg.jump, g, &desc);
        uint64_t* rec = lds.streams + threadIdx.x * 4;
        rec[0] = g.sh; rec[1] = g.sl; rec[2] = g.ih; rec[3] = g.il;
      }
      if constexpr (!FULL) __syncthreads();
    }
    if constexpr (!GRID && FULL) {
      if (ir < N) {
        for (int p = 0; p < cfg.n_params; p++) {
          const nsg_param_cfg& pc = cfg.params[p];
          if (!pc.uses_rng || pc.fn_slot != p || upd_lds_index(cfg, p) >= kMaxLdsUpd) continue;
          Pcg u;
          pcg_load(b.rng_upd + (int64_t)p * 4 * N, N, ir, u);
          uint64_t* ur = lds.ustreams + ((int64_t)upd_lds_index(cfg, p) * kBlock + threadIdx.x) * 4;
          ur[0] = u.sh; ur[1] = u.sl; ur[2] = u.ih; ur[3] = u.il;
        }
      }
      __syncthreads();
    }
    for (int k = 0; k < k_steps; k++) {
      StepOut out;
      // the LAST step writes into the env's own output rows (so the handle's buffers describe the env
      // after the rollout exactly as after nsg_step); nsg_rollout then copies them into the last slice
      const bool last = k == k_steps - 1;
      if (last) {
        out = dflt;
        if (GRID) out.obs = nullptr;
      } else {
      // a row the caller does not record is not stored by the steps before the last (IoMode::opt_out): 27 B per CartPole env-step that
      // the next step would only overwrite
      out.obs = ro.obs ? ro.obs + (int64_t)k * N * D : nullptr;
      out.reward = ro.reward ? ro.reward + (int64_t)k * N : nullptr;
      out.terminated = ro.terminated ? ro.terminated + (int64_t)k * N : nullptr;
      out.truncated = ro.truncated ? ro.truncated + (int64_t)k * N : nullptr;
      out.env_change = ro.env_change ? ro.env_change + (int64_t)k * P * N : nullptr;
      out.delta_change = ro.delta_change ? ro.delta_change + (int64_t)k * P * N : nullptr;
      }
      const void* act = FA ? (const void*)((const float*)actions + (int64_t)k * N) : (const void*)((const int32_t*)actions + (int64_t)k * N);
      if constexpr (POL) {
        // ---- the decision: one action per lane, from the lane's own registers (a lane beyond the batch decides nothing) ----
        int ai = 0;
        float af = 0.f;
        if (ir < N) {
          const int kind = KIND >= 0 ? KIND : pa->pol.kind;
          if (kind == NSG_POL_TABLE) {
            if constexpr (FA) af = ldg((const float*)act, (uint32_t)ir * 4u);
            else ai = ldg((const int32_t*)act, (uint32_t)ir * 4u);
          } else if (kind == NSG_POL_UNIFORM) {
            const uint64_t bits = pol_bits(pa->pol.seed, (uint64_t)(pa->pol.index0 + ir), (uint64_t)(uint32_t)(pa->pol.step0 + k));
            if constexpr (FA) af = pol_uniform_float(bits, pa->act_lo, pa->act_hi);
            else ai = pol_uniform_discrete(bits, pa->n_actions);
          } else if (kind == NSG_POL_BY_STATE) {
            if constexpr (GRID) ai = ldg((const int32_t*)pa->pol.data, (uint32_t)(k == 0 ? cell0 : gl.cell) * 4u);
          } else {
            if constexpr (!GRID) pol_linear<D, FA>(*pa, ls.o, ai, af);
          }
          if (pa->pol.actions_out) {
            if constexpr (FA) stg_out((float*)pa->pol.actions_out + (int64_t)k * N, (uint32_t)ir * 4u, af);
            else stg_out((int32_t*)pa->pol.actions_out + (int64_t)k * N, (uint32_t)ir * 4u, ai);
          }
        }
        double rw;
        unsigned fl;
        if constexpr (GRID) {
          gl.ai = ai;
          step_grid<ENV, FULL>(cfg, b, N, tb, zg, act, out, ir, ir < N, wc, gl, IoMode{k == 0, k == k_steps - 1, k > 0, false, false, false, !last, true});
          rw = gl.rw; fl = gl.fl;
        } else {
          ls.ai = ai; ls.af = af;
          step_chunk<ENV, FULL>(cfg, b, N, tb, zg, act, out, c * kBlock, parity, lds, wc, ls, IoMode{k == 0, k == k_steps - 1, k > 0, true, false, false, !last, true});
          rw = ls.rw; fl = ls.fl;
        }
        // ---- the accounts: tot_reward += reward * gamma ** depth (MCTS.py:179), total_reward += reward (run_experiment.py:117) ----
        if (ir < N && acc_alive && (fl & 4u)) {
          const double g = (pa->acc.discount && acc_len < pa->acc.n_discount) ? ldg(pa->acc.discount, (uint32_t)acc_len * 8u) : 1.0;
          acc_ret = acc_ret + rw * g;
          acc_len++;
          if (fl & 3u) acc_alive = false;
        }
      } else if constexpr (GRID) {
        const int64_t ig = c * kBlock + threadIdx.x;
        step_grid<ENV, FULL>(cfg, b, N, tb, zg, act, out, ig, ig < N, wc, gl, IoMode{k == 0, k == k_steps - 1, k > 0, false, false, false, !last});
      } else {
        step_chunk<ENV, FULL>(cfg, b, N, tb, zg, act, out, c * kBlock, parity, lds, wc, ls, IoMode{k == 0, k == k_steps - 1, k > 0, true, false, false, !last});
      }
      parity ^= 1;
    }
    if constexpr (POL) {
      if (ir < N) {
        const nsg_episode_acc& ac = pa->acc;
        if (ac.alive) stg(ac.alive, (uint32_t)ir, (uint8_t)(acc_alive ? 1 : 0));
        if (ac.ret) stg(ac.ret, (uint32_t)ir * 8u, acc_ret);
        if (ac.length) stg(ac.length, (uint32_t)ir * 4u, (int32_t)acc_len);
      }
    }
    if constexpr (!GRID && !FULL) __syncthreads();  // the next chunk refills the env streams
    if constexpr (!GRID && FULL) {
      __syncthreads();
      if (ir < N) {
        for (int p = 0; p < cfg.n_params; p++) {
          const nsg_param_cfg& pc = cfg.params[p];
          if (!pc.uses_rng || pc.fn_slot != p || upd_lds_index(cfg, p) >= kMaxLdsUpd) continue;
          const uint64_t* ur = lds.ustreams + ((int64_t)upd_lds_index(cfg, p) * kBlock + threadIdx.x) * 4;
          Pcg u = {ur[0], ur[1], 0, 0};
          pcg_store_state(b.rng_upd + (int64_t)p * 4 * N, N, ir, u);
        }
      }
      __syncthreads();  // the next chunk refills the records
    }
  }
  flush_counts(b.counters, block_rel, wc);
}

// The plain-arithmetic CartPole rollout sits at the 80-VGPR boundary (80 in round 1, 82 after round 2's hand-over changes: five
// wavefronts per SIMD instead of six, +4 % per step); the bound keeps it at six without spilling.
template <int ENV, bool FULL> constexpr int kRolloutMinWaves = (ENV == NSG_ENV_CARTPOLE && !FULL) ? 6 : 1;

template <int ENV, bool FULL>
__global__ __launch_bounds__(kBlock, (kRolloutMinWaves<ENV, FULL>)) void rollout_kernel(const Segment* __restrict__ seg, const void* __restrict__ actions,
                                                         int k_steps, nsg_rollout_out ro) {
  rollout_body<ENV, FULL>(seg->cfg, *seg, actions, k_steps, ro, (int)blockIdx.x, (int)gridDim.x);
}

#ifndef NSG_SPEC_BUILD
// (no register bound: the decision and the accounts ride on top of the table-driven rollout's registers)
template <int ENV, bool FULL>
__global__ __launch_bounds__(kBlock) void rollout_policy_kernel(const Segment* __restrict__ seg, int k_steps, nsg_rollout_out ro, PolicyArgs pa) {
  rollout_body<ENV, FULL, true>(seg->cfg, *seg, pa.pol.data, k_steps, ro, (int)blockIdx.x, (int)gridDim.x, &pa);
}
#endif

// ============================================================================================
// Resident stepper (nsg_resident_start): ONE launch that stays on the device and takes a step whenever the producer of the
// actions says the next action row is in place - for closed loops (policy kernel -> step -> policy kernel) in the launch-bound
// regime, where a dependent launch per step costs more than the step (C2 at 65 536 envs: 6.6 us per nsg_step launch, 2.7 us per
// step inside nsg_rollout).  One workgroup per 256-env chunk, all resident (the host refuses batches beyond that); state lives
// in registers / LDS between steps exactly as in rollout_body, and every step ALSO stores its persistent rows, so whenever the
// kernel leaves - max_steps reached, stop requested, or the wait for the next action row outlasted its budget - the handle's
// buffers describe the env as after that many nsg_step calls and a normal launch can carry on.
//
// Hand-shake (nsg_mailbox, device memory, agent scope), PER CHUNK: the producer writes chunk j's actions of step k, then
// act_seq[j] = k + 1 (release); the chunk's workgroup polls it (its first lane; acquire on success) - BOUNDED: the device's steady
// wall clock is read in the loop (resident_wait); after its step the workgroup fences (release) and publishes step_seq[j] = k + 1.
// No workgroup ever waits for another one of this kernel, and nothing in a step is a same-address atomic.
// ============================================================================================
struct ResidentArgs {
  nsg_mailbox* mb;
  int32_t max_steps;
  int32_t reserved;
  uint64_t budget_ticks;     // of the device's steady wall clock (hipDeviceAttributeWallClockRate): how long a workgroup waits for its next action row
  uint64_t grace_ticks;      // how long it keeps looking for one more row after `stop` has been raised (see resident_wait)
};

__device__ __forceinline__ uint64_t mb_peek(const uint64_t* p) {   // polling read: agent scope, no ordering (the fence follows success)
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mb_publish(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// One lane's bounded wait for `*seq >= want`.  Returns 1 (go), 0 (leave).  Nobody waits for ever: past `budget_ticks` the waiter
// raises mb->stop itself (NSG_MB_STARVED); from the moment a waiter sees `stop` raised - by itself, by another workgroup, by the host -
// it keeps polling for `grace_ticks` more and then leaves.  A producer reads `stop` before it publishes and publishes within a few
// microseconds of that read; with a grace period far beyond that, a row that a producer publishes for ALL chunks in the shadow of a
// stop (nsg_resident_publish) is seen by every workgroup (each polls until at least stop-time + grace): all of them take that step
// or none does.  (Chunks whose producers run independently of each other - the demo policy's workgroups - are wherever each of
// them got to when the stop reached it: mb->steps_done / steps_max, step_seq[j].)
__device__ __forceinline__ int resident_wait(nsg_mailbox* mb, const uint64_t* seq, uint64_t want, uint64_t budget_ticks, uint64_t grace_ticks,
                                             uint64_t starved_code) {
  const uint64_t t0 = (uint64_t)wall_clock64();
  uint64_t deadline = t0 + budget_ticks;
  bool draining = false;
  for (unsigned spin = 0;; spin++) {
    if (mb_peek(seq) >= want) return 1;   // (what was published before `seq` is read with coherent loads: no cache invalidate here)
    const uint64_t now = (uint64_t)wall_clock64();
    if (!draining && (spin & 7u) == 7u && mb_peek(&mb->stop) != 0u) {   // (every 8th poll: the hot path is one coherent load per poll)
      draining = true;
      deadline = now + grace_ticks;
    }
    if (now > deadline) {
      if (draining) return 0;
      uint64_t zero = 0u;    // first to give up says why; everybody (this lane included) then drains
      __hip_atomic_compare_exchange_strong(&mb->stop, &zero, starved_code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      draining = true;
      deadline = now + grace_ticks;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int ENV, bool FULL>
__device__ __forceinline__ void resident_body(const nsg_config& cfg, const Segment& sg, const void* __restrict__ actions, const ResidentArgs ra) {
  __shared__ int go_on;
  LdsTables lds;
  Tables tb;
  ZigLds zg;
  constexpr bool GRID = ENV == NSG_ENV_FROZENLAKE || ENV == NSG_ENV_CLIFFWALKING || ENV == NSG_ENV_BRIDGE;
  stage_tables<false, true, GRID ? 2 : 1>(sg, lds, tb, zg);
  WaveCounts wc;
  const nsg_buffers& b = sg.buf;
  const int64_t N = sg.N;
  StepOut out = default_out(b);
  if (GRID) out.obs = nullptr;
  const int64_t c = blockIdx.x;                       // one chunk per workgroup, all resident (checked by the host)
  const unsigned n_wg = gridDim.x;
  [[maybe_unused]] LaneState<GRID ? NSG_ENV_CARTPOLE : ENV> ls;
  [[maybe_unused]] GridLane<ENV == NSG_ENV_CLIFFWALKING ? 4 : 3> gl;
  [[maybe_unused]] const int64_t ir = c * kBlock + threadIdx.x;
  if constexpr (!GRID) {   // the chunk's env streams, derived once per launch (see rollout_body)
    if (ir < N) {
      Pcg g;
      const uint64_t count = (uint64_t)((uint32_t)ldg(b.episode, (uint32_t)ir * 4u) >> NSG_EP_COUNT_SHIFT);
      const u64x2 desc = {zg.sd0, zg.sd1};
      env_stream_at(b.rng_env, ir, count * (uint64_t)EnvTraits<GRID ? NSG_ENV_CARTPOLE : ENV>::RESET_DRAWS, zg.jump, g, &desc);
      uint64_t* rec = lds.streams + threadIdx.x * 4;
      rec[0] = g.sh; rec[1] = g.sl; rec[2] = g.ih; rec[3] = g.il;
    }
    if constexpr (FULL) {
      if (ir < N) {
        for (int p = 0; p < cfg.n_params; p++) {
          const nsg_param_cfg& pc = cfg.params[p];
          if (!pc.uses_rng || pc.fn_slot != p || upd_lds_index(cfg, p) >= kMaxLdsUpd) continue;
          Pcg u;
          pcg_load(b.rng_upd + (int64_t)p * 4 * N, N, ir, u);
          uint64_t* ur = lds.ustreams + ((int64_t)upd_lds_index(cfg, p) * kBlock + threadIdx.x) * 4;
          ur[0] = u.sh; ur[1] = u.sl; ur[2] = u.ih; ur[3] = u.il;
        }
      }
    }
    __syncthreads();
  }
  int parity = 0, taken = 0;
  for (int k = 0; k < ra.max_steps; k++) {
    if (threadIdx.x == 0) go_on = resident_wait(ra.mb, &ra.mb->act_seq[c], (uint64_t)k + 1u, ra.budget_ticks, ra.grace_ticks, NSG_MB_STARVED);
    __syncthreads();
    const int ok = go_on;
    __syncthreads();            // (go_on is rewritten by the next iteration's first lane)
    if (!ok) break;
    if constexpr (GRID) {
      // (a grid env's observation IS its cell row, its info["prob"] the prob row: both live in the persistent-store block - every step stores)
      // wt: the persistent rows (cell = the observation, prob) are written through as well
      step_grid<ENV, FULL>(cfg, b, N, tb, zg, actions, out, ir, ir < N, wc, gl, IoMode{k == 0, true, k > 0, false, true, true});
    } else {
      // outputs every step; the persistent rows stay in registers / LDS until the workgroup leaves (flush below)
      step_chunk<ENV, FULL>(cfg, b, N, tb, zg, actions, out, c * kBlock, parity, lds, wc, ls, IoMode{k == 0, false, k > 0, true, false, true});
    }
    parity ^= 1;
    taken++;
    // This chunk's outputs were written THROUGH the L2 one by one (IoMode::coh): once every lane's stores have been acknowledged
    // (vmcnt(0)) they are where the consumer's coherent loads look, and the sequence word - written through as well, after the
    // barrier - cannot overtake them.  No cache-wide writeback.
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&ra.mb->step_seq[c], (uint64_t)k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (!GRID) {   // leaving: the persistent rows go back (what a step with io.store does, step_chunk)
    if (taken > 0 && ir < N) {
      using T = EnvTraits<GRID ? NSG_ENV_CARTPOLE : ENV>;
      const uint32_t o4 = (uint32_t)ir * 4u, o8 = (uint32_t)ir * 8u;
#pragma unroll
      for (int q = 0; q < T::PHYS; q++) stg(b.phys, blk_off8(T::PHYS, q, ir), ls.s[q]);
      stg(b.t, o4, ls.t);
      stg(b.episode, o4, (int32_t)ls.st);
      if (cfg.n_params > 0) stg(b.theta, o8, ls.th0);
      if (cfg.n_params > 1) stg(b.theta + N, o8, ls.th1);
      if ((cfg.flags & NSG_F_TRACK_RETURNS) && T::RETURN_PER_STEP == 0.f) stg(b.ep_return, o4, ls.er);
    }
  }
  if constexpr (!GRID && FULL) {   // the update-fn streams held in LDS go back to their rows
    __syncthreads();
    if (ir < N) {
      for (int p = 0; p < cfg.n_params; p++) {
        const nsg_param_cfg& pc = cfg.params[p];
        if (!pc.uses_rng || pc.fn_slot != p || upd_lds_index(cfg, p) >= kMaxLdsUpd) continue;
        const uint64_t* ur = lds.ustreams + ((int64_t)upd_lds_index(cfg, p) * kBlock + threadIdx.x) * 4;
        Pcg u = {ur[0], ur[1], 0, 0};
        pcg_store_state(b.rng_upd + (int64_t)p * 4 * N, N, ir, u);
      }
    }
  }
  flush_counts(b.counters, (int)blockIdx.x, wc);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {       // the last workgroup out reports how the launch ended and how many steps the chunks have taken
    atomicMax((unsigned long long*)&ra.mb->taken_max, (unsigned long long)taken);
    atomicMax((unsigned long long*)&ra.mb->taken_min_inv, 0xffffffffULL - (unsigned long long)taken);
    const unsigned long long seen = atomicAdd((unsigned long long*)&ra.mb->leave, 1ULL);
    if (seen + 1ULL == (unsigned long long)n_wg) {
      const uint64_t mx = mb_peek(&ra.mb->taken_max), mn = 0xffffffffULL - mb_peek(&ra.mb->taken_min_inv);
      const uint64_t stop = mb_peek(&ra.mb->stop);
      mb_publish(&ra.mb->steps_done, mn);
      mb_publish(&ra.mb->steps_max, mx);
      mb_publish(&ra.mb->status, mn == (uint64_t)ra.max_steps ? (uint64_t)NSG_MB_FINISHED : stop == NSG_MB_STARVED ? (uint64_t)NSG_MB_STARVED : (uint64_t)NSG_MB_STOPPED);
    }
  }
}

// Heterogeneous fused rollout (nsg_rollout_group): K steps of every member in ONE launch, a block range per member like
// step_group_kernel; each member's persistent rows stay in registers / LDS for the K steps exactly as in nsg_rollout.
struct RolloutOuts {
  nsg_rollout_out o[NSG_MAX_SEGMENTS];
};

#ifndef NSG_SPEC_BUILD
template <int ENV, bool FULL>
__global__ __launch_bounds__(kBlock) void resident_kernel(const Segment* __restrict__ seg, const void* __restrict__ actions, ResidentArgs ra) {
  resident_body<ENV, FULL>(seg->cfg, *seg, actions, ra);
}

// A stand-in for the caller's policy (tools / tests / bench: the closed loop needs SOMETHING on the other side of the mailbox): a
// resident kernel that, for every step, waits for the stepper's step_seq (bounded by the same rules), writes the discrete action
// ((obs[i][watch] > 0) + k) mod n_actions for every env, reads `stop` and - if it is clear - publishes act_seq.  One workgroup per chunk.
__global__ __launch_bounds__(kBlock) void resident_demo_policy_kernel(const float* __restrict__ obs, int obs_dim, int watch, int32_t* __restrict__ actions,
                                                                     int64_t N, int n_actions, ResidentArgs ra) {
  __shared__ int go_on;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  for (int k = 0; k < ra.max_steps; k++) {
    // the observation of step k - 1 (the reset observation before the first step: nothing to wait for)
    if (threadIdx.x == 0) go_on = resident_wait(ra.mb, &ra.mb->step_seq[blockIdx.x], (uint64_t)k, ra.budget_ticks, ra.grace_ticks, NSG_MB_STARVED);
    __syncthreads();
    const int ok = go_on;
    __syncthreads();
    if (!ok) return;
    if (i < N) {   // coherent accesses on both sides (IoMode::coh): the observation past the L2, the action through it
      const float x = __hip_atomic_load(obs + i * obs_dim + watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(actions + i, (int32_t)(((x > 0.f ? 1 : 0) + k) % n_actions), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    // a producer never publishes once it has seen `stop` (resident_wait's grace period relies on it)
    if (threadIdx.x == 0 && mb_peek(&ra.mb->stop) == 0u)
      __hip_atomic_store(&ra.mb->act_seq[blockIdx.x], (uint64_t)k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The producer side for callers whose policy is ORDINARY kernels launched per step (or the host): enqueued behind them on their
// stream, it publishes "the action rows of step k are in place" for every chunk - unless stop has been raised.
__global__ void resident_publish_kernel(nsg_mailbox* mb, int n_chunks, uint64_t seq) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_chunks && mb_peek(&mb->stop) == 0u) __hip_atomic_store(&mb->act_seq[j], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool FULL>
__global__ __launch_bounds__(kBlock) void rollout_group_kernel(const Segment* __restrict__ segs, int nseg, ActionPtrs acts, int k_steps,
                                                               RolloutOuts outs) {
  const int sidx = group_segment_of_block(segs, nseg);
  const Segment& sg = segs[sidx];
  const void* actions = acts.p[sidx];
  const nsg_rollout_out& ro = outs.o[sidx];
  const int rel = (int)blockIdx.x - sg.block_begin;
  const int cnt = sg.block_count;
  switch (sg.cfg.env_type) {
    case NSG_ENV_CARTPOLE: rollout_body<NSG_ENV_CARTPOLE, FULL>(sg.cfg, sg, actions, k_steps, ro, rel, cnt); break;
    case NSG_ENV_PENDULUM: rollout_body<NSG_ENV_PENDULUM, FULL>(sg.cfg, sg, actions, k_steps, ro, rel, cnt); break;
    case NSG_ENV_ACROBOT: rollout_body<NSG_ENV_ACROBOT, FULL>(sg.cfg, sg, actions, k_steps, ro, rel, cnt); break;
    case NSG_ENV_MOUNTAINCAR: rollout_body<NSG_ENV_MOUNTAINCAR, FULL>(sg.cfg, sg, actions, k_steps, ro, rel, cnt); break;
    case NSG_ENV_MOUNTAINCAR_CONT: rollout_body<NSG_ENV_MOUNTAINCAR_CONT, FULL>(sg.cfg, sg, actions, k_steps, ro, rel, cnt); break;
    case NSG_ENV_FROZENLAKE: rollout_body<NSG_ENV_FROZENLAKE, FULL>(sg.cfg, sg, actions, k_steps, ro, rel, cnt); break;
    case NSG_ENV_CLIFFWALKING: rollout_body<NSG_ENV_CLIFFWALKING, FULL>(sg.cfg, sg, actions, k_steps, ro, rel, cnt); break;
    default: rollout_body<NSG_ENV_BRIDGE, FULL>(sg.cfg, sg, actions, k_steps, ro, rel, cnt); break;
  }
}
#endif

}  // namespace nsg
