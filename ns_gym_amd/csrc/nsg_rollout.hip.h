// nsg_rollout.hip.h — K wrapper steps per launch (caller supplies actions[K][N]).
//
// Every lane owns one env for the whole launch, so consecutive steps of an env are ordered by
// program order and no inter-workgroup synchronisation is needed.  Per-step outputs go to the
// k-th slice of the caller's trajectory buffers; the persistent state round-trips through the
// env's own SoA rows (L2-resident between consecutive steps of the same workgroup).
#pragma once
#include "nsg_kernels.hip.h"

namespace nsg {

template <int ENV, bool FULL>
__global__ __launch_bounds__(kBlock) void rollout_kernel(const Segment* __restrict__ seg, const void* __restrict__ actions,
                                                         int k_steps, nsg_rollout_out ro) {
  LdsTables lds;
  const Segment& sg = *seg;
  Tables tb;
  ZigLds zg;
  stage_tables(sg, lds, tb, zg);
  WaveCounts wc;
  const int64_t N = sg.N;
  const int P = sg.cfg.n_params > 0 ? sg.cfg.n_params : 1;
  const StepOut dflt = default_out(sg.buf);
  constexpr bool GRID = ENV == NSG_ENV_FROZENLAKE || ENV == NSG_ENV_CLIFFWALKING || ENV == NSG_ENV_BRIDGE;
  constexpr int D = GRID ? 1 : EnvTraits<GRID ? NSG_ENV_CARTPOLE : ENV>::OBS;
  constexpr bool FA = ENV == NSG_ENV_PENDULUM || ENV == NSG_ENV_MOUNTAINCAR_CONT;
  const int64_t chunks = (N + kBlock - 1) / kBlock;
  int parity = 0;
  for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    for (int k = 0; k < k_steps; k++) {
      StepOut out;
      out.obs = ro.obs ? ro.obs + (int64_t)k * N * D : (GRID ? nullptr : dflt.obs);
      out.reward = ro.reward ? ro.reward + (int64_t)k * N : dflt.reward;
      out.terminated = ro.terminated ? ro.terminated + (int64_t)k * N : dflt.terminated;
      out.truncated = ro.truncated ? ro.truncated + (int64_t)k * N : dflt.truncated;
      out.env_change = ro.env_change ? ro.env_change + (int64_t)k * P * N : dflt.env_change;
      out.delta_change = ro.delta_change ? ro.delta_change + (int64_t)k * P * N : dflt.delta_change;
      const void* act = FA ? (const void*)((const float*)actions + (int64_t)k * N) : (const void*)((const int32_t*)actions + (int64_t)k * N);
      step_block<ENV, FULL>(sg, tb, zg, act, out, c * kBlock, parity, lds, wc);
      parity ^= 1;
    }
  }
  flush_counts(sg, lds, wc);
}

}  // namespace nsg
