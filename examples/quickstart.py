#!/usr/bin/env python3
"""The reference's tutorial loop (tutorial.ipynb: wrap CartPole, make `masspole` drift, step with notifications) -
once as the N = 1 drop-in with the reference's class names, once as a batch of 2^18 envs on the GPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ns_gym_amd as nsg
from ns_gym_amd.schedulers import ContinuousScheduler, PeriodicScheduler
from ns_gym_amd.update_functions import IncrementUpdate, RandomWalk
from ns_gym_amd.wrappers import NSClassicControlWrapper

# --- N = 1: the reference's wrapper, returns and dict layout ---------------------------------------------------------
params = {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1), "gravity": RandomWalk(PeriodicScheduler(period=3), sigma=0.5)}
env = NSClassicControlWrapper(nsg.make("CartPole-v1"), params, change_notification=True, delta_change_notification=True)
obs, info = env.reset(seed=42)
for _ in range(5):
    obs, reward, terminated, truncated, info = env.step(env.action_space.sample())
    print(f"t={obs['relative_time']} state={obs['state'].round(4)} changed={obs['env_change']} delta={obs['delta_change']} "
          f"masspole={env.unwrapped.masspole:.2f}")
    if terminated or truncated:
        obs, info = env.reset()
plan = env.get_planning_env()          # frozen snapshot for a planner (classic_control.py:120-136)
print("planning copy: is_sim_env =", plan.is_sim_env, " t =", plan.t)
env.close()

# --- the same wrapper configuration over 2^18 envs, one fused launch per step ----------------------------------------
N = 1 << 18
vec = nsg.VecNSEnv(nsg.make("CartPole-v1"), params, num_envs=N, change_notification=True, delta_change_notification=True,
                   track_returns=True)
obs, info = vec.reset(seed=0)          # env i == the N = 1 wrapper after reset(seed=i)
for _ in range(200):
    actions = torch.randint(0, 2, (N,), dtype=torch.int32, device="cuda")
    obs, reward, terminated, truncated, info = vec.step(actions)     # finished envs reset themselves on their next step
c = vec.counters()
returns, lengths = vec.episode_returns()
print(f"{c['env_steps']:,} env-steps, {c['episodes']:,} episodes finished, {c['updates_applied']:,} parameter updates; "
      f"mean return of the last finished episodes {returns[lengths > 0].mean().item():.1f}")
vec.close()

# --- the tutorial's agent loop (tutorial.ipynb cells 12, 26: `action = policy[observation]` on a FrozenLake whose slipperiness
#     decays): 2^16 episodes at once, the policy looked up and the returns summed INSIDE the stepping kernel -------------------
import numpy as np  # noqa: E402

from ns_gym_amd.evaluate import run_episodes  # noqa: E402
from ns_gym_amd.policies import TabularPolicy  # noqa: E402
from ns_gym_amd.update_functions import DistributionDecrementUpdate  # noqa: E402

lake = nsg.VecNSEnv(nsg.make("FrozenLake-v1", is_slippery=False, max_episode_steps=50), {"P": DistributionDecrementUpdate(ContinuousScheduler(), k=0.1)},
                    num_envs=1 << 16, change_notification=True, delta_change_notification=True, initial_prob_dist=[1, 0, 0])
go = {"L": 0, "D": 1, "R": 2, "U": 3}
stationary_policy = TabularPolicy([go[c] for c in "DRDL" "DLDL" "RDDL" "LRRL"])      # the 4x4 map's shortest safe path (what value iteration finds without slip)
cols = run_episodes(lake, stationary_policy, seed=0, as_arrays=True)                  # one launch per 64 steps; rows as NumPy columns
print(f"stationary policy on the drifting lake: success rate {cols['total_reward'].mean():.3f} over {len(cols['num_steps']):,} episodes, "
      f"mean length {cols['num_steps'].mean():.1f} ({lake.policy_kernels} kernels)")
lake.close()
