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
