#!/usr/bin/env python3
"""A context sweep in ONE batch.  The reference builds one env per context value - `make_env_with_context`: a
`ContinuousScheduler(start=0, end=0)` + `StepWiseUpdate(scheduler, [value])` that installs the value at t = 0 of every episode
(ns_gym/context_switching.py:20-69) - and loops `for context: for episode:` to chart how a policy generalises
(`calculate_generalized_performance`).  θ is per-env state here, so the whole chart is one batch: env i carries context i // E
(persistent_params keeps θ across resets; a NoUpdate fn names the tuned parameter), a linear policy runs inside the stepping
kernel, and `run_episodes` returns every episode's return."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ns_gym_amd as nsg
from ns_gym_amd.evaluate import run_episodes
from ns_gym_amd.policies import LinearPolicy
from ns_gym_amd.schedulers import ContinuousScheduler
from ns_gym_amd.update_functions import NoUpdate

C, E = 1024, 64                                   # context values, episodes per context
contexts = np.linspace(0.05, 5.0, C)              # pole masses; the policy below was tuned by hand around masspole = 0.1
env = nsg.VecNSEnv(nsg.make("CartPole-v1"), {"masspole": NoUpdate(ContinuousScheduler())}, num_envs=C * E, persistent_params=True)
env.reset(seed=0)
env.theta[0].copy_(torch.from_numpy(np.repeat(contexts, E)).to(env.device))      # env i lives in context i // E
policy = LinearPolicy([[0.3, -0.8, -2.0, -1.1, 0.05], [-0.3, 0.8, 2.0, 1.1, -0.05]])
cols = run_episodes(env, policy, seed=0, as_arrays=True)                          # 65 536 closed-loop episodes, 8 launches of 64 steps
ret = cols["total_reward"].reshape(C, E).mean(axis=1)
for lo in range(0, C, C // 8):
    print(f"masspole {contexts[lo]:5.2f} .. {contexts[lo + C // 8 - 1]:5.2f}: mean return {ret[lo:lo + C // 8].mean():6.1f}")
print(f"{C * E:,} episodes, {int(cols['num_steps'].sum()):,} env-steps in {cols['time']:.3f} s ({env.policy_kernels} kernels)")
env.close()
