#!/usr/bin/env python3
"""`libm_exact=True`: the same closed loop in the two arithmetics a batch can run.

A balanced CartPole is an unstable plant: two trajectories under IDENTICAL actions separate by x 1.12 per step, so the last ulp of a
sin - the kernels' own sincos is < 1 ulp, but not libm's last bit in 3 % of its evaluations - is a different action ~270 steps later
and a different episode after that.  The default arithmetic follows the reference's SEMANTICS exactly and its float32 state to
1e-5 until then; `libm_exact=True` evaluates sin / cos / scalar ** 2 / exp / log1p the way glibc does, rounding for rounding, so
the float64 state IS the reference's for as long as the loop runs (tests/test_gpu_libm_exact.py compares it with the oracle; this
example only shows the two arithmetics parting, and what the exact one costs)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ns_gym_amd as nsg
from ns_gym_amd.policies import EpisodeAccounts, LinearPolicy
from ns_gym_amd.schedulers import PeriodicScheduler
from ns_gym_amd.update_functions import RandomWalk

N, K, CHUNKS = 65_536, 500, 6
policy = LinearPolicy([[0.3, -0.8, -2.0, -1.1, 0.05], [-0.3, 0.8, 2.0, 1.1, -0.05]])      # keeps most poles up to the TimeLimit


def build(**kw):
    env = nsg.VecNSEnv(nsg.make("CartPole-v1"), {"gravity": RandomWalk(PeriodicScheduler(period=3))}, N,
                       change_notification=True, delta_change_notification=True, specialize=True, **kw)
    env.reset(seed=11)
    return env, EpisodeAccounts(env, gamma=None)


(fast, acc_f), (exact, acc_e) = build(), build(libm_exact=True)
parted = torch.zeros(N, dtype=torch.bool, device=fast.device)
for c in range(CHUNKS):
    t = []
    for env, acc in ((fast, acc_f), (exact, acc_e)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.rollout_policy(policy, K, accounts=acc, step0=c * K)      # K closed-loop steps in ONE launch, the policy inside the kernel
        torch.cuda.synchronize(); t.append((time.perf_counter() - t0) / K * 1e6)
    parted |= (fast.t != exact.t) | (fast.state != exact.state).any(dim=1)
    ulp = (fast.phys != exact.phys).any(dim=0) & ~parted
    print(f"after {(c + 1) * K:5d} steps: {int(parted.sum()):6d} of {N} envs on a different trajectory, {int(ulp.sum()):6d} more differ below the "
          f"float32 observation; " + (f"{t[0]:.2f} vs {t[1]:.2f} us per step (default / exact)" if c else "(the first launches build the units)"))
print(f"episodes finished: default {fast.counters()['episodes']:,}, exact {exact.counters()['episodes']:,}; kernels: {exact.policy_kernels}")
fast.close(); exact.close()
