#!/usr/bin/env python3
"""Planner-shaped workload (the reference's heaviest step consumer, MCTS.py:131,162-181: deep-copy the env,
roll a simulation out, repeat): R root envs, S simulations per decision, each = fork(into=copy) + one fused
K-step rollout with random actions on the copy.  Reports simulated env-steps per second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ns_gym_amd import make
from ns_gym_amd.schedulers import ContinuousScheduler
from ns_gym_amd.update_functions import DistributionDecrementUpdate, IncrementUpdate
from ns_gym_amd.vec_env import VecNSEnv

S, K = 64, 32
for name, mk, tp, n_act, R in (
        ("CartPole", lambda: make("CartPole-v1"), lambda: {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.01)}, 2, 1 << 16),
        ("FrozenLake 8x8", lambda: make("FrozenLake-v1", map_name="8x8"), lambda: {"P": DistributionDecrementUpdate(ContinuousScheduler(), 0.01)}, 4, 1 << 16)):
    env = VecNSEnv(mk(), tp(), R, change_notification=True, delta_change_notification=True, specialize=True)
    env.reset(seed=0)
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = torch.randint(0, n_act, (K, R), dtype=torch.int32, device="cuda", generator=g)
    a0 = acts[0]
    for _ in range(5):
        env.step(a0)
    copy = env.get_planning_env()
    copy.rollout(acts, record=("reward", "terminated", "truncated"))
    decisions = 5
    for d in range(decisions + 1):   # decision 0 is untimed: it loads torch's reduction kernels (tens of ms in a fresh process)
        if d == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        ret = torch.zeros(R, device="cuda")
        for s in range(S):
            env.fork(theta_mode=0, into=copy)
            out = copy.rollout(acts, record=("reward", "terminated", "truncated"))
            ret += out["reward"].sum(0)
        env.step(a0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {R} roots x {S} simulations x {K} steps per decision, one fork + rollout PER SIMULATION: "
          f"{decisions * S * K * R / dt / 1e9:.1f} G simulated env-steps/s ({dt / decisions * 1e3:.2f} ms per decision)")
    # the same work with all simulations of a decision as ONE batch: fork(repeat=S) + one rollout launch
    big = env.fork(theta_mode=0, repeat=S)
    acts_big = acts.repeat(1, S)
    for _ in range(3):   # let the caching allocator settle on the two output buffers it alternates between
        out = big.rollout(acts_big, record=("reward", "terminated", "truncated"))
        ret = out["reward"].sum(0).view(S, R).mean(0)
    torch.cuda.synchronize()
    decisions = 10
    t0 = time.perf_counter()
    for d in range(decisions):
        env.fork(theta_mode=0, into=big)
        out = big.rollout(acts_big, record=("reward", "terminated", "truncated"))
        ret = out["reward"].sum(0).view(S, R).mean(0)
        env.step(a0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: ... all {S} simulations as ONE batch of {S * R} copies (fork(repeat={S})): "
          f"{decisions * S * K * R / dt / 1e9:.1f} G simulated env-steps/s ({dt / decisions * 1e3:.2f} ms per decision)")
    env.close(); copy.close(); big.close()
