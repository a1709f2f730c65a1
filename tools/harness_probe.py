#!/usr/bin/env python3
"""run_episodes (one episode per env, device-side accounting) against the bare fused rollouts it is built on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ns_gym_amd as nsg
from ns_gym_amd.evaluate import run_episodes
from ns_gym_amd.schedulers import ContinuousScheduler
from ns_gym_amd.update_functions import IncrementUpdate

for n in (1 << 16, 1 << 20):
    env = nsg.VecNSEnv(nsg.make("CartPole-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)}, n,
                       change_notification=True, delta_change_notification=True)
    run_episodes(env, seed=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = run_episodes(env, seed=1)
    dt = time.perf_counter() - t0
    steps = sum(r[2] for r in rows)
    longest = max(r[2] for r in rows)
    acts = torch.randint(0, 2, (64, n), dtype=torch.int32, device="cuda")
    env.reset(seed=1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range((longest + 63) // 64):
        env.rollout(acts, record=("reward", "terminated", "truncated"))
    torch.cuda.synchronize()
    dr = time.perf_counter() - t1
    print(f"N={n}: run_episodes {dt * 1e3:.1f} ms ({steps / dt / 1e9:.2f} G episode env-steps/s, longest episode {longest} steps; "
          f"building {n} result rows on the host included); the same number of bare rollout launches: {dr * 1e3:.1f} ms", flush=True)
    env.close()
