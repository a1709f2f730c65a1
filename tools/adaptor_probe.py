#!/usr/bin/env python3
"""Host time per step() of the N = 1 adaptors (the reference's own calling pattern: one Python call per env step,
observation dict out): NSClassicControlWrapper on CartPole and NSFrozenLakeWrapper, and the parts a step is made of."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ns_gym_amd as nsg
from ns_gym_amd.schedulers import ContinuousScheduler
from ns_gym_amd.update_functions import IncrementUpdate, DistributionDecrementUpdate
from ns_gym_amd.wrappers import NSClassicControlWrapper, NSFrozenLakeWrapper


def rate(env, n_act, steps=3000):
    rng = np.random.default_rng(0)
    acts = rng.integers(0, n_act, size=steps + 200)
    env.reset(seed=0)
    for k in range(200):
        o, r, term, trunc, info = env.step(int(acts[k]))
        if term or trunc:
            env.reset()
    t0 = time.perf_counter()
    for k in range(200, 200 + steps):
        o, r, term, trunc, info = env.step(int(acts[k]))
        if term or trunc:
            env.reset()
    return (time.perf_counter() - t0) / steps * 1e6


cp = NSClassicControlWrapper(nsg.make("CartPole-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)},
                             change_notification=True, delta_change_notification=True)
print(f"NSClassicControlWrapper(CartPole, masspole Increment): {rate(cp, 2):6.1f} us per step() incl. resets", flush=True)
v = cp._vec
a = torch.zeros(1, dtype=torch.int32, device=v.device)
for name, f in (("action fill_", lambda: a.fill_(1)), ("VecNSEnv.step", lambda: v.step(a)), ("host_rows", lambda: v.host_rows())):
    for _ in range(100):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        f()
    torch.cuda.synchronize()
    print(f"   {name:14s} {(time.perf_counter() - t0) / 2000 * 1e6:6.1f} us per call", flush=True)
fl = NSFrozenLakeWrapper(nsg.make("FrozenLake-v1"), {"P": DistributionDecrementUpdate(ContinuousScheduler(), k=0.01)},
                         change_notification=True, delta_change_notification=True, initial_prob_dist=[1.0, 0.0, 0.0])
print(f"NSFrozenLakeWrapper(4x4, P Decrement):                 {rate(fl, 4):6.1f} us per step() incl. resets", flush=True)
cp.close(); fl.close()
