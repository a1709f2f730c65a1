#!/usr/bin/env python3
"""Long-horizon parity soak: a BASELINE config stepped T times (tens of thousands of steps, thousands of autoresets per env)
on the GPU and by the oracle's OpenMP stepper, every compared row checked every `--every` steps.
    python tests/soak/soak.py --spec c2_cartpole_gravity_rw --n 65536 --steps 20000
    python tests/soak/soak.py --spec c4_acrobot_mass2_inc --n 262144 --steps 3000 --exact      # NSG_F_LIBM_EXACT unit: every BIT of the float64 state"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from oracle.oracle import OracleVecEnv
from ns_gym_amd.vec_env import VecNSEnv
from tests.util import TRAJ_SPECS, GpuView, OracleView, compare_views, make_env_from_spec

ap = argparse.ArgumentParser()
ap.add_argument("--spec", default="c2_cartpole_gravity_rw")
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--steps", type=int, default=20000)
ap.add_argument("--every", type=int, default=1000)
ap.add_argument("--rollout", type=int, default=0, help="drive the GPU side through nsg_rollout, K fused steps per launch")
ap.add_argument("--exact", action="store_true", help="libm_exact=True, and the comparison is bit for bit (float64 state, observation, reward, theta, t, flags)")
args = ap.parse_args()
spec = TRAJ_SPECS[args.spec]
is_fl = spec["env_id"] == "FrozenLake-v1"
env = make_env_from_spec(lambda *a, **k: VecNSEnv(*a, **k), spec, n=args.n, track_returns=True, specialize=True, **({"libm_exact": True} if args.exact else {}))
orc = make_env_from_spec(OracleVecEnv, spec, n=args.n, track_returns=True)
seeds = np.arange(args.n, dtype=np.uint64) + np.uint64(777)
env.reset(seed=seeds)
orc.reset(seed=seeds)
g = torch.Generator(device="cuda").manual_seed(3)
threads = min(16, os.cpu_count() or 1)
t0 = time.time()
K = max(args.rollout, 1)
for k0 in range(0, args.steps, K):
    shape = (K, args.n)
    acts = (torch.rand(shape, device="cuda", generator=g) * 4 - 2) if env.action_is_float else \
        torch.randint(0, env.n_actions, shape, dtype=torch.int32, device="cuda", generator=g)
    if args.rollout:
        env.rollout(acts, record=("reward",))
    else:
        env.step(acts[0])
    host = acts.cpu().numpy()
    for j in range(K):
        orc.step_mt(host[j], threads)
    k = k0 + K - 1
    if (k + 1) % args.every < K or k >= args.steps - 1:
        if args.exact:
            from tests.test_gpu_libm_exact import _same_state

            _same_state(env, orc, f"{args.spec}: step {k}")
        else:
            compare_views(GpuView(env)._out(), OracleView(orc)._out(), is_fl, f"{args.spec}: step {k}")
        print(f"step {k + 1}: all {args.n} envs agree{' in every bit' if args.exact else ''} ({env.counters()['episodes']:,} episodes so far, {time.time() - t0:.0f} s)", flush=True)
print("soak ok")
