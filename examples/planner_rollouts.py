#!/usr/bin/env python3
"""What the planning-env snapshot is for (MCTS.py:131,162-181 deep-copies the env once per simulation and rolls it out):
here every root env gets S Monte-Carlo rollouts of depth K per candidate first action, all in ONE fork + ONE fused
rollout launch, and acts greedily on the estimated returns."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ns_gym_amd as nsg
from ns_gym_amd.schedulers import ContinuousScheduler
from ns_gym_amd.update_functions import IncrementUpdate

R, S, K, A = 4096, 32, 24, 2          # roots, simulations per action, rollout depth, actions
env = nsg.VecNSEnv(nsg.make("CartPole-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.01)}, num_envs=R,
                   change_notification=True, delta_change_notification=True, track_returns=True)
env.reset(seed=0)
sims = env.fork(theta_mode=0, repeat=A * S)             # copy j <- root j mod R, each with its own streams
first = torch.arange(A, dtype=torch.int32, device="cuda").repeat_interleave(S * R)        # block a of S*R copies tries action a
gamma = 0.99 ** torch.arange(K, dtype=torch.float32, device="cuda")[:, None]
t0, steps = time.perf_counter(), 0
for decision in range(100):
    env.fork(theta_mode=0, into=sims)                   # overwrite the copies in place: one launch
    acts = torch.randint(0, A, (K, A * S * R), dtype=torch.int32, device="cuda")
    acts[0] = first
    out = sims.rollout(acts, record=("reward", "terminated", "truncated"))
    alive = torch.cumsum((out["terminated"] | out["truncated"]).to(torch.int32), 0) == 0   # still in the first episode
    alive = torch.cat([torch.ones_like(alive[:1]), alive[:-1]])                            # the finishing step still counts
    value = (out["reward"] * alive * gamma).sum(0).view(A, S, R).mean(1)                   # [A, R]
    env.step(value.argmax(0).to(torch.int32))
    steps += K * A * S * R
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ret, length = env.episode_returns()
done = int((length > 0).sum())
tail = (f"mean length of the last finished episode {length[length > 0].float().mean().item():.0f}" if done
        else "every root is still in its first episode after 100 decisions")
print(f"{R} roots x 100 decisions: {steps / dt / 1e9:.1f} G simulated env-steps/s; episodes finished so far: {done}, {tail} "
      f"(random policy: ~22 steps per episode)")
env.close(); sims.close()
