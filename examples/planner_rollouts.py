#!/usr/bin/env python3
"""What the planning-env snapshot and the fused policy rollout are for.  The reference's MCTS deep-copies the env once per
simulation, steps the expanded action and rolls the copy out with uniformly random actions, summing discounted rewards
(MCTS.py:131,204,162-181).  Here every root env gets S such simulations per candidate first action, ALL of them in one fork into
a standing batch of copies plus one fused launch (`ns_gym_amd.planning.Simulator`: the actions are drawn and the discounted
returns kept inside the kernel), and acts greedily on the estimated returns."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ns_gym_amd as nsg
from ns_gym_amd.planning import Simulator
from ns_gym_amd.schedulers import ContinuousScheduler
from ns_gym_amd.update_functions import IncrementUpdate

R, S, K, A, GAMMA = 4096, 32, 24, 2, 0.99          # roots, simulations per action, rollout depth, actions, discount
env = nsg.VecNSEnv(nsg.make("CartPole-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.01)}, num_envs=R,
                   change_notification=True, delta_change_notification=True, track_returns=True)
env.reset(seed=0)
sim = Simulator(env, sims=A * S, depth=K, gamma=GAMMA)                                    # copy j <- root j mod R, each with its own streams
first = torch.arange(A, dtype=torch.int32, device="cuda").repeat_interleave(S)[:, None].expand(A * S, R).contiguous()   # simulation block a tries action a
sim.run(seed=0, first_actions=first)                                                       # (untimed: the copies' specialised unit is compiled on first use)
torch.cuda.synchronize()
t0, steps = time.perf_counter(), 0
for decision in range(100):
    out = sim.run(seed=decision, first_actions=first)                                      # one nsg_fork + one step + one nsg_rollout_policy
    q = out["first_reward"].double() + GAMMA * torch.where(out["first_done"], torch.zeros_like(out["ret"]), out["ret"])   # [A * S, R]
    env.step(q.view(A, S, R).mean(1).argmax(0).to(torch.int32))
    steps += (K + 1) * A * S * R
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ret, length = env.episode_returns()
done = int((length > 0).sum())
tail = (f"mean length of the last finished episode {length[length > 0].float().mean().item():.0f}" if done
        else "every root is still in its first episode after 100 decisions")
print(f"{R} roots x 100 decisions: {steps / dt / 1e9:.1f} G simulated env-steps/s; episodes finished so far: {done}, {tail} "
      f"(random policy: ~22 steps per episode)")
sim.close(); env.close()
