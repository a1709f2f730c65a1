"""Monte-Carlo simulations for planners, batched: the part of the reference's MCTS that steps the environment
(ns_gym/benchmark_algorithms/MCTS.py) - one deep copy of the planning env per simulation (`search`, :131), the chance node's step
(`_expand`, :204), and the default policy's rollout (`_default_policy`, :162-181: uniformly random actions,
`tot_reward += reward * gamma ** depth` while not terminated, not truncated and depth < d) - for R root envs x S simulations as
TWO launches per decision: `nsg_fork` into a standing batch of S * R copies, and one fused `nsg_rollout_policy` that draws the
actions in the kernel and keeps only the discounted returns.  The search tree itself (selection, backup) is the agent's business
and stays where the caller keeps it.

    sim = Simulator(planning_env, sims=100, depth=50, gamma=0.99)          # README quickstart: d = 50, m = 100
    out = sim.run(seed=k, first_actions=a)                                 # a: [S, R] actions of the expanded chance nodes, or None
    q = out["first_reward"] + gamma * out["ret"]  ...                      # whatever the planner backs up

What the copies inherit follows the reference's `__deepcopy__` of a planning env - including, for FrozenLake, that a copy of a
copy steps with the table of `initial_prob_dist` (include/nsgym_hip.h: nsg_fork)."""
from __future__ import annotations

from typing import Optional

import torch

from .policies import EpisodeAccounts, Policy, UniformRandom


class Simulator:
    """A standing batch of `sims` copies of every env of `plan_env` (copy j <- env j mod R) and their accounts."""

    def __init__(self, plan_env, sims: int, depth: int, gamma: float, policy: Optional[Policy] = None):
        self.src, self.sims, self.depth, self.gamma = plan_env, int(sims), int(depth), float(gamma)
        self.copies = plan_env.fork(theta_mode=0, repeat=self.sims)            # deepcopy(self.env) per simulation (MCTS.py:131)
        self.acc = EpisodeAccounts(self.copies, gamma=self.gamma, horizon=self.depth + 1)
        self.policy = policy
        self.R = plan_env.num_envs

    def run(self, seed: int = 0, first_actions: Optional[torch.Tensor] = None, entropy: Optional[int] = None) -> dict:
        """One decision's simulations.  Returns [S, R] tensors: `ret` (the default policy's discounted return; 0 where the expanded
        node was terminal - the reference returns that node's own reward there, `first_reward`), `length`, and with
        `first_actions` the chance-node step's `first_reward` / `first_done`."""
        c, S, R = self.copies, self.sims, self.R
        self.src.fork(theta_mode=0, into=c, entropy=entropy)
        out = {}
        alive = None
        if first_actions is not None:                                          # _expand(chance node): sim_env.step(action) (MCTS.py:204)
            _, r, term, trunc, _ = c.step(first_actions.reshape(-1))
            done = term | trunc
            out["first_reward"], out["first_done"] = r.view(S, R).clone(), done.view(S, R).clone()
            alive = ~done
        else:                                                                  # a copy of an env whose episode is over has nothing to simulate
            alive = ~(c.buf["terminated"].bool() | c.buf["truncated"].bool())
        self.acc.restart(alive=alive)
        pol = self.policy if self.policy is not None else UniformRandom(seed=seed)
        c.rollout_policy(pol, self.depth, accounts=self.acc)                   # _default_policy (MCTS.py:162-181), all S * R at once
        out["ret"], out["length"] = self.acc.ret.view(S, R), self.acc.length.view(S, R)
        return out

    def close(self):
        self.copies.close()


__all__ = ["Simulator"]
