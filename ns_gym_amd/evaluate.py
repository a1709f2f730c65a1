"""Episode harness with the reference's result format (ns_gym/evaluate/run_experiment.py:91-148,
206-217), batched: one episode per env instance, all stepped by the fused kernel.

The reference runs `run_episode(env, agent, seed, ...)` in a `multiprocessing.Pool` (one process per
episode) and streams rows `[total_reward, SARNS, num_steps, seed, sample_id, time]` to a CSV.  Here
the N episodes of a batch run in lock-step on the GPU; a finished env is masked out (the kernel's
next-step autoreset keeps it stepping, its later steps are ignored), and the rows are written in the
same column order with the same header.
"""
from __future__ import annotations

import csv
import time
from typing import Callable, Optional

import numpy as np
import torch

from .utils import type_mismatch_checker  # noqa: F401  (run_experiment.py imports it next to the harness)

CSV_HEADER = ["total_reward", "State-Action-Reward-NextState", "num_steps", "seed", "sample_id", "time"]


def random_policy(env) -> Callable:
    """Uniform random actions generated on the device (the rollout policy of MCTS.py:162-181)."""
    if env.action_is_float:
        lo, hi = env.spec.env_type.action_low, env.spec.env_type.action_high
        return lambda obs: torch.rand(env.num_envs, device=env.device) * (hi - lo) + lo
    return lambda obs: torch.randint(0, env.n_actions, (env.num_envs,), dtype=torch.int32, device=env.device)


def run_episodes(env, policy: Optional[Callable] = None, seed: int = 0, max_steps: Optional[int] = None,
                 record_sarns: bool = False, sample_id=None) -> list:
    """One episode per env of `env` (a VecNSEnv).  `policy(state_tensor) -> action tensor [N]`.
    Returns rows `[total_reward, SARNS, num_steps, seed, sample_id, time]` (run_experiment.py:133-141);
    SARNS is a list of (state, action, reward, next_state) tuples when `record_sarns`, else []."""
    n = env.num_envs
    policy = policy or random_policy(env)
    limit = max_steps if max_steps is not None else (env.cfg.max_episode_steps or 10_000)
    obs, _ = env.reset(seed=seed)
    state = obs["state"].clone()
    alive = torch.ones(n, dtype=torch.bool, device=env.device)
    total = torch.zeros(n, dtype=torch.float64, device=env.device)
    steps = torch.zeros(n, dtype=torch.int64, device=env.device)
    traj = []
    t0 = time.time()
    for _ in range(limit + 1):     # the reference breaks at max_steps + 1 (run_experiment.py:127-129)
        a = policy(state)
        obs, r, term, trunc, _info = env.step(a)
        nxt = obs["state"]
        total += torch.where(alive, r.to(torch.float64), torch.zeros_like(total))
        steps += alive.to(torch.int64)
        if record_sarns:
            traj.append((state.cpu().numpy(), torch.as_tensor(a).cpu().numpy(), r.cpu().numpy().copy(),
                         nxt.cpu().numpy().copy(), alive.cpu().numpy().copy()))
        alive = alive & ~(term | trunc)
        state = nxt.clone()
        if not bool(alive.any()):
            break
    wall = time.time() - t0
    total, steps = total.cpu().numpy(), steps.cpu().numpy()
    seeds = (np.arange(n) + int(seed)) if np.isscalar(seed) else np.asarray(seed)
    ids = list(range(n)) if sample_id is None else list(sample_id)
    rows = []
    for i in range(n):
        sarns = []
        if record_sarns:
            for s, a, r, s2, al in traj:
                if al[i]:
                    sarns.append((np.asarray(s[i]).tolist(), np.asarray(a[i]).tolist(), float(r[i]), np.asarray(s2[i]).tolist()))
        rows.append([float(total[i]), sarns, int(steps[i]), int(seeds[i]), ids[i], wall])
    return rows


def write_results_csv(path: str, rows: list) -> None:
    """Same header and column order as the reference's results file (run_experiment.py:206-217)."""
    with open(path, mode="w", newline="") as f:
        w = csv.writer(f)
        w.writerow(CSV_HEADER)
        w.writerows(rows)
