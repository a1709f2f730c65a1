"""Policies that a fused rollout evaluates INSIDE the launch (`nsg_rollout_policy`, include/nsgym_hip.h), and the per-env episode
accounts such a rollout keeps in registers.

The reference's step consumers are Python loops `action = policy(observation); observation, reward, ... = env.step(action)`:
MCTS._default_policy (ns_gym/benchmark_algorithms/MCTS.py:162-181: uniformly random actions, `tot_reward += reward * gamma ** depth`),
run_episode (ns_gym/evaluate/run_experiment.py:108-129), the tutorial's run_episode with a tabular policy (tutorial.ipynb cell 12:
`action = policy[observation]`).  A policy that looks only at its own env's observation does not have to leave the kernel: the
objects below describe such policies; `VecNSEnv.rollout_policy(policy, K, ...)` runs K closed-loop steps of every env in ONE launch.

Each class also evaluates itself on the host / with torch (`__call__`, `actions`) - bit for bit what the kernel computes - so that
the same object drives `env.step()` loops, the oracle and the tests.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _abi as A

_M64 = (1 << 64) - 1


def _mix(x: np.ndarray) -> np.ndarray:
    x = x ^ (x >> np.uint64(30))
    x = x * np.uint64(0xBF58476D1CE4E5B9)
    x = x ^ (x >> np.uint64(27))
    x = x * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def policy_bits(seed: int, env_index, step) -> np.ndarray:
    """`nsg_policy_bits` vectorised: two rounds of the splitmix64 finaliser over (key, env, step).  A pure function of its three
    arguments - any sharding of a batch and any chunking of the steps draw the same actions."""
    with np.errstate(over="ignore"):
        e = np.asarray(env_index, dtype=np.uint64)
        k = np.asarray(step, dtype=np.uint64)
        a = _mix(np.uint64(seed & _M64) + np.uint64(0x9E3779B97F4A7C15) * (e + np.uint64(1)))
        return _mix(a + np.uint64(0xD1B54A32D192ED03) * (k + np.uint64(1)))


def _as_i64(c: int) -> int:
    return c - (1 << 64) if c >= (1 << 63) else c


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    """Logical right shift of int64 bit patterns (torch shifts arithmetically)."""
    return (x >> s) & ((1 << (64 - s)) - 1)


def policy_bits_torch(seed: int, env_index: torch.Tensor, step: torch.Tensor) -> torch.Tensor:
    """`policy_bits` with torch int64 arithmetic (wrap-around products are the unsigned ones' bit patterns): the same 64 bits, as
    int64, on whatever device the index tensors live on - action tables for 2^20 envs without a host round trip."""
    def mix(x):
        x = x ^ _lsr(x, 30)
        x = x * _as_i64(0xBF58476D1CE4E5B9)
        x = x ^ _lsr(x, 27)
        x = x * _as_i64(0x94D049BB133111EB)
        return x ^ _lsr(x, 31)

    a = mix(_as_i64(seed & _M64) + _as_i64(0x9E3779B97F4A7C15) * (env_index.to(torch.int64) + 1))
    return mix(a + _as_i64(0xD1B54A32D192ED03) * (step.to(torch.int64) + 1))


class Policy:
    """Base of the in-kernel policies.  `kind` is the `nsg_policy.kind`; `_data(env)` the device tensor the kernel reads."""
    kind = -1

    def _data(self, env) -> Optional[torch.Tensor]:
        return None

    def _n_data(self, env) -> int:
        return 0

    def _struct(self, env, step0: int, actions_out) -> A.Policy:
        d = self._data(env)
        return A.Policy(kind=self.kind, step0=int(step0), seed=getattr(self, "seed", 0) & _M64, index0=int(getattr(self, "index0", 0)),
                        data=d.data_ptr() if d is not None else None, n_data=self._n_data(env), reserved0=0,
                        actions_out=actions_out.data_ptr() if actions_out is not None else None)

    def actions(self, env, step0: int, k: int, state=None) -> torch.Tensor:
        raise NotImplementedError


class UniformRandom(Policy):
    """Uniformly random actions - the rollout policy of MCTS._default_policy (`np.random.choice(self.possible_actions)`,
    MCTS.py:176), of the reference's tests (`action_space.sample()`) and of BASELINE's "random-action rollouts".  The reference
    draws them from unseeded global generators; here step k of env i is `policy_bits(seed, index0 + i, k)`: discrete
    `((bits >> 32) * n_actions) >> 32`, continuous `low + (high - low) * float32(bits >> 40) * 2^-24` in float32.  Open loop: the
    whole table can be produced ahead (`actions`), which is how the fused path is checked against `VecNSEnv.rollout`."""
    kind = A.NSG_POL_UNIFORM

    def __init__(self, seed: int = 0, index0: int = 0):
        self.seed, self.index0 = int(seed), int(index0)

    def table(self, env, step0: int, k: int) -> np.ndarray:
        """The [k, N] actions of steps step0 .. step0 + k - 1 (NumPy, host)."""
        n = env.num_envs
        bits = policy_bits(self.seed, self.index0 + np.arange(n, dtype=np.uint64)[None, :], (step0 + np.arange(k, dtype=np.uint64))[:, None])
        if env.action_is_float:
            lo, hi = np.float32(env.spec.env_type.action_low), np.float32(env.spec.env_type.action_high)
            u = (bits >> np.uint64(40)).astype(np.float32) * np.float32(5.9604644775390625e-08)
            return (lo + (hi - lo) * u).astype(np.float32)
        return (((bits >> np.uint64(32)) * np.uint64(env.n_actions)) >> np.uint64(32)).astype(np.int32)

    def actions(self, env, step0: int, k: int, state=None) -> torch.Tensor:
        """The same [k, N] table computed on the env's device."""
        dev = env.device
        bits = policy_bits_torch(self.seed, (self.index0 + torch.arange(env.num_envs, device=dev))[None, :],
                                 (step0 + torch.arange(k, device=dev))[:, None])
        if env.action_is_float:
            lo, hi = float(env.spec.env_type.action_low), float(env.spec.env_type.action_high)
            u = _lsr(bits, 40).to(torch.float32) * 5.9604644775390625e-08
            return torch.tensor(lo, dtype=torch.float32, device=dev) + torch.tensor(hi - lo, dtype=torch.float32, device=dev) * u
        return ((_lsr(bits, 32) * env.n_actions) >> 32).to(torch.int32)


class TabularPolicy(Policy):
    """`action = policy[observation]` for the grid envs (tutorial.ipynb cell 12; the value-iteration policies of cell 6)."""
    kind = A.NSG_POL_BY_STATE

    def __init__(self, table):
        self.table = np.asarray(table, dtype=np.int32).reshape(-1).copy()
        self._dev = {}

    def _dev_table(self, device):
        t = self._dev.get(device)
        if t is None:
            t = self._dev[device] = torch.from_numpy(self.table).to(device)
        return t

    def _data(self, env):
        return self._dev_table(env.device)

    def _n_data(self, env):
        return int(self.table.size)

    def __call__(self, state: torch.Tensor) -> torch.Tensor:
        """The same look-up as a torch gather (for `step()` loops): state [N] int -> actions [N] int32."""
        return self._dev_table(state.device)[state.long()]

    def actions(self, env, step0, k, state=None):
        assert k == 1 and state is not None
        return self(state).to(torch.int32).reshape(1, -1)


class LinearPolicy(Policy):
    """A linear policy on the float32 observation of a classic-control env: `score_j = W[j, D] + sum_d W[j, d] * obs[d]` accumulated in
    float32 in that order; discrete action spaces take the first argmax over j (rows = n_actions), continuous ones
    `clip(score_0, low, high)` (one row).  The weights are uniform over the batch: the kernel reads them through scalar loads."""
    kind = A.NSG_POL_LINEAR

    def __init__(self, weights):
        self.W = np.asarray(weights, dtype=np.float32)
        assert self.W.ndim == 2, "weights: [rows, obs_dim + 1] (bias last)"
        self._dev = {}

    def _dev_weights(self, device):
        t = self._dev.get(device)
        if t is None:
            t = self._dev[device] = torch.from_numpy(self.W.copy()).to(device).contiguous()
        return t

    def _data(self, env):
        assert self.W.shape[1] == env.obs_dim + 1, f"weights need obs_dim + 1 = {env.obs_dim + 1} columns (bias last)"
        return self._dev_weights(env.device)

    def _n_data(self, env):
        return int(self.W.shape[0])

    def scores(self, obs: np.ndarray) -> np.ndarray:
        """[N, rows] float32 scores, rounded like the kernel (one float32 operation at a time)."""
        obs = np.asarray(obs, dtype=np.float32)
        d = self.W.shape[1] - 1
        out = np.empty((obs.shape[0], self.W.shape[0]), dtype=np.float32)
        for j in range(self.W.shape[0]):
            sc = np.full(obs.shape[0], self.W[j, d], dtype=np.float32)
            for q in range(d):
                sc = (sc + (self.W[j, q] * obs[:, q]).astype(np.float32)).astype(np.float32)
            out[:, j] = sc
        return out

    def decide(self, obs: np.ndarray, action_is_float: bool, low: float = 0.0, high: float = 0.0) -> np.ndarray:
        sc = self.scores(obs)
        if action_is_float:
            return np.clip(sc[:, 0], np.float32(low), np.float32(high)).astype(np.float32)
        return np.argmax(sc, axis=1).astype(np.int32)       # first maximum, like the kernel

    def __call__(self, state: torch.Tensor, action_is_float: bool = False, low: float = 0.0, high: float = 0.0) -> torch.Tensor:
        """The same decision with torch kernels (for `step()` loops): one float32 multiply and one float32 add per term, in the
        kernel's order (separate elementwise kernels: nothing is contracted into an FMA)."""
        W = self._dev_weights(state.device)
        d = W.shape[1] - 1
        sc = W[:, d].unsqueeze(0).expand(state.shape[0], -1).clone()
        for q in range(d):
            sc = sc + W[:, q].unsqueeze(0) * state[:, q:q + 1]
        if action_is_float:
            return sc[:, 0].clamp(low, high)
        return torch.argmax(sc, dim=1).to(torch.int32)       # (ties: the first maximum, like the kernel, for the two- and three-way cases here)


class EpisodeAccounts:
    """The per-env accounts a fused policy rollout keeps (`nsg_episode_acc`): `ret += reward64 * gamma ** length` and `length += 1` on
    every step taken while `alive`; `alive` is cleared by the step that returns terminated or truncated.  `discount[j] = gamma ** j`
    is computed HERE with Python's float power - the reference's own expression (`reward * self.gamma ** depth`, MCTS.py:179) - and
    only looked up by the kernel.  gamma None: plain sums (`total_reward += reward`, run_experiment.py:117)."""

    def __init__(self, env, gamma: Optional[float] = None, horizon: Optional[int] = None):
        n, dev = env.num_envs, env.device
        self.ret = torch.zeros(n, dtype=torch.float64, device=dev)
        self.length = torch.zeros(n, dtype=torch.int32, device=dev)
        self.alive = torch.ones(n, dtype=torch.uint8, device=dev)
        self.gamma = gamma
        self.discount = None
        if gamma is not None:
            h = int(horizon if horizon is not None else (env.cfg.max_episode_steps or 1000) + 1)
            self.discount = torch.tensor([float(gamma) ** j for j in range(h)], dtype=torch.float64, device=dev)

    def restart(self, alive: Optional[torch.Tensor] = None):
        self.ret.zero_()
        self.length.zero_()
        if alive is None:
            self.alive.fill_(1)
        else:
            self.alive.copy_(alive.to(torch.uint8))
        return self

    def _struct(self) -> A.EpisodeAcc:
        return A.EpisodeAcc(ret=self.ret.data_ptr(), length=self.length.data_ptr(), alive=self.alive.data_ptr(),
                            discount=self.discount.data_ptr() if self.discount is not None else None,
                            n_discount=int(self.discount.numel()) if self.discount is not None else 0, reserved0=0)


__all__ = ["Policy", "UniformRandom", "TabularPolicy", "LinearPolicy", "EpisodeAccounts", "policy_bits"]
