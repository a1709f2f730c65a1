"""Single-env adaptors with the reference's class names and call signatures.

`NSClassicControlWrapper` / `NSFrozenLakeWrapper` wrap a `VecNSEnv` with N = 1 and return
exactly what the reference wrappers return for one env: the NS observation dict with Python /
NumPy scalars, a float (or `Reward`) reward, Python bools, and the info dict
(ns_gym/wrappers/classic_control.py:60-109, ns_gym/wrappers/toy_text.py:342-399,
ns_gym/base.py:296-410).  They exist so that code written against the reference runs
unchanged (BASELINE config C1); throughput comes from `VecNSEnv` with large N.

Every value comes from the HIP kernels; reading it back synchronises the stream.
"""
from __future__ import annotations

import warnings
from typing import Any, Union

import numpy as np
import torch

from .base import Reward
from .envs import BaseEnvSpec, from_gym_env
from .vec_env import ConstraintViolationWarning, VecNSEnv

__all__ = ["NSClassicControlWrapper", "NSFrozenLakeWrapper", "NSCliffWalkingWrapper", "NSBridgeWrapper",
           "ConstraintViolationWarning"]


class _Unwrapped:
    """`env.unwrapped.<attr>` of the reference: live view of the base env's attributes."""

    def __init__(self, owner: "_NSSingle"):
        import weakref

        # weak: wrapper <-> view must not form a cycle, or a dropped planning copy would wait for the cycle collector
        # before its device handle is recycled (_NSSingle._copy)
        object.__setattr__(self, "_ref", weakref.ref(owner))

    @property
    def _o(self):
        o = self._ref()
        if o is None:
            raise ReferenceError("the wrapper this `unwrapped` view belongs to no longer exists")
        return o

    def __getattr__(self, name):
        o = self._o
        et = o.spec.env_type
        if name == "__class__":
            return type(self)
        if name == "get_planning_env":      # the wrapper installs itself on the base env (base.py:294): planners call it there
            return o.get_planning_env
        if name in et.theta_names and not o._vec.is_frozenlake:
            if name in o._vec.param_names:
                return float(o._vec.theta[o._vec.param_names.index(name), 0].item())
            return float(o._vec.cfg.base_theta[et.theta_names.index(name)])
        if et.class_name == "CartPoleEnv" and name == "total_mass":
            return self.masspole + self.masscart            # classic_control.py:426-435
        if et.class_name == "CartPoleEnv" and name == "polemass_length":
            return self.length * self.masspole              # classic_control.py:436-444
        if name == "state" and not o._vec.is_frozenlake:
            return o._vec.phys[:, 0].cpu().numpy().copy()
        if name == "s" and o._vec.is_frozenlake:
            return int(o._vec.state[0].item())
        if name in ("P", "P_left", "P_right") and o.spec.class_name == "Bridge":
            return o._dist(name)
        if name == "split_probs" and o.spec.class_name == "Bridge":
            return o._split_mode                           # envs/Bridge.py:149-158
        if name == "P" and o._vec.is_frozenlake:
            return o._build_P()
        if name in ("nrow", "ncol") and o._vec.is_frozenlake:
            return int(getattr(o._vec.cfg, name))
        if name == "desc" and o._vec.is_frozenlake:
            return np.asarray(o.spec.desc, dtype="c")
        raise AttributeError(name)

    @property
    def spec(self):
        return self._o.spec


class _NSSingle:
    """Shared N = 1 plumbing (NSWrapper surface, ns_gym/base.py:206-502)."""

    def __init__(self, env, tunable_params, change_notification=False, delta_change_notification=False,
                 in_sim_change=False, _vec=None, **kwargs: Any):
        self.spec: BaseEnvSpec = from_gym_env(env)
        self.env = env   # gymnasium.Wrapper's attribute: the env this wrapper was built around
        self._act = None
        self._host = None
        kwargs.setdefault("violation_mask", True)   # the per-step rejected-update row feeds the ConstraintViolationWarning
        # a single wrapper never resets on its own: step() after `done` goes on exactly like the reference's, which forwards to
        # gymnasium (base.py:313) - the reference's own tests step without looking at `done` (tests/test_step_reset.py:570-577)
        kwargs.setdefault("autoreset", False)
        self._vec = _vec if _vec is not None else VecNSEnv(
            self.spec, tunable_params, 1, change_notification=change_notification,
            delta_change_notification=delta_change_notification, in_sim_change=in_sim_change, **kwargs)
        v = self._vec
        self.tunable_params = tunable_params
        self.change_notification = change_notification
        self.delta_change_notification = delta_change_notification
        self.in_sim_change = in_sim_change
        self.scalar_reward = v.scalar_reward
        self.persistent_params = v.persistent_params
        self.unwrapped = _Unwrapped(self)
        self.delta_t = 1
        self.action_space = v.single_action_space
        self.observation_space = v.single_observation_space   # Dict(state, env_change, delta_change, relative_time)
        try:   # with gymnasium installed the spaces ARE gymnasium's, so the reference's planners' isinstance checks hold (MCTS.py:113)
            self.action_space = self.action_space.to_gymnasium()
            self.observation_space = self.observation_space.to_gymnasium()
        except ImportError:
            pass

    # state mirrored from the vector env
    t = property(lambda self: int(self._vec.t[0].item()))
    has_reset = property(lambda self: self._vec.has_reset)
    frozen = property(lambda self: self._vec.frozen)
    is_sim_env = property(lambda self: self._vec.is_sim_env)

    def _scalars(self):
        """The wrapper's scalar observation / info dicts (base.py:343-363) out of ONE read-back of the env's rows
        (`VecNSEnv.host_rows`); the snapshot stays in `self._host` for the subclass's extra info fields."""
        v = self._vec
        h = self._host = v.host_rows()
        names = v.param_names
        st = int(h["cell"][0]) if v.is_grid else h["obs"][:v.obs_dim].copy()
        gt_ec = {p: int(h["env_change"][j]) for j, p in enumerate(names)}
        gt_dc = {p: float(h["delta_change"][j]) for j, p in enumerate(names)}
        hide = v.frozen or (v.is_sim_env and not v.in_sim_change)   # VecNSEnv._masked (base.py:316-321)
        show_ec, show_dc = v.change_notification and not hide, v.delta_change_notification and not hide
        out = {
            "state": st,
            "env_change": gt_ec if show_ec else {p: 0 for p in names},
            "delta_change": gt_dc if show_dc else {p: 0.0 for p in names},
            "relative_time": int(h["t"][0]),
        }
        inf = {"Ground Truth Env Change": dict(gt_ec), "Ground Truth Delta Change": dict(gt_dc)}
        return out, inf

    def reset(self, *, seed: int | None = None, options: dict | None = None):
        self._vec.reset(seed=None if seed is None else [int(seed)], options=options)
        o, inf = self._scalars()
        # the reference's reset info carries the zero dicts (base.py:397-408)
        return o, inf

    def _step(self, action):
        v = self._vec
        if self._act is None:
            # one pinned host word: the kernel reads the action straight out of it (host memory pinned through HIP is mapped into
            # the device's address space at the same address) - no fill kernel, no host-to-device copy, one launch per step
            self._act = torch.zeros(1, dtype=torch.float32 if v.action_is_float else torch.int32).pin_memory()
            self._act_np = self._act.numpy()
        self._act_np[0] = float(np.asarray(action, dtype=np.float64).reshape(-1)[0]) if v.action_is_float else int(action)
        v._step_raw(self._act.data_ptr())
        o, inf = self._scalars()
        if v.may_raise:      # what the reference raises inside step() (LCBounded exhaustion, see VecNSEnv.check_errors)
            v.check_errors()
        h = self._host
        r = float(h["reward"][0])
        terminated, truncated = bool(h["terminated"][0]), bool(h["truncated"][0])
        if not v.is_grid:
            if "violation" not in h:
                v.check_constraints()
            new = int(h["violation"].sum()) if "violation" in h else 0   # this step's rejected updates (violation_mask row)
            if new:  # aggregated ConstraintViolationWarning (classic_control.py:212-234), same text as VecNSEnv.check_constraints
                v._viol_seen += new
                warnings.warn(f"{new} parameter updates violated a physical constraint and were not applied", ConstraintViolationWarning)
        elif bool((h["theta"] < 0).any()):
            # The reference's step() raises out of SciPy's W1 as soon as an update function hands back a pmf with a
            # negative weight (base.py:192-203, utils.py:87-94) - DistributionIncrementUpdate has no lower clamp, so a
            # negative k gets there (documented in the reference's tests/test_gridworld_wrappers.py:192-199).
            raise ValueError("All weights must be non-negative.")
        if not self.scalar_reward:
            r = Reward(reward=r, env_change=o["env_change"], delta_change=o["delta_change"],
                       relative_time=o["relative_time"])
        return o, r, terminated, truncated, inf

    def freeze(self, mode: bool = True):
        self._vec.freeze(mode)
        return self

    def unfreeze(self):
        return self.freeze(False)

    def get_default_params(self):
        return self._vec.get_default_params()

    def _wrap(self, vec):
        import copy as _copy

        new = _copy.copy(self)            # shallow: shares the descriptor objects, not device state
        new._vec = vec
        new._act = new._host = None
        new.unwrapped = _Unwrapped(new)
        return new

    # Planners in the reference's style deep-copy the env once per simulation and drop the copy (MCTS.py:131,162-181).  A
    # dropped copy's device handle goes back to a small pool on the root env and the next copy overwrites it in place
    # (`fork(into=)`: one launch) instead of paying allocation + handle creation + destruction every time.
    _POOL_MAX = 8

    def _copy(self, theta_mode):
        import weakref

        src = self._vec
        root = src._fork_root()
        pool = root.__dict__.setdefault("_copy_pool", [])
        vec = pool.pop() if pool else None
        if vec is not None:
            src.fork(theta_mode=theta_mode, into=vec)
        else:
            vec = src.fork(theta_mode=theta_mode)
        new = self._wrap(vec)
        new._recycle = weakref.finalize(new, _NSSingle._give_back, weakref.ref(root), vec)
        return new

    @staticmethod
    def _give_back(root_ref, vec):
        try:
            root = root_ref()
            pool = root.__dict__.get("_copy_pool") if root is not None and getattr(root, "_h", None) else None
            if pool is not None and len(pool) < _NSSingle._POOL_MAX and getattr(vec, "_h", None):
                pool.append(vec)
            else:
                vec.close()
        except Exception:   # interpreter shutdown: the runtime underneath may already be gone
            pass

    def get_planning_env(self):
        """Planning copy (classic_control.py:120-136 / toy_text.py:471-481): the current θ if the agent is told the deltas
        (or this already is a copy), otherwise the initial θ."""
        assert self.has_reset, "The environment must be reset before getting the planning environment."
        v = self._vec
        return self._copy(0 if (v.is_sim_env or v.delta_change_notification) else 1)

    def __deepcopy__(self, memo):
        """`deepcopy(env)` sets is_sim_env on the copy (classic_control.py:138-186)."""
        return self._copy(0)

    def close(self):
        fin = getattr(self, "_recycle", None)
        if fin is not None:    # a copy: its handle returns to the root's pool (or is destroyed if the pool is full), once
            if fin.alive:
                fin()
                self._vec = None   # the handle may already serve another copy: this wrapper must not touch it again
            return
        for vec in self._vec.__dict__.pop("_copy_pool", []):
            vec.close()
        self._vec.close()


class NSClassicControlWrapper(_NSSingle):
    """Non-stationary wrapper for the classic-control envs
    (ns_gym/wrappers/classic_control.py:15-109), N = 1 view of the fused HIP stepper."""

    def __init__(self, env, tunable_params, change_notification: bool = False,
                 delta_change_notification: bool = False, in_sim_change: bool = False, **kwargs: Any):
        spec = from_gym_env(env)
        from .envs import TUNABLE_PARAMS

        assert spec.class_name in TUNABLE_PARAMS.keys() and spec.class_name not in ("FrozenLakeEnv", "CliffWalkingEnv", "Bridge"), (
            f"{spec.class_name} is not a supported environment"
        )
        for key in tunable_params.keys():
            assert key in TUNABLE_PARAMS[spec.class_name].keys(), (
                f"{key} is not a tunable parameter for {spec.class_name}"
            )
        super().__init__(spec, tunable_params, change_notification, delta_change_notification, in_sim_change, **kwargs)
        self.initial_params = {k: getattr(self.unwrapped, k) for k in tunable_params.keys()}

    def step(self, action: Union[float, int]):
        obs, reward, terminated, truncated, info = self._step(action)
        info["prob"] = 1.0  # classic_control.py:98
        return obs, reward, terminated, truncated, info


class NSFrozenLakeWrapper(_NSSingle):
    """Non-stationary FrozenLake wrapper (ns_gym/wrappers/toy_text.py:265-399), N = 1 view."""

    def __init__(self, env, tunable_params, change_notification: bool = False,
                 delta_change_notification: bool = False, in_sim_change: bool = False,
                 initial_prob_dist=[1, 0, 0], modified_rewards: Union[dict, None] = None, **kwargs: Any):
        spec = from_gym_env(env)
        assert spec.class_name == "FrozenLakeEnv", f"{spec.class_name} is not a FrozenLake environment"
        super().__init__(spec, tunable_params, change_notification, delta_change_notification, in_sim_change,
                         initial_prob_dist=initial_prob_dist, modified_rewards=modified_rewards, **kwargs)
        self.initial_prob_dist = initial_prob_dist
        self.modified_rewards = modified_rewards
        self.ncol, self.nrow = int(self._vec.cfg.ncol), int(self._vec.cfg.nrow)
        self.nA, self.nS = 4, self.ncol * self.nrow
        self.LEFT, self.DOWN, self.RIGHT, self.UP = 0, 1, 2, 3

    @property
    def transition_prob(self):
        return [float(x) for x in self._vec.theta[:, 0].tolist()]

    def step(self, action: int):
        obs, reward, terminated, truncated, info = self._step(action)
        info["prob"] = float(self._host["prob"][0])
        info["transition_prob"] = [float(x) for x in self._host["theta"]]  # toy_text.py:379
        return obs, reward, terminated, truncated, info

    def reset(self, *, seed: int | None = None, options: dict | None = None):
        obs, info = super().reset(seed=seed, options=options)
        info["prob"] = 1
        return obs, info

    def _build_P(self):
        """The transition table the wrapper installs on the base env (toy_text.py:426-469),
        rebuilt on demand from the device-side table probabilities."""
        tp = [float(x) for x in self._vec.table_prob[:, 0].tolist()]
        desc = self.spec.desc
        P = {s: {a: [] for a in range(4)} for s in range(self.nS)}

        def inc(row, col, a):
            if a == 0:
                col = max(col - 1, 0)
            elif a == 1:
                row = min(row + 1, self.nrow - 1)
            elif a == 2:
                col = min(col + 1, self.ncol - 1)
            elif a == 3:
                row = max(row - 1, 0)
            return row, col

        for row in range(self.nrow):
            for col in range(self.ncol):
                s = row * self.ncol + col
                for a in range(4):
                    if desc[row][col] in "GH":
                        P[s][a].append((1.0, s, 0, True))
                        continue
                    for ind, b in enumerate([a, (a + 1) % 4, (a - 1) % 4]):
                        nr, nc = inc(row, col, b)
                        letter = desc[nr][nc]
                        rew = float(self.modified_rewards[letter]) if self.modified_rewards else float(letter == "G")
                        P[s][a].append((tp[ind], nr * self.ncol + nc, rew, letter in "GH"))
        return P


class NSCliffWalkingWrapper(_NSSingle):
    """Non-stationary CliffWalking wrapper (ns_gym/wrappers/toy_text.py:14-262), N = 1 view:
    4-way slip [a, a+1, a-1, a+2], cliff teleport, `modified_rewards`, `terminal_cliff`."""

    def __init__(self, env, tunable_params, change_notification: bool = False,
                 delta_change_notification: bool = False, in_sim_change: bool = False,
                 initial_prob_dist=[1, 0, 0, 0], modified_rewards: Union[dict, None] = None,
                 terminal_cliff: bool = False, **kwargs: Any):
        spec = from_gym_env(env)
        assert spec.class_name == "CliffWalkingEnv", f"{spec.class_name} is not a CliffWalking environment"
        super().__init__(spec, tunable_params, change_notification, delta_change_notification, in_sim_change,
                         initial_prob_dist=initial_prob_dist, modified_rewards=modified_rewards,
                         terminal_cliff=terminal_cliff, **kwargs)
        self.initial_prob_dist = initial_prob_dist
        self.modified_rewards = modified_rewards or {"H": -100, "G": 0, "F": -1, "S": -1}
        self.terminal_cliff = terminal_cliff
        self.shape = (int(self._vec.cfg.nrow), int(self._vec.cfg.ncol))
        self.nS, self.nA = self.shape[0] * self.shape[1], 4
        self.start_state_index = (self.shape[0] - 1) * self.shape[1]

    @property
    def transition_prob(self):
        return [float(x) for x in self._vec.theta[:, 0].tolist()]

    def step(self, action: int):
        obs, reward, terminated, truncated, info = self._step(action)
        info["prob"] = float(self._host["prob"][0])
        info["transition_prob"] = [float(x) for x in self._host["theta"]]  # toy_text.py:192
        return obs, reward, terminated, truncated, info

    def reset(self, *, seed: int | None = None, options: dict | None = None):
        obs, info = super().reset(seed=seed, options=options)
        info["prob"] = 1
        return obs, info

    def _build_P(self):
        """`unwrapped.P` of the reference (toy_text.py:86-148): {s: {a: [(prob, s', reward, terminated) x 4]}} in
        slip order [a, a+1, a-1, a+2], rebuilt on demand from the device-side table probabilities."""
        tp = [float(x) for x in self._vec.table_prob[:, 0].tolist()]
        nrow, ncol = self.shape
        delta = {0: (-1, 0), 1: (0, 1), 2: (1, 0), 3: (0, -1)}          # UP RIGHT DOWN LEFT (toy_text.py:74-76)
        mr = self.modified_rewards
        P = {}
        for s in range(self.nS):
            row, col = divmod(s, ncol)
            P[s] = {}
            for a in range(self.nA):
                entries = []
                for ind, b in enumerate([a, (a + 1) % 4, (a - 1) % 4, (a + 2) % 4]):
                    nr = min(max(row + delta[b][0], 0), nrow - 1)
                    nc = min(max(col + delta[b][1], 0), ncol - 1)
                    cliff = nr == nrow - 1 and 1 <= nc <= ncol - 2
                    goal = nr == nrow - 1 and nc == ncol - 1
                    reward = mr["H"] if cliff else mr["G"] if goal else mr["F"]
                    terminated = bool(self.terminal_cliff) if cliff else goal
                    entries.append((tp[ind], self.start_state_index if cliff else nr * ncol + nc, reward, terminated))
                P[s][a] = entries
        return P


class NSBridgeWrapper(_NSSingle):
    """Non-stationary Bridge wrapper (ns_gym/wrappers/toy_text.py:524-715), N = 1 view.  Uniform mode
    `{"P": fn}` or split mode `{"P_left": fn_l, "P_right": fn_r}` (either side may be omitted)."""

    def __init__(self, env, tunable_params, change_notification: bool = False,
                 delta_change_notification: bool = False, in_sim_change: bool = False,
                 initial_prob_dist=[1, 0, 0], modified_rewards: Union[dict, None] = None, **kwargs: Any):
        spec = from_gym_env(env)
        assert spec.class_name == "Bridge", f"{spec.class_name} is not the Bridge environment"
        super().__init__(spec, tunable_params, change_notification, delta_change_notification, in_sim_change,
                         initial_prob_dist=initial_prob_dist, **kwargs)
        self._split_mode = ("P_left" in tunable_params) or ("P_right" in tunable_params)
        self.initial_prob_dist = initial_prob_dist

    def _dist(self, name, host=None):
        v = self._vec
        if name in v.param_names:
            j = v.param_names.index(name)
            if host is not None:   # this step's read-back (N = 1: row k of theta is element k)
                return [float(x) for x in host["theta"][3 * j:3 * j + 3]]
            return [float(x) for x in v.theta[3 * j:3 * j + 3, 0].tolist()]
        side = 1 if name == "P_right" else 0
        return [float(v.cfg.initial_prob[side][k]) for k in range(3)]

    def step(self, action: int):
        obs, reward, terminated, truncated, info = self._step(action)
        reward = int(reward) if self.scalar_reward else reward   # Bridge returns int rewards (envs/Bridge.py:101)
        col = obs["state"] % int(self._vec.cfg.ncol)
        info["prob"] = (self._dist("P_left" if col < int(self._vec.cfg.ncol) // 2 else "P_right", self._host)
                        if self._split_mode else self._dist("P", self._host))
        return obs, reward, terminated, truncated, info
