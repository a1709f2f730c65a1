"""Minimal observation/action space descriptors (gymnasium is optional at run time).  They carry
what the reference's wrappers expose through `gym.spaces` (ns_gym/base.py:275-292): shapes,
dtypes, bounds, `n`, and a seeded `sample()`.  When gymnasium is importable, `to_gymnasium()`
converts them."""
from __future__ import annotations

import numpy as np


class Space:
    def __init__(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)


class Discrete(Space):
    def __init__(self, n, seed=None):
        super().__init__(seed)
        self.n, self.shape, self.dtype = int(n), (), np.int64

    def sample(self):
        return int(self._rng.integers(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n

    def to_gymnasium(self):
        import gymnasium as gym

        return gym.spaces.Discrete(self.n)

    def __repr__(self):
        return f"Discrete({self.n})"


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        super().__init__(seed)
        self.dtype = np.dtype(dtype)
        self.shape = tuple(np.shape(low)) if shape is None else tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), self.shape).astype(self.dtype)
        self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), self.shape).astype(self.dtype)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi, size=self.shape).astype(self.dtype)

    def to_gymnasium(self):
        import gymnasium as gym

        return gym.spaces.Box(self.low, self.high, shape=self.shape, dtype=self.dtype.type)

    def __repr__(self):
        return f"Box({self.shape}, {self.dtype})"


class Dict(Space):
    def __init__(self, spaces):
        super().__init__()
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def keys(self):
        return self.spaces.keys()

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}

    def to_gymnasium(self):
        import gymnasium as gym

        return gym.spaces.Dict({k: s.to_gymnasium() for k, s in self.spaces.items()})

    def __repr__(self):
        return "Dict(" + ", ".join(f"{k}: {v!r}" for k, v in self.spaces.items()) + ")"


_F32MAX = float(np.finfo(np.float32).max)


def base_spaces(class_name: str, desc=None):
    """(observation_space, action_space) of the base MDP [UPSTREAM gymnasium 1.2.1]."""
    if class_name == "CartPoleEnv":
        th = 12 * 2 * np.pi / 360
        high = np.array([4.8, np.inf, th * 2, np.inf], dtype=np.float32)
        return Box(-high, high), Discrete(2)
    if class_name == "PendulumEnv":
        high = np.array([1.0, 1.0, 8.0], dtype=np.float32)
        return Box(-high, high), Box(-2.0, 2.0, shape=(1,))
    if class_name == "AcrobotEnv":
        high = np.array([1.0, 1.0, 1.0, 1.0, 4 * np.pi, 9 * np.pi], dtype=np.float32)
        return Box(-high, high), Discrete(3)
    if class_name == "MountainCarEnv":
        return Box(np.array([-1.2, -0.07], dtype=np.float32), np.array([0.6, 0.07], dtype=np.float32)), Discrete(3)
    if class_name == "Continuous_MountainCarEnv":
        return (Box(np.array([-1.2, -0.07], dtype=np.float32), np.array([0.6, 0.07], dtype=np.float32)),
                Box(-1.0, 1.0, shape=(1,)))
    n = len(desc) * len(desc[0])
    return Discrete(n), Discrete(4)


def ns_observation_space(state_space, param_names):
    """The NS observation Dict of NSWrapper (ns_gym/base.py:275-292)."""
    return Dict({
        "state": state_space,
        "env_change": Dict({p: Discrete(2) for p in param_names}),
        "delta_change": Dict({p: Box(-np.inf, np.inf, shape=()) for p in param_names}),
        "relative_time": Box(0, np.inf, shape=()),
    })
