"""Base-MDP catalogue: what `gym.make(id)` + `TUNABLE_PARAMS` give the reference wrappers.

θ names / order follow ATTRIBUTE_MAP (ns_gym/base.py:611-635); defaults are the
construction-time attribute values TUNABLE_PARAMS holds (base.py:1156; documented at
docs/source/env_pages/classic_control/{cartpole,pendulum,acrobot,mountaincar}.md).
TimeLimit horizons are gymnasium 1.2.1's registrations [UPSTREAM].
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import _abi as A


@dataclass(frozen=True)
class EnvType:
    env_type: int
    class_name: str
    theta_names: tuple
    theta_defaults: tuple
    phys_dim: int
    obs_dim: int
    n_actions: int          # 0 = continuous (1-D Box)
    action_low: float = 0.0
    action_high: float = 0.0


ENV_TYPES = {
    "CartPoleEnv": EnvType(A.ENV_CARTPOLE, "CartPoleEnv",
                           ("gravity", "masscart", "masspole", "force_mag", "tau", "length"),
                           (9.8, 1.0, 0.1, 10.0, 0.02, 0.5), 4, 4, 2),
    "PendulumEnv": EnvType(A.ENV_PENDULUM, "PendulumEnv", ("m", "l", "dt", "g"),
                           (1.0, 1.0, 0.05, 10.0), 2, 3, 0, -2.0, 2.0),
    "AcrobotEnv": EnvType(A.ENV_ACROBOT, "AcrobotEnv",
                          ("dt", "LINK_LENGTH_1", "LINK_LENGTH_2", "LINK_MASS_1", "LINK_MASS_2",
                           "LINK_COM_POS_1", "LINK_COM_POS_2", "LINK_MOI"),
                          (0.2, 1.0, 1.0, 1.0, 1.0, 0.5, 0.5, 1.0), 4, 6, 3),
    "MountainCarEnv": EnvType(A.ENV_MOUNTAINCAR, "MountainCarEnv", ("gravity", "force"),
                              (0.0025, 0.001), 2, 2, 3),
    "Continuous_MountainCarEnv": EnvType(A.ENV_MOUNTAINCAR_CONT, "Continuous_MountainCarEnv", ("power",),
                                         (0.0015,), 2, 2, 0, -1.0, 1.0),
    "FrozenLakeEnv": EnvType(A.ENV_FROZENLAKE, "FrozenLakeEnv", ("P",), (), 0, 1, 4),
    "CliffWalkingEnv": EnvType(A.ENV_CLIFFWALKING, "CliffWalkingEnv", ("P",), (), 0, 1, 4),
    # in-tree env of the reference (ns_gym/envs/Bridge.py); TUNABLE_PARAMS["Bridge"] at ns_gym/base.py:1161
    "Bridge": EnvType(A.ENV_BRIDGE, "Bridge", ("P", "P_left", "P_right"), (), 0, 1, 4),
}
GRID_CLASSES = ("FrozenLakeEnv", "CliffWalkingEnv", "Bridge")

#: mirror of ns_gym.base.TUNABLE_PARAMS for the hot-path env classes
TUNABLE_PARAMS = {
    name: ({k: ([1.0, 0.0, 0.0] if name == "Bridge" else None) for k in et.theta_names} if name in GRID_CLASSES
           else dict(zip(et.theta_names, et.theta_defaults)))
    for name, et in ENV_TYPES.items()
}

BRIDGE_MAP = ["HHHHHHHH", "FFFFFHHH", "GFHFSFFG", "FFFFFHHH", "HHHHHHHH"]   # ns_gym/envs/Bridge.py:11

FROZENLAKE_MAPS = {
    "4x4": ["SFFF", "FHFH", "FFFH", "HFFG"],
    "8x8": ["SFFFFFFF", "FFFFFFFF", "FFFHFFFF", "FFFFFHFF", "FFFHFFFF", "FHHFFFHF", "FHFFHFHF", "FFFHFFFG"],
}

_REGISTRY = {
    "CartPole-v1": ("CartPoleEnv", 500),
    "Pendulum-v1": ("PendulumEnv", 200),
    "Acrobot-v1": ("AcrobotEnv", 500),
    "MountainCar-v0": ("MountainCarEnv", 200),
    "MountainCarContinuous-v0": ("Continuous_MountainCarEnv", 999),
    "FrozenLake-v1": ("FrozenLakeEnv", 100),
    "FrozenLake8x8-v1": ("FrozenLakeEnv", 200),
    "CliffWalking-v1": ("CliffWalkingEnv", None),       # no TimeLimit in gymnasium's registration [UPSTREAM]
    "ns_gym/Bridge-v0": ("Bridge", 100),                # ns_gym/__init__.py:17-21
}


@dataclass
class BaseEnvSpec:
    """What the NS wrappers need to know about the wrapped base env.  Stands where the
    reference takes a live `gym.make(...)` object; `from_gym_env` accepts one when
    gymnasium is installed."""

    env_id: str
    class_name: str
    max_episode_steps: int | None
    desc: list | None = None          # FrozenLake map rows
    theta_overrides: dict = field(default_factory=dict)

    @property
    def env_type(self) -> EnvType:
        return ENV_TYPES[self.class_name]

    @property
    def unwrapped(self):
        return self


# User-registered ids -> entry points: the role gymnasium's `register` / `registry` play for the reference's users, who
# register a factory that builds base env + NS wrapper and later `gym.make` it (tests/test_registration.py:78-117).
registry: dict = {}


def register(id: str, entry_point, **_gymnasium_options) -> None:
    """`gymnasium.envs.registration.register` counterpart: `make(id, **kw)` calls `entry_point(**kw)`.  gymnasium's own
    options (`disable_env_checker`, `order_enforce`, `max_episode_steps` ...) are accepted and have no effect here:
    the entry point returns the finished wrapper."""
    if not callable(entry_point):
        raise TypeError("register(): entry_point must be callable (string entry points need gymnasium's importer)")
    registry[id] = entry_point


def make(env_id: str, max_episode_steps: int | None = None, **kwargs):
    """`gym.make` counterpart for the supported ids (same ids, same TimeLimit horizons) and for ids added with `register`."""
    if env_id in registry:
        if max_episode_steps is not None:
            kwargs["max_episode_steps"] = max_episode_steps
        return registry[env_id](**kwargs)
    if env_id not in _REGISTRY:
        raise KeyError(f"{env_id} is not a supported environment; supported: {sorted(_REGISTRY)}")
    class_name, steps = _REGISTRY[env_id]
    desc = None
    if class_name == "FrozenLakeEnv":
        d = kwargs.pop("desc", None)
        map_name = kwargs.pop("map_name", "8x8" if env_id == "FrozenLake8x8-v1" else "4x4")
        kwargs.pop("is_slippery", None)   # the NS wrapper overwrites P entirely (toy_text.py:337-340)
        kwargs.pop("render_mode", None)
        desc = [str(r) for r in (d if d is not None else FROZENLAKE_MAPS[map_name])]
    elif class_name == "CliffWalkingEnv":
        kwargs.pop("render_mode", None)
        kwargs.pop("is_slippery", None)
        desc = ["F" * 12] * 3 + ["S" + "H" * 10 + "G"]   # shape (4, 12), cliff = row 3 cols 1..10 [UPSTREAM]
    elif class_name == "Bridge":
        kwargs.pop("render_mode", None)
        desc = list(BRIDGE_MAP)
    else:
        kwargs.pop("render_mode", None)
    if kwargs:
        raise TypeError(f"make({env_id!r}): unsupported keyword arguments {sorted(kwargs)}")
    return BaseEnvSpec(env_id, class_name, max_episode_steps if max_episode_steps is not None else steps, desc)


def _gym_desc(un, class_name):
    if class_name == "FrozenLakeEnv":
        return ["".join(chr(c[0]) if isinstance(c, (bytes, np.bytes_)) else str(c) for c in row) for row in un.desc]
    if class_name == "CliffWalkingEnv":
        nr, nc = un.shape
        return ["F" * nc] * (nr - 1) + ["S" + "H" * (nc - 2) + "G"]
    return ["".join(chr(c[0]) if isinstance(c, (bytes, np.bytes_)) else str(c) for c in row) for row in un.map]


def from_gym_env(env) -> BaseEnvSpec:
    """Describe a live gymnasium env (duck-typed; gymnasium itself is optional)."""
    if isinstance(env, BaseEnvSpec):
        return env
    if isinstance(env, str):
        return make(env)
    un = env.unwrapped
    class_name = un.__class__.__name__
    assert class_name in ENV_TYPES, f"{class_name} is not a supported environment"
    spec = getattr(env, "spec", None)
    steps = getattr(spec, "max_episode_steps", None) if spec is not None else None
    env_id = getattr(spec, "id", class_name) if spec is not None else class_name
    et = ENV_TYPES[class_name]
    desc = None
    overrides = {}
    if class_name in GRID_CLASSES:
        desc = _gym_desc(un, class_name)
    else:
        for name, default in zip(et.theta_names, et.theta_defaults):
            v = float(getattr(un, name, default))
            if v != default:
                overrides[name] = v
    return BaseEnvSpec(env_id, class_name, steps, desc, overrides)
