"""Compile (base env, tunable_params, wrapper flags) into the C-ABI config + constant tables.

Mirrors the validation of the reference constructors:
NSWrapper.__init__ (ns_gym/base.py:222-294), NSClassicControlWrapper.__init__
(ns_gym/wrappers/classic_control.py:27-58), NSFrozenLakeWrapper.__init__
(ns_gym/wrappers/toy_text.py:282-340).
"""
from __future__ import annotations

import copy
import math

import numpy as np

from . import _abi as A
from .base import Scheduler, TableBuilder, UpdateDistributionFn, UpdateFn
from .envs import BaseEnvSpec, from_gym_env

_INF = "inf"


def compile_config(env, tunable_params: dict, *, change_notification=False, delta_change_notification=False,
                   in_sim_change=False, scalar_reward=True, persistent_params=False, track_returns=False, initial_prob_dist=None, modified_rewards=None, is_sim_env=False):
    """Returns (Config, tables_blob: bytes, BaseEnvSpec, param_names)."""
    spec: BaseEnvSpec = from_gym_env(env)
    et = spec.env_type
    if delta_change_notification:
        assert change_notification, "If change_notification is True, delta_change_notification must be True"
    from .envs import TUNABLE_PARAMS

    allowed = TUNABLE_PARAMS.get(spec.class_name, {})
    assert set(tunable_params.keys()) <= set(allowed.keys()), (
        f"Tunable parameters {list(tunable_params.keys())} not all in default tunable parameters "
        f"{list(allowed.keys())} for environment {spec.class_name}"
    )
    if len(tunable_params) > A.NSG_MAX_PARAMS:
        raise ValueError(f"at most {A.NSG_MAX_PARAMS} tunable parameters")
    is_fl = et.env_type == A.ENV_FROZENLAKE
    cfg = A.Config()
    cfg.abi_version = A.NSG_ABI_VERSION
    cfg.env_type = et.env_type
    cfg.n_params = len(tunable_params)
    cfg.max_episode_steps = int(spec.max_episode_steps) if spec.max_episode_steps else 0
    flags = 0
    if change_notification:
        flags |= A.F_CHANGE_NOTIFICATION
    if delta_change_notification:
        flags |= A.F_DELTA_NOTIFICATION
    if persistent_params:
        flags |= A.F_PERSISTENT_PARAMS
    if track_returns:
        flags |= A.F_TRACK_RETURNS
    if in_sim_change:
        flags |= A.F_IN_SIM_CHANGE
    if is_sim_env:
        flags |= A.F_SIM_ENV
    tables = TableBuilder()
    if is_fl:
        assert "P" in tunable_params, "NSFrozenLakeWrapper requires tunable_params['P']"
        ipd = [1, 0, 0] if initial_prob_dist is None else list(initial_prob_dist)
        assert sum(ipd) == 1 or math.isclose(sum(ipd), 1), "The sum of transition probabilities must be 1."
        assert len(ipd) == 3, (
            "The length of the transition probability distribution must be 3. Each action can have at most 3 possible outcomes."
        )
        for k in range(3):
            cfg.initial_prob[k] = float(ipd[k])
        desc = spec.desc
        cfg.nrow, cfg.ncol = len(desc), len(desc[0])
        raw = "".join(desc).encode()
        assert set(raw) <= set(b"SFHG"), "FrozenLake desc may contain only S, F, H, G"
        cfg.desc_tab_off = tables.add_bytes(raw)
        if modified_rewards:
            flags |= A.F_MODIFIED_REWARDS
            for j, letter in enumerate("SFHG"):
                cfg.letter_reward[j] = float(modified_rewards[letter])
    else:
        for k, (name, default) in enumerate(zip(et.theta_names, et.theta_defaults)):
            cfg.base_theta[k] = float(spec.theta_overrides.get(name, default))
    cfg.flags = flags
    names = list(tunable_params.keys())
    for j, (name, fn) in enumerate(tunable_params.items()):
        assert isinstance(fn, UpdateFn), f"tunable_params[{name!r}] must be an UpdateFn, got {type(fn)}"
        if is_fl:
            assert isinstance(fn, UpdateDistributionFn), "FrozenLake 'P' needs an UpdateDistributionFn"
        else:
            assert not isinstance(fn, UpdateDistributionFn), f"{name}: scalar parameter needs a scalar UpdateFn"
        pc = cfg.params[j]
        pc.theta_slot = 0 if is_fl else et.theta_names.index(name)
        pc.rng_child = j
        pc.sched_end = float("inf")
        fields = {}
        fields.update(fn.scheduler._compile(tables, cfg.max_episode_steps or None))
        fields.update(fn._compile(tables))
        for key, val in fields.items():
            if key == "u":
                for q, x in enumerate(val):
                    pc.u[q] = x
            else:
                setattr(pc, key, val)
    return cfg, tables.blob(), spec, names


# --------------------------------------------------------------------------- neutral JSON specs
# (used by tests/golden: the same spec instantiates the reference's classes, the oracle and
# this package's same-named classes)


def _dec(v):
    if v == _INF:
        return np.inf
    if isinstance(v, dict) and "__set__" in v:
        return set(v["__set__"])
    if isinstance(v, dict) and "__tuples__" in v:
        return [tuple(x) for x in v["__tuples__"]]
    return v


def build_fn(fn_spec: dict):
    from . import schedulers as S
    from . import update_functions as U

    sname, skw = fn_spec["scheduler"]
    sched = getattr(S, sname)(**{k: _dec(v) for k, v in skw.items()})
    uname, ukw = fn_spec["update"]
    return getattr(U, uname)(sched, **copy.deepcopy({k: _dec(v) for k, v in ukw.items()}))


def build_tunable_params(params_spec: dict) -> dict:
    return {name: build_fn(fs) for name, fs in params_spec.items()}
