"""User-defined schedulers and update functions: the reference's own extension idiom for the hot path.

The reference lets a user plug in a subclass of `base.Scheduler` that defines `_check(t)` and a subclass of `base.UpdateFn` /
`base.UpdateDistributionFn` that defines `_update(param, t)` (ns_gym/base.py:50-95, 98-203; tutorial.ipynb cells 38-44).  A Python
method cannot run inside the kernel.  What the kernels can do is READ TABLES: a fire pattern as a bit table over t
(NSG_SCHED_TABLE) and a sequence of values consumed in fire order (NSG_UPD_STEPWISE / NSG_UPD_D_STEPWISE).  So wherever the θ chain
such an object produces is THE SAME FOR EVERY ENV of the batch and every episode, that chain is sampled once on the host - by
running the θ side of one reference wrapper episode, in the reference's call order - and stored in those tables:

    for t = 0 .. horizon                                              wrapper time (base.py:314, reset: :379)
      for (name, fn) in tunable_params.items()                         classic_control.py:80-85, toy_text.py:362-364
        fired = fn.scheduler.start <= t <= fn.scheduler.end and fn.scheduler._check(t)      base.py:79-81
        new   = fn._update(copy.copy(cur[name]), t) if fired else cur[name]                 base.py:139-149
        fn.prev_param, fn.prev_time = cur[name], t
      rejected = constraint checker over this step's proposals          classic_control.py:87-92, 193-422
      cur[name] = new unless rejected

The objects sampled are a DEEP COPY of the user's, taken at construction: the reference wrapper restarts every episode from
`deepcopy(init_initial_params)` (base.py:381-384), so an object that keeps state (a call counter, a seeded generator of its own)
restarts with the episode there too.  The kernel evaluates flag, delta (`new - cur`, or W1) and the constraint check itself from the
tabled proposal, in float64, exactly as it does for the built-in kinds.

The contract, checked where it can be and otherwise stated: the chain must not depend on anything that differs between envs or
episodes.  Refused with `NsgError` naming the property:
  * the chain is not reproducible (two fresh deep copies disagree: the method draws from a global / unseeded generator or reads
    outside state - the tutorial's `StochasticScheduler`); use `RandomScheduler(probability=...)` for per-env Bernoulli firing;
  * an update function that owns an `rng` attribute (the wrapper re-seeds it from the reset seed and carries it across episodes,
    base.py:151-158, 412-431: its chain differs per env and per episode);
  * a user-defined update function behind a stochastic built-in scheduler (per-env stream) or with `persistent_params=True`
    (θ survives a reset while t restarts: its value at (episode, t) depends on every earlier episode's length);
  * a stateful user-defined scheduler with `persistent_params=True` (the objects are not re-copied at reset there);
  * an overridden `_get_delta_change` (the kernels compute `new - cur` / W1);
  * Acrobot link-length / centre-of-mass parameters whose cross-check partner (classic_control.py:241-357) is driven by a built-in
    update function (that partner's proposals exist on the device only).
"""
from __future__ import annotations

import copy

import numpy as np

from ._lib import NsgError
from .base import Scheduler, UpdateDistributionFn, UpdateFn

_PKG = __name__.rsplit(".", 1)[0]


def _defined_by_user(cls, method: str) -> bool:
    f = getattr(cls, method, None)
    return f is not None and not str(getattr(f, "__module__", "")).startswith(_PKG)


def is_user_scheduler(s) -> bool:
    """A `base.Scheduler` subclass whose `_check` is the user's own (ns_gym/base.py:83-95)."""
    return isinstance(s, Scheduler) and _defined_by_user(type(s), "_check")


def is_user_update_fn(fn) -> bool:
    """A `base.UpdateFn` / `UpdateDistributionFn` subclass whose `_update` is the user's own (ns_gym/base.py:160-170)."""
    return isinstance(fn, UpdateFn) and _defined_by_user(type(fn), "_update")


def host_fires(scheduler, t) -> bool:
    """`Scheduler.__call__` (base.py:67-81) on the host, for schedulers that have a host-side `_check`: user-defined ones and the
    deterministic built-ins (pure functions of t).  The stochastic built-ins own a per-env device stream and have none."""
    if scheduler.start <= t <= scheduler.end:
        return bool(scheduler._check(t))
    return False


def has_host_check(s) -> bool:
    return is_user_scheduler(s) or not getattr(s, "_stochastic", False)


# ---- physical constraints (classic_control.py:193-422) as a function of this step's proposals --------------------------------
# Same decisions as the kernels' `constraint_mask` / `own_constraint_violated` (csrc/nsg_envs.hip.h, nsg_kernels.hip.h); needed on
# the host because a rejected proposal leaves θ where it was and the NEXT `_update` call must be given that value.
_ACRO_PARTNERS = {"LINK_LENGTH_1": ("LINK_COM_POS_1",), "LINK_COM_POS_1": ("LINK_LENGTH_1",), "LINK_COM_POS_2": ("LINK_LENGTH_2",)}


def constraint_violated(class_name: str, name: str, new_vals: dict, cur: dict) -> bool:
    """Is the proposal `new_vals[name]` rejected?  `new_vals`: this step's proposals of every tuned parameter the caller knows;
    `cur`: the current value of every θ of the env (tuned or not)."""
    v = new_vals[name]
    if class_name == "CartPoleEnv":                                         # :208-235
        return v < 0 if name == "gravity" else (v <= 0 if name in ("length", "masscart", "masspole") else False)
    if class_name == "PendulumEnv":                                         # :389-420
        return v < 0 if name == "g" else (v <= 0 if name in ("m", "l", "dt") else False)
    if class_name == "MountainCarEnv":                                      # :359-376
        return v <= 0 if name in ("gravity", "force") else False
    if class_name == "Continuous_MountainCarEnv":                           # :378-387
        return v <= 0 if name == "power" else False
    if class_name == "AcrobotEnv":                                          # :237-357
        if name == "LINK_LENGTH_1":
            if v <= 0:
                return True
            if "LINK_COM_POS_1" in new_vals and new_vals["LINK_COM_POS_1"] > v:
                return True
            return v < cur["LINK_COM_POS_1"]
        if name == "LINK_LENGTH_2":      # the centre-of-mass cross-checks of this branch are dead code in the reference (:267)
            return v <= 0
        if name in ("LINK_MASS_1", "LINK_MASS_2"):
            return v <= 0
        if name in ("LINK_COM_POS_1", "LINK_COM_POS_2"):
            length = "LINK_LENGTH_1" if name.endswith("1") else "LINK_LENGTH_2"
            if v <= 0:
                return True
            if length in new_vals and new_vals[length] < v:
                return True
            return v > cur[length]
        return False
    return False   # grid wrappers have no constraint checker


# ---- the sampling ---------------------------------------------------------------------------------------------------------------
class Tabulated:
    """What one parameter's user-defined objects were sampled into."""

    __slots__ = ("fired", "values")

    def __init__(self, fired, values):
        self.fired = fired      # uint8[horizon + 1]: the scheduler's answer at t (range gate applied)
        self.values = values    # None (built-in update fn) or the proposals in fire order: list of float / of n-lists

    def same_as(self, other) -> bool:
        if not np.array_equal(self.fired, other.fired) or (self.values is None) != (other.values is None):
            return False
        if self.values is None:
            return True
        a, b = np.asarray(self.values, dtype=np.float64), np.asarray(other.values, dtype=np.float64)
        return a.shape == b.shape and a.tobytes() == b.tobytes()     # bit for bit (NaN == NaN)


def _as_float(x, who):
    try:
        return float(x)
    except (TypeError, ValueError):
        raise NsgError(f"{who}._update returned {type(x).__name__}; a scalar update function must return a number") from None


def _as_dist(x, n, who):
    try:
        row = [float(v) for v in x]
    except (TypeError, ValueError):
        raise NsgError(f"{who}._update returned {type(x).__name__}; a distribution update function must return a list of "
                       f"{n} numbers") from None
    if len(row) != n:
        raise NsgError(f"{who}._update returned {len(row)} probabilities; this environment's slip distribution has {n}")
    return row


def _simulate(tunable_params, names, class_name, theta0, all_theta, n_dist, horizon, order=None):
    """One reference-wrapper episode's θ side for the parameters in `names` (dict order kept), t = 0 .. horizon."""
    tp = copy.deepcopy({k: tunable_params[k] for k in tunable_params})   # one deepcopy: shared objects stay shared
    cur = {k: copy.deepcopy(theta0[k]) for k in names}
    cur_all = dict(all_theta)
    fired = {k: np.zeros(horizon + 1, dtype=np.uint8) for k in names}
    values = {k: ([] if is_user_update_fn(tp[k]) else None) for k in names}
    for t in (range(horizon + 1) if order is None else order):
        new_vals, flags = {}, {}
        for k, fn in tp.items():
            if k not in names:
                continue
            f = host_fires(fn.scheduler, t)
            fired[k][t] = 1 if f else 0
            if values[k] is None:
                continue
            new = fn._update(copy.copy(cur[k]), t) if f else cur[k]      # base.py:139-149
            fn.prev_param, fn.prev_time = cur[k], t
            new_vals[k], flags[k] = new, f
        num = {k: (v if n_dist else _as_float(v, type(tp[k]).__name__)) for k, v in new_vals.items()}
        # every proposal is judged against the PRE-step values, then the accepted ones are written (classic_control.py:87-92)
        rejected = {k: (not n_dist) and constraint_violated(class_name, k, num, cur_all) for k in new_vals}
        for k, new in new_vals.items():
            if flags[k]:
                values[k].append(_as_dist(new, n_dist, type(tp[k]).__name__) if n_dist else num[k])
            if not rejected[k]:
                cur[k] = new
                if not n_dist:
                    cur_all[k] = num[k]
    return {k: Tabulated(fired[k], values[k]) for k in names}


def tabulate(tunable_params: dict, *, class_name: str, theta0: dict, all_theta: dict, n_dist: int, horizon, persistent: bool) -> dict:
    """{param name: Tabulated} for every parameter that involves a user-defined scheduler or update function ({} if none).

    theta0: the value each tuned parameter starts an episode with (float, or the initial distribution as given);
    all_theta: every θ of a classic-control env (name -> construction value), for Acrobot's cross-checks;
    n_dist: 0 for scalar parameters, else the support size of the slip distribution; horizon: largest t to sample."""
    names = [k for k, fn in tunable_params.items() if is_user_update_fn(fn) or is_user_scheduler(fn.scheduler)]
    if not names:
        return {}
    for k in names:
        fn = tunable_params[k]
        who = f"tunable_params[{k!r}]"
        if is_user_update_fn(fn):
            cls = type(fn).__name__
            if bool(n_dist) != isinstance(fn, UpdateDistributionFn):
                raise NsgError(f"{who}: {cls} must subclass " + ("UpdateDistributionFn" if n_dist else "UpdateFn (scalar parameter)"))
            if not has_host_check(fn.scheduler):
                raise NsgError(f"{who}: the user-defined update function {cls} cannot be fused behind {type(fn.scheduler).__name__}: that "
                               f"scheduler draws from a per-env stream, so after k fires θ differs from env to env and a Python `_update(param, t)` "
                               f"cannot be tabulated; use a deterministic scheduler (or a Scheduler subclass whose `_check` is reproducible)")
            if persistent:
                raise NsgError(f"{who}: the user-defined update function {cls} cannot be fused with persistent_params=True: θ survives a reset "
                               f"while t restarts at 0, so its value at (episode, t) depends on the length of every earlier episode of that env")
            if hasattr(fn, "rng"):
                raise NsgError(f"{who}: {cls} owns an `rng` attribute: the wrapper re-seeds it from the reset seed and carries it across episodes "
                               f"(ns_gym/base.py:151-158, 412-431), so its θ chain differs per env and per episode and cannot be tabulated; the "
                               f"built-in stochastic update functions (RandomWalk, OrnsteinUhlenbeck, ...) draw per env on the device")
            base_delta = UpdateDistributionFn._get_delta_change if n_dist else UpdateFn._get_delta_change
            if type(fn)._get_delta_change is not base_delta:
                raise NsgError(f"{who}: {cls} overrides `_get_delta_change`; the kernels compute the delta themselves ("
                               + ("1-Wasserstein distance" if n_dist else "`updated - param`") + ", ns_gym/base.py:172-203)")
            if class_name == "AcrobotEnv":
                for partner in _ACRO_PARTNERS.get(k, ()):
                    if partner in tunable_params and not is_user_update_fn(tunable_params[partner]):
                        raise NsgError(f"{who}: Acrobot's constraint checker compares {k} with this step's proposal for {partner} "
                                       f"(classic_control.py:241-357), which a built-in update function computes on the device only; drive both "
                                       f"with user-defined update functions, or neither")
    if horizon is None:
        raise NsgError("user-defined schedulers / update functions are sampled into tables over t = 0 .. horizon, and this env has no "
                       "TimeLimit to bound t: set `<object>.nsg_horizon = <largest t reached>` on one of them")
    horizon = int(horizon)
    kw = dict(class_name=class_name, theta0=theta0, all_theta=all_theta, n_dist=n_dist, horizon=horizon)
    try:
        a = _simulate(tunable_params, names, **kw)
        b = _simulate(tunable_params, names, **kw)
    except NsgError:
        raise
    except Exception as e:    # the user's own method failed while being sampled: say where
        raise type(e)(f"{e} (raised while ns_gym_amd sampled the user-defined scheduler / update function over t = 0 .. {horizon}; "
                      f"the reference would raise it inside step())") from e
    for k in names:
        if not a[k].same_as(b[k]):
            what = "`_update`" if a[k].values is not None and np.array_equal(a[k].fired, b[k].fired) else "`_check`"
            raise NsgError(f"tunable_params[{k!r}]: two samplings of the user-defined {what} over the same t = 0 .. {horizon} disagree: "
                           f"it draws from a global or unseeded random generator, or reads state outside the object, so there is no one chain "
                           f"to tabulate and the kernels cannot reproduce it.  For per-env random firing use RandomScheduler(probability=...) / "
                           f"MemorylessScheduler; for per-env random values RandomWalk / RandomCategorical and friends")
    if persistent:   # (only user-defined SCHEDULERS get here) the objects are not re-copied at reset: `_check` must be a function of t alone
        # asked about the same t in two other orders (back to front; a fixed shuffle): a function of t alone answers the same
        orders = (list(range(horizon, -1, -1)), [int(x) for x in np.random.default_rng(0).permutation(horizon + 1)])
        others = [_simulate(tunable_params, names, order=o, **kw) for o in orders]
        for k in names:
            if any(not np.array_equal(a[k].fired, c[k].fired) for c in others):
                raise NsgError(f"tunable_params[{k!r}]: the user-defined scheduler {type(tunable_params[k].scheduler).__name__} answers differently "
                               f"when asked in another order: it keeps state between calls, and with persistent_params=True the wrapper does not "
                               f"re-copy it at reset (ns_gym/base.py:392-395), so its fire pattern is not a function of t")
    return a


def horizon_hint(tunable_params: dict):
    """`nsg_horizon` of any user-defined object (for envs without a TimeLimit), else None."""
    hs = [getattr(o, "nsg_horizon") for fn in tunable_params.values() for o in (fn, fn.scheduler) if getattr(o, "nsg_horizon", None) is not None]
    return max(int(h) for h in hs) if hs else None
