"""Distribution update functions with the reference's names and signatures
(ns_gym/update_functions/distribution.py), for the slip distributions of the grid wrappers: n = 3
(NSFrozenLakeWrapper, NSBridgeWrapper) or n = 4 (NSCliffWalkingWrapper) (ns_gym/wrappers/toy_text.py)."""
from __future__ import annotations

from .. import _abi as A
from ..base import Scheduler, UpdateDistributionFn


def _u(*vals):
    u = [0.0] * 10
    for i, v in enumerate(vals):
        u[i] = float(v)
    return u


def _rows(rows, what, tables):
    """Distributions as float rows of the env's support size n (3: FrozenLake/Bridge, 4: CliffWalking;
    `tables.nd` is set by the config compiler before the update fns are compiled)."""
    n = getattr(tables, "nd", 3)
    out = []
    for r in rows:
        if len(r) != n:
            raise ValueError(f"{what}: every distribution must have length {n} for this environment")
        out.append([float(x) for x in r])
    return out


class DistributionIncrementUpdate(UpdateDistributionFn):
    """p₀ = min(1, p₀ + k), rest uniform (distribution.py:41-67)."""

    def __init__(self, scheduler: Scheduler, k: float) -> None:
        super().__init__(scheduler)
        self.k = k

    def _compile(self, tables):
        return {"upd_kind": A.UPD_D_INCREMENT, "u": _u(self.k)}


class DistributionDecrementUpdate(UpdateDistributionFn):
    """p₀ = max(0, p₀ − k), rest uniform (distribution.py:70-97)."""

    def __init__(self, scheduler: Scheduler, k: float) -> None:
        super().__init__(scheduler)
        self.k = k

    def _compile(self, tables):
        return {"upd_kind": A.UPD_D_DECREMENT, "u": _u(self.k)}


class DistributionStepWiseUpdate(UpdateDistributionFn):
    """Next distribution of `update_values` on each fire (distribution.py:100-130)."""

    def __init__(self, scheduler: Scheduler, update_values: list) -> None:
        super().__init__(scheduler)
        self.update_values = update_values

    def _compile(self, tables):
        rows = _rows(self.update_values, "DistributionStepWiseUpdate", tables)
        off, _ = tables.add_values([x for r in rows for x in r] or [0.0] * 4)
        return {"upd_kind": A.UPD_D_STEPWISE, "val_tab_off": off, "val_tab_len": len(rows)}


class DistributionCyclicUpdate(UpdateDistributionFn):
    """Cycle through `dist_list` (distribution.py:334-356)."""

    def __init__(self, scheduler: Scheduler, dist_list: list) -> None:
        super().__init__(scheduler)
        self.dist_list = dist_list
        self._index = 0

    def _compile(self, tables):
        rows = _rows(self.dist_list, "DistributionCyclicUpdate", tables)
        if not rows:
            raise ValueError("DistributionCyclicUpdate: dist_list must not be empty")
        off, _ = tables.add_values([x for r in rows for x in r])
        return {"upd_kind": A.UPD_D_CYCLIC, "val_tab_off": off, "val_tab_len": len(rows)}


class DistributionNoUpdate(UpdateDistributionFn):
    """p unchanged (distribution.py:217-231)."""

    def __init__(self, scheduler: Scheduler) -> None:
        super().__init__(scheduler)

    def _compile(self, tables):
        return {"upd_kind": A.UPD_D_NOUPDATE}


class UniformDrift(UpdateDistributionFn):
    """(1−rate)·p + rate/n (distribution.py:234-261)."""

    def __init__(self, scheduler: Scheduler, rate: float) -> None:
        super().__init__(scheduler)
        self.rate = rate

    def _compile(self, tables):
        return {"upd_kind": A.UPD_D_UNIFORMDRIFT, "u": _u(self.rate)}


class TargetReversion(UpdateDistributionFn):
    """p + θ(target − p) (distribution.py:264-293)."""

    def __init__(self, scheduler: Scheduler, target: list, theta: float) -> None:
        super().__init__(scheduler)
        self.target = target
        self.theta = theta

    def _compile(self, tables):
        (t,) = _rows([self.target], "TargetReversion", tables)
        return {"upd_kind": A.UPD_D_TARGETREV, "u": _u(*t, self.theta)}


class DistributionLinearInterpolation(UpdateDistributionFn):
    """start + (end − start)·min(t/T, 1) (distribution.py:296-331)."""

    def __init__(self, scheduler: Scheduler, start_dist: list, end_dist: list, T: int) -> None:
        super().__init__(scheduler)
        self.start_dist, self.end_dist, self.T = start_dist, end_dist, T

    def _compile(self, tables):
        s, e = _rows([self.start_dist, self.end_dist], "DistributionLinearInterpolation", tables)
        return {"upd_kind": A.UPD_D_LERP, "u": _u(*s, *e, self.T)}


class RandomCategorical(UpdateDistributionFn):
    """A fresh Dirichlet(1,…,1) sample on every fire: `list(rng.dirichlet(np.ones(n)))`
    (distribution.py:11-38); owns a PCG64 stream like the reference's `self.rng`."""

    def __init__(self, scheduler: Scheduler, seed=None) -> None:
        super().__init__(scheduler)
        self.seed_value = seed

    def _compile(self, tables):
        d = {"upd_kind": A.UPD_D_RANDOMCAT, "uses_rng": 1}
        if self.seed_value is not None:
            d.update(has_fn_seed=1, fn_seed=int(self.seed_value))
        return d


class LCBoundedDistrubutionUpdate(UpdateDistributionFn):
    """Lipschitz-bounded update (distribution.py:133-183): the inner update is re-sampled until
    W1(p, p') <= L * |t - prev_time| (at most 1e5 tries).  The inner fn is `RandomCategorical(scheduler)`
    by default, or any UpdateDistributionFn class constructible from a scheduler alone
    (`DistributionNoUpdate`).  `seed` (an extension: the reference leaves the inner sampler on OS
    entropy) fixes the inner stream; it is rewound by a non-persistent reset and never re-seeded by
    `reset(seed=...)`, because the wrapper only sees the outer fn, which has no `rng` (base.py:151-158)."""

    def __init__(self, scheduler, L: float, update_fn=None, seed=None) -> None:
        super().__init__(scheduler)
        self.L = L
        if update_fn is None:
            self.update_fn = RandomCategorical(scheduler)
        else:
            assert isinstance(update_fn, type) and issubclass(update_fn, UpdateDistributionFn), (
                "update_fn must be a subclass of base.UpdateDistributionFn"
            )
            self.update_fn = update_fn(scheduler)
        self.inner_seed = seed

    @property
    def _uses_rng(self):
        return isinstance(self.update_fn, RandomCategorical)

    def _compile(self, tables):
        if isinstance(self.update_fn, RandomCategorical):
            inner = 0
        elif isinstance(self.update_fn, DistributionNoUpdate):
            inner = 1
        else:
            from .._lib import NsgError

            raise NsgError(f"LCBoundedDistrubutionUpdate over {type(self.update_fn).__name__}: the kernels' rejection loop draws its candidates "
                           f"from RandomCategorical (or accepts DistributionNoUpdate at once); another inner update function would have to "
                           f"run as Python inside a per-env loop of up to 1e5 tries (distribution.py:167-183)")
        d = {"upd_kind": A.UPD_D_LCBOUNDED, "u": _u(self.L, inner), "uses_rng": 1 if inner == 0 else 0}
        if self.inner_seed is not None:
            d.update(has_fn_seed=1, fn_seed=int(self.inner_seed))
        return d


__all__ = [
    "DistributionCyclicUpdate", "DistributionDecrementUpdate", "DistributionIncrementUpdate",
    "DistributionLinearInterpolation", "DistributionNoUpdate", "DistributionStepWiseUpdate",
    "LCBoundedDistrubutionUpdate", "RandomCategorical", "TargetReversion", "UniformDrift",
]
