"""Schedulers with the reference's names and constructor signatures (ns_gym/schedulers.py).

Deterministic schedulers that are pure functions of t compile either to a closed form the
kernel evaluates (Continuous, Periodic, Burst) or to a bit table over t staged in LDS
(Discrete, Window, Custom).  Their `_check(t)` (the reference's method name, schedulers.py:52-53,73-74,88-89,139-140,197-198)
exists for ONE purpose: when a user-defined update function sits behind such a scheduler, its θ chain is sampled on the host
(`ns_gym_amd.extension`) and needs the fire pattern there.  Calling a built-in scheduler object evaluates it on the device.
A user-defined scheduler is a `base.Scheduler` subclass that defines `_check` (ns_gym/base.py:83-95); see `ns_gym_amd.extension`.
"""
from __future__ import annotations

import numpy as np

from . import _abi as A
from .base import Scheduler



class ContinuousScheduler(Scheduler):
    """Fires at every step in range (ns_gym/schedulers.py:46-53)."""

    def __init__(self, start=0, end=np.inf) -> None:
        super().__init__(start, end)

    def _check(self, t):
        return True

    def _compile(self, tables, horizon):
        return {"sched_kind": A.SCHED_CONTINUOUS, **self._range()}


class PeriodicScheduler(Scheduler):
    """Fires when t % period == 0 (ns_gym/schedulers.py:77-89)."""

    def __init__(self, period: int, start=0, end=np.inf) -> None:
        super().__init__(start, end)
        self.period = period

    def _check(self, t):
        return t % self.period == 0

    def _compile(self, tables, horizon):
        if int(self.period) != self.period or self.period <= 0:
            raise ValueError("PeriodicScheduler: period must be a positive integer")
        return {"sched_kind": A.SCHED_PERIODIC, "sched_i0": int(self.period), **self._range()}


class BurstScheduler(Scheduler):
    """on_duration firing steps then off_duration silent steps, cyclically
    (ns_gym/schedulers.py:119-140)."""

    def __init__(self, on_duration: int, off_duration: int, start=0, end=np.inf) -> None:
        super().__init__(start, end)
        self.on_duration = on_duration
        self.off_duration = off_duration
        self.cycle = on_duration + off_duration

    def _check(self, t):
        return (t % self.cycle) < self.on_duration

    def _compile(self, tables, horizon):
        if self.cycle <= 0:
            raise ValueError("BurstScheduler: on_duration + off_duration must be positive")
        return {"sched_kind": A.SCHED_BURST, "sched_i0": int(self.on_duration),
                "sched_i1": int(self.off_duration), **self._range()}


_TABLE_BITS = 8 * 12288     # what one scheduler may take of the kernels' 16-KiB constant tables


def _fit_table(n: int, horizon, who: str):
    """Length of a scheduler's bit table over t = 0 .. n-1 and what the kernels answer beyond it (0: never fires - exact when the
    table covers every listed time; 2: unknown - counted and raised, like a sampled CustomScheduler).  A table that does not fit
    the constant-table budget is cut at the largest t the batch can reach (`horizon`: 2 x max_episode_steps, see
    CustomScheduler); times beyond it cannot be reached anyway, and a t that does get there (a copy of a copy) is reported, not
    answered silently."""
    if n <= _TABLE_BITS:
        return n, 0
    if horizon and int(horizon) + 1 <= _TABLE_BITS:
        return int(horizon) + 1, 2
    raise ValueError(f"{who}: times up to {n - 1} do not fit the kernels' constant tables ({_TABLE_BITS} steps) and this env has "
                     f"no TimeLimit to bound t; list times below {_TABLE_BITS}")


class DiscreteScheduler(Scheduler):
    """Fires at the listed time steps (ns_gym/schedulers.py:56-74)."""

    def __init__(self, event_list: set, start=0, end=np.inf) -> None:
        super().__init__(start, end)
        self.event_list = event_list
        assert min(event_list) >= start, "Scheduler start time occurs after first event in event list"
        assert max(event_list) <= end, "Scheduler end time occurs before last event in event list"

    def _check(self, t):
        return t in self.event_list

    def _compile(self, tables, horizon):
        n, beyond = _fit_table(int(max(self.event_list)) + 1, horizon, "DiscreteScheduler")
        bits = np.zeros(max(n, 1), dtype=np.uint8)
        for e in self.event_list:
            if 0 <= e < n and int(e) == e:
                bits[int(e)] = 1
        off, ln = tables.add_bits(bits)
        return {"sched_kind": A.SCHED_TABLE, "sched_tab_off": off, "sched_tab_len": ln, "sched_i0": beyond,
                **self._range()}


class WindowScheduler(Scheduler):
    """Fires inside any inclusive (start, end) window (ns_gym/schedulers.py:180-198)."""

    def __init__(self, windows: list, start=0, end=np.inf) -> None:
        super().__init__(start, end)
        self.windows = windows

    def _check(self, t):
        return any(w_start <= t <= w_end for w_start, w_end in self.windows)

    def _compile(self, tables, horizon):
        finite = [w_end for _, w_end in self.windows if np.isfinite(w_end)]
        starts = [w_start for w_start, _ in self.windows]
        n, cut = _fit_table(int(max(finite + starts + [0])) + 2, horizon, "WindowScheduler")
        t = np.arange(n)
        bits = np.zeros(n, dtype=np.uint8)
        for w_start, w_end in self.windows:
            bits |= ((w_start <= t) & (t <= w_end)).astype(np.uint8)
        # beyond an uncut table: inside an open-ended window (1) or nothing (0); beyond a cut one: unknown (2)
        beyond = cut or int(any(not np.isfinite(w_end) for _, w_end in self.windows))
        off, ln = tables.add_bits(bits)
        return {"sched_kind": A.SCHED_TABLE, "sched_tab_off": off, "sched_tab_len": ln, "sched_i0": beyond,
                **self._range()}


class CustomScheduler(Scheduler):
    """User-defined event function of t (ns_gym/schedulers.py:31-43).  An arbitrary Python callable cannot run in the
    kernel: it is sampled once into a bit table over t = 0 .. horizon.  The reference calls `event_function(t)` for ANY t,
    so the table must cover every t the batch can reach:
      * `horizon=` given: that many steps;
      * env with a TimeLimit: 2 x max_episode_steps (an episode ends at max_episode_steps; a planning copy taken late in an
        episode keeps the source's t while its own TimeLimit restarts, classic_control.py:168-180, so it can reach
        t_src + max_episode_steps - 1);
      * env WITHOUT a TimeLimit and no `horizon=`: refused at construction (ValueError) - t is unbounded there.
    A t beyond the table (a copy of a copy, a `reset`-less run) is not answered silently: the scheduler does not fire, the
    kernels count it (NSG_CNT_SCHED_OVERRUN) and `VecNSEnv.check_errors()` / the N = 1 adaptors raise."""

    def __init__(self, event_function, start=0, end=np.inf, horizon: int | None = None) -> None:
        super().__init__(start, end)
        self.event_function = event_function
        self.horizon = horizon

    def _check(self, t):
        return self.event_function(t)

    def _compile(self, tables, horizon):
        if self.horizon is not None:
            h = int(self.horizon)
        elif horizon:
            h = int(horizon)      # compile_config passes 2 x max_episode_steps, or the t a host-side call asks about
        else:
            raise ValueError("CustomScheduler: this env has no TimeLimit, so t is unbounded; pass horizon=<largest t the "
                             "event function will be asked about> (the callable is sampled into a table, it cannot run on the GPU)")
        if h + 1 > 8 * 12288:
            raise ValueError(f"CustomScheduler: a horizon of {h} steps does not fit the kernels' 16-KiB constant tables")
        bits = np.array([1 if self.event_function(t) else 0 for t in range(h + 1)], dtype=np.uint8)
        off, ln = tables.add_bits(bits)
        return {"sched_kind": A.SCHED_TABLE, "sched_tab_off": off, "sched_tab_len": ln, "sched_i0": 2,
                **self._range()}


class _StochasticScheduler(Scheduler):
    """Schedulers that own a NumPy-compatible PCG64 stream (`self.rng` in the reference).  The stream
    is part of the wrapper's deep-copied initial params: a non-persistent reset rewinds it, and
    `reset(seed=...)` never re-seeds it (ns_gym/base.py:151-158,381-384)."""

    seed_value = None
    _stochastic = True   # Scheduler.__call__ keeps this object's stream / transition_time between calls

    def _seed_fields(self):
        return {"has_sched_seed": 1, "sched_seed": int(self.seed_value)} if self.seed_value is not None else {}


class RandomScheduler(_StochasticScheduler):
    """Fires with a fixed probability at every step in range (ns_gym/schedulers.py:9-28)."""

    def __init__(self, probability: float = 0.5, start=0, end=np.inf, seed=None) -> None:
        super().__init__(start, end)
        self.probability = probability
        self.seed_value = seed

    def _compile(self, tables, horizon):
        return {"sched_kind": A.SCHED_RANDOM, "sched_p0": float(self.probability), **self._seed_fields(), **self._range()}


class DecayingProbabilityScheduler(_StochasticScheduler):
    """Fires with probability p0·exp(−λt) (ns_gym/schedulers.py:143-177)."""

    def __init__(self, initial_probability: float, decay_rate: float, start=0, end=np.inf, seed=None) -> None:
        super().__init__(start, end)
        self.initial_probability = initial_probability
        self.decay_rate = decay_rate
        self.seed_value = seed

    def _compile(self, tables, horizon):
        return {"sched_kind": A.SCHED_DECAYING, "sched_p0": float(self.initial_probability),
                "sched_p1": float(self.decay_rate), **self._seed_fields(), **self._range()}


class MemorylessScheduler(_StochasticScheduler):
    """Geometric inter-event times (ns_gym/schedulers.py:92-116)."""

    def __init__(self, p: float, start=0, end=np.inf, seed=None) -> None:
        super().__init__(start, end)
        self.p = p
        self.seed_value = seed

    def _compile(self, tables, horizon):
        if not (0.0 < float(self.p) <= 1.0):
            raise ValueError("MemorylessScheduler: p must be in (0, 1]")
        return {"sched_kind": A.SCHED_MEMORYLESS, "sched_p0": float(self.p), **self._seed_fields(), **self._range()}


__all__ = [
    "BurstScheduler", "ContinuousScheduler", "CustomScheduler", "DecayingProbabilityScheduler",
    "DiscreteScheduler", "MemorylessScheduler", "PeriodicScheduler", "RandomScheduler", "WindowScheduler",
]
