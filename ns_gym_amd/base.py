"""Core host-side types, mirroring `ns_gym/base.py` of the reference for the hot path.

`Scheduler`, `UpdateFn`, `UpdateDistributionFn` keep the reference's names, constructor
arguments and error behaviour (ns_gym/base.py:50-203) but are *descriptors*: they hold the
parameters that `ns_gym_amd.spec` compiles into the constant table of the fused HIP kernel.
Evaluating one (`scheduler(t)`, `update_fn(param, t)`) runs the device θ-engine through the
C-ABI (`nsg_theta_trace`); there is no CPU implementation in this package.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Union

import numpy as np



@dataclass(frozen=True)
class Reward:
    """Non-scalar reward carrying the notification flags (ns_gym/base.py:33-47)."""

    reward: Union[int, float]
    env_change: dict
    delta_change: Union[float, dict, None]
    relative_time: Union[int, float]


class Scheduler:
    """When to update a parameter; start and end inclusive (ns_gym/base.py:50-95)."""

    _kind: int = -1

    def __init__(self, start=0, end=np.inf) -> None:
        self.start = start
        self.end = end

    def _check(self, t: int) -> bool:
        """Core scheduling logic of a subclass (ns_gym/base.py:83-95): called only for start <= t <= end.  A USER-DEFINED
        subclass defines this and nothing else; it is sampled into a bit table over t when a batch is built
        (`ns_gym_amd.extension`).  The deterministic built-ins define it too (used when a user-defined update function behind
        them is sampled); the stochastic built-ins draw from a per-env device stream and have no host-side answer."""
        from ._lib import NsgError

        raise NsgError(f"{type(self).__name__} defines no `_check(t)`: subclass ns_gym_amd.base.Scheduler and implement it "
                       f"(ns_gym/base.py:83-95), or use one of ns_gym_amd.schedulers")

    # -- compile-time description ------------------------------------------------------
    def _compile(self, tables: "TableBuilder", horizon: int | None) -> dict:
        """Built-in schedulers override this.  A user-defined subclass (one that defines `_check`) never gets here:
        `spec.compile_config` samples it through `ns_gym_amd.extension.tabulate`."""
        from ._lib import NsgError

        raise NsgError(f"{type(self).__name__} cannot be compiled for the kernels: it neither is a built-in scheduler nor defines `_check(t)`")

    def _range(self) -> dict:
        return {"sched_start": float(self.start), "sched_end": float(self.end)}

    # -- evaluation ------------------------------------------------------------------------
    def __call__(self, t: int) -> bool:
        """`start <= t <= end and _check(t)` (ns_gym/base.py:67-81).  Built-in schedulers are evaluated by the device θ-engine;
        a stochastic one (Random, DecayingProbability, Memoryless) keeps its stream / transition_time between calls like
        the reference object does (schedulers.py:25-28,107-116,173-177).  A user-defined subclass runs its own `_check`."""
        from . import extension

        if extension.is_user_scheduler(self):
            return extension.host_fires(self, t)
        from . import functional

        if getattr(self, "_stochastic", False):
            if not hasattr(self, "_call_state"):
                self._call_state = {}
            return bool(functional.schedule_fires(self, [t], state=self._call_state)[0])
        return bool(functional.schedule_fires(self, [t])[0])


class UpdateFn:
    """How a scalar parameter is updated when its scheduler fires (ns_gym/base.py:98-182)."""

    _kind: int = -1
    _is_distribution = False

    def __init__(self, scheduler: Scheduler) -> None:
        assert isinstance(scheduler, Scheduler), (
            f"Expected scheduler to be a subclass of Scheduler, got {type(scheduler)}"
        )
        self.scheduler = scheduler
        self.prev_param = None
        self.prev_time = -1

    def _compile(self, tables: "TableBuilder") -> dict:
        """Built-in update functions override this.  A user-defined subclass (one that defines `_update`) never gets here:
        `spec.compile_config` samples its chain through `ns_gym_amd.extension.tabulate`."""
        from ._lib import NsgError

        raise NsgError(f"{type(self).__name__} cannot be compiled for the kernels: it neither is a built-in update function nor defines "
                       f"`_update(param, t)` (ns_gym/base.py:160-170)")

    def _update(self, param: Any, t: int) -> Any:
        """The update rule of a subclass (ns_gym/base.py:160-170), called when the scheduler fires.  A USER-DEFINED subclass
        defines this; the built-in update functions are arithmetic inside the kernels and have no host-side `_update`."""
        from ._lib import NsgError

        raise NsgError(f"{type(self).__name__}._update runs on the device only (csrc/nsg_theta.hip.h); call the object, or step a batch")

    def _get_delta_change(self, param: Any, updated_param: Any, t: int) -> float:
        """`updated_param - param` (ns_gym/base.py:172-182)."""
        return updated_param - param

    def _call_user(self, param: Any, t) -> tuple[Any, int, float]:
        """`UpdateFn.__call__` of the reference (ns_gym/base.py:139-149) for a user-defined subclass: its own Python."""
        import copy

        from . import extension

        s = self.scheduler
        fire = extension.host_fires(s, t) if extension.has_host_check(s) else bool(s(t))
        if fire:
            updated = self._update(copy.copy(param), t)
            delta = self._get_delta_change(param, updated, t)
            self.prev_param, self.prev_time = param, t
            return (updated, 1, delta)
        self.prev_param, self.prev_time = param, t
        return (param, 0, 0.0)

    @property
    def _uses_rng(self) -> bool:
        return hasattr(self, "seed_value")

    def seed(self, seed) -> None:
        """Re-seed the function's stream: `self.rng = np.random.default_rng(seed)` of the reference
        (base.py:151-158); no-op for deterministic functions."""
        if self._uses_rng:
            self.seed_value = seed
            if hasattr(self, "_call_state"):
                self._call_state.pop("rng", None)   # the next call starts from default_rng(seed); cursors are kept

    def __call__(self, param: Any, t: Union[int, float]) -> tuple[Any, int, float]:
        """`(param', fired, delta)` for one (param, t) via the device θ-engine.  The object is stateful like
        the reference's (base.py:124-149): its rng stream continues, StepWise / Cyclic lists advance, LCBounded
        remembers prev_time and a stochastic scheduler keeps its own state from call to call (the state lives in
        device tensors owned by this object; `seed()` re-seeds the stream)."""
        assert isinstance(t, (int, float)), (
            f"Expected t to be an int or float, got {type(t)}, Arrays operations need to inherit from UpdateDistributionFn"
        )
        from . import extension, functional

        if extension.is_user_update_fn(self):
            return self._call_user(param, t)
        if not hasattr(self, "_call_state"):
            self._call_state = {}
        th, fired, delta = functional.theta_trace(self, param, t0=int(t), T=1, state=self._call_state)
        self.prev_param = param
        self.prev_time = t
        f, d = int(fired.reshape(-1)[0]), float(delta.reshape(-1)[0])
        if self._is_distribution:
            return ([float(x) for x in th.reshape(-1)] if f else param, f, d)
        return (float(th.reshape(-1)[0]) if f else param, f, d)


class UpdateDistributionFn(UpdateFn):
    """Update functions over a distribution represented as a list (ns_gym/base.py:185-203);
    delta is the 1-Wasserstein distance over integer support (ns_gym/utils.py:55-94)."""

    _is_distribution = True

    def __call__(self, param: Any, t: Union[int, float]) -> Any:
        assert isinstance(param, list), f"param must be a list, got {type(param)}"
        return super().__call__(param, t)

    def _get_delta_change(self, param: Any, updated_param: Any, t: int) -> float:
        """1-Wasserstein distance between the two pmfs (ns_gym/base.py:192-203)."""
        from .utils import wasserstein_distance

        return wasserstein_distance(param, updated_param)


class TableBuilder:
    """Constant-table blob shared by all envs of a batch: bit tables (u32 words), value
    tables (f64) and byte tables, each 8-byte aligned.  Uploaded once by nsg_create and
    staged through LDS by the kernels."""

    def __init__(self) -> None:
        self._chunks: list[bytes] = []
        self._size = 0

    def _add(self, raw: bytes) -> int:
        off = self._size
        pad = (-len(raw)) % 8
        self._chunks.append(raw + b"\0" * pad)
        self._size += len(raw) + pad
        return off

    def add_bits(self, bits) -> tuple[int, int]:
        bits = np.asarray(bits, dtype=np.uint8)
        n = int(bits.size)
        words = np.zeros((n + 31) // 32 or 1, dtype=np.uint32)
        for i in np.flatnonzero(bits):
            words[i >> 5] |= np.uint32(1) << np.uint32(i & 31)
        off = self._add(words.tobytes())
        return off // 4, n

    def add_values(self, vals) -> tuple[int, int]:
        v = np.ascontiguousarray(np.asarray(vals, dtype=np.float64))
        off = self._add(v.tobytes())
        return off // 8, int(v.shape[0]) if v.ndim else 1

    def add_bytes(self, raw: bytes) -> int:
        return self._add(bytes(raw))

    def blob(self) -> bytes:
        return b"".join(self._chunks) if self._chunks else b"\0" * 8


def __getattr__(name):
    # `from ns_gym.base import NSWrapper` (base.py:206): here the common base of the N = 1 adaptors, which lives with
    # them in wrappers.py (it needs the vector env, and with it torch - not loaded by importing this module)
    if name == "NSWrapper":
        from .wrappers import _NSSingle

        return _NSSingle
    if name == "TUNABLE_PARAMS":   # base.py:1156 in the reference; the table lives with the env descriptors here
        from .envs import TUNABLE_PARAMS

        return TUNABLE_PARAMS
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
