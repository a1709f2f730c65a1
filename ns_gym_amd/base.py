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

from . import _abi as A


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

    # -- compile-time description ------------------------------------------------------
    def _compile(self, tables: "TableBuilder", horizon: int | None) -> dict:
        raise NotImplementedError

    def _range(self) -> dict:
        return {"sched_start": float(self.start), "sched_end": float(self.end)}

    # -- evaluation (device) -------------------------------------------------------------
    def __call__(self, t: int) -> bool:
        """`start <= t <= end and _check(t)` evaluated by the device θ-engine."""
        from . import functional

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
        raise NotImplementedError

    @property
    def _uses_rng(self) -> bool:
        return hasattr(self, "seed_value")

    def __call__(self, param: Any, t: Union[int, float]) -> tuple[Any, int, float]:
        """`(param', fired, delta)` for one (param, t) via the device θ-engine (stateless use:
        list cursors / RNG streams start fresh on every call; use `functional.theta_trace`
        for a fed-back trajectory)."""
        assert isinstance(t, (int, float)), (
            f"Expected t to be an int or float, got {type(t)}, Arrays operations need to inherit from UpdateDistributionFn"
        )
        from . import functional

        th, fired, delta = functional.theta_trace(self, param, t0=int(t), T=1)
        self.prev_param = param
        self.prev_time = t
        if self._is_distribution:
            return ([float(x) for x in th[0]] if fired[0] else param, int(fired[0]), float(delta[0]))
        return (float(th[0]) if fired[0] else param, int(fired[0]), float(delta[0]))


class UpdateDistributionFn(UpdateFn):
    """Update functions over a distribution represented as a list (ns_gym/base.py:185-203);
    delta is the 1-Wasserstein distance over integer support (ns_gym/utils.py:55-94)."""

    _is_distribution = True

    def __call__(self, param: Any, t: Union[int, float]) -> Any:
        assert isinstance(param, list), f"param must be a list, got {type(param)}"
        return super().__call__(param, t)


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
