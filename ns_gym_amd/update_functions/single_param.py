"""Scalar update functions with the reference's names and signatures
(ns_gym/update_functions/single_param.py).  Descriptors only: the arithmetic runs in the
fused HIP kernel (ns_gym_amd/csrc/nsg_theta.hip.h)."""
from __future__ import annotations

from typing import Union

from .. import _abi as A
from ..base import Scheduler, UpdateFn


def _u(*vals):
    u = [0.0] * 10
    for i, v in enumerate(vals):
        u[i] = float(v)
    return u


class _Seeded(UpdateFn):
    """Update fn that owns a NumPy-compatible PCG64 stream (`self.rng` in the reference)."""

    def __init__(self, scheduler, seed=None):
        super().__init__(scheduler)
        self.seed_value = seed

    def _rng_fields(self):
        d = {"uses_rng": 1}
        if self.seed_value is not None:
            d.update(has_fn_seed=1, fn_seed=int(self.seed_value))
        return d


class IncrementUpdate(UpdateFn):
    """θ + k (single_param.py:154-175)."""

    def __init__(self, scheduler: Scheduler, k: float) -> None:
        super().__init__(scheduler)
        self.k = k

    def _compile(self, tables):
        return {"upd_kind": A.UPD_INCREMENT, "u": _u(self.k)}


class DecrementUpdate(UpdateFn):
    """θ − k (single_param.py:178-199)."""

    def __init__(self, scheduler, k) -> None:
        super().__init__(scheduler)
        self.k = k

    def _compile(self, tables):
        return {"upd_kind": A.UPD_DECREMENT, "u": _u(self.k)}


class DeterministicTrend(UpdateFn):
    """θ + slope·t (single_param.py:20-40)."""

    def __init__(self, scheduler: Scheduler, slope: float) -> None:
        super().__init__(scheduler)
        self.slope = slope

    def _compile(self, tables):
        return {"upd_kind": A.UPD_TREND, "u": _u(self.slope)}


class PolynomialTrend(UpdateFn):
    """θ + Σ aᵢ t^(i+1) (single_param.py:451-473)."""

    def __init__(self, scheduler: Scheduler, coeffs: list) -> None:
        super().__init__(scheduler)
        self.coeffs = coeffs

    def _compile(self, tables):
        off, ln = tables.add_values(list(self.coeffs) or [0.0])
        return {"upd_kind": A.UPD_POLY, "val_tab_off": off, "val_tab_len": len(self.coeffs)}


class GeometricProgression(UpdateFn):
    """θ·r (single_param.py:290-307)."""

    def __init__(self, scheduler, r):
        super().__init__(scheduler)
        self.r = r

    def _compile(self, tables):
        return {"upd_kind": A.UPD_GEOMETRIC, "u": _u(self.r)}


class ExponentialDecay(UpdateFn):
    """θ·exp(−λ t) (single_param.py:266-287)."""

    def __init__(self, scheduler: Scheduler, decay_rate: float) -> None:
        super().__init__(scheduler)
        self.decay_rate = decay_rate

    def _compile(self, tables):
        return {"upd_kind": A.UPD_EXPDECAY, "u": _u(self.decay_rate)}


class OscillatingUpdate(UpdateFn):
    """θ + δ·sin(t) (single_param.py:243-264)."""

    def __init__(self, scheduler: Scheduler, delta: float) -> None:
        super().__init__(scheduler)
        self.delta = delta

    def _compile(self, tables):
        return {"upd_kind": A.UPD_OSCILLATING, "u": _u(self.delta)}


class SigmoidTransition(UpdateFn):
    """a + (b−a)/(1+exp(−k(t−t0))), replaces θ (single_param.py:349-385)."""

    def __init__(self, scheduler: Scheduler, a: float, b: float, k: float, t0: float) -> None:
        super().__init__(scheduler)
        self.a, self.b, self.k, self.t0 = a, b, k, t0

    def _compile(self, tables):
        return {"upd_kind": A.UPD_SIGMOID, "u": _u(self.a, self.b, self.k, self.t0)}


class LinearInterpolation(UpdateFn):
    """start + (end−start)·min(t/T, 1), replaces θ (single_param.py:476-508)."""

    def __init__(self, scheduler: Scheduler, start_val: float, end_val: float, T: int) -> None:
        super().__init__(scheduler)
        self.start_val, self.end_val, self.T = start_val, end_val, T

    def _compile(self, tables):
        return {"upd_kind": A.UPD_LERP, "u": _u(self.start_val, self.end_val, self.T)}


class StepWiseUpdate(UpdateFn):
    """Next value of `param_list` on each fire; an exhausted list leaves θ unchanged but
    still reports fired=1 (single_param.py:202-223)."""

    def __init__(self, scheduler: Scheduler, param_list: list) -> None:
        super().__init__(scheduler)
        self.param_list = param_list

    def _compile(self, tables):
        off, _ = tables.add_values(list(self.param_list) or [0.0])
        return {"upd_kind": A.UPD_STEPWISE, "val_tab_off": off, "val_tab_len": len(self.param_list)}


class CyclicUpdate(UpdateFn):
    """Cycle through `value_list` (single_param.py:388-408)."""

    def __init__(self, scheduler: Scheduler, value_list: list) -> None:
        super().__init__(scheduler)
        self.value_list = value_list
        self._index = 0

    def _compile(self, tables):
        if len(self.value_list) == 0:
            raise ValueError("CyclicUpdate: value_list must not be empty")
        off, _ = tables.add_values(list(self.value_list))
        return {"upd_kind": A.UPD_CYCLIC, "val_tab_off": off, "val_tab_len": len(self.value_list)}


class NoUpdate(UpdateFn):
    """θ unchanged, fired=1, delta 0 (single_param.py:226-240)."""

    def __init__(self, scheduler: Scheduler) -> None:
        super().__init__(scheduler)

    def _compile(self, tables):
        return {"upd_kind": A.UPD_NOUPDATE}


class RandomWalk(_Seeded):
    """θ + N(μ, σ) (single_param.py:84-113)."""

    def __init__(self, scheduler: Scheduler, mu: Union[float, int] = 0, sigma: Union[float, int] = 1,
                 seed=None):
        super().__init__(scheduler, seed)
        self.mu, self.sigma = mu, sigma

    def _compile(self, tables):
        return {"upd_kind": A.UPD_RANDOMWALK, "u": _u(self.mu, self.sigma), **self._rng_fields()}


class RandomWalkWithDrift(_Seeded):
    """α + θ + N(μ, σ) (single_param.py:116-151)."""

    def __init__(self, scheduler: Scheduler, alpha: float, mu: float, sigma: float,
                 seed: Union[int, None] = None) -> None:
        super().__init__(scheduler, seed)
        self.alpha, self.mu, self.sigma = alpha, mu, sigma

    def _compile(self, tables):
        return {"upd_kind": A.UPD_RW_DRIFT, "u": _u(self.alpha, self.mu, self.sigma), **self._rng_fields()}


class RandomWalkWithDriftAndTrend(_Seeded):
    """α + θ + N(μ, σ) + slope·t (single_param.py:43-81)."""

    def __init__(self, scheduler: Scheduler, alpha: float, mu: float, sigma: float, slope: float,
                 seed: Union[int, None] = None) -> None:
        super().__init__(scheduler, seed)
        self.alpha, self.mu, self.sigma, self.slope = alpha, mu, sigma, slope

    def _compile(self, tables):
        return {"upd_kind": A.UPD_RW_DRIFT_TREND, "u": _u(self.alpha, self.mu, self.sigma, self.slope),
                **self._rng_fields()}


class OrnsteinUhlenbeck(_Seeded):
    """θ + θr(μ−θ) + N(0, σ); no draw when σ ≤ 0 (single_param.py:310-346)."""

    def __init__(self, scheduler: Scheduler, theta: float, mu: float, sigma: float = 0.0,
                 seed: Union[int, None] = None) -> None:
        super().__init__(scheduler, seed)
        self.theta, self.mu, self.sigma = theta, mu, sigma

    def _compile(self, tables):
        return {"upd_kind": A.UPD_OU, "u": _u(self.theta, self.mu, self.sigma), **self._rng_fields()}


class BoundedRandomWalk(_Seeded):
    """clip(θ + N(μ, σ), lo, hi) (single_param.py:411-448)."""

    def __init__(self, scheduler: Scheduler, mu: float, sigma: float, lo: float, hi: float,
                 seed: Union[int, None] = None) -> None:
        super().__init__(scheduler, seed)
        self.mu, self.sigma, self.lo, self.hi = mu, sigma, lo, hi

    def _compile(self, tables):
        return {"upd_kind": A.UPD_BOUNDED_RW, "u": _u(self.mu, self.sigma, self.lo, self.hi),
                **self._rng_fields()}


__all__ = [
    "BoundedRandomWalk", "CyclicUpdate", "DecrementUpdate", "DeterministicTrend", "ExponentialDecay",
    "GeometricProgression", "IncrementUpdate", "LinearInterpolation", "NoUpdate", "OrnsteinUhlenbeck",
    "OscillatingUpdate", "PolynomialTrend", "RandomWalk", "RandomWalkWithDrift",
    "RandomWalkWithDriftAndTrend", "SigmoidTransition", "StepWiseUpdate",
]
