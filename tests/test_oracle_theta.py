"""Pins the oracle's θ-schedule engine against vectors produced by the
reference's own Scheduler / UpdateFn classes (tests/golden/{schedulers,update_traces}.npz)."""
import numpy as np
import pytest

from ns_gym_amd.spec import build_fn
from oracle import oracle as O
from tests.util import MANIFEST, load

SCHED = MANIFEST["scheduler_specs"]
SCALAR = MANIFEST["scalar_update_specs"]
DIST = MANIFEST["dist_update_specs"]


@pytest.mark.parametrize("name", sorted(SCHED))
def test_scheduler_fire_pattern(name):
    g = load("schedulers.npz")
    fn = build_fn({"scheduler": SCHED[name], "update": ["NoUpdate", {}]})
    T = g[name].shape[0]
    _, fired, _ = O.theta_trace(fn, 1.0, t0=0, T=T)
    np.testing.assert_array_equal(fired[:, 0], g[name])


_TRACE_KEYS = sorted({k.rsplit("__", 1)[0] for k in load("update_traces.npz").files})


@pytest.mark.parametrize("key", _TRACE_KEYS)
def test_update_trace(key):
    g = load("update_traces.npz")
    uname, sname = key.split("__")
    dist = uname in DIST or uname == "d4_randomcat"
    upd = ["RandomCategorical", {"seed": 21}] if uname == "d4_randomcat" else (DIST if dist else SCALAR)[uname]
    fn = build_fn({"scheduler": SCHED[sname], "update": upd})
    T = g[key + "__fired"].shape[0]
    if dist:
        th0 = [0.4, 0.3, 0.3] if uname == "d_increment" else [1.0, 0.0, 0.0, 0.0] if uname == "d4_randomcat" else [1.0, 0.0, 0.0]
    else:
        th0 = 9.8
    th, fired, delta = O.theta_trace(fn, th0, t0=0, T=T)
    np.testing.assert_array_equal(fired[:, 0], g[key + "__fired"])
    want = g[key + "__theta"]
    got = th[:, :, 0] if dist else th[:, 0]
    transcendental = uname in ("expdecay", "oscillating", "sigmoid") or sname == "decaying"
    if transcendental:   # libm vs NumPy's SIMD exp/sin may differ in the last ulp
        np.testing.assert_allclose(got, want, rtol=1e-13)
        np.testing.assert_allclose(delta[:, 0], g[key + "__delta"], rtol=1e-9, atol=1e-13)
    else:                # everything else is bit-exact, W1 deltas included
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(delta[:, 0], g[key + "__delta"])


def test_survey_probe_values():
    # Increment trajectory and fp64 deltas; W1 of the step function and Decrement
    from ns_gym_amd.schedulers import ContinuousScheduler, DiscreteScheduler
    from ns_gym_amd.update_functions import (DistributionDecrementUpdate, DistributionStepWiseUpdate,
                                             IncrementUpdate)

    th, f, d = O.theta_trace(IncrementUpdate(ContinuousScheduler(), k=0.1), 0.1, T=3)
    assert list(th[:, 0]) == [0.2, 0.30000000000000004, 0.4]
    assert list(d[:, 0]) == [0.1, 0.10000000000000003, 0.09999999999999998]
    fn = DistributionStepWiseUpdate(DiscreteScheduler({50}), [[0.6, 0.2, 0.2]])
    th, f, d = O.theta_trace(fn, [1.0, 0.0, 0.0], T=52)
    assert f[:, 0].sum() == 1 and f[50, 0] == 1 and d[50, 0] == pytest.approx(0.6, abs=1e-15)
    assert list(th[51, :, 0]) == [0.6, 0.2, 0.2]
    th, f, d = O.theta_trace(DistributionDecrementUpdate(ContinuousScheduler(), 0.05), [1.0, 0.0, 0.0], T=1)
    assert d[0, 0] == 0.07500000000000007
