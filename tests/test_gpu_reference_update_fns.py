"""Known-answer tests per Scheduler / UpdateFn class in the style of the reference's
tests/test_schedulers.py and tests/test_update_functions.py: every `_update` / `_check` re-stated as one
line of plain Python (reference file:line beside it) and compared with what the host-side objects return -
i.e. with what the device θ-engine computes.  (The T = 64 traces generated from the reference classes
themselves are in tests/golden/update_traces.npz / schedulers.npz; this file is the readable counterpart.)"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _imp():
    from ns_gym_amd import schedulers as S
    from ns_gym_amd import update_functions as U

    return S, U


def on():
    S, _ = _imp()
    return S.ContinuousScheduler()


def off():
    S, _ = _imp()
    return S.ContinuousScheduler(start=1000, end=2000)


# ------------------------------------------------------------------ schedulers (ns_gym/schedulers.py)
def test_continuous_scheduler():
    S, _ = _imp()   # :46-53 + range gate base.py:67-81
    s = S.ContinuousScheduler(start=3, end=6)
    assert [s(t) for t in range(9)] == [False] * 3 + [True] * 4 + [False] * 2
    assert S.ContinuousScheduler()(10 ** 6) is True
    assert S.ContinuousScheduler(start=5, end=5)(5) is True
    assert isinstance(s(4), bool) and s(2) is False


def test_discrete_scheduler():
    S, _ = _imp()   # :56-74
    s = S.DiscreteScheduler({2, 5, 11})
    assert [t for t in range(15) if s(t)] == [2, 5, 11]
    with pytest.raises(AssertionError):
        S.DiscreteScheduler({1, 4}, start=2)
    with pytest.raises(AssertionError):
        S.DiscreteScheduler({1, 40}, end=20)
    assert S.DiscreteScheduler({0, 9}, start=0, end=9)(9) is True


def test_periodic_scheduler():
    S, _ = _imp()   # :77-89
    s = S.PeriodicScheduler(period=4)
    assert [t for t in range(13) if s(t)] == [0, 4, 8, 12]
    assert all(S.PeriodicScheduler(period=1)(t) for t in range(5))
    s = S.PeriodicScheduler(period=3, start=4, end=10)
    assert [t for t in range(15) if s(t)] == [6, 9]
    assert S.PeriodicScheduler(period=1000)(3000) is True and S.PeriodicScheduler(period=1000)(2999) is False


def test_burst_scheduler():
    S, _ = _imp()   # :119-140
    s = S.BurstScheduler(on_duration=2, off_duration=3)
    assert [int(s(t)) for t in range(10)] == [1, 1, 0, 0, 0, 1, 1, 0, 0, 0]
    assert [int(S.BurstScheduler(1, 1)(t)) for t in range(6)] == [1, 0, 1, 0, 1, 0]
    assert [int(S.BurstScheduler(3, 1, start=2, end=8)(t)) for t in range(10)] == [0, 0, 1, 0, 1, 1, 1, 0, 1, 0]
    assert S.BurstScheduler(2, 3)(10 ** 6) is True     # 10^6 % 5 == 0


def test_window_scheduler():
    S, _ = _imp()   # :180-198
    s = S.WindowScheduler([(2, 4), (8, 8), (10, 11)])
    assert [t for t in range(14) if s(t)] == [2, 3, 4, 8, 10, 11]
    assert not any(S.WindowScheduler([])(t) for t in range(10))
    assert [t for t in range(14) if S.WindowScheduler([(0, 12)], start=5, end=7)(t)] == [5, 6, 7]


def test_custom_scheduler():
    S, _ = _imp()   # :31-43
    assert [t for t in range(10) if S.CustomScheduler(lambda t: t % 2 == 1)(t)] == [1, 3, 5, 7, 9]
    assert not any(S.CustomScheduler(lambda t: False)(t) for t in range(5))
    assert [t for t in range(10) if S.CustomScheduler(lambda t: True, start=3, end=5)(t)] == [3, 4, 5]


# ------------------------------------------------------------------ scalar update fns (update_functions/single_param.py)
def test_no_update_and_call_interface():
    S, U = _imp()
    out = U.NoUpdate(on())(3.5, 0)                      # :239-240; 3-tuple (param, fired, delta), base.py:124-149
    assert out == (3.5, 1, 0.0)
    assert U.IncrementUpdate(off(), k=1.0)(3.5, 0) == (3.5, 0, 0.0)
    fn = U.IncrementUpdate(on(), k=0.25)
    fn(1.0, 4)
    assert fn.prev_param == 1.0 and fn.prev_time == 4   # base.py:143-148
    with pytest.raises(AssertionError):
        U.IncrementUpdate("not a scheduler", k=1.0)
    with pytest.raises(AssertionError):
        fn(1.0, np.int64(3))                            # base.py:136-138: NumPy ints are rejected
    with pytest.raises(AssertionError):
        U.DistributionNoUpdate(on())(0.5, 0)            # base.py:189: distributions must be lists


def test_increment_decrement_trend():
    _, U = _imp()
    assert U.IncrementUpdate(on(), k=0.1)(0.2, 0) == (0.2 + 0.1, 1, (0.2 + 0.1) - 0.2)     # :173-175
    assert U.IncrementUpdate(on(), k=-2.0)(1.0, 0)[0] == -1.0
    p = 0.0
    fn = U.IncrementUpdate(on(), k=0.1)
    for t in range(5):
        p = fn(p, t)[0]
    assert p == 0.1 + 0.1 + 0.1 + 0.1 + 0.1             # float accumulation like the reference: 0.5 exactly?
    assert U.DecrementUpdate(on(), k=0.3)(1.0, 0)[0] == 1.0 - 0.3                           # :197-199
    assert U.DeterministicTrend(on(), slope=0.5)(2.0, 3)[0] == 2.0 + 0.5 * 3                # :38-40
    assert U.DeterministicTrend(on(), slope=0.5)(2.0, 0) == (2.0, 1, 0.0)
    assert U.DeterministicTrend(on(), slope=-1.5)(2.0, 2)[0] == 2.0 - 3.0


def test_geometric_polynomial_interpolation():
    _, U = _imp()
    assert U.GeometricProgression(on(), r=1.5)(2.0, 0)[0] == 3.0                             # :305-307
    assert U.GeometricProgression(on(), r=0.5)(2.0, 7)[0] == 1.0
    for coeffs, t in (([2.0], 3), ([0.0, 1.0], 4), ([0.0, 0.0, 1.0], 2), ([1.0, -0.5, 0.25], 3), ([1.0, 2.0], 0)):
        want = 1.0 + sum(a * t ** (i + 1) for i, a in enumerate(coeffs))                     # :471-473
        assert U.PolynomialTrend(on(), coeffs)(1.0, t)[0] == want, (coeffs, t)
    li = lambda: U.LinearInterpolation(on(), start_val=2.0, end_val=10.0, T=8)              # noqa: E731  :506-508
    assert [li()(0.0, t)[0] for t in (0, 4, 8, 20)] == [2.0, 6.0, 10.0, 10.0]
    assert U.LinearInterpolation(on(), 10.0, 2.0, 4)(0.0, 1)[0] == 10.0 + (2.0 - 10.0) * 0.25


def test_transcendental_updates_within_one_ulp():
    _, U = _imp()
    for t in (0, 1, 2, 7):
        got = U.OscillatingUpdate(on(), delta=0.5)(1.0, t)[0]                                # :262-264
        assert abs(got - (1.0 + 0.5 * math.sin(t))) <= 2.3e-16
        got = U.ExponentialDecay(on(), decay_rate=0.3)(4.0, t)[0]                            # :285-287
        assert abs(got - 4.0 * math.exp(-0.3 * t)) <= 4.5e-16 * 4.0
    sg = lambda t: U.SigmoidTransition(on(), a=1.0, b=3.0, k=2.0, t0=5.0)(0.0, t)[0]        # noqa: E731  :383-385
    for t in (0, 5, 9, 50):
        assert abs(sg(t) - (1.0 + 2.0 * (1.0 / (1.0 + math.exp(-2.0 * (t - 5.0)))))) <= 4.5e-16
    assert sg(5) == 2.0 and abs(sg(0) - 1.0) < 1e-4 and abs(sg(50) - 3.0) < 1e-12


def test_noise_free_limits_of_the_stochastic_updates():
    _, U = _imp()
    assert U.RandomWalk(on(), mu=0, sigma=0, seed=1)(2.0, 0)[0] == 2.0                       # :110-113
    assert U.RandomWalkWithDrift(on(), alpha=0.5, mu=0, sigma=0, seed=1)(2.0, 0)[0] == 2.5   # :148-151
    assert U.RandomWalkWithDriftAndTrend(on(), alpha=0.5, mu=0, sigma=0, slope=0.25, seed=1)(2.0, 4)[0] == 0.5 + 2.0 + 0.0 + 0.25 * 4  # :78-81
    ou = lambda p: U.OrnsteinUhlenbeck(on(), theta=0.5, mu=1.0, sigma=0.0)(p, 0)[0]         # noqa: E731  :344-346
    assert ou(3.0) == 3.0 + 0.5 * (1.0 - 3.0) and ou(-1.0) == -1.0 + 0.5 * 2.0 and ou(1.0) == 1.0
    b = U.BoundedRandomWalk(on(), mu=0, sigma=5.0, lo=-1.0, hi=1.0, seed=3)                  # :446-448
    p = 0.0
    for t in range(40):
        p = b(p, t)[0]
        assert -1.0 <= p <= 1.0
    assert U.BoundedRandomWalk(on(), mu=100.0, sigma=0.0, lo=0.0, hi=1.0, seed=1)(0.5, 0)[0] == 1.0
    assert U.BoundedRandomWalk(on(), mu=-100.0, sigma=0.0, lo=0.0, hi=1.0, seed=1)(0.5, 0)[0] == 0.0


@pytest.mark.parametrize("make", [
    lambda U: U.IncrementUpdate(off(), k=1.0), lambda U: U.DecrementUpdate(off(), k=1.0), lambda U: U.DeterministicTrend(off(), slope=1.0),
    lambda U: U.RandomWalk(off(), seed=1), lambda U: U.StepWiseUpdate(off(), [9.0]), lambda U: U.OrnsteinUhlenbeck(off(), 0.5, 1.0, 0.1, seed=1),
    lambda U: U.SigmoidTransition(off(), 0.0, 1.0, 1.0, 0.0), lambda U: U.CyclicUpdate(off(), [9.0]),
    lambda U: U.BoundedRandomWalk(off(), 0.0, 1.0, -1.0, 1.0, seed=1), lambda U: U.PolynomialTrend(off(), [1.0]),
    lambda U: U.LinearInterpolation(off(), 0.0, 1.0, 4)])
def test_no_update_when_scheduler_false(make):
    _, U = _imp()
    assert make(U)(1.25, 3) == (1.25, 0, 0.0)


# ------------------------------------------------------------------ distribution update fns (update_functions/distribution.py)
def test_distribution_increment_decrement():
    _, U = _imp()
    q, fired, delta = U.DistributionIncrementUpdate(on(), k=0.1)([0.5, 0.25, 0.25], 0)       # :61-67
    p0 = min(1, 0.5 + 0.1)
    assert q == [p0, (1 - p0) / 2, (1 - p0) / 2] and fired == 1
    assert abs(delta - (abs(0.5 - p0) + abs(0.75 - (p0 + (1 - p0) / 2)))) < 1e-15            # W1 over {0,1,2}, utils.py:55-94
    assert U.DistributionIncrementUpdate(on(), k=0.9)([0.5, 0.25, 0.25], 0)[0] == [1, 0.0, 0.0]
    q = U.DistributionDecrementUpdate(on(), k=0.2)([0.5, 0.25, 0.25], 0)[0]                  # :88-97
    assert q == [0.5 - 0.2, (1 - (0.5 - 0.2)) / 2, (1 - (0.5 - 0.2)) / 2]
    assert U.DistributionDecrementUpdate(on(), k=0.9)([0.5, 0.25, 0.25], 0)[0] == [0, 0.5, 0.5]
    q4 = U.DistributionDecrementUpdate(on(), k=0.1)([0.7, 0.1, 0.1, 0.1], 0)[0]              # CliffWalking's 4-way support
    assert q4 == [0.7 - 0.1] + [(1 - (0.7 - 0.1)) / 3] * 3


def test_distribution_stepwise_cyclic_noupdate():
    _, U = _imp()
    sw = U.DistributionStepWiseUpdate(on(), [[0.6, 0.2, 0.2], [0.2, 0.4, 0.4]])             # :116-130
    p = [1.0, 0.0, 0.0]
    seen = []
    for t in range(4):
        p, fired, delta = sw(p, t)
        seen.append((list(p), fired))
    assert [s[0] for s in seen] == [[0.6, 0.2, 0.2], [0.2, 0.4, 0.4], [0.2, 0.4, 0.4], [0.2, 0.4, 0.4]]
    assert [s[1] for s in seen] == [1, 1, 1, 1]
    assert U.DistributionStepWiseUpdate(on(), [[0.6, 0.2, 0.2]])([1.0, 0.0, 0.0], 0)[2] == 0.6   # W1 = 0.6
    cy = U.DistributionCyclicUpdate(on(), [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])               # :353-356
    assert [cy([0.3, 0.3, 0.4], t)[0] for t in range(3)] == [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0]]
    assert U.DistributionNoUpdate(on())([0.2, 0.3, 0.5], 0) == ([0.2, 0.3, 0.5], 1, 0.0)     # :230-231


def test_uniform_drift_target_reversion_interpolation():
    _, U = _imp()
    p = [0.7, 0.2, 0.1]
    assert U.UniformDrift(on(), rate=0.25)(list(p), 0)[0] == [(1 - 0.25) * x + 0.25 * (1.0 / 3) for x in p]   # :256-261
    assert U.UniformDrift(on(), rate=0.0)(list(p), 0)[0] == p
    assert U.UniformDrift(on(), rate=1.0)(list(p), 0)[0] == [1.0 / 3] * 3
    tgt = [0.1, 0.1, 0.8]
    assert U.TargetReversion(on(), target=tgt, theta=0.5)(list(p), 0)[0] == [x + 0.5 * (g - x) for x, g in zip(p, tgt)]  # :289-293
    assert U.TargetReversion(on(), target=tgt, theta=1.0)(list(p), 0)[0] == [x + 1.0 * (g - x) for x, g in zip(p, tgt)]
    assert U.TargetReversion(on(), target=tgt, theta=0.0)(list(p), 0)[0] == p
    a, b = [1.0, 0.0, 0.0], [0.2, 0.4, 0.4]
    for t, frac in ((0, 0.0), (2, 0.5), (4, 1.0), (9, 1.0)):                                 # :326-331
        assert U.DistributionLinearInterpolation(on(), a, b, T=4)([0.3, 0.3, 0.4], t)[0] == [s + (e - s) * frac for s, e in zip(a, b)]
    q = U.UniformDrift(on(), rate=0.3)([0.25, 0.25, 0.25, 0.25], 0)[0]
    assert len(q) == 4 and abs(sum(q) - 1.0) < 1e-15


def test_random_categorical_follows_numpy_dirichlet():
    _, U = _imp()   # :28-38: list(rng.dirichlet(np.ones(len(param))))
    fn = U.RandomCategorical(on(), seed=13)
    rng = np.random.default_rng(13)
    p = [1.0, 0.0, 0.0]
    for t in range(5):
        want = list(rng.dirichlet(np.ones(3)))
        q, fired, delta = fn(p, t)
        assert fired == 1 and np.allclose(q, want, rtol=0, atol=4e-16) and abs(sum(q) - 1.0) < 1e-15
        p = q
    a = [U.RandomCategorical(on(), seed=2)([0.5, 0.5, 0.0], 0)[0] for _ in range(2)]
    assert a[0] == a[1] and len(a[0]) == 3


@pytest.mark.parametrize("make", [
    lambda U: U.DistributionIncrementUpdate(off(), 0.1), lambda U: U.DistributionDecrementUpdate(off(), 0.1),
    lambda U: U.UniformDrift(off(), 0.5), lambda U: U.TargetReversion(off(), [0.1, 0.1, 0.8], 0.5),
    lambda U: U.DistributionLinearInterpolation(off(), [1.0, 0.0, 0.0], [0.0, 0.0, 1.0], 4),
    lambda U: U.DistributionCyclicUpdate(off(), [[0.0, 0.0, 1.0]])])
def test_distribution_no_update_when_scheduler_false(make):
    _, U = _imp()
    assert make(U)([0.5, 0.25, 0.25], 3) == ([0.5, 0.25, 0.25], 0, 0.0)
