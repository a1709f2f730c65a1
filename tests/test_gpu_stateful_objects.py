"""Host-side Scheduler / UpdateFn objects are STATEFUL like the reference's (ns_gym/base.py:67-81,124-149,
ns_gym/schedulers.py:25-28,107-116): called repeatedly, their streams continue, lists advance, prev_time is
remembered.  The state lives in device tensors (nsg_theta_trace_stateful); expectations are computed with
NumPy's own Generator - the library the reference's objects draw from - and with plain Python arithmetic
re-stating each `_update` (file:line cited per test)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _imp():
    from ns_gym_amd import schedulers as S
    from ns_gym_amd import update_functions as U

    return S, U


def test_random_walk_stream_continues_across_calls():
    S, U = _imp()   # single_param.py:110-113: param + rng.normal(mu, sigma)
    fn = U.RandomWalk(S.ContinuousScheduler(), mu=0.5, sigma=2.0, seed=42)
    rng = np.random.default_rng(42)
    p = 1.0
    for t in range(8):
        want = p + rng.normal(0.5, 2.0)
        got, fired, delta = fn(p, t)
        assert fired == 1 and got == want and delta == want - p
        p = got
    # two objects with the same seed agree; a different seed diverges (test_update_functions.py seed tests)
    a, b, c = (U.RandomWalk(S.ContinuousScheduler(), seed=s) for s in (7, 7, 8))
    xs = [[f(0.0, t)[0] for t in range(4)] for f in (a, b, c)]
    assert xs[0] == xs[1] and xs[0] != xs[2]


def test_seed_method_restarts_the_stream():
    S, U = _imp()   # base.py:151-158
    fn = U.RandomWalk(S.ContinuousScheduler(), seed=1)
    first = [fn(0.0, t)[0] for t in range(3)]
    fn.seed(1)
    assert [fn(0.0, t)[0] for t in range(3)] == first
    fn.seed(99)
    rng = np.random.default_rng(99)
    assert fn(0.0, 0)[0] == 0.0 + rng.normal(0, 1)


def test_stochastic_fn_only_draws_when_its_scheduler_fires():
    S, U = _imp()   # base.py:139-141: _update is called on fired steps only
    fn = U.RandomWalk(S.PeriodicScheduler(period=3), seed=5)
    rng = np.random.default_rng(5)
    p = 9.8
    for t in range(10):
        got, fired, delta = fn(p, t)
        if t % 3 == 0:
            want = p + rng.normal(0, 1)
            assert fired == 1 and got == want
        else:
            assert fired == 0 and got == p and delta == 0.0
        p = got


def test_stepwise_and_cyclic_lists_advance():
    S, U = _imp()   # single_param.py:217-223, 402-408
    fn = U.StepWiseUpdate(S.ContinuousScheduler(), [5.0, 6.0, 7.0])
    outs = [fn(1.0, t) for t in range(5)]
    assert [o[0] for o in outs] == [5.0, 6.0, 7.0, 1.0, 1.0]       # exhausted: value unchanged ...
    assert [o[1] for o in outs] == [1, 1, 1, 1, 1]                 # ... but still reported as fired
    assert outs[3][2] == 0.0
    cyc = U.CyclicUpdate(S.ContinuousScheduler(), [1.0, 2.0, 3.0])
    assert [cyc(0.0, t)[0] for t in range(7)] == [1.0, 2.0, 3.0, 1.0, 2.0, 3.0, 1.0]


def test_ornstein_uhlenbeck_and_bounded_walk_follow_numpy():
    S, U = _imp()
    ou = U.OrnsteinUhlenbeck(S.ContinuousScheduler(), mu=2.0, theta=0.3, sigma=0.1, seed=3)   # single_param.py:345-346
    rng = np.random.default_rng(3)
    p = 0.0
    for t in range(6):
        want = p + 0.3 * (2.0 - p) + 0.1 * rng.normal()
        got = ou(p, t)[0]
        assert abs(got - want) <= 2e-16 * max(1.0, abs(want))
        p = got
    bw = U.BoundedRandomWalk(S.ContinuousScheduler(), mu=0.0, sigma=1.0, lo=-0.5, hi=0.5, seed=4)     # single_param.py:447-448
    rng = np.random.default_rng(4)
    p = 0.0
    for t in range(6):
        want = float(np.clip(p + rng.normal(0, 1.0), -0.5, 0.5))
        got = bw(p, t)[0]
        assert got == want
        p = got


def test_random_scheduler_is_a_stream_of_uniforms():
    S, _ = _imp()   # schedulers.py:27-28: rng.random() < probability, one draw per in-range call
    s = S.RandomScheduler(probability=0.4, seed=7, start=2, end=30)
    rng = np.random.default_rng(7)
    for t in [0, 1, 2, 3, 5, 5, 9, 30, 31, 12]:
        want = (rng.random() < 0.4) if 2 <= t <= 30 else False
        assert s(t) is want
    assert [S.RandomScheduler(probability=1.0, seed=1)(t) for t in range(5)] == [True] * 5
    assert [S.RandomScheduler(probability=0.0, seed=1)(t) for t in range(5)] == [False] * 5


def test_decaying_probability_scheduler_follows_numpy():
    S, _ = _imp()   # schedulers.py:175-177
    s = S.DecayingProbabilityScheduler(initial_probability=0.9, decay_rate=0.2, seed=11)
    rng = np.random.default_rng(11)
    for t in range(25):
        assert s(t) is bool(rng.random() < 0.9 * np.exp(-0.2 * t))


def test_memoryless_scheduler_resamples_after_each_fire():
    S, _ = _imp()   # schedulers.py:104-116
    for p, seed in ((0.3, 2), (0.05, 9), (0.8, 4)):
        s = S.MemorylessScheduler(p=p, seed=seed)
        rng = np.random.default_rng(seed)
        transition = int(rng.geometric(p))
        fires = 0
        for t in range(120):
            want = t == transition
            if want:
                transition = int(rng.geometric(p)) + t
                fires += 1
            assert s(t) is want, (p, seed, t)
        assert fires >= 1


def test_update_fn_with_stochastic_scheduler_keeps_both_streams():
    S, U = _imp()
    fn = U.RandomWalk(S.RandomScheduler(probability=0.5, seed=21), mu=0, sigma=1, seed=22)
    srng, urng = np.random.default_rng(21), np.random.default_rng(22)
    p = 0.0
    for t in range(20):
        fire = srng.random() < 0.5
        want = p + urng.normal(0, 1) if fire else p
        got, fired, _ = fn(p, t)
        assert fired == int(fire) and got == want
        p = got


def test_lcbounded_remembers_prev_time():
    S, U = _imp()   # distribution.py:150-183: the Lipschitz bound scales with t - prev_time
    fn = U.LCBoundedDistrubutionUpdate(S.DiscreteScheduler({0, 1, 7}), L=0.02, seed=5)
    p = [0.6, 0.2, 0.2]
    w1 = []
    for t in range(9):
        q, fired, delta = fn(p, t)
        if fired:
            assert abs(sum(q) - 1.0) < 1e-12
            w1.append((t, delta))
        p = q
    assert [t for t, _ in w1] == [0, 1, 7]
    assert all(d <= 0.02 * 1 + 1e-12 for t, d in w1)   # prev_time is recorded on EVERY call (base.py:143-148): dt == 1
