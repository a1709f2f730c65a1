"""Conditions for which the reference RAISES inside step(): the kernels cannot raise, they count, and the host raises when it polls.

  * LCBoundedDistrubutionUpdate: no candidate within the Lipschitz bound after 1e5 tries -> ValueError
    (ns_gym/update_functions/distribution.py:168-182).
  * CustomScheduler: the reference calls event_function(t) for ANY t (ns_gym/schedulers.py:31-43); here the callable is sampled
    over 2 x TimeLimit (or horizon=), which covers a planning copy taken late in an episode (classic_control.py:168-180: the
    copy keeps t, its TimeLimit restarts); a t beyond the table is counted and reported, never answered silently."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_lcbounded_exhaustion_raises_like_the_reference():
    import torch

    from ns_gym_amd import make
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.update_functions import LCBoundedDistrubutionUpdate
    from ns_gym_amd.vec_env import VecNSEnv
    from ns_gym_amd.wrappers import NSFrozenLakeWrapper

    # L = 0: the bound is W1 <= 0, which no Dirichlet draw meets -> every fire exhausts its 1e5 tries
    mk = lambda: {"P": LCBoundedDistrubutionUpdate(ContinuousScheduler(), L=0.0)}  # noqa: E731
    env = VecNSEnv(make("FrozenLake-v1"), mk(), 64, initial_prob_dist=[1.0, 0.0, 0.0])
    assert env.may_raise
    env.reset(seed=0)
    env.check_errors()                                   # nothing yet
    env.step(torch.zeros(64, dtype=torch.int32, device="cuda"))
    c = env.counters()
    assert c["lc_exhausted"] == 64 and c["scheduler_overruns"] == 0
    np.testing.assert_array_equal(env.theta.cpu().numpy(), np.repeat([[1.0], [0.0], [0.0]], 64, axis=1))   # left unchanged
    with pytest.raises(ValueError, match="Lipschitz-continuous update after 100000 attempts"):
        env.check_errors()
    env.check_errors()                                   # reported once
    env.close()
    # the N = 1 adaptor raises from step(), where the reference does
    single = NSFrozenLakeWrapper(make("FrozenLake-v1"), mk(), initial_prob_dist=[1.0, 0.0, 0.0])
    single.reset(seed=0)
    with pytest.raises(ValueError, match="Lipschitz-continuous"):
        single.step(0)
    # and so does the host-side object called directly: fn(param, t) (base.py:124-149 -> distribution.py:168-182)
    fn = LCBoundedDistrubutionUpdate(ContinuousScheduler(), L=0.0)
    with pytest.raises(ValueError, match="Lipschitz-continuous"):
        fn([1.0, 0.0, 0.0], 0)
    # a satisfiable bound does not raise
    ok = VecNSEnv(make("FrozenLake-v1"), {"P": LCBoundedDistrubutionUpdate(ContinuousScheduler(), L=0.5)}, 64, initial_prob_dist=[1.0, 0.0, 0.0])
    ok.reset(seed=0)
    for _ in range(5):
        ok.step(torch.zeros(64, dtype=torch.int32, device="cuda"))
    ok.check_errors()
    assert ok.counters()["lc_exhausted"] == 0
    ok.close()


def test_custom_scheduler_in_a_late_planning_copy_and_beyond_its_table():
    import torch

    from ns_gym_amd import make
    from ns_gym_amd.schedulers import CustomScheduler
    from ns_gym_amd.update_functions import IncrementUpdate
    from ns_gym_amd.vec_env import VecNSEnv

    event = lambda t: t % 7 == 3 or t in (199, 200, 201, 348)  # noqa: E731
    n = 300
    # Pendulum never terminates: TimeLimit 200 -> the callable is sampled over t = 0 .. 400
    env = VecNSEnv(make("Pendulum-v1"), {"m": IncrementUpdate(CustomScheduler(event), k=0.01)}, n, change_notification=True,
                   delta_change_notification=True, in_sim_change=True)
    assert env.may_raise and env.cfg.params[0].sched_tab_len == 401
    env.reset(seed=0)
    act = torch.zeros(n, dtype=torch.float32, device="cuda")
    for t in range(150):
        env.step(act)
        assert bool(env.gt_env_change[0, 0].item()) == bool(event(t)), t
    late = env.fork(theta_mode=0, entropy=1)              # t = 150 carried over, the copy's TimeLimit restarts: runs to t = 349
    fired = []
    for k in range(199):
        late.step(act)
        fired.append(bool(late.gt_env_change[0, 0].item()))
    assert int(late.t[0].item()) == 349 and not bool(late.truncated[0].item())
    assert fired == [bool(event(t)) for t in range(150, 349)]          # the reference would have called event(t) for these t
    late.check_errors()                                               # all inside the table
    again = late.fork(theta_mode=0, entropy=2)            # a copy of the copy: t = 349 + up to 199 -> beyond 400
    for k in range(60):
        again.step(act)
    assert int(again.t[0].item()) == 409
    c = again.counters()
    assert c["scheduler_overruns"] == n * (409 - 401)     # t = 401 .. 408 were asked about and not answered
    with pytest.raises(ValueError, match="beyond the horizon"):
        again.check_errors()
    # an explicit horizon removes the limit
    wide = VecNSEnv(make("Pendulum-v1"), {"m": IncrementUpdate(CustomScheduler(event, horizon=2000), k=0.01)}, 8, in_sim_change=True)
    assert wide.cfg.params[0].sched_tab_len == 2001
    for e in (env, late, again, wide):
        e.close()


def test_host_side_custom_scheduler_call_reaches_any_t():
    from ns_gym_amd.schedulers import CustomScheduler

    s = CustomScheduler(lambda t: t in (5, 5000, 70000))
    assert [t for t in (4, 5, 4999, 5000, 70000) if s(t)] == [5, 5000, 70000]
