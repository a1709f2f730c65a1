"""User-defined Scheduler / UpdateFn subclasses (the reference's extension idiom: ns_gym/base.py:50-203, tutorial.ipynb cells
38-44) through the KERNELS: the fixtures were produced by the REFERENCE wrappers driving the very same classes
(tests/golden/user_plugins.py built on ns_gym.base; make_golden.py USER_SPECS); here the classes are built on ns_gym_amd.base,
sampled into tables (ns_gym_amd.extension) and stepped on the GPU - as the batch of the fixture's seeds, as N = 1 adaptors, and
scattered inside a 4096-env batch whose other envs are checked against the oracle."""
import numpy as np
import pytest
import torch

import ns_gym_amd.base as base
from tests.golden import user_plugins
from tests.util import USER_SPECS, GpuView, OracleView, build_params, check_trajectory, compare_views, load, make_env_from_spec

pytestmark = pytest.mark.gpu

P = user_plugins.plugin_classes(base)


def _vec(*a, **k):
    from ns_gym_amd.vec_env import VecNSEnv

    return VecNSEnv(*a, **k)


def _strict(spec):
    # distributions and every θ that involves only + - * / are bit-exact (the tabled proposals are the reference's own numbers);
    # the CartPole fixture also carries a RandomWalk (ziggurat wedge / tail through device exp / log1p: a few ulp)
    return spec["env_id"] == "FrozenLake-v1"


@pytest.mark.parametrize("name", sorted(USER_SPECS))
@pytest.mark.parametrize("specialize", [False, True])
def test_reference_trajectories_with_user_subclasses(name, specialize):
    spec = USER_SPECS[name]
    env = make_env_from_spec(_vec, spec, specialize=specialize)
    assert env.specialized == specialize
    check_trajectory(GpuView(env), spec, load(f"traj_{name}.npz"), strict_theta=_strict(spec))
    assert env.counters()["scheduler_overruns"] == 0
    env.close()


@pytest.mark.parametrize("name", ["user_cartpole_every5_custom", "user_frozenlake_oscillating"])
def test_n1_adaptors_with_user_subclasses(name):
    """The reference's own calling convention: one wrapper object, Python scalars, `reset()` after `done`."""
    from ns_gym_amd import make
    from ns_gym_amd.wrappers import NSClassicControlWrapper, NSFrozenLakeWrapper

    spec, rec = USER_SPECS[name], load(f"traj_{name}.npz")
    is_fl = spec["env_id"] == "FrozenLake-v1"
    pnames = list(spec["params"])
    T = 150
    for i, seed in enumerate(spec["seeds"][:2]):
        W = NSFrozenLakeWrapper if is_fl else NSClassicControlWrapper
        env = W(make(spec["env_id"], **spec.get("make_kwargs", {})), build_params(spec["params"]), **spec["flags"], **spec.get("wrapper_kwargs", {}))
        obs, info = env.reset(seed=int(seed))
        need_reset = False
        for k in range(T):
            if need_reset:
                obs, info = env.reset()
                r, term, trunc = 0.0, False, False
            else:
                obs, r, term, trunc, info = env.step(int(rec["actions"][k, i]))
            tag = f"seed {seed} step {k}"
            if is_fl:
                assert obs["state"] == rec["state"][k + 1, i, 0], tag
                assert [float(x) for x in env.transition_prob] == list(rec["theta"][k + 1, i]), tag
            else:
                np.testing.assert_allclose(obs["state"], rec["state"][k + 1, i], rtol=1e-5, atol=1e-5, err_msg=tag)
                got = [getattr(env.unwrapped, p) for p in pnames]
                np.testing.assert_allclose(got, rec["theta"][k + 1, i], rtol=1e-12, err_msg=tag)
            assert [obs["env_change"][p] for p in pnames] == list(rec["env_change"][k + 1, i]), tag
            np.testing.assert_allclose([obs["delta_change"][p] for p in pnames], rec["delta_change"][k + 1, i], rtol=1e-6, atol=1e-7, err_msg=tag)
            assert obs["relative_time"] == rec["relative_time"][k + 1, i], tag
            assert (r, term, trunc) == (rec["reward"][k, i], bool(rec["terminated"][k, i]), bool(rec["truncated"][k, i])), tag
            need_reset = term or trunc
        env.close()


@pytest.mark.parametrize("name", sorted(USER_SPECS))
def test_fixture_envs_inside_a_4096_env_batch(name):
    """The fixture's envs sit at scattered indices of a 4096-env batch (arbitrary per-env seeds); their rows must be the
    reference's, everybody else's the oracle's - every env, every step."""
    from oracle.oracle import OracleVecEnv

    spec, rec = USER_SPECS[name], load(f"traj_{name}.npz")
    is_fl = spec["env_id"] == "FrozenLake-v1"
    n, T = 4096, 200
    rng = np.random.default_rng(7)
    seeds = rng.integers(1000, 2**40, size=n).astype(np.uint64)
    where = np.sort(rng.choice(n, size=len(spec["seeds"]), replace=False))
    seeds[where] = np.asarray(spec["seeds"], dtype=np.uint64)
    from tests.golden.make_golden import make_actions

    acts = make_actions(spec["env_id"], T, n).copy()
    acts[:, where] = rec["actions"][:T]
    g = GpuView(make_env_from_spec(_vec, spec, n=n))
    o = OracleView(make_env_from_spec(OracleVecEnv, spec, n=n))
    compare_views(g.reset(seeds), o.reset(seeds), is_fl, "reset")
    for k in range(T):
        a, b = g.step(acts[k]), o.step(acts[k])
        compare_views(a, b, is_fl, f"step {k}")
        tag = f"fixture envs, step {k}"
        if is_fl:
            np.testing.assert_array_equal(a["state"].reshape(-1)[where], rec["state"][k + 1, :, 0], err_msg=tag)
            np.testing.assert_array_equal(a["theta"][:, where].T, rec["theta"][k + 1], err_msg=tag)
        else:
            np.testing.assert_allclose(a["state"][where], rec["state"][k + 1], rtol=1e-5, atol=1e-5, err_msg=tag)
            np.testing.assert_allclose(a["theta"][:, where].T, rec["theta"][k + 1], rtol=1e-12, err_msg=tag)
        np.testing.assert_array_equal(a["env_change"][:, where].T, rec["gt_env_change"][k + 1], err_msg=tag)
        np.testing.assert_array_equal(a["terminated"][where], rec["terminated"][k], err_msg=tag)
        np.testing.assert_array_equal(a["truncated"][where], rec["truncated"][k], err_msg=tag)
        np.testing.assert_array_equal(a["t"][where], rec["relative_time"][k + 1], err_msg=tag)
    g.env.close()


def test_fused_rollout_equals_steps_for_a_user_config():
    from tests.golden.make_golden import make_actions

    spec = USER_SPECS["user_cartpole_every5_custom"]
    n, K = 3000, 96
    acts = torch.from_numpy(make_actions(spec["env_id"], K, n)).cuda()
    a, b = make_env_from_spec(_vec, spec, n=n), make_env_from_spec(_vec, spec, n=n)
    a.reset(seed=11); b.reset(seed=11)
    out = a.rollout(acts, record=("obs", "reward", "terminated", "env_change", "delta_change"))
    for k in range(K):
        b.step(acts[k])
        assert torch.equal(out["obs"][k], b.state) and torch.equal(out["env_change"][k][: b.cfg.n_params], b.gt_env_change), k
        assert torch.equal(out["delta_change"][k][: b.cfg.n_params], b.gt_delta_change), k
    for row in ("phys", "theta", "t", "cursor", "rng_upd", "episode"):
        assert torch.equal(a.buf[row], b.buf[row]), row
    a.close(); b.close()


def test_planning_copies_of_a_user_config():
    """A copy shares the SOURCE's sampled tables (nothing is compiled again, whatever happened to the Python objects since), a
    frozen copy keeps θ, an in_sim_change copy continues the chain; a copy that restarts from the initial θ and keeps evolving
    is refused with the reason."""
    from ns_gym_amd import make
    from ns_gym_amd._lib import NsgError
    from oracle.oracle import OracleVecEnv

    sched = P["EveryNthCall"](3)
    tp = lambda s: {"masscart": P["Sawtooth"](P["Every"](every=5)), "length": P["Momentum"](s, k=0.002)}  # noqa: E731
    n = 512
    kw = dict(change_notification=True, delta_change_notification=True, in_sim_change=True)
    params = tp(sched)
    g = _vec(make("CartPole-v1"), params, n, **kw)
    o = OracleVecEnv(make("CartPole-v1"), tp(P["EveryNthCall"](3)), n, **kw)
    g.reset(seed=5); o.reset(seed=5)
    acts = np.random.default_rng(1).integers(2, size=(40, n)).astype(np.int32)
    for k in range(12):
        g.step(torch.from_numpy(acts[k])); o.step(acts[k])
    sched.calls = 12345          # the user keeps playing with the object: the copy must not care
    gc, oc = g.fork(theta_mode=0, entropy=99), o.fork(theta_mode=0, entropy=99)
    assert bytes(gc.tables) == bytes(g.tables)
    for k in range(12, 40):
        gc.step(torch.from_numpy(acts[k])); oc.step(acts[k])
        np.testing.assert_array_equal(gc.theta.cpu().numpy(), oc.a["theta"], err_msg=f"copy step {k}")
        np.testing.assert_array_equal(gc.gt_env_change.cpu().numpy(), oc.a["env_change"][:2], err_msg=f"copy step {k}")
        np.testing.assert_allclose(gc.state.cpu().numpy(), oc.a["obs"], rtol=1e-5, atol=1e-5)
    with pytest.raises(NsgError, match="restarts from the initial θ.*keeps.*evolving"):
        g.fork(theta_mode=1)
    frozen_src = _vec(make("CartPole-v1"), tp(P["EveryNthCall"](3)), n, change_notification=True)   # in_sim_change=False: frozen copies, any θ
    frozen_src.reset(seed=5)
    for k in range(12):
        frozen_src.step(torch.from_numpy(acts[k]))
    fc = frozen_src.get_planning_env()     # no delta notification: the initial θ, frozen
    th = fc.theta.clone()
    for k in range(12, 30):
        fc.step(torch.from_numpy(acts[k]))
    assert torch.equal(fc.theta, th) and float(th[0, 0]) == 1.0 and int(fc.gt_env_change.sum()) == 0
    for e in (gc, g, fc, frozen_src):
        e.close()


def test_a_copy_that_outruns_the_sampled_horizon_is_reported_not_answered():
    from ns_gym_amd import make

    fn = P["OscillatingSlip"](P["Every"](every=2), head=1.0)     # stays one-hot: the walker goes UP and never terminates
    fn.nsg_horizon = 30
    env = _vec(make("CliffWalking-v1"), {"P": fn}, 64, initial_prob_dist=[1.0, 0.0, 0.0, 0.0])
    env.reset(seed=0)
    acts = torch.zeros(64, dtype=torch.int32, device="cuda")
    for _ in range(31):
        env.step(acts)
    env.check_errors()                      # t = 0 .. 30: all inside the table
    th = env.theta.clone()
    env.step(acts)                          # t = 31: beyond it (CliffWalking has no TimeLimit; nothing terminated: action 0 = UP)
    assert torch.equal(env.theta, th)
    with pytest.raises(ValueError, match="beyond the horizon"):
        env.check_errors()
    env.close()
