"""autoreset=False on the GPU: the kernels follow the REFERENCE wrappers far past their first `done` (fixtures:
make_golden.py NORESET_SPECS - the reference stepped without any reset), as a batch, through the fused rollout, through the
N = 1 adaptors (whose default this is), and at scale against the oracle."""
import numpy as np
import pytest
import torch

from tests.test_oracle_grid import check_grid
from tests.util import MANIFEST, GpuView, OracleView, build_params, check_trajectory, compare_views, load, make_env_from_spec

pytestmark = pytest.mark.gpu

NORESET = MANIFEST["noreset_specs"]
NORESET_GRID = MANIFEST["noreset_grid_specs"]


def _vec(*a, **k):
    from ns_gym_amd.vec_env import VecNSEnv

    return VecNSEnv(*a, **k)


@pytest.mark.parametrize("name", sorted(NORESET))
@pytest.mark.parametrize("specialize", [False, True])
def test_kernels_follow_the_reference_past_done(name, specialize):
    spec = NORESET[name]
    env = make_env_from_spec(_vec, spec, autoreset=False, specialize=specialize)
    check_trajectory(GpuView(env), spec, load(f"traj_{name}.npz"), strict_theta=spec["env_id"] == "FrozenLake-v1")
    env.close()


@pytest.mark.parametrize("name", sorted(NORESET_GRID))
def test_grid_kernels_follow_the_reference_past_done(name):
    spec = NORESET_GRID[name]
    env = make_env_from_spec(_vec, spec, autoreset=False)
    check_grid(GpuView(env), spec, load(f"grid_{name}.npz"))
    env.close()


@pytest.mark.parametrize("name", ["noreset_cartpole", "noreset_frozenlake_4x4", "noreset_pendulum"])
def test_single_wrappers_never_reset_on_their_own(name):
    """The N = 1 adaptors: the reference's calling convention, no reset() after `done`, no warning - the reference's own tests
    step like this (tests/test_step_reset.py:570-577, 722-752)."""
    import warnings

    from ns_gym_amd import make
    from ns_gym_amd.wrappers import NSClassicControlWrapper, NSFrozenLakeWrapper

    spec, rec = NORESET[name], load(f"traj_{name}.npz")
    is_fl = spec["env_id"] == "FrozenLake-v1"
    cont = spec["env_id"] == "Pendulum-v1"
    pnames = list(spec["params"])
    W = NSFrozenLakeWrapper if is_fl else NSClassicControlWrapper
    env = W(make(spec["env_id"], **spec.get("make_kwargs", {})), build_params(spec["params"]), **spec["flags"], **spec.get("wrapper_kwargs", {}))
    assert env._vec.autoreset is False
    obs, info = env.reset(seed=int(spec["seeds"][0]))
    with warnings.catch_warnings():
        warnings.simplefilter("error", UserWarning)      # (a ConstraintViolationWarning is not a UserWarning)
        for k in range(spec["T"]):
            a = rec["actions"][k, 0]
            obs, r, term, trunc, info = env.step(np.array([a], dtype=np.float32) if cont else int(a))
            tag = f"step {k}"
            if is_fl:
                assert obs["state"] == rec["state"][k + 1, 0, 0], tag
            else:
                np.testing.assert_allclose(obs["state"], rec["state"][k + 1, 0], rtol=1e-5, atol=1e-5, err_msg=tag)
            assert obs["relative_time"] == rec["relative_time"][k + 1, 0] == k + 1, tag
            assert (term, trunc) == (bool(rec["terminated"][k, 0]), bool(rec["truncated"][k, 0])), tag
            np.testing.assert_allclose(r, rec["reward"][k, 0], rtol=1e-5, atol=1e-6, err_msg=tag)
            assert [obs["env_change"][p] for p in pnames] == list(rec["env_change"][k + 1, 0]), tag
    # reset() starts a new episode: t = 0, θ back to its initial value, flags clear
    obs, info = env.reset()
    assert obs["relative_time"] == 0 and env.t == 0
    if not is_fl:
        assert getattr(env.unwrapped, pnames[0]) == env.initial_params[pnames[0]]
    env.close()


def test_fused_rollout_equals_steps_without_autoreset():
    from tests.golden.make_golden import make_actions

    for name in ("noreset_cartpole", "noreset_frozenlake_4x4"):
        spec = NORESET[name]
        n, K = 2500, 80
        acts = torch.from_numpy(make_actions(spec["env_id"], K, n)).cuda()
        a = make_env_from_spec(_vec, spec, n=n, autoreset=False)
        b = make_env_from_spec(_vec, spec, n=n, autoreset=False)
        a.reset(seed=3); b.reset(seed=3)
        out = a.rollout(acts, record=("obs", "reward", "terminated", "truncated"))
        for k in range(K):
            b.step(acts[k])
            assert torch.equal(out["reward"][k], b.reward) and torch.equal(out["terminated"][k], b.terminated), (name, k)
            assert torch.equal(out["obs"][k].reshape(-1), b.state.reshape(-1)), (name, k)
        assert torch.equal(a._arena[: a._arena_head], b._arena[: b._arena_head]), name
        assert int(b.t.min()) == K == int(b.t.max())
        a.close(); b.close()


@pytest.mark.parametrize("name", ["noreset_cartpole", "noreset_acrobot", "noreset_frozenlake_4x4", "noreset_mountaincar_continuous"])
def test_kernels_equal_oracle_at_scale_without_autoreset(name):
    from oracle.oracle import OracleVecEnv
    from tests.golden.make_golden import make_actions

    spec = NORESET[name]
    is_fl = spec["env_id"] == "FrozenLake-v1"
    n, T = 4096, 90 if spec["env_id"] != "MountainCarContinuous-v0" else 220
    g = GpuView(make_env_from_spec(_vec, spec, n=n, autoreset=False))
    o = OracleView(make_env_from_spec(OracleVecEnv, spec, n=n, autoreset=False))
    seeds = np.arange(n, dtype=np.uint64) + np.uint64(77)
    acts = make_actions(spec["env_id"], T, n)
    compare_views(g.reset(seeds), o.reset(seeds), is_fl, "reset")
    for k in range(T):
        compare_views(g.step(acts[k]), o.step(acts[k]), is_fl, f"step {k}")
    assert int(o.env.a["t"].min()) == T
    c = g.env.counters()
    assert [c["episodes"], c["updates_applied"], c["env_steps"]] == [int(x) for x in o.env.a["counters"].sum(axis=1)[[0, 1, 3]]]
    g.env.close()


def test_episode_accounting_needs_the_autoreset():
    from ns_gym_amd import make
    from ns_gym_amd._lib import NsgError

    with pytest.raises(NsgError, match="not combinable with NSG_F_NO_AUTORESET"):
        _vec(make("CartPole-v1"), {}, 8, autoreset=False, track_returns=True)


def test_planning_copy_of_a_finished_single_wrapper_starts_unterminated():
    """deepcopy builds a NEW base env and resets it (classic_control.py:168-178): the copy's CartPole pays 1.0 on its own first
    terminated step even if the source has long been down."""
    import copy

    from ns_gym_amd import make
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.update_functions import IncrementUpdate
    from ns_gym_amd.wrappers import NSClassicControlWrapper

    env = NSClassicControlWrapper(make("CartPole-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)},
                                  change_notification=True, delta_change_notification=True)
    env.reset(seed=0)
    rewards = []
    for k in range(40):
        _, r, term, _, _ = env.step(k % 2)
        rewards.append(r)
    assert term and rewards[-1] == 0.0
    sim = copy.deepcopy(env)
    _, r, term, _, _ = sim.step(0)
    assert term and r == 1.0           # the copy's first terminated step
    _, r, term, _, _ = sim.step(0)
    assert term and r == 0.0
    sim.close(); env.close()
