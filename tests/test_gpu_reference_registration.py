"""The reference's registration scenarios (tests/test_registration.py): users register a factory that builds
base env + NS wrapper under a custom id, `make` it, and then deep-copy it / ask for a planning env (the path
MCTS-style planners take).  Same assertions, with ns_gym_amd's `register` / `registry` / `make` in gymnasium's role."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CLASSIC = ["CartPole-v1", "Acrobot-v1", "MountainCar-v0", "MountainCarContinuous-v0", "Pendulum-v1"]
GRID = ["CliffWalking-v1", "FrozenLake-v1"]
OBS_KEYS = ("state", "env_change", "delta_change", "relative_time")


def _classic_params(env_id):
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.update_functions import IncrementUpdate

    up, down = IncrementUpdate(ContinuousScheduler(), k=0.1), IncrementUpdate(ContinuousScheduler(), k=-0.1)
    return {"CartPole-v1": {"masspole": up, "gravity": up}, "Acrobot-v1": {"LINK_LENGTH_1": up, "LINK_MASS_2": up},
            "MountainCar-v0": {"gravity": down, "force": up}, "MountainCarContinuous-v0": {"power": up},
            "Pendulum-v1": {"m": up, "g": up}}[env_id]


def _grid_params():
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.update_functions import DistributionIncrementUpdate

    return {"P": DistributionIncrementUpdate(ContinuousScheduler(), k=-0.1)}


def _wrapper_class(env_id):
    from ns_gym_amd import wrappers as w

    return {"CliffWalking-v1": w.NSCliffWalkingWrapper, "FrozenLake-v1": w.NSFrozenLakeWrapper}.get(env_id, w.NSClassicControlWrapper)


class _Registered:
    """Registers `custom_id` -> factory(base id, params, wrapper kwargs) for the duration of a `with` block."""

    def __init__(self, env_id, params, custom_id, **wrapper_kwargs):
        self.env_id, self.params, self.custom_id, self.kw = env_id, params, custom_id, wrapper_kwargs

    def __enter__(self):
        import ns_gym_amd as nsg

        cls, env_id, params, kw = _wrapper_class(self.env_id), self.env_id, self.params, self.kw

        def factory(**make_kwargs):
            return cls(nsg.make(env_id, **make_kwargs), params, **kw)

        nsg.registry.pop(self.custom_id, None)
        nsg.register(id=self.custom_id, entry_point=factory, disable_env_checker=True, order_enforce=False)
        return nsg.make(self.custom_id)

    def __exit__(self, *exc):
        import ns_gym_amd as nsg

        nsg.registry.pop(self.custom_id, None)


def _single_wrapper_layer(env, cls):
    assert isinstance(env, cls), type(env).__name__
    inner = env.env
    while hasattr(inner, "env"):
        assert not isinstance(inner, cls), "wrapped twice"
        inner = inner.env
    assert not isinstance(inner, cls)


@pytest.mark.parametrize("env_id", CLASSIC + GRID)
def test_made_env_is_one_wrapper_around_the_base_env(env_id):
    params = _grid_params() if env_id in GRID else _classic_params(env_id)
    with _Registered(env_id, params, f"TestReg-{env_id}") as env:
        _single_wrapper_layer(env, _wrapper_class(env_id))
        env.close()


@pytest.mark.parametrize("env_id", CLASSIC + GRID)
def test_deepcopy_of_a_made_env_is_a_sim_env_of_the_same_class(env_id):
    params = _grid_params() if env_id in GRID else _classic_params(env_id)
    with _Registered(env_id, params, f"TestRegDC-{env_id}") as env:
        env.reset(seed=42)
        for _ in range(3 if env_id in CLASSIC else 1):
            env.step(env.action_space.sample())
        sim = copy.deepcopy(env)
        assert sim.is_sim_env is True and not env.is_sim_env
        _single_wrapper_layer(sim, _wrapper_class(env_id))
        sim.close(); env.close()


@pytest.mark.parametrize("env_id", CLASSIC + GRID)
def test_planning_env_of_a_made_env(env_id):
    params = _grid_params() if env_id in GRID else _classic_params(env_id)
    with _Registered(env_id, params, f"TestRegPE-{env_id}") as env:
        env.reset(seed=42)
        env.step(env.action_space.sample())
        plan = env.get_planning_env()
        assert plan.is_sim_env is True
        _single_wrapper_layer(plan, _wrapper_class(env_id))
        plan.close(); env.close()


@pytest.mark.parametrize("env_id", CLASSIC)
def test_wrapper_kwargs_survive_the_copy_and_the_copy_steps(env_id):
    """Registered with both notification flags: the copy must be built without them leaking anywhere they do not belong
    and must step with the full observation dict."""
    with _Registered(env_id, _classic_params(env_id), f"TestRegKW-{env_id}", change_notification=True,
                     delta_change_notification=True) as env:
        env.reset(seed=42)
        env.step(env.action_space.sample())
        sim = copy.deepcopy(env)
        assert sim.is_sim_env is True and isinstance(sim, _wrapper_class(env_id))
        assert sim.change_notification and sim.delta_change_notification
        obs, _, _, _, _ = sim.step(sim.action_space.sample())
        assert isinstance(obs, dict) and all(k in obs for k in OBS_KEYS)
        sim.close(); env.close()


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Pendulum-v1"])
def test_non_scalar_reward_propagates_through_the_copy(env_id):
    from ns_gym_amd.base import Reward

    with _Registered(env_id, _classic_params(env_id), f"TestRegSR-{env_id}", scalar_reward=False) as env:
        env.reset(seed=42)
        _, reward, _, _, _ = env.step(env.action_space.sample())
        assert isinstance(reward, Reward)
        sim = copy.deepcopy(env)
        assert sim.scalar_reward is False
        _, sim_reward, _, _, _ = sim.step(sim.action_space.sample())
        assert isinstance(sim_reward, Reward)
        sim.close(); env.close()


@pytest.mark.parametrize("env_id", CLASSIC)
def test_short_episode_then_reset_starts_clean(env_id):
    with _Registered(env_id, _classic_params(env_id), f"TestRegEp-{env_id}") as env:
        env.reset(seed=42)
        for _ in range(20):
            _, _, done, trunc, _ = env.step(env.action_space.sample())
            if done or trunc:
                break
        obs, info = env.reset(seed=42)
        assert env.t == 0 and obs["relative_time"] == 0
        assert isinstance(obs, dict) and all(k in obs for k in OBS_KEYS)
        env.close()


def test_made_env_updates_its_parameters_when_stepped():
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.update_functions import IncrementUpdate

    k = 0.5
    with _Registered("CartPole-v1", {"masspole": IncrementUpdate(ContinuousScheduler(start=0), k=k)}, "TestRegStep-CartPole-v1") as env:
        env.reset(seed=42)
        before = env.unwrapped.masspole
        env.step(0)
        assert np.isclose(env.unwrapped.masspole, before + k)
        env.close()
