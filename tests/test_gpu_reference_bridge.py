"""The NSBridgeWrapper scenarios of the reference's tests/test_bridge.py (uniform and split mode,
snapshot semantics, planning copies), re-stated against the drop-in adaptor.  The bare-env tests of
that file (attribute setters of the reference's own Bridge object) have no counterpart: here the base
env is a descriptor and its state lives in device rows.

Scenario -> reference test (file:line):
  uniform init / validation / dict obs / reset restores   test_bridge.py:253,258,265,274
  split init / independent drift / reset / one side only  :298,308,319,328
  P is a current snapshot; deepcopy freezes it            :378,390,407
  planning env keeps P frozen and steps without crashing   :432      deepcopy preserves both sides :458
"""
import copy

import pytest

pytestmark = pytest.mark.gpu


def _api():
    import ns_gym_amd as nsg
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.update_functions import DistributionNoUpdate, UniformDrift
    from ns_gym_amd.wrappers import NSBridgeWrapper

    return nsg, NSBridgeWrapper, ContinuousScheduler, DistributionNoUpdate, UniformDrift


def uniform_wrapped():
    nsg, W, Cont, NoUp, _ = _api()
    return W(nsg.make("ns_gym/Bridge-v0"), {"P": NoUp(Cont())}, initial_prob_dist=[0.8, 0.1, 0.1])


def split_wrapped():
    nsg, W, Cont, NoUp, Drift = _api()
    return W(nsg.make("ns_gym/Bridge-v0"), {"P_left": Drift(Cont(), rate=0.05), "P_right": NoUp(Cont())},
             initial_prob_dist=([0.9, 0.05, 0.05], [0.5, 0.25, 0.25]))


def drift_uniform():
    nsg, W, Cont, _, Drift = _api()
    return W(nsg.make("ns_gym/Bridge-v0"), {"P": Drift(Cont(), rate=0.05)}, initial_prob_dist=[0.9, 0.05, 0.05])


SEED = 12   # with LEFT/RIGHT alternation this env's episode survives 16 steps of drifting slips (found with the oracle)


def pace(env, n, start=0):
    """n deterministic steps along the bridge row; a finished episode would be auto-reset by the next step()."""
    for k in range(start, start + n):
        _, _, term, trunc, _ = env.step(0 if k % 2 == 0 else 2)
        assert not (term or trunc), "the fixed seed keeps the episode alive"


def test_bridge_in_tunable_params_registry():
    nsg, *_ = _api()
    assert {"P", "P_left", "P_right"} <= set(nsg.TUNABLE_PARAMS["Bridge"])


def test_uniform_init_and_validation():
    nsg, W, Cont, NoUp, _ = _api()
    env = uniform_wrapped()
    assert env.unwrapped.split_probs is False and env.unwrapped.P == [0.8, 0.1, 0.1]
    with pytest.raises(AssertionError):
        W(nsg.make("ns_gym/Bridge-v0"), {"not_a_real_param": NoUp(Cont())})


def test_uniform_step_returns_dict_obs_and_reset_restores():
    env = drift_uniform()
    obs, info = env.reset(seed=0)
    assert isinstance(obs, dict) and set(obs) == {"state", "env_change", "delta_change", "relative_time"}
    obs, reward, term, trunc, info = env.step(0)
    assert isinstance(obs, dict) and isinstance(obs["state"], int) and info["prob"] == env.unwrapped.P
    assert env.unwrapped.P != [0.9, 0.05, 0.05]     # UniformDrift moved it on the very first step
    for _ in range(5):
        _, _, term, trunc, _ = env.step(0)
        if term or trunc:
            break
    env.reset(seed=0)
    assert env.unwrapped.P == [0.9, 0.05, 0.05]


def test_split_mode_drift_is_per_side():
    env = split_wrapped()
    env.reset(seed=0)
    assert env.unwrapped.split_probs is True
    left, right = list(env.unwrapped.P_left), list(env.unwrapped.P_right)
    for _ in range(20):
        _, _, term, trunc, _ = env.step(0)
        if term or trunc:
            break
    assert env.unwrapped.P_left != left and env.unwrapped.P_right == right
    env.reset(seed=0)
    assert env.unwrapped.P_left == [0.9, 0.05, 0.05] and env.unwrapped.P_right == [0.5, 0.25, 0.25]


def test_split_only_one_side_specified():
    nsg, W, Cont, _, Drift = _api()
    env = W(nsg.make("ns_gym/Bridge-v0"), {"P_left": Drift(Cont(), rate=0.1)}, initial_prob_dist=([1.0, 0.0, 0.0], [0.6, 0.2, 0.2]))
    env.reset(seed=0)
    assert env.unwrapped.split_probs is True
    right = list(env.unwrapped.P_right)
    for _ in range(10):
        _, _, term, trunc, _ = env.step(0)
        if term or trunc:
            break
    assert env.unwrapped.P_right == right == [0.6, 0.2, 0.2]


def test_uniform_P_is_current_snapshot_and_deepcopy_freezes_it():
    env = drift_uniform()
    env.reset(seed=SEED)
    a, b, c = list(env.unwrapped.P), list(env.unwrapped.P), list(env.unwrapped.P)
    assert a == b == c
    pace(env, 1)
    assert list(env.unwrapped.P) != a
    pace(env, 4, start=1)
    snap = copy.deepcopy(env)
    assert snap.is_sim_env
    at_capture = list(snap.unwrapped.P)
    assert at_capture == list(env.unwrapped.P)
    pace(env, 10, start=5)
    assert list(snap.unwrapped.P) == at_capture and list(env.unwrapped.P) != at_capture


def test_split_deepcopy_preserves_and_freezes_both_sides():
    nsg, W, Cont, _, Drift = _api()
    env = W(nsg.make("ns_gym/Bridge-v0"), {"P_left": Drift(Cont(), rate=0.05), "P_right": Drift(Cont(), rate=0.02)},
            initial_prob_dist=([0.9, 0.05, 0.05], [0.7, 0.15, 0.15]))
    env.reset(seed=SEED)
    for k in range(5):
        env.step(k % 4)
    snap = copy.deepcopy(env)
    assert snap.is_sim_env and snap.unwrapped.split_probs is True
    L, R = list(snap.unwrapped.P_left), list(snap.unwrapped.P_right)
    assert L == list(env.unwrapped.P_left) and R == list(env.unwrapped.P_right)
    env.freeze(False)
    for k in range(3):   # few steps: whatever happens to the episode, both sides have drifted away from the snapshot
        env.step(0 if k % 2 == 0 else 2)
    assert list(snap.unwrapped.P_left) == L and list(snap.unwrapped.P_right) == R
    assert list(env.unwrapped.P_left) != L and list(env.unwrapped.P_right) != R


def test_planning_env_freezes_P_and_steps():
    env = drift_uniform()
    env.reset(seed=SEED)
    pace(env, 10)
    pe = env.get_planning_env()
    before = list(pe.unwrapped.P)
    for _ in range(20):
        obs, _, term, trunc, _ = pe.step(pe.action_space.sample())
        if term or trunc:
            pe.reset()
    assert list(pe.unwrapped.P) == before
