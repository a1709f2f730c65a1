"""Pins the oracle's wrapper semantics (update-before-step ordering,
constraint rejection, dependency resolver, notification ground truth, TimeLimit, next-step
autoreset == reset(seed=None), seeding by SeedSequence(seed).spawn, persistent_params)
against trajectories produced by the reference's NSClassicControlWrapper /
NSFrozenLakeWrapper running over the restated base envs."""
import numpy as np
import pytest

from oracle.oracle import OracleVecEnv
from tests.util import TRAJ_SPECS, OracleView, check_trajectory, load, make_env_from_spec


@pytest.mark.parametrize("name", sorted(TRAJ_SPECS))
def test_trajectory_matches_reference_wrapper(name):
    spec = TRAJ_SPECS[name]
    rec = load(f"traj_{name}.npz")
    env = make_env_from_spec(OracleVecEnv, spec)
    # no tolerance for the classic-control envs: observation, reward and theta in every bit (what that took: oracle/Makefile's
    # -fno-builtin-sin/-cos/-pow - DESIGN.md section 2).  The grid envs' integer path was always compared exactly.
    check_trajectory(OracleView(env), spec, rec, strict=True)


def test_reset_semantics():
    """reset(seed) re-seeds env + update-fn streams, reset() continues them, θ restored
    (ns_gym/base.py:365-431; tests/test_step_reset.py:559-1122 of the reference)."""
    g = load("reset_semantics.npz")
    spec = TRAJ_SPECS["cartpole_two_params"]
    env = make_env_from_spec(OracleVecEnv, spec, n=1, change_notification=None) if False else \
        make_env_from_spec(OracleVecEnv, {**spec, "flags": {"change_notification": True, "delta_change_notification": True}}, n=1)
    acts = g["actions"]
    states, thetas, deltas = [], [], []

    def snap():
        states.append(env.a["obs"][0].copy()); thetas.append(env.a["theta"][:, 0].copy())
        deltas.append(env.a["delta_change"][:2, 0].astype(np.float64))

    env.reset(seed=np.array([42], dtype=np.uint64)); snap()
    for k in range(7):
        env.step(acts[k:k + 1]); snap()
    env.reset(); snap()
    for k in range(7, 14):
        env.step(acts[k:k + 1]); snap()
    env.reset(seed=np.array([42], dtype=np.uint64)); snap()
    for k in range(7):
        env.step(acts[k:k + 1]); snap()
    np.testing.assert_allclose(np.array(states), g["state"], atol=1e-6)
    np.testing.assert_allclose(np.array(thetas), g["theta"], rtol=1e-13)
    np.testing.assert_allclose(np.array(deltas), g["delta"], rtol=1e-6, atol=1e-7)
