"""Pins the oracle's planning-env semantics against what the reference's own
get_planning_env() / __deepcopy__ produce (tests/golden/plan_*.npz): θ carried over or reset to the
construction values, frozen unless in_sim_change, t preserved while the TimeLimit count restarts,
list cursors copied, source env untouched."""
import numpy as np
import pytest

from oracle.oracle import OracleVecEnv
from tests.util import MANIFEST, OracleView, load, make_env_from_spec

PLAN = MANIFEST["planning_specs"]


def run_planning(factory, view_cls, name, fork, strict=False):
    run_planning_rec(factory, view_cls, PLAN[name], load(f"plan_{name}.npz"), fork, strict=strict)


def run_planning_rec(factory, view_cls, spec, rec, fork, strict=False):
    """`strict`: observation, reward (as float32) and theta of the copy's steps in every bit (the oracle; the kernels' libm_exact units)."""
    env = make_env_from_spec(factory, {**spec, "seeds": [spec["seed"]]}, n=1)
    v = view_cls(env)
    is_fl = spec["env_id"] in ("FrozenLake-v1", "CliffWalking-v1")
    v.reset(np.array([spec["seed"]], dtype=np.uint64))
    acts = rec["actions"]
    pre = int(rec["pre"])
    for k in range(pre):
        v.step(acts[k:k + 1])
    theta_mode = 1 if (spec["kind"] == "planning" and not spec["flags"].get("delta_change_notification")) else 0
    sim = fork(env, theta_mode)
    if spec.get("levels", 1) == 2:      # a copy of the copy (MCTS.search: deepcopy(self.env) of the planning env it was given)
        sim = fork(sim, 0)
    sv = view_cls(sim)
    out = sv._out()
    assert out["t"][0] == rec["fork_t"]
    np.testing.assert_allclose(out["theta"][:, 0], rec["fork_theta"], rtol=1e-13)
    if is_fl:
        sim.seed_streams(np.array([rec["fork_env_seed"]], dtype=np.uint64), 0)
    for j in range(len(rec["reward"])):
        out = sv.step(acts[pre + j:pre + j + 1])
        tag = f"fork step {j}"
        if is_fl:
            assert out["state"].reshape(-1)[0] == rec["state"][j, 0], tag
            assert out["prob"][0] == pytest.approx(rec["prob"][j], rel=1e-6), tag
        elif strict:
            np.testing.assert_array_equal(np.asarray(out["state"][0], dtype=np.float32), np.asarray(rec["state"][j], dtype=np.float32), err_msg=tag)
            np.testing.assert_array_equal(out["theta"][:, 0], rec["theta"][j], err_msg=tag)
            assert np.float32(out["reward"][0]) == np.float32(rec["reward"][j]), tag
        else:
            np.testing.assert_allclose(out["state"][0], rec["state"][j], rtol=1e-5, atol=1e-5, err_msg=tag)
        assert out["t"][0] == rec["relative_time"][j], tag
        np.testing.assert_array_equal(out["env_change"][:, 0], rec["gt_env_change"][j], err_msg=tag)
        np.testing.assert_allclose(out["delta_change"][:, 0], rec["gt_delta_change"][j], rtol=1e-5, atol=1e-7, err_msg=tag)
        np.testing.assert_allclose(out["theta"][:, 0], rec["theta"][j], rtol=1e-12, err_msg=tag)
        assert out["reward"][0] == pytest.approx(rec["reward"][j], rel=1e-5, abs=1e-5), tag
        assert out["terminated"][0] == rec["terminated"][j] and out["truncated"][0] == rec["truncated"][j], tag
    src = v._out()
    np.testing.assert_allclose(src["theta"][:, 0], rec["src_theta_after"], rtol=1e-13)
    assert src["t"][0] == rec["src_t_after"]


@pytest.mark.parametrize("name", sorted(PLAN))
def test_oracle_planning_env_matches_reference(name):
    run_planning(OracleVecEnv, OracleView, name, lambda env, mode: env.fork(theta_mode=mode, entropy=99), strict=True)
