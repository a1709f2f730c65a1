"""Pins the oracle's CliffWalking / Bridge paths against trajectories from the
reference's NSCliffWalkingWrapper / NSBridgeWrapper (tests/golden/grid_*.npz).  CliffWalking is
bit-exact incl. the env-stream categorical draws; Bridge draws its slip from the unseeded global
np.random, so its exact fixtures use one-hot slip distributions (uniform and split mode)."""
import numpy as np
import pytest

from oracle.oracle import OracleVecEnv
from tests.util import MANIFEST, OracleView, load, make_env_from_spec

GRID = MANIFEST["grid_specs"]


def grid_spec(name):
    spec = dict(GRID[name])
    wk = dict(spec.get("wrapper_kwargs", {}))
    ipd = wk.get("initial_prob_dist")
    if isinstance(ipd, dict) and "__pair__" in ipd:
        wk["initial_prob_dist"] = (list(ipd["__pair__"][0]), list(ipd["__pair__"][1]))
    spec["wrapper_kwargs"] = wk
    return spec


def check_grid(view, spec, rec):
    seeds = np.asarray(spec["seeds"], dtype=np.uint64)
    out = view.reset(seeds)
    P = len(spec["params"])

    def cmp(out, k, kk):
        tag = f"index {k}"
        np.testing.assert_array_equal(out["state"].reshape(-1), rec["state"][k, :, 0], err_msg=tag)
        np.testing.assert_array_equal(out["t"], rec["relative_time"][k], err_msg=tag)
        np.testing.assert_array_equal(out["env_change"][:P].T, rec["gt_env_change"][k], err_msg=tag)
        np.testing.assert_allclose(out["delta_change"][:P].T, rec["gt_delta_change"][k], rtol=1e-6, atol=1e-7, err_msg=tag)
        np.testing.assert_array_equal(out["theta"].T, rec["theta"][k], err_msg=tag)
        if kk is not None:
            np.testing.assert_allclose(out["reward"], rec["reward"][kk], rtol=1e-6, err_msg=tag)
            np.testing.assert_array_equal(out["terminated"], rec["terminated"][kk], err_msg=tag)
            np.testing.assert_array_equal(out["truncated"], rec["truncated"][kk], err_msg=tag)
            if "prob" in rec.files:
                np.testing.assert_allclose(out["prob"], rec["prob"][kk], rtol=1e-6, err_msg=tag)

    cmp(out, 0, None)
    for k in range(spec["T"]):
        cmp(view.step(rec["actions"][k]), k + 1, k)


@pytest.mark.parametrize("name", sorted(GRID))
def test_grid_trajectory_matches_reference_wrapper(name):
    spec = grid_spec(name)
    env = make_env_from_spec(OracleVecEnv, spec)
    check_grid(OracleView(env), spec, load(f"grid_{name}.npz"))


def test_bridge_slip_distribution_statistics():
    """The non-degenerate slip draw (np.random.choice(p=P)) can only be checked in distribution."""
    from ns_gym_amd import make
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.update_functions import DistributionNoUpdate

    n = 200_000
    env = OracleVecEnv(make("ns_gym/Bridge-v0"), {"P": DistributionNoUpdate(ContinuousScheduler())}, n,
                       initial_prob_dist=[0.6, 0.3, 0.1])
    env.reset(seed=0)
    env.step(np.full(n, 2, dtype=np.int32))   # RIGHT from (2,4): a -> (2,5); a+1=UP -> (1,4); a-1=DOWN -> (3,4)
    cells = env.a["cell"]
    freq = [np.mean(cells == c) for c in (2 * 8 + 5, 1 * 8 + 4, 3 * 8 + 4)]
    np.testing.assert_allclose(freq, [0.6, 0.3, 0.1], atol=5e-3)
