"""autoreset=False: nothing resets inside step(); a finished env keeps stepping exactly like the reference's single wrappers,
which forward every step() to gymnasium whatever `done` said (ns_gym/base.py:313).  The fixtures were produced by the REFERENCE
wrappers stepped far past their first `done` without a reset (make_golden.py NORESET_SPECS); here the oracle."""
import numpy as np
import pytest

from oracle.oracle import OracleVecEnv
from tests.test_oracle_grid import check_grid
from tests.util import MANIFEST, OracleView, check_trajectory, load, make_env_from_spec

NORESET = MANIFEST["noreset_specs"]
NORESET_GRID = MANIFEST["noreset_grid_specs"]


@pytest.mark.parametrize("name", sorted(NORESET))
def test_oracle_follows_the_reference_past_done(name):
    spec, rec = NORESET[name], load(f"traj_{name}.npz")
    assert int(rec["was_reset"].sum()) == 0 and int((rec["terminated"] | rec["truncated"]).sum()) > 30
    env = make_env_from_spec(OracleVecEnv, spec, autoreset=False)
    check_trajectory(OracleView(env), spec, rec, strict=True)


@pytest.mark.parametrize("name", sorted(NORESET_GRID))
def test_oracle_grid_follows_the_reference_past_done(name):
    spec, rec = NORESET_GRID[name], load(f"grid_{name}.npz")
    env = make_env_from_spec(OracleVecEnv, spec, autoreset=False)
    check_grid(OracleView(env), spec, rec)


def test_cartpole_pays_one_on_the_terminating_step_and_zero_afterwards():
    rec = load("traj_noreset_cartpole.npz")
    term, rew = rec["terminated"].astype(bool), rec["reward"]
    first = term.argmax(axis=0)
    for i, k in enumerate(first):
        assert term[k:, i].all() and (rew[: k + 1, i] == 1.0).all() and (rew[k + 1:, i] == 0.0).all()
    # t and θ run on past `done`
    assert (rec["relative_time"][-1] == rec["relative_time"].shape[0] - 1).all()
    assert (np.diff(rec["theta"][:, 0, 0]) > 0).all()


def test_flag_is_part_of_the_compiled_config_and_excludes_episode_accounting():
    from ns_gym_amd import _abi as A
    from ns_gym_amd import make
    from ns_gym_amd.spec import compile_config

    cfg, _, _, _ = compile_config(make("CartPole-v1"), {}, autoreset=False)
    assert cfg.flags & A.F_NO_AUTORESET
    cfg, _, _, _ = compile_config(make("CartPole-v1"), {})
    assert not cfg.flags & A.F_NO_AUTORESET
