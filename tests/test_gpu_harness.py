"""The device-side episode harness (ns_gym_amd.evaluate.run_episodes; reference: ns_gym/evaluate/run_experiment.py:108-141,
206-217) against the ORACLE's stepper: for every env the row's total_reward, num_steps and seed must be what the reference's
`run_episode` loop produces - reset(seed), step until done or truncated, sum the rewards, count the steps - computed here by
stepping the CPU oracle with the same action table and cutting each env's trajectory at its first episode end.

Covered: the fused open-loop path (nsg_rollout chunks, alive masks and sums on the device, end test one chunk late) for
CartPole (C1's config), Pendulum (never terminates: every episode runs into the TimeLimit and through the max_steps + 1
break rule), FrozenLake (C3's config); the closed-loop path (a policy that looks at the state); SARNS records; chunk sizes
that do and do not divide the horizon."""
import numpy as np
import pytest

from tests.util import TRAJ_SPECS, make_env_from_spec

pytestmark = pytest.mark.gpu


def _vec(*a, **k):
    from ns_gym_amd.vec_env import VecNSEnv

    return VecNSEnv(*a, **k)


def _oracle_rows(spec, n, seed, acts, limit):
    """The reference loop per env, on the oracle: rows (total_reward, num_steps) and the per-step records."""
    from oracle.oracle import OracleVecEnv

    orc = make_env_from_spec(OracleVecEnv, spec, n=n)
    orc.reset(seed=np.arange(n, dtype=np.uint64) + np.uint64(seed))
    total, steps, alive = np.zeros(n, dtype=np.float64), np.zeros(n, dtype=np.int64), np.ones(n, dtype=bool)
    rewards = []
    for k in range(min(limit + 1, len(acts))):          # `if num_steps == max_steps + 1: break` (run_experiment.py:127-129)
        orc.step(acts[k])
        r = orc.a["reward"].astype(np.float64)
        total += np.where(alive, r, 0.0)
        steps += alive
        rewards.append(np.where(alive, r, np.nan))
        alive &= ~((orc.a["terminated"] | orc.a["truncated"]).astype(bool))
        if not alive.any():
            break
    return total, steps, np.array(rewards)


@pytest.mark.parametrize("name,chunk", [("c1_cartpole_masspole_inc", 64), ("c1_cartpole_masspole_inc", 7), ("c4_pendulum_m_inc", 64),
                                        ("c4_pendulum_m_inc", 50), ("c3_frozenlake_step50", 32), ("mountaincar", 64)])
def test_open_loop_rows_equal_the_reference_loop_on_the_oracle(name, chunk):
    import torch

    from ns_gym_amd.evaluate import run_episodes

    spec = TRAJ_SPECS[name]
    n, seed = 3000, 77
    env = make_env_from_spec(_vec, spec, n=n)
    limit = env.cfg.max_episode_steps
    g = torch.Generator(device="cuda").manual_seed(5)
    if env.action_is_float:
        acts = torch.rand((limit + 1, n), device="cuda", generator=g) * 4 - 2
    else:
        acts = torch.randint(0, env.n_actions, (limit + 1, n), dtype=torch.int32, device="cuda", generator=g)
    rows = run_episodes(env, seed=seed, actions=acts, chunk=chunk)
    total, steps, _ = _oracle_rows(spec, n, seed, acts.cpu().numpy(), limit)
    assert len(rows) == n and all(len(r) == 6 for r in rows)
    np.testing.assert_array_equal([r[2] for r in rows], steps)                       # num_steps
    np.testing.assert_array_equal([r[3] for r in rows], np.arange(n) + seed)         # seed column
    np.testing.assert_array_equal([r[4] for r in rows], np.arange(n))                # sample_id
    if spec["env_id"] in ("CartPole-v1", "FrozenLake-v1", "MountainCar-v0"):
        np.testing.assert_array_equal([r[0] for r in rows], total)                   # sums of +-1 / 0 / 1: exact
    else:
        np.testing.assert_allclose([r[0] for r in rows], total, rtol=1e-6, atol=1e-4)  # float32 per-step rewards within 1e-5 each
    assert max(steps) <= limit + 1 and min(steps) >= 1
    if name.startswith("c4_pendulum"):
        assert (steps == limit).all()      # never terminates: truncated by the TimeLimit on step `limit`
    env.close()


def test_default_random_policy_is_fused_and_ends_early():
    """No `actions`, no `policy`: uniform random actions drawn per chunk on the device; CartPole episodes are over after a few
    dozen steps, so the harness must stop long before the 501-step horizon (lagged end test), with consistent rows."""
    from ns_gym_amd.evaluate import run_episodes

    env = make_env_from_spec(_vec, TRAJ_SPECS["c1_cartpole_masspole_inc"], n=4096)
    rows = run_episodes(env, seed=1, chunk=32)
    steps = np.array([r[2] for r in rows])
    assert (np.array([r[0] for r in rows]) == steps).all()          # CartPole pays 1.0 per step
    assert steps.max() < 200 and steps.min() >= 5
    launched = env.counters()["env_steps"] + env.counters()["episodes"]
    assert launched <= 4096 * (steps.max() + 3 * 32)                  # at most ~2 chunks past the last episode's end
    # the same columns as arrays (no Python row per env): an open-loop table makes the two runs identical
    import torch
    table = torch.randint(0, 2, (300, 4096), dtype=torch.int32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    rows = run_episodes(env, seed=7, actions=table, sample_id=range(100, 100 + 4096))
    cols = run_episodes(env, seed=7, actions=table, sample_id=range(100, 100 + 4096), as_arrays=True)
    assert cols["total_reward"].tolist() == [r[0] for r in rows] and cols["num_steps"].tolist() == [r[2] for r in rows]
    assert cols["seed"].tolist() == [r[3] for r in rows] and cols["sample_id"].tolist() == [r[4] for r in rows]
    env.close()


def test_closed_loop_policy_and_sarns_records():
    import torch

    from ns_gym_amd.evaluate import run_episodes

    spec = TRAJ_SPECS["c1_cartpole_masspole_inc"]
    n, seed = 512, 9

    def policy(state):            # looks at the observation: push towards the side the pole leans to
        return (state[:, 2] > 0).to(torch.int32)

    env = make_env_from_spec(_vec, spec, n=n)
    rows = run_episodes(env, policy, seed=seed, record_sarns=True, chunk=16)
    # the same policy on the oracle, env by env
    from oracle.oracle import OracleVecEnv

    orc = make_env_from_spec(OracleVecEnv, spec, n=n)
    orc.reset(seed=np.arange(n, dtype=np.uint64) + np.uint64(seed))
    total, steps, alive = np.zeros(n), np.zeros(n, dtype=np.int64), np.ones(n, dtype=bool)
    first_states = orc.a["obs"].copy()
    for k in range(env.cfg.max_episode_steps + 1):
        a = (orc.a["obs"][:, 2] > 0).astype(np.int32)
        orc.step(a)
        total += np.where(alive, orc.a["reward"], 0.0)
        steps += alive
        alive &= ~((orc.a["terminated"] | orc.a["truncated"]).astype(bool))
        if not alive.any():
            break
    np.testing.assert_array_equal([r[2] for r in rows], steps)
    np.testing.assert_array_equal([r[0] for r in rows], total)
    for i in (0, 1, n - 1):
        sarns = rows[i][1]
        assert len(sarns) == steps[i] and len(sarns[0]) == 4
        np.testing.assert_allclose(sarns[0][0], first_states[i], rtol=1e-6, atol=1e-6)       # S of the first record = reset state
        for (s, a, r, s2), (s_next, *_rest) in zip(sarns[:-1], sarns[1:]):
            assert s2 == s_next and r == 1.0                                                 # chained records
    env.close()


def test_sarns_from_the_fused_path_and_csv(tmp_path):
    import csv

    import torch

    from ns_gym_amd.evaluate import CSV_HEADER, run_episodes, write_results_csv

    spec = TRAJ_SPECS["c3_frozenlake_step50"]
    n = 200
    env = make_env_from_spec(_vec, spec, n=n)
    acts = torch.randint(0, 4, (101, n), dtype=torch.int32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    rows = run_episodes(env, seed=4, actions=acts, record_sarns=True, chunk=24)
    total, steps, _ = _oracle_rows(spec, n, 4, acts.cpu().numpy(), 100)
    np.testing.assert_array_equal([r[2] for r in rows], steps)
    np.testing.assert_array_equal([r[0] for r in rows], total)
    a_host = acts.cpu().numpy()
    for i in (0, 17, n - 1):
        sarns = rows[i][1]
        assert len(sarns) == steps[i] and sarns[0][0] == 0                     # FrozenLake starts in cell 0
        assert [rec[1] for rec in sarns] == a_host[: steps[i], i].tolist()     # the actions this env took
        assert all(s2 == nxt[0] for (_, _, _, s2), nxt in zip(sarns[:-1], sarns[1:]))
    p = tmp_path / "res.csv"
    write_results_csv(str(p), rows)
    with open(p) as f:
        rd = list(csv.reader(f))
    assert rd[0] == CSV_HEADER and len(rd) == n + 1 and float(rd[1][0]) == total[0] and int(rd[1][2]) == steps[0]
    env.close()
