"""nsg_rollout (K fused steps per launch, persistent rows held in registers) must be
indistinguishable from K nsg_step launches: same per-step outputs, same final state, same
counters — and therefore equal to the oracle."""
import numpy as np
import pytest

from tests.util import TRAJ_SPECS, make_env_from_spec

pytestmark = pytest.mark.gpu


def _vec(*a, **k):
    from ns_gym_amd.vec_env import VecNSEnv

    return VecNSEnv(*a, **k)


@pytest.mark.parametrize("name,n,K", [
    ("c1_cartpole_masspole_inc", 5000, 97), ("cartpole_two_params", 4096, 64), ("cartpole_constraint", 3000, 50),
    ("c2_cartpole_gravity_rw", 8192, 40), ("c4_pendulum_m_inc", 4096, 230), ("acrobot_constraints", 2048, 40),
    ("mountaincar", 2048, 210), ("c3_frozenlake_step50", 8192, 120), ("cartpole_persistent", 2048, 80),
    ("frozenlake_decrement", 4096, 150), ("frozenlake_randomcat", 2048, 60), ("frozenlake_lcbounded_persistent", 2048, 60),
    ("frozenlake_4x4_drift_rewards", 3000, 130), ("cartpole_shared_scheduler_and_list", 2048, 60),
])
def test_rollout_equals_single_steps(name, n, K):
    import torch

    from tests.golden.make_golden import make_actions

    spec = TRAJ_SPECS[name]
    a, b = make_env_from_spec(_vec, spec, n=n, track_returns=True), make_env_from_spec(_vec, spec, n=n, track_returns=True)
    a.reset(seed=99)
    b.reset(seed=99)
    acts = torch.from_numpy(make_actions(spec["env_id"], K, n)).cuda()
    rec = ("obs", "reward", "terminated", "truncated", "env_change", "delta_change")
    # two launches of K/2 to cover register state being written back and re-loaded
    k1 = K // 2
    out1 = b.rollout(acts[:k1], record=rec)
    out2 = b.rollout(acts[k1:], record=rec)
    traj = {k: torch.cat([out1[k], out2[k]]) for k in rec}
    P = max(a.cfg.n_params, 1)
    for k in range(K):
        obs, r, te, tr, info = a.step(acts[k])
        assert torch.equal(traj["obs"][k], a.state), f"obs step {k}"
        assert torch.equal(traj["reward"][k], r) and torch.equal(traj["terminated"][k], te) and torch.equal(traj["truncated"][k], tr)
        assert torch.equal(traj["env_change"][k], a.gt_env_change[:P]) and torch.equal(traj["delta_change"][k], a.gt_delta_change[:P])
    for field in ("theta", "t", "state"):
        assert torch.equal(getattr(a, field), getattr(b, field)), field
    if not a.is_grid:
        assert torch.equal(a.phys, b.phys)
    for row in ("status", "episode", "rng_env"):
        if a.buf[row] is not None:
            assert torch.equal(a.buf[row], b.buf[row]), row
    for row in ("ep_return", "last_return", "last_length"):   # return rows exist only where the return is not implied by the length
        if a.buf[row] is not None:
            assert torch.equal(a.buf[row], b.buf[row]), row
    for x, y in zip(a.episode_returns(), b.episode_returns()):
        assert torch.equal(x, y)
    assert a.counters() == b.counters()
    a.close(); b.close()


@pytest.mark.parametrize("name,n,K", [("cliff_decrement", 4096, 90), ("cliff_terminal_stepwise_rewards", 3000, 70),
                                      ("bridge_split_onehot", 2048, 60), ("bridge_uniform_onehot", 2048, 110)])
def test_grid_rollout_equals_single_steps(name, n, K):
    """CliffWalking / Bridge: cell, t, status, episode return, the env stream and the table probabilities ride in
    registers across the K fused steps; the result must equal K single launches row for row."""
    import torch

    from tests.test_oracle_grid import grid_spec

    spec = grid_spec(name)
    a, b = make_env_from_spec(_vec, spec, n=n, track_returns=True), make_env_from_spec(_vec, spec, n=n, track_returns=True)
    a.reset(seed=5)
    b.reset(seed=5)
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.randint(0, 4, (K, n), dtype=torch.int32, device="cuda", generator=g)
    rec = ("obs", "reward", "terminated", "truncated", "env_change", "delta_change")
    k1 = K // 3
    outs = [b.rollout(acts[:k1], record=rec), b.rollout(acts[k1:], record=rec)]
    traj = {k: torch.cat([o[k] for o in outs]) for k in rec}
    P = max(a.cfg.n_params, 1)
    for k in range(K):
        obs, r, te, tr, info = a.step(acts[k])
        assert torch.equal(traj["obs"][k], a.state) and torch.equal(traj["reward"][k], r), f"step {k}"
        assert torch.equal(traj["terminated"][k], te) and torch.equal(traj["truncated"][k], tr)
        assert torch.equal(traj["env_change"][k], a.gt_env_change[:P]) and torch.equal(traj["delta_change"][k], a.gt_delta_change[:P])
    for row in ("cell", "t", "status", "episode", "theta", "table_prob", "rng_env", "prob", "ep_return", "last_return", "last_length", "cursor"):
        if a.buf[row] is not None:
            assert torch.equal(a.buf[row], b.buf[row]), row
    assert a.counters() == b.counters()
    a.close(); b.close()


@pytest.mark.parametrize("specialize", [False, True], ids=["generic-group-kernel", "specialised-group-unit"])
def test_group_rollout_equals_single_group_steps(specialize):
    """nsg_rollout_group - K fused steps of a Pendulum, an Acrobot, a FrozenLake and a full-engine CartPole member in ONE launch -
    leaves every member exactly where K nsg_step_group launches (and K nsg_step launches per member) leave it, and records the same
    trajectory: bit-identical rows, streams and counters."""
    import torch

    from ns_gym_amd.vec_env import VecNSEnv, rollout_group, step_group, step_group_kind
    from tests.golden.make_golden import make_actions
    from tests.util import TRAJ_SPECS, make_env_from_spec

    names = ["c4_pendulum_m_inc", "c4_acrobot_mass2_inc", "c3_frozenlake_step50", "c2_cartpole_gravity_rw"]
    ns = [5000, 3000, 4096, 2500]
    K, reps = 48, 3
    mk = lambda: [make_env_from_spec(lambda *a, **k: VecNSEnv(*a, **k), TRAJ_SPECS[nm], n=n, track_returns=True, specialize=specialize)  # noqa: E731
                  for nm, n in zip(names, ns)]
    fused, stepped = mk(), mk()
    for e in fused + stepped:
        e.reset(seed=77)
    acts = [torch.from_numpy(make_actions(TRAJ_SPECS[nm]["env_id"], K * reps, n)).cuda() for nm, n in zip(names, ns)]
    rec = ("obs", "reward", "terminated", "truncated", "env_change", "delta_change")
    for r in range(reps):
        outs = rollout_group(fused, [a[r * K:(r + 1) * K] for a in acts], record=rec)
        for k in range(K):
            step_group(stepped, [a[r * K + k] for a in acts])
            for e, o in zip(stepped, outs):
                assert torch.equal(o["reward"][k], e.reward) and torch.equal(o["terminated"][k], e.terminated) and torch.equal(o["truncated"][k], e.truncated)
                assert torch.equal(o["obs"][k], e.state)
                P = e.cfg.n_params
                assert torch.equal(o["env_change"][k][:P], e.gt_env_change) and torch.equal(o["delta_change"][k][:P], e.gt_delta_change)
    assert step_group_kind(fused).startswith("specialised") if specialize else step_group_kind(fused) == "generic-full"
    for a, b, nm in zip(fused, stepped, names):
        for row in ("phys", "cell", "theta", "table_prob", "t", "status", "episode", "rng_env", "rng_upd", "cursor", "obs", "reward", "terminated",
                    "truncated", "env_change", "delta_change", "prob", "ep_return", "last_return", "last_length"):
            if a.buf[row] is not None:
                assert torch.equal(a.buf[row], b.buf[row]), (nm, row)
        assert a.counters() == b.counters(), nm
    for e in fused + stepped:
        e.close()
