"""nsg_rollout_policy: K closed-loop steps in one launch - the policy is evaluated inside the kernel, the episode accounts stay in
registers.  Whatever the action source, the launch must be indistinguishable from the loop it replaces (MCTS._default_policy,
MCTS.py:162-181; run_episode, run_experiment.py:108-129; the tutorial's tabular run_episode, cell 12):

  * UniformRandom / action tables: bit-identical to `VecNSEnv.rollout` over the same action table - every recorded row, every
    persistent row, the counters - for every env type, generic kernels and the specialised unit, in one launch or chunked;
  * TabularPolicy / LinearPolicy (closed loops): bit-identical to `step()` loops driven by the same policy evaluated on the host;
  * the accounts: Python's `tot_reward += reward * gamma ** depth` over the recorded rewards, bit for bit (float64).
"""
import numpy as np
import pytest

from tests.util import TRAJ_SPECS, make_env_from_spec

pytestmark = pytest.mark.gpu

REC = ("obs", "reward", "terminated", "truncated", "env_change", "delta_change")


def _vec(*a, **k):
    from ns_gym_amd.vec_env import VecNSEnv

    return VecNSEnv(*a, **k)


def _same_batch(a, b):
    import torch

    for field in ("theta", "t", "state"):
        assert torch.equal(getattr(a, field), getattr(b, field)), field
    if not a.is_grid:
        assert torch.equal(a.phys, b.phys)
    for row in ("status", "episode", "rng_env", "rng_upd", "cursor", "ep_return", "last_return", "last_length", "reward", "terminated", "truncated"):
        if a.buf.get(row) is not None:
            assert torch.equal(a.buf[row], b.buf[row]), row
    assert a.counters() == b.counters()


def _spec_of(name):
    if name in TRAJ_SPECS:
        return TRAJ_SPECS[name]
    from tests.test_oracle_grid import grid_spec

    return grid_spec(name)


@pytest.mark.parametrize("specialize", [False, True])
@pytest.mark.parametrize("name,n,K", [
    ("c1_cartpole_masspole_inc", 5000, 97), ("c2_cartpole_gravity_rw", 8192, 40), ("cartpole_two_params", 4096, 64),
    ("c4_pendulum_m_inc", 4096, 230), ("acrobot_constraints", 2048, 40), ("mountaincar", 2048, 210),
    ("c3_frozenlake_step50", 8192, 120), ("frozenlake_randomcat", 2048, 60), ("cliff_decrement", 4096, 90), ("bridge_split_onehot", 2048, 60),
])
def test_uniform_policy_equals_table_rollout(name, n, K, specialize):
    """The in-kernel uniform draws are the table `UniformRandom.table()` produces on the host; a fused policy rollout over them - in
    one launch, and again chunked with the step counter carried over - equals nsg_rollout over that table in every row."""
    import torch

    from ns_gym_amd.policies import EpisodeAccounts, UniformRandom

    spec = _spec_of(name)
    a, b, c = (make_env_from_spec(_vec, spec, n=n, track_returns=True, specialize=specialize) for _ in range(3))
    for e in (a, b, c):
        e.reset(seed=99)
    pol = UniformRandom(seed=1234, index0=7)
    table = pol.table(a, 0, K)
    assert table.shape == (K, n)
    if a.action_is_float:
        lo, hi = a.spec.env_type.action_low, a.spec.env_type.action_high
        assert table.min() >= lo and table.max() <= hi and table.std() > 0.2 * (hi - lo)
    else:
        assert set(np.unique(table)) == set(range(a.n_actions))
    ref = a.rollout(torch.from_numpy(table).cuda(), record=REC)
    acc = EpisodeAccounts(b, gamma=0.97, horizon=K + 1)
    out = b.rollout_policy(pol, K, record=REC, accounts=acc, record_actions=True)
    assert torch.equal(out["actions"].cpu(), torch.from_numpy(table))
    for k in REC:
        assert torch.equal(out[k], ref[k]), k
    _same_batch(a, b)
    assert b.policy_kernels == ("config-specialised" if (specialize and b.specialized) else "generic")
    # chunked: three launches, the counter carried over, the accounts carried over
    acc_c = EpisodeAccounts(c, gamma=0.97, horizon=K + 1)
    k1, k2 = K // 3, K // 2
    parts = [c.rollout_policy(pol, k1, record=REC, accounts=acc_c, step0=0), c.rollout_policy(pol, k2 - k1, record=REC, accounts=acc_c, step0=k1),
             c.rollout_policy(pol, K - k2, record=REC, accounts=acc_c, step0=k2)]
    for k in REC:
        assert torch.equal(torch.cat([p[k] for p in parts]), ref[k]), k
    _same_batch(a, c)
    for f in ("ret", "length", "alive"):
        assert torch.equal(getattr(acc, f), getattr(acc_c, f)), f
    # the accounts against Python's own loop over the recorded rewards (float32-exact rewards only: CartPole, Acrobot, MountainCar, grid defaults)
    rew = ref["reward"].cpu().numpy().astype(np.float64)
    done = (ref["terminated"] | ref["truncated"]).cpu().numpy()
    exact = spec["env_id"] not in ("Pendulum-v1", "MountainCarContinuous-v0") and not spec.get("wrapper_kwargs", {}).get("modified_rewards")
    ret, length, alive = acc.ret.cpu().numpy(), acc.length.cpu().numpy(), acc.alive.cpu().numpy()
    for i in list(range(0, n, max(1, n // 300))):
        tot, depth, live = 0.0, 0, True
        for k in range(K):
            if not live:
                break
            tot += rew[k, i] * 0.97 ** depth
            depth += 1
            if done[k, i]:
                live = False
        assert length[i] == depth and bool(alive[i]) == live
        if exact:
            assert ret[i] == tot, (i, ret[i], tot)
        else:
            assert abs(ret[i] - tot) <= 1e-5 * max(1.0, abs(tot))
    for e in (a, b, c):
        e.close()


@pytest.mark.parametrize("specialize", [False, True])
def test_table_kind_is_nsg_rollout(specialize):
    import ctypes as C

    import torch

    from ns_gym_amd import _abi as A
    from ns_gym_amd import _lib
    from tests.golden.make_golden import make_actions

    spec = TRAJ_SPECS["c2_cartpole_gravity_rw"]
    n, K = 4096, 33
    a, b = (make_env_from_spec(_vec, spec, n=n, specialize=specialize) for _ in range(2))
    a.reset(seed=3); b.reset(seed=3)
    acts = torch.from_numpy(make_actions(spec["env_id"], K, n)).cuda()
    ref = a.rollout(acts, record=REC)

    class Table:
        kind = A.NSG_POL_TABLE

        def _struct(self, env, step0, actions_out):
            return A.Policy(kind=self.kind, step0=0, seed=0, index0=0, data=acts.data_ptr(), n_data=0, reserved0=0, actions_out=None)

    out = b.rollout_policy(Table(), K, record=REC)
    for k in REC:
        assert torch.equal(out[k], ref[k]), k
    _same_batch(a, b)
    a.close(); b.close()


@pytest.mark.parametrize("name,seed", [("c3_frozenlake_step50", 1), ("frozenlake_decrement", 2), ("frozenlake_4x4_drift_rewards", 3),
                                       ("cliff_decrement", 4), ("bridge_split_onehot", 5)])
@pytest.mark.parametrize("specialize", [False, True])
def test_tabular_policy_closed_loop(name, seed, specialize):
    """`action = policy[observation]` (tutorial.ipynb cell 12) fused: equal to a step() loop that looks the action up in the same
    table on the host side of the launch boundary."""
    import torch

    from ns_gym_amd.policies import EpisodeAccounts, TabularPolicy

    spec = _spec_of(name)
    n, K = 3000, 130
    a, b = (make_env_from_spec(_vec, spec, n=n, specialize=specialize) for _ in range(2))
    a.reset(seed=seed); b.reset(seed=seed)
    nS = a.cfg.nrow * a.cfg.ncol
    rng = np.random.default_rng(seed)
    pol = TabularPolicy(rng.integers(0, 4, size=nS))
    acc = EpisodeAccounts(b, gamma=None)
    # a few plain steps first, so that the first in-kernel decision reads a cell row that is not the start state
    warm = torch.from_numpy(rng.integers(0, 4, size=(3, n)).astype(np.int32)).cuda()
    for k in range(3):
        a.step(warm[k]); b.step(warm[k])
    acc.restart(alive=~(b.buf["terminated"].bool() | b.buf["truncated"].bool()))
    alive0 = acc.alive.clone().bool()
    out = b.rollout_policy(pol, K, record=("obs", "reward", "terminated", "truncated"), accounts=acc, record_actions=True)
    tot = torch.zeros(n, dtype=torch.float64, device="cuda")
    steps = torch.zeros(n, dtype=torch.int32, device="cuda")
    alive = alive0.clone()
    for k in range(K):
        act = pol(a.state).to(torch.int32)
        needs_reset = (a.buf["status"] & 1).bool()          # a pending autoreset takes no action and changes no account
        obs, r, te, tr, _ = a.step(act)
        took = ~needs_reset
        assert torch.equal(out["actions"][k][took], act[took]), k
        assert torch.equal(out["obs"][k], a.state) and torch.equal(out["reward"][k], r)
        assert torch.equal(out["terminated"][k], te) and torch.equal(out["truncated"][k], tr)
        live = alive & took
        tot += torch.where(live, r.to(torch.float64), torch.zeros_like(tot))
        steps += live.to(torch.int32)
        alive = alive & ~(live & (te | tr))
    _same_batch(a, b)
    assert torch.equal(acc.length, steps) and torch.equal(acc.alive.bool(), alive)
    if not spec.get("wrapper_kwargs", {}).get("modified_rewards"):
        assert torch.equal(acc.ret, tot)
    else:
        assert torch.allclose(acc.ret, tot, rtol=1e-6, atol=1e-6)
    a.close(); b.close()


@pytest.mark.parametrize("name", ["c1_cartpole_masspole_inc", "c2_cartpole_gravity_rw", "c4_pendulum_m_inc", "acrobot_constraints", "mountaincar"])
@pytest.mark.parametrize("specialize", [False, True])
def test_linear_policy_closed_loop(name, specialize):
    """A linear policy on the float32 observation, fused: equal to a step() loop whose actions the host computes from the
    observation of the previous step with the same float32 operation order."""
    import torch

    from ns_gym_amd.policies import EpisodeAccounts, LinearPolicy

    spec = TRAJ_SPECS[name]
    n, K = 2048, 90
    a, b = (make_env_from_spec(_vec, spec, n=n, specialize=specialize) for _ in range(2))
    a.reset(seed=17); b.reset(seed=17)
    rng = np.random.default_rng(5)
    rows = 1 if a.action_is_float else a.n_actions
    pol = LinearPolicy(rng.normal(size=(rows, a.obs_dim + 1)).astype(np.float32))
    lo, hi = a.spec.env_type.action_low, a.spec.env_type.action_high
    acc = EpisodeAccounts(b, gamma=0.999, horizon=K + 1)
    out = b.rollout_policy(pol, K, record=("obs", "reward", "terminated", "truncated"), accounts=acc, record_actions=True)
    for k in range(K):
        act_np = pol.decide(a.state.cpu().numpy(), a.action_is_float, lo, hi)
        act = torch.from_numpy(act_np).cuda()
        assert torch.equal(pol(a.state, a.action_is_float, lo, hi), act)       # the torch evaluation (for step() loops) decides alike
        needs_reset = (a.buf["episode"] & 1).bool() if a.spec.class_name != "PendulumEnv" else (a.t >= a.cfg.max_episode_steps)
        obs, r, te, tr, _ = a.step(act)
        took = ~needs_reset
        assert torch.equal(out["actions"][k][took], act[took]), (k, name)
        assert torch.equal(out["obs"][k], a.state), k
        assert torch.equal(out["reward"][k], r) and torch.equal(out["terminated"][k], te) and torch.equal(out["truncated"][k], tr)
    _same_batch(a, b)
    assert int(acc.length.min()) >= 1
    a.close(); b.close()


def test_policy_argument_errors():
    from ns_gym_amd._lib import NsgError
    from ns_gym_amd.policies import LinearPolicy, TabularPolicy

    cp = make_env_from_spec(_vec, TRAJ_SPECS["c1_cartpole_masspole_inc"], n=256)
    fl = make_env_from_spec(_vec, TRAJ_SPECS["c3_frozenlake_step50"], n=256)
    cp.reset(seed=0); fl.reset(seed=0)
    with pytest.raises(NsgError, match="grid envs only"):
        cp.rollout_policy(TabularPolicy(np.zeros(64)), 4)
    with pytest.raises(NsgError, match="one action per cell"):
        fl.rollout_policy(TabularPolicy(np.zeros(10)), 4)
    with pytest.raises(NsgError, match="classic-control"):
        fl.rollout_policy(LinearPolicy(np.zeros((4, 2), dtype=np.float32)), 4)
    with pytest.raises(NsgError, match="weight rows"):
        cp.rollout_policy(LinearPolicy(np.zeros((3, 5), dtype=np.float32)), 4)
    cp.close(); fl.close()


# ---- the reference's own loops (tests/golden/policy_*.npz; the oracle is pinned to the same files by tests/test_oracle_policy_rollouts.py) ----
from tests.policy_cases import EPISODE_CASES, MCTS_CASES, HipSide, run_episode_case, run_mcts_case  # noqa: E402


@pytest.mark.parametrize("specialize", [False, True])
@pytest.mark.parametrize("name", sorted(MCTS_CASES))
def test_default_policy_matches_reference_mcts(name, specialize):
    """`MCTS._default_policy` (MCTS.py:162-181) on copies made the way `MCTS.search` makes them: the reference's `tot_reward` and
    step count, bit for bit, from ONE nsg_rollout_policy launch per fixture."""
    run_mcts_case(HipSide, name, specialize=specialize)


@pytest.mark.parametrize("specialize", [False, True])
@pytest.mark.parametrize("name", sorted(EPISODE_CASES))
def test_closed_loop_matches_reference_run_episode(name, specialize):
    """`run_episode` (run_experiment.py:91-148) with a linear agent: total reward, number of steps and every action of every
    episode, from ONE launch."""
    run_episode_case(HipSide, name, specialize=specialize)


# ---- the harness and the planner-side helper on top of the fused rollouts ----------------------------------------------------
def test_run_episodes_with_fused_policies_equals_the_step_loop():
    """`run_episodes` with a policy descriptor (one launch per K-step chunk) returns the rows of the same policy driven through
    `step()` as a Python callable - totals, step counts and SARNS records."""
    import torch

    from ns_gym_amd.evaluate import run_episodes
    from ns_gym_amd.policies import TabularPolicy

    spec = _spec_of("frozenlake_decrement")
    n = 600
    rng = np.random.default_rng(3)
    env = make_env_from_spec(_vec, spec, n=n)
    pol = TabularPolicy(rng.integers(0, 4, size=env.cfg.nrow * env.cfg.ncol))
    fused = run_episodes(env, pol, seed=21, record_sarns=True, chunk=16)
    loop = run_episodes(env, lambda state: pol(state).to(torch.int32), seed=21, record_sarns=True, chunk=16)
    assert [r[0] for r in fused] == [r[0] for r in loop] and [r[2] for r in fused] == [r[2] for r in loop]
    for i in (0, 5, n - 1):
        assert fused[i][1] == loop[i][1] and len(fused[i][1]) == fused[i][2]
    env.close()


def test_default_policy_of_run_episodes_is_the_in_kernel_uniform_source():
    from ns_gym_amd.evaluate import run_episodes
    from ns_gym_amd.policies import UniformRandom

    spec = TRAJ_SPECS["c1_cartpole_masspole_inc"]
    env = make_env_from_spec(_vec, spec, n=2048)
    a = run_episodes(env, seed=5, as_arrays=True, chunk=32)
    import torch
    table = UniformRandom(seed=5).actions(env, 0, 501)
    b = run_episodes(env, seed=5, actions=table, as_arrays=True, chunk=50)
    assert a["num_steps"].tolist() == b["num_steps"].tolist() and a["total_reward"].tolist() == b["total_reward"].tolist()
    assert (a["total_reward"] == a["num_steps"]).all() and a["num_steps"].max() < 200
    env.close()


@pytest.mark.parametrize("name", ["c1_cartpole_masspole_inc", "frozenlake_decrement"])
def test_simulator_equals_the_loops_it_replaces(name):
    """planning.Simulator (fork into a standing batch + one fused rollout) against the same simulations made one by one the way
    MCTS.search makes them: deepcopy, the chance node's step, then the default policy's loop through step()."""
    import torch

    from ns_gym_amd.planning import Simulator
    from ns_gym_amd.policies import UniformRandom

    spec = _spec_of(name)
    R, S, d, gamma = 96, 5, 25, 0.95
    env = make_env_from_spec(_vec, {**spec, "flags": {**spec["flags"], "change_notification": True, "delta_change_notification": True}}, n=R)
    env.reset(seed=11)
    rng = np.random.default_rng(0)
    for _ in range(3):
        env.step(torch.from_numpy(rng.integers(0, env.n_actions, size=R).astype(np.int32)).cuda())
    plan = env.get_planning_env()
    sim = Simulator(plan, sims=S, depth=d, gamma=gamma)
    first = torch.from_numpy(rng.integers(0, env.n_actions, size=(S, R)).astype(np.int32)).cuda()
    out = sim.run(seed=77, first_actions=first, entropy=555)
    pol = UniformRandom(seed=77)
    # the same, one simulation at a time - with the env streams of both sides set to the same seeds, so that the slip outcomes agree
    sim.src.fork(theta_mode=0, into=sim.copies, entropy=555)
    sim.copies.seed_streams(np.arange(S * R, dtype=np.uint64) + np.uint64(1000), which="env")
    _, r0, te0, tr0, _ = sim.copies.step(first.reshape(-1))
    r0 = r0.clone()                                  # (step() hands out views of the handle's own rows: the rollout below rewrites them)
    alive0 = ~(te0 | tr0)
    sim.acc.restart(alive=alive0)
    sim.copies.rollout_policy(UniformRandom(seed=77), d, accounts=sim.acc)
    ret = sim.acc.ret.view(S, R).cpu().numpy()
    length = sim.acc.length.view(S, R).cpu().numpy()
    table = pol.table(sim.copies, 0, d).reshape(d, S, R)
    for s in range(S):
        c = plan.fork(theta_mode=0, entropy=555)
        c.seed_streams(np.arange(R, dtype=np.uint64) + np.uint64(1000 + s * R), which="env")
        _, r, te, tr, _ = c.step(first[s])
        assert torch.equal(r, r0.view(S, R)[s])
        tot = np.zeros(R); depth = np.zeros(R, dtype=np.int64); live = ~(te | tr).cpu().numpy()
        for k in range(d):
            _, r, te, tr, _ = c.step(torch.from_numpy(table[k, s]).cuda())
            rr = r.cpu().numpy().astype(np.float64)
            tot = np.where(live, tot + rr * gamma ** depth, tot)
            depth = depth + live
            live = live & ~(te | tr).cpu().numpy()
        np.testing.assert_array_equal(length[s], depth)
        np.testing.assert_allclose(ret[s], tot, rtol=1e-12, atol=1e-12)
        c.close()
    assert out["ret"].shape == (S, R) and out["first_reward"].shape == (S, R)
    sim.close(); plan.close(); env.close()


def test_sharded_policy_rollouts_equal_the_unsharded_job():
    """Seeds and in-kernel action draws are functions of the GLOBAL env index (`reset(seed=base + i)`, `UniformRandom(index0=...)`):
    a job cut into shards - one handle per GPU in the multi-GPU layout (ns_gym_amd/distributed.py) - walks the same trajectories and
    keeps the same accounts as the unsharded batch."""
    import torch

    from ns_gym_amd.policies import EpisodeAccounts, UniformRandom

    spec = TRAJ_SPECS["c2_cartpole_gravity_rw"]
    n, K, base = 6144, 80, 40
    full = make_env_from_spec(_vec, spec, n=n)
    full.reset(seed=base)
    acc = EpisodeAccounts(full, gamma=0.99, horizon=K + 1)
    ref = full.rollout_policy(UniformRandom(seed=3), K, record=("obs", "reward", "terminated"), accounts=acc)
    cuts = [0, 1024, 4096, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = make_env_from_spec(_vec, spec, n=hi - lo)
        part.reset(seed=np.arange(lo, hi, dtype=np.uint64) + np.uint64(base))
        pacc = EpisodeAccounts(part, gamma=0.99, horizon=K + 1)
        out = part.rollout_policy(UniformRandom(seed=3, index0=lo), K, record=("obs", "reward", "terminated"), accounts=pacc)
        for k in out:
            assert torch.equal(out[k], ref[k][:, lo:hi]), (k, lo)
        assert torch.equal(pacc.ret, acc.ret[lo:hi]) and torch.equal(pacc.length, acc.length[lo:hi])
        assert torch.equal(part.theta, full.theta[:, lo:hi])
        part.close()
    full.close()


def test_context_sweep_by_per_env_theta_equals_the_reference_idiom():
    """The reference installs a context value with ContinuousScheduler(start=0, end=0) + StepWiseUpdate([value]) - one env per
    value (context_switching.py:49-52).  θ being per-env state, a batch with persistent_params and the values written into its θ
    row walks the same trajectories (only the t = 0 notification differs): checked per context against the reference's idiom."""
    import torch

    from ns_gym_amd.envs import make
    from ns_gym_amd.policies import EpisodeAccounts, LinearPolicy
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.update_functions import NoUpdate, StepWiseUpdate

    values, E, K = [0.05, 0.4, 1.7, 3.0], 256, 300
    pol = LinearPolicy([[0.3, -0.8, -2.0, -1.1, 0.05], [-0.3, 0.8, 2.0, 1.1, -0.05]])
    sweep = _vec(make("CartPole-v1"), {"masspole": NoUpdate(ContinuousScheduler())}, len(values) * E, persistent_params=True)
    seeds = np.concatenate([np.arange(E, dtype=np.uint64) + np.uint64(50)] * len(values))
    sweep.reset(seed=seeds)
    sweep.theta[0].copy_(torch.from_numpy(np.repeat(values, E)).cuda())
    acc = EpisodeAccounts(sweep, gamma=None)
    out = sweep.rollout_policy(pol, K, record=("obs", "reward", "terminated", "truncated"), accounts=acc)
    for j, v in enumerate(values):
        one = _vec(make("CartPole-v1"), {"masspole": StepWiseUpdate(ContinuousScheduler(start=0, end=0), [v])}, E)
        one.reset(seed=50)
        a1 = EpisodeAccounts(one, gamma=None)
        o1 = one.rollout_policy(pol, K, record=("obs", "reward", "terminated", "truncated"), accounts=a1)
        sl = slice(j * E, (j + 1) * E)
        # identical while the first episode runs (afterwards the reference idiom re-installs the value at t = 0 of the next episode -
        # one step with the default mass - where the persistent row simply keeps it)
        first = a1.length.cpu().numpy()
        assert torch.equal(acc.length[sl], a1.length) and torch.equal(acc.ret[sl], a1.ret)
        for i in (0, 17, E - 1):
            L = int(first[i])
            assert torch.equal(out["obs"][:L, j * E + i], o1["obs"][:L, i]) and torch.equal(out["terminated"][:L, j * E + i], o1["terminated"][:L, i])
        one.close()
    assert not torch.equal(out["obs"][60, :E], out["obs"][60, E:2 * E])      # same seeds, different contexts: different trajectories
    sweep.close()
