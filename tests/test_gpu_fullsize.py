"""BASELINE.json's full sizes (N = 2^20 per GPU) through size-independent properties:
  * env independence: any subset of a 2^20-env batch equals the oracle run on those seeds alone
    (same seeds, same per-env actions) — bit-exact integers / flags, 1e-5 on float32 state;
  * determinism: two identical runs agree bit for bit;
  * accounting: ballot-reduced counters == sums of the per-env outputs; FrozenLake slip
    distributions stay normalised; done-mask compaction == nonzero(terminated | truncated)."""
import numpy as np
import pytest

from tests.util import TRAJ_SPECS, OracleView, compare_views, make_env_from_spec

pytestmark = pytest.mark.gpu

N = 1 << 20


def _vec(*a, **k):
    from ns_gym_amd.vec_env import VecNSEnv

    return VecNSEnv(*a, **k)


@pytest.mark.parametrize("name,T", [("c1_cartpole_masspole_inc", 120), ("c3_frozenlake_step50", 130),
                                    ("c2_cartpole_gravity_rw", 40)])
def test_subset_of_full_batch_equals_oracle(name, T):
    import torch

    from oracle.oracle import OracleVecEnv

    spec = TRAJ_SPECS[name]
    is_fl = spec["env_id"] == "FrozenLake-v1"
    env = make_env_from_spec(_vec, spec, n=N, track_returns=True)
    env.reset(seed=1000)
    pick = np.sort(np.random.default_rng(0).choice(N, size=2048, replace=False))
    pick = np.unique(np.concatenate([pick, [0, 1, 63, 64, N - 1, N - 64, N - 65]]))
    n_act = 4 if is_fl else 2
    g = torch.Generator(device="cuda").manual_seed(5)
    acts = torch.randint(0, n_act, (T, N), dtype=torch.int32, device="cuda", generator=g)
    sub = acts[:, torch.from_numpy(pick).cuda()].cpu().numpy()
    o = OracleView(make_env_from_spec(OracleVecEnv, spec, n=len(pick), track_returns=True))
    ob = o.reset((pick + 1000).astype(np.uint64))
    tot_done = tot_fired = 0
    for k in range(T):
        obs, rew, term, trunc, info = env.step(acts[k])
        ob = o.step(sub[k])
        tot_done += int((term | trunc).sum())
        tot_fired += int(sum(v.sum() for v in info["Ground Truth Env Change"].values()))
        if k % 10 == 0 or k == T - 1:
            idx = torch.from_numpy(pick).cuda()
            P = max(env.cfg.n_params, 1)
            a = {"state": env.state[idx].cpu().numpy(), "reward": rew[idx].cpu().numpy(),
                 "terminated": term[idx].cpu().numpy().astype(np.uint8), "truncated": trunc[idx].cpu().numpy().astype(np.uint8),
                 "env_change": env.gt_env_change[:P, idx].cpu().numpy(), "delta_change": env.gt_delta_change[:P, idx].cpu().numpy(),
                 "t": env.t[idx].cpu().numpy(), "theta": env.theta[:, idx].cpu().numpy()}
            if is_fl:
                a["prob"] = env.prob[idx].cpu().numpy()
            compare_views(a, ob, is_fl, f"step {k}")
    c = env.counters()
    assert c["episodes"] == tot_done and c["updates_applied"] == tot_fired
    assert c["env_steps"] + c["episodes"] - int((env.terminated | env.truncated).sum()) == N * T  # every call is a step or a reset
    want = torch.nonzero(env.terminated | env.truncated).flatten().cpu().numpy()
    np.testing.assert_array_equal(np.sort(env.done_indices().cpu().numpy()), want)
    if is_fl:
        s = env.theta.sum(dim=0)
        assert float((s - 1).abs().max()) < 1e-12
    env.close()


@pytest.mark.parametrize("name,n,T", [("c1_cartpole_masspole_inc", N, 200), ("c2_cartpole_gravity_rw", 65536, 1000),
                                      ("c3_frozenlake_step50", N, 200), ("c4_pendulum_m_inc", 262144, 200)])
def test_every_env_of_the_full_batch_equals_oracle(name, n, T):
    """Not a sample: ALL envs of the BASELINE configurations at BASELINE's own batch sizes and step counts (C1's config at
    C5's per-GPU size; C2 65 536 x 1000; C3 2^20 x 200; C4's Pendulum half 2^18 x 200 - the Acrobot half has its own test
    below) against the oracle (its OpenMP stepper on the host's cores), every compared row at checkpoints and at the end."""
    N = n
    import os

    import torch

    from oracle.oracle import OracleVecEnv
    from tests.util import GpuView

    spec = TRAJ_SPECS[name]
    is_fl = spec["env_id"] == "FrozenLake-v1"
    env = make_env_from_spec(_vec, spec, n=N, track_returns=True, specialize=True)
    orc = make_env_from_spec(OracleVecEnv, spec, n=N, track_returns=True)
    seeds = np.arange(N, dtype=np.uint64) + np.uint64(4242)
    env.reset(seed=seeds)
    orc.reset(seed=seeds)
    view, oview = GpuView(env), OracleView(orc)
    g = torch.Generator(device="cuda").manual_seed(17)
    threads = min(16, os.cpu_count() or 1)
    for k in range(T):
        if env.action_is_float:
            a = torch.rand(N, device="cuda", generator=g) * 4 - 2
        else:
            a = torch.randint(0, env.n_actions, (N,), dtype=torch.int32, device="cuda", generator=g)
        env.step(a)
        orc.step_mt(a.cpu().numpy(), threads)
        if k % 50 == 0 or k == T - 1:
            compare_views(view._out(), oview._out(), is_fl, f"{name}: all {N} envs, step {k}")
    env.close()


@pytest.mark.parametrize("n", [49151, 49152, 163840, 163841, 393216, 393217, 393472, 1048575, 1048577, 1310720])
@pytest.mark.parametrize("name", ["c1_cartpole_masspole_inc", "c2_cartpole_gravity_rw"])
def test_batch_sizes_either_side_of_the_launch_policies(name, n):
    """The specialised CartPole kernels change shape with the batch size: resets in-lane for 49 152 .. 163 840 envs
    (nsg_specialize), one workgroup per chunk up to 1536 chunks (393 216 envs), 1536 workgroups walking 2-3 chunks up to 2^20
    (step_grid_for), one workgroup per chunk again beyond 2^20 envs (launch_grid_for).  Every env of batches one env either side of each threshold (and of a ragged last chunk) against the
    oracle for 60 steps - long enough for every env to have finished at least one episode."""
    import os

    import torch

    from oracle.oracle import OracleVecEnv
    from tests.util import GpuView

    spec = TRAJ_SPECS[name]
    env = make_env_from_spec(_vec, spec, n=n, track_returns=True, specialize=True)
    orc = make_env_from_spec(OracleVecEnv, spec, n=n, track_returns=True)
    env.reset(seed=31)
    orc.reset(seed=np.arange(n, dtype=np.uint64) + np.uint64(31))
    view, oview = GpuView(env), OracleView(orc)
    g = torch.Generator(device="cuda").manual_seed(n)
    threads = min(16, os.cpu_count() or 1)
    for k in range(60):
        a = torch.randint(0, env.n_actions, (n,), dtype=torch.int32, device="cuda", generator=g)
        env.step(a)
        orc.step_mt(a.cpu().numpy(), threads)
        if k % 20 == 19:
            compare_views(view._out(), oview._out(), False, f"{name}: all {n} envs, step {k}")
    assert env.counters()["episodes"] >= n
    env.close()


ACROBOT_MAX_SPLITS = 16      # envs whose episodes END one step apart, of 262 144 envs over 200 steps (seen: 0-5)
ACROBOT_MAX_DRIFTS = 64      # envs whose float32 observation drifts past the bar inside one long episode (measured: 16)
ACROBOT_DRIFT_MIN_AGE = 100  # ... and only in an episode at least this many steps old (measured: first at 115)


class AcrobotAllowance:
    """Per-step bookkeeping of the STATED allowance for C4's Acrobot half (see the docstring of
    test_c4_acrobot_full_horizon_with_stated_allowance): feed it the batch and the oracle after every step, `conclude()`
    asserts the classification and the bounds."""

    def __init__(self, n):
        self.n = n
        self.split = np.zeros(n, dtype=bool)
        self.first = {}

    def after_step(self, k, env, orc):
        from tests.util import STATE_ATOL, STATE_RTOL, THETA_RTOL

        st, so = env.state.cpu().numpy(), orc.a["obs"]
        state_ok = (np.abs(st - so) <= STATE_ATOL + STATE_RTOL * np.abs(so)).all(axis=1)
        th, tho = env.theta.cpu().numpy()[0], orc.a["theta"][0]
        term = env.terminated.cpu().numpy().astype(np.uint8)
        flags_ok = ((term == orc.a["terminated"])
                    & (env.truncated.cpu().numpy().astype(np.uint8) == orc.a["truncated"]) & (env.t.cpu().numpy() == orc.a["t"])
                    & (env.gt_env_change.cpu().numpy()[0] == orc.a["env_change"][0])
                    & (np.abs(env.reward.cpu().numpy() - orc.a["reward"]) <= 1e-5)
                    & (np.abs(th - tho) <= 1e-12 + THETA_RTOL * np.abs(tho)))
        bad = ~(state_ok & flags_ok) & ~self.split
        if bad.any():
            err = (np.abs(st - so) / (STATE_ATOL + STATE_RTOL * np.abs(so))).max(axis=1)
            for i in np.nonzero(bad)[0]:
                term_differs = bool(term[i]) != bool(orc.a["terminated"][i])
                self.first[int(i)] = dict(step=k, term_differs=term_differs, state_ok=bool(state_ok[i]), flags_ok=bool(flags_ok[i]),
                                          age=int(orc.a["t"][i]), err_in_bars=float(err[i]))
        self.split |= bad

    def conclude(self, what):
        first = self.first
        print(f"{what}: {int(self.split.sum())} of {self.n} envs left the bar:")
        for i, d in sorted(first.items(), key=lambda kv: kv[1]["step"]):
            print("   env", i, d)
        n_term = sum(1 for d in first.values() if d["term_differs"] and d["state_ok"])
        n_drift = sum(1 for d in first.values() if not d["term_differs"] and d["flags_ok"] and not d["state_ok"])
        assert n_term + n_drift == len(first), "an env left the bar in a way that is neither a termination-boundary split nor a gradual drift"
        assert n_term <= ACROBOT_MAX_SPLITS and n_drift <= ACROBOT_MAX_DRIFTS, (n_term, n_drift)
        for d in first.values():
            if not d["term_differs"]:   # amplified inside an old episode, never in a young one
                assert d["age"] >= ACROBOT_DRIFT_MIN_AGE, d
        return n_term, n_drift


def test_c4_acrobot_full_horizon_with_stated_allowance():
    """C4's Acrobot half at BASELINE's own size and step count: 262 144 envs x 200 steps, EVERY env compared with the oracle
    after EVERY step.  The bar for an env is the suite's usual one (float32 state |d| <= 1e-5 * max(1, |x|), flags / t exact,
    theta 1e-5 relative).  Stated allowance: the double pendulum is chaotic, and C4's config makes it stiffer with every step
    (LINK_MASS_2 += 0.1 per step, RK4 at dt = 0.2): the last-ulp differences between the kernels' sincos and libm's (both
    <= 1 ulp; 3 % of the evaluations differ) are amplified step by step inside an episode until, in episodes more than a hundred
    steps old, a few envs' float32 observations part by more than the bar - or the termination test
    `-cos(th1) - cos(th1 + th2) > 1.0` falls on different sides, so that the two implementations end the episode one step
    apart.  Every reset wipes the difference (the initial states are bit-identical).  Asserted:
      * an env may leave the bar in exactly two ways: (a) its `terminated` flags disagree on a step where its state still
        agrees (termination boundary), at most ACROBOT_MAX_SPLITS envs; (b) its observation drifts past the bar with every
        flag, t, reward and theta still equal, at most ACROBOT_MAX_DRIFTS envs, and only in an episode that is at least
        ACROBOT_DRIFT_MIN_AGE steps old - a per-step arithmetic error would show in young episodes first;
      * every other env (>= 99.97 %) is within the bar after every one of the 200 steps."""
    import os

    import torch

    from oracle.oracle import OracleVecEnv

    n, T = 262144, 200
    spec = TRAJ_SPECS["c4_acrobot_mass2_inc"]
    env = make_env_from_spec(_vec, spec, n=n, track_returns=True, specialize=True)
    orc = make_env_from_spec(OracleVecEnv, spec, n=n, track_returns=True)
    seeds = np.arange(n, dtype=np.uint64) + np.uint64(4242)
    env.reset(seed=seeds)
    orc.reset(seed=seeds)
    g = torch.Generator(device="cuda").manual_seed(17)
    threads = min(16, os.cpu_count() or 1)
    allow = AcrobotAllowance(n)
    for k in range(T):
        a = torch.randint(0, env.n_actions, (n,), dtype=torch.int32, device="cuda", generator=g)
        env.step(a)
        orc.step_mt(a.cpu().numpy(), threads)
        allow.after_step(k, env, orc)
    allow.conclude(f"Acrobot 2^18 x {T}")
    env.close()


@pytest.mark.parametrize("specialize", [False, True], ids=["generic-group-kernel", "specialised-group-unit"])
def test_c4_mixed_group_launch_at_full_size_equals_oracle(specialize):
    """BASELINE config C4 as north_star states it: Pendulum 262 144 + Acrobot 262 144 stepped by ONE heterogeneous launch
    (`nsg_step_group`: 2048 workgroups, a block range per env type) for 200 steps, EVERY env of both members against the
    oracle - the Pendulum member within the suite's bar after every step, the Acrobot member after every step under the
    stated allowance of the test above - once through the precompiled group kernel and once through the unit that is
    compiled for the ordered pair of the members' configurations (`nsg_spec_group`); which of the two ran is read back from
    the library, not assumed."""
    import os

    import torch

    from ns_gym_amd.vec_env import step_group, step_group_kind
    from oracle.oracle import OracleVecEnv
    from tests.util import GpuView

    n, T = 262144, 200
    names = ("c4_pendulum_m_inc", "c4_acrobot_mass2_inc")
    envs = [make_env_from_spec(_vec, TRAJ_SPECS[nm], n=n, track_returns=True, specialize=specialize) for nm in names]
    orcs = [make_env_from_spec(OracleVecEnv, TRAJ_SPECS[nm], n=n, track_returns=True) for nm in names]
    assert [e.specialized for e in envs] == [specialize, specialize]
    seeds = np.arange(n, dtype=np.uint64) + np.uint64(4242)
    for e, o in zip(envs, orcs):
        e.reset(seed=seeds)
        o.reset(seed=seeds)
    pend, acro = envs
    pview, poview = GpuView(pend), OracleView(orcs[0])
    g = torch.Generator(device="cuda").manual_seed(29)
    threads = min(16, os.cpu_count() or 1)
    allow = AcrobotAllowance(n)
    for k in range(T):
        ap = torch.rand(n, device="cuda", generator=g) * 4 - 2
        aa = torch.randint(0, acro.n_actions, (n,), dtype=torch.int32, device="cuda", generator=g)
        step_group([pend, acro], [ap, aa])
        orcs[0].step_mt(ap.cpu().numpy(), threads)
        orcs[1].step_mt(aa.cpu().numpy(), threads)
        allow.after_step(k, acro, orcs[1])
        if k % 10 == 0 or k == T - 1:
            compare_views(pview._out(), poview._out(), False, f"C4 group launch, Pendulum member: all {n} envs, step {k}")
    assert step_group_kind([pend, acro]) == ("specialised (prebuilt)" if specialize else "generic")   # C4's own unit ships with the library
    allow.conclude(f"C4 group launch ({'specialised unit' if specialize else 'generic kernel'}), Acrobot member 2^18 x {T}")
    # the launch's ballot counters: every call of every env was a step or a reset
    for e in envs:
        c = e.counters()
        assert c["env_steps"] + c["episodes"] - int((e.terminated | e.truncated).sum()) == n * T
    for e in envs:
        e.close()


def test_full_batch_is_deterministic():
    import torch

    spec = TRAJ_SPECS["c2_cartpole_gravity_rw"]
    outs = []
    for _ in range(2):
        env = make_env_from_spec(_vec, spec, n=N)
        env.reset(seed=7)
        g = torch.Generator(device="cuda").manual_seed(11)
        for k in range(30):
            env.step(torch.randint(0, 2, (N,), dtype=torch.int32, device="cuda", generator=g))
        outs.append((env.state.clone(), env.theta.clone(), env.t.clone()))
        env.close()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["c1_cartpole_masspole_inc", "c3_frozenlake_step50"])
def test_maximum_batch_per_handle(name):
    """N = 2^27, the largest batch one handle takes (rows are addressed with 32-bit byte offsets; the
    32-byte stream records reach offset 2^32 - 32): the envs at the very end of the batch must behave
    exactly like the same seeds stepped in a small batch."""
    import torch

    spec = TRAJ_SPECS[name]
    is_fl = spec["env_id"] == "FrozenLake-v1"
    big_n, tail, T = 1 << 27, 1000, 12
    free, _ = torch.cuda.mem_get_info()
    if free < 40 << 30:
        pytest.skip("needs ~25 GB of device memory")
    big = make_env_from_spec(_vec, spec, n=big_n, track_returns=True)
    small = make_env_from_spec(_vec, spec, n=tail, track_returns=True)
    big.reset(seed=5)                                                     # env i <- seed 5 + i
    small.reset(seed=(np.arange(big_n - tail, big_n, dtype=np.uint64) + np.uint64(5)))
    g = torch.Generator(device="cuda").manual_seed(3)
    acts = torch.randint(0, 4 if is_fl else 2, (big_n,), dtype=torch.int32, device="cuda", generator=g)
    for k in range(T):
        a = torch.roll(acts, k)                                           # a different action vector per step, no new 512 MB draws
        big.step(a)
        small.step(a[big_n - tail:].contiguous())
    for attr in ("state", "t", "reward", "terminated", "truncated", "theta"):
        x, y = getattr(big, attr), getattr(small, attr)
        assert torch.equal(x[..., big_n - tail:] if attr == "theta" else x[big_n - tail:], y), attr
    c = big.counters()
    assert c["env_steps"] + c["episodes"] - int((big.terminated | big.truncated).sum()) == big_n * T
    big.close(); small.close()


def test_c4_group_rollout_at_full_size_equals_group_steps():
    """C4 at BASELINE's size through the fused group rollout: Pendulum 262 144 + Acrobot 262 144, 200 steps as 5 launches of
    K = 40 (`nsg_rollout_group`, the specialised group unit), against the same 200 steps as single `nsg_step_group` launches -
    which the test above holds against the oracle: every persistent row and every recorded step bit-identical."""
    import torch

    from ns_gym_amd.vec_env import rollout_group, step_group, step_group_kind

    n, K, reps = 262144, 40, 5
    names = ("c4_pendulum_m_inc", "c4_acrobot_mass2_inc")
    fused = [make_env_from_spec(_vec, TRAJ_SPECS[nm], n=n, track_returns=True, specialize=True) for nm in names]
    stepped = [make_env_from_spec(_vec, TRAJ_SPECS[nm], n=n, track_returns=True, specialize=True) for nm in names]
    for e in fused + stepped:
        e.reset(seed=4242)
    g = torch.Generator(device="cuda").manual_seed(31)
    for r in range(reps):
        ap = torch.rand((K, n), device="cuda", generator=g) * 4 - 2
        aa = torch.randint(0, 3, (K, n), dtype=torch.int32, device="cuda", generator=g)
        outs = rollout_group(fused, [ap, aa], record=("obs", "reward", "terminated", "truncated"))
        for k in range(K):
            step_group(stepped, [ap[k], aa[k]])
            if k % 13 == 0 or k == K - 1:
                for e, o in zip(stepped, outs):
                    assert torch.equal(o["obs"][k], e.state) and torch.equal(o["reward"][k], e.reward)
                    assert torch.equal(o["terminated"][k], e.terminated) and torch.equal(o["truncated"][k], e.truncated)
    assert step_group_kind(fused).startswith("specialised")
    for a, b, nm in zip(fused, stepped, names):
        for row in ("phys", "theta", "t", "episode", "obs", "reward", "terminated", "truncated", "ep_return", "last_return", "last_length"):
            assert torch.equal(a.buf[row], b.buf[row]), (nm, row)
        assert a.counters() == b.counters(), nm
    for e in fused + stepped:
        e.close()


def test_fused_policy_rollouts_at_full_size():
    """BASELINE's sizes through the fused closed loops (nsg_rollout_policy), by properties that do not need an oracle run:
      * C1's config, 2^20 envs x 200 steps, uniform in-kernel actions: every row equals nsg_rollout over the same action table
        (computed on the device by the torch mirror of the draw), and CartPole's accounts have a closed form - the discounted
        return of an episode of L steps is the L-th sequential partial sum of the discount table, bit for bit;
      * C3's config, 2^20 envs x 200 steps, a tabular policy decided in the kernel: equals 200 single steps whose actions a torch
        gather looks up in the same table; a FrozenLake account is gamma ** (L - 1) if the episode reached the goal, else 0."""
    import torch

    from ns_gym_amd.policies import EpisodeAccounts, TabularPolicy, UniformRandom

    K, chunks, gamma = 50, 4, 0.99
    disc = np.array([gamma ** j for j in range(K * chunks + 1)])
    # ---- C1 ----
    spec = TRAJ_SPECS["c1_cartpole_masspole_inc"]
    a, b = make_env_from_spec(_vec, spec, n=N), make_env_from_spec(_vec, spec, n=N)
    a.reset(seed=7); b.reset(seed=7)
    pol = UniformRandom(seed=99)
    acc = EpisodeAccounts(b, gamma=gamma, horizon=K * chunks + 1)
    for c in range(chunks):
        table = pol.actions(a, c * K, K)
        if c == 0:
            assert np.array_equal(table[:3, :4096].cpu().numpy(), pol.table(a, 0, 3)[:, :4096])     # the torch mirror IS the NumPy one
        ra = a.rollout(table, record=("reward", "terminated"))
        rb = b.rollout_policy(pol, K, record=("reward", "terminated"), accounts=acc, step0=c * K)
        assert torch.equal(ra["reward"], rb["reward"]) and torch.equal(ra["terminated"], rb["terminated"])
    for row in ("theta", "t", "state", "phys"):
        assert torch.equal(getattr(a, row), getattr(b, row)), row
    assert torch.equal(a.buf["episode"], b.buf["episode"]) and a.counters() == b.counters()
    length = acc.length.cpu().numpy()
    assert length.min() >= 8 and length.max() <= K * chunks and not acc.alive.cpu().numpy()[length < K * chunks].any()
    partial = np.zeros(K * chunks + 1)
    s = 0.0
    for j in range(K * chunks):          # tot_reward += 1.0 * gamma ** depth, one addition at a time
        s = s + 1.0 * disc[j]
        partial[j + 1] = s
    np.testing.assert_array_equal(acc.ret.cpu().numpy(), partial[length])
    a.close(); b.close()
    # ---- C3 ----
    spec = TRAJ_SPECS["c3_frozenlake_step50"]
    a, b = make_env_from_spec(_vec, spec, n=N), make_env_from_spec(_vec, spec, n=N)
    a.reset(seed=3); b.reset(seed=3)
    tab = TabularPolicy(np.random.default_rng(1).integers(0, 4, size=64))
    acc = EpisodeAccounts(b, gamma=gamma, horizon=K * chunks + 1)
    reached = torch.zeros(N, dtype=torch.bool, device="cuda")
    live = torch.ones(N, dtype=torch.bool, device="cuda")
    for c in range(chunks):
        b.rollout_policy(tab, K, accounts=acc, step0=c * K)
        for _ in range(K):
            needs_reset = (a.buf["status"] & 1).bool()
            _, r, te, tr, _ = a.step(tab(a.state).to(torch.int32))
            took = live & ~needs_reset
            reached |= took & (r > 0)
            live &= ~(took & (te | tr))
    for row in ("theta", "t", "state"):
        assert torch.equal(getattr(a, row), getattr(b, row)), row
    assert torch.equal(a.buf["status"], b.buf["status"]) and torch.equal(a.buf["rng_env"], b.buf["rng_env"]) and a.counters() == b.counters()
    assert torch.equal(acc.alive.bool(), live)
    length = acc.length.cpu().numpy()
    want = np.where(reached.cpu().numpy(), 1.0 * disc[np.maximum(length, 1) - 1], 0.0)
    np.testing.assert_array_equal(acc.ret.cpu().numpy(), want)
    a.close(); b.close()
