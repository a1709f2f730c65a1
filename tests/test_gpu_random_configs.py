"""Seeded sweep over random wrapper configurations: every env type x 1-3 tuned parameters x a random
(Scheduler, UpdateFn) pair per parameter (shared objects now and then) x random flags x a ragged batch
size, HIP kernels against the oracle step by step.  The golden fixtures pin single components against the
reference; this sweep looks for interactions between them (constraint rejection + stochastic state, shared
objects + persistent params, autoreset + list cursors ...).  Generic and config-specialised kernels alternate."""
import numpy as np
import pytest

from tests.golden.make_golden import DIST_UPDATE_SPECS, SCALAR_UPDATE_SPECS, SCHEDULER_SPECS, make_actions
from tests.util import GpuView, OracleView, compare_views

pytestmark = pytest.mark.gpu

TUNABLE = {   # name -> a magnitude that keeps the physics away from overflow over a short run
    "CartPole-v1": {"gravity": 9.8, "masscart": 1.0, "masspole": 0.1, "force_mag": 10.0, "tau": 0.02, "length": 0.5},
    "Pendulum-v1": {"m": 1.0, "l": 1.0, "dt": 0.05, "g": 10.0},
    "Acrobot-v1": {"LINK_LENGTH_1": 1.0, "LINK_MASS_1": 1.0, "LINK_MASS_2": 1.0, "LINK_COM_POS_1": 0.5, "LINK_MOI": 1.0},
    "MountainCar-v0": {"gravity": 0.0025, "force": 0.001},
    "MountainCarContinuous-v0": {"power": 0.0015},
}
GRID = {"FrozenLake-v1": 3, "CliffWalking-v1": 4, "ns_gym/Bridge-v0": 3}
# update fns whose magnitude is independent of the parameter's scale (the others are rescaled below)
SCALAR_KINDS = ["increment", "decrement", "trend", "geometric", "expdecay", "noupdate", "randomwalk_mu_sigma", "rw_drift",
                "ou_nosigma", "stepwise", "cyclic", "lerp", "poly"]
SCHED_KINDS = ["continuous", "continuous_5_20", "periodic3", "periodic4_s2_e30", "discrete", "burst_3_2", "window", "random_p3",
               "random_p5_s3_e40"]


def scalar_fn_spec(rng, kind, scale):
    name, kw = SCALAR_UPDATE_SPECS[kind]
    kw = dict(kw)
    s = scale * 0.02
    if kind in ("increment", "decrement"):
        kw["k"] = s
    elif kind == "trend":
        kw["slope"] = s * 0.05
    elif kind == "poly":
        kw["coeffs"] = [s * 0.05, -s * 1e-3]
    elif kind == "geometric":
        kw["r"] = 1.0 + 0.01 * float(rng.uniform(-1, 1))
    elif kind in ("randomwalk_mu_sigma", "rw_drift"):
        kw.update(mu=0.0, sigma=s, seed=int(rng.integers(1, 1000)))
        if kind == "rw_drift":
            kw["alpha"] = s * 0.1
    elif kind == "ou_nosigma":
        kw.update(theta=0.2, mu=scale * 1.1)
    elif kind in ("stepwise", "cyclic"):
        vals = [scale * float(x) for x in (1.1, 0.9, 1.25, 1.0)]
        kw = {"param_list": vals} if kind == "stepwise" else {"value_list": vals}
    elif kind == "lerp":
        kw.update(start_val=scale, end_val=scale * 1.3, T=25)
    return [name, kw]


def random_spec(rng):
    env_id = str(rng.choice(list(TUNABLE) + list(GRID)))
    params = {}
    if env_id in GRID:
        nd = GRID[env_id]
        kind = str(rng.choice([k for k in DIST_UPDATE_SPECS if k not in ("d_lcbounded",)]))
        name, kw = DIST_UPDATE_SPECS[kind]
        kw = dict(kw)
        if nd == 4:   # CliffWalking's 4-way support
            for key in ("update_values", "dist_list"):
                if key in kw:
                    kw[key] = [[0.7, 0.1, 0.1, 0.1], [0.25, 0.25, 0.25, 0.25]]
            if "target" in kw:
                kw["target"] = [0.1, 0.4, 0.3, 0.2]
            if "start_dist" in kw:
                kw.update(start_dist=[1.0, 0.0, 0.0, 0.0], end_dist=[0.4, 0.2, 0.2, 0.2])
        pname = "P"
        if env_id == "ns_gym/Bridge-v0" and rng.random() < 0.5:
            pname = str(rng.choice(["P_left", "P_right"]))
        params[pname] = {"scheduler": SCHEDULER_SPECS[str(rng.choice(SCHED_KINDS))], "update": [name, kw]}
        wk = {}
        if env_id == "ns_gym/Bridge-v0" and pname != "P":
            wk["initial_prob_dist"] = {"__pair__": [[0.8, 0.1, 0.1], [0.6, 0.2, 0.2]]}
        elif rng.random() < 0.5:
            wk["initial_prob_dist"] = [0.7, 0.1, 0.1, 0.1] if nd == 4 else [0.7, 0.2, 0.1]
    else:
        names = list(rng.choice(list(TUNABLE[env_id]), size=int(rng.integers(1, min(3, len(TUNABLE[env_id])) + 1)), replace=False))
        for j, pn in enumerate(names):
            pn = str(pn)
            similar = 0.25 <= TUNABLE[env_id][pn] / TUNABLE[env_id][str(names[0])] <= 4.0
            if j > 0 and rng.random() < 0.2 and similar:   # (a fn sized for g = 10 would throw dt = 0.05 into chaos)
                params[pn] = {"same_as": str(names[0])}          # one UpdateFn object under two names
                continue
            fs = {"update": scalar_fn_spec(rng, str(rng.choice(SCALAR_KINDS)), TUNABLE[env_id][pn])}
            if j > 0 and "same_as" not in params[str(names[0])] and rng.random() < 0.2:
                fs["scheduler_of"] = str(names[0])               # one Scheduler object in two update fns
            else:
                fs["scheduler"] = SCHEDULER_SPECS[str(rng.choice(SCHED_KINDS))]
            params[pn] = fs
        wk = {}
    cn = bool(rng.random() < 0.7)
    flags = {"change_notification": cn, "delta_change_notification": bool(cn and rng.random() < 0.6),
             "persistent_params": bool(rng.random() < 0.3)}
    return {"env_id": env_id, "params": params, "flags": flags, "wrapper_kwargs": wk,
            "make_kwargs": {"map_name": "8x8"} if env_id == "FrozenLake-v1" and rng.random() < 0.5 else {}}


def _decode(spec):
    wk = dict(spec["wrapper_kwargs"])
    if isinstance(wk.get("initial_prob_dist"), dict):
        a, b = wk["initial_prob_dist"]["__pair__"]
        wk["initial_prob_dist"] = (a, b)
    return wk


@pytest.mark.parametrize("case", range(int(__import__("os").environ.get("NSG_SWEEP_CASES", "96"))))
def test_random_configuration_matches_oracle(case):
    from ns_gym_amd.envs import make
    from ns_gym_amd.spec import build_tunable_params
    from ns_gym_amd.vec_env import VecNSEnv
    from oracle.oracle import OracleVecEnv

    rng = np.random.default_rng(10_000 + case)
    spec = random_spec(rng)
    n = int(rng.choice([1, 63, 64, 65, 200, 257, 700]))
    T = 45
    kw = {**spec["flags"], **_decode(spec), "track_returns": True}
    g = GpuView(VecNSEnv(make(spec["env_id"], **spec["make_kwargs"]), build_tunable_params(spec["params"]), n,
                         specialize=bool(case % 2), **kw))
    o = OracleView(OracleVecEnv(make(spec["env_id"], **spec["make_kwargs"]), build_tunable_params(spec["params"]), n, **kw))
    seeds = rng.integers(0, 2 ** 40, size=n).astype(np.uint64)
    is_grid = spec["env_id"] in GRID
    tag = f"case {case}: {spec}"
    compare_views(g.reset(seeds), o.reset(seeds), is_grid, tag + " reset")
    acts = make_actions(spec["env_id"], T, n)
    for k in range(T):
        a, b = g.step(acts[k]), o.step(acts[k])
        if spec["env_id"] == "Acrobot-v1":   # chaotic: compare only the envs still in step
            same = a["t"] == b["t"]
            a, b = ({key: (v[..., same] if v.ndim and v.shape[-1] == n else v[same] if v.ndim and v.shape[0] == n else v)
                     for key, v in d.items()} for d in (a, b))
        compare_views(a, b, is_grid, tag + f" step {k}")
    c = g.env.counters()
    oc = o.env.a["counters"].sum(axis=1)
    if spec["env_id"] != "Acrobot-v1":
        assert [c["episodes"], c["updates_applied"], c["constraint_violations"], c["env_steps"]] == [int(x) for x in oc[:4]], tag
    g.env.close()


@pytest.mark.parametrize("case", range(int(__import__("os").environ.get("NSG_SWEEP_FORK_CASES", "48"))))
def test_random_configuration_planning_copy_matches_oracle(case):
    """The same sweep with a planning copy taken mid-run (random theta_mode, random in_sim_change, the same
    entropy on both sides): copy and source are stepped on, both must keep matching the oracle's."""
    from ns_gym_amd.envs import make
    from ns_gym_amd.spec import build_tunable_params
    from ns_gym_amd.vec_env import VecNSEnv
    from oracle.oracle import OracleVecEnv

    rng = np.random.default_rng(30_000 + case)
    spec = random_spec(rng)
    n = int(rng.choice([1, 64, 65, 300]))
    pre, post = int(rng.integers(1, 25)), 25
    kw = {**spec["flags"], **_decode(spec), "in_sim_change": bool(rng.random() < 0.5)}
    if spec["env_id"] in ("CliffWalking-v1", "ns_gym/Bridge-v0"):
        kw.pop("persistent_params", None)
    g = GpuView(VecNSEnv(make(spec["env_id"], **spec["make_kwargs"]), build_tunable_params(spec["params"]), n,
                         specialize=bool(case % 2), **kw))
    o = OracleView(OracleVecEnv(make(spec["env_id"], **spec["make_kwargs"]), build_tunable_params(spec["params"]), n, **kw))
    seeds = rng.integers(0, 2 ** 40, size=n).astype(np.uint64)
    is_grid = spec["env_id"] in GRID
    tag = f"fork case {case}: {spec} kw={kw}"
    g.reset(seeds), o.reset(seeds)
    acts = make_actions(spec["env_id"], pre + post, n)
    for k in range(pre):
        g.step(acts[k]), o.step(acts[k])
    mode, entropy = int(rng.integers(0, 2)), int(rng.integers(0, 2 ** 62))
    gf, of = GpuView(g.env.fork(theta_mode=mode, entropy=entropy)), OracleView(o.env.fork(theta_mode=mode, entropy=entropy))
    if case % 3 == 2:      # a copy OF the copy (MCTS.search deep-copies the planning env it is given): which P table, whose theta
        first = gf.env
        gf, of = GpuView(first.fork(theta_mode=0, entropy=entropy + 1)), OracleView(of.env.fork(theta_mode=0, entropy=entropy + 1))
        first.close()
        tag += " (copy of a copy)"
    compare_views(gf._out(), of._out(), is_grid, tag + " at fork")
    acro = spec["env_id"] == "Acrobot-v1"
    for k in range(pre, pre + post):
        for x, y, who in ((gf, of, "copy"), (g, o, "source")):
            a, b = x.step(acts[k]), y.step(acts[k])
            if acro:
                same = a["t"] == b["t"]
                a, b = ({key: (v[..., same] if v.ndim and v.shape[-1] == n else v[same] if v.ndim and v.shape[0] == n else v)
                         for key, v in d.items()} for d in (a, b))
            compare_views(a, b, is_grid, tag + f" {who} step {k}")
    for e in (g.env, gf.env):
        e.close()


@pytest.mark.parametrize("case", range(int(__import__("os").environ.get("NSG_SWEEP_ROLLOUT_CASES", "48"))))
def test_random_configuration_fused_rollout_matches_oracle(case):
    """The sweep through nsg_rollout (K fused steps, persistent rows and streams in registers / LDS): every
    step's outputs and the final rows against the oracle stepped K times."""
    import torch

    from ns_gym_amd.envs import make
    from ns_gym_amd.spec import build_tunable_params
    from ns_gym_amd.vec_env import VecNSEnv
    from oracle.oracle import OracleVecEnv

    rng = np.random.default_rng(40_000 + case)
    spec = random_spec(rng)
    if spec["env_id"] == "Acrobot-v1":
        spec["env_id"], spec["params"] = "Pendulum-v1", {"m": {"scheduler": SCHEDULER_SPECS["periodic3"], "update": scalar_fn_spec(rng, "randomwalk_mu_sigma", 1.0)}}
    n = int(rng.choice([1, 64, 65, 300, 513]))
    K1, K2 = int(rng.integers(1, 20)), int(rng.integers(1, 30))
    kw = {**spec["flags"], **_decode(spec), "track_returns": True}
    env = VecNSEnv(make(spec["env_id"], **spec["make_kwargs"]), build_tunable_params(spec["params"]), n, specialize=bool(case % 2), **kw)
    o = OracleView(OracleVecEnv(make(spec["env_id"], **spec["make_kwargs"]), build_tunable_params(spec["params"]), n, **kw))
    g = GpuView(env)
    seeds = rng.integers(0, 2 ** 40, size=n).astype(np.uint64)
    is_grid = spec["env_id"] in GRID
    tag = f"rollout case {case}: {spec}"
    g.reset(seeds), o.reset(seeds)
    acts = make_actions(spec["env_id"], K1 + K2, n)
    rec = ("obs", "reward", "terminated", "truncated", "env_change", "delta_change")
    k = 0
    for K in (K1, K2):   # two launches: the second starts from what the first wrote back
        out = env.rollout(torch.from_numpy(acts[k:k + K]).cuda(), record=rec)
        for j in range(K):
            b = o.step(acts[k + j])
            st = out["obs"][j].cpu().numpy()
            if is_grid:
                np.testing.assert_array_equal(st.reshape(-1), b["state"].reshape(-1), err_msg=f"{tag} step {k + j}")
            else:
                np.testing.assert_allclose(st.reshape(b["state"].shape), b["state"], rtol=1e-5, atol=1e-5, err_msg=f"{tag} step {k + j}")
            np.testing.assert_allclose(out["reward"][j].cpu().numpy(), b["reward"], rtol=1e-5, atol=1e-5, err_msg=tag)
            np.testing.assert_array_equal(out["terminated"][j].cpu().numpy().astype(np.uint8), b["terminated"], err_msg=f"{tag} step {k + j}")
            np.testing.assert_array_equal(out["truncated"][j].cpu().numpy().astype(np.uint8), b["truncated"], err_msg=f"{tag} step {k + j}")
            P = env.cfg.n_params
            np.testing.assert_array_equal(out["env_change"][j].cpu().numpy()[:P], b["env_change"], err_msg=f"{tag} step {k + j}")
        k += K
        compare_views(g._out(), o._out(), is_grid, tag + f" after {k} steps")
    c = env.counters()
    assert [c["episodes"], c["updates_applied"], c["constraint_violations"], c["env_steps"]] == [int(x) for x in o.env.a["counters"].sum(axis=1)[:4]], tag
    env.close()


@pytest.mark.parametrize("case", range(int(__import__("os").environ.get("NSG_SWEEP_POLICY_CASES", "48"))))
def test_random_configuration_policy_rollout_matches_oracle(case):
    """The sweep through nsg_rollout_policy: a random configuration, a random in-kernel action source (uniform draws, a table over the
    cells, a linear policy on the observation), discounted accounts, two launches - against the oracle's restatement of the same loop
    (orc_rollout_policy): the actions taken, the float64 accounts and the final rows."""
    import torch

    from ns_gym_amd import _abi as A
    from ns_gym_amd.envs import make
    from ns_gym_amd.policies import EpisodeAccounts, LinearPolicy, TabularPolicy, UniformRandom
    from ns_gym_amd.spec import build_tunable_params
    from ns_gym_amd.vec_env import VecNSEnv
    from oracle.oracle import OracleVecEnv

    rng = np.random.default_rng(50_000 + case)
    spec = random_spec(rng)
    if spec["env_id"] == "Acrobot-v1":      # (chaotic under closed loops: a last-ulp difference flips a decision; its open loops are swept above)
        spec["env_id"], spec["params"] = "MountainCar-v0", {"force": {"scheduler": SCHEDULER_SPECS["periodic3"], "update": scalar_fn_spec(rng, "increment", 0.001)}}
    n = int(rng.choice([1, 64, 65, 300, 513]))
    K1, K2 = int(rng.integers(1, 30)), int(rng.integers(1, 40))
    kw = {**spec["flags"], **_decode(spec)}
    env = VecNSEnv(make(spec["env_id"], **spec["make_kwargs"]), build_tunable_params(spec["params"]), n, specialize=bool(case % 2), **kw)
    o = OracleView(OracleVecEnv(make(spec["env_id"], **spec["make_kwargs"]), build_tunable_params(spec["params"]), n, **kw))
    g = GpuView(env)
    seeds = rng.integers(0, 2 ** 40, size=n).astype(np.uint64)
    is_grid = spec["env_id"] in GRID
    g.reset(seeds), o.reset(seeds)
    which = int(rng.integers(0, 2))
    if which == 0:
        pol = UniformRandom(seed=int(rng.integers(0, 2 ** 62)), index0=int(rng.integers(0, 2 ** 30)))
        okind, odata, okw = A.NSG_POL_UNIFORM, None, {"seed": pol.seed, "index0": pol.index0}
    elif is_grid:
        pol = TabularPolicy(rng.integers(0, 4, size=env.cfg.nrow * env.cfg.ncol))
        okind, odata, okw = A.NSG_POL_BY_STATE, pol.table, {}
    else:
        pol = LinearPolicy(rng.normal(size=(1 if env.action_is_float else env.n_actions, env.obs_dim + 1)).astype(np.float32))
        okind, odata, okw = A.NSG_POL_LINEAR, pol.W, {}
    gamma = float(rng.choice([1.0, 0.99, 0.9]))
    tag = f"policy case {case}: {spec} kind {okind} gamma {gamma}"
    acc = EpisodeAccounts(env, gamma=gamma, horizon=K1 + K2 + 1)
    oacc = {"ret": np.zeros(n), "length": np.zeros(n, dtype=np.int32), "alive": np.ones(n, dtype=np.uint8),
            "discount": np.array([gamma ** j for j in range(K1 + K2 + 1)], dtype=np.float64)}
    k = 0
    smooth = spec["env_id"] in ("Pendulum-v1", "MountainCarContinuous-v0")     # float64 rewards through sin / cos or the float action
    for K in (K1, K2):
        out = env.rollout_policy(pol, K, accounts=acc, step0=k, record_actions=True)
        oa, _, _ = o.env.rollout_policy(okind, K, data=odata, step0=k, accounts=oacc, **okw)
        ga = out["actions"].cpu().numpy()
        if smooth and okind == A.NSG_POL_LINEAR:
            np.testing.assert_allclose(ga, oa, rtol=0, atol=2e-5, err_msg=tag)
        else:
            np.testing.assert_array_equal(ga, oa, err_msg=tag)
        k += K
        compare_views(g._out(), o._out(), is_grid, tag + f" after {k} steps")
        np.testing.assert_array_equal(acc.length.cpu().numpy(), oacc["length"], err_msg=tag)
        np.testing.assert_array_equal(acc.alive.cpu().numpy(), oacc["alive"], err_msg=tag)
        if smooth:
            np.testing.assert_allclose(acc.ret.cpu().numpy(), oacc["ret"], rtol=1e-6, atol=1e-6, err_msg=tag)
        else:
            np.testing.assert_array_equal(acc.ret.cpu().numpy(), oacc["ret"], err_msg=tag)
    env.close()
