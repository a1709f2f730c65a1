"""`libm_exact=True` (NSG_F_LIBM_EXACT): the integrators evaluate sin / cos - and the `x ** 2` the reference hands to libm's pow -
with libm's own algorithms and roundings (ns_gym_amd/csrc/nsg_libm.hip.h), so the float64 STATE of a classic-control batch equals the oracle's - and, the oracle's state
being the reference's bit for bit (tests/test_oracle_vs_reference_live.py), the reference's - in every bit, for as long as it is
stepped.  No tolerance and no allowance anywhere in this file: open loops with autoreset, a CLOSED loop on the unstable plant that
separates the default arithmetic from libm's after ~270 steps (profiles/NOTEBOOK.md), Acrobot at C4's size over its full horizon."""
import numpy as np
import pytest

from tests.util import TRAJ_SPECS, GpuView, check_trajectory, load, make_env_from_spec

pytestmark = pytest.mark.gpu


def _vec(*a, **k):
    from ns_gym_amd.vec_env import VecNSEnv

    return VecNSEnv(*a, **k)


def _orc(*a, **k):
    from oracle.oracle import OracleVecEnv

    return OracleVecEnv(*a, **k)


def _same_state(env, orc, tag):
    F = orc.a["phys"].shape[0]
    got = env.phys.cpu().numpy()
    assert np.array_equal(got.view(np.uint64), orc.a["phys"][:got.shape[0]].view(np.uint64)), tag + ": float64 state"
    assert np.array_equal(env.state.cpu().numpy().view(np.uint32), orc.a["obs"].view(np.uint32)), tag + ": float32 observation"
    assert np.array_equal(env.t.cpu().numpy(), orc.a["t"]), tag + ": t"
    assert np.array_equal(env.theta.cpu().numpy().view(np.uint64), orc.a["theta"][:env.theta.shape[0]].view(np.uint64)), tag + ": theta"
    assert np.array_equal(env.buf["reward"].cpu().numpy().view(np.uint32), orc.a["reward"].view(np.uint32)), tag + ": reward"
    assert np.array_equal(env.buf["terminated"].cpu().numpy(), orc.a["terminated"]) and np.array_equal(env.buf["truncated"].cpu().numpy(), orc.a["truncated"]), tag


@pytest.mark.parametrize("name", [n for n in sorted(TRAJ_SPECS) if TRAJ_SPECS[n]["env_id"].split("-")[0] in
                                  ("CartPole", "Pendulum", "Acrobot", "MountainCar", "MountainCarContinuous")])
def test_committed_reference_trajectories_bit_for_bit(name):
    """Straight against the numbers the reference's wrappers produced (tests/golden/traj_*.npz), no oracle in between and no tolerance:
    float32 observation, reward and float64 theta of every step in every bit - update functions through sin / exp and RandomWalk's
    normal variates included (pendulum_all_params, the c2 / shared-RandomWalk CartPoles)."""
    spec = TRAJ_SPECS[name]
    env = make_env_from_spec(_vec, spec, libm_exact=True)
    assert env.libm_exact and env.specialized
    check_trajectory(GpuView(env), spec, load(f"traj_{name}.npz"), strict=True)
    env.close()


def test_no_autoreset_and_user_extension_fixtures_bit_for_bit():
    """The other two families of reference-generated trajectories through exact units, strictly: wrappers stepped far past `done`
    (NSG_F_NO_AUTORESET) and wrappers driving user-defined Scheduler / UpdateFn subclasses - classic-control cases."""
    from tests.util import MANIFEST, USER_SPECS, build_params

    done = 0
    for name, spec in MANIFEST["noreset_specs"].items():
        if _classic(spec):
            env = make_env_from_spec(_vec, spec, autoreset=False, libm_exact=True)
            check_trajectory(GpuView(env), spec, load(f"traj_{name}.npz"), strict=True)
            env.close(); done += 1
    for name, spec in USER_SPECS.items():
        if _classic(spec):
            env = make_env_from_spec(_vec, spec, libm_exact=True)
            check_trajectory(GpuView(env), spec, load(f"traj_{name}.npz"), strict=True)
            env.close(); done += 1
    assert done >= 6, done


def test_single_wrapper_with_libm_exact_follows_the_reference_trajectory_bit_for_bit():
    """The N = 1 adaptor (the reference's class name and return values) forwards `libm_exact`: C1's recorded reference trajectory, 1000
    steps with resets, observation bits and float64 masspole equal at every step."""
    import ns_gym_amd as nsg
    from ns_gym_amd.schedulers import ContinuousScheduler
    from ns_gym_amd.update_functions import IncrementUpdate
    from ns_gym_amd.wrappers import NSClassicControlWrapper

    spec, rec = TRAJ_SPECS["c1_cartpole_masspole_inc"], load("traj_c1_cartpole_masspole_inc.npz")
    env = NSClassicControlWrapper(nsg.make("CartPole-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)},
                                  change_notification=True, delta_change_notification=True, libm_exact=True, autoreset=True)
    assert env._vec.libm_exact
    obs, info = env.reset(seed=spec["seeds"][0])
    assert np.array_equal(np.asarray(obs["state"], dtype=np.float32).view(np.uint32), rec["state"][0, 0].view(np.uint32))
    for k in range(spec["T"]):
        obs, r, term, trunc, info = env.step(int(rec["actions"][k, 0]))
        assert np.array_equal(np.asarray(obs["state"], dtype=np.float32).view(np.uint32), rec["state"][k + 1, 0].view(np.uint32)), k
        assert float(r) == rec["reward"][k, 0] and term == bool(rec["terminated"][k, 0]) and trunc == bool(rec["truncated"][k, 0]), k
        assert env.unwrapped.masspole == rec["theta"][k + 1, 0, 0], k
    env.close()


@pytest.mark.parametrize("name,T", [("c1_cartpole_masspole_inc", 300), ("c2_cartpole_gravity_rw", 300), ("cartpole_two_params", 200),
                                    ("c4_pendulum_m_inc", 450), ("pendulum_l_and_g", 300), ("c4_acrobot_mass2_inc", 300), ("acrobot_constraints", 25),
                                    ("mountaincar", 450)])
def test_open_loop_state_equals_oracle_bit_for_bit(name, T):
    # (acrobot_constraints: short - its drifting link parameters blow RK4 up after ~40 steps, into angles of 1e60 rad and beyond, where
    # gymnasium's wrap() loop never returns and there is no reference behaviour left to equal)
    import torch

    from tests.golden.make_golden import make_actions

    from tests.golden.make_golden import SCHEDULER_SPECS

    spec = TRAJ_SPECS[name] if name in TRAJ_SPECS else {      # (`l ** 2` in the pendulum's acceleration is a scalar power too)
        "env_id": "Pendulum-v1", "params": {"g": {"scheduler": SCHEDULER_SPECS["periodic3"], "update": ["RandomWalk", {"sigma": 2.0}]},
                                            "l": {"scheduler": SCHEDULER_SPECS["continuous"], "update": ["OscillatingUpdate", {"delta": 0.05}]}},
        "flags": {"change_notification": True, "delta_change_notification": True}}
    n = 4096
    env = make_env_from_spec(_vec, spec, n=n, libm_exact=True)
    assert env.libm_exact and env.specialized
    orc = make_env_from_spec(_orc, spec, n=n)
    env.reset(seed=21); orc.reset(seed=21)
    acts = make_actions(spec["env_id"], T, n)
    for k in range(T):
        env.step(torch.from_numpy(acts[k]).cuda())
        orc.step(acts[k])
        if k % 25 == 24 or k == T - 1:
            _same_state(env, orc, f"{name} step {k}")
    # the fused rollout and the fused policy rollout are the same arithmetic
    if T >= 100:
        env.rollout(torch.from_numpy(acts[:40]).cuda(), record=("reward",))
        for k in range(40):
            orc.step(acts[k])
        _same_state(env, orc, f"{name} after a fused rollout")
    env.close()


def _classic(spec):
    return spec["env_id"].split("-")[0] in ("CartPole", "Pendulum", "Acrobot", "MountainCar", "MountainCarContinuous")


def test_the_references_planner_and_harness_loops_bit_for_bit():
    """The fixtures the reference's own MCTS._default_policy and run_episode produced (tests/golden/policy_*.npz), classic-control
    cases, through exact units with NO exception: Pendulum's discounted returns, episode totals and every action its linear agent
    chose included (the default kernels need 1e-7 / 2e-5 there); the planning-copy fixtures (get_planning_env / deepcopy) likewise."""
    from tests.policy_cases import EPISODE_CASES, MCTS_CASES, HipSide, run_episode_case, run_mcts_case
    from tests.test_gpu_planning import _PlanView
    from tests.test_oracle_planning import PLAN, run_planning

    done = 0
    for name, spec in MCTS_CASES.items():
        if _classic(spec):
            run_mcts_case(HipSide, name, strict=True, libm_exact=True); done += 1
    for name, spec in EPISODE_CASES.items():
        if _classic(spec):
            run_episode_case(HipSide, name, strict=True, libm_exact=True); done += 1
    for name, spec in PLAN.items():
        if _classic(spec):
            run_planning(lambda *a, **k: _vec(*a, libm_exact=True, **k), _PlanView, name, lambda env, mode: env.fork(theta_mode=mode, entropy=99), strict=True); done += 1
    assert done >= 8, done


@pytest.mark.parametrize("case", range(int(__import__("os").environ.get("NSG_SWEEP_EXACT_CASES", "40"))))
def test_random_classic_control_configuration_is_exact(case):
    """The random-configuration sweep of tests/test_gpu_random_configs.py (any env type, 1-3 parameters, every scheduler and update
    function of the catalogue, every wrapper flag) for the classic-control envs through exact units: where that sweep compares at
    1e-5 and skips Acrobot envs that have left the oracle's episode, this one demands every bit of every env - float64 state,
    observation, reward, theta, t, flags - and the counters, Acrobot included."""
    import torch

    from ns_gym_amd.envs import make
    from ns_gym_amd.spec import build_tunable_params
    from tests.golden.make_golden import make_actions
    from tests.test_gpu_random_configs import GRID, _decode, random_spec

    rng = np.random.default_rng(70_000 + case)
    for _ in range(50):
        spec = random_spec(rng)
        if spec["env_id"] not in GRID:
            break
    n = int(rng.choice([1, 63, 64, 65, 200, 257, 700]))
    T = 60
    kw = {**spec["flags"], **_decode(spec), "track_returns": True}
    env = _vec(make(spec["env_id"], **spec["make_kwargs"]), build_tunable_params(spec["params"]), n, libm_exact=True, **kw)
    orc = _orc(make(spec["env_id"], **spec["make_kwargs"]), build_tunable_params(spec["params"]), n, **kw)
    seeds = rng.integers(0, 2 ** 40, size=n).astype(np.uint64)
    env.reset(seed=seeds); orc.reset(seed=seeds)
    tag = f"case {case}: {spec}"
    _same_state(env, orc, tag + " reset")
    acts = make_actions(spec["env_id"], T, n)
    for k in range(T):
        env.step(torch.from_numpy(acts[k]).cuda()); orc.step(acts[k])
        if k % 10 == 9 or k == T - 1:
            _same_state(env, orc, tag + f" step {k}")
    c = env.counters()
    oc = orc.a["counters"].sum(axis=1)
    assert [c["episodes"], c["updates_applied"], c["constraint_violations"], c["env_steps"]] == [int(x) for x in oc[:4]], tag
    env.close()


def test_closed_loop_on_the_unstable_plant_stays_exact():
    """C2's config under the balancing linear policy: the case in which the default sincos and libm's part ways after ~270 steps
    (0.4 % of the envs within 4000 steps).  With libm's arithmetic: every env, every row, every account, 3000 steps."""
    from ns_gym_amd import _abi as A
    from ns_gym_amd.policies import EpisodeAccounts, LinearPolicy

    spec = TRAJ_SPECS["c2_cartpole_gravity_rw"]
    n, K, chunks = 8192, 500, 6
    env = make_env_from_spec(_vec, spec, n=n, libm_exact=True)
    orc = make_env_from_spec(_orc, spec, n=n)
    env.reset(seed=11); orc.reset(seed=11)
    Wm = np.array([[0.3, -0.8, -2.0, -1.1, 0.05], [-0.3, 0.8, 2.0, 1.1, -0.05]], dtype=np.float32)
    pol = LinearPolicy(Wm)
    acc = EpisodeAccounts(env, gamma=None)
    oacc = {"ret": np.zeros(n), "length": np.zeros(n, dtype=np.int32), "alive": np.ones(n, dtype=np.uint8), "discount": None}
    for c in range(chunks):
        out = env.rollout_policy(pol, K, accounts=acc, step0=c * K, record_actions=True)
        oacts, _, _ = orc.rollout_policy(A.NSG_POL_LINEAR, K, data=Wm, step0=c * K, accounts=oacc)
        assert np.array_equal(out["actions"].cpu().numpy(), oacts), f"chunk {c}: actions"
        _same_state(env, orc, f"closed loop after {(c + 1) * K} steps")
        assert np.array_equal(acc.ret.cpu().numpy(), oacc["ret"]) and np.array_equal(acc.length.cpu().numpy(), oacc["length"])
    assert env.counters()["episodes"] >= n * 5      # the policy does keep most poles up to the TimeLimit, again and again
    env.close()


def test_acrobot_at_c4_size_full_horizon_without_an_allowance():
    """C4's Acrobot member, 262 144 envs x 500 steps (every episode to its TimeLimit, LINK_MASS_2 grown from 1 to 6): the chaotic system for which the default arithmetic carries a stated allowance
    (tests/test_gpu_fullsize.py).  With libm's sin / cos / pow every float64 state of every env equals the oracle's, at every check."""
    import os

    import torch

    from ns_gym_amd import make, workloads as W
    from tests.golden.make_golden import make_actions

    n, T = 1 << 18, 500
    w = W.WORKLOADS["acro"]
    env = W.build("acro", n, track_returns=False, libm_exact=True, seed=None)
    orc = _orc(make(w["env_id"]), w["params"](), n, change_notification=True, delta_change_notification=True)
    env.reset(seed=5); orc.reset(seed=5)
    g = torch.Generator(device="cuda").manual_seed(9)
    acts = torch.randint(0, 3, (T, n), dtype=torch.int32, device="cuda", generator=g)
    host = acts.cpu().numpy()
    threads = min(16, os.cpu_count() or 1)
    for k in range(T):
        env.step(acts[k])
        orc.step_mt(host[k], threads)
        if k % 50 == 49:
            _same_state(env, orc, f"acrobot step {k}")
    env.close()


@pytest.mark.parametrize("name,theta_mode,in_sim_change", [("c2_cartpole_gravity_rw", 0, True), ("c4_acrobot_mass2_inc", 0, False), ("c4_pendulum_m_inc", 1, False)])
def test_planning_copies_of_an_exact_batch_are_exact(name, theta_mode, in_sim_change):
    """nsg_fork of a libm_exact batch: the copy carries the flag, gets its own exact unit and equals the oracle's copy bit for bit, like its
    source - the planner-shaped use (fork, then roll the copy forward)."""
    import torch

    from tests.golden.make_golden import make_actions

    n, pre, post = 2048, 30, 120
    spec = dict(TRAJ_SPECS[name])
    spec["flags"] = {**spec["flags"], "in_sim_change": in_sim_change}
    env = make_env_from_spec(_vec, spec, n=n, libm_exact=True)
    orc = make_env_from_spec(_orc, spec, n=n)
    env.reset(seed=3); orc.reset(seed=3)
    acts = make_actions(spec["env_id"], pre + post, n)
    for k in range(pre):
        env.step(torch.from_numpy(acts[k]).cuda()); orc.step(acts[k])
    sim, osim = env.fork(theta_mode=theta_mode, entropy=4242), orc.fork(theta_mode=theta_mode, entropy=4242)
    assert sim.libm_exact and sim.specialized and sim.is_sim_env
    _same_state(sim, osim, f"{name}: at fork")
    for k in range(pre, pre + post):
        sim.step(torch.from_numpy(acts[k]).cuda()); osim.step(acts[k])
    _same_state(sim, osim, f"{name}: copy after {post} steps")
    env.step(torch.from_numpy(acts[0]).cuda()); orc.step(acts[0])
    _same_state(env, orc, f"{name}: source after the fork")
    sim.close(); env.close()


def test_c4_as_one_group_launch_at_full_size_is_exact():
    """BASELINE C4 in its own launch shape - Pendulum 262 144 + Acrobot 262 144 in ONE nsg_step_group launch per step - with both members
    exact: every float64 state of both members equals the oracle's over 200 steps, then 64 more through one nsg_rollout_group launch."""
    import os

    import torch

    from ns_gym_amd import make, workloads as W
    from ns_gym_amd.vec_env import rollout_group, step_group

    n, T, K = 1 << 18, 200, 64
    threads = min(16, os.cpu_count() or 1)
    envs, orcs = [], []
    for name in ("pend", "acro"):
        w = W.WORKLOADS[name]
        e = W.build(name, n, track_returns=False, libm_exact=True, seed=None)
        o = _orc(make(w["env_id"]), w["params"](), n, change_notification=True, delta_change_notification=True)
        e.reset(seed=13); o.reset(seed=13)
        envs.append(e); orcs.append(o)
    g = torch.Generator(device="cuda").manual_seed(4)
    a_p = (torch.rand((T + K, n), device="cuda", generator=g) * 4 - 2).float()
    a_a = torch.randint(0, 3, (T + K, n), dtype=torch.int32, device="cuda", generator=g)
    hp, ha = a_p.cpu().numpy(), a_a.cpu().numpy()
    for k in range(T):
        step_group(envs, [a_p[k], a_a[k]])
        orcs[0].step_mt(hp[k], threads); orcs[1].step_mt(ha[k], threads)
        if k % 50 == 49:
            _same_state(envs[0], orcs[0], f"C4 group, pendulum, step {k}")
            _same_state(envs[1], orcs[1], f"C4 group, acrobot, step {k}")
    rollout_group(envs, [a_p[T:], a_a[T:]])
    for k in range(T, T + K):
        orcs[0].step_mt(hp[k], threads); orcs[1].step_mt(ha[k], threads)
    _same_state(envs[0], orcs[0], "C4 group rollout, pendulum")
    _same_state(envs[1], orcs[1], "C4 group rollout, acrobot")
    for e in envs:
        e.close()


def test_angles_beyond_the_restated_range_are_harmless():
    """Pendulum's exact unit evaluates libm's sin / cos with the argument ranges merged into one body, which every lane runs: an angle
    beyond the restated range (|x| >= 105414336 - no episode gets there, a poked state does), inf or NaN must not reach that body's
    table index.  Such lanes get the fallback's answer (a value in [-1, 1], or NaN), their neighbours stay exact."""
    import torch

    from ns_gym_amd import make, workloads as W

    n = 4096
    w = W.WORKLOADS["pend"]
    env = W.build("pend", n, libm_exact=True, track_returns=False, seed=3)
    orc = _orc(make(w["env_id"]), w["params"](), n, change_notification=True, delta_change_notification=True)
    orc.reset(seed=3)
    wild = torch.tensor([105414336.0, 1e9, -1e10, 3.3e12, 2.0 ** 45, 1.5 * 2.0 ** 45, -2.0 ** 52, 1e100, -1e300, float("inf"), float("-inf"), float("nan")],
                        dtype=torch.float64, device="cuda")
    idx = torch.arange(0, n, n // wild.numel(), device="cuda")[:wild.numel()]
    env.phys[0, idx] = wild
    a = W.random_actions(env)
    for _ in range(3):
        env.step(a); orc.step(a.cpu().numpy())
    torch.cuda.synchronize()
    obs = env.state.cpu().numpy()
    poked = np.zeros(n, dtype=bool); poked[idx.cpu().numpy()] = True
    assert np.all((np.abs(obs[poked, :2]) <= 1.0) | np.isnan(obs[poked, :2]))
    got, want = env.phys.cpu().numpy()[:, ~poked], orc.a["phys"][:2][:, ~poked]
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    env.close()


@pytest.mark.parametrize("exact", [False, True])
def test_blown_up_acrobot_states_end_every_launch(exact):
    """acrobot_constraints lets the link parameters drift until RK4 blows up (~40 steps): angles of 1e60 rad, inf, NaN.  gymnasium's
    wrap() loop never returns there, so there is no reference behaviour to equal - but every launch has to end, in both arithmetics
    (nsg_wrap_pi's passes are bounded; beyond 2^56 rad the angle is left as it came), and the envs that did not blow up go on."""
    import time

    import torch

    from tests.golden.make_golden import make_actions

    spec = TRAJ_SPECS["acrobot_constraints"]
    n, T = 4096, 300
    env = make_env_from_spec(_vec, {**spec, "flags": {**spec["flags"], "autoreset": False}}, n=n, **({"libm_exact": True} if exact else {"specialize": True}))
    env.reset(seed=21)
    acts = torch.from_numpy(make_actions(spec["env_id"], T, n)).cuda()
    t0 = time.perf_counter()
    for k in range(T):
        env.step(acts[k])
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 20.0
    ph = env.phys.cpu().numpy()
    assert (~np.isfinite(ph)).any() or np.abs(ph[:2]).max() > np.pi        # some env did blow up (or stands unwrapped beyond 2^56) ...
    assert np.isfinite(ph).all(axis=0).sum() > 0                            # ... and not all of them
    assert int(env.t.min()) == T
    env.close()


def test_resident_stepper_of_an_exact_batch_is_exact():
    """The resident closed loop (nsg_resident_start: one kernel, K steps, the policy in another kernel) of a libm_exact batch runs the exact
    arithmetic too: bit-identical to K step() calls of a second exact batch - which the tests above hold equal to the oracle."""
    import torch

    from ns_gym_amd import workloads as W
    from ns_gym_amd.vec_env import ResidentStepper
    from tests.test_gpu_resident import _policy, _same

    n, K = 1 << 16, 600
    ref = W.build("c2", n, seed=11, track_returns=False, libm_exact=True)
    env = W.build("c2", n, seed=11, track_returns=False, libm_exact=True)
    a = torch.zeros(n, dtype=torch.int32, device="cuda")
    for k in range(K):
        _policy(ref, k, a)
        ref.step(a)
    loop = ResidentStepper(env, torch.zeros(n, dtype=torch.int32, device="cuda"), wait_budget_us=50_000)
    loop.start(K)
    loop.demo_policy(K, stream=torch.cuda.Stream())
    assert loop.result() == ("finished", K)
    _same(env, ref, f"exact c2 after {K} resident steps")
    # ... and not the default arithmetic's: a default batch under the same loop has parted from it by now
    dflt = W.build("c2", n, seed=11, track_returns=False)
    for k in range(K):
        _policy(dflt, k, a)
        dflt.step(a)
    assert not torch.equal(dflt.phys, ref.phys)
    env.close(); ref.close(); dflt.close()


def test_baseline_configurations_have_prebuilt_exact_units():
    """The exact units of the BASELINE configurations ship with the library (built and inspected by `build()`, ns_gym_amd/prebuilt.py):
    bit-exact arithmetic without a runtime compiler on the box."""
    from ns_gym_amd import workloads as W
    from ns_gym_amd.vec_env import step_group, step_group_kind

    envs = {name: W.build(name, libm_exact=True) for name in ("c1", "c2", "pend", "acro")}
    for name, e in envs.items():
        assert e.libm_exact and e.kernels == "config-specialised (prebuilt)", (name, e.kernels)
    pair = [envs["pend"], envs["acro"]]
    step_group(pair, [W.random_actions(e) for e in pair])
    assert step_group_kind(pair) == "specialised (prebuilt)"
    for e in envs.values():
        e.close()


def test_exact_mode_is_refused_where_it_cannot_run():
    import torch

    from ns_gym_amd._lib import NsgError
    from ns_gym_amd import workloads as W
    from ns_gym_amd.vec_env import step_group

    p, a = W.build("pend", 4096, libm_exact=True), W.build("acro", 4096)
    with pytest.raises(NsgError, match="one launch runs one arithmetic"):     # a mixed member list
        step_group([p, a], [W.random_actions(p), W.random_actions(a)])
    p.step(W.random_actions(p))      # on its own unit it steps
    fl = W.build("c3", 4096, libm_exact=True)       # the grid envs have nothing to choose: the flag is not even set for them ...
    assert not fl.libm_exact and not fl.specialized
    with pytest.raises(NsgError, match="every member specialised"):       # ... an exact launch needs its own unit, so every member's
        step_group([p, fl], [W.random_actions(p), W.random_actions(fl)])
    fl.specialize()
    step_group([p, fl], [W.random_actions(p), W.random_actions(fl)])      # ... and then a grid member shares a launch with exact ones
    torch.cuda.synchronize()
    p.close(); a.close(); fl.close()
