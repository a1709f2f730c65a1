"""nsg_specialize: kernels compiled for one configuration (hiprtc, the config a compile-time constant)
must be indistinguishable from the generic kernels - bit for bit, every row - and therefore satisfy
the same golden-trajectory parity.  Also covers rollouts, planning copies and the code-object cache."""
import numpy as np
import pytest

from tests.test_oracle_grid import grid_spec
from tests.util import TRAJ_SPECS, GpuView, check_trajectory, load, make_env_from_spec

pytestmark = pytest.mark.gpu

ROWS = ("phys", "cell", "theta", "table_prob", "t", "status", "episode", "rng_env", "rng_upd", "rng_sched", "sched_next", "cursor", "obs",
        "reward", "terminated", "truncated", "env_change", "delta_change", "prob", "ep_return", "last_return", "last_length",
        "done_bits")


def _vec(*a, **k):
    from ns_gym_amd.vec_env import VecNSEnv

    return VecNSEnv(*a, **k)


def _same_rows(a, b, where):
    import torch

    for r in ROWS:
        x, y = a.buf[r], b.buf[r]
        assert (x is None) == (y is None), r
        if x is not None:
            assert torch.equal(x, y), f"{where}: row {r} differs between the generic and the specialised kernels"
    assert a.counters() == b.counters(), where


SPECS = [("c1_cartpole_masspole_inc", 5000, 120), ("c2_cartpole_gravity_rw", 4099, 60), ("cartpole_two_params", 2048, 60),
         ("cartpole_constraint", 2048, 60), ("cartpole_persistent", 1000, 80), ("cartpole_random_sched", 1024, 60),
         ("c4_pendulum_m_inc", 4096, 230), ("pendulum_all_params", 1024, 60), ("acrobot_constraints", 1024, 40),
         ("mountaincar", 1024, 210), ("mountaincar_continuous", 1024, 60), ("c3_frozenlake_step50", 8192, 120),
         ("frozenlake_randomcat", 1024, 60), ("frozenlake_lcbounded", 1024, 60),
         ("cartpole_shared_randomwalk", 1024, 60), ("cartpole_shared_scheduler_and_list", 1024, 60)]


@pytest.mark.parametrize("name,n,T", SPECS)
def test_specialised_equals_generic_bitwise(name, n, T):
    import torch

    from tests.golden.make_golden import make_actions

    spec = TRAJ_SPECS[name]
    a = make_env_from_spec(_vec, spec, n=n, track_returns=True)
    b = make_env_from_spec(_vec, spec, n=n, track_returns=True, specialize=True)
    assert b.specialized and not a.specialized
    a.reset(seed=7)
    b.reset(seed=7)
    acts = torch.from_numpy(make_actions(spec["env_id"], T, n)).cuda()
    for k in range(T // 2):
        a.step(acts[k])
        b.step(acts[k])
        if k % 16 == 0:
            _same_rows(a, b, f"{name} step {k}")
    _same_rows(a, b, f"{name} after {T // 2} steps")
    # the fused rollout of the specialised unit against single generic steps
    rec = ("obs", "reward", "terminated", "truncated")
    out = b.rollout(acts[T // 2:], record=rec)
    for j, k in enumerate(range(T // 2, T)):
        obs, r, te, tr, _ = a.step(acts[k])
        assert torch.equal(out["reward"][j], r) and torch.equal(out["terminated"][j], te) and torch.equal(out["truncated"][j], tr)
        assert torch.equal(out["obs"][j].reshape(a.state.shape), a.state), f"{name} rollout step {k}"
    _same_rows(a, b, f"{name} after the rollout")
    a.close(); b.close()


@pytest.mark.parametrize("name", ["c1_cartpole_masspole_inc", "c2_cartpole_gravity_rw", "c3_frozenlake_step50", "c4_acrobot_mass2_inc"])
def test_specialised_kernels_reproduce_the_reference_trajectories(name):
    spec = TRAJ_SPECS[name]
    rec = load(f"traj_{name}.npz")
    env = make_env_from_spec(_vec, spec, specialize=True)
    assert env.specialized
    check_trajectory(GpuView(env), spec, rec)
    env.close()


def test_grid_variants_and_planning_copies():
    import torch

    from tests.golden.make_golden import make_actions

    for name in ("cliff_terminal_stepwise_rewards", "bridge_split_onehot"):
        spec = grid_spec(name)
        a = make_env_from_spec(_vec, spec, n=777)
        b = make_env_from_spec(_vec, spec, n=777, specialize=True)
        a.reset(seed=3); b.reset(seed=3)
        acts = torch.from_numpy(make_actions(spec["env_id"], 40, 777)).cuda()
        for k in range(40):
            a.step(acts[k]); b.step(acts[k])
        _same_rows(a, b, name)
        a.close(); b.close()
    # a frozen planning copy of a specialised batch is itself specialised (its own config: is_sim_env)
    spec = TRAJ_SPECS["c1_cartpole_masspole_inc"]
    a = make_env_from_spec(_vec, spec, n=1000)
    b = make_env_from_spec(_vec, spec, n=1000, specialize=True)
    a.reset(seed=5); b.reset(seed=5)
    acts = torch.from_numpy(make_actions(spec["env_id"], 30, 1000)).cuda()
    for k in range(10):
        a.step(acts[k]); b.step(acts[k])
    fa, fb = a.fork(theta_mode=0, entropy=1234), b.fork(theta_mode=0, entropy=1234)
    assert fb.specialized and not fa.specialized
    for k in range(10, 30):
        fa.step(acts[k]); fb.step(acts[k])
    _same_rows(fa, fb, "planning copy")
    for e in (a, b, fa, fb):
        e.close()


def test_code_objects_are_shared_and_cached_on_disk(tmp_path, monkeypatch):
    import time

    spec = TRAJ_SPECS["c4_pendulum_m_inc"]
    monkeypatch.setenv("NSG_SPEC_CACHE", str(tmp_path))
    # NSG_SPEC_FLAGS is part of the key: a distinct value forces a fresh compilation in this process
    monkeypatch.setenv("NSG_SPEC_FLAGS", "-DNSG_TEST_CACHE_KEY=1")
    t0 = time.time()
    a = make_env_from_spec(_vec, spec, n=256, specialize=True)
    t_first = time.time() - t0
    files = list(tmp_path.glob("nsg_*.hsaco"))
    assert len(files) == 1 and files[0].stat().st_size > 4096
    t0 = time.time()
    b = make_env_from_spec(_vec, spec, n=512, specialize=True)   # same config, other batch size: same module
    t_second = time.time() - t0
    assert b.specialized and t_second < max(0.5 * t_first, 0.2)
    assert len(list(tmp_path.glob("nsg_*.hsaco"))) == 1
    a.close(); b.close()


def test_default_cache_directory_off_switch_and_damaged_objects(tmp_path, monkeypatch):
    """Unset NSG_SPEC_CACHE = the user's cache directory; "off" = no disk cache; a cached object that does not load
    (damaged file, other GPU generation) is rebuilt in place instead of failing the handle."""
    import os
    import subprocess
    import sys

    spec = TRAJ_SPECS["c4_pendulum_m_inc"]
    cache = tmp_path / "ns_gym_amd"
    monkeypatch.delenv("NSG_SPEC_CACHE", raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path))
    monkeypatch.setenv("NSG_SPEC_FLAGS", "-DNSG_TEST_CACHE_KEY=2")
    # another process builds the unit first: this process then has to find it on disk
    child = ("from tests.util import TRAJ_SPECS, make_env_from_spec\n"
             "from ns_gym_amd.vec_env import VecNSEnv\n"
             "e = make_env_from_spec(lambda *a, **k: VecNSEnv(*a, **k), TRAJ_SPECS['c4_pendulum_m_inc'], n=256, specialize=True)\n"
             "assert e.specialized\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", child], cwd=root, check=True, env=dict(os.environ), timeout=300)
    files = list(cache.glob("nsg_*.hsaco"))
    assert len(files) == 1 and files[0].stat().st_size > 4096
    files[0].write_bytes(b"not a code object " * 64)               # damaged on disk
    a = make_env_from_spec(_vec, spec, n=256, specialize=True)      # same key: found, does not load, rebuilt
    assert a.specialized
    assert files[0].stat().st_size > 4096 and not files[0].read_bytes().startswith(b"not a code object")
    a.close()
    monkeypatch.setenv("NSG_SPEC_CACHE", "off")
    monkeypatch.setenv("NSG_SPEC_FLAGS", "-DNSG_TEST_CACHE_KEY=3")
    b = make_env_from_spec(_vec, spec, n=256, specialize=True)
    assert b.specialized and len(list(cache.glob("nsg_*.hsaco"))) == 1      # nothing new on disk
    b.close()


def test_specialised_group_launch_equals_generic_group_launch():
    """nsg_step_group over specialised members runs ONE unit compiled for the ordered tuple of their
    configs (BASELINE C4: Pendulum + Acrobot, plus a FrozenLake and a full-engine CartPole segment)."""
    import torch

    from ns_gym_amd.vec_env import step_group
    from tests.golden.make_golden import make_actions

    names = ["c4_pendulum_m_inc", "c4_acrobot_mass2_inc", "c3_frozenlake_step50", "c2_cartpole_gravity_rw"]
    ns = [5000, 3000, 4096, 2500]
    T = 60
    gen = [make_env_from_spec(_vec, TRAJ_SPECS[nm], n=n, track_returns=True) for nm, n in zip(names, ns)]
    spc = [make_env_from_spec(_vec, TRAJ_SPECS[nm], n=n, track_returns=True, specialize=True) for nm, n in zip(names, ns)]
    for e in gen + spc:
        e.reset(seed=21)
    acts = [torch.from_numpy(make_actions(TRAJ_SPECS[nm]["env_id"], T, n)).cuda() for nm, n in zip(names, ns)]
    for k in range(T):
        step_group(gen, [a[k] for a in acts])
        step_group(spc, [a[k] for a in acts])
    for a, b, nm in zip(gen, spc, names):
        _same_rows(a, b, f"group member {nm}")
    # a different membership (order matters: block ranges) gets its own unit and still agrees
    step_group(gen[::-1], [a[0] for a in acts[::-1]])
    step_group(spc[::-1], [a[0] for a in acts[::-1]])
    for a, b, nm in zip(gen, spc, names):
        _same_rows(a, b, f"reordered group member {nm}")
    for e in gen + spc:
        e.close()


def test_large_batch_without_runtime_compiler_warns_and_runs_generic():
    """No libhiprtc (simulated with NSG_NO_HIPRTC=1 in a child process): a batch that would be specialised by default falls back
    to the precompiled generic kernels WITH a SpecializationUnavailableWarning; asking for specialize=True raises."""
    import os
    import subprocess
    import sys

    child = ("import warnings, torch\n"
             "from ns_gym_amd import workloads as W\n"
             "from ns_gym_amd._lib import NsgError\n"
             "from ns_gym_amd.vec_env import SpecializationUnavailableWarning\n"
             "with warnings.catch_warnings(record=True) as w:\n"
             "    warnings.simplefilter('always')\n"
             "    e = W.build('c1', 65536)\n"
             "assert any(issubclass(x.category, SpecializationUnavailableWarning) for x in w), [str(x.message) for x in w]\n"
             "assert not e.specialized\n"
             "ref = W.build('c1', 65536, specialize=False)\n"
             "a = W.random_actions(e)\n"
             "for _ in range(30): e.step(a); ref.step(a)\n"
             "assert torch.equal(e.state, ref.state) and torch.equal(e.theta, ref.theta)\n"
             "try:\n"
             "    W.build('c1', 4096, specialize=True); raise SystemExit('specialize=True must raise without the compiler')\n"
             "except NsgError as err:\n"
             "    assert 'libhiprtc' in str(err)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", child], cwd=root, check=True, env=dict(os.environ, NSG_NO_HIPRTC="1", NSG_SPEC_CACHE="off", NSG_PREBUILT_DIR="off"), timeout=300)   # (C1's unit ships prebuilt: switched off here)


def test_units_from_a_filled_cache_load_without_the_runtime_compiler(tmp_path):
    """Deployment on a box without libhiprtc: a cache directory filled elsewhere (here: by a first child process that has the
    compiler) serves the specialised units - the second child runs with NSG_NO_HIPRTC=1 and still gets them."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = ("import torch\n"
             "from ns_gym_amd import workloads as W\n"
             "e = W.build('c2', 8192, specialize=True)\n"
             "assert e.specialized\n"
             "a = W.random_actions(e)\n"
             "for _ in range(5): e.step(a)\n"
             "torch.cuda.synchronize()\n")
    env = dict(os.environ, NSG_SPEC_CACHE=str(tmp_path), NSG_PREBUILT_DIR="off")   # (C2's unit also ships prebuilt: switched off here)
    subprocess.run([sys.executable, "-c", child], cwd=root, check=True, env=env, timeout=300)
    assert len(list(tmp_path.glob("nsg_*.hsaco"))) == 1
    subprocess.run([sys.executable, "-c", child], cwd=root, check=True, env=dict(env, NSG_NO_HIPRTC="1"), timeout=300)
