"""Host-side mirror of the reference interface: constructor validation and error behaviour
(AssertionError / TypeError types of ns_gym/base.py:252-261, schedulers.py:66-71,
wrappers/toy_text.py:329-334), config compilation, table building.  No GPU, no oracle."""
import numpy as np
import pytest

from ns_gym_amd import TUNABLE_PARAMS, make
from ns_gym_amd import _abi as A
from ns_gym_amd.base import Scheduler, TableBuilder
from ns_gym_amd.schedulers import (BurstScheduler, ContinuousScheduler, CustomScheduler, DiscreteScheduler,
                                   PeriodicScheduler, RandomScheduler, WindowScheduler)
from ns_gym_amd.spec import compile_config
from ns_gym_amd.update_functions import (CyclicUpdate, DistributionStepWiseUpdate, IncrementUpdate, RandomWalk,
                                         StepWiseUpdate)


def test_tunable_params_match_reference_defaults():
    # docs/source/env_pages/classic_control/*.md of the reference; SURVEY §8(a) a18
    assert TUNABLE_PARAMS["CartPoleEnv"] == {"gravity": 9.8, "masscart": 1.0, "masspole": 0.1, "force_mag": 10.0,
                                             "tau": 0.02, "length": 0.5}
    assert TUNABLE_PARAMS["PendulumEnv"] == {"m": 1.0, "l": 1.0, "dt": 0.05, "g": 10.0}
    assert list(TUNABLE_PARAMS["AcrobotEnv"]) == ["dt", "LINK_LENGTH_1", "LINK_LENGTH_2", "LINK_MASS_1", "LINK_MASS_2",
                                                  "LINK_COM_POS_1", "LINK_COM_POS_2", "LINK_MOI"]
    assert TUNABLE_PARAMS["MountainCarEnv"] == {"gravity": 0.0025, "force": 0.001}
    assert TUNABLE_PARAMS["Continuous_MountainCarEnv"] == {"power": 0.0015}
    assert list(TUNABLE_PARAMS["FrozenLakeEnv"]) == ["P"] and list(TUNABLE_PARAMS["CliffWalkingEnv"]) == ["P"]
    assert TUNABLE_PARAMS["Bridge"] == {"P": [1.0, 0.0, 0.0], "P_left": [1.0, 0.0, 0.0], "P_right": [1.0, 0.0, 0.0]}  # base.py:1161


def test_update_fn_requires_scheduler_instance():
    with pytest.raises(AssertionError):
        IncrementUpdate("not a scheduler", k=1)


def test_delta_notification_requires_change_notification():
    with pytest.raises(AssertionError):
        compile_config(make("CartPole-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), 0.1)},
                       change_notification=False, delta_change_notification=True)


def test_unknown_parameter_name_rejected():
    with pytest.raises(AssertionError):
        compile_config(make("CartPole-v1"), {"not_a_param": IncrementUpdate(ContinuousScheduler(), 0.1)})
    with pytest.raises(AssertionError):
        compile_config(make("Pendulum-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), 0.1)})
    with pytest.raises(KeyError):
        make("Ant-v5")


def test_frozenlake_distribution_asserts():
    fn = lambda: DistributionStepWiseUpdate(ContinuousScheduler(), [[0.6, 0.2, 0.2]])  # noqa: E731
    with pytest.raises(AssertionError):
        compile_config(make("FrozenLake-v1"), {"P": fn()}, initial_prob_dist=[0.5, 0.2, 0.2])
    with pytest.raises(AssertionError):
        compile_config(make("FrozenLake-v1"), {"P": fn()}, initial_prob_dist=[0.5, 0.5])
    with pytest.raises(AssertionError):
        compile_config(make("FrozenLake-v1"), {"P": IncrementUpdate(ContinuousScheduler(), 0.1)})
    with pytest.raises(ValueError):
        compile_config(make("FrozenLake-v1"), {"P": DistributionStepWiseUpdate(ContinuousScheduler(), [[0.5, 0.5]])})


def test_discrete_scheduler_ctor_asserts():
    with pytest.raises(AssertionError):
        DiscreteScheduler({1, 5}, start=3)
    with pytest.raises(AssertionError):
        DiscreteScheduler({1, 50}, end=10)
    with pytest.raises(ValueError):  # min() of empty set, like the reference (schedulers.py:66)
        DiscreteScheduler(set())


def test_config_compilation_cartpole():
    tp = {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1), "gravity": RandomWalk(PeriodicScheduler(3), seed=5)}
    cfg, blob, spec, names = compile_config(make("CartPole-v1"), tp, change_notification=True,
                                            delta_change_notification=True, persistent_params=True)
    assert names == ["masspole", "gravity"]
    assert cfg.env_type == A.ENV_CARTPOLE and cfg.n_params == 2 and cfg.max_episode_steps == 500
    assert cfg.flags == A.F_CHANGE_NOTIFICATION | A.F_DELTA_NOTIFICATION | A.F_PERSISTENT_PARAMS
    assert list(cfg.base_theta)[:6] == [9.8, 1.0, 0.1, 10.0, 0.02, 0.5]
    p0, p1 = cfg.params[0], cfg.params[1]
    assert (p0.theta_slot, p0.sched_kind, p0.upd_kind, p0.rng_child, p0.uses_rng) == (2, A.SCHED_CONTINUOUS, A.UPD_INCREMENT, 0, 0)
    assert p0.u[0] == 0.1 and p0.sched_start == 0.0 and p0.sched_end == float("inf")
    assert (p1.theta_slot, p1.sched_kind, p1.upd_kind, p1.rng_child, p1.uses_rng) == (0, A.SCHED_PERIODIC, A.UPD_RANDOMWALK, 1, 1)
    assert p1.sched_i0 == 3 and (p1.u[0], p1.u[1]) == (0.0, 1.0) and p1.has_fn_seed == 1 and p1.fn_seed == 5


def test_bit_and_value_tables():
    tb = TableBuilder()
    off, ln = tb.add_bits([0, 1, 0, 0, 1] + [0] * 40 + [1])
    words = np.frombuffer(tb.blob(), dtype=np.uint32)
    assert ln == 46 and words[off] == 0b10010 and words[off + 1] == 1 << (45 - 32)
    voff, vlen = tb.add_values([1.5, 2.5, -3.0])
    vals = np.frombuffer(tb.blob(), dtype=np.float64)
    assert vlen == 3 and list(vals[voff:voff + 3]) == [1.5, 2.5, -3.0]
    assert len(tb.blob()) % 8 == 0


def _table_bits(cfg, blob, p=0):
    pc = cfg.params[p]
    words = np.frombuffer(blob, dtype=np.uint32)[pc.sched_tab_off:]
    return [(int(words[t >> 5]) >> (t & 31)) & 1 for t in range(pc.sched_tab_len)], pc


def test_table_schedulers_compile_to_reference_fire_patterns():
    mk = lambda s: compile_config(make("CartPole-v1"), {"gravity": IncrementUpdate(s, 1.0)})  # noqa: E731
    cfg, blob, _, _ = mk(DiscreteScheduler({1, 7, 8, 50}))
    bits, pc = _table_bits(cfg, blob)
    assert pc.sched_kind == A.SCHED_TABLE and [t for t, b in enumerate(bits) if b] == [1, 7, 8, 50] and pc.sched_i0 == 0
    cfg, blob, _, _ = mk(WindowScheduler([(2, 4), (10, 10), (30, np.inf)], start=3, end=50))
    bits, pc = _table_bits(cfg, blob)
    assert [t for t, b in enumerate(bits) if b][:5] == [2, 3, 4, 10, 30] and pc.sched_i0 == 1
    assert (pc.sched_start, pc.sched_end) == (3.0, 50.0)
    cfg, blob, _, _ = mk(CustomScheduler(lambda t: t % 7 == 3))
    bits, pc = _table_bits(cfg, blob)
    # sampled over 2 x TimeLimit (a planning copy taken late in an episode runs on to t_src + 500 - 1); beyond it: "unknown" (2)
    assert pc.sched_tab_len == 1001 and [t for t, b in enumerate(bits) if b][:3] == [3, 10, 17] and pc.sched_i0 == 2
    cfg, blob, _, _ = mk(CustomScheduler(lambda t: t == 2, horizon=64))
    assert cfg.params[0].sched_tab_len == 65
    # an env without a TimeLimit has no bound on t: the callable cannot be sampled without being told how far (reference: it is
    # simply called with any t, ns_gym/schedulers.py:31-43)
    from ns_gym_amd.update_functions import DistributionNoUpdate
    with pytest.raises(ValueError, match="no TimeLimit"):
        compile_config(make("CliffWalking-v1", max_episode_steps=None), {"P": DistributionNoUpdate(CustomScheduler(lambda t: t == 2))},
                       initial_prob_dist=[1, 0, 0, 0])
    compile_config(make("CliffWalking-v1", max_episode_steps=None), {"P": DistributionNoUpdate(CustomScheduler(lambda t: t == 2, horizon=5000))},
                   initial_prob_dist=[1, 0, 0, 0])
    cfg, _, _, _ = mk(BurstScheduler(3, 2, start=1))
    assert (cfg.params[0].sched_kind, cfg.params[0].sched_i0, cfg.params[0].sched_i1) == (A.SCHED_BURST, 3, 2)
    # a listed time far beyond anything the batch can reach (the reference simply never gets there): the table is cut at the
    # reachable horizon (2 x TimeLimit) and a t beyond it is "unknown" (2: counted and raised), not a 125-KB table
    cfg, blob, _, _ = mk(DiscreteScheduler({7, 10 ** 6}))
    bits, pc = _table_bits(cfg, blob)
    assert pc.sched_tab_len == 1001 and [t for t, b in enumerate(bits) if b] == [7] and pc.sched_i0 == 2
    cfg, blob, _, _ = mk(DiscreteScheduler({7, 90000}))            # fits: kept whole, nothing fires beyond it (0)
    assert cfg.params[0].sched_tab_len == 90001 and cfg.params[0].sched_i0 == 0
    cfg, blob, _, _ = mk(WindowScheduler([(5, 8), (10 ** 6, 10 ** 6 + 5)]))
    bits, pc = _table_bits(cfg, blob)
    assert pc.sched_tab_len == 1001 and [t for t, b in enumerate(bits) if b] == [5, 6, 7, 8] and pc.sched_i0 == 2
    with pytest.raises(ValueError, match="no TimeLimit"):
        compile_config(make("CliffWalking-v1", max_episode_steps=None), {"P": DistributionNoUpdate(DiscreteScheduler({3, 10 ** 6}))},
                       initial_prob_dist=[1, 0, 0, 0])


def test_value_list_update_fns_and_unsupported_kinds():
    cfg, blob, _, _ = compile_config(make("CartPole-v1"), {"length": StepWiseUpdate(ContinuousScheduler(), [1.5, 0.7]),
                                                           "tau": CyclicUpdate(ContinuousScheduler(), [0.01, 0.03])})
    vals = np.frombuffer(blob, dtype=np.float64)
    p0, p1 = cfg.params[0], cfg.params[1]
    assert list(vals[p0.val_tab_off:p0.val_tab_off + p0.val_tab_len]) == [1.5, 0.7]
    assert list(vals[p1.val_tab_off:p1.val_tab_off + p1.val_tab_len]) == [0.01, 0.03]
    cfg, _, _, _ = compile_config(make("CartPole-v1"), {"gravity": IncrementUpdate(RandomScheduler(0.5, start=2, seed=9), 1.0)})
    pc = cfg.params[0]
    assert (pc.sched_kind, pc.sched_p0, pc.has_sched_seed, pc.sched_seed, pc.sched_start) == (A.SCHED_RANDOM, 0.5, 1, 9, 2.0)
    from ns_gym_amd.update_functions import LCBoundedDistrubutionUpdate, UniformDrift
    cfg, _, _, _ = compile_config(make("FrozenLake-v1"), {"P": LCBoundedDistrubutionUpdate(ContinuousScheduler(), L=0.1)})
    assert (cfg.params[0].upd_kind, cfg.params[0].u[0], cfg.params[0].uses_rng) == (A.UPD_D_LCBOUNDED, 0.1, 1)
    with pytest.raises(TypeError):      # like the reference: update_fn(scheduler) with a ctor that needs more arguments
        LCBoundedDistrubutionUpdate(ContinuousScheduler(), L=0.1, update_fn=UniformDrift)
    with pytest.raises(AssertionError):
        LCBoundedDistrubutionUpdate(ContinuousScheduler(), L=0.1, update_fn=IncrementUpdate)
    assert isinstance(ContinuousScheduler(), Scheduler)


def test_frozenlake_config():
    cfg, blob, spec, _ = compile_config(make("FrozenLake-v1", map_name="8x8", is_slippery=False),
                                        {"P": DistributionStepWiseUpdate(DiscreteScheduler({50}), [[0.6, 0.2, 0.2]])},
                                        initial_prob_dist=[1.0, 0.0, 0.0], modified_rewards={"H": -1, "G": 1, "F": 0, "S": 0})
    assert (cfg.nrow, cfg.ncol, cfg.max_episode_steps) == (8, 8, 100)
    desc = blob[cfg.desc_tab_off:cfg.desc_tab_off + 64]
    assert desc[:8] == b"SFFFFFFF" and desc[-1:] == b"G" and desc.count(b"H") == 10
    assert cfg.flags & A.F_MODIFIED_REWARDS and list(cfg.letter_reward) == [0.0, 0.0, -1.0, 1.0]
    assert list(cfg.initial_prob[0])[:3] == [1.0, 0.0, 0.0]


def test_product_has_no_cpu_fallback():
    """The product path fails loudly without a GPU (never routes through the oracle)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ns_gym_amd._lib import NsgError
    from ns_gym_amd.vec_env import VecNSEnv

    with pytest.raises(NsgError):
        VecNSEnv(make("CartPole-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), 0.1)}, 4)
    with pytest.raises(NsgError):
        ContinuousScheduler()(3)   # Scheduler.__call__ evaluates on the device only
    import glob
    import os

    import ns_gym_amd

    pkg = os.path.dirname(ns_gym_amd.__file__)
    files = glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True) + glob.glob(os.path.join(pkg, "csrc", "*"))
    assert len(files) > 10
    for f in files:
        if os.path.isfile(f) and not f.endswith((".so", ".s", ".txt")):
            src = open(f, errors="replace").read()
            assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src.replace(
                "never routes through the oracle", ""), f"{f} must not touch oracle/"


def test_cliffwalking_and_bridge_configs():
    from ns_gym_amd.update_functions import DistributionCyclicUpdate, DistributionDecrementUpdate

    cfg, blob, spec, names = compile_config(make("CliffWalking-v1"), {"P": DistributionDecrementUpdate(ContinuousScheduler(), 0.1)},
                                            terminal_cliff=True)
    assert (cfg.env_type, cfg.nrow, cfg.ncol, cfg.max_episode_steps) == (A.ENV_CLIFFWALKING, 4, 12, 0)
    assert list(cfg.initial_prob[0]) == [1.0, 0.0, 0.0, 0.0] and cfg.flags & A.F_TERMINAL_CLIFF
    assert list(cfg.letter_reward) == [-1.0, -1.0, -100.0, 0.0]          # S F H G defaults (toy_text.py:56-59)
    desc = blob[cfg.desc_tab_off:cfg.desc_tab_off + 48]
    assert desc[36:] == b"S" + b"H" * 10 + b"G" and desc[:36] == b"F" * 36
    with pytest.raises(ValueError):   # 3-vectors do not fit the 4-way CliffWalking distribution
        compile_config(make("CliffWalking-v1"), {"P": DistributionCyclicUpdate(ContinuousScheduler(), [[1, 0, 0]])})
    cfg, blob, spec, names = compile_config(make("ns_gym/Bridge-v0"),
                                            {"P_right": DistributionCyclicUpdate(ContinuousScheduler(), [[0, 1, 0]])},
                                            initial_prob_dist=([1, 0, 0], [0, 0, 1]))
    assert (cfg.env_type, cfg.nrow, cfg.ncol, cfg.max_episode_steps, cfg.n_params) == (A.ENV_BRIDGE, 5, 8, 100, 1)
    assert cfg.params[0].theta_slot == 2 and list(cfg.initial_prob[1])[:3] == [0.0, 0.0, 1.0]
    with pytest.raises(AssertionError):
        compile_config(make("ns_gym/Bridge-v0"), {"P": DistributionCyclicUpdate(ContinuousScheduler(), [[0, 1, 0]]),
                                                  "P_left": DistributionCyclicUpdate(ContinuousScheduler(), [[0, 1, 0]])})


def test_register_and_make_custom_ids():
    """`register` / `registry` / `make` in gymnasium's role for user-registered NS envs (reference: tests/test_registration.py:78-117)."""
    import ns_gym_amd as nsg

    calls = []
    nsg.register(id="HostApi-Custom-v0", entry_point=lambda **kw: calls.append(kw) or "built", disable_env_checker=True, order_enforce=False)
    try:
        assert "HostApi-Custom-v0" in nsg.registry
        assert nsg.make("HostApi-Custom-v0", flavour=3) == "built" and calls == [{"flavour": 3}]
    finally:
        del nsg.registry["HostApi-Custom-v0"]
    with pytest.raises(KeyError):
        nsg.make("HostApi-Custom-v0")
    with pytest.raises(TypeError):
        nsg.register(id="HostApi-Custom-v1", entry_point="module:factory")
    assert nsg.make("CartPole-v1").env_id == "CartPole-v1"   # built-in ids unaffected


def test_utils_wasserstein_distance_equals_scipy_bit_for_bit():
    """ns_gym_amd.utils.wasserstein_distance restates SciPy's CDF-distance sum (ns_gym/utils.py:55-94) term for term."""
    from scipy.stats import wasserstein_distance as scipy_w1

    from ns_gym_amd.utils import wasserstein_distance

    rng = np.random.default_rng(3)
    assert wasserstein_distance([1, 0, 0], [0, 0, 1]) == 2.0
    assert wasserstein_distance([1.0, 0.0, 0.0], [0.6, 0.2, 0.2]) == scipy_w1([0, 1, 2], [0, 1, 2], [1.0, 0.0, 0.0], [0.6, 0.2, 0.2])
    for n in list(range(1, 12)) + [16, 17, 33, 64, 129]:
        for _ in range(40):
            u, v = rng.random(n) * rng.integers(1, 5), rng.dirichlet(np.ones(n))
            idx = np.arange(n, dtype=float)
            assert wasserstein_distance(u, v) == float(scipy_w1(idx, idx, u_weights=u, v_weights=v)), n
    with pytest.raises(ValueError):
        wasserstein_distance([0.5, 0.5], [1.0, 0.0, 0.0])
    with pytest.raises(ValueError):
        wasserstein_distance([1.2, -0.2, 0.0], [1.0, 0.0, 0.0])   # the reference documents SciPy's refusal (tests/test_gridworld_wrappers.py:192-199)


def test_utils_helpers_and_reference_import_locations():
    import ns_gym_amd as nsg
    from ns_gym_amd import utils
    from ns_gym_amd.base import Reward

    assert utils.n_choose_k(5, 2) == 10 and utils.n_choose_k(4, 0) == 1
    row = [(1 / 3, 4, 0.0, False), (1 / 3, 0, 0.0, False), (1 / 3, 1, 1.0, True)]
    assert utils.state_action_update(row, [0.6, 0.2, 0.2]) == [(0.6, 4, 0.0, False), (0.2, 0, 0.0, False), (0.2, 1, 1.0, True)]
    assert utils.type_mismatch_checker({"state": 3, "relative_time": 1}, Reward(1.0, {}, {}, 1)) == (3, 1.0)
    assert utils.type_mismatch_checker() == (None, None) and utils.type_mismatch_checker(5, 2.0) == (5, 2.0)
    assert 0 <= utils.categorical_sample([0.2, 0.3, 0.5]) < 3
    assert isinstance(nsg.__version__, str) and nsg.utils is utils
    from ns_gym_amd.evaluate import type_mismatch_checker   # the harness's import location
    assert type_mismatch_checker is utils.type_mismatch_checker


def test_reference_names_resolve_from_base():
    """`from ns_gym.base import ...` names (base.py:33-47, 50, 98, 185, 1156) resolve from ns_gym_amd.base."""
    from ns_gym_amd import envs
    from ns_gym_amd.base import TUNABLE_PARAMS, Reward, Scheduler, UpdateDistributionFn, UpdateFn   # noqa: F401

    assert TUNABLE_PARAMS is envs.TUNABLE_PARAMS
    assert set(TUNABLE_PARAMS["CartPoleEnv"]) == {"gravity", "masscart", "masspole", "force_mag", "tau", "length"}
