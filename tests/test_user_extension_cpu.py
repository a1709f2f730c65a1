"""User-defined Scheduler / UpdateFn subclasses - the reference's extension idiom (ns_gym/base.py:50-203, tutorial.ipynb cells
38-44) - on the CPU: how they are sampled into tables (ns_gym_amd.extension), that the ORACLE stepping those tables reproduces the
fixtures the REFERENCE wrappers produced with the same classes, and that everything that cannot be fused is refused with a reason.
(The kernels themselves: tests/test_gpu_user_extension.py.)"""
import copy

import numpy as np
import pytest

import ns_gym_amd.base as base
from ns_gym_amd import _abi as A
from ns_gym_amd import extension, make
from ns_gym_amd._lib import NsgError
from ns_gym_amd.schedulers import (ContinuousScheduler, DiscreteScheduler, MemorylessScheduler, PeriodicScheduler,
                                   RandomScheduler)
from ns_gym_amd.spec import compile_config
from ns_gym_amd.update_functions import IncrementUpdate
from tests.golden import user_plugins
from tests.util import USER_SPECS, OracleView, check_trajectory, load, make_env_from_spec

P = user_plugins.plugin_classes(base)


def _bits(blob, pc):
    w = np.frombuffer(blob, dtype=np.uint32)[pc.sched_tab_off:]
    return np.array([(int(w[t >> 5]) >> (t & 31)) & 1 for t in range(pc.sched_tab_len)], dtype=np.uint8)


def _vals(blob, pc, nd=1):
    return np.frombuffer(blob, dtype=np.float64)[pc.val_tab_off:pc.val_tab_off + pc.val_tab_len * nd].reshape(pc.val_tab_len, nd)


# ------------------------------------------------------------------ the oracle on the reference's fixtures
@pytest.mark.parametrize("name", sorted(USER_SPECS))
def test_oracle_reproduces_reference_with_user_subclasses(name):
    from oracle.oracle import OracleVecEnv

    spec = USER_SPECS[name]
    env = make_env_from_spec(OracleVecEnv, spec)
    check_trajectory(OracleView(env), spec, load(f"traj_{name}.npz"), strict=True)


# ------------------------------------------------------------------ what the tables hold
def test_user_scheduler_is_sampled_into_a_bit_table_with_unknown_beyond():
    cfg, blob, _, _ = compile_config(make("CartPole-v1"), {"gravity": IncrementUpdate(P["Every"](every=5, start=7, end=60), k=0.1)})
    pc = cfg.params[0]
    assert (pc.sched_kind, pc.upd_kind, pc.sched_i0) == (A.SCHED_TABLE, A.UPD_INCREMENT, 2)
    assert pc.sched_tab_len == 2 * 500 + 1            # 2 x TimeLimit (what a late planning copy can reach), inclusive
    want = np.array([1 if (7 <= t <= 60 and t % 5 == 0) else 0 for t in range(pc.sched_tab_len)], dtype=np.uint8)
    np.testing.assert_array_equal(_bits(blob, pc), want)


def test_user_update_fn_chain_follows_reference_call_order_and_constraint_rejections():
    fn = P["Sawtooth"](P["Every"](every=5), up=0.25, down=0.6, block=5)
    cfg, blob, _, _ = compile_config(make("CartPole-v1"), {"masscart": fn})
    pc = cfg.params[0]
    assert (pc.sched_kind, pc.upd_kind) == (A.SCHED_TABLE, A.UPD_STEPWISE)
    # the same chain by hand: UpdateFn.__call__ (base.py:139-149) + the rejection rule of classic_control.py:87-92
    cur, want = 1.0, []
    for t in range(0, 1001, 5):
        new = cur + 0.25 if (t // 5) % 2 == 0 else cur - 0.6
        want.append(new)
        if not new <= 0:
            cur = new
    got = _vals(blob, pc)[:, 0]
    np.testing.assert_array_equal(got, np.array(want))
    assert (got <= 0).any(), "the chain must contain rejected proposals for this test to mean anything"
    # the user's own object was not touched by the sampling
    assert fn.prev_time == -1 and fn.prev_param is None


def test_stateful_shared_scheduler_is_called_in_dict_order():
    sched = P["EveryNthCall"](3)
    tp = {"force_mag": IncrementUpdate(sched, k=0.5), "length": P["Momentum"](sched, k=0.002)}
    cfg, blob, _, _ = compile_config(make("CartPole-v1"), tp)
    a, b = _bits(blob, cfg.params[0]), _bits(blob, cfg.params[1])
    calls = np.arange(1, 2 * a.size + 1)
    np.testing.assert_array_equal(a, (calls[0::2] % 3 == 0).astype(np.uint8))   # force_mag sees calls 1, 3, 5, ...
    np.testing.assert_array_equal(b, (calls[1::2] % 3 == 0).astype(np.uint8))   # length sees calls 2, 4, 6, ...
    assert sched.calls == 0
    assert cfg.params[1].sched_slot == 1 and cfg.params[1].fn_slot == 1          # tables carry no shared state


def test_distribution_chain_and_tutorial_oscillator():
    cfg, blob, _, _ = compile_config(make("FrozenLake-v1", is_slippery=False), {"P": P["OscillatingSlip"](PeriodicScheduler(7))},
                                     initial_prob_dist=[1, 0, 0])
    pc = cfg.params[0]
    assert pc.upd_kind == A.UPD_D_STEPWISE and pc.val_tab_len == len(range(0, 201, 7))
    v = _vals(blob, pc, 3)
    np.testing.assert_array_equal(v[0::2], np.tile([0.4, 0.3, 0.3], (len(v[0::2]), 1)))
    np.testing.assert_array_equal(v[1::2], np.tile([1.0, 0.0, 0.0], (len(v[1::2]), 1)))


def test_horizon_falls_back_to_one_time_limit_when_two_do_not_fit():
    # CartPole, TimeLimit 500, three parameters firing every step: 3 x 1001 values of 8 bytes do not fit 16 KiB, 3 x 501 do
    tp = {n: P["Momentum"](ContinuousScheduler(), k=1e-4) for n in ("masspole", "length", "force_mag")}
    cfg, blob, _, _ = compile_config(make("CartPole-v1"), tp)
    assert all(cfg.params[j].sched_tab_len == 501 and cfg.params[j].val_tab_len == 501 for j in range(3)) and len(blob) <= A.MAX_TABLE_BYTES
    tp["gravity"] = P["Momentum"](ContinuousScheduler(), k=1e-4)      # a fourth one: not even one TimeLimit fits
    tp["masscart"] = P["Momentum"](ContinuousScheduler(), k=1e-4)
    with pytest.raises(NsgError, match="bytes of tables.*at most 16384"):
        compile_config(make("CartPole-v1"), tp)


def test_env_without_time_limit_needs_a_horizon():
    fn = P["Sharpen"](PeriodicScheduler(3))
    with pytest.raises(NsgError, match="no TimeLimit.*nsg_horizon"):
        compile_config(make("CliffWalking-v1"), {"P": fn}, initial_prob_dist=[0.7, 0.1, 0.1, 0.1])
    fn.nsg_horizon = 300
    cfg, _, _, _ = compile_config(make("CliffWalking-v1"), {"P": fn}, initial_prob_dist=[0.7, 0.1, 0.1, 0.1])
    assert cfg.params[0].sched_tab_len == 301 and cfg.params[0].val_tab_len == 101


# ------------------------------------------------------------------ refusals: always with the property that blocks fusion
def test_global_rng_scheduler_is_refused_like_the_tutorials():
    with pytest.raises(NsgError, match=r"two samplings of the user-defined `_check`.*disagree.*RandomScheduler"):
        compile_config(make("FrozenLake-v1"), {"P": P["OscillatingSlip"](P["GlobalCoin"]())}, initial_prob_dist=[1, 0, 0])
    with pytest.raises(NsgError, match="two samplings"):     # ... also in front of a built-in update function
        compile_config(make("CartPole-v1"), {"gravity": IncrementUpdate(P["GlobalCoin"](), k=0.1)})


def test_nondeterministic_update_is_refused():
    class Jitter(base.UpdateFn):
        def _update(self, param, t):
            return param + np.random.normal()

    with pytest.raises(NsgError, match=r"two samplings of the user-defined `_update`.*RandomWalk"):
        compile_config(make("CartPole-v1"), {"gravity": Jitter(ContinuousScheduler())})


def test_update_fn_with_rng_attribute_is_refused():
    class OwnStream(base.UpdateFn):
        def __init__(self, scheduler, seed=0):
            super().__init__(scheduler)
            self.rng = np.random.default_rng(seed)

        def _update(self, param, t):
            return param + self.rng.normal()

    with pytest.raises(NsgError, match=r"owns an `rng` attribute.*re-seeds"):
        compile_config(make("CartPole-v1"), {"gravity": OwnStream(ContinuousScheduler())})


@pytest.mark.parametrize("sched", [RandomScheduler(0.3, seed=1), MemorylessScheduler(0.2, seed=1)])
def test_user_update_behind_a_per_env_scheduler_is_refused(sched):
    with pytest.raises(NsgError, match="draws from a per-env stream"):
        compile_config(make("CartPole-v1"), {"masspole": P["Momentum"](sched, k=0.01)})


def test_persistent_params_refusals():
    with pytest.raises(NsgError, match="persistent_params=True: θ survives a reset"):
        compile_config(make("CartPole-v1"), {"masspole": P["Momentum"](ContinuousScheduler(), k=0.01)}, persistent_params=True)
    with pytest.raises(NsgError, match="keeps state between calls.*persistent_params=True"):
        compile_config(make("CartPole-v1"), {"masspole": IncrementUpdate(P["EveryNthCall"](3), k=0.01)}, persistent_params=True)
    # a scheduler that IS a function of t is fine there
    cfg, _, _, _ = compile_config(make("CartPole-v1"), {"masspole": IncrementUpdate(P["Every"](4), k=0.01)}, persistent_params=True)
    assert cfg.params[0].sched_kind == A.SCHED_TABLE


def test_overridden_delta_is_refused():
    class OwnDelta(base.UpdateFn):
        def _update(self, param, t):
            return param + 1

        def _get_delta_change(self, param, updated_param, t):
            return abs(updated_param - param)

    with pytest.raises(NsgError, match="overrides `_get_delta_change`"):
        compile_config(make("CartPole-v1"), {"gravity": OwnDelta(ContinuousScheduler())})


def test_acrobot_cross_check_partner_must_be_user_defined_too():
    tp = {"LINK_LENGTH_1": P["Momentum"](ContinuousScheduler(), k=-0.01),
          "LINK_COM_POS_1": IncrementUpdate(ContinuousScheduler(), k=0.01)}
    with pytest.raises(NsgError, match="compares LINK_LENGTH_1 with this step's proposal for LINK_COM_POS_1"):
        compile_config(make("Acrobot-v1"), tp)
    tp["LINK_COM_POS_1"] = P["Momentum"](ContinuousScheduler(), k=0.01)
    cfg, blob, _, _ = compile_config(make("Acrobot-v1"), tp)          # both sampled jointly: fine
    # the two chains cross at some t: from there on the length proposals are rejected and the value stays
    ln, com = _vals(blob, cfg.params[0])[:, 0], _vals(blob, cfg.params[1])[:, 0]
    assert (com > ln).any()


def test_wrong_kind_and_shape_of_user_update():
    with pytest.raises(AssertionError, match="needs an UpdateDistributionFn"):
        compile_config(make("FrozenLake-v1"), {"P": P["Momentum"](ContinuousScheduler(), k=0.1)})

    class TwoOnly(base.UpdateDistributionFn):
        def _update(self, param, t):
            return [0.5, 0.5]

    with pytest.raises(NsgError, match="returned 2 probabilities.*has 3"):
        compile_config(make("FrozenLake-v1"), {"P": TwoOnly(ContinuousScheduler())})

    class NotANumber(base.UpdateFn):
        def _update(self, param, t):
            return "heavy"

    with pytest.raises(NsgError, match="must return a number"):
        compile_config(make("CartPole-v1"), {"gravity": NotANumber(ContinuousScheduler())})


def test_errors_of_the_users_own_method_say_where_they_came_from():
    class Broken(base.UpdateFn):
        def _update(self, param, t):
            return 1.0 / (t - 3)

    with pytest.raises(ZeroDivisionError, match="raised while ns_gym_amd sampled"):
        compile_config(make("CartPole-v1"), {"gravity": Broken(ContinuousScheduler())})


def test_objects_that_define_neither_method_name_it():
    class Nothing(base.Scheduler):
        pass

    with pytest.raises(NsgError, match="defines no `_check"):
        Nothing()._check(0)
    with pytest.raises(NsgError, match="neither is a built-in scheduler nor defines `_check"):
        compile_config(make("CartPole-v1"), {"gravity": IncrementUpdate(Nothing(), k=0.1)})

    class NoUpdateRule(base.UpdateFn):
        pass

    with pytest.raises(NsgError, match="neither is a built-in update function nor defines"):
        compile_config(make("CartPole-v1"), {"gravity": NoUpdateRule(ContinuousScheduler())})


# ------------------------------------------------------------------ calling the objects directly (the tutorial does)
def test_user_objects_are_callable_on_the_host_like_the_references():
    s = P["Every"](every=4, start=2, end=10)
    assert [t for t in range(14) if s(t)] == [4, 8]
    fn = P["OscillatingSlip"](P["Every"](every=3))
    p, log = [1, 0, 0], []
    for t in range(7):
        p, fired, delta = fn(p, t)
        log.append((list(p), fired, round(delta, 12)))
    assert log[0] == ([0.4, 0.3, 0.3], 1, 0.9) and log[1] == ([0.4, 0.3, 0.3], 0, 0.0) and log[3] == ([1, 0, 0], 1, 0.9)
    assert fn.prev_time == 6
    with pytest.raises(AssertionError, match="param must be a list"):
        fn((1, 0, 0), 0)
    with pytest.raises(AssertionError, match="Expected t to be an int or float"):
        P["Momentum"](ContinuousScheduler(), k=1.0)(1.0, np.int64(3))
    th, fired, delta = P["Momentum"](DiscreteScheduler({2}), k=1.0)(1.0, 2)     # a built-in deterministic scheduler: its host `_check`
    assert (th, fired, delta) == (2.0, 1, 1.0)


def test_builtin_host_checks_equal_their_compiled_tables():
    """The deterministic built-ins' `_check` exists only to sample user-defined update functions behind them: it must say what
    the kernels' closed forms / tables say (here: what the reference's fixtures say)."""
    from tests.util import MANIFEST
    from ns_gym_amd.spec import build_fn

    g = load("schedulers.npz")
    for name, sspec in MANIFEST["scheduler_specs"].items():
        s = build_fn({"scheduler": sspec, "update": ["NoUpdate", {}]}).scheduler
        if getattr(s, "_stochastic", False):
            assert not extension.has_host_check(s)
            continue
        np.testing.assert_array_equal([extension.host_fires(s, t) for t in range(g[name].shape[0])], g[name].astype(bool), err_msg=name)


def test_sampling_leaves_the_users_objects_alone_and_copies_state_at_construction():
    sched = P["EveryNthCall"](2)
    sched(0)                                    # the user played with the object before building the env: calls == 1
    fn = IncrementUpdate(sched, k=1.0)
    before = copy.deepcopy(sched.__dict__)
    cfg, blob, _, _ = compile_config(make("CartPole-v1"), {"gravity": fn})
    assert sched.__dict__ == before
    np.testing.assert_array_equal(_bits(blob, cfg.params[0])[:6], [1, 0, 1, 0, 1, 0])   # continues from calls == 1, like a deepcopy would
