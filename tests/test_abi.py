"""The C-ABI library loads (no GPU needed) and exports every symbol include/nsgym_hip.h declares;
struct layouts agree between the header (via the library), the ctypes mirror and the oracle."""
import ctypes as C
import os
import re

import pytest

from ns_gym_amd import _abi as A
from ns_gym_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "nsgym_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nsg_[a-z_0-9]+)\s*\(", src)))


def test_exports_every_declared_symbol(lib):
    declared = _declared_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/nsgym_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == sorted(declared)


def test_struct_layouts_agree(lib):
    from oracle import oracle as O

    assert lib.nsg_abi_version() == A.NSG_ABI_VERSION
    assert lib.nsg_sizeof_config() == C.sizeof(A.Config) == O.lib().orc_sizeof_config()
    assert lib.nsg_sizeof_buffers() == C.sizeof(A.Buffers) == O.lib().orc_sizeof_buffers()
    assert lib.nsg_sizeof_layout() == C.sizeof(A.Layout)


def test_layout_query_is_pure_host_arithmetic(lib):
    from ns_gym_amd import make
    from ns_gym_amd.schedulers import ContinuousScheduler, PeriodicScheduler
    from ns_gym_amd.spec import compile_config
    from ns_gym_amd.update_functions import DistributionDecrementUpdate, IncrementUpdate, RandomWalk

    cfg, _, _, _ = compile_config(make("CartPole-v1"), {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1),
                                                        "gravity": RandomWalk(PeriodicScheduler(3))}, track_returns=True)
    lay = A.Layout()
    assert lib.nsg_layout_query(C.byref(cfg), 1000, C.byref(lay)) == 0
    assert (lay.phys, lay.theta, lay.obs, lay.rng_env, lay.rng_upd, lay.cursor) == (4096, 2000, 4000, 2002, 8000, 0)   # phys is chunk-blocked: 4 rows x 4 chunks x 256
    # classic-control envs: an episode word per env and NO per-env stream state (rng_env = descriptor + (seed, spawn key) records)
    assert (lay.episode, lay.status) == (1000, 0)
    assert (lay.phys_dim, lay.obs_dim, lay.n_actions, lay.action_is_float) == (4, 4, 2, 0)
    # CartPole pays +1 on every step: its episode return is its length - no running-return row, no last_return row
    assert (lay.ep_return, lay.last_return, lay.last_length) == (0, 0, 1000)
    assert lay.counters == A.CNT_COUNT * A.CNT_SHARDS and lay.done_bits == 16
    cfg, _, _, _ = compile_config(make("Pendulum-v1"), {"m": IncrementUpdate(ContinuousScheduler(), k=0.1)}, track_returns=True)
    assert lib.nsg_layout_query(C.byref(cfg), 1000, C.byref(lay)) == 0
    assert (lay.ep_return, lay.last_return, lay.last_length) == (1000, 1000, 1000)
    cfg, _, _, _ = compile_config(make("FrozenLake-v1", map_name="8x8"), {"P": DistributionDecrementUpdate(ContinuousScheduler(), 0.1)})
    assert lib.nsg_layout_query(C.byref(cfg), 64, C.byref(lay)) == 0
    assert (lay.cell, lay.theta, lay.table_prob, lay.obs, lay.prob, lay.phys, lay.rng_env) == (64, 192, 768, 0, 64, 0, 1024)   # chunk-blocked rows pad to 256 envs
    assert (lay.episode, lay.status) == (0, 64)   # grid envs: a status byte, PCG64 state rows (one draw per env per step)
    assert lib.nsg_layout_query(C.byref(cfg), 0, C.byref(lay)) != 0
    assert b"2^27" in lib.nsg_last_error()
    # maximum batch per handle: rows are addressed with 32-bit byte offsets (32-byte stream records)
    assert lib.nsg_layout_query(C.byref(cfg), 1 << 27, C.byref(lay)) == 0 and lay.theta == 3 << 27
    assert lib.nsg_layout_query(C.byref(cfg), (1 << 27) + 1, C.byref(lay)) == -22


def test_errors_do_not_throw_across_the_abi(lib):
    assert lib.nsg_step(None, None, None) != 0
    assert b"NULL" in lib.nsg_last_error()
    assert lib.nsg_destroy(None) == 0


def _compiled(env_id, tp, **kw):
    from ns_gym_amd import make
    from ns_gym_amd.spec import compile_config

    cfg, tables, _, _ = compile_config(make(env_id), tp, **kw)
    return cfg, bytes(tables)


def _create(lib, cfg, tables, n=64):
    h = C.c_void_p()
    rc = lib.nsg_create(C.byref(cfg), tables, len(tables), n, C.byref(h))
    return rc, lib.nsg_last_error().decode(), h


MALFORMED = [
    ("abi_version", lambda c: setattr(c, "abi_version", A.NSG_ABI_VERSION + 1), "abi_version"),
    ("env_type", lambda c: setattr(c, "env_type", 99), "env_type"),
    ("n_params", lambda c: setattr(c, "n_params", 1000), "n_params"),
    ("theta_slot", lambda c: setattr(c.params[0], "theta_slot", 77), "theta_slot"),
    ("same slot twice", lambda c: setattr(c.params[1], "theta_slot", c.params[0].theta_slot), "configured twice"),
    ("update kind", lambda c: setattr(c.params[0], "upd_kind", 9999), "update kind"),
    ("scheduler kind", lambda c: setattr(c.params[0], "sched_kind", 9999), "scheduler kind"),
    ("period 0", lambda c: setattr(c.params[1], "sched_i0", 0), "period"),
    ("uses_rng", lambda c: setattr(c.params[1], "uses_rng", 0), "uses_rng"),
    ("fn_slot", lambda c: setattr(c.params[0], "fn_slot", 5), "fn_slot"),
    ("sched_slot", lambda c: setattr(c.params[1], "sched_slot", -3), "sched_slot"),
]


@pytest.mark.parametrize("what,mutate,needle", MALFORMED, ids=[m[0] for m in MALFORMED])
def test_malformed_configs_are_refused_before_any_device_work(lib, what, mutate, needle):
    """nsg_create validates the whole config on the host first: every malformed field comes back as NSG_EINVAL with a
    message that names it - nothing throws, nothing reaches the GPU (this test runs without one)."""
    from ns_gym_amd.schedulers import ContinuousScheduler, PeriodicScheduler
    from ns_gym_amd.update_functions import IncrementUpdate, RandomWalk

    cfg, tables = _compiled("CartPole-v1", {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1),
                                            "gravity": RandomWalk(PeriodicScheduler(period=3))})
    mutate(cfg)
    rc, msg, h = _create(lib, cfg, tables)
    assert rc == -22 and not h.value, (what, rc, msg)
    assert needle in msg, (what, msg)


def test_malformed_grid_and_table_configs_are_refused(lib):
    from ns_gym_amd.schedulers import ContinuousScheduler, MemorylessScheduler, RandomScheduler
    from ns_gym_amd.update_functions import CyclicUpdate, DistributionDecrementUpdate, IncrementUpdate

    cfg, tables = _compiled("FrozenLake-v1", {"P": DistributionDecrementUpdate(ContinuousScheduler(), k=0.05)})
    cfg.nrow = 0
    rc, msg, _ = _create(lib, cfg, tables)
    assert rc == -22 and "grid map" in msg
    cfg, tables = _compiled("FrozenLake-v1", {"P": DistributionDecrementUpdate(ContinuousScheduler(), k=0.05)})
    cfg.n_params = 2
    rc, msg, _ = _create(lib, cfg, tables)
    assert rc == -22 and "exactly one" in msg
    cfg, tables = _compiled("FrozenLake-v1", {"P": DistributionDecrementUpdate(ContinuousScheduler(), k=0.05)})
    rc, msg, _ = _create(lib, cfg, tables[:8])          # the map does not fit the blob that was handed over
    assert rc == -22 and "out of range" in msg
    cfg, tables = _compiled("CartPole-v1", {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)})
    cfg.params[0].upd_kind = A.NSG_UPD_D_DECREMENT if hasattr(A, "NSG_UPD_D_DECREMENT") else 9999
    rc, msg, _ = _create(lib, cfg, tables)
    assert rc == -22                                     # a distribution update on a scalar parameter
    cfg, tables = _compiled("CartPole-v1", {"length": CyclicUpdate(ContinuousScheduler(), [0.4, 0.5, 0.6])})
    cfg.params[0].val_tab_len = 0
    rc, msg, _ = _create(lib, cfg, tables)
    assert rc == -22 and "empty cyclic" in msg
    cfg.params[0].val_tab_len = 1 << 20
    rc, msg, _ = _create(lib, cfg, tables)
    assert rc == -22 and "out of range" in msg
    cfg, tables = _compiled("CartPole-v1", {"masspole": IncrementUpdate(RandomScheduler(probability=0.3, seed=1), k=0.1)})
    cfg.params[0].sched_p0 = float("nan")
    rc, msg, _ = _create(lib, cfg, tables)
    assert rc == -22 and "NaN" in msg
    cfg, tables = _compiled("CartPole-v1", {"masspole": IncrementUpdate(MemorylessScheduler(p=0.5, seed=1), k=0.1)})
    cfg.params[0].sched_p0 = 1.5
    rc, msg, _ = _create(lib, cfg, tables)
    assert rc == -22 and "Memoryless" in msg
    # argument-level refusals of the other entry points, still without a device
    out = C.c_void_p()
    good, gt = _compiled("CartPole-v1", {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)})
    assert lib.nsg_create(C.byref(good), gt, len(gt), 0, C.byref(out)) == -22
    assert lib.nsg_create(C.byref(good), None, 16, 64, C.byref(out)) == -22
    assert lib.nsg_create(C.byref(good), gt, len(gt), 64, None) == -22
    assert lib.nsg_rollout(None, None, 4, None, None) == -22
    assert lib.nsg_fork(None, None, 0, 0, None) == -22
    assert lib.nsg_step_group(None, 0, None, None) == -22
    assert lib.nsg_specialize(None) == -22 and lib.nsg_is_specialized(None) == 0
    assert lib.nsg_seed_streams(None, None, 0, None) == -22
    assert lib.nsg_compact_done(None, None, None, None) == -22
