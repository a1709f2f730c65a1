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
    assert (lay.phys, lay.theta, lay.obs, lay.rng_env, lay.rng_upd, lay.cursor) == (4096, 2000, 4000, 4000, 8000, 0)   # phys is chunk-blocked: 4 rows x 4 chunks x 256
    assert (lay.phys_dim, lay.obs_dim, lay.n_actions, lay.action_is_float) == (4, 4, 2, 0)
    assert lay.ep_return == 1000 and lay.counters == A.CNT_COUNT * A.CNT_SHARDS and lay.done_bits == 16
    cfg, _, _, _ = compile_config(make("FrozenLake-v1", map_name="8x8"), {"P": DistributionDecrementUpdate(ContinuousScheduler(), 0.1)})
    assert lib.nsg_layout_query(C.byref(cfg), 64, C.byref(lay)) == 0
    assert (lay.cell, lay.theta, lay.table_prob, lay.obs, lay.prob, lay.phys, lay.rng_env) == (64, 192, 768, 0, 64, 0, 1024)   # chunk-blocked rows pad to 256 envs
    assert lib.nsg_layout_query(C.byref(cfg), 0, C.byref(lay)) != 0
    assert b"2^27" in lib.nsg_last_error()
    # maximum batch per handle: rows are addressed with 32-bit byte offsets (32-byte stream records)
    assert lib.nsg_layout_query(C.byref(cfg), 1 << 27, C.byref(lay)) == 0 and lay.theta == 3 << 27
    assert lib.nsg_layout_query(C.byref(cfg), (1 << 27) + 1, C.byref(lay)) == -22


def test_errors_do_not_throw_across_the_abi(lib):
    assert lib.nsg_step(None, None, None) != 0
    assert b"NULL" in lib.nsg_last_error()
    assert lib.nsg_destroy(None) == 0
