#!/usr/bin/env python3
"""Register / LDS / scratch usage of the config-specialised kernels (nsg_spec_build runs hiprtc WITHOUT a GPU):
    tools/spec_resources.py [work ...]            (work names of tools/kbench.py; NSG_SPEC_FLAGS is honoured)
Prints, per kernel, the numbers the code object's metadata carries (llvm-readelf --notes)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns_gym_amd import _lib, make  # noqa: E402
from ns_gym_amd.spec import compile_config  # noqa: E402

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def work_table():
    from ns_gym_amd.schedulers import ContinuousScheduler, DiscreteScheduler, PeriodicScheduler
    from ns_gym_amd.update_functions import DistributionStepWiseUpdate, IncrementUpdate, RandomWalk

    return {
        "c1": ("CartPole-v1", lambda: {"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)}, {}),
        "c2": ("CartPole-v1", lambda: {"gravity": RandomWalk(PeriodicScheduler(period=3))}, {}),
        "c3": ("FrozenLake-v1", lambda: {"P": DistributionStepWiseUpdate(DiscreteScheduler({50}), [[0.6, 0.2, 0.2]])}, {"map_name": "8x8"}),
        "pend": ("Pendulum-v1", lambda: {"m": IncrementUpdate(ContinuousScheduler(), k=0.01)}, {}),
        "acro": ("Acrobot-v1", lambda: {"LINK_MASS_2": IncrementUpdate(ContinuousScheduler(), k=0.1)}, {}),
        "mcar": ("MountainCar-v0", lambda: {"force": IncrementUpdate(ContinuousScheduler(), k=1e-6)}, {}),
    }


def resources(name):
    env_id, tp, mkw = work_table()[name]
    kw = dict(change_notification=True, delta_change_notification=True, track_returns=True)
    if env_id == "FrozenLake-v1":
        kw["initial_prob_dist"] = [1.0, 0.0, 0.0]
    cfg = compile_config(make(env_id, **mkw), tp(), **kw)[0]
    lib = _lib.load()
    code, size = C.c_void_p(), C.c_size_t()
    rc = lib.nsg_spec_build(C.byref(cfg), b"gfx950", C.byref(code), C.byref(size))
    assert rc == 0, lib.nsg_last_error().decode()
    data = C.string_at(code, size.value)
    lib.nsg_spec_free(code)
    with tempfile.NamedTemporaryFile(suffix=".hsaco") as f:
        f.write(data)
        f.flush()
        notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    out = {}
    for blk in notes.split("- .agpr_count")[1:]:
        g = lambda key: re.search(rf"\.{key}:\s*(\S+)", blk)  # noqa: E731
        out[g("name").group(1)] = {k: int(g(k).group(1)) for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                                                   "private_segment_fixed_size", "group_segment_fixed_size")}
    return out


if __name__ == "__main__":
    for w in sys.argv[1:] or ["c1"]:
        for k, v in resources(w).items():
            print(w, k, v, "waves/SIMD by VGPRs:", min(8, 512 // ((v["vgpr_count"] + 7) // 8 * 8)))
