"""Config-specialised code objects (nsg_spec_build): the hiprtc unit compiles for gfx950 WITHOUT a
GPU and exports the two kernels nsg_specialize() looks up.  Runs the compiler only - no compute."""
import ctypes as C

import pytest

from tests.test_oracle_grid import grid_spec
from tests.util import TRAJ_SPECS

NAMES = ["c1_cartpole_masspole_inc", "c2_cartpole_gravity_rw", "c3_frozenlake_step50", "c4_pendulum_m_inc",
         "c4_acrobot_mass2_inc", "cartpole_random_sched", "frozenlake_randomcat"]


def _build(spec, **flags):
    from ns_gym_amd import _lib
    from ns_gym_amd.envs import make
    from ns_gym_amd.spec import build_tunable_params, compile_config

    lib = _lib.load()
    cfg = compile_config(make(spec["env_id"], **spec.get("make_kwargs", {})), build_tunable_params(spec["params"]),
                         **{**spec["flags"], **spec.get("wrapper_kwargs", {}), **flags})[0]
    code, size = C.c_void_p(), C.c_size_t()
    rc = lib.nsg_spec_build(C.byref(cfg), b"gfx950", C.byref(code), C.byref(size))
    if rc == -95 and b"libhiprtc" in lib.nsg_last_error():
        pytest.skip("libhiprtc.so not available in this environment")
    assert rc == 0, lib.nsg_last_error().decode()
    data = C.string_at(code, size.value)
    lib.nsg_spec_free(code)
    return data


@pytest.mark.parametrize("name", NAMES)
def test_specialised_unit_compiles_for_gfx950(name):
    data = _build(TRAJ_SPECS[name])
    assert data[:4] == b"\x7fELF" and len(data) > 4096
    assert b"nsg_spec_step" in data and b"nsg_spec_rollout" in data
    # a specialised unit carries only its own two kernels
    assert b"init_kernel" not in data and b"fork_kernel" not in data


def test_planning_copy_and_grid_variants_compile():
    _build(TRAJ_SPECS["c1_cartpole_masspole_inc"], is_sim_env=True)          # frozen planning copy (derived rows, t_fork)
    _build(TRAJ_SPECS["cartpole_constraint"], violation_mask=True)
    _build(grid_spec("cliff_terminal_stepwise_rewards"))
    _build(grid_spec("bridge_split_onehot"))


def test_bad_config_is_rejected_not_compiled():
    from ns_gym_amd import _abi as A, _lib

    lib = _lib.load()
    cfg = A.Config()
    cfg.abi_version = 0
    code, size = C.c_void_p(), C.c_size_t()
    assert lib.nsg_spec_build(C.byref(cfg), b"gfx950", C.byref(code), C.byref(size)) == -22
    assert not code.value and size.value == 0


def _build_group(names, expect_rc=0):
    from ns_gym_amd import _lib
    from ns_gym_amd.envs import make
    from ns_gym_amd.spec import build_tunable_params, compile_config

    lib = _lib.load()
    cfgs = []
    for name in names:
        spec = TRAJ_SPECS[name]
        cfgs.append(compile_config(make(spec["env_id"], **spec.get("make_kwargs", {})), build_tunable_params(spec["params"]),
                                   **{**spec["flags"], **spec.get("wrapper_kwargs", {})})[0])
    arr = (C.c_void_p * len(cfgs))(*[C.cast(C.pointer(c), C.c_void_p) for c in cfgs])
    code, size = C.c_void_p(), C.c_size_t()
    rc = lib.nsg_spec_build_group(arr, len(cfgs), b"gfx950", C.byref(code), C.byref(size))
    if rc == -95 and b"libhiprtc" in lib.nsg_last_error():
        pytest.skip("libhiprtc.so not available in this environment")
    assert rc == expect_rc, lib.nsg_last_error().decode()
    if rc:
        return lib.nsg_last_error().decode()
    data = C.string_at(code, size.value)
    lib.nsg_spec_free(code)
    return data


def test_group_unit_compiles_for_gfx950():
    """The heterogeneous launch's specialised unit (nsg_spec_group) for C4's pair of configs: compiles without a GPU."""
    data = _build_group(("c4_pendulum_m_inc", "c4_acrobot_mass2_inc", "c3_frozenlake_step50"))
    assert data[:4] == b"\x7fELF" and b"nsg_spec_group" in data


def _notes(data):
    """{kernel name: {metadata key: int}} from the code object's notes (llvm-readelf)."""
    import os
    import re
    import subprocess
    import tempfile

    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not available")
    with tempfile.NamedTemporaryFile(suffix=".hsaco") as f:
        f.write(data)
        f.flush()
        out = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True, check=True).stdout
    res = {}
    for blk in out.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        res[name] = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size):\s+(\d+)", blk)}
    return res


@pytest.mark.parametrize("name", NAMES)
def test_no_kernel_of_a_shipped_unit_spills_or_owns_scratch(name):
    """The spill guard covers EVERY kernel of a unit - nsg_spec_step and nsg_spec_rollout alike (round 2 checked the step
    kernel only): what nsg_spec_build hands out has no spilled vector register and no scratch segment in any kernel."""
    notes = _notes(_build(TRAJ_SPECS[name]))
    assert set(notes) == {"nsg_spec_step", "nsg_spec_rollout"}
    for kernel, k in notes.items():
        assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (name, kernel, k)


def test_group_unit_does_not_spill_and_a_spilling_one_is_refused(monkeypatch):
    """Same rule for the heterogeneous launch's unit (nsg_spec_group): C4's pair (+ a FrozenLake member) has no spill; the
    same unit forced under an 8-wavefront register bound (64 VGPRs: Acrobot's RK4 cannot fit) spills and is REFUSED, so
    nsg_step_group stays on the generic group kernel instead of launching it."""
    names = ("c4_pendulum_m_inc", "c4_acrobot_mass2_inc", "c3_frozenlake_step50")
    notes = _notes(_build_group(names))
    assert set(notes) == {"nsg_spec_group", "nsg_spec_group_rollout"}
    for kernel, k in notes.items():
        assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (kernel, k)
    monkeypatch.setenv("NSG_SPEC_FLAGS", "-DNSG_MIN_WAVES=8")
    assert "spills" in _build_group(names, expect_rc=-95)


def test_a_unit_whose_rollout_or_step_spills_is_refused(monkeypatch):
    """Random-configuration case 61 (the config whose spilling build returned wrong states on MI355X) under a forced
    6-wavefront bound: both attempts of spec_compile spill, nsg_spec_build refuses with NSG_EUNSUPPORTED."""
    import numpy as np

    from ns_gym_amd import _lib
    from ns_gym_amd.envs import make
    from ns_gym_amd.spec import build_tunable_params, compile_config
    from tests.test_gpu_random_configs import _decode, random_spec

    spec = random_spec(np.random.default_rng(10_061))
    cfg = compile_config(make(spec["env_id"], **spec["make_kwargs"]), build_tunable_params(spec["params"]),
                         **{**spec["flags"], **_decode(spec), "track_returns": True})[0]
    lib = _lib.load()
    monkeypatch.setenv("NSG_SPEC_FLAGS", "-DNSG_MIN_WAVES=6")
    code, size = C.c_void_p(), C.c_size_t()
    rc = lib.nsg_spec_build(C.byref(cfg), b"gfx950", C.byref(code), C.byref(size))
    if rc == -95 and b"libhiprtc" in lib.nsg_last_error():
        pytest.skip("libhiprtc.so not available in this environment")
    assert rc == -95 and b"spills" in lib.nsg_last_error() and not code.value


def test_no_specialised_step_kernel_spills_vector_registers():
    """CartPole's specialised step is built under a 6-wavefront register bound (80 VGPRs); a config that does not fit is
    rebuilt without it (nsg_specialize.host.h: spec_compile).  A spilling build is slow and - case 61 of
    tests/test_gpu_random_configs.py, two update fns with their own streams - was miscompiled by the runtime compiler."""
    import numpy as np

    from tests.test_gpu_random_configs import _decode, random_spec

    light = _notes(_build(TRAJ_SPECS["c1_cartpole_masspole_inc"]))["nsg_spec_step"]
    assert light["vgpr_spill_count"] == 0 and light["vgpr_count"] <= 80            # the bound holds where it can
    for case in (61, 3, 17):
        spec = random_spec(np.random.default_rng(10_000 + case))
        spec = {**spec, "wrapper_kwargs": _decode(spec)}
        for name, k in _notes(_build(spec, track_returns=True)).items():
            assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (case, name, k)


def test_without_the_runtime_compiler_the_build_is_refused_not_faked():
    """A box without libhiprtc (simulated: NSG_NO_HIPRTC=1, read once per process): nsg_spec_build answers NSG_EUNSUPPORTED and
    names the missing library - the caller (VecNSEnv) then stays on the generic kernels and warns."""
    import os
    import subprocess
    import sys

    child = ("import ctypes as C\n"
             "from ns_gym_amd import _lib\n"
             "from ns_gym_amd.envs import make\n"
             "from ns_gym_amd.spec import build_tunable_params, compile_config\n"
             "from tests.util import TRAJ_SPECS\n"
             "spec = TRAJ_SPECS['c1_cartpole_masspole_inc']\n"
             "cfg = compile_config(make(spec['env_id']), build_tunable_params(spec['params']), **spec['flags'])[0]\n"
             "lib = _lib.load(); code, size = C.c_void_p(), C.c_size_t()\n"
             "rc = lib.nsg_spec_build(C.byref(cfg), b'gfx950', C.byref(code), C.byref(size))\n"
             "assert rc == -95 and b'libhiprtc' in lib.nsg_last_error() and not code.value, (rc, lib.nsg_last_error())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", child], cwd=root, check=True, env=dict(os.environ, NSG_NO_HIPRTC="1"), timeout=300)
