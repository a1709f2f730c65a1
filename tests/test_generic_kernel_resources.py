"""Resource usage of the PRECOMPILED kernels, read from the library that actually ships (the device code object is taken out
of `libnsgym_hip.so`'s fat binary): no kernel spills a vector register or owns scratch memory - the class of build that was
miscompiled in round 2 (profiles/r03_case61_spill_evidence.md) must not reach the product path through the generic kernels
either - and the register budgets the launch policy relies on hold (nsgym_hip.hip: step_grid_for)."""
import os
import re
import subprocess
import tempfile

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def kernels():
    from ns_gym_amd import _lib

    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"):
        if not os.path.exists(os.path.join(LLVM, tool)):
            pytest.skip(f"{tool} not available")
    _lib.load()
    d = tempfile.mkdtemp()
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", _lib.LIB_PATH, os.path.join(d, "copy.so")])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    out = {}
    for blk in notes.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", blk)}
    return out


def test_no_precompiled_kernel_spills_vector_registers_or_owns_scratch(kernels):
    assert len(kernels) >= 50
    bad = {k: v for k, v in kernels.items() if v["vgpr_spill_count"] or v["private_segment_fixed_size"]}
    assert not bad, bad


def test_register_budgets_the_launch_policy_relies_on(kernels):
    def one(pat):
        hits = [v for k, v in kernels.items() if re.search(pat, k)]
        assert len(hits) == 1, (pat, len(hits))
        return hits[0]

    # step_kernel<ENV, FULL>: ENV 0 CartPole, 1 Pendulum, 2 Acrobot, 3 / 4 MountainCar(+Continuous), 5-7 grid envs
    assert one(r"step_kernelILi0ELb0").get("vgpr_count") <= 80          # six workgroups per CU: the 1536-workgroup launch
    assert one(r"rollout_kernelILi0ELb0").get("vgpr_count") <= 80
    for env in (0, 1, 3, 4, 5, 6, 7):                                    # full theta-engine: four wavefronts per SIMD
        assert one(rf"step_kernelILi{env}ELb1").get("vgpr_count") <= 128, env
