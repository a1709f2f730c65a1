"""Build-time side of the prebuilt config-specialised units (`python -m ns_gym_amd.prebuilt`, `__graft_entry__.build()`).

`nsg_specialize` normally hands the kernel sources to the runtime compiler on the GPU box.  For the configurations this package
is measured on - BASELINE.json's C1-C5 as `ns_gym_amd.workloads` builds them - the units are instead compiled when the LIBRARY is
built (no GPU needed), for the target the MI355X reports (`gfx950:sramecc+:xnack-`), by the library's own generator
(`nsg_spec_prebuild`: same embedded sources, same options, same spill rule as `nsg_specialize`), written to `ns_gym_amd/prebuilt/`
under the key `nsg_specialize` looks up BEFORE it asks hiprtc, and inspected: every kernel's registers, spills, scratch and LDS out
of the code object's metadata go to `ns_gym_amd/prebuilt/resource_usage.txt`, and a unit with a VGPR spill or scratch fails the
build.  So the code object a benchmark times is one that could be read beforehand; other configurations keep using hiprtc."""
from __future__ import annotations

import ctypes as C
import json
import os
import re
import subprocess
import sys

from . import _lib
from .spec import compile_config

ARCH = "gfx950:sramecc+:xnack-"      # hipDeviceProp_t::gcnArchName of an MI355X (part of the unit's key)
DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prebuilt")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"

# (tag, workload, envs, track_returns): the batch size matters only where the launch policy depends on it (CartPole batches of
# 49 152 - 163 840 envs reset in-lane; classic-control batches of >= 2^24 envs stream their state rows)
SINGLES = [
    ("C1 / C5 CartPole masspole Increment, 2^20 envs (any size outside the two policy ranges)", "c1", 1 << 20, True),
    ("C1 at 65 536 envs (in-lane resets)", "c1", 1 << 16, True),
    ("C1 at 2^24 envs (roofline_hbm_resident: state rows streamed)", "c1", 1 << 24, True),
    ("C2 CartPole gravity RandomWalk / Periodic(3), 65 536 envs (BASELINE's size: in-lane resets)", "c2", 1 << 16, True),
    ("C2 at 2^20 envs", "c2", 1 << 20, True),
    ("C3 FrozenLake 8x8 step at t = 50, 2^20 envs", "c3", 1 << 20, True),
    ("C4 member Pendulum, 2^18 envs", "pend", 1 << 18, True),
    ("C4 member Acrobot, 2^18 envs", "acro", 1 << 18, True),
    ("C4 member Pendulum without episode accounting", "pend", 1 << 18, False),
    ("C4 member Acrobot without episode accounting", "acro", 1 << 18, False),
]
# the fused policy rollouts (nsg_rollout_policy) of the same configurations: closed loops and planner simulations at the BASELINE sizes
# (tag, workload, envs, track_returns, action-source kinds: 1 uniform, 2 by-state, 3 linear - one small unit each)
POLICIES = [
    ("C1 fused policy rollout, 2^20 envs", "c1", 1 << 20, True, (1, 3)),
    ("C2 fused policy rollout, 65 536 envs (closed loop at BASELINE's size)", "c2", 1 << 16, True, (1, 3)),
    ("C3 fused policy rollout, 2^20 envs", "c3", 1 << 20, True, (1, 2)),
    ("C4 member Pendulum, fused policy rollout", "pend", 1 << 18, True, (1, 3)),
    ("C4 member Acrobot, fused policy rollout", "acro", 1 << 18, True, (1, 3)),
]
POLICY_KIND_NAMES = {0: "action table", 1: "uniform", 2: "by state", 3: "linear"}
GROUPS = [
    ("C4 Pendulum + Acrobot in one launch (nsg_step_group / nsg_rollout_group)", [("pend", 1 << 18), ("acro", 1 << 18)], True, False),
    ("C4 without episode accounting", [("pend", 1 << 18), ("acro", 1 << 18)], False, False),
    ("C4 in one launch, NSG_F_LIBM_EXACT (both members)", [("pend", 1 << 18), ("acro", 1 << 18)], True, True),
]
# the same configurations with NSG_F_LIBM_EXACT (`libm_exact=True`: libm's sin / cos / pow / exp / log1p, rounding for rounding), so
# that the bit-exact arithmetic of the BASELINE configurations needs no runtime compiler either
EXACT = [
    ("C1 / C5 with NSG_F_LIBM_EXACT, 2^20 envs", "c1", 1 << 20, True),
    ("C2 with NSG_F_LIBM_EXACT, 65 536 envs", "c2", 1 << 16, True),
    ("C4 member Pendulum with NSG_F_LIBM_EXACT, 2^18 envs", "pend", 1 << 18, True),
    ("C4 member Acrobot with NSG_F_LIBM_EXACT, 2^18 envs", "acro", 1 << 18, True),
]


def _config(name, track_returns, libm_exact=False):
    from . import make
    from .workloads import WORKLOADS

    w = WORKLOADS[name]
    return compile_config(make(w["env_id"], **w["make_kwargs"]), w["params"](), change_notification=True,
                          delta_change_notification=True, track_returns=track_returns, libm_exact=libm_exact, **w["wrapper_kwargs"])[0]


def _kernel_resources(path):
    notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
    out = {}
    for blk in notes.split("- .agpr_count")[1:]:
        g = lambda key: re.search(rf"\.{key}:\s*(\S+)", blk)  # noqa: E731
        out[g("name").group(1)] = {k: int(g(k).group(1)) for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                                                   "private_segment_fixed_size", "group_segment_fixed_size")}
    return out


def build_all(directory: str = DIR, arch: str = ARCH, verbose: bool = False) -> dict:
    """Compile every listed unit into `directory` (stale units of earlier builds are removed first), inspect them, write
    `resource_usage.txt` and `manifest.json` next to them.  Raises if a unit is refused or a kernel spills / owns scratch."""
    lib = _lib.load()
    os.makedirs(directory, exist_ok=True)
    for f in os.listdir(directory):
        if f.endswith(".hsaco") or ".hsaco.tmp" in f:
            os.remove(os.path.join(directory, f))
    manifest, lines = {}, []

    def newest(before):
        new = sorted(set(os.listdir(directory)) - before)
        assert len(new) == 1, new
        return new[0]

    def inspect(fname, tag):
        res = _kernel_resources(os.path.join(directory, fname))
        assert res, f"{fname}: no kernel metadata found"
        lines.append(f"## {tag}  [{fname}]")
        for kern, r in sorted(res.items()):
            lines.append(f"   {kern:<24} vgpr {r['vgpr_count']:>3}  sgpr {r['sgpr_count']:>3}  vgpr_spill {r['vgpr_spill_count']}  "
                         f"sgpr_spill {r['sgpr_spill_count']}  scratch {r['private_segment_fixed_size']} B  static LDS {r['group_segment_fixed_size']} B")
            if r["vgpr_spill_count"] or r["private_segment_fixed_size"]:
                raise RuntimeError(f"prebuilt unit {fname} ({tag}): kernel {kern} spills vector registers / owns scratch: {r}")
        manifest[fname] = {"what": tag, "arch": arch, "kernels": res}

    for tag, name, n, track, exact in [(*e, False) for e in SINGLES] + [(*e, True) for e in EXACT]:
        cfg = _config(name, track, exact)
        before = set(os.listdir(directory))
        _lib.check(lib.nsg_spec_prebuild(C.byref(cfg), n, arch.encode(), directory.encode()), f"nsg_spec_prebuild({name}, {n})")
        new = set(os.listdir(directory)) - before
        if not new:      # the same key as an earlier entry (the batch size did not change the policy)
            continue
        inspect(newest(before), tag)
        if verbose:
            print("prebuilt", tag, file=sys.stderr)
    for tag, name, n, track, kinds in POLICIES:
        cfg = _config(name, track)
        for kind in kinds:
            before = set(os.listdir(directory))
            _lib.check(lib.nsg_spec_prebuild_policy(C.byref(cfg), n, kind, arch.encode(), directory.encode()), f"nsg_spec_prebuild_policy({name}, {n}, {kind})")
            inspect(newest(before), f"{tag}: {POLICY_KIND_NAMES[kind]}")
            if verbose:
                print("prebuilt", tag, POLICY_KIND_NAMES[kind], file=sys.stderr)
    for tag, members, track, exact in GROUPS:
        cfgs = [_config(name, track, exact) for name, _ in members]
        arr = (C.c_void_p * len(cfgs))(*[C.addressof(c) for c in cfgs])
        ns = (C.c_int64 * len(cfgs))(*[n for _, n in members])
        before = set(os.listdir(directory))
        _lib.check(lib.nsg_spec_prebuild_group(arr, ns, len(cfgs), arch.encode(), directory.encode()), f"nsg_spec_prebuild_group({members})")
        inspect(newest(before), tag)
        if verbose:
            print("prebuilt", tag, file=sys.stderr)
    with open(os.path.join(directory, "resource_usage.txt"), "w") as f:
        f.write(f"# Kernels of the prebuilt config-specialised units (ns_gym_amd/prebuilt.py), target {arch}; from the code objects' metadata\n"
                "# (llvm-readelf --notes).  A unit with a VGPR spill or scratch memory is never written.\n")
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(directory, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    return manifest


if __name__ == "__main__":
    m = build_all(verbose=True)
    print(f"{len(m)} prebuilt units in {DIR}")
