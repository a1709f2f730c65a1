"""Loader of the C-ABI shared library (ns_gym_amd/libnsgym_hip.so).  There is no fallback:
if the HIP library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

from . import _abi as A

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NSG_LIB", os.path.join(_PKG, "libnsgym_hip.so"))  # NSG_LIB: experiment builds (tools/kbench.py)
_lib = None

EXPORTS = [
    "nsg_abi_version", "nsg_last_error", "nsg_sizeof_config", "nsg_sizeof_buffers", "nsg_sizeof_layout", "nsg_pcg64_jump_table",
    "nsg_layout_query", "nsg_create", "nsg_bind", "nsg_reset", "nsg_reset_seeded", "nsg_step", "nsg_rollout", "nsg_step_group", "nsg_step_group_kind", "nsg_rollout_group",
    "nsg_fork", "nsg_seed_streams", "nsg_resident_start", "nsg_resident_publish", "nsg_resident_demo_policy",
    "nsg_table_prob_dirty", "nsg_compact_done", "nsg_theta_trace", "nsg_theta_trace_stateful", "nsg_rng_fill", "nsg_time_steps", "nsg_calib_copy_f64", "nsg_read_back", "nsg_destroy",
    "nsg_specialize", "nsg_is_specialized", "nsg_spec_origin", "nsg_spec_prebuild", "nsg_spec_prebuild_group", "nsg_spec_build", "nsg_spec_build_group", "nsg_spec_build_resident",
    "nsg_rollout_policy", "nsg_rollout_policy_kind", "nsg_spec_build_policy", "nsg_spec_prebuild_policy", "nsg_policy_bits",
    "nsg_spec_free",
]


class NsgError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess

    src_dir = os.path.join(_PKG, "csrc")
    newest = max(os.path.getmtime(os.path.join(src_dir, f)) for f in os.listdir(src_dir)
                 if f.endswith((".hip", ".h")))
    inc = os.path.join(os.path.dirname(_PKG), "include")
    newest = max([newest] + [os.path.getmtime(os.path.join(inc, f)) for f in os.listdir(inc)])
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        subprocess.check_call(["make", "-C", src_dir, "-s"])
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NsgError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C ns_gym_amd/csrc).  ns_gym_amd has no CPU fallback."
        )
    try:  # share the HIP runtime PyTorch has already loaded (same SONAME libamdhip64.so.7)
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64p = C.c_void_p, C.c_int32, C.c_int64, C.c_void_p
    lib.nsg_abi_version.restype = C.c_int
    lib.nsg_last_error.restype = C.c_char_p
    for f in ("nsg_sizeof_config", "nsg_sizeof_buffers", "nsg_sizeof_layout"):
        getattr(lib, f).restype = C.c_size_t
    lib.nsg_layout_query.argtypes = [C.POINTER(A.Config), i64, C.POINTER(A.Layout)]
    lib.nsg_create.argtypes = [C.POINTER(A.Config), vp, C.c_size_t, i64, C.POINTER(vp)]
    lib.nsg_bind.argtypes = [vp, C.POINTER(A.Buffers)]
    lib.nsg_reset.argtypes = [vp, vp, vp, vp]
    lib.nsg_reset_seeded.argtypes = [vp, C.c_uint64, vp]
    lib.nsg_pcg64_jump_table.argtypes = [vp]
    lib.nsg_pcg64_jump_table.restype = None
    lib.nsg_step.argtypes = [vp, vp, vp]
    lib.nsg_rollout.argtypes = [vp, vp, i32, C.POINTER(A.RolloutOut), vp]
    lib.nsg_step_group.argtypes = [C.POINTER(vp), i32, C.POINTER(vp), vp]
    lib.nsg_step_group_kind.argtypes = [C.POINTER(vp), i32]
    lib.nsg_rollout_group.argtypes = [C.POINTER(vp), i32, C.POINTER(vp), i32, C.POINTER(A.RolloutOut), vp]
    lib.nsg_fork.argtypes = [vp, vp, C.c_uint64, i32, vp]
    lib.nsg_resident_start.argtypes = [vp, vp, vp, i32, C.c_uint32, vp]
    lib.nsg_resident_demo_policy.argtypes = [vp, i32, vp, vp, i32, C.c_uint32, vp]
    lib.nsg_resident_publish.argtypes = [vp, vp, i32, vp]
    lib.nsg_seed_streams.argtypes = [vp, vp, i32, vp]
    lib.nsg_compact_done.argtypes = [vp, vp, vp, vp]
    lib.nsg_table_prob_dirty.argtypes = [vp, vp]
    lib.nsg_theta_trace.argtypes = [vp, i32, i32, i32, i32, vp, u64p, vp, vp, vp, vp]
    lib.nsg_theta_trace_stateful.argtypes = [vp, i32, i32, i32, i32, vp, C.POINTER(A.TraceState), vp, vp, vp, vp]
    lib.nsg_rng_fill.argtypes = [i32, vp, i32, i32, i32, vp, vp, vp]
    lib.nsg_time_steps.argtypes = [vp, vp, i32, vp, C.POINTER(C.c_float)]
    lib.nsg_calib_copy_f64.argtypes = [vp, vp, i64, vp]
    lib.nsg_read_back.argtypes = [vp, vp, i64, C.c_uint64, vp]
    lib.nsg_destroy.argtypes = [vp]
    for f in EXPORTS[6:]:
        getattr(lib, f).restype = C.c_int
    lib.nsg_spec_free.restype = None
    lib.nsg_rollout_policy.argtypes = [vp, C.POINTER(A.Policy), i32, C.POINTER(A.RolloutOut), C.POINTER(A.EpisodeAcc), vp]
    lib.nsg_rollout_policy_kind.argtypes = [vp, i32]
    lib.nsg_spec_build_policy.argtypes = [C.c_void_p, i32, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.nsg_policy_bits.restype = C.c_uint64
    lib.nsg_policy_bits.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    lib.nsg_spec_free.argtypes = [C.c_void_p]
    lib.nsg_spec_build.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.nsg_spec_build_resident.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.nsg_spec_build_group.argtypes = [C.POINTER(C.c_void_p), i32, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.nsg_specialize.argtypes = [C.c_void_p]
    lib.nsg_is_specialized.argtypes = [C.c_void_p]
    lib.nsg_spec_origin.argtypes = [C.c_void_p]
    lib.nsg_spec_prebuild.argtypes = [C.c_void_p, i64, C.c_char_p, C.c_char_p]
    lib.nsg_spec_prebuild_group.argtypes = [C.POINTER(C.c_void_p), C.POINTER(i64), i32, C.c_char_p, C.c_char_p]
    lib.nsg_spec_prebuild_policy.argtypes = [C.c_void_p, i64, i32, C.c_char_p, C.c_char_p]
    if lib.nsg_abi_version() != A.NSG_ABI_VERSION:
        raise NsgError("libnsgym_hip.so ABI version mismatch")
    if (lib.nsg_sizeof_config() != C.sizeof(A.Config) or lib.nsg_sizeof_buffers() != C.sizeof(A.Buffers)
            or lib.nsg_sizeof_layout() != C.sizeof(A.Layout)):
        raise NsgError("libnsgym_hip.so struct layout differs from ns_gym_amd/_abi.py")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().nsg_last_error().decode(errors="replace")
        raise NsgError(f"{what or 'nsg call'} failed ({rc}): {msg}")
