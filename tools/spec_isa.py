#!/usr/bin/env python3
"""Static instruction mix of the config-specialised kernels (no GPU): tools/spec_isa.py <work> [--dump out.s]
Counts the instructions of nsg_spec_step / nsg_spec_rollout by class (NSG_SPEC_FLAGS is honoured)."""
import collections
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns_gym_amd import _lib, make, workloads as W  # noqa: E402
from ns_gym_amd.spec import compile_config  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disasm(name):
    w = W.WORKLOADS[name]
    kw = dict(change_notification=True, delta_change_notification=True, track_returns=True, **w["wrapper_kwargs"])
    cfg = compile_config(make(w["env_id"], **w["make_kwargs"]), w["params"](), **kw)[0]
    lib = _lib.load()
    code, size = C.c_void_p(), C.c_size_t()
    assert lib.nsg_spec_build(C.byref(cfg), b"gfx950", C.byref(code), C.byref(size)) == 0, lib.nsg_last_error().decode()
    data = C.string_at(code, size.value)
    lib.nsg_spec_free(code)
    with tempfile.NamedTemporaryFile(suffix=".hsaco") as f:
        f.write(data)
        f.flush()
        return subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout


def main():
    name = sys.argv[1]
    text = disasm(name)
    if "--dump" in sys.argv:
        open(sys.argv[sys.argv.index("--dump") + 1], "w").write(text)
    kern = None
    mix = collections.defaultdict(collections.Counter)
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
        if m:
            kern = m.group(1)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)", line)
        if kern and m:
            op = m.group(1)
            cls = ("valu_f64" if re.search(r"_f64|v_rcp_f64|v_rndne_f64|v_div", op) and op.startswith("v_") else
                   "valu_other" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith("s_load") and not op.startswith("s_waitcnt") else
                   "smem" if op.startswith("s_load") else "vmem" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else
                   "lds" if op.startswith("ds_") else "other")
            mix[kern][cls] += 1
            mix[kern]["total"] += 1
    for k, c in mix.items():
        print(name, k, dict(c))


if __name__ == "__main__":
    main()
