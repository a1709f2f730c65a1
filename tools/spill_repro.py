#!/usr/bin/env python3
"""Evidence tool for the spilled-register miscompilation of round 2 (random-configuration case 61): builds a specialised
unit through hiprtc exactly as nsg_specialize does, but WITH the 6-wavefront register bound kept even though the kernel
spills, writes the code object and its disassembly, and prints every scratch access of nsg_spec_step together with the
exec-mask writes around it.  No GPU needed.
    tools/spill_repro.py <dumped unit (NSG_SPEC_DUMP)> <out dir> [extra -D options]"""
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.environ.get("NSG_SRC_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # NSG_SRC_ROOT: a pinned source tree
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
HEADERS = [("nsgym_hip.h", "include/nsgym_hip.h")] + [(f, f"ns_gym_amd/csrc/{f}") for f in (
    "nsg_math.hip.h", "nsg_rng.hip.h", "nsg_theta.hip.h", "nsg_envs.hip.h", "nsg_kernels.hip.h", "nsg_rollout.hip.h")]


def hiprtc_build(src: str, extra=()):
    rtc = C.CDLL("libhiprtc.so")
    prog = C.c_void_p()
    names = (C.c_char_p * len(HEADERS))(*[n.encode() for n, _ in HEADERS])
    texts = (C.c_char_p * len(HEADERS))(*[open(os.path.join(ROOT, p), "rb").read() for _, p in HEADERS])
    assert rtc.hiprtcCreateProgram(C.byref(prog), src.encode(), b"nsg_spec.hip", len(HEADERS), texts, names) == 0
    opts = ["--offload-arch=" + os.environ.get("NSG_ARCH", "gfx950"), "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function", "-DNSG_BLOCK=256", *extra]
    arr = (C.c_char_p * len(opts))(*[o.encode() for o in opts])
    rc = rtc.hiprtcCompileProgram(prog, len(opts), arr)
    if rc != 0:
        n = C.c_size_t()
        rtc.hiprtcGetProgramLogSize(prog, C.byref(n))
        log = C.create_string_buffer(n.value + 1)
        rtc.hiprtcGetProgramLog(prog, log)
        raise SystemExit(log.value.decode()[:3000])
    n = C.c_size_t()
    rtc.hiprtcGetCodeSize(prog, C.byref(n))
    code = C.create_string_buffer(n.value)
    rtc.hiprtcGetCode(prog, code)
    return code.raw


def main():
    unit, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    src = open(unit).read()
    if "#define NSG_MIN_WAVES 6" not in src:   # the dump is the rebuilt (unbounded) unit: put the bound back
        src = src.replace("#define NSG_SPEC_BUILD 1\n", "#define NSG_SPEC_BUILD 1\n#define NSG_MIN_WAVES 6\n", 1)
    code = hiprtc_build(src, sys.argv[3:])
    hs = os.path.join(out, "bounded.hsaco")
    open(hs, "wb").write(code)
    dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", hs], capture_output=True, text=True, check=True).stdout
    open(os.path.join(out, "bounded.s"), "w").write(dis)
    notes = subprocess.run([OBJDUMP.replace("objdump", "readelf"), "--notes", hs], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count")[1:]:
        g = lambda k: re.search(rf"\.{k}:\s*(\S+)", blk).group(1)  # noqa: E731
        print(g("name"), {k: g(k) for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")})


if __name__ == "__main__":
    main()
