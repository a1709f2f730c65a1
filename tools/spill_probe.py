#!/usr/bin/env python3
"""GPU-side evidence for round 2's wrong-results event (random-configuration case 61, a CartPole config whose specialised
step kernel spilled 54-65 vector registers under the 6-wavefront bound).  Runs the config's specialised build against the
GENERIC kernels (bit-identical rows expected) in variants that separate the suspects:
  A  bound kept, spilling build (NSG_SPEC_ALLOW_SPILL=1: diagnostic switch)           - does the event reproduce?
  B  A with in-lane resets (-DNSG_CARTPOLE_INLANE=1): no LDS hand-over, no barriers   - is the hand-over involved?
  C  Acrobot forced under an 8-wavefront bound (spills; no hand-over in that env type) - does ANY spilling unit misbehave?
  D  a 40-line stand-alone hiprtc kernel that keeps 96 doubles per lane live across a divergent branch under
     __launch_bounds__(256, 8) (spills by construction), checked against its closed form - is it scratch memory itself?
  I  A with the hand-over's two s_barrier patched to s_nop in the code object         - are the barriers involved?
  N  A with the three instructions patched that put the state's spill stores under a partial exec mask - is THAT the cause?
Each variant runs in its own child process (NSG_SPEC_FLAGS / the diagnostic switch are read at compile time).  I and N patch
byte patterns of ONE specific build (the sources of the commit named in profiles/r03_case61_spill_evidence.md, built on an
MI355X box: point NSG_LIB at a library built from that commit); on any other build they report 0 patched instructions.
NSG_PROBE_ONLY=A,N selects variants.  Findings: profiles/r03_case61_spill_evidence.md."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(variant):
    import numpy as np
    import torch

    from ns_gym_amd.envs import make
    from ns_gym_amd.spec import build_tunable_params
    from ns_gym_amd.vec_env import VecNSEnv
    from tests.golden.make_golden import make_actions
    from tests.test_gpu_random_configs import _decode, random_spec
    from tests.util import TRAJ_SPECS

    if variant != "C":
        spec = random_spec(np.random.default_rng(10_061))
        kw = {**spec["flags"], **_decode(spec), "track_returns": True}
        env_id, mk, tp = spec["env_id"], spec["make_kwargs"], lambda: build_tunable_params(spec["params"])
    else:
        spec = TRAJ_SPECS["c4_acrobot_mass2_inc"]
        kw = {**spec["flags"], "track_returns": True}
        env_id, mk, tp = spec["env_id"], spec.get("make_kwargs", {}), lambda: build_tunable_params(spec["params"])
    out = {"variant": variant, "flags": os.environ.get("NSG_SPEC_FLAGS", ""), "allow_spill": os.environ.get("NSG_SPEC_ALLOW_SPILL", "")}
    res = []
    if "--build-only" in sys.argv:
        spc = VecNSEnv(make(env_id, **mk), tp(), 65, specialize=True, **kw)
        spc.close()
        return
    for n in (65, 64, 256, 5000):
        T = 45
        gen = VecNSEnv(make(env_id, **mk), tp(), n, specialize=False, **kw)
        spc = VecNSEnv(make(env_id, **mk), tp(), n, specialize=True, **kw)
        seeds = np.random.default_rng(5).integers(0, 2 ** 40, size=n).astype(np.uint64)
        gen.reset(seed=seeds); spc.reset(seed=seeds)
        assert torch.equal(gen.buf["phys"], spc.buf["phys"]) and torch.equal(gen.buf["obs"], spc.buf["obs"]), "rows differ after reset already"
        acts = make_actions(env_id, T, n)
        first_bad = None
        for k in range(T):
            a = torch.from_numpy(acts[k]).cuda()
            gen.step(a); spc.step(a)
            bad = [row for row in ("phys", "theta", "t", "obs", "reward", "terminated", "truncated", "episode", "delta_change", "env_change", "rng_upd")
                   if gen.buf[row] is not None and not torch.equal(gen.buf[row], spc.buf[row])]
            if bad and first_bad is None:
                d = (gen.buf["obs"] != spc.buf["obs"]).view(n, -1).any(dim=1)
                first_bad = {"step": k, "rows": bad, "envs_with_wrong_obs": int(d.sum()),
                             "spc_obs_env0": spc.buf["obs"].view(n, -1)[0].tolist(), "gen_obs_env0": gen.buf["obs"].view(n, -1)[0].tolist()}
                break
        res.append({"n": n, "first_mismatch": first_bad})
        gen.close(); spc.close()
    out["runs"] = res
    print("RESULT " + json.dumps(out), flush=True)


STANDALONE = r'''
extern "C" __global__ __launch_bounds__(256, 8) void spill_kernel(const double* __restrict__ in, double* __restrict__ out, int n, int rounds) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v[96];
#pragma unroll
  for (int k = 0; k < 96; k++) v[k] = in[i] + (double)k;          // 96 live doubles per lane: far beyond 64 VGPRs
  for (int r = 0; r < rounds; r++) {
    if ((i + r) & 1) {                                             // divergent region that touches every value
#pragma unroll
      for (int k = 0; k < 96; k++) v[k] = v[k] * 1.0000001 + (double)(k & 3);
    } else {
#pragma unroll
      for (int k = 0; k < 96; k++) v[k] = v[k] - (double)(k & 1);
    }
    __syncthreads();
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 96; k++) s += v[k] * (double)(k + 1);
  out[i] = s;
}
'''


def standalone():
    """Variant D: scratch memory under hipModuleLaunchKernel, nothing of this library involved."""
    import ctypes as C

    import numpy as np
    import torch

    rtc = C.CDLL("libhiprtc.so")
    hip = C.CDLL("libamdhip64.so")
    prog = C.c_void_p()
    assert rtc.hiprtcCreateProgram(C.byref(prog), STANDALONE.encode(), b"spill.hip", 0, None, None) == 0
    opts = [b"--offload-arch=gfx950", b"-O3", b"-ffp-contract=off"]
    assert rtc.hiprtcCompileProgram(prog, len(opts), (C.c_char_p * len(opts))(*opts)) == 0
    n = C.c_size_t()
    rtc.hiprtcGetCodeSize(prog, C.byref(n))
    code = C.create_string_buffer(n.value)
    rtc.hiprtcGetCode(prog, code)
    open("/tmp/spill_standalone.hsaco", "wb").write(code.raw)
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", "/tmp/spill_standalone.hsaco"], capture_output=True, text=True).stdout
    import re
    meta = {k: int(re.search(rf"\.{k}:\s*(\d+)", notes).group(1)) for k in ("vgpr_count", "vgpr_spill_count", "private_segment_fixed_size")}
    torch.zeros(1, device="cuda")
    mod, fn = C.c_void_p(), C.c_void_p()
    assert hip.hipModuleLoadData(C.byref(mod), code.raw) == 0
    assert hip.hipModuleGetFunction(C.byref(fn), mod, b"spill_kernel") == 0
    N, rounds = 100000, 7
    x = torch.rand(N, dtype=torch.float64, device="cuda")
    y = torch.zeros(N, dtype=torch.float64, device="cuda")
    a0, a1, a2, a3 = C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_int(N), C.c_int(rounds)
    args = (C.c_void_p * 4)(C.cast(C.byref(a0), C.c_void_p), C.cast(C.byref(a1), C.c_void_p), C.cast(C.byref(a2), C.c_void_p), C.cast(C.byref(a3), C.c_void_p))
    assert hip.hipModuleLaunchKernel(fn, (N + 255) // 256, 1, 1, 256, 1, 1, 0, None, args, None) == 0
    torch.cuda.synchronize()
    xs = x.cpu().numpy()
    v = xs[:, None] + np.arange(96, dtype=np.float64)[None, :]
    idx = np.arange(N)
    for r in range(rounds):
        odd = ((idx + r) & 1).astype(bool)
        v_odd = v * 1.0000001 + (np.arange(96) & 3)[None, :]
        v_even = v - (np.arange(96) & 1)[None, :]
        v = np.where(odd[:, None], v_odd, v_even)
    want = np.zeros(N)
    for k in range(96):
        want = want + v[:, k] * (k + 1)
    got = y.cpu().numpy()
    print("RESULT " + json.dumps({"variant": "D", "meta": meta, "lanes_wrong": int((got != want).sum()), "n": N}), flush=True)


def main():
    if len(sys.argv) > 1:
        return standalone() if sys.argv[1] == "D" else child(sys.argv[1])
    variants = {"A": {"NSG_SPEC_ALLOW_SPILL": "1"},
                "B": {"NSG_SPEC_ALLOW_SPILL": "1", "NSG_SPEC_FLAGS": "-DNSG_CARTPOLE_INLANE=1"},
                "C": {"NSG_SPEC_ALLOW_SPILL": "1", "NSG_SPEC_FLAGS": "-DNSG_MIN_WAVES=8"},
                "D": {},
                "I_A_with_barriers_patched_to_nops": {"NSG_SPEC_ALLOW_SPILL": "1"},
                "N_A_with_the_state_spill_stores_executed_for_every_lane": {"NSG_SPEC_ALLOW_SPILL": "1"}}
    only = os.environ.get("NSG_PROBE_ONLY", "").split(",") if os.environ.get("NSG_PROBE_ONLY") else None
    for v, extra in variants.items():
        if only and v[0] not in only:
            continue
        env = dict(os.environ, NSG_SPEC_CACHE="off", **extra)
        if v[0] in "IN":   # build once into a private disk cache, patch the code object there, run from the cache
            import glob
            import tempfile

            cache = tempfile.mkdtemp(prefix="nsg_probe_cache_")
            env["NSG_SPEC_CACHE"] = cache
            subprocess.run([sys.executable, os.path.abspath(__file__), "A", "--build-only"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
            h = bytes.fromhex
            if v[0] == "I":    # the hand-over's two s_barrier -> s_nop 0 (the first s_barrier of the kernel, after the table staging, stays)
                subs, skip_first = [(h("00008ABF"), h("000080BF"))], 1
            else:  # the decisive one: the state spill stores, which the compiler sank into the then-side of `if (i >= N)`
                # (exec = lanes beyond the batch: empty for a full chunk), are made to execute for every lane - the region's exec
                # write and its one-instruction skip become no-ops, and the else-side mask is derived from the untouched s[2:3]
                subs, skip_first = [(h("0201febe" "020088bf"), h("000080bf" "000080bf")),            # s_mov_b64 exec, s[2:3]; s_cbranch_execz 2 -> s_nop; s_nop
                                    (h("002180be" "8070707e" "7e00fe88"), h("002180be" "8070707e" "7e02fe89"))], 0   # ...; s_xor_b64 exec, exec, s[0:1] -> s_andn2_b64 exec, exec, s[2:3]
            patched = 0
            for f in glob.glob(os.path.join(cache, "*.hsaco")):
                blob = bytearray(open(f, "rb").read())
                keep = os.path.join(ROOT, "gpurun_out", "case61_spilling_unit_as_built_on_the_gpu_box.hsaco")
                if os.path.isdir(os.path.dirname(keep)):
                    open(keep, "wb").write(blob)
                sym = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", f], capture_output=True, text=True).stdout
                hdr = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", f], capture_output=True, text=True).stdout
                import re
                m = re.search(r"\.text\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", hdr)
                taddr, toff = int(m.group(1), 16), int(m.group(2), 16)
                m = re.search(r"([0-9a-f]{16})\s+(\d+)\s+FUNC\s+\w+\s+\w+\s+\d+\s+nsg_spec_step\b", sym)
                lo = int(m.group(1), 16) - taddr + toff
                hi = lo + int(m.group(2))
                for pat, rep in subs:
                    seen, k = 0, blob.find(pat, lo)
                    while 0 <= k < hi:
                        if k % 4 == 0:
                            seen += 1
                            if seen > skip_first:
                                blob[k:k + len(pat)] = rep
                                patched += 1
                        k = blob.find(pat, k + 4)
                open(f, "wb").write(blob)
            env["NSG_PROBE_NOTE"] = f"patched {patched} instruction(s) of nsg_spec_step"
        p = subprocess.run([sys.executable, os.path.abspath(__file__), v[0] if v[0] in "ABCD" else "A"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        print(v, env.get("NSG_PROBE_NOTE", ""), lines[-1] if lines else f"rc={p.returncode} {p.stderr[-1500:]}", flush=True)


if __name__ == "__main__":
    main()
