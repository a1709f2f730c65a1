#!/usr/bin/env python3
"""Why does the same C1 kernel at 2^24 envs take 366-383 us per launch inside bench.py and 408 us inside tools/kbench.py on the
same box?  Same library call (nsg_time_steps); what differs is the process's history before the batch is allocated.  Each
variant runs in its own child process and reports the median of three 100-launch timings."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 1 << 24


def child(variant):
    import torch

    from ns_gym_amd import workloads as W

    dev = torch.device("cuda:0")
    keep = []
    if variant == "after_small_env":          # bench.py's history: a 2^20-env batch lived (and was closed) first
        e = W.build("c1", 1 << 20, specialize=True)
        a = W.random_actions(e)
        for _ in range(50):
            e.step(a)
        e.close()
        del e, a
    elif variant == "after_small_env_kept":   # ... and is still alive
        e = W.build("c1", 1 << 20, specialize=True)
        keep.append(e)
    elif variant == "after_1g_dummy":         # a 1 GiB tensor allocated first (shifts where the arena lands)
        keep.append(torch.empty(1 << 30, dtype=torch.uint8, device=dev))
    elif variant == "after_1g_dummy_freed":
        x = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        del x
    elif variant == "empty_cache_after_small":
        e = W.build("c1", 1 << 20, specialize=True)
        e.close()
        del e
        torch.cuda.empty_cache()
    big = W.build("c1", N, specialize=True)
    a = W.random_actions(big)
    for _ in range(30):
        big.step(a)
    reps = sorted(big.time_steps(a, 100) * 1e3 for _ in range(3))
    print("RESULT " + json.dumps({"variant": variant, "median_us": reps[1], "reps_us": reps, "arena_ptr_mod_2M": big._arena.data_ptr() % (2 << 20),
                                  "arena_ptr": hex(big._arena.data_ptr()), "actions_ptr": hex(a.data_ptr())}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for v in ("fresh", "after_small_env", "after_small_env_kept", "after_1g_dummy", "after_1g_dummy_freed", "empty_cache_after_small", "fresh"):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), v], capture_output=True, text=True, timeout=600, cwd=ROOT)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
            print(lines[-1] if lines else f"{v}: rc={p.returncode} {p.stderr[-800:]}", flush=True)
