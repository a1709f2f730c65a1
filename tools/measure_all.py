#!/usr/bin/env python3
"""One-call measurement of every row of DESIGN.md §4 (same GPU box for all numbers):
step API and fused rollouts (K = 64), generic and config-specialised kernels, N = 2^20, plus every BASELINE configuration at
the size BASELINE quotes it at (C2: 65 536 envs; C4: Pendulum 2^18 + Acrobot 2^18 in one group launch - the `c4_group` rows;
C1 and C3 are quoted at 2^20)."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kb = [sys.executable, os.path.join(root, "tools", "kbench.py")]
rows = {}


def run(tag, extra):
    out = subprocess.run(kb + extra, capture_output=True, text=True).stdout
    for line in out.splitlines():
        parts = line.split(" ", 1)
        if len(parts) == 2 and parts[1].startswith("{"):
            rows[f"{parts[0]}:{tag}"] = json.loads(parts[1])


for spec in (False, True):
    s = ["--spec"] if spec else []
    t = "spec" if spec else "generic"
    run(f"step:{t}", ["--work", "c1,c2,c3,pend,acro", "--n", str(1 << 20)] + s)
    run(f"rollout64:{t}", ["--work", "c1,c2,c3,pend,acro", "--n", str(1 << 20), "--rollout", "64"] + s)
    run(f"c4size:{t}", ["--work", "pend,acro", "--n", str(1 << 18)] + s)
    # every BASELINE config at ITS size: C2 is quoted at 65 536 envs (one wavefront per SIMD: the latency regime)
    run(f"own_size_step:{t}", ["--work", "c2", "--n", str(1 << 16)] + s)
    run(f"own_size_rollout64:{t}", ["--work", "c2", "--n", str(1 << 16), "--rollout", "64"] + s)
    # ... and as a CLOSED loop without a launch per step: the resident stepper with a resident policy kernel on the other side
    run(f"own_size_resident:{t}", ["--work", "c2", "--n", str(1 << 16), "--resident", "4000"] + s)
print(json.dumps(rows, indent=1))
