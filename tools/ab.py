#!/usr/bin/env python3
"""Interleaved A/B timing of library builds (one subprocess per build per round, same box)."""
import json, os, subprocess, sys
builds = sys.argv[1].split(",")
work = sys.argv[2] if len(sys.argv) > 2 else "c1"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
res = {b: [] for b in builds}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for r in range(rounds):
    for b in builds:
        env = dict(os.environ, NSG_LIB=os.path.join(root, "tools", f"exp_{b}.so"))
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "kbench.py"), "--work", work, "--iters", "300"],
                             env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith(work.split(",")[0] + " "):
                res[b].append(json.loads(line.split(" ", 1)[1])["us"])
for b in builds:
    v = sorted(res[b])
    print(b, "min %.2f med %.2f" % (v[0], v[len(v) // 2]), ["%.1f" % x for x in res[b]])
