#!/usr/bin/env python3
"""Interleaved A/B timing of library builds (one subprocess per variant per round, same box).

    tools/ab.py VARIANTS [WORK] [ROUNDS] [N_ENVS] [ITERS] [ROLLOUT_K]      (ROLLOUT_K > 0: nsg_rollout, us per fused step)
    VARIANT = <build>[:spec[:<hiprtc flags>]]   build = tools/exp_<build>.so ("lib" = the in-tree library)
e.g.  tools/ab.py lib,lib:spec,lib:spec:-DNSG_BATCH_LOADS=0 c1 3
"""
import json, os, subprocess, sys
variants = sys.argv[1].split(",")
work = sys.argv[2] if len(sys.argv) > 2 else "c1"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
n_envs = sys.argv[4] if len(sys.argv) > 4 else str(1 << 20)
iters = sys.argv[5] if len(sys.argv) > 5 else "300"
rollout = sys.argv[6] if len(sys.argv) > 6 else "0"
res = {b: [] for b in variants}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for r in range(rounds):
    for v in variants:
        parts = v.split(":", 2)
        env = dict(os.environ)
        if parts[0] != "lib":
            env["NSG_LIB"] = os.path.join(root, "tools", f"exp_{parts[0]}.so")
        cmd = [sys.executable, os.path.join(root, "tools", "kbench.py"), "--work", work, "--iters", iters, "--n", n_envs]
        if rollout != "0":
            cmd += ["--rollout", rollout]
        if len(parts) > 1 and parts[1] == "spec":
            cmd.append("--spec")
            if len(parts) > 2:
                env["NSG_SPEC_FLAGS"] = parts[2]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith(work.split(",")[0] + " "):
                d = json.loads(line.split(" ", 1)[1])
                res[v].append(d["us"] if "us" in d else d["us_per_step"])
for v in variants:
    x = sorted(res[v])
    print(f"n={n_envs} {work}" + (f" rollout{rollout}" if rollout != "0" else ""), v, "min %.2f med %.2f" % (x[0], x[len(x) // 2]) if x else "no result", ["%.1f" % y for y in res[v]], flush=True)
