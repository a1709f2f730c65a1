#!/usr/bin/env python3
"""Generate tests/golden/frozenlake_intree_anchor.json from the FrozenLake layout and move geometry the reference tree itself
holds: `MAPS`, the action constants and `NSFrozenLakeV0.inc / to_s / to_m / reachable_states`
(ns_gym/benchmark_algorithms/rats-experiments/code/envs/nsfrozenlake_v0.py:9-33, 215-240, 262-277).

gymnasium - where `FrozenLake-v1` takes its maps and its `inc` from - is not in the tree, and the wrapper
fixtures under tests/golden/ were generated on top of the build's own restated base env; this legacy file is the one place in
the reference where the 4x4 / 8x8 maps, the LEFT/DOWN/RIGHT/UP encoding and the clamped moves are written down.

Runs only in the build container (needs /root/reference, read-only).  Nothing is copied: the module cannot be imported (it
needs the legacy `gym`, `six`, matplotlib and a relative package), so the literal `MAPS` / action constants are read with
`ast.literal_eval`, and the four small methods are compiled FROM THE FILE'S OWN AST and driven for every (cell, action); the
NUMBERS they return are stored.

Usage:  python tests/golden/make_frozenlake_anchor.py
"""
from __future__ import annotations

import ast
import json
import os
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("NSG_REFERENCE", "/root/reference")
SRC = os.path.join(REFERENCE, "ns_gym", "benchmark_algorithms", "rats-experiments", "code", "envs", "nsfrozenlake_v0.py")


def main():
    tree = ast.parse(open(SRC).read(), SRC)
    consts = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            if node.targets[0].id in ("MAPS", "LEFT", "DOWN", "RIGHT", "UP"):
                consts[node.targets[0].id] = ast.literal_eval(node.value)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "NSFrozenLakeV0")
    wanted = ("inc", "to_s", "to_m", "reachable_states")
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert sorted(f.name for f in fns) == sorted(wanted)
    ns = {"np": np, "State": type("State", (), {})}
    exec(compile(ast.Module(body=fns, type_ignores=[]), SRC, "exec"), ns)   # the reference's own method bodies
    out = {"source": "ns_gym/benchmark_algorithms/rats-experiments/code/envs/nsfrozenlake_v0.py",
           "actions": {k: consts[k] for k in ("LEFT", "DOWN", "RIGHT", "UP")}, "maps": {}}
    for name in ("4x4", "8x8"):
        rows = consts["MAPS"][name]
        nrow, ncol = len(rows), len(rows[0])
        env = types.SimpleNamespace(nrow=nrow, ncol=ncol, nS=nrow * ncol, is_slippery=True)
        for f in wanted:
            setattr(env, f, types.MethodType(ns[f], env))
        nxt = [[None] * 4 for _ in range(nrow * ncol)]      # [cell][action] -> cell reached by the move itself
        reach = [[None] * 4 for _ in range(nrow * ncol)]    # [cell][action] -> sorted cells reachable when slippery
        for s in range(nrow * ncol):
            r, c = env.to_m(s)
            assert env.to_s(r, c) == s
            for a in range(4):
                nr, nc = env.inc(r, c, a)
                nxt[s][a] = int(env.to_s(nr, nc))
                reach[s][a] = [int(k) for k in np.nonzero(env.reachable_states(s, a))[0]]
        out["maps"][name] = {"desc": rows, "next_cell": nxt, "slippery_reachable": reach}
    path = os.path.join(HERE, "frozenlake_intree_anchor.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
