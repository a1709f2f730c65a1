#!/usr/bin/env python3
"""What NSG_F_LIBM_EXACT costs: the same workload through its default specialised unit and through the unit built with libm's own
sin / cos (csrc/nsg_libm.hip.h), per BASELINE config at its own size - nsg_step, a K = 64 rollout over an action table, and a K = 64
fused policy rollout - plus the registers each unit allocates.  Prints one JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ns_gym_amd import workloads as W  # noqa: E402
from ns_gym_amd.policies import EpisodeAccounts, UniformRandom  # noqa: E402
from tools.policy_probe import rand_actions, timed  # noqa: E402


def main():
    K = 64
    res = {}
    for name, n in (("c1", 1 << 20), ("c2", 1 << 16), ("c2", 1 << 20), ("pend", 1 << 18), ("acro", 1 << 18), ("mcar", 1 << 20)):
        if name not in W.WORKLOADS:
            continue
        row = {}
        for mode, kw in (("default", {}), ("libm_exact", {"libm_exact": True})):
            e = W.build(name, n, specialize=True, **kw)
            acts = rand_actions(e, K)
            pol = UniformRandom(seed=1) if not e.action_is_float else None
            acc = EpisodeAccounts(e, gamma=0.99, horizon=K + 1)
            for _ in range(3):
                e.rollout(acts, record=("reward", "terminated", "truncated"))
                if pol:
                    e.rollout_policy(pol, K, accounts=acc.restart())
            row[mode] = {"nsg_step_us": round(e.time_steps(acts[0], 200) * 1e3, 2),
                         "rollout64_us_per_step": round(timed(lambda: e.rollout(acts, record=("reward", "terminated", "truncated")), inner=4) * 1e3 / K, 3)}
            if pol:
                row[mode]["rollout_policy64_us_per_step"] = round(timed(lambda: e.rollout_policy(pol, K, accounts=acc.restart()), inner=4) * 1e3 / K, 3)
            row[mode]["kernels"] = e.kernels
            e.close()
        for k in row["default"]:
            if k.endswith("_us") or k.endswith("_per_step"):
                row.setdefault("ratio", {})[k] = round(row["libm_exact"][k] / row["default"][k], 3)
        res[f"{name}@{n}"] = row
        print(name, n, json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if not torch.cuda.is_available():
        sys.exit("needs the GPU")
    main()
