#!/usr/bin/env python3
"""Kernel timing harness (GPU box): average device time per nsg_step launch for the
BASELINE.json configurations, measured with hipEvents over back-to-back launches."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ns_gym_amd.vec_env import rollout_group, step_group  # noqa: E402

from ns_gym_amd import workloads as W  # noqa: E402

# name -> algorithmic bytes per env-step, fp64-internal variants of SURVEY §8(d) (ns_gym_amd/workloads.py)
WORK = {k: (v["env_id"], v["params"], v["make_kwargs"], v["bytes_per_env_step"]) for k, v in W.WORKLOADS.items()}


EXACT = False


def mk(name, n, track=True, spec=False):
    return W.build(name, n, track_returns=track, specialize=spec, **({"libm_exact": True} if EXACT else {}))


def actions(e, n):
    if e.action_is_float:
        return (torch.rand(n, device="cuda") * 4 - 2).float()
    return torch.randint(0, e.n_actions, (n,), dtype=torch.int32, device="cuda")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--work", default="c1,c2,c3,pend,acro")
    ap.add_argument("--no-track", action="store_true")
    ap.add_argument("--spec", action="store_true", help="config-specialised kernels (nsg_specialize)")
    ap.add_argument("--exact", action="store_true", help="libm_exact=True: the NSG_F_LIBM_EXACT units (classic-control workloads)")
    ap.add_argument("--rollout", type=int, default=0, help="K fused steps per launch (nsg_rollout) instead of nsg_step")
    ap.add_argument("--resident", type=int, default=0, help="K steps through ONE resident launch (nsg_resident_start): closed loop with the library's "
                                                              "resident demo policy on a second stream, and open loop (action rows published in advance)")
    args = ap.parse_args()
    global EXACT
    EXACT = args.exact
    res = {}
    for name in args.work.split(","):
        n = args.n
        e = mk(name, n, track=not args.no_track and not args.resident, spec=args.spec)
        a = actions(e, n)
        for _ in range(30):
            e.step(a)
        torch.cuda.synchronize()
        if args.resident:
            from ns_gym_amd.vec_env import ResidentStepper

            K = args.resident
            loop = ResidentStepper(e, a.clone(), wait_budget_us=50_000)
            pol = torch.cuda.Stream()
            out = {}
            for mode in ("closed_loop", "open_loop"):
                best = 1e9
                for rep in range(4):          # (the first run pays the kernels' first launches)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record(loop.stream)
                    loop.start(K, prefilled=K if mode == "open_loop" else 0)
                    if mode == "closed_loop":
                        loop.demo_policy(K, stream=pol)
                    e1.record(loop.stream)
                    status, steps = loop.result()
                    assert (status, steps) == ("finished", K), (status, steps)
                    if rep:
                        best = min(best, e0.elapsed_time(e1) * 1e3 / K)
                out[mode + "_us_per_step"] = best
            out["Gsteps/s_closed_loop"] = n / (out["closed_loop_us_per_step"] * 1e-6) / 1e9
            out["what"] = (f"one nsg_resident_start launch, {K} steps; closed loop = the resident demo policy (one workgroup per chunk) on the other "
                           "side of the mailbox, observation -> action -> step; open loop = rows published in advance (no hand-over wait)")
            res[name] = out
            print(name, json.dumps(out), flush=True)
            e.close()
            continue
        if args.rollout:
            K = args.rollout
            acts = torch.stack([actions(e, n) for _ in range(K)])
            for _ in range(2):
                e.rollout(acts)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                reps = max(args.iters // K, 2)
                for _ in range(reps):
                    e.rollout(acts)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / (reps * K))
            ms = best
        else:
            ms = min(e.time_steps(a, args.iters) for _ in range(3))
        if args.rollout:
            # a fused rollout keeps the persistent rows in registers: per env-step it moves the action (4 B) and the recorded
            # outputs (obs + reward 4 + terminated 1 + truncated 1) only - NOT the step API's algorithmic bytes
            per = 4 + (4 if e.is_grid else 4 * e.obs_dim) + 6
            gbs = per * n / (ms * 1e-3) / 1e9
            res[name] = {"us_per_step": ms * 1e3, "Gsteps/s": n / (ms * 1e-3) / 1e9, "hbm_bytes_per_env_step": per, "GB/s": gbs,
                         "frac_of_8TBs": gbs / 8000, "bound": "instruction issue + the reset hand-over's barriers (DESIGN.md), not HBM"}
        else:
            gbs = WORK[name][3] * n / (ms * 1e-3) / 1e9
            res[name] = {"us": ms * 1e3, "algorithmic_bytes_per_env_step": WORK[name][3], "GB/s": gbs, "frac": gbs / 8000,
                         "Gsteps/s": n / (ms * 1e-3) / 1e9}
        print(name, json.dumps(res[name]), flush=True)
        e.close()
    if "pend" in args.work and "acro" in args.work:  # C4: heterogeneous launch, 2^18 each
        n = 1 << 18
        ep, ea = mk("pend", n, spec=args.spec), mk("acro", n, spec=args.spec)
        ap_, aa = actions(ep, n), actions(ea, n)
        for _ in range(20):
            step_group([ep, ea], [ap_, aa])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            step_group([ep, ea], [ap_, aa])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        gbs = (83 + 127) * n / (ms * 1e-3) / 1e9
        print("c4_group", json.dumps({"us": ms * 1e3, "GB/s": gbs, "frac": gbs / 8000, "Gsteps/s": 2 * n / (ms * 1e-3) / 1e9}), flush=True)
        if args.rollout:   # the same pair through ONE fused launch of K steps (nsg_rollout_group)
            K = args.rollout
            acts = [torch.stack([actions(e, n) for _ in range(K)]) for e in (ep, ea)]
            for _ in range(2):
                rollout_group([ep, ea], acts)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                reps = max(args.iters // K, 2)
                for _ in range(reps):
                    rollout_group([ep, ea], acts)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / (reps * K))
            print("c4_group_rollout", json.dumps({"us_per_step": best * 1e3, "Gsteps/s": 2 * n / (best * 1e-3) / 1e9,
                                                  "launch": "one nsg_rollout_group launch per K steps, both members"}), flush=True)


if __name__ == "__main__":
    main()
