#!/usr/bin/env python3
"""Closed loop at the launch-bound size (BASELINE C2 at 65 536 envs): policy -> step -> policy, three ways:
  (1) two launches per step on one stream (a trivial policy kernel + nsg_step): what a closed-loop caller pays today;
  (2) the same loop captured in a HIP graph;
  (3) the resident stepper (nsg_resident_start) with the library's resident demo policy on a second stream.
us per step each; rows of (3) are checked against (1).   tools/resident_probe.py [work] [n_envs] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ns_gym_amd import workloads as W
from ns_gym_amd.vec_env import ResidentStepper

work = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
res = {"work": work, "envs": n, "steps": K}


def policy(env, k, out):
    torch.remainder((env.state[:, 2] > 0).to(torch.int32) + k, 2, out=out)


for spec in (False, True):
    tag = "spec" if spec else "generic"
    ref = W.build(work, n, specialize=spec, seed=3, track_returns=False)
    a = torch.zeros(n, dtype=torch.int32, device="cuda")
    for k in range(20):
        policy(ref, k, a); ref.step(a)
    ref.reset(seed=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        policy(ref, k, a); ref.step(a)
    torch.cuda.synchronize()
    res[f"{tag}:launches_per_step_us"] = (time.perf_counter() - t0) / K * 1e6
    if not spec:
        env = W.build(work, n, specialize=False, seed=3, track_returns=False)
        loop = ResidentStepper(env, torch.zeros(n, dtype=torch.int32, device="cuda"), wait_budget_us=5000)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        loop.mailbox.zero_()
        with torch.cuda.stream(sa):
            pass
        loop.start(K, stream=sa)
        e0.record(sa) if False else None
        loop.demo_policy(K, stream=sb)
        t0 = time.perf_counter()
        status, steps = loop.result()
        dt = time.perf_counter() - t0
        res["resident:status"] = status
        res["resident:steps_done"] = steps
        res["resident:us_per_step_host_clock_incl_launch"] = dt / max(steps, 1) * 1e6
        same = all(torch.equal(env.buf[r], ref.buf[r]) for r in ("phys", "theta", "t", "episode", "obs", "reward", "terminated", "truncated", "rng_upd")
                   if env.buf[r] is not None)
        res["resident:rows_equal_to_the_launch_loop"] = bool(same) if not spec else None
        # again, timed with events on the stepper's stream (second run: everything warm)
        env.reset(seed=3)
        torch.cuda.synchronize()
        e0.record(sa)
        loop.start(K, stream=sa)
        loop.demo_policy(K, stream=sb)
        e1.record(sa)
        status, steps = loop.result()
        res["resident:us_per_step_events"] = e0.elapsed_time(e1) * 1e3 / max(steps, 1)
        res["resident:status2"] = status
        env.close()
    ref.close()
print(json.dumps(res, indent=1))
