#!/usr/bin/env python3
"""Closed loop at the launch-bound size (BASELINE C2 at 65 536 envs): policy -> step -> policy.
  launches:   a one-kernel policy + nsg_step per step on one stream (what a closed-loop caller pays today), and as a HIP graph
  resident:   nsg_resident_start with the library's resident demo policy on a second stream (per-chunk hand-shake)
  open loop:  the resident stepper fed from ONE action row published in advance: what its steps cost without any hand-over wait
us per step each; the resident loop's rows are checked against the launch loop.   tools/resident_probe.py [work] [n_envs] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ns_gym_amd import workloads as W
from ns_gym_amd.vec_env import ResidentStepper

work = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
res = {"work": work, "envs": n, "steps": K}
ROWS = ("phys", "theta", "t", "episode", "obs", "reward", "terminated", "truncated", "rng_upd", "env_change", "delta_change")


@torch.compile(disable=True)
def policy(env, k, out):     # the demo policy's rule, as (few) torch kernels
    torch.remainder((env.state[:, 2] > 0).to(torch.int32) + k, 2, out=out)


def launch_loop(env, a, steps):
    for k in range(steps):
        policy(env, k, a); env.step(a)


for spec in (False, True):
    tag = "spec" if spec else "generic"
    ref = W.build(work, n, specialize=spec, seed=3, track_returns=False)
    a = torch.zeros(n, dtype=torch.int32, device="cuda")
    launch_loop(ref, a, 20)
    ref.reset(seed=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    launch_loop(ref, a, K)
    torch.cuda.synchronize()
    res[f"{tag}:policy(4 torch kernels)+nsg_step_us"] = (time.perf_counter() - t0) / K * 1e6
    # the same loop as a HIP graph of 64 iterations (5 kernels each): no host in the loop
    ref.reset(seed=3)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        launch_loop(ref, a, 2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch_loop(ref, a, 64)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(K // 64, 1)):
        g.replay()
    torch.cuda.synchronize()
    res[f"{tag}:policy(4 torch kernels)+nsg_step_as_a_graph_us"] = (time.perf_counter() - t0) / (max(K // 64, 1) * 64) * 1e6
    # the step alone, back to back (no policy): the floor of any launch-per-step loop
    t0 = time.perf_counter()
    for k in range(K):
        ref.step(a)
    torch.cuda.synchronize()
    res[f"{tag}:nsg_step_alone_us"] = (time.perf_counter() - t0) / K * 1e6
    ref.reset(seed=3)
    launch_loop(ref, a, K)
    torch.cuda.synchronize()
    env = W.build(work, n, specialize=spec, seed=3, track_returns=False)
    loop = ResidentStepper(env, torch.zeros(n, dtype=torch.int32, device="cuda"), wait_budget_us=20000)
    sa, sb = loop.stream, torch.cuda.Stream()     # the stepper on the loop's high-priority stream, the producer on an ordinary one
    for rep in range(2):     # the first run pays the kernels' first launches
        env.reset(seed=3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(sa)
        loop.start(K, stream=sa)
        loop.demo_policy(K, stream=sb)
        e1.record(sa)
        status, steps = loop.result()
        torch.cuda.synchronize()
        res[f"{tag}:resident:closed_loop_run{rep}"] = {"status": status, "steps_done": steps, "us_per_step": e0.elapsed_time(e1) * 1e3 / max(steps, 1)}
    res[f"{tag}:resident:rows_equal_to_the_launch_loop"] = all(bool(torch.equal(env.buf[r], ref.buf[r])) for r in ROWS if env.buf[r] is not None)
    # open loop: every action row "published" in advance (the same row each step): no hand-over wait at all
    for rep in range(2):
        env.reset(seed=3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(sa)
        loop.start(K, stream=sa, prefilled=K)
        e1.record(sa)
        status, steps = loop.result()
        res[f"{tag}:resident:open_loop_run{rep}"] = {"status": status, "steps_done": steps, "us_per_step": e0.elapsed_time(e1) * 1e3 / max(steps, 1)}
    # starvation: nobody publishes - the kernel must leave inside its budget (2 ms) + grace (0.2 ms)
    loop2 = ResidentStepper(env, torch.zeros(n, dtype=torch.int32, device="cuda"), wait_budget_us=2000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop2.start(50, stream=sa)
    status, steps = loop2.result()
    res[f"{tag}:resident:starved"] = {"status": status, "steps_done": steps, "host_ms_until_it_left": (time.perf_counter() - t0) * 1e3}
    env.close(); ref.close()
print(json.dumps(res, indent=1))
