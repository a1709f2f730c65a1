#!/usr/bin/env python3
"""Launch-bound regime: (1) BASELINE C2 at its own size (65 536 envs): stream launches vs a captured HIP graph of the
same step loop vs the fused rollout; (2) BASELINE C4 (Pendulum + Acrobot, 2^18 each): one heterogeneous
`nsg_step_group` launch vs the two handles stepped concurrently on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.kbench import mk, actions
from ns_gym_amd.vec_env import step_group


def wall(f, reps):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


K = 64
for spec in (False, True):
    n = 1 << 16
    e = mk("c2", n, spec=spec)
    acts = torch.stack([actions(e, n) for _ in range(K)])
    for k in range(8):
        e.step(acts[k])
    def loop():
        for k in range(K):
            e.step(acts[k])
    t_stream = wall(loop, 20) / K
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        e.step(acts[0])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(K):
            e.step(acts[k])
    t_graph = wall(g.replay, 20) / K
    t_roll = wall(lambda: e.rollout(acts, record=("reward", "terminated", "truncated")), 20) / K
    print(f"C2 N=65536 spec={spec}: stream launches {t_stream*1e6:.2f} us/step | graph of {K} steps {t_graph*1e6:.2f} us/step | "
          f"nsg_rollout K={K} {t_roll*1e6:.2f} us/step", flush=True)
    e.close()

n = 1 << 18
for spec in (False, True):
    ep, ea = mk("pend", n, spec=spec), mk("acro", n, spec=spec)
    ap_, aa = actions(ep, n), actions(ea, n)
    t_group = wall(lambda: step_group([ep, ea], [ap_, aa]), 200)
    def serial():
        ea.step(aa); ep.step(ap_)
    t_serial = wall(serial, 200)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    def two_streams():
        with torch.cuda.stream(s1):
            ea.step(aa)
        with torch.cuda.stream(s2):
            ep.step(ap_)
    t_two = wall(two_streams, 200)
    print(f"C4 2^18+2^18 spec={spec}: nsg_step_group {t_group*1e6:.2f} us | two launches on one stream {t_serial*1e6:.2f} us | "
          f"on two streams {t_two*1e6:.2f} us", flush=True)
    ep.close(); ea.close()
