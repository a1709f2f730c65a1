#!/usr/bin/env python3
"""C4 as two CONCURRENT kernels (Pendulum 2^18 and Acrobot 2^18 on two HIP streams, forked and joined inside one HIP graph:
one graph launch = one step of both batches) against the one nsg_step_group launch.  PAIRS pairs per graph keep the host out of
the measurement."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.kbench import mk, actions
from ns_gym_amd.vec_env import step_group

n = 1 << 18
PAIRS, REPS = 10, 40
ep, ea = mk("pend", n, spec=True), mk("acro", n, spec=True)
ap_, aa = actions(ep, n), actions(ea, n)
for _ in range(20):
    step_group([ep, ea], [ap_, aa])
torch.cuda.synchronize()


def run_graph(build):
    cap, side = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        build(cap, side)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (REPS * PAIRS) * 1e3


def b_group(cap, side):
    for _ in range(PAIRS):
        step_group([ep, ea], [ap_, aa])


def b_serial(cap, side):
    for _ in range(PAIRS):
        ea.step(aa)
        ep.step(ap_)


def b_joined(cap, side):      # fork, two kernels side by side, join - per pair
    for _ in range(PAIRS):
        f = torch.cuda.Event(); f.record(cap)
        side.wait_event(f)
        with torch.cuda.stream(side):
            ep.step(ap_)
            j = torch.cuda.Event(); j.record(side)
        ea.step(aa)
        cap.wait_event(j)


def b_free(cap, side):        # fork once, PAIRS kernels per stream, join once
    f = torch.cuda.Event(); f.record(cap)
    side.wait_event(f)
    with torch.cuda.stream(side):
        for _ in range(PAIRS):
            ep.step(ap_)
        j = torch.cuda.Event(); j.record(side)
    for _ in range(PAIRS):
        ea.step(aa)
    cap.wait_event(j)


for name, b in (("one nsg_step_group launch per pair", b_group), ("two launches, one stream", b_serial),
                ("two streams, joined per pair", b_joined), ("two streams, joined per 10 pairs", b_free)):
    print(f"{name:36s}: {run_graph(b):6.2f} us per pair", flush=True)
ep.close(); ea.close()
