#!/usr/bin/env python3
"""Where do the ~40 us go that a 20-step timed window (the driver's `bench.py --steps 20 --warmup 5`) carries beyond 20 kernel
durations?  Host time stamps around the same sequence bench.py runs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ns_gym_amd import workloads as W  # noqa: E402
from ns_gym_amd.distributed import global_actions  # noqa: E402

n = 1 << 20
env = W.build("c1", n, specialize=True)
pool = [global_actions(k, 0, n, 2, device="cuda") for k in range(8)]
for k in range(16):
    env.step(pool[k % 8])
ev = torch.cuda.Event()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def drain(e):
    while not e.query():
        pass
    torch.cuda.synchronize()


pc = time.perf_counter
for trial in range(6):
    for k in range(2):
        env.step(pool[k])
    ev.record(); drain(ev)
    t0 = pc(); e0.record(); t1 = pc()
    env.step(pool[0]); t2 = pc()
    for k in range(1, 20):
        env.step(pool[k % 8])
    t3 = pc(); e1.record(); t4 = pc()
    while not e1.query():
        pass
    t5 = pc(); torch.cuda.synchronize(); t6 = pc()
    kern = e0.elapsed_time(e1) * 1e3
    print(f"trial {trial}: total {(t6 - t0) * 1e6:7.1f} us | e0.record {(t1 - t0) * 1e6:5.1f} | first step() call {(t2 - t1) * 1e6:5.1f} | 19 more calls {(t3 - t2) * 1e6:6.1f} "
          f"| e1.record {(t4 - t3) * 1e6:5.1f} | polling until done {(t5 - t4) * 1e6:7.1f} | synchronize {(t6 - t5) * 1e6:5.1f} | events e0->e1 {kern:7.1f} us = {kern / 20:.2f} per step", flush=True)
