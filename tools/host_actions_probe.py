#!/usr/bin/env python3
"""C1 at 2^20 envs when the ACTIONS live in host memory (the boundary handing over host buffers): a host-to-device copy per step,
or the kernel reading the pinned host buffer directly.  Never the bench's `value` - that has its inputs resident in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.kbench import mk

n = 1 << 20
e = mk("c1", n, spec=True)
host = [torch.randint(0, 2, (n,), dtype=torch.int32).pin_memory() for _ in range(8)]
dev = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(2)]
resident = [h.cuda() for h in host]


def run(body, iters=400):
    for k in range(40):
        body(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(iters):
        body(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def copy_then_step(k):
    d = dev[k & 1]
    d.copy_(host[k & 7], non_blocking=True)
    e.step(d)


def zero_copy(k):
    e._step_raw(host[k & 7].data_ptr())


for name, f in (("actions resident in HBM", lambda k: e.step(resident[k & 7])), ("H2D copy of 4 MiB per step, then step", copy_then_step),
                ("kernel reads the pinned host buffer itself", zero_copy)):
    us = run(f)
    print(f"{name:44s}: {us:7.1f} us per step = {n / us / 1e3:6.1f} G env-steps/s", flush=True)
e.close()
