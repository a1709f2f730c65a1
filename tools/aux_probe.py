#!/usr/bin/env python3
"""Device time of the kernels around the hot path at N = 2^20 (C1 config): reset(seed), reset(), fork(into=),
done-index compaction, seed_streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools.kbench import mk, actions

n = 1 << 20
e = mk("c1", n, spec=True)
a = actions(e, n)
for _ in range(20):
    e.step(a)
copy = e.fork()
seeds = np.arange(n, dtype=np.uint64)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("reset(seed=array)   %.1f us (incl. the H2D copy of the seeds)" % timed(lambda: e.reset(seed=seeds)))
print("reset()             %.1f us" % timed(lambda: e.reset()))
for _ in range(5):
    e.step(a)
print("fork(into=)         %.1f us" % timed(lambda: e.fork(into=copy)))
print("seed_streams(env)   %.1f us" % timed(lambda: e.seed_streams(7, "env")))
e.step(a)
print("done_indices()      %.1f us (incl. the count read-back)" % timed(lambda: e.done_indices()))
print("counters()          %.1f us (sum over 16384 shards + read-back)" % timed(lambda: e.counters()))
