import sys, time, torch
sys.path.insert(0, '/root/repo')
sys.path.insert(0, '.')
from tools.kbench import mk, actions
for name, n in (("c2", 65536), ("pend", 262144), ("acro", 262144), ("c3", 1 << 20)):
    for spec in (False, True):
        e = mk(name, n, spec=spec)
        a = actions(e, n)
        for _ in range(50): e.step(a)
        torch.cuda.synchronize()
        dev_ms = min(e.time_steps(a, 500) for _ in range(3))
        t0 = time.perf_counter()
        for _ in range(2000): e.step(a)
        torch.cuda.synchronize()
        host = (time.perf_counter() - t0) / 2000
        K = 64
        acts = torch.stack([actions(e, n) for _ in range(K)])
        e.rollout(acts); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): e.rollout(acts, record=("obs", "reward", "terminated", "truncated", "env_change", "delta_change"))
        torch.cuda.synchronize()
        ro = (time.perf_counter() - t0) / (20 * K)
        print(f"{name} N={n} spec={spec}: kernel {dev_ms*1e3:.2f} us/step ({n/dev_ms/1e6:.2f} G/s) | python step() loop {host*1e6:.2f} us/step ({n/host/1e9:.2f} G/s) | rollout K=64 {ro*1e6:.2f} us/step ({n/ro/1e9:.2f} G/s)")
        e.close()
