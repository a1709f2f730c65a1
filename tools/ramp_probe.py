#!/usr/bin/env python3
"""Launch time of nsg_step against how far the batch is from its reset.
  python tools/ramp_probe.py            blocks of 20 launches right after reset()
  python tools/ramp_probe.py ages       200-launch averages after 0, 512, 2048, ... pre-rolled steps (episode counts grow, so
                                        the reset path's jump-ahead walks more table digits)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.kbench import mk, actions
n = 1 << 20
e = mk("c1", n, spec=True)
a = actions(e, n)
if len(sys.argv) > 1 and sys.argv[1] == "ages":
    pre = torch.stack([a] * 64)
    for age in (0, 512, 2048, 8192, 32768, 131072):
        e.reset(seed=1)
        for _ in range(age // 64):
            e.rollout(pre, record=("reward",))
        torch.cuda.synchronize()
        ep = e.buf["episode"]
        cnt = (ep >> 1).to(torch.float64)
        print(f"age {age:7d} steps: mean episode count {cnt.mean().item():8.1f} max {int(cnt.max().item()):6d}   "
              + " ".join(f"{e.time_steps(a, 200) * 1e3:.2f}" for _ in range(3)) + " us per launch", flush=True)
else:
    for rep in range(2):
        e.reset(seed=rep)
        print("after reset:", " ".join(f"{e.time_steps(a, 20) * 1e3:.1f}" for _ in range(12)), "us per launch (blocks of 20)")
e.close()
