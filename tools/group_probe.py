#!/usr/bin/env python3
"""C4 heterogeneous launch probe (GPU box): one nsg_step_group launch over Pendulum 2^18 + Acrobot 2^18 against the two
handles stepped one after the other, specialised and generic; prints nsg_last_error() after the first group call (a failed
group-unit build falls back to the generic kernel silently)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.kbench import mk, actions
from ns_gym_amd.vec_env import step_group
from ns_gym_amd import _lib

n = 1 << 18
for spec in (True, False):
    ep, ea = mk("pend", n, spec=spec), mk("acro", n, spec=spec)
    ap_, aa = actions(ep, n), actions(ea, n)
    step_group([ep, ea], [ap_, aa])
    print("spec" if spec else "generic", "last_error after first group call:", repr(_lib.load().nsg_last_error()[:200]))
    for order, label in (([ep, ea], "pend,acro"), ([ea, ep], "acro,pend")):
        acts = [ap_, aa] if order[0] is ep else [aa, ap_]
        for _ in range(20):
            step_group(order, acts)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            step_group(order, acts)
        e1.record(); torch.cuda.synchronize()
        print(f"  group [{label}]: {e0.elapsed_time(e1) / 300 * 1e3:.2f} us")
    for _ in range(20):
        ep.step(ap_); ea.step(aa)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        ea.step(aa); ep.step(ap_)
    e1.record(); torch.cuda.synchronize()
    print(f"  separate launches (acro then pend): {e0.elapsed_time(e1) / 300 * 1e3:.2f} us; pend alone {ep.time_steps(ap_, 300) * 1e3:.2f}, acro alone {ea.time_steps(aa, 300) * 1e3:.2f}")
    ep.close(); ea.close()
