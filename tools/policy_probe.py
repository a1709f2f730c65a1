#!/usr/bin/env python3
"""Fused policy rollouts (nsg_rollout_policy) on the GPU box: what a closed loop costs per env-step when the decision does not
leave the kernel, against the ways the same loop ran before - and the planner-shaped workload (MCTS.py:131,162-181) built on it.

  1. kernel level, per BASELINE config at its own size: K = 64 fused steps per launch, UNIFORM in-kernel actions, accounts only
     (nothing recorded) vs nsg_rollout over an action table with reward / terminated / truncated recorded (what the harness used
     to launch) vs nsg_step;
  2. closed loops: C2 at 65 536 envs with a linear policy (in-kernel) vs step() + the same policy as torch kernels; C3 at 2^20 with
     a tabular policy (in-kernel) vs step() + a torch gather;
  3. the planner shape, README quickstart d = 50, m = 100: R roots x 100 simulations x 50 steps per decision:
     planning.Simulator (fork into a standing batch + one fused launch) vs fork + torch.randint table + nsg_rollout + torch reductions.
Prints one JSON object."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ns_gym_amd import workloads as W  # noqa: E402
from ns_gym_amd.planning import Simulator  # noqa: E402
from ns_gym_amd.policies import EpisodeAccounts, LinearPolicy, TabularPolicy, UniformRandom  # noqa: E402


def timed(fn, reps=5, inner=1):
    """Best-of device time per call of fn (ms), HIP events on torch's current stream (the library launches there)."""
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / inner)
    return best


def rand_actions(e, k):
    if e.action_is_float:
        return (torch.rand((k, e.num_envs), device="cuda") * 4 - 2).float()
    return torch.randint(0, e.n_actions, (k, e.num_envs), dtype=torch.int32, device="cuda")


def kernel_level(res, only=None):
    K = 64
    for name, n in (("c1", 1 << 20), ("c2", 1 << 16), ("c3", 1 << 20), ("pend", 1 << 18), ("acro", 1 << 18)):
        if only and name not in only:
            continue
        e = W.build(name, n, specialize=True)
        e.reset(seed=0)
        acts = rand_actions(e, K)
        for _ in range(3):
            e.rollout(acts, record=("reward", "terminated", "truncated"))
        pol = UniformRandom(seed=1)
        acc = EpisodeAccounts(e, gamma=0.99, horizon=K + 1)
        for _ in range(3):
            e.rollout_policy(pol, K, accounts=acc)
        step = e.time_steps(acts[0], 200) * 1e3
        t_tab = timed(lambda: e.rollout(acts, record=("reward", "terminated", "truncated")), inner=4) * 1e3 / K
        t_pol = timed(lambda: e.rollout_policy(pol, K, accounts=acc.restart()), inner=4) * 1e3 / K
        t_pol_rec = timed(lambda: e.rollout_policy(pol, K, record=("reward", "terminated", "truncated"), accounts=acc.restart()), inner=4) * 1e3 / K
        res["kernel_level"][name] = {"envs": n, "nsg_step_us": round(step, 2), "rollout_table_recorded_us_per_step": round(t_tab, 3),
                                     "rollout_policy_uniform_accounts_only_us_per_step": round(t_pol, 3),
                                     "rollout_policy_uniform_recorded_us_per_step": round(t_pol_rec, 3),
                                     "Gsteps_per_s_policy": round(n / t_pol / 1e3, 2), "kernels": e.policy_kernels}
        print(name, json.dumps(res["kernel_level"][name]), file=sys.stderr, flush=True)
        e.close()


def closed_loops(res):
    K = 64
    # C2 at BASELINE's own size with a linear policy on the observation
    n = 1 << 16
    e = W.build("c2", n, specialize=True)
    e.reset(seed=0)
    Wm = np.array([[0.3, -0.8, -2.0, -1.1, 0.05], [-0.3, 0.8, 2.0, 1.1, -0.05]], dtype=np.float32)
    pol = LinearPolicy(Wm)
    acc = EpisodeAccounts(e, gamma=None)
    for _ in range(3):
        e.rollout_policy(pol, K, accounts=acc)
    fused = timed(lambda: e.rollout_policy(pol, K, accounts=acc), inner=8) * 1e3 / K
    Wt = torch.from_numpy(Wm).cuda()

    def torch_loop():
        s = e.state
        for _ in range(K):
            a = torch.argmax(s @ Wt[:, :4].T + Wt[:, 4], dim=1).to(torch.int32)
            e._step_raw(a.data_ptr())
            s = e.state

    for _ in range(2):
        torch_loop()
    py = timed(torch_loop, reps=3) * 1e3 / K
    res["closed_loop"]["c2_65536_linear_policy"] = {"fused_us_per_step": round(fused, 3), "step_plus_torch_policy_us_per_step": round(py, 2),
                                                    "Gsteps_per_s_fused": round(n / fused / 1e3, 2)}
    print(json.dumps(res["closed_loop"]["c2_65536_linear_policy"]), file=sys.stderr, flush=True)
    e.close()
    # C3 at its own size with a tabular policy (tutorial.ipynb cell 12)
    n = 1 << 20
    e = W.build("c3", n, specialize=True)
    e.reset(seed=0)
    tab = TabularPolicy(np.random.default_rng(0).integers(0, 4, size=64))
    acc = EpisodeAccounts(e, gamma=None)
    for _ in range(3):
        e.rollout_policy(tab, K, accounts=acc)
    fused = timed(lambda: e.rollout_policy(tab, K, accounts=acc), inner=4) * 1e3 / K
    tt = tab._dev_table(e.device)

    def torch_loop3():
        for _ in range(K):
            a = tt[e.state.long()]
            e._step_raw(a.data_ptr())

    for _ in range(2):
        torch_loop3()
    py = timed(torch_loop3, reps=3) * 1e3 / K
    res["closed_loop"]["c3_2p20_tabular_policy"] = {"fused_us_per_step": round(fused, 3), "step_plus_torch_gather_us_per_step": round(py, 2),
                                                   "Gsteps_per_s_fused": round(n / fused / 1e3, 2)}
    print(json.dumps(res["closed_loop"]["c3_2p20_tabular_policy"]), file=sys.stderr, flush=True)
    e.close()


def planner(res):
    S, d, gamma = 100, 50, 0.99          # README quickstart: MCTS(d=50, m=100)
    for name, R in (("c1", 4096), ("c3", 4096)):
        e = W.build(name, R, specialize=True)
        e.reset(seed=0)
        a0 = rand_actions(e, 1)[0]
        for _ in range(5):
            e.step(a0)
        plan = e.get_planning_env()
        sim = Simulator(plan, sims=S, depth=d, gamma=gamma)
        first = torch.randint(0, e.n_actions, (S, R), dtype=torch.int32, device="cuda")
        for k in range(3):
            sim.run(seed=k, first_actions=first)
        t_new = timed(lambda: sim.run(seed=7, first_actions=first), reps=4)
        # the same decision the way it ran before: action table from torch, nsg_rollout with the flags recorded, torch reductions
        big = sim.copies
        disc = torch.tensor([gamma ** j for j in range(d)], dtype=torch.float64, device="cuda").view(d, 1)

        def old():
            plan.fork(theta_mode=0, into=big)
            _, r, te, tr, _ = big.step(first.reshape(-1))
            alive = ~(te | tr)
            acts = torch.randint(0, e.n_actions, (d, S * R), dtype=torch.int32, device="cuda")
            out = big.rollout(acts, record=("reward", "terminated", "truncated"))
            done = out["terminated"] | out["truncated"]
            before = torch.cumsum(done.to(torch.int32), dim=0) - done.to(torch.int32)
            live = alive.unsqueeze(0) & (before == 0)
            return (out["reward"].to(torch.float64) * disc * live).sum(dim=0)

        for _ in range(2):
            old()
        t_old = timed(old, reps=4)
        steps = S * R * (d + 1)
        res["planner"][name] = {"roots": R, "sims": S, "depth": d, "copies": S * R,
                                "fused_ms_per_decision": round(t_new, 3), "fused_G_sim_steps_per_s": round(steps / t_new / 1e6, 1),
                                "table_rollout_plus_torch_ms_per_decision": round(t_old, 3), "table_G_sim_steps_per_s": round(steps / t_old / 1e6, 1)}
        print(name, json.dumps(res["planner"][name]), file=sys.stderr, flush=True)
        sim.close(); plan.close(); e.close()


def main():
    res = {"kernel_level": {}, "closed_loop": {}, "planner": {},
           "what": "K = 64 fused steps per launch; times are HIP-event device times, best of 3-5; policy units: config-specialised (hiprtc on first use)"}
    which = sys.argv[1:] or ["kernel", "closed", "planner"]      # "kernel:c1,acro" restricts the kernel-level part (counter passes)
    t0 = time.time()
    for w in which:
        if w.startswith("kernel"):
            kernel_level(res, only=w.split(":")[1].split(",") if ":" in w else None)
    if "closed" in which:
        closed_loops(res)
    if "planner" in which:
        planner(res)
    res["wall_s"] = round(time.time() - t0, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
