#!/usr/bin/env python3
"""Long-horizon parity of a CLOSED loop: a BASELINE config under an in-kernel linear / tabular policy (nsg_rollout_policy) for thousands
of steps on the GPU, and the oracle's restatement of the same loop (orc_rollout_policy) on the host; after every chunk all envs'
rows and episode accounts are compared.  A closed loop is the hardest case for parity: one differing observation bit can flip a
decision, after which the two trajectories have nothing to do with each other - so every env that left the bar is counted and shown.

    python tests/soak/soak_policy.py c2 65536 4000 500        # workload, envs, steps, chunk
    python tests/soak/soak_policy.py c2 65536 4000 500 exact  # the same through the NSG_F_LIBM_EXACT unit (libm's sin / cos / exp, bit for bit)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from ns_gym_amd import _abi as A
from ns_gym_amd import make, workloads as W
from ns_gym_amd.policies import EpisodeAccounts, LinearPolicy, TabularPolicy
from oracle.oracle import OracleVecEnv


def main():
    name, n, T, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    w = W.WORKLOADS[name]
    exact = len(sys.argv) > 5 and sys.argv[5] == "exact"
    env = W.build(name, n, track_returns=False, specialize=True, seed=None, **({"libm_exact": True} if exact else {}))
    print(f"{name}: {n} envs, {T} steps, libm_exact={env.libm_exact}", flush=True)
    orc = OracleVecEnv(make(w["env_id"], **w["make_kwargs"]), w["params"](), n, change_notification=True, delta_change_notification=True,
                       **w["wrapper_kwargs"])
    env.reset(seed=11)
    orc.reset(seed=11)
    if env.is_grid:
        table = np.random.default_rng(0).integers(0, 4, size=env.cfg.nrow * env.cfg.ncol).astype(np.int32)
        pol, okind, odata = TabularPolicy(table), A.NSG_POL_BY_STATE, table
    else:
        Wm = np.array([[0.3, -0.8, -2.0, -1.1, 0.05], [-0.3, 0.8, 2.0, 1.1, -0.05]], dtype=np.float32)
        pol, okind, odata = LinearPolicy(Wm), A.NSG_POL_LINEAR, Wm
    acc = EpisodeAccounts(env, gamma=None)
    oacc = {"ret": np.zeros(n), "length": np.zeros(n, dtype=np.int32), "alive": np.ones(n, dtype=np.uint8), "discount": None}
    t0 = time.time()
    bad_total = np.zeros(n, dtype=bool)
    needs_reset_start = np.zeros(n, dtype=bool)
    ages = []          # episode age (wrapper time t) at which an env's actions FIRST differed from the oracle's
    for k0 in range(0, T, K):
        t_start = env.t.cpu().numpy().copy()
        out = env.rollout_policy(pol, K, record=("terminated", "truncated"), accounts=acc, step0=k0, record_actions=True)
        # accounts restart every chunk on both sides, so that they keep counting episodes that start later
        oacts, _, _ = orc.rollout_policy(okind, K, data=odata, step0=k0, accounts=oacc)
        diff = out["actions"].cpu().numpy() != oacts                       # [K, n]
        fresh = diff.any(axis=0) & ~bad_total
        if fresh.any():
            done = (out["terminated"] | out["truncated"]).cpu().numpy()
            pending = needs_reset_start
            for i in np.nonzero(fresh)[0]:
                t, reset_next = int(t_start[i]), bool(pending[i])
                for k in range(K):
                    if reset_next:          # this call performs the pending autoreset: no action is taken
                        t, reset_next = 0, False
                        continue
                    if diff[k, i]:
                        ages.append(t)
                        break
                    t += 1
                    reset_next = bool(done[k, i])
        needs_reset_start = (out["terminated"][-1] | out["truncated"][-1]).cpu().numpy()
        g_state = env.state.cpu().numpy().reshape(n, -1)
        o_state = (orc.a["cell"].reshape(n, 1) if env.is_grid else orc.a["obs"].reshape(n, -1))
        same_t = env.t.cpu().numpy() == orc.a["t"]
        if env.is_grid:
            same_s = (g_state == o_state).all(axis=1)
        else:
            same_s = (np.abs(g_state - o_state) <= 1e-5 * np.maximum(1.0, np.abs(o_state))).all(axis=1)
        same_th = np.allclose(env.theta.cpu().numpy(), orc.a["theta"][:env.theta.shape[0]], rtol=1e-12, atol=0) if same_t.all() else None
        same_acc = (acc.length.cpu().numpy() == oacc["length"]) & (acc.ret.cpu().numpy() == oacc["ret"]) & (acc.alive.cpu().numpy() == oacc["alive"])
        bad = ~(same_t & same_s & same_acc)
        bad_total |= bad
        print(f"step {k0 + K}: envs off the bar {int(bad.sum())} (t {int((~same_t).sum())}, state {int((~same_s).sum())}, accounts {int((~same_acc).sum())}); "
              f"theta equal: {same_th}; episodes so far (GPU counters) {env.counters()['episodes']:,}; {time.time() - t0:.0f} s", flush=True)
        acc.restart(); oacc["ret"][:] = 0; oacc["length"][:] = 0; oacc["alive"][:] = 1
    print(f"{name}: {n} envs x {T} closed-loop steps ({n * T / 1e6:.0f} M decisions): {int(bad_total.sum())} envs ever left the bar"
          + ("" if not bad_total.any() else f" (first: {np.nonzero(bad_total)[0][:8].tolist()})"))
    if ages:
        a = np.array(ages)
        print(f"episode age when an env's actions first differed from the oracle's: min {a.min()}, median {int(np.median(a))}, max {a.max()} "
              f"({len(a)} envs; an unstable plant under equal actions amplifies a last-ulp state difference by e^(lambda tau) per step)")


if __name__ == "__main__":
    main()
