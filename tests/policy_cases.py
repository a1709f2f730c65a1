"""Shared runners for the closed-loop fixtures (tests/golden/policy_*.npz, made by tests/golden/make_policy_rollouts.py from the
reference's own `MCTS._default_policy` and `run_episode`): the same sequence of calls is driven through the C oracle (CPU tests)
and through the HIP library (GPU tests) behind two thin adapters."""
import json
import os

import numpy as np

from tests.util import GOLDEN, build_params, load

with open(os.path.join(GOLDEN, "policy_rollouts.json")) as _f:
    POLICY_MANIFEST = json.load(_f)
MCTS_CASES = POLICY_MANIFEST["mcts"]
EPISODE_CASES = POLICY_MANIFEST["episode"]


def _make(factory, spec, n, **extra):
    from ns_gym_amd.envs import make

    env = make(spec["env_id"], **spec.get("make_kwargs", {}))
    return factory(env, build_params(spec["params"]), n, **spec["flags"], **spec.get("wrapper_kwargs", {}), **extra)


class OracleSide:
    """oracle.OracleVecEnv behind the calls the runners make."""

    def __init__(self, spec, n):
        from oracle.oracle import OracleVecEnv

        self.env = _make(OracleVecEnv, spec, n)

    def reset(self, seeds):
        self.env.reset(seed=np.asarray(seeds, dtype=np.uint64))

    def step(self, actions):
        self.env.step(actions)

    def fork(self, theta_mode):
        o = OracleSide.__new__(OracleSide)
        o.env = self.env.fork(theta_mode=theta_mode, entropy=4321)
        return o

    def seed_env_streams(self, seeds):
        self.env.seed_streams(np.asarray(seeds, dtype=np.uint64), 0)

    def _accounts(self, gamma, horizon):
        n = self.env.N
        disc = None if gamma is None else np.array([float(gamma) ** j for j in range(horizon)], dtype=np.float64)
        return {"ret": np.zeros(n), "length": np.zeros(n, dtype=np.int32), "alive": np.ones(n, dtype=np.uint8), "discount": disc}

    def table_rollout(self, actions, gamma, horizon):
        from ns_gym_amd import _abi as A

        acc = self._accounts(gamma, horizon)
        dt = np.float32 if self.env.action_is_float else np.int32
        self.env.rollout_policy(A.NSG_POL_TABLE, actions.shape[0], data=np.ascontiguousarray(actions, dtype=dt), accounts=acc)
        return acc["ret"], acc["length"], acc["alive"]

    def linear_rollout(self, W, k_steps, gamma=None):
        from ns_gym_amd import _abi as A

        acc = self._accounts(gamma, k_steps + 1)
        acts, _, took = self.env.rollout_policy(A.NSG_POL_LINEAR, k_steps, data=np.ascontiguousarray(W, dtype=np.float32), accounts=acc)
        return acc["ret"], acc["length"], acc["alive"], acts


class HipSide:
    """ns_gym_amd.VecNSEnv behind the same calls (everything through the C-ABI)."""

    def __init__(self, spec, n, specialize=False, libm_exact=False):
        from ns_gym_amd.vec_env import VecNSEnv

        kw = {"libm_exact": True} if libm_exact and spec["env_id"] not in ("FrozenLake-v1", "CliffWalking-v1") else {}
        self.env = _make(VecNSEnv, spec, n, specialize=specialize, **kw)

    def reset(self, seeds):
        self.env.reset(seed=np.asarray(seeds, dtype=np.uint64))

    def step(self, actions):
        import torch

        self.env.step(torch.from_numpy(np.ascontiguousarray(actions)).to(self.env.device))

    def fork(self, theta_mode):
        o = HipSide.__new__(HipSide)
        o.env = self.env.fork(theta_mode=theta_mode, entropy=4321)
        return o

    def seed_env_streams(self, seeds):
        self.env.seed_streams(np.asarray(seeds, dtype=np.uint64), which="env")

    def table_rollout(self, actions, gamma, horizon):
        import torch

        from ns_gym_amd import _abi as A
        from ns_gym_amd.policies import EpisodeAccounts, Policy

        dt = torch.float32 if self.env.action_is_float else torch.int32
        tab = torch.from_numpy(np.ascontiguousarray(actions)).to(self.env.device, dtype=dt).contiguous()

        class Table(Policy):
            kind = A.NSG_POL_TABLE

            def _data(self, env):
                return tab

        acc = EpisodeAccounts(self.env, gamma=gamma, horizon=horizon)
        self.env.rollout_policy(Table(), int(tab.shape[0]), accounts=acc)
        return acc.ret.cpu().numpy(), acc.length.cpu().numpy(), acc.alive.cpu().numpy()

    def linear_rollout(self, W, k_steps, gamma=None):
        from ns_gym_amd.policies import EpisodeAccounts, LinearPolicy

        acc = EpisodeAccounts(self.env, gamma=gamma, horizon=k_steps + 1)
        out = self.env.rollout_policy(LinearPolicy(W), k_steps, accounts=acc, record_actions=True)
        return acc.ret.cpu().numpy(), acc.length.cpu().numpy(), acc.alive.cpu().numpy(), out["actions"].cpu().numpy()


def run_mcts_case(side_cls, name, strict=False, **kw):
    """MCTS.search's simulation set-up (MCTS.py:131) and MCTS._default_policy (MCTS.py:162-181) for all fixture envs at once:
    reset(seed), the warm-up steps, get_planning_env(), deepcopy, then d steps of the recorded actions with the discounted-return
    account.  The reference's `tot_reward` and the number of steps its loop took must come out bit for bit."""
    spec, rec = MCTS_CASES[name], load(f"policy_{name}.npz")
    n, d, gamma = spec["n"], spec["d"], spec["gamma"]
    real = side_cls(spec, n, **kw)
    real.reset(rec["seeds"])
    for k in range(spec["pre"]):
        real.step(rec["pre_actions"][k])
    plan = real.fork(0 if spec["flags"].get("delta_change_notification") else 1)    # get_planning_env(): classic_control.py:120-136
    sim = plan.fork(0)                                                               # deepcopy(self.env): MCTS.py:131
    if spec["env_id"] in ("FrozenLake-v1", "CliffWalking-v1"):
        sim.seed_env_streams(rec["sim_env_seeds"])
    ret, length, alive = sim.table_rollout(rec["actions"], gamma, d + 1)
    np.testing.assert_array_equal(length, rec["steps"])
    exact = strict or spec["env_id"] not in ("Pendulum-v1",)     # (strict: the oracle, the kernels' libm_exact units)
    if exact:
        np.testing.assert_array_equal(ret, rec["ret"])
    else:
        np.testing.assert_allclose(ret, rec["ret"], rtol=1e-7)
    # a loop that stopped short of depth d stopped because its episode ended
    assert not alive[rec["steps"] < d].any()
    return ret, length, alive


def run_episode_case(side_cls, name, strict=False, **kw):
    """run_episode (run_experiment.py:91-148) for all fixture envs at once: reset(seed = base + i), then the closed loop with the
    linear agent until done / truncated (max_steps + 1 steps at most); `sum(episode_reward)`, `num_steps` and every action the
    reference's agent chose must come out."""
    spec, rec = EPISODE_CASES[name], load(f"policy_{name}.npz")
    n = spec["n"]
    side = side_cls(spec, n, **kw)
    side.reset(spec["seed"] + np.arange(n, dtype=np.uint64))
    K = spec["max_steps"] + 1
    ret, length, alive, acts = side.linear_rollout(np.asarray(spec["weights"], dtype=np.float32), K, gamma=None)
    np.testing.assert_array_equal(length, rec["num_steps"])
    T = rec["actions"].shape[0]
    for i in range(n):
        if spec["env_id"] == "Pendulum-v1" and not strict:   # (its observation goes through cos / sin: last-ulp differences between libm and the default kernels' sincos)
            np.testing.assert_allclose(acts[:length[i], i], rec["actions"][:length[i], i], rtol=0, atol=2e-5, err_msg=f"env {i}")
        else:
            np.testing.assert_array_equal(acts[:length[i], i], rec["actions"][:length[i], i], err_msg=f"env {i}")
    if spec["env_id"] == "Pendulum-v1" and not strict:
        np.testing.assert_allclose(ret, rec["total_reward"], rtol=1e-6)
    else:
        np.testing.assert_array_equal(ret, rec["total_reward"])
    assert T == int(rec["num_steps"].max())
    return ret, length
