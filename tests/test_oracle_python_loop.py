"""The object-model Python loop used as the "reference-style" CPU figure (oracle/python_loop.py) against the C oracle:
same seeds, same actions, next-step autoreset - C1 (Increment/Continuous) and C2 (RandomWalk/Periodic(3))."""
import numpy as np
import pytest

from oracle import python_loop as PL
from oracle.oracle import OracleVecEnv


@pytest.mark.parametrize("cfg", ["c1", "c2"])
def test_python_object_loop_equals_the_c_oracle(cfg):
    from ns_gym_amd import make
    from ns_gym_amd.schedulers import ContinuousScheduler, PeriodicScheduler
    from ns_gym_amd.update_functions import IncrementUpdate, RandomWalk

    n, T = 6, 120
    tp = ({"masspole": IncrementUpdate(ContinuousScheduler(), k=0.1)} if cfg == "c1"
          else {"gravity": RandomWalk(PeriodicScheduler(period=3))})
    o = OracleVecEnv(make("CartPole-v1"), tp, n, change_notification=True, delta_change_notification=True)
    envs = [PL.make_c1() if cfg == "c1" else PL.make_c2() for _ in range(n)]
    seeds = np.arange(n, dtype=np.uint64) + np.uint64(40)
    o.reset(seed=seeds)
    obs = [e.reset(seed=int(s))[0] for e, s in zip(envs, seeds)]
    np.testing.assert_allclose(np.stack([x["state"] for x in obs]), o.a["obs"], rtol=1e-6, atol=1e-7)
    acts = np.random.default_rng(1).integers(2, size=(T, n)).astype(np.int32)
    done = [False] * n
    name = "masspole" if cfg == "c1" else "gravity"
    for k in range(T):
        o.step(acts[k])
        for i, e in enumerate(envs):
            if done[i]:
                ob, _ = e.reset()
                term = trunc = False
            else:
                ob, r, term, trunc, info = e.step(int(acts[k, i]))
                assert info["Ground Truth Env Change"][name] == o.a["env_change"][0, i], (k, i)
            done[i] = term or trunc
            np.testing.assert_allclose(ob["state"], o.a["obs"][i], rtol=1e-5, atol=1e-5, err_msg=f"step {k} env {i}")
            assert ob["relative_time"] == o.a["t"][i] and term == bool(o.a["terminated"][i]) and trunc == bool(o.a["truncated"][i])
            np.testing.assert_allclose(getattr(e.unwrapped, name), o.a["theta"][0, i], rtol=1e-13)
